#!/usr/bin/env python
"""wav -> log-mel (1024/256, 80 mel): fused psnd_logmel_fwd vs psnd_stft_fwd + psnd_mel_fwd; end-to-end algorithmic
bytes 4*N*T + 4*N*M*F (1344 B per frame)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd import kernels as K
from oracle import features as ofe
dev = torch.device('cuda:0')
n, h, M, T = 1024, 256, 80, 44100
plan = K.stft_plan(n, ofe.analysis_window(n)).to(dev)
mplan = K.mel_plan(ofe.mel_filterbank(22050, n, M, 0.0, 8000.0)).to(dev)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e-3)
    return best
for N in (4, 32, 256, 1024, 2048):
    wav = torch.randn(N, T, device=dev) * 0.07
    F = K.frame_count(T, n, h)
    t1 = timeit(lambda: K.logmel_forward(wav, n, h, plan, mplan, M, 0, 0.0, K.LOG_E, 1e-6, None, -11.5, 6.9))
    def two():
        mag = K.stft_forward(wav, n, h, plan)['mag']
        return K.mel_forward(mag, mplan, M, K.LOG_E, 1e-6, None, -11.5, 6.9)
    t2 = timeit(two)
    b = 4 * N * T + 4 * N * M * F
    print('N=%4d: fused %.1f us (%.0f GB/s end-to-end, %.2f Mframes/s) | stft+mel %.1f us | x%.2f' % (N, t1 * 1e6, b / t1 / 1e9, N * F / t1 / 1e6, t2 * 1e6, t2 / t1), flush=True)
