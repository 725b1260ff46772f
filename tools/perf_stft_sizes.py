#!/usr/bin/env python
"""psnd_stft_fwd (magnitude) across FFT sizes at HBM-sized working sets: algorithmic GB/s = (4NT + 4NKF) / t."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
def hann(n):
    m = np.arange(n); return (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
for n, h, N, T in [(4096, 1024, 4, 1323000), (4096, 1024, 32, 1323000), (2048, 512, 64, 1323000), (2048, 240, 1024, 16384), (1024, 256, 1024, 44100), (512, 128, 1024, 44100), (512, 50, 1024, 16384), (256, 64, 1024, 44100)]:
    wav = torch.randn(N, T, device=dev) * 0.07
    plan = K.stft_plan(n, hann(n)).to(dev)
    F = K.frame_count(T, n, h); Kb = n // 2 + 1
    mag = torch.empty(N, Kb, F, device=dev)
    def run():
        check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
    for _ in range(3): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e-3)
    b = 4 * N * T + 4 * N * Kb * F
    print('n=%d hop=%d N=%d T=%d (%.0f MB): %.1f us  %.0f GB/s  %.1f%% of 8 TB/s' % (n, h, N, T, b / 1e6, best * 1e6, b / best / 1e9, b / best / 8e12 * 100), flush=True)
    del wav, mag
