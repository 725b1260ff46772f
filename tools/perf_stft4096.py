#!/usr/bin/env python
"""config 5 (BASELINE.json configs[4]): psnd_stft_fwd magnitude, n_fft 4096 / hop 1024, 30 s clips at 44.1 kHz.
algorithmic GB/s = (4NT + 4NKF) / t.  PSND_STFT4096_V1=1 selects the first-generation kernel (8-frame tiles, one workgroup per CU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
n, h, T = 4096, 1024, 1323000
m = np.arange(n); w = (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
plan = K.stft_plan(n, w).to(dev)
for N in (16, 32):
    wav = torch.randn(N, T, device=dev) * 0.07
    F = K.frame_count(T, n, h); Kb = n // 2 + 1
    mag = torch.empty(N, Kb, F, device=dev)
    def run():
        check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
    for _ in range(3): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e-3)
    b = 4 * N * T + 4 * N * Kb * F
    print('%s n=4096 hop=1024 N=%d (%.0f MB): %.1f us  %.0f GB/s  %.1f%% of 8 TB/s' % (
        'v1' if os.environ.get('PSND_STFT4096_V1') else 'v2', N, b / 1e6, best * 1e6, b / best / 1e9, b / best / 8e12 * 100), flush=True)
    del wav, mag
