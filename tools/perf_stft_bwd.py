#!/usr/bin/env python
"""psnd_stft_bwd (gmag -> gwav, magnitude path) and psnd_istft at HBM-sized working sets.
bwd bytes: 4NKF (gmag) + 4NT (wav, recompute) + 4NT (gwav) (+ 4NT zero fill)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
dev = torch.device('cuda:0')
def hann(n):
    m = np.arange(n); return (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
def timeit(fn, iters=10, warm=3):
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
for n, h, N, T in [(1024, 256, 32, 44100), (1024, 256, 1024, 44100), (4096, 1024, 16, 1323000)]:
    wav = torch.randn(N, T, device=dev) * 0.07
    plan = K.stft_plan(n, hann(n)).to(dev)
    F = K.frame_count(T, n, h); Kb = n // 2 + 1
    gmag = torch.randn(N, Kb, F, device=dev)
    t = timeit(lambda: K.stft_backward(wav, n, h, plan, gmag=gmag))
    b = 4 * N * Kb * F + 8 * N * T
    print('stft_bwd n=%d N=%d T=%d: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)' % (n, N, T, t * 1e6, b / t / 1e9, b / t / 8e12 * 100), flush=True)
    del wav, gmag
for n, h, N, F in [(1024, 256, 32, 173), (1024, 256, 1024, 173)]:
    plan = K.stft_plan(n, hann(n)).to(dev)
    mag = torch.rand(N, n // 2 + 1, F, device=dev)
    ph = torch.rand(N, n // 2 + 1, F, device=dev) * 6.28
    win = torch.from_numpy(hann(n)).to(dev)
    t = timeit(lambda: K.istft(mag, ph, n, h, plan, 1e-9, win))
    b = 8 * N * (n // 2 + 1) * F + 4 * N * (F - 1) * h
    print('istft n=%d N=%d F=%d: %.1f us  %.0f GB/s' % (n, N, F, t * 1e6, b / t / 1e9), flush=True)
# (re, im) gradient path (STFTTorchAudio)
for n, h, N, T in [(1024, 256, 1024, 44100)]:
    wav = torch.randn(N, T, device=dev) * 0.07
    plan = K.stft_plan(n, hann(n)).to(dev)
    F = K.frame_count(T, n, h); Kb = n // 2 + 1
    gre = torch.randn(N, Kb, F, device=dev); gim = torch.randn(N, Kb, F, device=dev)
    t = timeit(lambda: K.stft_backward(wav, n, h, plan, gre=gre, gim=gim))
    print('stft_bwd(re,im) n=%d N=%d: %.1f us' % (n, N, t * 1e6), flush=True)
