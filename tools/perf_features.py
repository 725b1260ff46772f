#!/usr/bin/env python
"""Quick GPU timing probe for the STFT / mel kernels (HIP events on torch's current stream).
Prints algorithmic GB/s = (4*N*T + 4*N*K*F) / t for the STFT stage."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K

def hann(n):
    m = np.arange(n); return (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

dev = torch.device('cuda:0')
print(torch.cuda.get_device_name(0))
cases = [(1024,256,32,44100), (1024,256,256,44100), (1024,256,1024,44100), (1024,256,2048,44100*2),
         (512,128,512,44100), (2048,512,256,88200), (256,64,512,44100), (4096,1024,4,1323000)]
if len(sys.argv) > 1: cases = cases[:int(sys.argv[1])]
for n, h, N, T in cases:
    wav = torch.randn(N, T, device=dev) * 0.07
    plan = K.stft_plan(n, hann(n)).to(dev)
    F = K.frame_count(T, n, h); Kb = n//2+1
    o = K.stft_forward(wav, n, h, plan)  # allocate once
    mag = o['mag']
    import ctypes
    from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
    def run():
        check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
    t = timeit(run)
    byts = 4*N*T + 4*N*Kb*F
    print('stft n=%d hop=%d N=%d T=%d F=%d: %.1f us  %.1f GB/s (%.1f%% of 8 TB/s)  %.2f Mframes/s' % (n,h,N,T,F,t*1e6, byts/t/1e9, byts/t/8e12*100, N*F/t/1e6), flush=True)
    if n == 1024:
        # mel stage
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import features as ofe
        W = ofe.mel_filterbank(22050, 1024, 80, 0, 8000)
        mp = K.mel_plan(W).to(dev)
        out, _ = K.mel_forward(mag, mp, 80, K.LOG_E, 1e-6, None, -11.5, 6.9)
        def run2():
            check(lib().psnd_mel_fwd(ptr(mag), N, F, 80, Kb, ptr(mp), 1, 1e-6, -1.0, -11.5, 6.9, ptr(out), None, stream_ptr(dev)), 'mel')
        t2 = timeit(run2)
        b2 = 4*N*Kb*F + 4*N*80*F
        print('   mel: %.1f us  %.1f GB/s' % (t2*1e6, b2/t2/1e9), flush=True)
        tt = timeit(lambda: torch.stft(wav, n, h, n, torch.hann_window(n, device=dev), True, 'reflect', False, True, return_complex=True).abs())
        print('   torch.stft(rocFFT)+abs: %.1f us' % (tt*1e6), flush=True)
    del wav, mag, o
# copy ceiling
x = torch.empty(256*1024*1024//4, device=dev); y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x))
print('copy 256MiB: %.1f us  %.1f GB/s (r+w)' % (t*1e6, 2*x.numel()*4/t/1e9))
