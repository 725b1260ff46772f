"""per-launch time series of a long back-to-back run (HIP events around each launch): how the (N, F, K) STFT kernels, the mel kernel and a plain
copy of the same bytes behave once the burst is over.  argv: launches"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window
dev = torch.device('cuda:0')
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def smi(tag):
    try:
        o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.splitlines() if any(k in l for k in ('sclk', 'mclk', 'fclk', 'Power', 'junction', 'hotspot', 'Junction'))]
        print(tag, ' | '.join(keep)[:600], flush=True)
    except Exception as e:
        print(tag, 'rocm-smi failed', e)


def series(name, launch, nbytes, L=L, gap=None):
    torch.cuda.synchronize()
    evs = []
    for i in range(L):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        evs.append((e0, e1))
        if gap and i % gap == gap - 1:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    ch = [ts[i:i + 16].mean() for i in range(0, L, 16)]
    print('%-28s best %.1f  chunks of 16 (us): %s  -> sustained(last half) %.1f us = %.3f of 8 TB/s' % (
        name, ts.min(), ' '.join('%.0f' % c for c in ch), ts[L // 2:].mean(), nbytes / (ts[L // 2:].mean() * 1e-6) / 8e12), flush=True)
    return ts


smi('idle:')
for n_fft, N, T in ((1024, 1024, 44100), (4096, 32, 1323000)):
    hop = n_fft // 4
    plan = K.stft_plan(n_fft, periodic_window('hann', n_fft).astype(np.float32)).to(dev)
    x = 0.07 * torch.randn(N, T, device=dev)
    F = K.frame_count(T, n_fft, hop)
    o = torch.empty((N, F, n_fft // 2 + 1), device=dev)
    nbytes = 4 * N * T + 4 * N * (n_fft // 2 + 1) * F
    series('stft nfk %d' % n_fft, lambda: K.stft_mag_nfk(x, n_fft, hop, plan, out=o), nbytes)
    smi('after %d:' % n_fft)
    series('stft nfk %d (sync / 8)' % n_fft, lambda: K.stft_mag_nfk(x, n_fft, hop, plan, out=o), nbytes, gap=8)
    # a copy moving the same bytes (reads wav-sized + writes mag-sized is not expressible; copy nbytes/2 -> nbytes total traffic)
    a = torch.empty(nbytes // 8, device=dev); b = torch.empty_like(a)
    series('copy same bytes', lambda: b.copy_(a), nbytes)
    del a, b, x, o
smi('end:')
