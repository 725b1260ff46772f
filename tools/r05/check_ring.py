"""round 5: the LDS-ring config-5 kernel (psnd_stft_r.hip) against the register-load kernel it replaces (PSND_STFT4096_NORING=1):
bit-identity is NOT expected (the transposes move the same values; the arithmetic is the same -> it should in fact be bit-identical),
then HIP-event timings of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pytorch_sound_amd import kernels as K, _lib
from pytorch_sound_amd.models.transforms import periodic_window

dev = torch.device('cuda:0')
w = periodic_window('hann', 4096).astype(np.float32)
plan = K.stft_plan(4096, w).to(dev)


def both(x, framing=0):
    os.environ.pop('PSND_STFT4096_NORING', None); _lib.refresh_switches()
    a = K.stft_mag_nfk(x, 4096, 1024, plan, framing)
    torch.cuda.synchronize()
    os.environ['PSND_STFT4096_NORING'] = '1'; _lib.refresh_switches()
    b = K.stft_mag_nfk(x, 4096, 1024, plan, framing)
    torch.cuda.synchronize()
    os.environ.pop('PSND_STFT4096_NORING', None); _lib.refresh_switches()
    return a, b


for (N, T, framing) in [(1, 9000, 0), (3, 44100, 0), (2, 30000, 1), (21, 60000, 0), (300, 8192, 0), (2, 1323000, 0), (32, 1323000, 0), (5, 4096 * 40, 1)]:
    x = 0.07 * torch.randn(N, T, device=dev)
    a, b = both(x, framing)
    d = float((a - b).abs().max())
    print('N %d T %d framing %d: F %d  max|ring - regs| = %g (max %g) finite %s' % (N, T, framing, a.shape[1], d, float(b.max()), bool(torch.isfinite(a).all())), flush=True)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.array(ts)


for N in (32, 16, 64):
    T = 1323000
    x = 0.07 * torch.randn(N, T, device=dev)
    F = K.frame_count(T, 4096, 1024)
    o = torch.empty((N, F, 2049), device=dev)
    nbytes = 4 * N * T + 4 * N * 2049 * F
    for name, env in (('ring', None), ('regs', '1'), ('ring', None), ('regs', '1')):
        if env: os.environ['PSND_STFT4096_NORING'] = env
        else: os.environ.pop('PSND_STFT4096_NORING', None)
        _lib.refresh_switches()
        ts = timeit(lambda: K.stft_mag_nfk(x, 4096, 1024, plan, out=o))
        print('N %d %s: mean %.1f us min %.1f us -> %.3f of 8 TB/s (mean)' % (N, name, ts.mean(), ts.min(), nbytes / ts.mean() / 1e-6 / 8e12), flush=True)
os.environ.pop('PSND_STFT4096_NORING', None)
