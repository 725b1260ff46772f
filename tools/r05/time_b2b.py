"""bench-style timing (bench.py _nfk_roofline): 13 back-to-back launches, HIP events around each, mean of the last 10.  argv: n_fft clips"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window
dev = torch.device('cuda:0')
n_fft = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hop = n_fft // 4
T = 1323000 if n_fft == 4096 else 44100
plan = K.stft_plan(n_fft, periodic_window('hann', n_fft).astype(np.float32)).to(dev)
x = 0.07 * torch.randn(N, T, device=dev)
F = K.frame_count(T, n_fft, hop)
o = torch.empty((N, F, n_fft // 2 + 1), device=dev)
nbytes = 4 * N * T + 4 * N * (n_fft // 2 + 1) * F
res = []
for rep in range(3):
    torch.cuda.synchronize()
    evs = []
    for i in range(13):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.stft_mag_nfk(x, n_fft, hop, plan, out=o); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = float(np.mean([a.elapsed_time(b) for a, b in evs[3:]])) * 1e-3
    res.append(t)
print('%s n_fft %d N %d ABLATE=%s NORING=%s: b2b launch us %s -> frac %.3f' % (os.path.basename(os.environ.get('PSND_LIB', 'default')), n_fft, N,
      os.environ.get('PSND_ABLATE', '-'), os.environ.get('PSND_STFT4096_NORING', '-'), ' '.join('%.1f' % (v * 1e6) for v in res), nbytes / min(res) / 8e12), flush=True)
