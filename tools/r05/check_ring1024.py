"""round 5: stft_fwd_n1024r_kernel (ring + loader wave) against stft_fwd_n1024q_kernel (PSND_STFT1024_NORING=1): same arithmetic -> bit-identical"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K, _lib
from pytorch_sound_amd.models.transforms import periodic_window
dev = torch.device('cuda:0')
plan = K.stft_plan(1024, periodic_window('hann', 1024).astype(np.float32)).to(dev)
def both(x, framing=0):
    os.environ.pop('PSND_STFT1024_NORING', None); _lib.refresh_switches()
    a = K.stft_mag_nfk(x, 1024, 256, plan, framing); torch.cuda.synchronize()
    os.environ['PSND_STFT1024_NORING'] = '1'; _lib.refresh_switches()
    b = K.stft_mag_nfk(x, 1024, 256, plan, framing); torch.cuda.synchronize()
    os.environ.pop('PSND_STFT1024_NORING', None); _lib.refresh_switches()
    return a, b
for (N, T, framing) in [(1, 1024, 0), (1, 2052, 0), (4, 44100, 0), (3, 8192, 1), (37, 10000, 0), (1, 516, 0), (64, 44100, 0), (300, 3000, 0), (1024, 44100, 0), (2, 1323000, 0)]:
    x = 0.07 * torch.randn(N, T, device=dev)
    a, b = both(x, framing)
    print('N %d T %d framing %d: F %d  max|ring - q| = %g (max %g) finite %s' % (N, T, framing, a.shape[1], float((a - b).abs().max()), float(b.max()), bool(torch.isfinite(a).all())), flush=True)
