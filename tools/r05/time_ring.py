"""HIP-event timing of psnd_stft_mag_nfk at config 5 (32 x 30 s) for the library selected by PSND_LIB; PSND_ABLATE=2 drops the stores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 1323000
plan = K.stft_plan(4096, periodic_window('hann', 4096).astype(np.float32)).to(dev)
x = 0.07 * torch.randn(N, T, device=dev)
o = torch.empty((N, K.frame_count(T, 4096, 1024), 2049), device=dev)
fn = lambda: K.stft_mag_nfk(x, 4096, 1024, plan, out=o)
for _ in range(5): fn()
ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts = np.array(ts)
print('%-28s ABLATE=%s NORING=%s: mean %.1f med %.1f min %.1f us' % (os.path.basename(os.environ.get('PSND_LIB', 'default')), os.environ.get('PSND_ABLATE', '-'),
      os.environ.get('PSND_STFT4096_NORING', '-'), ts.mean(), np.median(ts), ts.min()), flush=True)
if os.environ.get('PSND_SERIES'):
    print(' '.join('%.0f' % v for v in ts))
