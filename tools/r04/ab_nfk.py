"""same-process A/B of libpsnd variants is not possible (one library per process): this script times ONE library; run it once per PSND_LIB
in one gpurun call.  Prints mean / median / min of psnd_stft_mag_nfk over `reps` launches, interleaved with nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window
n_fft, N, T = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
hop = n_fft // 4
dev = torch.device('cuda:0')
plan = K.stft_plan(n_fft, periodic_window('hann', n_fft).astype(np.float32)).to(dev)
x = 0.07 * torch.randn(N, T, device=dev)
F, Kb = K.frame_count(T, n_fft, hop), n_fft // 2 + 1
out = torch.empty((N, F, Kb), device=dev)
for _ in range(10):
    K.stft_mag_nfk(x, n_fft, hop, plan, out=out)
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.stft_mag_nfk(x, n_fft, hop, plan, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts = np.array(ts)
nb = 4 * N * T + 4 * N * Kb * F
print('%-28s n_fft %d N %d: mean %.1f median %.1f min %.1f us -> %.3f (median) of 8 TB/s' % (
    os.path.basename(os.environ.get('PSND_LIB', 'libpsnd_hip.so')), n_fft, N, ts.mean(), np.median(ts), ts.min(), nb / np.median(ts) / 1e-6 / 8e12))
