"""(N, K, F) at config 5: is 'the (N, F, K) ring kernel + a transposing pass' a road to the reference layout?  HIP events, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window
dev = torch.device('cuda:0')
N, T = 32, 1323000
plan = K.stft_plan(4096, periodic_window('hann', 4096).astype(np.float32)).to(dev)
x = 0.07 * torch.randn(N, T, device=dev)
F = K.frame_count(T, 4096, 1024)
nfk = torch.empty((N, F, 2049), device=dev)
nkf = torch.empty((N, 2049, F), device=dev)
fns = {'nkf kernel (stft_fwd_n4096w_kernel)': lambda: K.stft_forward(x, 4096, 1024, plan, out_mag=nkf),
       'nfk ring kernel (stft_fwd_n4096r_kernel)': lambda: K.stft_mag_nfk(x, 4096, 1024, plan, out=nfk),
       'transposing copy (N,F,K)->(N,K,F), 339 MB each way (library copy kernel)': lambda: nkf.copy_(nfk.transpose(1, 2))}
ts = {k: [] for k in fns}
for k, f in fns.items():
    for _ in range(3): f()
torch.cuda.synchronize()
for r in range(12):
    evs = []
    for k, f in fns.items():
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); evs.append((k, e0, e1))
    torch.cuda.synchronize()
    for k, e0, e1 in evs: ts[k].append(e0.elapsed_time(e1) * 1e3)
for k in fns: print('%-80s median %.1f us' % (k, np.median(ts[k])))
