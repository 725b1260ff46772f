#!/usr/bin/env python
"""Phase timeline of conv_wgrad_kernel at the config-2 shape (needs the -DPSND_TRACE variant, see trace_stft.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
N, L, C, HP, k, dil = 32, 173, 256, 8, 3, int(os.environ.get('DIL', '1'))
Lp = L + 2 * HP
x = torch.randn(N, Lp, C, device=dev).to(torch.bfloat16)
g = torch.randn(N, Lp, C, device=dev).to(torch.bfloat16)
S = lib().psnd_conv1d_cl_wgrad_splits(N, Lp, C, C, k)
gw = torch.empty(S, k, C, C, device=dev)
gb = torch.empty(S, C, device=dev)
nwg = 4096
trace = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
def run():
    check(lib().psnd_conv1d_cl_wgrad(ptr(g), None, None, 0.1, ptr(x), N, Lp, C, C, k, -dil, dil, ptr(gw), ptr(gb), None,
                                     stream_ptr(dev)), 'wgrad')
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
print('wgrad: %.2f us per call (back-to-back)' % (s.elapsed_time(e) / 50 * 1e3))
os.environ['PSND_TRACE_PTR'] = hex(trace.data_ptr())
run(); torch.cuda.synchronize()
full = trace.cpu().numpy().reshape(nwg, 4, 8)
tr = full[:, :, :5]
ok = (tr != 0).all(axis=(1, 2))
tr = tr[ok]
full = full[ok].astype(np.float64)
print('  chunk 4: stage (incl. the wait for its loads) %.0f, barrier wait %.0f' % ((full[:, :, 6] - full[:, :, 5]).mean(), (full[:, :, 7] - full[:, :, 6]).mean()))
names = ['entry -> chunk 0 staged', 'row loop', 'atomics issue', 'drain']
d = np.diff(tr, axis=2).astype(np.float64)
for i in range(4):
    print('  %-26s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (names[i], d[:, :, i].mean(), *np.percentile(d[:, :, i], [10, 50, 90])))
life = (tr[:, :, 4] - tr[:, :, 0]).astype(np.float64)
print('  %-26s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f ; traced WGs %d' % ('wave lifetime', life.mean(), *np.percentile(life, [10, 50, 90]), tr.shape[0]))
