#!/usr/bin/env python
"""Phase timeline of conv_cl_kernel at the config-2 shape (needs the -DPSND_TRACE variant, see trace_stft.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import cl
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
N, L, C, HP, k, dil = 32, 173, 256, 8, 3, int(os.environ.get('DIL', '1'))
Lp = L + 2 * HP
x = torch.randn(N, Lp, C, device=dev).to(torch.bfloat16)
w = (torch.randn(k, C, C, device=dev) * 0.05).to(torch.bfloat16)
bias = torch.zeros(C, device=dev)
out = torch.empty_like(x)
nwg = ((N * Lp + 63) // 64) * (C // 64)
trace = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
def run():
    check(lib().psnd_conv1d_cl(ptr(x), None, None, 0.0, ptr(w), ptr(bias), None, None, N, Lp, L, HP, C, C, k, -dil, dil, 0.1, 0.0,
                               ptr(out), None, None, stream_ptr(dev)), "conv")
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
print('conv_cl %d WGs: %.2f us per launch (back-to-back)' % (nwg, s.elapsed_time(e) / 50 * 1e3))
os.environ['PSND_TRACE_PTR'] = hex(trace.data_ptr())
run(); torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(nwg, 4, 8)[:, :, :6]
ok = (tr != 0).all(axis=(1, 2))
tr = tr[ok]
names = ['entry -> stage 0 in LDS', 'barrier', 'main loop (rest)', 'epilogue issue', 'drain']
d = np.diff(tr, axis=2).astype(np.float64)
for i in range(5):
    print('  %-26s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (names[i], d[:, :, i].mean(), *np.percentile(d[:, :, i], [10, 50, 90])))
life = (tr[:, :, 5] - tr[:, :, 0]).astype(np.float64)
print('  %-26s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f ; traced WGs %d' % ('wave lifetime', life.mean(), *np.percentile(life, [10, 50, 90]), tr.shape[0]))
