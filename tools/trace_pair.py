#!/usr/bin/env python
"""where a workgroup of psnd_conv1d_cl_pair spends its cycles (s_memtime stamps, PSND_PAIR_TRACE_PTR): config-2 shape"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
N, L, HP, C, k, d = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 173, 5, 256, 3, 3
Lp = (L + 2 * HP + 7) // 8 * 8
R = N * Lp
x = torch.zeros(N, Lp, C, device=dev, dtype=torch.bfloat16)
x[:, HP:HP + L] = torch.randn(N, L, C, device=dev).to(torch.bfloat16)
w1 = (torch.randn(3, C, C, device=dev) / 28).to(torch.bfloat16)
w2 = (torch.randn(3, C, C, device=dev) / 28).to(torch.bfloat16)
b = torch.zeros(C, device=dev)
mid, raw, act = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
ntile = (R + 13) // 14
tr = torch.zeros(ntile * 8, dtype=torch.int64, device=dev)
def run():
    check(lib().psnd_conv1d_cl_pair(ptr(x), ptr(w1), ptr(b), None, 1.0, 0.1, ptr(mid), ptr(w2), ptr(b), None, 1.0, ptr(x), N, Lp, L, HP, C, k,
                                    -d, d, -1, 1, 0.1, ptr(raw), ptr(act), stream_ptr(dev)), 'pair')
for _ in range(3): run()
torch.cuda.synchronize()
os.environ['PSND_PAIR_TRACE_PTR'] = str(tr.data_ptr())
run(); torch.cuda.synchronize()
del os.environ['PSND_PAIR_TRACE_PTR']
t = tr.view(ntile, 8).cpu().double()
t = t[t[:, 0] > 0]
ntile = t.shape[0]
names = ['-> tile in LDS', 'conv 1 loop', 'epilogue 1 + mid out', 'conv 2 loop', 'epilogue 2']
d_ = t[:, 1:6] - t[:, 0:5]
print('workgroups', ntile, ' s_memtime ticks = shader cycles (wave 0 of every workgroup)')
for i, n in enumerate(names):
    print('%-24s mean %8.1f  min %8.1f  max %8.1f' % (n, d_[:, i].mean(), d_[:, i].min(), d_[:, i].max()))
print('epilogue 1: math + LDS writes %.1f, loads issued + barrier %.1f, mid store issue %.1f' % ((t[:, 6] - t[:, 2]).mean(), (t[:, 7] - t[:, 6]).mean(), (t[:, 3] - t[:, 7]).mean()))
print('whole workgroup mean %.1f ticks; first start to last end %.1f ticks' % ((t[:, 5] - t[:, 0]).mean(), t[:, 5].max() - t[:, 0].min()))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
print('launch back to back: %.2f us' % (s.elapsed_time(e) / 50 * 1e3))
