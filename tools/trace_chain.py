#!/usr/bin/env python
"""where a workgroup of psnd_conv1d_cl_chain spends its cycles (s_memtime stamps, PSND_PAIR_TRACE_PTR): config-2 shape, a ResBlock1"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd import _lib
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
N, L, HP, C, k = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 173, 5, 256, 3
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Lp = (L + 2 * HP + 7) // 8 * 8
R = N * Lp
x = torch.zeros(N, Lp, C, device=dev, dtype=torch.bfloat16)
x[:, HP:HP + L] = torch.randn(N, L, C, device=dev).to(torch.bfloat16)
ws = [(torch.randn(3, C, C, device=dev) / 28).to(torch.bfloat16) for _ in range(2 * npairs)]
b = torch.zeros(C, device=dev)
outs = [[torch.empty_like(x) for _ in range(3)] for _ in range(npairs)]
dils = [int(a) for a in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 3, 5, 1][:npairs]
arr = (_lib.ChainPair * npairs)()
taps = []
for i, d in enumerate(arr):
    d.W1, d.bias1, d.act1_slope, d.mid_out = ws[2 * i].data_ptr(), b.data_ptr(), 0.1, outs[i][0].data_ptr()
    d.W2, d.bias2, d.off1, d.dstep1, d.off2, d.dstep2 = ws[2 * i + 1].data_ptr(), b.data_ptr(), -dils[i], dils[i], -1, 1
    d.act2_slope, d.out_raw, d.out_act = 0.1, outs[i][1].data_ptr(), outs[i][2].data_ptr()
    taps += [-dils[i], dils[i], -1, 1]
mr = ctypes.c_int(0)
ts = lib().psnd_conv1d_cl_chain_plan(C, k, npairs, (ctypes.c_int * len(taps))(*taps), R, ctypes.byref(mr))
print('row tile', 32 * mr.value, 'dilations', dils)
ntile = (R + ts - 1) // ts
tr = torch.zeros(ntile * 8 * 16, dtype=torch.int64, device=dev)
def run():
    check(lib().psnd_conv1d_cl_chain(ptr(x), ptr(x), ctypes.addressof(arr), npairs, N, Lp, L, HP, C, k, stream_ptr(dev)), 'chain')
for _ in range(3): run()
torch.cuda.synchronize()
os.environ['PSND_PAIR_TRACE_PTR'] = str(tr.data_ptr())
run(); torch.cuda.synchronize()
del os.environ['PSND_PAIR_TRACE_PTR']
tw = tr.view(ntile, 8, 16).cpu().double()
t = tw[:, 0]
print('workgroups', ntile, 'owned rows', ts, ' s_memtime ticks (100 MHz * ? - relative) wave 0 of every workgroup')
names = ['-> tiles in LDS'] + [n % (i + 1) for i in range(npairs) for n in ('pair %d conv 1 loop', 'pair %d epilogue 1 + barrier', 'pair %d copy-out + conv 2 loop', 'pair %d epilogue 2 + barrier')]
d_ = t[:, 1:2 + 4 * npairs] - t[:, 0:1 + 4 * npairs]
for i, n in enumerate(names):
    print('%-34s mean %8.1f  min %8.1f  max %8.1f' % (n, d_[:, i].mean(), d_[:, i].min(), d_[:, i].max()))
last = 1 + 4 * npairs
print('per wave, mean over workgroups, ticks since the workgroup\'s wave-0 start: rows = stamps (0 start, 1 tiles in LDS, then per pair: conv 1 done, barrier, conv 2 done, barrier)')
rel = (tw - tw[:, :1, :1]).mean(0)
for i in list(range(last + 1)) + [14, 15]:
    print('stamp %2d: ' % i + ' '.join('%7.0f' % rel[w, i] for w in range(8)))
print('whole workgroup mean %.1f ticks; first start to last end %.1f ticks' % ((t[:, last] - t[:, 0]).mean(), t[:, last].max() - t[:, 0].min()))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
print('launch back to back: %.2f us' % (s.elapsed_time(e) / 50 * 1e3))
