#!/usr/bin/env python
"""Phase timeline of stft_fwd_n1024_kernel from s_memtime stamps (needs a -DPSND_TRACE build of the library:
   tools/build_variant.sh trace -DPSND_TRACE ; PSND_LIB=tools/mb/variants/libpsnd_trace.so python tools/trace_stft.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
n, h, N, T = 1024, 256, int(os.environ.get('NCLIPS', '1024')), 44100
dev = torch.device('cuda:0')
m = np.arange(n); w = (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
wav = torch.randn(N, T, device=dev) * 0.07
plan = K.stft_plan(n, w).to(dev)
F = K.frame_count(T, n, h); Kb = n//2+1
mag = torch.empty(N, Kb, F, device=dev)
ntile = (F + 15) // 16
nwg = (N * ntile + 7) // 8 * 8
trace = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
def run():
    check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
for _ in range(3): run()
torch.cuda.synchronize()
os.environ['PSND_TRACE_PTR'] = hex(trace.data_ptr())
run(); torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(nwg, 4, 8)
tr = tr[(tr[:, :, 7] != 0).all(1)]
if os.environ.get('PSND_TRACE_ITER', '0') != '0':
    tr = tr[:, :, 2:]   # prologue stamps belong to another tile
t0 = tr[:, :, 0].min()
names0 = ['prologue: loads -> LDS', 'barrier (span visible)', 'taps+window+fft32', 'xchg half0 + read A', 'xchg half1+fftA+read B+commit', 'fft B + post_emit + store issue', 'drain (vmcnt0) | to next tile start']
names = names0 if tr.shape[2] == 8 else names0[2:]
d = np.diff(tr, axis=2).astype(np.float64)
print('waves traced: %d ; kernel span %.1f kcycles (s_memtime units)' % (tr.shape[0]*4, (tr[:, :, -1].max() - t0)/1e3))
for i in range(tr.shape[2] - 1):
    print('  %-30s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (names[i], d[:, :, i].mean(), *np.percentile(d[:, :, i], [10, 50, 90])))
life = (tr[:, :, -1] - tr[:, :, 0]).astype(np.float64)
print('  %-30s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % ('wave lifetime', life.mean(), *np.percentile(life, [10, 50, 90])))
e = np.sort(tr[:, 0, 0] - t0)

