#!/usr/bin/env python
"""Phase timeline of stft_bwd_n1024_mag_kernel (needs the -DPSND_TRACE variant, see trace_stft.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
n, h, N, T = 1024, 256, int(os.environ.get('NCLIPS', '256')), 44100
dev = torch.device('cuda:0')
m = np.arange(n); w = (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
wav = torch.randn(N, T, device=dev) * 0.07
plan = K.stft_plan(n, w).to(dev)
F = K.frame_count(T, n, h)
gmag = torch.randn(N, n // 2 + 1, F, device=dev)
nwg = (N * ((F + 15) // 16) + 7) // 8 * 8
trace = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(3): K.stft_backward(wav, n, h, plan, gmag=gmag)
torch.cuda.synchronize()
os.environ['PSND_TRACE_PTR'] = hex(trace.data_ptr())
K.stft_backward(wav, n, h, plan, gmag=gmag); torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(nwg, 4, 8)
tr = tr[(tr != 0).all(axis=(1, 2))]
names = ['prologue (span, tables, gmag)', 'fwd pass 1 + exchange', 'pairs: fwd r16, split, adj, inv r16', 'lanes: inv r32, window', 'zero span + ds_add', 'span -> gwav (issue)', 'drain']
d = np.diff(tr, axis=2).astype(np.float64)
for i in range(7):
    print('  %-38s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (names[i], d[:, :, i].mean(), *np.percentile(d[:, :, i], [10, 50, 90])))
life = (tr[:, :, 7] - tr[:, :, 0]).astype(np.float64)
print('  %-38s mean %8.0f ; traced WGs %d' % ('wave lifetime', life.mean(), tr.shape[0]))
