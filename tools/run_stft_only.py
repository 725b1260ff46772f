#!/usr/bin/env python
"""run only psnd_stft_fwd (mag) a few times on a large batch - target for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = n // 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 44100
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device('cuda:0')
m = np.arange(n); w = (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
wav = torch.randn(N, T, device=dev) * 0.07
plan = K.stft_plan(n, w).to(dev)
F = K.frame_count(T, n, h); Kb = n//2+1
mag = torch.empty(N, Kb, F, device=dev)
for _ in range(iters):
    check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
torch.cuda.synchronize()
print('done', N*F, 'frames', 4*N*T + 4*N*Kb*F, 'bytes')
