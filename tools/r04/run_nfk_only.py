"""a few launches of psnd_stft_mag_nfk for rocprofv3: run_nfk_only.py <n_fft> <clips> <T> <reps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window
n_fft, N, T, reps = (int(a) for a in sys.argv[1:5])
hop = n_fft // 4
dev = torch.device('cuda:0')
plan = K.stft_plan(n_fft, periodic_window('hann', n_fft).astype(np.float32)).to(dev)
x = 0.07 * torch.randn(N, T, device=dev)
out = torch.empty((N, K.frame_count(T, n_fft, hop), n_fft // 2 + 1), device=dev)
for _ in range(reps):
    K.stft_mag_nfk(x, n_fft, hop, plan, out=out)
torch.cuda.synchronize()
