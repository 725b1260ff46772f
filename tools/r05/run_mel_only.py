"""a few launches of psnd_mel_fwd (1024 clips x 2 s of magnitudes -> 80 log-mel bands) for rocprofv3: run_mel_only.py <ignored> <clips> <ignored> <reps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.utils.mel import mel_filterbank
N, reps = int(sys.argv[2]), int(sys.argv[4])
dev = torch.device('cuda:0')
plan = K.mel_plan(mel_filterbank(22050, 1024, 80, 0.0, 8000.0)).to(dev)
mag = torch.rand(N, 513, 173, device=dev)
out = torch.empty(N, 80, 173, device=dev)
for _ in range(reps):
    K.mel_forward(mag, plan, 80, K.LOG_E, 1e-6, None, -11.5, 6.9, out=out)
torch.cuda.synchronize()
