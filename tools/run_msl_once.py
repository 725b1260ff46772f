#!/usr/bin/env python
"""a few multi_stft_loss forward+backward passes at one shape, for rocprofv3 (N, T from argv)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sound_amd.models import sound as S  # noqa: E402

N, T = int(sys.argv[1]), int(sys.argv[2])
PARAMS = [(1024, 600, 120), (2048, 1200, 240), (512, 240, 50)]
target = 0.1 * torch.randn(N, T, device='cuda:0')
pred = (target + 0.01 * torch.randn(N, T, device='cuda:0')).requires_grad_(True)
for _ in range(6):
    pred.grad = None
    S.multi_stft_loss(pred, target, PARAMS)[0].backward()
torch.cuda.synchronize()
