#!/usr/bin/env python
"""a few forward+backward passes of hifi_gan_v1 on the CL kernel path (16 x 32 mel frames) - rocprofv3 target"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd.models import build_model
from pytorch_sound_amd.models.vocoders import hifi_gan  # noqa: F401
g = build_model(sys.argv[1] if len(sys.argv) > 1 else 'hifi_gan_v1').cuda()
x = torch.randn(16, 80, 32, device='cuda')
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    g.zero_grad(set_to_none=True)
    g(x).abs().mean().backward()
torch.cuda.synchronize()
