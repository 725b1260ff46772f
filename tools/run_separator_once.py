#!/usr/bin/env python
"""config-2 separator forward + backward (32 x 513 x 173), a few iterations - rocprofv3 / PMC target for the conv body's kernels"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd.models import build_model
from pytorch_sound_amd.models import separator  # noqa: F401
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_model('conv_separator_voicebank').to(dev)
mag = torch.rand(32, 513, 173, device=dev) * 4
tgt = torch.rand(32, 513, 173, device=dev)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    model.zero_grad()
    (model(mag) - tgt).abs().mean().backward()
torch.cuda.synchronize()
print('done')
