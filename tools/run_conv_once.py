#!/usr/bin/env python
"""a few forward convs and paired backward launches at the config-2 shape (one 256 -> 256, k = 3 conv, 32 x 173 frames) - PMC target"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda:0')
print(bench._conv_roofline(dev, 32, 173, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 5))
