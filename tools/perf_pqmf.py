#!/usr/bin/env python
"""PQMF analysis + synthesis, forward and backward (transforms.py:492-560): polyphase HIP kernels against the reference's
conv1d / conv_transpose1d formulation on the same GPU (MIOpen)."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sound_amd.models.transforms import PQMF  # noqa: E402

dev = torch.device('cuda:0')
pq = PQMF().to(dev)


def torch_analysis(x):
    x = torch.nn.functional.conv1d(pq.pad_fn(x), pq.analysis_filter)
    return torch.nn.functional.conv1d(x, pq.updown_filter, stride=pq.subbands)


def torch_synthesis(x):
    x = torch.nn.functional.conv_transpose1d(x, pq.updown_filter * pq.subbands, stride=pq.subbands)
    return torch.nn.functional.conv1d(pq.pad_fn(x), pq.synthesis_filter)


def timeit(f, n):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, T in ((16, 8192), (64, 1 << 18)):
    x = torch.randn(B, 1, T, device=dev, requires_grad=True)

    def run(an, sy):
        x.grad = None
        sy(an(x)).sum().backward()
    a = timeit(lambda: run(pq.analysis, pq.synthesis), 20)
    b = timeit(lambda: run(torch_analysis, torch_synthesis), 20)
    print('B=%d T=%d analysis+synthesis fwd+bwd: hip %.3f ms, torch %.3f ms (x%.1f); bytes moved (hip, min) %.0f MB'
          % (B, T, a, b, b / a, 8 * B * T * 4 / 1e6), flush=True)
