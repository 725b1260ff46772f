#!/usr/bin/env python
"""config 3 shape: hifi_gan_v1 generator forward + backward, batch 16 x 32 mel frames (8192 samples):
channels-last bf16 kernel path (Generator.forward_cl) vs the library path (fp32, and bf16 autocast)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd.models import build_model
from pytorch_sound_amd.models.vocoders import hifi_gan  # noqa: F401 (registers the architectures)
dev = torch.device('cuda:0')
arch = sys.argv[1] if len(sys.argv) > 1 else 'hifi_gan_v1'
g = build_model(arch).to(dev)
x = torch.randn(16, 80, 32, device=dev)
def step(autocast=False):
    g.zero_grad(set_to_none=True)
    if autocast:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = g(x)
    else:
        y = g(x)
    y.float().abs().mean().backward()
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
g.use_cl = True
t_cl = timeit(step)
g.use_cl = False
t_lib = timeit(step)
t_lib16 = timeit(lambda: step(True))
print('%s fwd+bwd, 16 x 8192 samples: CL kernels %.2f ms | library fp32 %.2f ms | library bf16 autocast %.2f ms' % (arch, t_cl, t_lib, t_lib16))
