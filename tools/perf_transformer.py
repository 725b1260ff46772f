#!/usr/bin/env python
"""config 4 shape: PositionalEncoding -> MultiHeadAttention(C=256, H=4) -> PointwiseFeedForward over an 80-mel input projected
by a 1x1 conv, batch 32, one length bucket (T frames), padding mask; forward + backward.  HIP kernels (GroupNorm(1,C)+residual,
masked softmax over keys) vs the torch formulation of the same modules on the same GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd.models import modules as M
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, N = 256, 4, 32
proj = torch.nn.Conv1d(80, C, 1).to(dev)
pe = M.PositionalEncoding(C, 2048).to(dev)
mha = M.MultiHeadAttention(C, H, 0.0).to(dev)
ffn = M.PointwiseFeedForward(C, 0.0).to(dev)
params = list(proj.parameters()) + list(mha.parameters()) + list(ffn.parameters())
def step(x, mask):
    for p in params: p.grad = None
    h = pe(proj(x))
    h, att = mha(h, mask)
    y = ffn(h)
    (y.abs().mean() + 1e-3 * att.mean()).backward()
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
hip_ok = M._hip_ok
for T in (173, 690, 1292):
    x = torch.randn(N, 80, T, device=dev)
    lens = torch.linspace(0.8 * T, T, N).long()
    mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    M._hip_ok = hip_ok
    t1 = timeit(lambda: step(x, mask))
    M._hip_ok = lambda t: False
    t2 = timeit(lambda: step(x, mask))
    print('T=%4d frames (att %4.0f MB): HIP norm/softmax kernels %.2f ms | torch formulation %.2f ms | x%.2f' % (T, H * N * T * T * 4 / 1e6, t1, t2, t2 / t1), flush=True)
