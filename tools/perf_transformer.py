#!/usr/bin/env python
"""config 4 shape: PositionalEncoding -> MultiHeadAttention(C=256, H=4) -> PointwiseFeedForward over an 80-mel input projected
by a 1x1 conv, batch 32, one length bucket (T frames), padding mask; forward + backward, replayed as a hipGraph (GPU time, no
host launch overhead).  round 2: psnd_linear1x1_* + psnd_mha_* ; round 1: library GEMMs / bmm around the softmax + GroupNorm
kernels ; torch: the plain torch formulation."""
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
    for p in params:
        p.grad = None
    h = pe(M._conv1x1(proj, x))
    h, att = mha(h, mask)
    y = ffn(h)
    y.abs().mean().backward()


def graph_time(x, mask, iters=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step(x, mask)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(x, mask)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for T in (173, 431, 690, 1292):
    x = torch.randn(N, 80, T, device=dev)
    lens = torch.linspace(0.8 * T, T, N).long()
    mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
    res = {}
    keep = M._hip_ok
    for name, (torch_path, ret_att, ac) in (('kernels fp32', (False, True, False)), ('autocast bf16', (False, True, True)),
                                             ('autocast bf16, no att', (False, False, True)), ('torch', (True, True, False))):
        M._hip_ok = (lambda t: False) if torch_path else keep        # (the product module has no switch: the torch yardstick is a patch)
        mha.return_att = ret_att
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=ac):
            res[name] = graph_time(x, mask)
    M._hip_ok, mha.return_att = keep, True
    print('T=%4d frames (att %4.0f MB): ' % (T, H * N * T * T * 4 / 1e6) + ' | '.join('%s %.2f ms' % kv for kv in res.items()), flush=True)
