#!/usr/bin/env python
"""config-4 block (PE -> MHA(256, 4) -> FFN, batch 32, T frames) forward + backward, a few iterations - rocprofv3 target"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd.models import modules as M
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1292
want_att = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
AC = len(sys.argv) > 3 and sys.argv[3] == 'bf16'
dev = torch.device('cuda:0')
torch.manual_seed(0)
C, H, N = 256, 4, 32
proj = torch.nn.Conv1d(80, C, 1).to(dev)
pe = M.PositionalEncoding(C, 2048).to(dev)
mha = M.MultiHeadAttention(C, H, 0.0).to(dev)
mha.return_att = want_att
ffn = M.PointwiseFeedForward(C, 0.0).to(dev)
params = list(proj.parameters()) + list(mha.parameters()) + list(ffn.parameters())
x = torch.randn(N, 80, T, device=dev)
lens = torch.linspace(0.8 * T, T, N).long()
mask = (torch.arange(T)[None, :] >= lens[:, None]).to(dev)
import time
for it in range(6):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    for p in params: p.grad = None
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=AC):
        h = pe(M._conv1x1(proj, x))
        h, att = mha(h, mask)
        y = ffn(h)
    y.abs().mean().backward()
torch.cuda.synchronize()
print('T=%d want_att=%s: %.2f ms per fwd+bwd' % (T, want_att, (time.perf_counter() - t0) / 4 * 1e3))
