#!/usr/bin/env python
"""psnd_groupnorm1_fwd/bwd and psnd_softmax_keys_fwd/bwd at config-4 sizes: algorithmic GB/s.
groupnorm fwd: read x, res (2 passes over x+res: stats, apply) + write y = 4*(2*2+1)*NCT ; bwd: gy, x, res (2 passes) + gx.
softmax fwd: in place, one read + one write of (B,T,T) = 8*B*T*T ; bwd: att, gatt read + gscores write = 12*B*T*T."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd import kernels as K
dev = torch.device('cuda:0')
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for N, C, T in [(32, 256, 1292), (32, 1024, 1292)]:
    x = torch.randn(N, C, T, device=dev, requires_grad=True); r = torch.randn(N, C, T, device=dev, requires_grad=True)
    g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    y = K.GroupNorm1.apply(x, r, g, b, 1e-5, True)
    gy = torch.randn_like(y)
    tf = timeit(lambda: K.GroupNorm1.apply(x, r, g, b, 1e-5, True))
    tb = timeit(lambda: torch.autograd.grad(K.GroupNorm1.apply(x, r, g, b, 1e-5, True), (x, r, g, b), gy)) - tf
    nb = N * C * T * 4
    print('groupnorm1 N=%d C=%d T=%d: fwd %.1f us (%.0f GB/s of 5 passes; %.0f GB/s minimum 3 passes) | bwd %.1f us' % (N, C, T, tf * 1e6, 5 * nb / tf / 1e9, 3 * nb / tf / 1e9, tb * 1e6))
for B, T in [(128, 690), (128, 1292)]:
    s_ = torch.randn(B, T, T, device=dev, requires_grad=True)
    mask = torch.zeros(B, T, dtype=torch.uint8, device=dev); mask[:, int(0.9 * T):] = 1
    a = K.SoftmaxKeys.apply(s_, mask, 0.125)
    ga = torch.randn_like(a)
    tf = timeit(lambda: K.SoftmaxKeys.apply(s_, mask, 0.125))
    tb = timeit(lambda: torch.autograd.grad(K.SoftmaxKeys.apply(s_, mask, 0.125), (s_,), ga)) - tf
    nb = B * T * T * 4
    print('softmax_keys B=%d T=%d (%.0f MB): fwd incl. the clone %.1f us (%.0f GB/s of 4 passes) | bwd %.1f us (%.0f GB/s of 3 passes)' % (B, T, nb / 1e6, tf * 1e6, 4 * nb / tf / 1e9, tb * 1e6, 3 * nb / tb / 1e9))
