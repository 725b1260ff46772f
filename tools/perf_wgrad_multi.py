#!/usr/bin/env python
"""psnd_conv1d_cl_wgrad_multi alone: 24 convs of the config-2 body (256 -> 256 channels, 3 taps, 32 clips x 173 frames), row ranges per conv
from PSND_WGRAD_MULTI_BLOCKS"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd import _lib
from pytorch_sound_amd._lib import lib, stream_ptr, check
dev = torch.device('cuda:0')
N, L, HP, C, k, n = 32, 173, 5, 256, 3, 24
Lp = (L + 2 * HP + 7) // 8 * 8
g = [torch.randn(N, Lp, C, device=dev).to(torch.bfloat16) for _ in range(n)]
x = [torch.randn(N, Lp, C, device=dev).to(torch.bfloat16) for _ in range(n)]
for blocks in [int(a) for a in sys.argv[1:]] or [384, 768, 1152, 1536, 2304, 3072]:
    os.environ['PSND_WGRAD_MULTI_BLOCKS'] = str(blocks)
    S = lib().psnd_conv1d_cl_wgrad_multi_splits(N, Lp, C, C, k, n)
    gw = [torch.empty(S, k, C, C, device=dev) for _ in range(n)]
    gb = [torch.empty(S, C, device=dev) for _ in range(n)]
    arr = (_lib.WgradDesc * n)()
    for i, d in enumerate(arr):
        dil = (1, 1, 3, 1, 5, 1)[i % 6]
        d.g, d.xa, d.gw_part, d.gbias_part, d.off0, d.dstep = g[i].data_ptr(), x[i].data_ptr(), gw[i].data_ptr(), gb[i].data_ptr(), -dil, dil
        d.Ca, d.Cb, d.k, d.splits = C, C, k, S
    run = lambda: check(lib().psnd_conv1d_cl_wgrad_multi(ctypes.addressof(arr), n, N, Lp, stream_ptr(dev)), 'multi')
    for _ in range(3): run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print('target %5d blocks -> %d row ranges per conv, %d workgroups, slabs %.0f MB: %.1f us' % (blocks, S, 16 * S * n, S * n * k * C * C * 4 / 1e6, s.elapsed_time(e) / 20 * 1e3), flush=True)
