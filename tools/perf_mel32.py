#!/usr/bin/env python
"""the mel projection at the config-2 step's size (32 clips x 513 bins x 173 frames): forward, and backward through autograd"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_sound_amd import kernels as K
from oracle import features as ofe
dev = torch.device('cuda:0')
n, M = 1024, 80
mplan = K.mel_plan(ofe.mel_filterbank(22050, n, M, 0.0, 8000.0)).to(dev)
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e-3)
    return best
for N in (32, 64):
    mag = torch.rand(N, 513, 173, device=dev)
    t1 = timeit(lambda: K.mel_forward(mag, mplan, M, K.LOG_E, 1e-6, None, -11.5, 6.9))
    m = mag.clone().requires_grad_(True)
    out = K.MelLog.apply(m, mplan, M, K.LOG_E, 1e-6, None, -11.5, 6.9) if hasattr(K, 'MelLog') else None
    print('N=%d: mel forward %.1f us' % (N, t1 * 1e6), flush=True)
    if out is not None:
        g = torch.randn_like(out)
        def bwd():
            m.grad = None
            out.backward(g, retain_graph=True)
        print('      mel backward (autograd node) %.1f us' % (timeit(bwd) * 1e6), flush=True)
