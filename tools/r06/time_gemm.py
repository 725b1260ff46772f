"""psnd_linear1x1_fwd / _bwd at the config-4 shapes (N = 32, T = 1292, bf16 operands): us per launch and fraction of the bytes / flops roofs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
N, T = 32, 1292
st = stream_ptr(dev)


def timeit(fn, n=30):
    for _ in range(60): fn()
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


for Cin, Cout in ((80, 256), (256, 768), (256, 256), (256, 1024), (1024, 256), (256, 80)):
    x = torch.randn(N, Cin, T, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05; b = torch.zeros(Cout, device=dev)
    y = torch.empty(N, Cout, T, device=dev); gy = torch.randn(N, Cout, T, device=dev); gx = torch.empty_like(x)
    S = int(lib().psnd_linear1x1_wgrad_slabs(N, Cin, Cout, T)); part = torch.empty(S, Cout, Cin, device=dev); gw = torch.empty_like(w); gb = torch.empty_like(b)
    for bf in (1, 0):
        tf = timeit(lambda: check(lib().psnd_linear1x1_fwd(ptr(x), ptr(w), ptr(b), N, Cin, Cout, T, 0, bf, ptr(y), st), 'f'))
        tx = timeit(lambda: check(lib().psnd_linear1x1_bwd(ptr(gy), None, ptr(x), ptr(w), N, Cin, Cout, T, bf, ptr(gx), None, None, None, st), 'bx'))
        tw = timeit(lambda: check(lib().psnd_linear1x1_bwd(ptr(gy), None, ptr(x), ptr(w), N, Cin, Cout, T, bf, None, ptr(gw), ptr(part), ptr(gb), st), 'bw'))
        fl = 2.0 * N * T * Cin * Cout
        by = 4.0 * N * T * (Cin + Cout)
        print('%4d -> %4d %s  fwd %6.1f us (%5.1f TF/s, %4.2f TB/s)  gx %6.1f us  gw %6.1f us (slabs %d)' % (
            Cin, Cout, 'bf16' if bf else 'fp32', tf, fl / tf / 1e6, by / tf / 1e6, tx, tw, S), flush=True)
