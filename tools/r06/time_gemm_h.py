"""the feed-forward pair's GEMMs with the hidden tensor stored as bf16 (psnd_linear1x1_fwd_ex / _bwd_ex, io_h) at the config-4 shape
(N = 32, T = 1292, C = 256 -> 1024 -> 256): us per launch; lab builds take PSND_GEMM_ABLATE (1 no fetches, 2 no MFMAs, 4 no epilogue)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
N, T, C, Hd = 32, int(os.environ.get("TT", "1292")), 256, 1024
st = stream_ptr(dev)


def timeit(fn, n=30):
    for _ in range(60): fn()
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


x = torch.randn(N, C, T, device=dev); w1 = torch.randn(Hd, C, device=dev) * 0.05; b1 = torch.zeros(Hd, device=dev)
w2 = torch.randn(C, Hd, device=dev) * 0.05; b2 = torch.zeros(C, device=dev)
LD = int(os.environ.get('LD', '0'))      # row pitch of the bf16 tensors (0: T)
h = torch.empty(N, Hd, LD or T, device=dev, dtype=torch.bfloat16); y = torch.empty(N, C, T, device=dev)
gy = torch.randn(N, C, T, device=dev); gh = torch.empty_like(h); gx = torch.empty_like(x)
L = lib()
S1 = int(L.psnd_linear1x1_wgrad_slabs(N, C, Hd, T)); S2 = int(L.psnd_linear1x1_wgrad_slabs(N, Hd, C, T))
p1 = torch.empty(S1, Hd, C, device=dev); p2 = torch.empty(S2, C, Hd, device=dev); gw1 = torch.empty_like(w1); gw2 = torch.empty_like(w2)
gb1 = torch.empty_like(b1); gb2 = torch.empty_like(b2)
t = {}
t['ffn1 fwd (out bf16)'] = timeit(lambda: check(L.psnd_linear1x1_fwd_ex(ptr(x), ptr(w1), ptr(b1), N, C, Hd, T, 1, 1, 2, LD, ptr(h), st), 'f1'))
t['ffn2 fwd (in bf16)'] = timeit(lambda: check(L.psnd_linear1x1_fwd_ex(ptr(h), ptr(w2), ptr(b2), N, Hd, C, T, 0, 1, 1, LD, ptr(y), st), 'f2'))
t['ffn2 gx (mask, out bf16)'] = timeit(lambda: check(L.psnd_linear1x1_bwd_ex(ptr(gy), None, ptr(h), ptr(w2), N, Hd, C, T, 1, 2, LD, None, ptr(h), ptr(gh), None, None, None, st), 'g2'))
t['ffn2 gw (x bf16)'] = timeit(lambda: check(L.psnd_linear1x1_bwd_ex(ptr(gy), None, ptr(h), ptr(w2), N, Hd, C, T, 1, 2, LD, None, None, None, ptr(gw2), ptr(p2), ptr(gb2), st), 'w2'))
t['ffn1 gx (gy bf16)'] = timeit(lambda: check(L.psnd_linear1x1_bwd_ex(ptr(gh), None, ptr(x), ptr(w1), N, C, Hd, T, 1, 1, LD, None, None, ptr(gx), None, None, None, st), 'g1'))
t['ffn1 gw (gy bf16)'] = timeit(lambda: check(L.psnd_linear1x1_bwd_ex(ptr(gh), None, ptr(x), ptr(w1), N, C, Hd, T, 1, 1, LD, None, None, None, ptr(gw1), ptr(p1), ptr(gb1), st), 'w1'))
print('  '.join('%s %.1f' % kv for kv in t.items()), flush=True)
