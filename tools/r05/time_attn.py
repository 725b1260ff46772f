"""times the three bf16 attention launches at the config-4 shape (32 clips, 4 heads of 64, T = 1292), one library per process (PSND_LIB):
HIP events around 20 launches each of psnd_mha_fwd, psnd_mha_bwd_parts(2) (key/value gradients) and (4) (query gradients)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
N, H, C, T = 32, 4, 256, 1292
torch.manual_seed(0)
kvq = torch.randn(N, 3 * C, T, device=dev)
gout = torch.randn(N, C, T, device=dev)
out = torch.empty(N, C, T, device=dev)
stats = torch.empty(H * N, T, 2, device=dev)
delta = torch.empty(H * N, T, device=dev)
gkvq = torch.empty_like(kvq)
st = stream_ptr(dev)
def fwd(): check(lib().psnd_mha_fwd(ptr(kvq), None, N, H, C, T, ptr(out), None, ptr(stats), 1, st), 'fwd')
def part(k): check(lib().psnd_mha_bwd_parts(ptr(kvq), None, ptr(out), None, ptr(stats), ptr(gout), None, N, H, C, T, ptr(delta), ptr(gkvq), 1, k, st), 'bwd')
fwd(); part(1)
res = {}
for name, fn in (('fwd', fwd), ('kv', lambda: part(2)), ('q', lambda: part(4))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    res[name] = a.elapsed_time(b) / 20 * 1e3
print(os.environ.get('PSND_LIB', 'default').split('/')[-1], ' '.join('%s %.1f us' % kv for kv in res.items()))
