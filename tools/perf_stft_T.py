import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
def hann(n):
    m = np.arange(n); return (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
def run(n, h, N, T, iters=20):
    wav = torch.randn(N, T, device=dev) * 0.07
    plan = K.stft_plan(n, hann(n)).to(dev)
    F = K.frame_count(T, n, h); Kb = n//2+1
    mag = torch.empty(N, Kb, F, device=dev)
    f = lambda: check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e)/iters*1e-3
    b = 4*N*T + 4*N*Kb*F
    print('n=%d N=%d T=%d F=%d: %.1f us %.0f GB/s (%.1f%%) %.0f Mframes/s' % (n, N, T, F, t*1e6, b/t/1e9, b/t/8e10, N*F/t/1e6), flush=True)
for T in [int(x) for x in sys.argv[1].split(',')]:
    run(1024, 256, int(sys.argv[2]) if len(sys.argv) > 2 else 1024, T)
