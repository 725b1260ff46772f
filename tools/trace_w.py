"""s_memtime stamps of psnd_stft_w.hip (library built with -DPSND_W_DEBUG: tools/build_variant.sh wdbg -DPSND_W_DEBUG): where the
cycles of one tile go, per wave.  s_memtime ticks at 100 MHz on gfx950: 1 tick = 10 ns."""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
dev = torch.device('cuda:0')
trace = torch.zeros(256 * 16 * 16, dtype=torch.int64, device=dev)
os.environ['PSND_W_TRACE_PTR'] = str(trace.data_ptr())
os.environ.setdefault('PSND_W_TRACE_ITER', '4')
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
n, h, T, N = 4096, 1024, 1323000, 32
w = (0.5 - 0.5*np.cos(2*np.pi*np.arange(n)/n)).astype(np.float32)
plan = K.stft_plan(n, w).to(dev)
wav = torch.randn(N, T, device=dev) * 0.07
F = K.frame_count(T, n, h)
mag = torch.empty(N, n // 2 + 1, F, device=dev)
for _ in range(3):
    check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(256, 16, 16)
names = ['top', 'win+r2+swap', 'fft1+tw', 'transpose', 'fft2', 'split', 'B1 (next loads issued)', 'stage', 'B2', 'flush', None, None, None, None, 'B3']
use = [i for i, n in enumerate(names) if n]
for b in (0, 9, 100):
    t = tr[b].astype(np.float64)
    t0 = t[:, 0].min()
    print('block', b, ': stamps relative to the earliest wave top, in ticks (10 ns) - mean over waves / min / max')
    prev = None
    for i in use:
        d = t[:, i] - t0
        print('  %-26s %8.0f %8.0f %8.0f    phase mean %7.0f' % (names[i], d.mean(), d.min(), d.max(), (t[:, i] - t[:, prev]).mean() if prev is not None else 0))
        prev = i
