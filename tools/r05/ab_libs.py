"""Same-process, interleaved A/B timing of psnd_stft_mag_nfk (or psnd_stft_fwd with --nkf) across several builds of libpsnd_hip.so:
    python tools/r05/ab_libs.py [--nfft 4096] [--clips 32] [--rounds 12] [--nkf] name=path[:ENV=VAL,...] ...
Every round launches each variant 3 times back to back (HIP events around each launch); the chip's clock state is shared by all the
variants of a round, so medians over rounds compare like with like even on a box that throttles.  `default` = the in-tree library."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K, _lib
from pytorch_sound_amd.models.transforms import periodic_window

args = sys.argv[1:]
opt = {'--nfft': 4096, '--clips': 32, '--rounds': 12}
nkf = False
specs = []
i = 0
while i < len(args):
    if args[i] in opt: opt[args[i]] = int(args[i + 1]); i += 2
    elif args[i] == '--nkf': nkf = True; i += 1
    else: specs.append(args[i]); i += 1
n_fft, N, rounds = opt['--nfft'], opt['--clips'], opt['--rounds']
hop = n_fft // 4
T = 1323000 if n_fft == 4096 else 44100
dev = torch.device('cuda:0')
plan = K.stft_plan(n_fft, periodic_window('hann', n_fft).astype(np.float32)).to(dev)
x = 0.07 * torch.randn(N, T, device=dev)
F, Kb = K.frame_count(T, n_fft, hop), n_fft // 2 + 1
o = torch.empty((N, F, Kb) if not nkf else (N, Kb, F), device=dev)
nbytes = 4 * N * T + 4 * N * Kb * F
P, I64, INT, FL = C.c_void_p, C.c_int64, C.c_int, C.c_float
variants = []
for sp in specs:
    name, _, rest = sp.partition('=')
    path, _, envs = rest.partition(':')
    path = path or 'default'
    h = _lib.lib() if path == 'default' else C.CDLL(os.path.abspath(path))
    if path != 'default':
        h.psnd_stft_mag_nfk.restype = INT; h.psnd_stft_mag_nfk.argtypes = [P, I64, I64, INT, INT, INT, P, FL, P, P]
        h.psnd_stft_fwd.restype = INT; h.psnd_stft_fwd.argtypes = [P, I64, I64, INT, INT, INT, P, FL, P, P, P, P, P]
        if hasattr(h, 'psnd_env_refresh'): h.psnd_env_refresh.restype = None
    env = dict(e.split('=') for e in envs.split(',') if e)
    variants.append((name, h, env))
st = _lib.stream_ptr(dev)
ALLENV = sorted({k for _, _, e in variants for k in e})

def launch(h, env):
    for k in ALLENV: os.environ.pop(k, None)
    os.environ.update(env)
    getattr(h, 'psnd_env_refresh', lambda: None)()
    if nkf: rc = h.psnd_stft_fwd(_lib.ptr(x), N, T, n_fft, hop, 0, _lib.ptr(plan), 0.0, _lib.ptr(o), None, None, None, st)
    else: rc = h.psnd_stft_mag_nfk(_lib.ptr(x), N, T, n_fft, hop, 0, _lib.ptr(plan), 0.0, _lib.ptr(o), st)
    assert rc == 0, rc

times = {n: [] for n, _, _ in variants}
for name, h, env in variants:
    for _ in range(3): launch(h, env)
torch.cuda.synchronize()
for r in range(rounds):
    order = variants if r % 2 == 0 else variants[::-1]
    evs = []
    for _ in range(8): launch(order[0][1], order[0][2])          # untimed: the clock ramps over the first launches after an idle gap
    for name, h, env in order:
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(h, env); e1.record()
            evs.append((name, e0, e1))
    torch.cuda.synchronize()
    for name, e0, e1 in evs: times[name].append(e0.elapsed_time(e1) * 1e3)
for name, _, _ in variants:
    t = np.array(times[name])
    print('%-22s med %.1f  mean %.1f  min %.1f us   frac(med) %.3f' % (name, np.median(t), t.mean(), t.min(), nbytes / np.median(t) / 1e-6 / 8e12), flush=True)
