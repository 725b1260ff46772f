"""python tools/r05/chk_variant.py lib.so ... - psnd_stft_mag_nfk of each library (and of the in-tree one: `default`) against the (N, K, F)
kernel of the in-tree library and the float64 oracle on a spread of shapes; the output buffer is poisoned with NaN first."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pytorch_sound_amd import kernels as K, _lib
from pytorch_sound_amd.models.transforms import periodic_window
from oracle import features as ofe
dev = torch.device('cuda:0')
P, I64, INT, FL = C.c_void_p, C.c_int64, C.c_int, C.c_float
plan = K.stft_plan(1024, periodic_window('hann', 1024).astype(np.float32)).to(dev)
st = _lib.stream_ptr(dev)
SHAPES = [(1, 1024, 256, 0), (3, 5000, 256, 0), (64, 44100, 256, 0), (1024, 44100, 256, 0), (5, 8192, 256, 1), (7, 30000, 128, 0),
          (2, 600, 256, 0), (33, 44099, 256, 0), (4, 44100, 200, 0), (9, 16000, 512, 0)]
for path in sys.argv[1:]:
    h = _lib.lib() if path == 'default' else C.CDLL(os.path.abspath(path))
    if path != 'default':
        h.psnd_stft_mag_nfk.restype = INT; h.psnd_stft_mag_nfk.argtypes = [P, I64, I64, INT, INT, INT, P, FL, P, P]
    for rep in range(3):
        for (N, T, hop, fr) in SHAPES:
            g = torch.Generator(device='cpu').manual_seed(N * 7 + T + rep)
            x = (0.07 * torch.randn(N, T, generator=g)).to(dev)
            ref = K.stft_forward(x, 1024, hop, plan, framing=fr)['mag'].transpose(1, 2).contiguous()
            o = torch.full_like(ref, float('nan'))
            rc = h.psnd_stft_mag_nfk(_lib.ptr(x), N, T, 1024, hop, fr, _lib.ptr(plan), 0.0, _lib.ptr(o), st)
            assert rc == 0
            torch.cuda.synchronize()
            bad = int(torch.isnan(o).sum())
            d = float((o - ref).abs().max()) / float(ref.max())
            orc = ofe.stft_mag_f64(x[:1].cpu().numpy(), 1024, hop, framing=fr) if N <= 64 else None
            do = None if orc is None else float(np.abs(o[:1].cpu().numpy().transpose(0, 2, 1) - orc).max() / np.abs(orc).max())
            flag = '' if (bad == 0 and d < 1e-5) else '   <<<<<< MISMATCH'
            if rep == 0 or flag:
                print('%-40s N=%d T=%d hop=%d fr=%d  vs nkf %.2e  vs oracle %s  nan %d%s' % (os.path.basename(path), N, T, hop, fr, d, do, bad, flag), flush=True)
