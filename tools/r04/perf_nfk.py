"""HIP-event timing of psnd_stft_mag_nfk against psnd_stft_fwd (NKF): n_fft 1024 (1024 clips x 2 s, 64 clips) and config 5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window

dev = torch.device('cuda:0')


def run(n_fft, hop, N, T, reps=30):
    w = periodic_window('hann', n_fft).astype(np.float32)
    plan = K.stft_plan(n_fft, w).to(dev)
    x = (0.07 * torch.randn(N, T, device=dev))
    F, Kb = K.frame_count(T, n_fft, hop), n_fft // 2 + 1
    nbytes = 4 * N * T + 4 * N * Kb * F
    o1 = torch.empty((N, Kb, F), device=dev)
    o2 = torch.empty((N, F, Kb), device=dev)
    res = {}
    for name, fn in (('nkf', lambda: K.stft_forward(x, n_fft, hop, plan, out_mag=o1)), ('nfk', lambda: K.stft_mag_nfk(x, n_fft, hop, plan, out=o2))):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = np.array(ts)
        res[name] = (ts.mean(), ts.min())
        print('n_fft %d hop %d N %d T %d  %s: mean %.1f us min %.1f us  -> %.3f of 8 TB/s (mean)  [%.1f MB]' % (
            n_fft, hop, N, T, name, ts.mean(), ts.min(), nbytes / ts.mean() / 1e-6 / 8e12, nbytes / 1e6), flush=True)
    return res


if __name__ == '__main__':
    which = sys.argv[1:] or ['1024', '4096']
    if '1024' in which:
        run(1024, 256, 1024, 44100)
        run(1024, 256, 64, 44100)
    if '4096' in which:
        run(4096, 1024, 32, 1323000)
        run(4096, 1024, 16, 1323000)
