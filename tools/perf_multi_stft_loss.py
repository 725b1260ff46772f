#!/usr/bin/env python
"""multi_stft_loss forward + backward (models/sound.py:106-133): the HIP path (psnd_stft_fwd/bwd + psnd_stft_loss_*)
against the reference's formulation on torch.stft (rocFFT + ~12 elementwise/reduction launches per resolution), same
GPU, same inputs.  Shapes: config 3 (16 x 8192 samples) and a bandwidth-sized batch."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sound_amd.models import sound as S  # noqa: E402

PARAMS = [(1024, 600, 120), (2048, 1200, 240), (512, 240, 50)]
dev = torch.device('cuda:0')


def torch_loss(pred, target, params, eps=1e-5):
    loss = sc = mg = 0.
    for n_fft, win, hop in params:
        w = torch.hann_window(win, device=pred.device)
        spec = lambda x: torch.stft(x, n_fft, hop, win, w, True, 'reflect', False, True, return_complex=True).abs()  # noqa: E731
        p, t = spec(pred), spec(target)
        n = t.size(1) * t.size(2)
        a = ((t - p).norm(dim=(1, 2)) / t.norm(dim=(1, 2))).mean()
        b = torch.norm(torch.log(t + eps) - torch.log(p + eps), p=1, dim=(1, 2)).mean() / n
        loss, sc, mg = loss + a + b, sc + a, mg + b
    return loss / len(params), sc / len(params), mg / len(params)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    for N, T, iters in ((16, 8192, 50), (256, 16384, 20), (1024, 16384, 10)):
        target = 0.1 * torch.randn(N, T, device=dev)
        pred = (target + 0.01 * torch.randn(N, T, device=dev)).requires_grad_(True)

        def run(f):
            pred.grad = None
            f(pred, target, PARAMS)[0].backward()
        a = timeit(lambda: run(S.multi_stft_loss), iters)
        b = timeit(lambda: run(torch_loss), iters)
        with torch.no_grad():
            fa = timeit(lambda: S.multi_stft_loss(pred, target, PARAMS), iters)
            fb = timeit(lambda: torch_loss(pred, target, PARAMS), iters)
        l1 = S.multi_stft_loss(pred, target, PARAMS)
        l2 = torch_loss(pred, target, PARAMS)
        print('N=%4d T=%6d  fwd+bwd: hip %.3f ms  torch %.3f ms (x%.1f) | fwd only: hip %.3f ms torch %.3f ms | loss %.6f vs %.6f'
              % (N, T, a, b, b / a, fa, fb, float(l1[0]), float(l2[0])), flush=True)


if __name__ == '__main__':
    main()
