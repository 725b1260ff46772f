#!/usr/bin/env python
"""config-4 shaped batch (32 clips, 2-15 s at 22.05 kHz): host zero-padding + H2D of the padded batch (the reference's
pad_collate_fn + .cuda()) against one pinned ragged buffer + psnd_pad_collate in HBM."""
import os
import sys
import time
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sound_amd.data import dataset as D  # noqa: E402

dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
for spread in ('bucketed (13-15 s)', 'unbucketed (2-15 s)'):
    lo = 13 if spread.startswith('bucketed') else 2
    clips = [rs.randn(int(rs.uniform(lo, 15) * 22050)).astype(np.float32) for _ in range(32)]
    items = [[c] for c in clips]

    def host_path():
        x = D.SpeechDataLoader.pad_collate_fn(items)[0].pin_memory()
        return x.to(dev, non_blocking=True)

    def ragged_path():
        return D.ragged_collate_fn(items)[0].to_device(dev, want_mask=True)

    for name, f in (('host pad + H2D', host_path), ('ragged + psnd_pad_collate', ragged_path)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        print('%-22s %-28s %.3f ms/batch  (payload %.1f MB, padded %.1f MB)' % (spread, name, (time.perf_counter() - t0) / 20 * 1e3,
              sum(len(c) for c in clips) * 4e-6, 32 * max(len(c) for c in clips) * 4e-6), flush=True)
rb = D.RaggedBatch(clips)
flat = rb.flat.to(dev)
meta = torch.stack([rb.offs, rb.lens]).to(dev)
from pytorch_sound_amd._lib import lib, check, ptr, stream_ptr  # noqa: E402
N, Tmax = len(rb), int(rb.lens.max())
out, mask = torch.empty(N, Tmax, device=dev), torch.empty(N, Tmax, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    check(lib().psnd_pad_collate(ptr(flat), ptr(meta[0]), ptr(meta[1]), N, Tmax, ptr(out), ptr(mask), stream_ptr(dev)), 'x')
e0.record()
for _ in range(20):
    check(lib().psnd_pad_collate(ptr(flat), ptr(meta[0]), ptr(meta[1]), N, Tmax, ptr(out), ptr(mask), stream_ptr(dev)), 'x')
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('psnd_pad_collate kernel: %.1f us, %.2f TB/s (reads %.1f MB, writes %.1f MB)' % (us, (flat.numel() + 2 * N * Tmax) * 4 / us / 1e6,
      flat.numel() * 4e-6, 2 * N * Tmax * 4e-6))
