#!/usr/bin/env python
"""BASELINE config 3 as a Trainer step on ONE GPU: 16 clips x 8192 samples (22.05 kHz), HiFi-GAN framing mel (1024/256/80),
hifi_gan_v1 generator, loss = L1(mel(G(mel(x))), mel(x)) [+ multi_stft_loss with --msl], Adam.
  hip   : Generator on the CL conv kernels, mel front end on psnd_stft_*/psnd_mel_*, psnd_adam_step, hipGraph replay
  torch : the same model on the library convolutions (fp32), torch.stft front end, torch fused Adam, eager
Prints ms/step and audio-s/s for both."""
import os
import sys
import tempfile
import time
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sound_amd.models import build_model  # noqa: E402
from pytorch_sound_amd.models.vocoders import hifi_gan  # noqa: E402,F401
from pytorch_sound_amd.interface.hifi_gan import MelSpectrogram  # noqa: E402
from pytorch_sound_amd.models.sound import multi_stft_loss  # noqa: E402
from pytorch_sound_amd.trainer import Trainer, LogType  # noqa: E402
from pytorch_sound_amd import optim as poptim  # noqa: E402
from pytorch_sound_amd import kernels as K  # noqa: E402

dev = torch.device('cuda:0')
N, T, SR = 16, 8192, 22050
MSL = '--msl' in sys.argv
PARAMS = [(1024, 600, 120), (2048, 1200, 240), (512, 240, 50)]


def make(hip):
    torch.manual_seed(0)
    gen = build_model('hifi_gan_v1').to(dev)
    gen.use_cl = hip
    mel = MelSpectrogram().to(dev)
    win = torch.hann_window(1024, device=dev)

    def mel_torch(w):
        p = (1024 - 256) // 2
        w = F.pad(w.unsqueeze(1), (p, p), mode='reflect').squeeze(1)
        s = torch.stft(w, 1024, 256, 1024, win, center=False, return_complex=True)
        mag = torch.sqrt(s.real ** 2 + s.imag ** 2 + 1e-9)
        return torch.log(torch.clamp(torch.matmul(mel.mel_filter, mag), min=1e-5))

    feat = mel if hip else mel_torch

    class Step(Trainer):
        def prepare(self, wav):
            with torch.no_grad():
                return wav, feat(wav)

        def forward(self, wav, m, is_logging=False):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=m.is_cuda):
                y = self.model(m)
            y = y.float().squeeze(1)
            loss = (K.l1_loss if hip else F.l1_loss)(feat(y), m)
            if MSL:
                loss = loss + multi_stft_loss(y, wav, PARAMS)[0]
            return loss, {'loss': (loss, LogType.SCALAR)}

    opt = poptim.Adam(gen.parameters(), lr=2e-4, betas=(0.8, 0.99)) if hip else \
        torch.optim.Adam(gen.parameters(), lr=2e-4, betas=(0.8, 0.99), fused=True)
    g = torch.Generator().manual_seed(1)
    pool = [(0.07 * torch.randn(N, T, generator=g)).clamp(-1, 1).to(dev) for _ in range(4)]
    pool = [(w,) for w in pool]
    tr = Step(gen, opt, pool, pool, max_step=10 ** 9, valid_max_step=1, save_interval=10 ** 9, log_interval=10 ** 9,
              save_dir=tempfile.mkdtemp(prefix='psnd_c3_'), seed=1)
    tr.graph_steps = hip
    gen.train()
    return tr


def run(tr, steps=30, warm=8):
    s = 0
    for _ in range(warm):
        s += 1
        tr.step = s
        tr.train(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s += 1
        tr.step = s
        tr.train(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


LEGS = (('hip', True),) if '--hip-only' in sys.argv else (('hip', True), ('torch', False))
for name, hip in LEGS:
    ms = run(make(hip))
    print('config 3%s, %-5s: %.2f ms/step = %.1f k audio-s/s' % (' + multi_stft_loss' if MSL else '', name, ms, N * T / SR / ms), flush=True)
