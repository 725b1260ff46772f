#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference (read-only, /root/reference).

Runs only in the build container (the reference never travels to the GPU box; the
fixtures and this script do).  Third-party packages the reference imports but this
image lacks (torchaudio, librosa, tensorboardX, unidecode, inflect) are stubbed HERE,
in this tool - never in reference code.  ``librosa.filters.mel`` is stubbed with the
oracle's restatement (oracle/features.py:mel_filterbank) so the fixtures pin the
matmul / log / clamp logic but NOT the filterbank values (DESIGN.md "parity unpinned"
for the filter values).

Fixtures are data only: seeded inputs and the reference's outputs.
"""
import os
import sys
import types
import tempfile
import logging
import io
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('PSND_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import torch  # noqa: E402
import scipy.signal  # noqa: E402
from oracle import features as ofe  # noqa: E402


def install_stubs():
    if not hasattr(scipy.signal, 'kaiser'):
        scipy.signal.kaiser = scipy.signal.windows.kaiser

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ta = mod('torchaudio')
    ta.functional = mod('torchaudio.functional')
    ta.transforms = mod('torchaudio.transforms', MelSpectrogram=object)
    lr = mod('librosa')
    lr.util = mod('librosa.util', pad_center=lambda w, size: ofe.pad_center(np.asarray(w), size))
    lr.filters = mod('librosa.filters',
                     mel=lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None:
                     ofe.mel_filterbank(sr, n_fft, n_mels, fmin, fmax))

    class _Writer:
        def __init__(self, *a, **k):
            self.scalars = []

        def add_scalar(self, tag, value, global_step=None):
            self.scalars.append((tag, float(value), global_step))

        def add_image(self, *a, **k):
            pass

        add_audio = add_text = add_image

    mod('tensorboardX', SummaryWriter=_Writer)
    mod('unidecode', unidecode=lambda s: s)
    # imported at module level by pytorch_sound/utils/sound.py (never called by the fixtures)
    mod('pretty_midi')
    mod('pyworld')
    mod('pysndfx', AudioEffectsChain=object)

    class _Engine:
        def number_to_words(self, *a, **k):
            return ''

    mod('inflect', engine=lambda: _Engine())
    # no GPU in the build container
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.manual_seed = lambda *a, **k: None


def t2n(d):
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def seeded_wav(seed, N, T, sr=22050):
    g = np.random.RandomState(seed)
    t = np.arange(T) / sr
    w = 0.0708 * g.randn(N, T) + 0.1 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t + 0.3)
    return np.clip(w, -1, 1).astype(np.float32)


def gen_stft():
    from pytorch_sound.models.transforms import STFT
    out = {}
    cases = [('n1024_h256', 1024, 256, None, (2, 4200)),
             ('n1024_h256_w800', 1024, 256, 800, (2, 3000)),
             ('n512_h128', 512, 128, None, (3, 1111)),
             ('n256_h64_w200', 256, 64, 200, (2, 700)),
             ('n2048_h512', 2048, 512, None, (1, 6000)),
             ('n4096_h1024', 4096, 1024, None, (1, 9000))]
    for i, (name, n, h, w, shape) in enumerate(cases):
        m = STFT(filter_length=n, hop_length=h, win_length=w)
        wav = seeded_wav(100 + i, *shape)
        mag, phase = m.transform(torch.from_numpy(wav))
        out[name + '/wav'] = wav
        out[name + '/mag'] = mag.numpy()
        out[name + '/phase'] = phase.numpy()
        out[name + '/params'] = np.array([n, h, w or n], np.int64)
        fb = m.forward_basis.numpy()[:, 0, :]
        K = n // 2 + 1
        rows = [0, 1, K - 1, K, K + 1, 2 * K - 1]
        out[name + '/basis_rows_idx'] = np.array(rows, np.int64)
        out[name + '/basis_rows'] = fb[rows]
        out[name + '/square_window'] = m.square_window.numpy()
        if n <= 1024:
            # G3: inverse (next row) on the consistent spectrum
            rec = m.inverse(mag, phase)
            out[name + '/inverse'] = rec.numpy()
    # backward through STFT.transform magnitude (autograd through conv1d + sqrt)
    m = STFT(1024, 256)
    wav = torch.from_numpy(seeded_wav(7, 2, 2500)).requires_grad_(True)
    mag, _ = m.transform(wav)
    g = torch.from_numpy(np.random.RandomState(8).randn(*mag.shape).astype(np.float32))
    (mag * g).sum().backward()
    out['bwd/wav'] = wav.detach().numpy()
    out['bwd/gmag'] = g.numpy()
    out['bwd/gwav'] = wav.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'stft.npz'), **out)


def gen_impulse():
    """G2: bit-exact frame indexing.  Impulse at sample p -> set of (frame, tap) hit."""
    from pytorch_sound.models.transforms import STFT
    out = {}
    n, h, T = 64, 16, 200
    m = STFT(n, h)
    # expose taps: replace the basis by identity taps so conv output row r = tap r value
    eye = torch.eye(n).unsqueeze(1)
    pos = [0, 1, 15, 16, 31, 32, 33, 100, T - 2, T - 1]
    for framing in (0, 1):
        hits = []
        for p in pos:
            x = torch.zeros(1, T)
            x[0, p] = 1.0
            if framing == 0:
                # reference centre framing: reuse its own pad + strided conv (transforms.py:55-66)
                m.forward_basis = eye
                xx = x.unsqueeze(1).unsqueeze(1)
                xx = torch.nn.functional.pad(xx, (m.pad_amount, m.pad_amount, 0, 0), mode='reflect').squeeze(1)
                y = torch.nn.functional.conv1d(xx, eye, stride=h)                       # 1,n,F
            else:
                # Audio2Mel framing (transforms.py:352-353) with a rectangular "DFT": taps via unfold
                pp = (n - h) // 2
                xx = torch.nn.functional.pad(x.unsqueeze(1), (pp, pp), 'reflect').squeeze(1)
                y = xx.unfold(-1, n, h).transpose(1, 2)                                  # 1,n,F
            hits.append(y[0].numpy().astype(np.int8))
        out['framing%d/taps' % framing] = np.stack(hits)                                  # P,n,F
    out['pos'] = np.array(pos, np.int64)
    out['params'] = np.array([n, h, T], np.int64)
    np.savez_compressed(os.path.join(OUT, 'impulse.npz'), **out)


def gen_logmel():
    from pytorch_sound.models.transforms import LogMelSpectrogram
    out = {}
    cases = [('default', dict(sample_rate=22050, mel_size=80, n_fft=1024, win_length=1024, hop_length=256,
                              min_db=-50, max_db=30, mel_min=0, mel_max=8000), (2, 5000)),
             ('noclamp', dict(sample_rate=22050, mel_size=80, n_fft=1024, win_length=1024, hop_length=256,
                              min_db=None, max_db=None, mel_min=0., mel_max=None), (1, 3000)),
             ('zero_db_disables', dict(sample_rate=16000, mel_size=40, n_fft=512, win_length=512, hop_length=128,
                                       min_db=0, max_db=0, mel_min=50., mel_max=7000.), (2, 2000)),
             ('silence', dict(sample_rate=22050, mel_size=80, n_fft=1024, win_length=1024, hop_length=256,
                              min_db=-50, max_db=30, mel_min=0, mel_max=8000), (1, 2048))]
    for i, (name, kw, shape) in enumerate(cases):
        m = LogMelSpectrogram(**kw)
        wav = seeded_wav(200 + i, *shape, sr=kw['sample_rate'])
        if name == 'silence':
            wav[:] = 0
        if name == 'default':
            wav[1] *= 40.0          # drive some mels above max_db clamp
        mel = m(torch.from_numpy(wav))
        out[name + '/wav'] = wav
        out[name + '/mel'] = mel.numpy()
        out[name + '/mel_filter'] = m.mel_filter.numpy()
        out[name + '/kw'] = np.array([kw['sample_rate'], kw['mel_size'], kw['n_fft'], kw['win_length'],
                                      kw['hop_length'],
                                      np.nan if kw['min_db'] is None else kw['min_db'],
                                      np.nan if kw['max_db'] is None else kw['max_db'],
                                      kw['mel_min'], np.nan if kw['mel_max'] is None else kw['mel_max']], np.float64)
    # backward to the waveform through mel/log/clamp
    m = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0, 8000)
    wav = torch.from_numpy(seeded_wav(300, 2, 2600)).requires_grad_(True)
    mel = m(wav)
    g = torch.from_numpy(np.random.RandomState(9).randn(*mel.shape).astype(np.float32))
    (mel * g).sum().backward()
    out['bwd/wav'] = wav.detach().numpy()
    out['bwd/gmel'] = g.numpy()
    out['bwd/gwav'] = wav.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'logmel.npz'), **out)


def gen_torch_stft():
    """a5/a6: STFTTorchAudio / Audio2Mel / interface MelSpectrogram cannot run on torch>=2
    (no return_complex).  Their semantics = view_as_real(torch.stft(..., return_complex=True));
    torch.stft of THIS torch is the value/framing oracle."""
    out = {}
    wav = seeded_wav(400, 2, 4096)
    win = torch.hann_window(1024)
    s = torch.stft(torch.from_numpy(wav), 1024, 256, 1024, win, True, 'reflect', False, True, return_complex=True)
    out['center/wav'] = wav
    out['center/re'] = s.real.numpy()
    out['center/im'] = s.imag.numpy()
    # win_length < n_fft (torch centre-pads the window like the reference's pad_center)
    win2 = torch.hann_window(600)
    s2 = torch.stft(torch.from_numpy(wav), 1024, 256, 600, win2, True, 'reflect', False, True, return_complex=True)
    out['center_w600/re'] = s2.real.numpy()
    out['center_w600/im'] = s2.imag.numpy()
    # HiFi-GAN framing
    p = (1024 - 256) // 2
    x = torch.nn.functional.pad(torch.from_numpy(wav).unsqueeze(1), (p, p), 'reflect').squeeze(1)
    s3 = torch.stft(x, 1024, 256, 1024, win, center=False, return_complex=True)
    out['hifigan/re'] = s3.real.numpy()
    out['hifigan/im'] = s3.imag.numpy()
    melW = ofe.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    out['hifigan/mel_filter'] = melW
    mag9 = torch.sqrt(torch.view_as_real(s3).pow(2).sum(-1) + 1e-9)
    out['hifigan/interface_mel'] = torch.log(torch.clamp(torch.matmul(torch.from_numpy(melW), mag9), min=1e-5)).numpy()
    melW2 = ofe.mel_filterbank(22050, 1024, 80, 0.0, None)
    out['hifigan/audio2mel_filter'] = melW2
    mag0 = torch.sqrt(s3.real ** 2 + s3.imag ** 2)
    out['hifigan/audio2mel'] = torch.log10(torch.clamp(torch.matmul(torch.from_numpy(melW2), mag0), min=1e-5)).numpy()
    np.savez_compressed(os.path.join(OUT, 'torch_stft.npz'), **out)


def gen_torch_stft_modules():
    """a5/a6/f1 at MODULE level: the reference's STFTTorchAudio (forward / transform with its differentiable phase / inverse),
    Audio2Mel and interface.hifi_gan.MelSpectrogram (is_center both ways), run from the imported reference.
    Environment stubs (this tool only): torch.stft / torch.istft get the torch-1.x calling convention the reference was written
    against (real (..., 2) views instead of complex tensors)."""
    real_stft, real_istft = torch.stft, torch.istft

    def stft_1x(*a, **k):
        if 'return_complex' in k:
            return real_stft(*a, **k)
        return torch.view_as_real(real_stft(*a, return_complex=True, **k))

    def istft_1x(x, *a, **k):
        if not torch.is_complex(x):
            x = torch.view_as_complex(x.contiguous())
        return real_istft(x, *a, **k)

    torch.stft, torch.istft = stft_1x, istft_1x
    try:
        from pytorch_sound.models import transforms as rt
        from pytorch_sound.interface import hifi_gan as rh
        out = {}
        wav = seeded_wav(410, 2, 3000)
        out['wav'] = wav
        for tag, kw in (('w1024', dict(filter_length=1024, hop_length=256)),
                        ('w600', dict(filter_length=1024, hop_length=256, win_length=600, n_fft=1024)),
                        ('n512', dict(filter_length=512, hop_length=128))):
            m = rt.STFTTorchAudio(**kw)
            x = torch.from_numpy(wav).requires_grad_(True)
            re, im = m(x)
            mag, ph = m.transform(x)
            g = np.random.RandomState(411).randn(2, *mag.shape).astype(np.float32)
            (mag * torch.from_numpy(g[0]) + ph * torch.from_numpy(g[1])).sum().backward()
            out[tag + '/re'], out[tag + '/im'] = re.detach().numpy(), im.detach().numpy()
            out[tag + '/mag'], out[tag + '/phase'] = mag.detach().numpy(), ph.detach().numpy()
            out[tag + '/g'] = g
            out[tag + '/gwav'] = x.grad.numpy()
            # inverse of the module's own analysis, and of an arbitrary (inconsistent) magnitude / phase pair
            out[tag + '/inverse'] = m.inverse(mag.detach(), ph.detach()).numpy()
            rs = np.random.RandomState(412)
            amag = np.abs(rs.randn(*mag.shape)).astype(np.float32)
            aph = rs.uniform(-np.pi, np.pi, mag.shape).astype(np.float32)
            # real DC / Nyquist bins, as any spectrum of a real signal has
            aph[:, 0] = 0
            aph[:, -1] = 0
            out[tag + '/amag'], out[tag + '/aphase'] = amag, aph
            out[tag + '/ainverse'] = m.inverse(torch.from_numpy(amag), torch.from_numpy(aph)).numpy()
        a2m = rt.Audio2Mel()
        out['audio2mel/out'] = a2m(torch.from_numpy(wav).unsqueeze(1)).numpy()
        a2m2 = rt.Audio2Mel(n_fft=512, hop_length=128, win_length=512, sampling_rate=16000, n_mel_channels=40, mel_fmin=50.0, mel_fmax=7000.0)
        out['audio2mel_b/out'] = a2m2(torch.from_numpy(wav).unsqueeze(1)).numpy()
        ms = rh.MelSpectrogram()
        out['interface/out'] = ms(torch.from_numpy(wav)).numpy()
        out['interface/out_center'] = ms(torch.from_numpy(wav), is_center=True).numpy()
        assert int(ms.pad_size) == 384
    finally:
        torch.stft, torch.istft = real_stft, real_istft
    np.savez_compressed(os.path.join(OUT, 'torch_stft_modules.npz'), **out)


def gen_modules():
    from pytorch_sound.models.modules import MultiHeadAttention, PointwiseFeedForward, PositionalEncoding
    out = {}
    torch.manual_seed(1234)
    C, H, T, N = 16, 4, 10, 3
    mha = MultiHeadAttention(C, H, 0.0)
    with torch.no_grad():
        mha.layernorm.weight.copy_(torch.randn(C) * 0.3 + 1)
        mha.layernorm.bias.copy_(torch.randn(C) * 0.1)
    for k, v in t2n(mha.state_dict()).items():
        out['mha/sd/' + k] = v
    x = torch.randn(N, C, T)
    lens = [10, 7, 4]
    mask = torch.zeros(N, T, dtype=torch.bool)
    for i, L in enumerate(lens):
        mask[i, L:] = True
    g = torch.randn(N, C, T)
    for tag, mk in (('nomask', None), ('mask', mask)):
        xi = x.clone().requires_grad_(True)
        mha.zero_grad()
        y, att = mha(xi, mk)
        (y * g).sum().backward()
        out['mha/%s/y' % tag] = y.detach().numpy()
        out['mha/%s/att' % tag] = att.detach().numpy()
        out['mha/%s/gx' % tag] = xi.grad.numpy()
        for k, p in mha.named_parameters():
            out['mha/%s/g/%s' % (tag, k)] = p.grad.numpy()
    out['mha/x'] = x.numpy()
    out['mha/g'] = g.numpy()
    out['mha/mask'] = mask.numpy()

    ffn = PointwiseFeedForward(C, 0.0)
    with torch.no_grad():
        ffn.layernorm.weight.copy_(torch.randn(C) * 0.3 + 1)
        ffn.layernorm.bias.copy_(torch.randn(C) * 0.1)
    for k, v in t2n(ffn.state_dict()).items():
        out['ffn/sd/' + k] = v
    xi = x.clone().requires_grad_(True)
    y = ffn(xi)
    (y * g).sum().backward()
    out['ffn/y'] = y.detach().numpy()
    out['ffn/gx'] = xi.grad.numpy()
    for k, p in ffn.named_parameters():
        out['ffn/g/' + k] = p.grad.numpy()

    pe = PositionalEncoding(C, 32)
    out['pe/table'] = pe.pe.numpy()
    out['pe/y'] = pe(x).numpy()
    np.savez_compressed(os.path.join(OUT, 'modules.npz'), **out)


def gen_hifigan():
    from argparse import Namespace
    from pytorch_sound.models import build_model
    from pytorch_sound.models.vocoders import hifi_gan  # noqa: F401  (registers)
    out = {}
    small = {
        'tiny1': Namespace(resblock='1', upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4],
                           upsample_initial_channel=32, resblock_kernel_sizes=[3, 7],
                           resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]]),
        'tiny2': Namespace(resblock='2', upsample_rates=[8, 4], upsample_kernel_sizes=[16, 8],
                           upsample_initial_channel=16, resblock_kernel_sizes=[3, 5],
                           resblock_dilation_sizes=[[1, 2], [2, 6]]),
    }
    for name, h in small.items():
        torch.manual_seed(77)
        gen = hifi_gan.Generator(h)
        with torch.no_grad():          # make weight_g differ from ||v|| so weight-norm is exercised
            for k, p in gen.named_parameters():
                if k.endswith('weight_g'):
                    p.mul_(1.0 + 0.2 * torch.rand_like(p))
                if k.endswith('bias'):
                    p.add_(0.05 * torch.randn_like(p))
        for k, v in t2n(gen.state_dict()).items():
            out['%s/sd/%s' % (name, k)] = v
        x = torch.randn(2, 80, 6).requires_grad_(True)
        y = gen(x)
        g = torch.randn_like(y)
        (y * g).sum().backward()
        out[name + '/x'] = x.detach().numpy()
        out[name + '/y'] = y.detach().numpy()
        out[name + '/g'] = g.numpy()
        out[name + '/gx'] = x.grad.numpy()
        for k, p in gen.named_parameters():
            out['%s/g/%s' % (name, k)] = p.grad.numpy()
    # registered archs: parameter counts, key lists and output shapes only (weights too big to ship)
    for arch in ('hifi_gan_v1', 'hifi_gan_v2', 'hifi_gan_v3'):
        torch.manual_seed(5)
        gen = build_model(arch)
        out[arch + '/n_params'] = np.array(sum(p.numel() for p in gen.parameters()), np.int64)
        out[arch + '/keys'] = np.array(sorted(gen.state_dict().keys()))
        out[arch + '/shapes'] = np.array([str(tuple(v.shape)) for k, v in sorted(gen.state_dict().items())])
        with torch.no_grad():
            out[arch + '/out_shape'] = np.array(gen(torch.randn(1, 80, 4)).shape, np.int64)
    # shipped v2 checkpoint (asset = data): output on a seeded mel
    ck = os.path.join(REF, 'assets', 'vocoders', 'hifi_gan_v2.pt')
    if os.path.exists(ck):
        gen = build_model('hifi_gan_v2')
        sd = torch.load(ck, map_location='cpu', weights_only=False)['generator']
        gen.load_state_dict(sd)
        torch.manual_seed(11)
        mel = torch.randn(1, 80, 8) * 2 - 5
        with torch.no_grad():
            y = gen(mel)
        out['v2ckpt/mel'] = mel.numpy()
        out['v2ckpt/y'] = y.numpy()
        out['v2ckpt/keys'] = np.array(sorted(sd.keys()))
    np.savez_compressed(os.path.join(OUT, 'hifigan.npz'), **out)


def gen_trainer():
    import pytorch_sound.trainer as rt
    from pytorch_sound.utils.commons import LOGGER
    out = {}
    buf = io.StringIO()
    hdl = logging.StreamHandler(buf)
    hdl.setFormatter(logging.Formatter('%(message)s'))
    LOGGER.addHandler(hdl)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1 = torch.nn.Linear(8, 16)
            self.l2 = torch.nn.Linear(16, 1)

        def forward(self, x):
            return self.l2(torch.tanh(self.l1(x)))

    class T(rt.Trainer):
        def forward(self, x, y, is_logging=False):
            o = self.model(x)
            loss = torch.nn.functional.mse_loss(o, y)
            if self.poison and self.step == 3 and self.model.training:
                loss = loss * float('nan')
            return loss, {'loss': (loss.item(), rt.LogType.SCALAR), 'mae': ((o - y).abs().mean().item(), rt.LogType.SCALAR)}

    def data(seed, n):
        g = torch.Generator().manual_seed(seed)
        return [(torch.randn(4, 8, generator=g), torch.randn(4, 1, generator=g)) for _ in range(n)]

    torch.manual_seed(2024)
    net = Net()
    init_sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k, v in t2n(init_sd).items():
        out['init/' + k] = v
    train, valid = data(1, 5), data(2, 3)
    for i, (x, y) in enumerate(train):
        out['train/x%d' % i], out['train/y%d' % i] = x.numpy(), y.numpy()
    for i, (x, y) in enumerate(valid):
        out['valid/x%d' % i], out['valid/y%d' % i] = x.numpy(), y.numpy()

    with tempfile.TemporaryDirectory() as d:
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
        sch = torch.optim.lr_scheduler.StepLR(opt, 2, 0.5)
        T.poison = True
        tr = T(net, opt, train, valid, max_step=6, valid_max_step=3, save_interval=3, log_interval=2,
               save_dir=d, save_prefix='exp', grad_clip=0.5, grad_norm=1.0, scheduler=sch, seed=99)
        best = tr.run()
        out['run/best_valid'] = np.array(float(best), np.float64)
        for k, v in t2n(net.state_dict()).items():
            out['final/' + k] = v
        files = []
        for r, _, fs in os.walk(d):
            for f in fs:
                files.append(os.path.relpath(os.path.join(r, f), d))
        out['run/files'] = np.array(sorted(files))
        ck = torch.load(os.path.join(d, 'models', 'exp', 'Net', 'step_000006.chkpt'), weights_only=False)
        out['run/ckpt_keys'] = np.array(sorted(ck.keys()))
        out['run/ckpt_model_keys'] = np.array(sorted(ck['model'].keys()))
        out['run/ckpt_step'] = np.array(ck['step'], np.int64)
        out['run/ckpt_seed'] = np.array(ck['seed'], np.int64)
        out['run/writer_scalars'] = np.array(['%s|%.9g|%s' % s for s in tr.writer.scalars])
        # resume: a fresh trainer on the same dir continues from step 6
        net2 = Net()
        opt2 = torch.optim.SGD(net2.parameters(), lr=0.05, momentum=0.9)
        sch2 = torch.optim.lr_scheduler.StepLR(opt2, 2, 0.5)
        T.poison = False
        tr2 = T(net2, opt2, train, valid, max_step=8, valid_max_step=3, save_interval=3, log_interval=2,
                save_dir=d, save_prefix='exp', grad_clip=0.5, grad_norm=1.0, scheduler=sch2, seed=5)
        out['resume/step'] = np.array(tr2.step, np.int64)
        out['resume/seed'] = np.array(tr2.seed, np.int64)
        tr2.run()
        for k, v in t2n(net2.state_dict()).items():
            out['resume_final/' + k] = v
    LOGGER.removeHandler(hdl)
    lines = [ln for ln in buf.getvalue().splitlines() if 'checkpoint' not in ln and 'No any checkpoint' not in ln]
    out['run/log_lines'] = np.array(lines)
    np.savez_compressed(os.path.join(OUT, 'trainer.npz'), **out)


def gen_sound():
    """f2: models/sound.py PreEmphasis + multi_stft_loss, run from the imported reference on CPU.
    Environment stubs (this tool only): `.cuda()` is the identity (install_stubs) and torch.stft gets the torch-1.x
    default the reference was written against (no return_complex argument -> real view of the complex result)."""
    import contextlib
    real_stft = torch.stft

    def stft_1x(*a, **k):
        if 'return_complex' in k:
            return real_stft(*a, **k)
        return torch.view_as_real(real_stft(*a, return_complex=True, **k))

    torch.stft = stft_1x
    try:
        from pytorch_sound.models import sound as rs
        out = {}
        x = torch.from_numpy(seeded_wav(500, 3, 2048)).unsqueeze(1).requires_grad_(True)
        pe = rs.PreEmphasis(0.97)
        y = pe(x)
        gy = torch.from_numpy(np.random.RandomState(501).randn(*y.shape).astype(np.float32))
        (y * gy).sum().backward()
        out['preemph/x'] = x.detach().numpy()
        out['preemph/y'] = y.detach().numpy()
        out['preemph/gy'] = gy.numpy()
        out['preemph/gx'] = x.grad.numpy()
        # VolNormConv forward / reverse and the tanh-RNN InversePreEmphasis (host-side / inference-side utilities)
        vn = rs.VolNormConv(400, 160, -11.5)
        w = torch.from_numpy(seeded_wav(506, 1, 4000))
        nw = vn.forward(w)
        out['volnorm/wav'] = w.numpy()
        out['volnorm/norm'] = nw.numpy()
        out['volnorm/std'] = vn.std_buffer.numpy()
        out['volnorm/reverse'] = vn.reverse(nw).numpy()
        ipe = rs.InversePreEmphasis(0.97)
        with torch.no_grad():
            out['ipreemph/y'] = ipe(y.detach()[:, :, :256]).numpy()
        # the three MelGAN / Parallel-WaveGAN style resolutions (n_fft, window size, hop size)
        params = [(1024, 600, 120), (2048, 1200, 240), (512, 240, 50)]
        target = torch.from_numpy(seeded_wav(502, 3, 8192))
        pred = (target + 0.05 * torch.from_numpy(seeded_wav(503, 3, 8192))).requires_grad_(True)
        with contextlib.redirect_stdout(io.StringIO()) as so:
            loss, sc, mag = rs.multi_stft_loss(pred, target, params)
        loss.backward()
        out['msl/params'] = np.asarray(params, np.int64)
        out['msl/pred'] = pred.detach().numpy()
        out['msl/target'] = target.numpy()
        out['msl/loss'] = np.asarray([float(loss), float(sc), float(mag)], np.float64)
        out['msl/gpred'] = pred.grad.numpy()
        out['msl/stdout'] = np.asarray(so.getvalue())
        # a second, small case: one resolution, batch 1, eps given
        t2 = torch.from_numpy(seeded_wav(504, 1, 1000))
        p2 = (0.5 * t2 + 0.1 * torch.from_numpy(seeded_wav(505, 1, 1000))).requires_grad_(True)
        with contextlib.redirect_stdout(io.StringIO()):
            l2, s2, m2 = rs.multi_stft_loss(p2, t2, [(256, 200, 64)], eps=1e-3)
        l2.backward()
        out['msl1/pred'] = p2.detach().numpy()
        out['msl1/target'] = t2.numpy()
        out['msl1/loss'] = np.asarray([float(l2), float(s2), float(m2)], np.float64)
        out['msl1/gpred'] = p2.grad.numpy()
    finally:
        torch.stft = real_stft
    np.savez_compressed(os.path.join(OUT, 'sound.npz'), **out)


def gen_data():
    """f3: data/dataset.py - pad_collate_fn, BucketRandomBatchSampler (seeded np.random), SpeechDataset crops."""
    import pandas as pd
    from pytorch_sound.data import dataset as rd
    from pytorch_sound.data.meta import MetaFrame, MetaType
    out = {}
    rs = np.random.RandomState(900)
    lens = [700, 1000, 333, 1000, 512]
    batch = [[rs.randn(n).astype(np.float32), rs.randn(80, n // 100).astype(np.float32), int(i * 3),
              rs.randn(4).astype(np.float32), rs.randn(2, 3, 1 + i).astype(np.float32)] for i, n in enumerate(lens)]
    for i, item in enumerate(batch):
        for j, x in enumerate(item):
            out['collate/in/%d/%d' % (i, j)] = np.asarray(x)
    res = rd.SpeechDataLoader.pad_collate_fn(batch)
    for j, x in enumerate(res):
        out['collate/out/%d' % j] = x.numpy()
    one = rd.SpeechDataLoader.pad_collate_fn(batch[:1])
    for j, x in enumerate(one):
        out['collate/one/%d' % j] = x.numpy()

    class _Src:
        def __len__(self):
            return 1000

    for tag, skip in (('all', False), ('skip', True)):
        np.random.seed(7)
        smp = rd.BucketRandomBatchSampler(_Src(), n_buckets=5, batch_size=16, skip_last_bucket=skip)
        out['sampler/%s/batches' % tag] = np.asarray(list(smp), np.int64)
        out['sampler/%s/len' % tag] = np.asarray(len(smp))
        out['sampler/%s/bucket_size' % tag] = np.asarray(smp.bucket_size)

    with tempfile.TemporaryDirectory() as d:
        rows = []
        for i, n in enumerate([3000, 2500, 4000, 2048]):
            a, b = os.path.join(d, 'a%d.npy' % i), os.path.join(d, 'b%d.npy' % i)
            np.save(a, rs.randn(n).astype(np.float32))
            np.save(b, rs.randn(n).astype(np.float32))
            rows.append({'mix': a, 'voice': b, 'speaker': i % 2, 'note': 'x'})
            out['dataset/mix/%d' % i] = np.load(a)
            out['dataset/voice/%d' % i] = np.load(b)

        class Meta(MetaFrame):
            sr = 22050

            def __init__(self):
                self._meta = pd.DataFrame(rows)

            @property
            def columns(self):
                return [(MetaType.AUDIO, 'mix'), (MetaType.AUDIO, 'voice'), (MetaType.SCALAR, 'speaker'), (MetaType.META, 'note')]

            @property
            def meta(self):
                return self._meta

            def make_meta(self):
                pass

            def __len__(self):
                return len(self._meta)

        for tag, kw in (('crop', dict(fix_len=2048, audio_mask=True)),
                        ('shuffle', dict(fix_len=1024, fix_shuffle=True, extra_features=[('mix', lambda x: np.abs(x).astype(np.float32))])),
                        ('whole', dict())):
            np.random.seed(11)
            ds = rd.SpeechDataset(Meta(), **kw)
            for i in range(len(ds)):
                for j, x in enumerate(ds[i]):
                    out['dataset/%s/%d/%d' % (tag, i, j)] = np.asarray(x)
            out['dataset/%s/nfields' % tag] = np.asarray(len(ds[0]))
    np.savez_compressed(os.path.join(OUT, 'data.npz'), **out)


def gen_filters():
    """f4: PQMF (transforms.py:462-560) from the imported reference: filters, analysis, synthesis, autograd gradients."""
    from pytorch_sound.models import transforms as rt
    out = {}
    for tag, kw in (('s4', dict()), ('s8', dict(subbands=8, taps=126, cutoff_ratio=0.07, beta=10.0))):
        pq = rt.PQMF(**kw)
        S = pq.subbands
        x = torch.from_numpy(seeded_wav(950 + S, 2, 2048 + 3)).unsqueeze(1).requires_grad_(True)     # T not a multiple of S
        a = pq.analysis(x)
        ga = torch.from_numpy(np.random.RandomState(951).randn(*a.shape).astype(np.float32))
        (a * ga).sum().backward()
        sb = a.detach().clone().requires_grad_(True)
        y = pq.synthesis(sb)
        gy = torch.from_numpy(np.random.RandomState(952).randn(*y.shape).astype(np.float32))
        (y * gy).sum().backward()
        out[tag + '/analysis_filter'] = pq.analysis_filter.numpy()
        out[tag + '/synthesis_filter'] = pq.synthesis_filter.numpy()
        out[tag + '/x'] = x.detach().numpy()
        out[tag + '/analysis'] = a.detach().numpy()
        out[tag + '/ga'] = ga.numpy()
        out[tag + '/gx'] = x.grad.numpy()
        out[tag + '/synthesis'] = y.detach().numpy()
        out[tag + '/gy'] = gy.numpy()
        out[tag + '/gsb'] = sb.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'filters.npz'), **out)


def gen_lstft():
    """LearnableSTFT (transforms.py:104-203, experimental in the reference): transform / inverse and the gradient wrt the bases."""
    from pytorch_sound.models import transforms as rt
    out = {}
    m = rt.LearnableSTFT(256, 64, 200)
    wav = torch.from_numpy(seeded_wav(960, 2, 2048))
    mag, phase = m.transform(wav)
    rec = m.inverse(mag, phase)
    (mag.sum() + rec.pow(2).sum()).backward()
    out['wav'] = wav.numpy()
    out['mag'] = mag.detach().numpy()
    out['phase'] = phase.detach().numpy()
    out['rec'] = rec.detach().numpy()
    out['g_forward_basis_rows'] = m.forward_basis.grad.numpy()[[0, 1, 64, 129, 200, 257]]
    out['g_inverse_basis_rows'] = m.inverse_basis.grad.numpy()[[0, 1, 64, 129, 200, 257]]
    out['state_keys'] = np.asarray(sorted(m.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, 'lstft.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    which = sys.argv[1:] or ['stft', 'impulse', 'logmel', 'torch_stft', 'torch_stft_modules', 'modules', 'hifigan', 'trainer', 'sound', 'data', 'filters', 'lstft']
    for w in which:
        print('generating', w, flush=True)
        globals()['gen_' + w]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
