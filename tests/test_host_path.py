"""the transform modules are device-agnostic in the reference (interface/hifi_gan.py:92 defaults to
device='cpu'; BASELINE configs[0] is "feature extraction only, batch=4 on CPU").  A CPU tensor takes the host formulation of
pytorch_sound_amd/host.py (torch ops written in the product - NOT oracle/); these tests pin it to the goldens produced by the
imported reference (tools/gen_golden.py) with the tolerances the GPU tests of the same rows hold, and then use the oracle as an
independent float64 check.  HIP tensors never reach host.py (tests/test_gpu_no_library_paths.py)."""
import numpy as np
import pytest
import torch

STFT_CASES = ['n1024_h256', 'n1024_h256_w800', 'n512_h128', 'n256_h64_w200', 'n2048_h512', 'n4096_h1024']


def _np(t):
    return t.detach().numpy()


@pytest.mark.parametrize('name', STFT_CASES)
def test_stft_transform_and_inverse_on_cpu(golden, name):
    from pytorch_sound.models.transforms import STFT
    g = golden('stft')
    n, h, w = (int(v) for v in g[name + '/params'])
    m = STFT(filter_length=n, hop_length=h, win_length=w)
    wav = torch.from_numpy(g[name + '/wav'])
    mag, phase = m.transform(wav)
    gm, gp = g[name + '/mag'], g[name + '/phase']
    assert mag.shape == gm.shape and phase.shape == gp.shape and not phase.requires_grad
    assert np.abs(_np(mag) - gm).max() <= 4e-6 * gm.max()             # the reference's own fp32 dense-DFT noise (test_oracle_golden)
    big = gm > 1e-2 * gm.max()
    d = np.angle(np.exp(1j * (_np(phase).astype(np.float64) - gp)))
    assert np.abs(d[big]).max() <= 1e-3
    if name + '/inverse' in g.files:
        rec = m.inverse(torch.from_numpy(gm), torch.from_numpy(gp))
        assert tuple(rec.shape) == g[name + '/inverse'].shape
        assert np.abs(_np(rec) - g[name + '/inverse']).max() <= 2e-6
    # round trip through the module (transforms.py:71-101 docstring use)
    y = m.inverse(*m.transform(wav))
    T = min(y.shape[1], wav.shape[1])
    assert (y[:, :T] - wav[:, :T]).abs().max().item() <= 2e-5


def test_stft_magnitude_gradient_on_cpu(golden):
    from pytorch_sound.models.transforms import STFT
    g = golden('stft')
    x = torch.from_numpy(g['bwd/wav']).requires_grad_(True)
    mag, phase = STFT(1024, 256).transform(x)
    (mag * torch.from_numpy(g['bwd/gmag'])).sum().backward()
    ref = g['bwd/gwav']
    assert np.abs(_np(x.grad) - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize('name', ['default', 'noclamp', 'zero_db_disables', 'silence'])
def test_logmel_on_cpu(golden, name):
    from pytorch_sound.models.transforms import LogMelSpectrogram
    g = golden('logmel')
    kw = g[name + '/kw']
    opt = lambda v: None if np.isnan(v) else float(v)  # noqa: E731
    m = LogMelSpectrogram(int(kw[0]), int(kw[1]), int(kw[2]), int(kw[3]), int(kw[4]), opt(kw[5]), opt(kw[6]), float(kw[7]), opt(kw[8]))
    m.mel_filter.copy_(torch.from_numpy(g[name + '/mel_filter']))       # the filter the reference run used (librosa stub = the restated one)
    out = m(torch.from_numpy(g[name + '/wav']))
    assert tuple(out.shape) == g[name + '/mel'].shape
    assert np.abs(_np(out) - g[name + '/mel']).max() <= 1e-4
    if name == 'silence':
        assert np.all(_np(out) == g[name + '/mel'])


def test_logmel_gradient_on_cpu(golden):
    from pytorch_sound.models.transforms import LogMelSpectrogram
    from pytorch_sound import settings
    g = golden('logmel')
    m = LogMelSpectrogram(settings.SAMPLE_RATE, settings.MEL_SIZE, settings.N_FFT, settings.WIN_LENGTH, settings.HOP_LENGTH,
                          float(settings.MIN_DB), float(settings.MAX_DB), float(settings.MEL_MIN), float(settings.MEL_MAX))
    x = torch.from_numpy(g['bwd/wav']).requires_grad_(True)
    (m(x) * torch.from_numpy(g['bwd/gmel'])).sum().backward()
    ref = g['bwd/gwav']
    assert np.abs(_np(x.grad) - ref).max() <= 2e-3 * np.abs(ref).max()


def test_baseline_config0_on_cpu():
    """BASELINE.json configs[0]: VoiceBank 22.05 kHz, 1024-pt STFT / 256 hop / 80-mel feature extraction, batch = 4 on CPU with the
    settings.py defaults, through the drop-in import path; checked against the float64 oracle."""
    from conftest import seeded_wav
    from oracle import features as ofe
    from pytorch_sound import settings
    from pytorch_sound.models.transforms import LogMelSpectrogram
    wav = seeded_wav(1234, 4, 44100)
    m = LogMelSpectrogram(settings.SAMPLE_RATE, settings.MEL_SIZE, settings.N_FFT, settings.WIN_LENGTH, settings.HOP_LENGTH,
                          float(settings.MIN_DB), float(settings.MAX_DB), float(settings.MEL_MIN), float(settings.MEL_MAX))
    out = m(torch.from_numpy(wav))
    assert tuple(out.shape) == (4, 80, 173)
    ref = ofe.logmel_f64(wav, settings.SAMPLE_RATE, settings.MEL_SIZE, settings.N_FFT, settings.WIN_LENGTH, settings.HOP_LENGTH,
                         float(settings.MIN_DB), float(settings.MAX_DB), float(settings.MEL_MIN), float(settings.MEL_MAX))
    assert np.abs(_np(out) - ref).max() <= 2e-4


MODULE_CASES = {'w1024': dict(filter_length=1024, hop_length=256),
                'w600': dict(filter_length=1024, hop_length=256, win_length=600, n_fft=1024),
                'n512': dict(filter_length=512, hop_length=128)}


@pytest.mark.parametrize('tag', sorted(MODULE_CASES))
def test_stft_torchaudio_on_cpu(golden, tag):
    from pytorch_sound.models.transforms import STFTTorchAudio
    g = golden('torch_stft_modules')
    m = STFTTorchAudio(**MODULE_CASES[tag])
    x = torch.from_numpy(g['wav']).requires_grad_(True)
    re, im = m(x)
    sc = np.abs(g[tag + '/mag']).max()
    assert np.abs(_np(re) - g[tag + '/re']).max() <= 4e-6 * sc and np.abs(_np(im) - g[tag + '/im']).max() <= 4e-6 * sc
    mag, ph = m.transform(x)
    assert mag.requires_grad and ph.requires_grad                       # transforms.py:311: the phase is NOT detached
    assert np.abs(_np(mag) - g[tag + '/mag']).max() <= 4e-6 * sc
    strong = g[tag + '/mag'] > 1e-2 * sc
    d = np.angle(np.exp(1j * (_np(ph).astype(np.float64) - g[tag + '/phase'])))
    assert np.abs(d[strong]).max() <= 1e-4
    gg = torch.from_numpy(g[tag + '/g'])
    (mag * gg[0] + ph * gg[1]).sum().backward()
    ref = g[tag + '/gwav']
    assert np.abs(_np(x.grad) - ref).max() <= 2e-3 * np.abs(ref).max()
    for a, b, o in (('mag', 'phase', 'inverse'), ('amag', 'aphase', 'ainverse')):
        y = m.inverse(torch.from_numpy(g[tag + '/' + a]), torch.from_numpy(g[tag + '/' + b]))
        ref = g[tag + '/' + o]
        assert tuple(y.shape) == ref.shape
        assert np.abs(_np(y) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_audio2mel_and_interface_mel_on_cpu(golden):
    from pytorch_sound.models.transforms import Audio2Mel
    from pytorch_sound.interface.hifi_gan import MelSpectrogram
    g = golden('torch_stft_modules')
    x = torch.from_numpy(g['wav'])
    out = Audio2Mel()(x.unsqueeze(1))
    assert tuple(out.shape) == g['audio2mel/out'].shape
    assert np.abs(_np(out) - g['audio2mel/out']).max() <= 2e-4
    out = Audio2Mel(n_fft=512, hop_length=128, win_length=512, sampling_rate=16000, n_mel_channels=40, mel_fmin=50.0,
                    mel_fmax=7000.0)(x.unsqueeze(1))
    assert np.abs(_np(out) - g['audio2mel_b/out']).max() <= 2e-4
    ms = MelSpectrogram()
    out = ms(x)
    assert tuple(out.shape) == g['interface/out'].shape
    assert np.abs(_np(out) - g['interface/out']).max() <= 2e-4
    out = ms(x, is_center=True)
    assert tuple(out.shape) == g['interface/out_center'].shape
    assert np.abs(_np(out) - g['interface/out_center']).max() <= 2e-4


def test_interface_hifigan_default_device_is_cpu(tmp_path, monkeypatch):
    """InterfaceHifiGAN() with the reference's default device='cpu' (interface/hifi_gan.py:92): encode and decode run on the host."""
    import pytorch_sound.interface.hifi_gan as ih
    from pytorch_sound.models import build_model
    torch.manual_seed(0)
    gen = build_model('hifi_gan_v3')
    ck = tmp_path / 'hifi_gan_v3.pt'
    torch.save({'generator': gen.state_dict()}, ck)
    itf = ih.InterfaceHifiGAN('hifi_gan_v3', chk_path=str(ck))
    from conftest import seeded_wav
    wav = torch.from_numpy(seeded_wav(5, 2, 8192))
    mel = itf.encode(wav)
    assert tuple(mel.shape) == (2, 80, 32) and torch.isfinite(mel).all()
    rec = itf.decode(mel)
    assert tuple(rec.shape) == (2, 1, 8192) and torch.isfinite(rec).all()
    # the encoder equals the stand-alone module
    assert torch.equal(mel, ih.MelSpectrogram()(wav))


def test_stft_state_dict_carries_the_reference_buffers(golden):
    """a checkpoint saved here must pass a strict load in the reference (transforms.py:49-51 registers forward_basis /
    inverse_basis): emitted by default up to n = 1024, bit-identical rows to the reference's forward_basis."""
    from pytorch_sound.models.transforms import STFT
    g = golden('stft')
    m = STFT(1024, 256)
    sd = m.state_dict()
    assert set(sd) == {'square_window', 'forward_basis', 'inverse_basis'}
    assert tuple(sd['forward_basis'].shape) == (1026, 1, 1024) and tuple(sd['inverse_basis'].shape) == (1026, 1, 1024)
    rows = g['n1024_h256/basis_rows_idx']
    assert np.abs(sd['forward_basis'].numpy()[rows, 0] - g['n1024_h256/basis_rows']).max() <= 1e-6
    assert np.array_equal(sd['square_window'].numpy(), g['n1024_h256/square_window'])
    m2 = STFT(1024, 256)
    m2.load_state_dict(sd, strict=True)                                  # and it loads back (the dense bases are dropped)
    assert set(STFT(4096, 1024).state_dict()) == {'square_window'}       # 2 x 67 MB: opt-in
