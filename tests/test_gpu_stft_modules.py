"""Module-level GPU parity of the torch.stft-convention front ends: the drop-in CLASSES
STFTTorchAudio (forward / transform with its differentiable phase / inverse), Audio2Mel and interface.hifi_gan.MelSpectrogram
(is_center both ways) against the outputs of the imported reference's classes (tests/golden/torch_stft_modules.npz and
torch_stft.npz; generator: tools/gen_golden.py) - the glue around the kernels (pad_size, the double padding of is_center=True,
fmax defaults, window centring, the (F - 1) * hop length convention of torch.istft) is what these tests pin.

Tolerances: (re, im, magnitude) 4e-6 of the largest bin (fp32 FFT round-off); phase 1e-4 rad on bins above 1 % of the largest
(atan2 is ill-conditioned below); log-mel 2e-4 absolute (as test_gpu_features); inverse 1e-5 absolute on unit-scale audio;
waveform gradient through (magnitude, phase) 2e-3 of its maximum (the reference's own fp32 atan2 gradient sits at the same
distance from float64, tests/test_oracle_golden.py::test_torch_stft_modules_stft).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {'w1024': dict(filter_length=1024, hop_length=256),
         'w600': dict(filter_length=1024, hop_length=256, win_length=600, n_fft=1024),
         'n512': dict(filter_length=512, hop_length=128)}


def _dev():
    assert torch.cuda.is_available(), 'GPU test run without a GPU'
    return torch.device('cuda:0')


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('tag', sorted(CASES))
def test_stft_torchaudio_module(golden, tag):
    from pytorch_sound_amd.models.transforms import STFTTorchAudio
    g = golden('torch_stft_modules')
    dev = _dev()
    m = STFTTorchAudio(**CASES[tag]).to(dev)
    x = torch.from_numpy(g['wav']).to(dev).requires_grad_(True)
    re, im = m(x)
    sc = np.abs(g[tag + '/mag']).max()
    assert re.shape == g[tag + '/re'].shape
    assert np.abs(_np(re) - g[tag + '/re']).max() <= 4e-6 * sc and np.abs(_np(im) - g[tag + '/im']).max() <= 4e-6 * sc
    mag, ph = m.transform(x)
    assert mag.requires_grad and ph.requires_grad                       # transforms.py:311: the phase is NOT detached here
    assert np.abs(_np(mag) - g[tag + '/mag']).max() <= 4e-6 * sc
    strong = g[tag + '/mag'] > 1e-2 * sc
    d = np.angle(np.exp(1j * (_np(ph).astype(np.float64) - g[tag + '/phase'])))
    assert np.abs(d[strong]).max() <= 1e-4
    gg = torch.from_numpy(g[tag + '/g']).to(dev)
    (mag * gg[0] + ph * gg[1]).sum().backward()
    ref = g[tag + '/gwav']
    assert np.abs(_np(x.grad) - ref).max() <= 2e-3 * np.abs(ref).max()
    # magnitude-only gradient takes the tuned magnitude adjoint: same answer as the (re, im) route
    x2 = torch.from_numpy(g['wav']).to(dev).requires_grad_(True)
    mag2, _ = m.transform(x2)
    (mag2 * gg[0]).sum().backward()
    x3 = torch.from_numpy(g['wav']).to(dev).requires_grad_(True)
    re3, im3 = m(x3)
    (torch.sqrt(re3 ** 2 + im3 ** 2) * gg[0]).sum().backward()
    assert np.abs(_np(x2.grad) - _np(x3.grad)).max() <= 5e-5 * np.abs(_np(x3.grad)).max()


@pytest.mark.parametrize('tag', sorted(CASES))
def test_stft_torchaudio_inverse(golden, tag):
    """f1: STFTTorchAudio.inverse (transforms.py:313-319) on psnd_istft - no library FFT for a GPU tensor."""
    from pytorch_sound_amd.models.transforms import STFTTorchAudio
    g = golden('torch_stft_modules')
    dev = _dev()
    m = STFTTorchAudio(**CASES[tag]).to(dev)
    for a, b, o in (('mag', 'phase', 'inverse'), ('amag', 'aphase', 'ainverse')):
        y = m.inverse(torch.from_numpy(g[tag + '/' + a]).to(dev), torch.from_numpy(g[tag + '/' + b]).to(dev))
        ref = g[tag + '/' + o]
        assert tuple(y.shape) == ref.shape                              # (F - 1) * hop samples
        assert np.abs(_np(y) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    # analysis -> synthesis round trip through the module
    x = torch.from_numpy(g['wav']).to(dev)
    y = m.inverse(*m.transform(x))
    T = min(y.shape[1], x.shape[1])
    assert (y[:, :T] - x[:, :T]).abs().max().item() <= 2e-5
    # differentiable (the inverse is linear in mag e^{i phase}): gradient vs torch.istft's own autograd on the host
    mag = torch.from_numpy(g[tag + '/amag']).to(dev).requires_grad_(True)
    ph = torch.from_numpy(g[tag + '/aphase']).to(dev).requires_grad_(True)
    from test_gpu_no_library_paths import forbid_library_ops
    with forbid_library_ops():                                          # forward AND backward: no library conv / FFT for a HIP tensor
        y = m.inverse(mag, ph)
        gy = torch.from_numpy(np.random.RandomState(7).randn(*y.shape).astype(np.float32)).to(dev)
        (y * gy).sum().backward()
    cm = torch.from_numpy(g[tag + '/amag']).double().requires_grad_(True)
    cp = torch.from_numpy(g[tag + '/aphase']).double().requires_grad_(True)
    kw = CASES[tag]
    n_fft = kw.get('n_fft') or kw.get('win_length') or kw['filter_length']
    win = kw.get('win_length') or kw['filter_length']
    yc = torch.istft(torch.polar(cm, cp), n_fft, kw['hop_length'], win, torch.hann_window(win, dtype=torch.float64))
    (yc * gy.cpu().double()).sum().backward()
    assert np.abs(_np(mag.grad) - _np(cm.grad)).max() <= 1e-4 * np.abs(_np(cm.grad)).max()
    # the DC / Nyquist phase does not reach a real signal: compare the rest
    assert np.abs(_np(ph.grad)[:, 1:-1] - _np(cp.grad)[:, 1:-1]).max() <= 1e-4 * np.abs(_np(cp.grad)).max()


def test_audio2mel_module(golden):
    from pytorch_sound_amd.models.transforms import Audio2Mel
    g = golden('torch_stft_modules')
    dev = _dev()
    x = torch.from_numpy(g['wav']).to(dev).unsqueeze(1)
    out = Audio2Mel().to(dev)(x)
    assert tuple(out.shape) == g['audio2mel/out'].shape
    assert np.abs(_np(out) - g['audio2mel/out']).max() <= 2e-4
    out = Audio2Mel(n_fft=512, hop_length=128, win_length=512, sampling_rate=16000, n_mel_channels=40, mel_fmin=50.0,
                    mel_fmax=7000.0).to(dev)(x)
    assert tuple(out.shape) == g['audio2mel_b/out'].shape
    assert np.abs(_np(out) - g['audio2mel_b/out']).max() <= 2e-4
    with pytest.raises(RuntimeError):
        Audio2Mel().to(dev)(x.squeeze(1))                               # (N, 1, T) only, as the reference's F.pad demands
    # the fixture captured from torch.stft itself (tests/golden/torch_stft.npz, consumed on the CPU so far)
    g2 = golden('torch_stft')
    out = Audio2Mel().to(dev)(torch.from_numpy(g2['center/wav']).to(dev).unsqueeze(1))
    assert np.abs(_np(out) - g2['hifigan/audio2mel']).max() <= 2e-4


def test_interface_melspectrogram_module(golden):
    from pytorch_sound_amd.interface.hifi_gan import MelSpectrogram
    g = golden('torch_stft_modules')
    dev = _dev()
    x = torch.from_numpy(g['wav']).to(dev)
    ms = MelSpectrogram().to(dev)
    assert ms.pad_size == 384
    out = ms(x)
    assert tuple(out.shape) == g['interface/out'].shape
    assert np.abs(_np(out) - g['interface/out']).max() <= 2e-4
    out = ms(x, is_center=True)                                         # pad_size reflect pad, THEN the centre pad of torch.stft
    assert tuple(out.shape) == g['interface/out_center'].shape
    assert np.abs(_np(out) - g['interface/out_center']).max() <= 2e-4
    g2 = golden('torch_stft')
    out = ms(torch.from_numpy(g2['center/wav']).to(dev))
    assert np.abs(_np(out) - g2['hifigan/interface_mel']).max() <= 2e-4
    # with gradient: the autograd route (magnitude -> mel kernel) equals the fused forward
    xg = x.clone().requires_grad_(True)
    out_g = ms(xg)
    assert np.abs(_np(out_g) - _np(ms(x))).max() <= 2e-5
    out_g.sum().backward()
    assert torch.isfinite(xg.grad).all()


def test_polar_bwd_kernel_matches_formula():
    """psnd_polar_bwd against the float64 formula, sizes that are / are not multiples of 4, with and without g_mag."""
    from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
    dev = _dev()
    rs = np.random.RandomState(3)
    for n in (1, 7, 1024, 4099, 513 * 173 * 3):
        mag = (np.abs(rs.randn(n)) + 0.1).astype(np.float32)
        ph = rs.uniform(-np.pi, np.pi, n).astype(np.float32)
        gm, gp = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
        for use_gm in (True, False):
            t = [torch.from_numpy(a).to(dev) for a in (gm, gp, mag, ph)]
            gre, gim = torch.full((n,), 7.0, device=dev), torch.full((n,), 7.0, device=dev)
            check(lib().psnd_polar_bwd(ptr(t[0]) if use_gm else None, ptr(t[1]), ptr(t[2]), ptr(t[3]), n, ptr(gre), ptr(gim),
                                       stream_ptr(dev)), 'psnd_polar_bwd')
            g0 = gm.astype(np.float64) if use_gm else 0.0
            q = gp.astype(np.float64) / mag
            ere = g0 * np.cos(ph.astype(np.float64)) - q * np.sin(ph.astype(np.float64))
            eim = g0 * np.sin(ph.astype(np.float64)) + q * np.cos(ph.astype(np.float64))
            sc = max(np.abs(ere).max(), np.abs(eim).max())
            assert np.abs(_np(gre) - ere).max() <= 2e-6 * sc and np.abs(_np(gim) - eim).max() <= 2e-6 * sc


# ---- round 5: filter_length that is not a power of two on HIP tensors (reference transforms.py:19-51 takes any) -------------------------------
@pytest.mark.parametrize('n,hop,win', [(800, 200, None), (1200, 300, None), (2400, 600, 2000), (1000, 250, None), (48, 12, None),
                                       (801, 200, None), (75, 25, None), (1025, 256, 1001)])     # round 6: odd sizes (K = int(n / 2 + 1), no Nyquist bin)
def test_stft_any_filter_length_on_hip_tensors(n, hop, win):
    """STFT(filter_length = 800 / 1200 / 2400 ...) used to raise on a HIP tensor; it now runs the reference's dense-basis formulation on the
    exact-fp32 matrix-core GEMM (pytorch_sound_amd/dense.py, psnd_linear1x1_*): magnitude and phase against the float64 oracle, the
    magnitude's gradient against the oracle's adjoint, inverse(transform(x)) == x, no library convolution / fft / bmm on the way.
    Tolerance 2e-5 of the largest bin (fp32 sums of n products)."""
    import numpy as np
    from conftest import seeded_wav
    from oracle import features as ofe
    from pytorch_sound_amd.models.transforms import STFT
    from test_gpu_no_library_paths import forbid_library_ops
    dev = torch.device('cuda:0')
    N, T = 3, 6 * n + 37
    wav_np = seeded_wav(n + hop, N, T)
    stft = STFT(n, hop, win).to(dev)
    x = torch.from_numpy(wav_np).to(dev).requires_grad_(True)
    with forbid_library_ops():
        mag, phase = stft.transform(x)
        g = torch.from_numpy(np.random.RandomState(1).randn(*mag.shape).astype(np.float32)).to(dev)
        (mag * g).sum().backward()
        rec = stft.inverse(mag.detach(), phase)
    ref = ofe.stft_mag_f64(wav_np, n, hop, win)
    assert mag.shape == ref.shape
    assert np.abs(mag.detach().cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    re, im = ofe.stft_reim_f64(wav_np, n, hop, win) if hasattr(ofe, 'stft_reim_f64') else (None, None)
    if re is not None:
        strong = ref > 1e-3 * ref.max()                                  # the phase of a near-zero bin is noise
        d = np.angle(np.exp(1j * (phase.cpu().numpy() - np.arctan2(im, re))))
        assert np.abs(d[strong]).max() <= 2e-3
    gref = ofe.stft_mag_bwd_f64(g.cpu().numpy().astype(np.float64), wav_np, n, hop, win)
    assert np.abs(x.grad.cpu().numpy() - gref).max() <= 5e-5 * np.abs(gref).max()
    Fr = mag.shape[2]
    L = (Fr - 1) * hop + n % 2                                            # n + hop (F - 1) - 2 int(n / 2) samples (transforms.py:96-99)
    assert rec.shape == (N, L)
    assert float((rec - x.detach()[:, :L]).abs().max()) <= 2e-5 * float(x.detach().abs().max()) + 2e-5
