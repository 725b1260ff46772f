"""GPU parity tests of the STFT / mel hot path: HIP kernels (through the C ABI) vs the CPU oracle
and the golden vectors captured from the imported reference.

Tolerances (stated per north_star): frame indexing bit-exact; floating point compared with the
float64 oracle at |err| <= 4e-6 * max|X| for FFT outputs (fp32 FFT round-off ~ log2(n) * 6e-8
relative to the frame's largest bin; the reference's own two code paths - dense-DFT conv1d vs
torch.stft - differ by 1.5e-6 of max), and <= 2e-5 absolute for log-mel values.
"""
import numpy as np
import pytest
import torch

from conftest import seeded_wav
from oracle import features as ofe

pytestmark = pytest.mark.gpu

FFT_RTOL = 4e-6


def _dev():
    assert torch.cuda.is_available(), 'GPU test run without a GPU'
    return torch.device('cuda:0')


def _k():
    from pytorch_sound_amd import kernels
    return kernels


def _stft(wav_np, n_fft, hop, win_length=None, framing=0, mag_eps=0.0, window=None, **want):
    K = _k()
    dev = _dev()
    w = ofe.analysis_window(n_fft, win_length) if window is None else np.asarray(window, np.float32)
    plan = K.stft_plan(n_fft, w).to(dev)
    out = K.stft_forward(torch.from_numpy(wav_np).to(dev), n_fft, hop, plan, framing, mag_eps, **want)
    torch.cuda.synchronize()
    return {k: (None if v is None else v.cpu().numpy()) for k, v in out.items()}


CASES = [
    # n_fft, hop, win, framing, N, T
    (1024, 256, None, 0, 2, 4200),
    (1024, 256, None, 0, 4, 44100),       # BASELINE config 1 shape
    (1024, 256, 800, 0, 2, 3000),
    (1024, 256, None, 1, 3, 8192),        # HiFi-GAN framing, config 3 segment
    (1024, 300, None, 0, 1, 5000),        # hop not dividing n_fft
    (1024, 255, 1000, 1, 2, 2049),        # odd hop, odd T -> unaligned rows
    (512, 128, None, 0, 3, 1111),
    (512, 128, 400, 1, 2, 4096),
    (256, 64, 200, 0, 2, 700),
    (256, 64, None, 1, 5, 1000),
    (2048, 512, None, 0, 1, 6000),
    (2048, 512, 1200, 1, 2, 9000),
    (4096, 1024, None, 0, 1, 9000),       # three-pass 4096 kernel (config 5), clip shorter than a few frames
    (4096, 1024, None, 0, 3, 44100),      # several tiles per clip, partial last tile (F = 44: 11 tiles)
    (4096, 1024, 3000, 1, 2, 30000),      # HiFi-GAN framing, short window
    (4096, 1022, None, 0, 1, 20000),      # even hop that is not a multiple of 4
    (4096, 1023, None, 0, 1, 20000),      # odd hop -> generic path
    (128, 32, None, 0, 2, 500),           # generic path
    (64, 16, None, 1, 2, 200),            # generic path
    (1024, 256, None, 0, 1, 513),         # minimum-ish T (> pad): every frame touches an edge
    (1024, 256, None, 0, 33, 2600),       # many clips, partial last tile
]
# n_fft = 512 on the span-staged kernel (32 frames per workgroup, two pass-1 rounds)
N512_CASES = [
    (512, 50, 240, 0, 3, 5000),           # multi_stft_loss resolution (sound.py:106-133): even hop, not a multiple of 4
    (512, 128, None, 0, 40, 2600),        # many clips, partial last tile (F = 21 of 32)
    (512, 200, None, 1, 2, 9000),         # span of 32 frames > 5 K samples: persistent variant, 8 span pieces
    (512, 129, None, 0, 2, 3000),         # odd hop -> two-pass kernel
    (512, 128, None, 0, 1, 257),          # T barely above the reflect pad
]
# n_fft = 2048 on the span-staged kernel (16 frames per workgroup, two pass-1 rounds of 8 frames, 32-point second pass)
N2048_CASES = [
    (2048, 240, 1200, 0, 3, 9000),        # multi_stft_loss resolution
    (2048, 512, None, 0, 20, 20000),      # many clips, partial last tile, 10 span pieces per thread
    (2048, 546, None, 1, 2, 30000),       # the largest span that fits (15 * 546 + 2048 = 10238 samples)
    (2048, 600, None, 0, 1, 20000),       # span too long -> two-pass kernel
    (2048, 512, None, 0, 1, 1025),        # T barely above the reflect pad
]
CASES += N512_CASES + N2048_CASES


@pytest.mark.parametrize('n_fft,hop,win,framing,N,T', CASES)
def test_stft_mag_vs_oracle(n_fft, hop, win, framing, N, T):
    wav = seeded_wav(n_fft + hop + T, N, T)
    got = _stft(wav, n_fft, hop, win, framing, want_mag=True)['mag']
    ref = ofe.stft_mag_f64(wav, n_fft, hop, win, framing)
    assert got.shape == ref.shape == (N, n_fft // 2 + 1, ofe.frame_count(T, n_fft, hop, framing))
    tol = FFT_RTOL * np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol


# n_fft = 4096, magnitude only, on the wave-per-frame kernel (psnd_stft_w.hip: 16 frames per workgroup, 64-byte store runs).  The
# dispatcher takes it from 2048 16-frame tiles on (config 5 at 32 clips); PSND_STFT4096_W=1 forces it at test sizes.
N4096W_CASES = [
    (4096, 1024, None, 0, 1, 9000),       # F = 9: one partial tile, frames 0-1 and 7-8 touch the clip edges (reflect), F % 4 != 0
    (4096, 1024, None, 0, 3, 44100),      # F = 44 (F % 4 == 0: 16-byte stores), 3 tiles per clip, partial last tile
    (4096, 1024, 3000, 1, 2, 30000),      # HiFi-GAN framing, short window
    (4096, 1022, None, 0, 1, 20000),      # even hop that is not a multiple of 4
    (4096, 512, None, 0, 2, 40000),       # hop = n / 8
    (4096, 1900, None, 0, 2, 70000),      # the widest hop the 8-byte sample loads cover
    (4096, 1024, None, 0, 1, 2049),       # T barely above the reflect pad: every frame touches an edge
    (4096, 1024, None, 0, 21, 60000),     # many clips: several persistent workgroups with more than one tile
]


@pytest.mark.parametrize('n_fft,hop,win,framing,N,T', N4096W_CASES)
def test_stft_4096_wave_kernel_vs_oracle(n_fft, hop, win, framing, N, T, monkeypatch, lab_lib):
    monkeypatch.setenv('PSND_STFT4096_W', '1')
    wav = seeded_wav(n_fft + hop + T, N, T)
    got = _stft(wav, n_fft, hop, win, framing, want_mag=True)['mag']
    ref = ofe.stft_mag_f64(wav, n_fft, hop, win, framing)
    assert got.shape == ref.shape == (N, n_fft // 2 + 1, ofe.frame_count(T, n_fft, hop, framing))
    assert np.abs(got - ref).max() <= FFT_RTOL * np.abs(ref).max()
    # mag_eps (interface/hifi_gan.py:55) and the 4-frame kernel agree with it as well
    got_eps = _stft(wav, n_fft, hop, win, framing, mag_eps=1e-9, want_mag=True)['mag']
    assert np.abs(got_eps - np.sqrt(ref * ref + 1e-9)).max() <= FFT_RTOL * np.abs(ref).max()
    monkeypatch.delenv('PSND_STFT4096_W')
    monkeypatch.setenv('PSND_STFT4096_V2', '1')
    old = _stft(wav, n_fft, hop, win, framing, want_mag=True)['mag']
    assert np.abs(got - old).max() <= 2 * FFT_RTOL * np.abs(ref).max()


def test_stft_4096_wave_kernel_full_size(monkeypatch, lab_lib):
    """config 5 at the size where the dispatcher picks the wave-per-frame kernel by itself (32 clips x 30 s: 2592 tiles): equal to the
    4-frame kernel on the same input, Parseval per frame, every element written (NaN-filled output)."""
    K = _k()
    dev = _dev()
    n_fft, hop, N, T = 4096, 1024, 32, 1323000
    g = torch.Generator(device='cpu').manual_seed(3)
    a = (0.0708 * torch.randn(N, T, generator=g)).to(dev)
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft)).to(dev)
    F = K.frame_count(T, n_fft, hop)
    out = torch.full((N, n_fft // 2 + 1, F), float('nan'), device=dev)
    K.stft_forward(a, n_fft, hop, plan, out_mag=out)
    assert torch.isfinite(out).all()
    monkeypatch.setenv('PSND_STFT4096_V2', '1')
    old = K.stft_forward(a, n_fft, hop, plan)['mag']
    assert float((out - old).abs().max()) <= 8e-6 * float(old.max())
    # three whole 30-s clips of the full-size launch against the float64 oracle
    for n in (0, 17, 31):
        ref = ofe.stft_mag_f64(a[n:n + 1].cpu().numpy(), n_fft, hop)
        assert np.abs(out[n:n + 1].cpu().numpy() - ref).max() <= FFT_RTOL * np.abs(ref).max()
    w = torch.from_numpy(ofe.analysis_window(n_fft)).to(dev).double()
    for n in (0, 17, 31):
        xp = torch.nn.functional.pad(a[n:n + 1].unsqueeze(1), (n_fft // 2, n_fft // 2), mode='reflect').squeeze(1).double()
        fr = xp.unfold(-1, n_fft, hop) * w
        rhs = n_fft * (fr * fr).sum(-1)
        m2 = out[n:n + 1].double() ** 2
        lhs = m2[:, 0] + m2[:, -1] + 2 * m2[:, 1:-1].sum(1)
        assert float(((lhs - rhs).abs() / rhs.abs().clamp_min(1e-12)).max()) <= 2e-5


@pytest.mark.parametrize('n_fft,hop,win,framing,N,T', [c for c in CASES if c[0] in (64, 256, 512, 1024, 2048)][:9] + N512_CASES + N2048_CASES[:3])
def test_stft_reim_phase_vs_oracle(n_fft, hop, win, framing, N, T):
    wav = seeded_wav(3 * n_fft + T, N, T)
    got = _stft(wav, n_fft, hop, win, framing, want_mag=True, want_phase=True, want_reim=True)
    re, im = ofe.stft_reim_f64(wav, n_fft, hop, win, framing)
    tol = FFT_RTOL * np.sqrt(re * re + im * im).max()
    assert np.abs(got['re'] - re).max() <= tol
    assert np.abs(got['im'] - im).max() <= tol
    assert np.abs(got['mag'] - np.sqrt(re * re + im * im)).max() <= tol
    # phase: compare as unit vectors where the bin is well above round-off (atan2 is ill-conditioned at 0)
    big = np.sqrt(re * re + im * im) > 1e-3 * np.sqrt(re * re + im * im).max()
    d = np.angle(np.exp(1j * (got['phase'] - np.arctan2(im, re))))
    assert np.abs(d[big]).max() <= 1e-3 * 4e-3 / 1e-3  # 4e-3 rad at |X| = 1e-3 max  (err ~ tol/|X|)
    # self-consistency of the kernel's own outputs: phase == atan2(im, re) of what it wrote
    d2 = np.angle(np.exp(1j * (got['phase'] - np.arctan2(got['im'], got['re']))))
    assert np.abs(d2).max() <= 2e-6


def test_stft_golden_reference(golden):
    """mag / phase of the imported reference's STFT.transform (tools/gen_golden.py G1)."""
    g = golden('stft')
    for name in ['n1024_h256', 'n1024_h256_w800', 'n512_h128', 'n256_h64_w200', 'n2048_h512', 'n4096_h1024']:
        n, h, w = (int(v) for v in g[name + '/params'])
        got = _stft(g[name + '/wav'], n, h, w, 0, want_mag=True, want_phase=True)
        ref = g[name + '/mag']
        # the reference itself is a float32 dense DFT: allow its own round-off (1e-5 of max)
        assert np.abs(got['mag'] - ref).max() <= 1e-5 * ref.max(), name
        big = ref > 1e-2 * ref.max()
        d = np.angle(np.exp(1j * (got['phase'] - g[name + '/phase'])))
        assert np.abs(d[big]).max() <= 2e-3, name


def test_frame_indexing_bit_exact():
    """Impulse at sample p, all-ones window: the DC bin of frame f equals the NUMBER of taps of
    frame f that read sample p under reflect indexing - small integers, exact in fp32."""
    for n_fft, hop, T in [(1024, 256, 3000), (512, 128, 1500), (256, 64, 700), (2048, 512, 5000), (64, 16, 200)]:
        for framing in (0, 1):
            F = ofe.frame_count(T, n_fft, hop, framing)
            idx = ofe.frame_sample_index(np.arange(F)[:, None], np.arange(n_fft)[None, :], T, n_fft, hop, framing)
            pad = ofe.pad_amount(n_fft, hop, framing)
            pos = [0, 1, hop - 1, hop, pad - 1, pad, pad + 1, T // 2, T - pad - 1, T - pad, T - 2, T - 1]
            wav = np.zeros((len(pos), T), np.float32)
            for i, p in enumerate(pos):
                wav[i, p] = 1.0
            got = _stft(wav, n_fft, hop, None, framing, window=np.ones(n_fft, np.float32), want_mag=False, want_reim=True)
            for i, p in enumerate(pos):
                count = (idx == p).sum(axis=1).astype(np.float32)          # F
                assert np.array_equal(got['re'][i, 0, :], count), (n_fft, framing, p)


@pytest.mark.parametrize('variant', ['wave', 'tile4', 'reim'])
def test_frame_indexing_bit_exact_n4096(variant, monkeypatch, lab_lib):
    """the same impulse contract at n_fft = 4096 on BOTH magnitude-only kernels of config 5 (`stft_fwd_n4096w_kernel` with its
    hand-resolved reflect loads, forced at this size, and `stft_fwd_n4096b_kernel`) and on the (re, im) instance: |DC| of frame f is
    the number of taps of frame f that read sample p - small integers, exact in fp32 (transforms.py:55-66)."""
    if variant == 'wave':
        monkeypatch.setenv('PSND_STFT4096_W', '1')
    if variant == 'tile4':
        monkeypatch.setenv('PSND_STFT4096_V2', '1')
    n_fft = 4096
    for hop, T in [(1024, 20000), (512, 9000), (1024, 2049)]:
        for framing in (0, 1):
            F = ofe.frame_count(T, n_fft, hop, framing)
            idx = ofe.frame_sample_index(np.arange(F)[:, None], np.arange(n_fft)[None, :], T, n_fft, hop, framing)
            pad = ofe.pad_amount(n_fft, hop, framing)
            pos = sorted({0, 1, hop - 1, hop, min(pad - 1, T - 1), min(pad, T - 1), min(pad + 1, T - 1), T // 2, max(T - pad - 1, 0),
                          max(T - pad, 0), T - 2, T - 1})
            wav = np.zeros((len(pos), T), np.float32)
            for i, p in enumerate(pos):
                wav[i, p] = 1.0
            if variant == 'reim':
                got = _stft(wav, n_fft, hop, None, framing, window=np.ones(n_fft, np.float32), want_mag=False, want_reim=True)['re']
            else:
                got = _stft(wav, n_fft, hop, None, framing, window=np.ones(n_fft, np.float32), want_mag=True)['mag']
            for i, p in enumerate(pos):
                count = (idx == p).sum(axis=1).astype(np.float32)
                assert np.array_equal(got[i, 0, :], count), (variant, hop, T, framing, p)


def test_config5_one_clip_vs_oracle(monkeypatch, lab_lib):
    """one 30-s clip of config 5 (44.1 kHz, 4096 / 1024) against ofe.stft_mag_f64 on both 4096 magnitude kernels."""
    wav = seeded_wav(55, 1, 1323000, 44100)
    ref = ofe.stft_mag_f64(wav, 4096, 1024)
    for env in ('PSND_STFT4096_W', 'PSND_STFT4096_V2'):
        monkeypatch.setenv(env, '1')
        got = _stft(wav, 4096, 1024, want_mag=True)['mag']
        monkeypatch.delenv(env)
        assert got.shape == ref.shape == (1, 2049, 1292)
        assert np.abs(got - ref).max() <= FFT_RTOL * np.abs(ref).max()


def test_frame_indexing_golden(golden):
    """Same contract against the taps recorded from the reference's own pad+conv (G2)."""
    g = golden('impulse')
    n, h, T = (int(v) for v in g['params'])
    for framing in (0, 1):
        taps = g['framing%d/taps' % framing]                                # P,n,F
        wav = np.zeros((len(g['pos']), T), np.float32)
        for i, p in enumerate(g['pos']):
            wav[i, p] = 1.0
        got = _stft(wav, n, h, None, framing, window=np.ones(n, np.float32), want_mag=False, want_reim=True)
        assert np.array_equal(got['re'][:, 0, :], taps.sum(axis=1).astype(np.float32))


@pytest.mark.parametrize('n_fft,hop,N,T,sr', [(1024, 256, 32, 44100, 22050),        # BASELINE config 2: 32 x 2 s
                                               (4096, 1024, 4, 1323000, 44100)])     # config 5: 30 s clips at 44.1 kHz
def test_stft_linearity_and_parseval_full_size(n_fft, hop, N, T, sr):
    """BASELINE full sizes: size-independent properties instead of the slow oracle."""
    K = _k()
    dev = _dev()
    a = torch.from_numpy(seeded_wav(1, N, T, sr)).to(dev)
    b = torch.from_numpy(seeded_wav(2, N, T, sr)).to(dev)
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft)).to(dev)
    A = K.stft_forward(a, n_fft, hop, plan, want_mag=True, want_reim=True)
    B = K.stft_forward(b, n_fft, hop, plan, want_mag=False, want_reim=True)
    S = K.stft_forward(0.5 * a - 2.0 * b, n_fft, hop, plan, want_mag=False, want_reim=True)
    scale = float(A['mag'].max())
    assert float((S['re'] - (0.5 * A['re'] - 2.0 * B['re'])).abs().max()) <= 8e-6 * scale
    assert float((S['im'] - (0.5 * A['im'] - 2.0 * B['im'])).abs().max()) <= 8e-6 * scale
    # Parseval per frame: |X0|^2 + |XC|^2 + 2 sum_{0<k<C} |Xk|^2 = n * sum_m (w x)^2
    w = torch.from_numpy(ofe.analysis_window(n_fft)).to(dev).double()
    xp = torch.nn.functional.pad(a.unsqueeze(1), (n_fft // 2, n_fft // 2), mode='reflect').squeeze(1).double()
    fr = xp.unfold(-1, n_fft, hop) * w                                       # N,F,n
    rhs = n_fft * (fr * fr).sum(-1)                                          # N,F
    m2 = A['mag'].double() ** 2
    lhs = m2[:, 0] + m2[:, -1] + 2 * m2[:, 1:-1].sum(1)
    assert float(((lhs - rhs).abs() / rhs.abs().clamp_min(1e-12)).max()) <= 2e-5
    assert A['mag'].shape == (N, n_fft // 2 + 1, T // hop + 1)


def _mel(mag_np, W, **kw):
    K = _k()
    dev = _dev()
    plan = K.mel_plan(W).to(dev)
    out, lin = K.mel_forward(torch.from_numpy(mag_np).to(dev), plan, W.shape[0], want_lin=True, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy(), lin.cpu().numpy(), plan


@pytest.mark.parametrize('sr,n_fft,M,fmin,fmax,F', [(22050, 1024, 80, 0, 8000, 173), (22050, 1024, 80, 0, None, 32),
                                                    (16000, 512, 40, 50, 7000, 61), (44100, 4096, 128, 0, None, 70),
                                                    (22050, 256, 13, 0, None, 5)])
def test_mel_fwd_bwd_vs_oracle(sr, n_fft, M, fmin, fmax, F):
    K = _k()
    dev = _dev()
    W = ofe.mel_filterbank(sr, n_fft, M, fmin, fmax)
    g = np.random.RandomState(F)
    mag = np.abs(g.randn(3, n_fft // 2 + 1, F)).astype(np.float32) * 3
    mag[0, :, 0] = 0                                                          # silence frame -> lower clamp
    lo, hi = ofe.db_to_ln(-50), ofe.db_to_ln(30)
    mag[1] *= 3.0e4                                                           # drive the upper clamp
    out, lin, plan = _mel(mag, W, log_kind=K.LOG_E, log_offset=1e-6, clamp_lo=lo, clamp_hi=hi)
    lin64 = np.matmul(W.astype(np.float64), mag.astype(np.float64))
    ref = np.clip(np.log(lin64 + 1e-6), lo, hi)
    assert np.abs(lin - lin64).max() <= 2e-6 * np.abs(lin64).max()
    assert np.abs(out - ref).max() <= 2e-5
    assert (out == np.float32(lo)).any() and (out == np.float32(hi)).any()
    # backward: gmag = W^T (gout * [lo <= y <= hi] / (lin + off))
    gout = g.randn(*out.shape).astype(np.float32)
    y = np.log(lin64 + 1e-6)
    dl = np.where((y < lo) | (y > hi), 0.0, 1.0 / (lin64 + 1e-6))
    gref = np.einsum('mk,nmf->nkf', W.astype(np.float64), gout * dl)
    got = K.mel_backward(torch.from_numpy(gout).to(dev), torch.from_numpy(lin).to(dev), plan, n_fft // 2 + 1,
                         K.LOG_E, 1e-6, None, lo, hi).cpu().numpy()
    # entries within float32 round-off of a clamp edge may legitimately flip; none here by construction
    assert np.abs(got - gref).max() <= 3e-6 * np.abs(gref).max()


def test_mel_dense_matrix_and_log10():
    """A dense (non-banded) projection and the Audio2Mel epilogue log10(clamp(.,1e-5))."""
    K = _k()
    g = np.random.RandomState(3)
    W = g.rand(24, 129).astype(np.float32)
    mag = np.abs(g.randn(2, 129, 77)).astype(np.float32) * 1e-3
    mag[0, :, :5] = 0
    out, lin, _ = _mel(mag, W, log_kind=K.LOG_10, log_offset=0.0, pre_clamp_min=1e-5)
    lin64 = np.matmul(W.astype(np.float64), mag.astype(np.float64))
    assert np.abs(out - np.log10(np.maximum(lin64, 1e-5))).max() <= 2e-5


def test_logmel_golden_reference(golden):
    """LogMelSpectrogram.forward of the imported reference (G4), through stft_fwd + mel_fwd."""
    K = _k()
    dev = _dev()
    g = golden('logmel')
    for name in ['default', 'noclamp', 'zero_db_disables', 'silence']:
        kw = g[name + '/kw']
        sr, M, n_fft, win, hop = (int(v) for v in kw[:5])
        min_db = None if np.isnan(kw[5]) else kw[5]
        max_db = None if np.isnan(kw[6]) else kw[6]
        wav = torch.from_numpy(g[name + '/wav']).to(dev)
        plan = K.stft_plan(win, ofe.analysis_window(win)).to(dev)
        mag = K.stft_forward(wav, win, hop, plan)['mag']
        mplan = K.mel_plan(g[name + '/mel_filter']).to(dev)
        out, _ = K.mel_forward(mag, mplan, M, K.LOG_E, 1e-6, None,
                               ofe.db_to_ln(min_db) if min_db else None, ofe.db_to_ln(max_db) if max_db else None)
        assert np.abs(out.cpu().numpy() - g[name + '/mel']).max() <= 2e-4, name


def test_logmel_fused_golden_reference(golden):
    """the same golden cases through the ONE-kernel path psnd_logmel_fwd (magnitude kept in LDS)"""
    K = _k()
    dev = _dev()
    g = golden('logmel')
    ran = 0
    for name in ['default', 'noclamp', 'zero_db_disables', 'silence']:
        kw = g[name + '/kw']
        sr, M, n_fft, win, hop = (int(v) for v in kw[:5])
        if win != 1024 or hop > 256:
            continue
        min_db = None if np.isnan(kw[5]) else kw[5]
        max_db = None if np.isnan(kw[6]) else kw[6]
        wav = torch.from_numpy(g[name + '/wav']).to(dev)
        plan = K.stft_plan(win, ofe.analysis_window(win)).to(dev)
        mplan = K.mel_plan(g[name + '/mel_filter']).to(dev)
        out = K.logmel_forward(wav, win, hop, plan, mplan, M, K.FRAMING_CENTER, 0.0, K.LOG_E, 1e-6, None,
                               ofe.db_to_ln(min_db) if min_db else None, ofe.db_to_ln(max_db) if max_db else None)
        assert np.abs(out.cpu().numpy() - g[name + '/mel']).max() <= 2e-4, name
        ran += 1
    assert ran >= 2


@pytest.mark.parametrize('N,T,hop,framing,M', [(3, 44100, 256, 0, 80), (2, 8192, 256, 1, 80), (1, 5000, 200, 0, 40),
                                                (5, 2 * 4096 + 77, 128, 0, 80), (1, 600, 256, 0, 80)])
def test_logmel_fused_equals_two_kernel_path(N, T, hop, framing, M):
    """psnd_logmel_fwd == psnd_stft_fwd -> psnd_mel_fwd on the same inputs (same arithmetic, same summation order in the
    mel product: <= 1 ulp-level differences, 2e-6 abs on the log scale) and vs the float64 oracle (2e-5, as the unfused path)"""
    K = _k()
    dev = _dev()
    wav_np = seeded_wav(N + hop, N, T)
    wav = torch.from_numpy(wav_np).to(dev)
    plan = K.stft_plan(1024, ofe.analysis_window(1024)).to(dev)
    W = ofe.mel_filterbank(22050, 1024, M, 0.0, 8000.0)
    mplan = K.mel_plan(W).to(dev)
    lo, hi = ofe.db_to_ln(-50), ofe.db_to_ln(30)
    fused = K.logmel_forward(wav, 1024, hop, plan, mplan, M, framing, 1e-9, K.LOG_E, 1e-6, None, lo, hi)
    mag = K.stft_forward(wav, 1024, hop, plan, framing=framing, mag_eps=1e-9)['mag']
    two, _ = K.mel_forward(mag, mplan, M, K.LOG_E, 1e-6, None, lo, hi)
    assert fused.shape == two.shape
    assert float((fused - two).abs().max()) <= 2e-6
    mag64 = ofe.stft_mag_f64(wav_np, 1024, hop, framing=framing, eps=1e-9)
    ref = np.clip(np.log(np.matmul(W.astype(np.float64), mag64) + 1e-6), lo, hi)
    assert np.abs(fused.cpu().numpy() - ref).max() <= 2e-5


def test_logmel_module_takes_the_fused_path():
    from pytorch_sound_amd.models.transforms import LogMelSpectrogram
    K = _k()
    dev = _dev()
    m = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0.0, 8000.0).to(dev)
    wav = torch.from_numpy(seeded_wav(5, 2, 22050)).to(dev)
    y = m(wav)                                         # no grad wanted -> fused
    wg = wav.clone().requires_grad_(True)
    y2 = m(wg)                                         # differentiable -> stft_fwd + mel_fwd with autograd
    assert y2.requires_grad and not y.requires_grad
    assert float((y - y2.detach()).abs().max()) <= 2e-6
    y2.sum().backward()
    assert bool(torch.isfinite(wg.grad).all())


# ------------------------------------------------------------------------------------------------
# STFT backward (adjoint) - psnd_stft_bwd
# ------------------------------------------------------------------------------------------------
BWD_CASES = [
    (1024, 256, None, 0, 2, 2500),
    (1024, 256, None, 0, 3, 9000),        # several tiles: plain-store interior + atomic head/tail
    (1024, 256, 800, 1, 2, 4096),
    (1024, 300, None, 0, 1, 5000),
    (1024, 2000, None, 0, 1, 9000),       # hop > n_fft: span does not fit -> direct atomics path
    (512, 128, None, 0, 2, 3000),         # magnitude gradient: span kernel with two lane-pass rounds; (re, im): two-pass kernel
    (512, 50, 240, 0, 3, 5000),           # multi_stft_loss resolution: ten frames per sample, hop not a multiple of 4
    (512, 128, None, 0, 40, 2600),        # many clips, partial last tile
    (512, 200, None, 1, 2, 9000),         # hop does not divide n_fft
    (512, 129, None, 0, 2, 3000),         # odd hop -> two-pass kernel
    (256, 64, 200, 1, 3, 1500),
    (2048, 512, None, 0, 1, 6000),        # magnitude gradient: span kernel with 512 threads; (re, im): two-pass kernel
    (2048, 240, 1200, 0, 3, 9000),        # multi_stft_loss resolution
    (2048, 512, None, 0, 20, 20000),      # many clips, partial last tile
    (2048, 1024, None, 1, 2, 30000),      # hop = n / 2
    (2048, 600, None, 0, 1, 20000),       # hop does not divide n_fft
    (2048, 513, None, 0, 1, 9000),        # odd hop -> two-pass kernel
    (4096, 1024, None, 0, 1, 9000),       # magnitude gradient: adjoint of the 4-frame n4096 kernel; (re, im): generic path
    (4096, 1024, None, 0, 3, 44100),      # several tiles per clip, partial last tile, plain-store interior + atomic rest
    (4096, 1024, 3000, 1, 2, 30000),      # HiFi-GAN framing, short window
    (4096, 512, None, 0, 1, 20000),       # hop = n / 8: eight frames per sample
    (4096, 1022, None, 0, 1, 20000),      # hop does not divide n_fft: generic path
    (64, 16, None, 0, 2, 300),            # generic path
    (1024, 256, None, 0, 1, 513),
]


def _bwd(wav_np, n_fft, hop, win, framing, gmag=None, gre=None, gim=None, mag_eps=0.0):
    K = _k()
    dev = _dev()
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft, win)).to(dev)
    tt = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    gw = K.stft_backward(tt(wav_np), n_fft, hop, plan, framing, mag_eps, tt(gmag), tt(gre), tt(gim))
    torch.cuda.synchronize()
    return gw.cpu().numpy()


@pytest.mark.parametrize('n_fft,hop,win,framing,N,T', BWD_CASES)
def test_stft_bwd_vs_oracle(n_fft, hop, win, framing, N, T):
    wav = seeded_wav(n_fft + 7 * hop + T, N, T)
    F = ofe.frame_count(T, n_fft, hop, framing)
    g = np.random.RandomState(T)
    gmag = g.randn(N, n_fft // 2 + 1, F).astype(np.float32)
    gre = g.randn(N, n_fft // 2 + 1, F).astype(np.float32)
    gim = g.randn(N, n_fft // 2 + 1, F).astype(np.float32)
    # (a) through re/im (linear map - tight tolerance)
    ref = ofe.stft_reim_bwd_f64(gre, gim, T, n_fft, hop, win, framing)
    got = _bwd(wav, n_fft, hop, win, framing, gre=gre, gim=gim)
    assert got.shape == (N, T)
    assert np.abs(got - ref).max() <= 4e-6 * np.abs(ref).max()
    # (b) through the magnitude (needs X/|X| of the recomputed spectrum)
    ref = ofe.stft_mag_bwd_f64(gmag, wav, n_fft, hop, win, framing)
    got = _bwd(wav, n_fft, hop, win, framing, gmag=gmag)
    assert np.abs(got - ref).max() <= 5e-5 * np.abs(ref).max()
    # (c) both at once = sum
    got2 = _bwd(wav, n_fft, hop, win, framing, gmag=gmag, gre=gre, gim=gim)
    ref2 = ref + ofe.stft_reim_bwd_f64(gre, gim, T, n_fft, hop, win, framing)
    assert np.abs(got2 - ref2).max() <= 5e-5 * np.abs(ref2).max()


def test_stft_bwd_golden_reference(golden):
    """autograd of the imported reference through pad + conv1d + sqrt (G1 'bwd')."""
    g = golden('stft')
    got = _bwd(g['bwd/wav'], 1024, 256, None, 0, gmag=g['bwd/gmag'])
    assert np.abs(got - g['bwd/gwav']).max() <= 1e-4 * np.abs(g['bwd/gwav']).max()


def test_stft_bwd_zero_bin_is_nan_like_reference():
    """sqrt has no eps in STFT.transform (transforms.py:67): an exactly-zero bin gives NaN grads in the
    reference (inf * 0); mag_eps > 0 (interface MelSpectrogram's 1e-9) gives finite ones."""
    wav = np.zeros((1, 3000), np.float32)
    gmag = np.ones((1, 513, ofe.frame_count(3000, 1024, 256, 0)), np.float32)
    assert np.isnan(_bwd(wav, 1024, 256, None, 0, gmag=gmag)).all()
    assert np.isfinite(_bwd(wav, 1024, 256, None, 0, gmag=gmag, mag_eps=1e-9)).all()


def test_stft_adjoint_identity_full_size():
    """<A x, G> == <x, A^T G> at BASELINE config 2 size (32 x 2 s), A = wav -> (re, im)."""
    K = _k()
    dev = _dev()
    n_fft, hop = 1024, 256
    x = torch.from_numpy(seeded_wav(5, 32, 44100)).to(dev)
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft)).to(dev)
    o = K.stft_forward(x, n_fft, hop, plan, want_mag=False, want_reim=True)
    gen = torch.Generator(device='cpu').manual_seed(3)
    gre = torch.randn(o['re'].shape, generator=gen).to(dev)
    gim = torch.randn(o['im'].shape, generator=gen).to(dev)
    gw = K.stft_backward(x, n_fft, hop, plan, gre=gre, gim=gim)
    lhs = (o['re'].double() * gre.double()).sum() + (o['im'].double() * gim.double()).sum()
    rhs = (x.double() * gw.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-5 * abs(float(lhs)) + 1e-3


# ------------------------------------------------------------------------------------------------
# inverse STFT - psnd_istft (STFT.inverse, "next" row f1) and the no-padding framing it needs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n_fft,hop,win,N,T', [(1024, 256, None, 2, 4096), (1024, 256, 800, 2, 2816), (512, 128, None, 3, 1024),
                                               (512, 128, None, 2, 9000), (512, 64, 300, 2, 5000),
                                               (256, 64, 200, 2, 640), (2048, 512, None, 1, 6144), (4096, 1024, None, 1, 8192),
                                               (1024, 256, None, 5, 9000)])
def test_istft_vs_oracle_and_roundtrip(n_fft, hop, win, N, T):
    K = _k()
    dev = _dev()
    wav = seeded_wav(T + n_fft, N, T)
    re, im = ofe.stft_reim_f64(wav, n_fft, hop, win)
    mag, phase = np.sqrt(re * re + im * im), np.arctan2(im, re)
    ref = ofe.istft_f64(mag, phase, n_fft, hop, win)
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft, win)).to(dev)
    got = K.istft(torch.from_numpy(mag.astype(np.float32)).to(dev), torch.from_numpy(phase.astype(np.float32)).to(dev),
                  n_fft, hop, plan).cpu().numpy()
    assert got.shape == ref.shape == (N, (mag.shape[2] - 1) * hop)
    assert np.abs(got - ref).max() <= 5e-6 * max(np.abs(ref).max(), 1.0)
    if win is None:      # COLA window: the inverse reproduces the waveform it came from
        L = got.shape[1]
        assert np.abs(got[:, :min(L, T)] - wav[:, :min(L, T)]).max() <= 1e-5


def test_istft_golden_reference(golden):
    """STFT.inverse of the imported reference (pinv synthesis basis + envelope), G3."""
    K = _k()
    dev = _dev()
    g = golden('stft')
    for name in ['n1024_h256', 'n1024_h256_w800', 'n512_h128', 'n256_h64_w200']:
        n, h, w = (int(v) for v in g[name + '/params'])
        plan = K.stft_plan(n, ofe.analysis_window(n, w)).to(dev)
        got = K.istft(torch.from_numpy(g[name + '/mag']).to(dev), torch.from_numpy(g[name + '/phase']).to(dev), n, h, plan)
        assert np.abs(got.cpu().numpy() - g[name + '/inverse']).max() <= 5e-6, name


def test_stft_nopad_framing_and_istft_backward():
    K = _k()
    dev = _dev()
    n_fft, hop = 1024, 256
    wav = seeded_wav(77, 2, 5000)
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft)).to(dev)
    o = K.stft_forward(torch.from_numpy(wav).to(dev), n_fft, hop, plan, K.FRAMING_NONE, want_mag=False, want_reim=True)
    re, im = ofe.stft_reim_f64(wav, n_fft, hop, None, ofe.NOPAD)
    assert o['re'].shape == re.shape == (2, 513, (5000 - 1024) // 256 + 1)
    tol = FFT_RTOL * np.sqrt(re * re + im * im).max()
    assert np.abs(o['re'].cpu().numpy() - re).max() <= tol and np.abs(o['im'].cpu().numpy() - im).max() <= tol
    # gradient of sum(g * istft(mag, phase)) wrt mag and phase vs finite differences of the float64 oracle
    F = 9
    rs = np.random.RandomState(1)
    mag = np.abs(rs.randn(1, 513, F)) + 0.1
    phase = rs.uniform(-3, 3, (1, 513, F))
    gout = rs.randn(1, (F - 1) * hop)
    mt = torch.from_numpy(mag.astype(np.float32)).to(dev).requires_grad_(True)
    pt = torch.from_numpy(phase.astype(np.float32)).to(dev).requires_grad_(True)
    win = torch.from_numpy(ofe.analysis_window(n_fft)).to(dev)
    out = K.istft(mt, pt, n_fft, hop, plan, 1e-9, win)
    (out * torch.from_numpy(gout.astype(np.float32)).to(dev)).sum().backward()
    f = lambda m, p: float((ofe.istft_f64(m, p, n_fft, hop) * gout).sum())  # noqa: E731
    for (k, fr) in [(0, 0), (5, 3), (100, 4), (512, 8), (256, 1)]:
        for arr, grad in ((mag, mt.grad), (phase, pt.grad)):
            d = np.zeros_like(arr)
            d[0, k, fr] = 1e-4
            num = (f(mag + d if arr is mag else mag, phase + d if arr is phase else phase)
                   - f(mag - d if arr is mag else mag, phase - d if arr is phase else phase)) / 2e-4
            assert abs(float(grad[0, k, fr]) - num) <= 2e-3 * max(1.0, abs(num)), (k, fr)


def test_stft_module_inverse_roundtrip():
    from pytorch_sound_amd.models.transforms import STFT
    dev = _dev()
    m = STFT(1024, 256).to(dev)
    wav = torch.from_numpy(seeded_wav(3, 2, 8192)).to(dev)
    mag, phase = m.transform(wav)
    rec = m.inverse(mag, phase)
    assert rec.shape == (2, 8192) and float((rec - wav).abs().max()) <= 1e-5
