"""psnd_stft_mag_nfk: the magnitude with the BIN axis fastest, (N, F, K) - against the float64 oracle (transposed), the reference
goldens, the impulse (bit-exact framing) contract and the (N, K, F) kernels on the same input.  Tolerance: as test_gpu_features
(4e-6 of the largest bin: fp32 FFT round-off)."""
import numpy as np
import pytest
import torch

from conftest import seeded_wav
from oracle import features as ofe

pytestmark = pytest.mark.gpu
FFT_RTOL = 4e-6
DEV = torch.device('cuda:0')


def _nfk(wav_np, n_fft, hop, win=None, framing=0, mag_eps=0.0, window=None):
    from pytorch_sound_amd import kernels as K
    w = ofe.analysis_window(n_fft, win) if window is None else np.asarray(window, np.float32)
    plan = K.stft_plan(n_fft, w).to(DEV)
    out = K.stft_mag_nfk(torch.from_numpy(wav_np).to(DEV), n_fft, hop, plan, framing, mag_eps)
    torch.cuda.synchronize()
    return out.cpu().numpy()


CASES = [
    # n_fft = 1024: the wave-per-four-frames kernel (psnd_stft_q.hip)
    (1024, 256, None, 0, 4, 44100),      # BASELINE config 1 / 2 clip: F = 173 (partial last quad / tile)
    (1024, 256, None, 0, 1, 8192),       # F = 33
    (1024, 256, 800, 0, 2, 3000),        # short window
    (1024, 256, None, 1, 3, 8192),       # HiFi-GAN framing (config 3)
    (1024, 256, None, 0, 1, 513),        # T barely above the pad: every frame reflects
    (1024, 128, None, 0, 2, 5000),
    (1024, 64, None, 0, 1, 4000),
    (1024, 200, None, 0, 2, 6000),       # hop % 4 == 0 but not a power of two
    (1024, 250, None, 0, 2, 6000),       # hop % 4 != 0
    (1024, 512, None, 0, 2, 9000),       # a hop the four-frame span does not cover -> generic kernel
    (1024, 256, None, 0, 37, 10000),     # many clips: persistent workgroups with several tiles
    # n_fft = 4096: the wave-per-frame kernel, stores straight from registers
    (4096, 1024, None, 0, 1, 9000),
    (4096, 1024, None, 0, 3, 44100),
    (4096, 1024, 3000, 1, 2, 30000),
    (4096, 1022, None, 0, 1, 20000),
    (4096, 512, None, 0, 2, 40000),
    (4096, 1024, None, 0, 1, 2049),
    (4096, 1024, None, 0, 21, 60000),
    # the other sizes: one frame per workgroup
    (512, 128, None, 0, 2, 3000), (2048, 512, None, 0, 2, 9000), (256, 64, 200, 1, 2, 1500), (64, 16, None, 0, 1, 300),
]


@pytest.mark.parametrize('n_fft,hop,win,framing,N,T', CASES)
def test_nfk_vs_oracle(n_fft, hop, win, framing, N, T):
    wav = seeded_wav(n_fft + hop + T, N, T)
    got = _nfk(wav, n_fft, hop, win, framing)
    ref = ofe.stft_mag_f64(wav, n_fft, hop, win, framing)
    assert got.shape == (N, ofe.frame_count(T, n_fft, hop, framing), n_fft // 2 + 1)
    assert np.abs(got.transpose(0, 2, 1) - ref).max() <= FFT_RTOL * np.abs(ref).max()
    got_eps = _nfk(wav, n_fft, hop, win, framing, mag_eps=1e-9)
    assert np.abs(got_eps.transpose(0, 2, 1) - np.sqrt(ref * ref + 1e-9)).max() <= FFT_RTOL * np.abs(ref).max()


def test_nfk_every_element_written_and_nothing_else():
    """NaN-filled output with guard rows on both sides: every element of (N, F, K) is written, nothing outside it."""
    from pytorch_sound_amd import kernels as K
    for n_fft, hop, N, T in [(1024, 256, 5, 44100), (4096, 1024, 3, 50000), (512, 128, 2, 3000)]:
        wav = torch.from_numpy(seeded_wav(7, N, T)).to(DEV)
        plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft)).to(DEV)
        F, Kb = K.frame_count(T, n_fft, hop), n_fft // 2 + 1
        buf = torch.full((N * F * Kb + 2 * 4096,), float('nan'), device=DEV)
        out = buf[4096:4096 + N * F * Kb].view(N, F, Kb)
        K.stft_mag_nfk(wav, n_fft, hop, plan, out=out)
        assert torch.isfinite(out).all()
        assert torch.isnan(buf[:4096]).all() and torch.isnan(buf[-4096:]).all()


@pytest.mark.parametrize('n_fft', [1024, 4096])
def test_nfk_frame_indexing_bit_exact(n_fft):
    """impulse at sample p, all-ones window: |DC| of frame f = the number of taps of frame f that read p (transforms.py:55-66)."""
    hop = n_fft // 4
    for T in (5 * n_fft - 37, n_fft // 2 + 1):
        for framing in (0, 1):
            F = ofe.frame_count(T, n_fft, hop, framing)
            idx = ofe.frame_sample_index(np.arange(F)[:, None], np.arange(n_fft)[None, :], T, n_fft, hop, framing)
            pad = ofe.pad_amount(n_fft, hop, framing)
            pos = sorted({0, 1, min(hop - 1, T - 1), min(hop, T - 1), min(pad - 1, T - 1), min(pad, T - 1), min(pad + 1, T - 1), T // 2,
                          max(T - pad - 1, 0), max(T - pad, 0), T - 2, T - 1})
            wav = np.zeros((len(pos), T), np.float32)
            for i, p in enumerate(pos):
                wav[i, p] = 1.0
            got = _nfk(wav, n_fft, hop, None, framing, window=np.ones(n_fft, np.float32))
            for i, p in enumerate(pos):
                assert np.array_equal(got[i, :, 0], (idx == p).sum(axis=1).astype(np.float32)), (n_fft, T, framing, p)


def test_nfk_golden_reference(golden):
    """the reference's own STFT.transform magnitudes (tests/golden/stft.npz), transposed"""
    g = golden('stft')
    for name in ('n1024_h256', 'n1024_h256_w800', 'n512_h128', 'n256_h64_w200', 'n2048_h512', 'n4096_h1024'):
        n, h, w = (int(v) for v in g[name + '/params'])
        got = _nfk(g[name + '/wav'], n, h, w)
        gm = g[name + '/mag']
        assert np.abs(got.transpose(0, 2, 1) - gm).max() <= 6e-6 * gm.max()


@pytest.mark.parametrize('n_fft,hop,N,T,sr', [(1024, 256, 64, 44100, 22050), (4096, 1024, 32, 1323000, 44100)])
def test_nfk_full_size_equals_nkf(n_fft, hop, N, T, sr):
    """BASELINE full sizes (config 2: 2 x 32 clips x 2 s; config 5: 32 x 30 s): equal to psnd_stft_fwd on the same input (which the
    oracle / Parseval tests of test_gpu_features pin), whole clips against the float64 oracle."""
    from pytorch_sound_amd import kernels as K
    g = torch.Generator(device='cpu').manual_seed(11)
    a = (0.0708 * torch.randn(N, T, generator=g)).to(DEV)
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft)).to(DEV)
    nkf = K.stft_forward(a, n_fft, hop, plan)['mag']
    nfk = K.stft_mag_nfk(a, n_fft, hop, plan)
    assert float((nfk.transpose(1, 2) - nkf).abs().max()) <= 8e-6 * float(nkf.max())
    for n in (0, N - 1):
        ref = ofe.stft_mag_f64(a[n:n + 1].cpu().numpy(), n_fft, hop)
        assert np.abs(nfk[n:n + 1].cpu().numpy().transpose(0, 2, 1) - ref).max() <= FFT_RTOL * np.abs(ref).max()


# ---- round 5: the LDS sample-ring kernel of config 5 (psnd_stft_r.hip: hop 1024, T % 4 == 0) ----------------------------------------------
RING_SHAPES = [
    # (N, T, framing): single frames, segments cut at clip boundaries (many short clips per workgroup), clips shorter than the pad (every
    # chunk a reflect gather), one long clip split over many workgroups, HiFi-GAN framing, config-5 clips
    (1, 4096, 0), (1, 2052, 0), (7, 2052, 0), (3, 44100, 0), (2, 30000, 1), (21, 60000, 0), (300, 8192, 0), (1000, 4100, 0),
    (5, 4096 * 40, 1), (1, 1323000, 0), (3, 1323000, 0),
]


@pytest.mark.parametrize('N,T,framing', RING_SHAPES)
def test_ring_kernel_is_bit_identical_to_the_register_load_kernel(N, T, framing, monkeypatch, lab_lib):
    """stft_fwd_n4096r_kernel (samples through the workgroup's LDS ring) and stft_fwd_n4096w_kernel<NFK> (every wave loads its frame) run the
    same arithmetic on the same values: equal bit for bit - and the latter is pinned to the oracle / goldens / impulse contract above."""
    from pytorch_sound_amd import kernels as K
    g = torch.Generator(device='cpu').manual_seed(N * 31 + T)
    x = (0.0708 * torch.randn(N, T, generator=g)).to(DEV)
    plan = K.stft_plan(4096, ofe.analysis_window(4096)).to(DEV)
    ring = K.stft_mag_nfk(x, 4096, 1024, plan, framing)
    monkeypatch.setenv('PSND_STFT4096_NORING', '1')
    regs = K.stft_mag_nfk(x, 4096, 1024, plan, framing)
    monkeypatch.delenv('PSND_STFT4096_NORING')
    assert torch.isfinite(ring).all()
    assert torch.equal(ring, regs)
    ref = ofe.stft_mag_f64(x[:1].cpu().numpy(), 4096, 1024, None, framing)
    assert np.abs(ring[:1].cpu().numpy().transpose(0, 2, 1) - ref).max() <= FFT_RTOL * np.abs(ref).max()


def test_ring_kernel_repeats_exactly():
    """the ring's hand-offs (loader wave -> frame waves, slot re-use, the transpose buffers' locks) are timing dependent; the result is not:
    30 launches on changing inputs, each equal to the first launch on the same input; poisoned output in between."""
    from pytorch_sound_amd import kernels as K
    plan = K.stft_plan(4096, ofe.analysis_window(4096)).to(DEV)
    g = torch.Generator(device='cpu').manual_seed(5)
    xs = [(0.0708 * torch.randn(n, t, generator=g)).to(DEV) for n, t in ((32, 330752), (9, 1323000), (130, 40000))]
    first = [K.stft_mag_nfk(x, 4096, 1024, plan).clone() for x in xs]
    outs = [torch.empty_like(f) for f in first]
    for rep in range(10):
        for x, f, o in zip(xs, first, outs):
            o.fill_(float('nan'))
            K.stft_mag_nfk(x, 4096, 1024, plan, out=o)
            assert torch.equal(o, f), rep


def test_q_kernel_full_chip_repeats_equal_nkf():
    """round 6: the n = 1024 (N, F, K) kernel issues its LDS table reads a phase ahead of their use with COUNTED waits; a count that is
    one too generous only shows when every CU is loaded (a first version passed every small shape and failed 1024 clips x 2 s).  The
    full-chip launch, three inputs, poisoned output, against the (N, K, F) kernel (pinned to the oracle in test_gpu_features)."""
    from pytorch_sound_amd import kernels as K
    n_fft, hop, N, T = 1024, 256, 1024, 44100
    plan = K.stft_plan(n_fft, ofe.analysis_window(n_fft)).to(DEV)
    for rep in range(3):
        g = torch.Generator(device='cpu').manual_seed(100 + rep)
        a = (0.0708 * torch.randn(N, T, generator=g)).to(DEV)
        nkf = K.stft_forward(a, n_fft, hop, plan)['mag']
        out = torch.full((N, K.frame_count(T, n_fft, hop), n_fft // 2 + 1), float('nan'), device=DEV)
        for _ in range(4):
            K.stft_mag_nfk(a, n_fft, hop, plan, out=out)
        assert not torch.isnan(out).any()
        assert float((out.transpose(1, 2) - nkf).abs().max()) <= 8e-6 * float(nkf.max()), rep
