"""PQMF / MelToMFCC / MFCC / SpectrogramMasker on the GPU: the polyphase kernels against the reference's golden outputs and
gradients (tests/golden/filters.npz) and the float64 oracle on ragged sizes; adjoint identity <A x, g> = <x, A^T g> at a
large size; the DCT on the matrix-core mel kernel against the oracle."""
import numpy as np
import pytest
import torch

from oracle import filters as of

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = [('s4', dict()), ('s8', dict(subbands=8, taps=126, cutoff_ratio=0.07, beta=10.0))]


@pytest.mark.parametrize('tag,kw', CASES)
def test_pqmf_golden(golden, tag, kw):
    from pytorch_sound_amd.models.transforms import PQMF
    g = golden('filters')
    pq = PQMF(**kw).to(DEV)
    x = torch.from_numpy(g[tag + '/x']).to(DEV).requires_grad_(True)
    a = pq.analysis(x)
    (a * torch.from_numpy(g[tag + '/ga']).to(DEV)).sum().backward()
    assert a.shape == g[tag + '/analysis'].shape
    assert np.abs(a.detach().cpu().numpy() - g[tag + '/analysis']).max() < 2e-6
    assert np.abs(x.grad.cpu().numpy() - g[tag + '/gx']).max() < 1e-5
    sb = torch.from_numpy(g[tag + '/analysis']).to(DEV).requires_grad_(True)
    y = pq.synthesis(sb)
    (y * torch.from_numpy(g[tag + '/gy']).to(DEV)).sum().backward()
    assert y.shape == g[tag + '/synthesis'].shape
    assert np.abs(y.detach().cpu().numpy() - g[tag + '/synthesis']).max() < 2e-6
    assert np.abs(sb.grad.cpu().numpy() - g[tag + '/gsb']).max() < 2e-5


@pytest.mark.parametrize('B,T,kw', [(3, 1000, dict()), (1, 257, dict(subbands=2, taps=30, cutoff_ratio=0.25)), (2, 8192 + 5, CASES[1][1]),
                                    (1, 5, dict())])
def test_pqmf_vs_oracle_ragged(B, T, kw):
    from pytorch_sound_amd.models.transforms import PQMF
    pq = PQMF(**kw).to(DEV)
    ha, hs = pq.analysis_filter.squeeze(1).double().cpu().numpy(), pq.synthesis_filter.squeeze(0).double().cpu().numpy()
    x = np.random.RandomState(T).randn(B, 1, T).astype(np.float32)
    a = pq.analysis(torch.from_numpy(x).to(DEV))
    want = of.pqmf_analysis(x, ha)
    assert a.shape == want.shape and np.abs(a.cpu().numpy() - want).max() < 2e-6 * max(1.0, np.abs(want).max())
    y = pq.synthesis(a)
    wy = of.pqmf_synthesis(a.cpu().numpy(), hs)
    assert y.shape == wy.shape and np.abs(y.cpu().numpy() - wy).max() < 4e-6 * max(1.0, np.abs(wy).max())


def test_pqmf_adjoint_identity_large():
    from pytorch_sound_amd.models.transforms import PQMF
    pq = PQMF().to(DEV)
    x = torch.randn(64, 1, 1 << 18, device=DEV, requires_grad=True)
    a = pq.analysis(x)
    g = torch.randn_like(a)
    (a * g).sum().backward()
    lhs = float((a.detach().double() * g.double()).sum())
    rhs = float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    sb = a.detach().requires_grad_(True)
    y = pq.synthesis(sb)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    lhs = float((y.detach().double() * gy.double()).sum())
    rhs = float((sb.detach().double() * sb.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


def test_mfcc_on_the_mel_kernel():
    from pytorch_sound_amd.models.transforms import MelToMFCC, MFCC
    m = MelToMFCC(13, 80).to(DEV)
    mel = torch.randn(4, 80, 173, device=DEV, requires_grad=True)
    out = m(mel)
    g = torch.randn_like(out)
    (out * g).sum().backward()
    want = of.mel_to_mfcc(mel.detach().cpu().numpy(), 13)
    assert out.shape == (4, 13, 173) and np.abs(out.detach().cpu().numpy() - want).max() < 2e-6 * np.abs(want).max()
    gw = np.matmul(of.create_dct(13, 80), g.cpu().numpy().astype(np.float64))
    assert np.abs(mel.grad.cpu().numpy() - gw).max() < 2e-6 * np.abs(gw).max()
    mf = MFCC(22050, 80, 1024, 1024, 13, 256, -80.0, 20.0).to(DEV)
    wav = 0.1 * torch.randn(2, 1, 8192, device=DEV)
    c = mf(wav)
    ref = m(mf.mel_func(wav.squeeze(1)))
    assert c.shape == (2, 13, 33) and torch.equal(c, ref)


def test_spectrogram_masker_device():
    from pytorch_sound_amd.models.transforms import SpectrogramMasker
    sm = SpectrogramMasker(1024, 256)
    mask = torch.zeros(3, 22050, device=DEV)
    mask[0, :5000] = 1
    mask[1, :] = 1
    out = sm(mask)
    assert out.device.type == 'cuda' and out.shape == (3, 87)
    assert out[1].min() == 1 and out[2, 3:].max() == 0 and out[0].sum() == np.ceil((5000 + 512) / 256)
    # against the reference's formulation (constant-weight conv + ceil) on the host, incl. windows that are not a power of two,
    # hops that do not divide them, fractional masks, and with the library ops forbidden for HIP tensors
    from test_gpu_no_library_paths import forbid_library_ops
    torch.manual_seed(3)
    for win, hop, T in ((1024, 256, 22050), (600, 120, 9000), (512, 50, 4001), (8, 4, 32), (2048, 512, 70000)):
        sm = SpectrogramMasker(win, hop)
        lens = torch.randint(1, T, (5,))
        m = (torch.arange(T)[None, :] < lens[:, None]).float()
        m[4] = 1.0
        m[3] = torch.rand(T) * (torch.arange(T) < lens[3])          # fractional values: any positive mean rounds up to 1
        ref = sm(m)                                                 # host: the conv formulation
        with forbid_library_ops():
            got = sm(m.to(DEV))
        assert got.shape == ref.shape and got.device.type == 'cuda'
        # (a fully valid frame is 1 + rounding in the conv formulation when 1 / win is not exact: ceil may give 2 there; the kernel's
        # exact division gives 1)
        assert float(got.max()) <= 1.0 and torch.equal(got.cpu(), ref.clamp(max=1.0))


def test_logmel_torchaudio_variant_vs_oracle():
    """LogMelSpectrogramTorchAudio: power spectrogram (hann(win) centre-padded to n_fft) x HTK triangles -> ln -> clamp, against the
    float64 oracle (STFT magnitude oracle squared, oracle/filters.py filterbank); forward 2e-5 on the log scale, waveform gradient 2e-4."""
    from pytorch_sound_amd.models.transforms import LogMelSpectrogramTorchAudio
    from conftest import seeded_wav
    from oracle import features as fe
    m = LogMelSpectrogramTorchAudio(22050, 80, 1024, 800, 256, -80.0, 20.0, 0.0, 8000.0).to(DEV)
    wav = seeded_wav(970, 3, 6000)
    x = torch.from_numpy(wav).to(DEV).requires_grad_(True)
    y = m(x)
    win = fe.pad_center(m.stft.window.cpu().numpy().astype(np.float64), 1024)
    mag = fe.stft_mag_f64(wav, 1024, 256, framing=fe.CENTER, window=win)
    fb = of.mel_filterbank_htk(22050, 1024, 80, 0.0, 8000.0)
    lin = np.matmul(fb, mag ** 2)
    want = np.clip(np.log(lin + 1e-6), np.log(10 ** -8.0), np.log(10 ** 2.0))
    assert y.shape == want.shape and np.abs(y.detach().cpu().numpy() - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    g = np.random.RandomState(1).randn(*want.shape)
    (y * torch.from_numpy(g).to(DEV)).sum().backward()
    inside = (np.log(lin + 1e-6) > np.log(10 ** -8.0)) & (np.log(lin + 1e-6) < np.log(10 ** 2.0))
    gmag = np.matmul(fb.T, g * inside / (lin + 1e-6)) * 2.0 * mag
    gw = fe.stft_mag_bwd_f64(gmag, wav, 1024, 256, framing=fe.CENTER, window=win)
    assert np.abs(x.grad.cpu().numpy() - gw).max() <= 2e-4 * np.abs(gw).max()
