"""PQMF / MFCC / SpectrogramMasker: oracle and the drop-in's host path against fixtures from the imported
reference (tests/golden/filters.npz); DCT known answers (values unpinned by the reference - torchaudio is absent)."""
import os
import numpy as np
import pytest
import torch

from oracle import filters as of

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'filters.npz'))
CASES = [('s4', dict()), ('s8', dict(subbands=8, taps=126, cutoff_ratio=0.07, beta=10.0))]


@pytest.mark.parametrize('tag,kw', CASES)
def test_oracle_pqmf_matches_reference(tag, kw):
    ha, hs = of.pqmf_filters(**kw)
    assert np.abs(ha - G[tag + '/analysis_filter'][:, 0]).max() < 1e-7 and np.abs(hs - G[tag + '/synthesis_filter'][0]).max() < 1e-7
    assert np.abs(of.pqmf_analysis(G[tag + '/x'], ha) - G[tag + '/analysis']).max() < 1e-6
    assert np.abs(of.pqmf_synthesis(G[tag + '/analysis'], hs) - G[tag + '/synthesis']).max() < 1e-6


@pytest.mark.parametrize('tag,kw', CASES)
def test_module_host_path_matches_reference(tag, kw):
    from pytorch_sound_amd.models.transforms import PQMF
    pq = PQMF(**kw)
    assert set(pq.state_dict()) == {'analysis_filter', 'synthesis_filter', 'updown_filter'}
    assert np.abs(pq.analysis_filter.numpy() - G[tag + '/analysis_filter']).max() < 1e-7
    x = torch.from_numpy(G[tag + '/x']).requires_grad_(True)
    a = pq.analysis(x)
    (a * torch.from_numpy(G[tag + '/ga'])).sum().backward()
    assert np.abs(a.detach().numpy() - G[tag + '/analysis']).max() < 1e-6 and np.abs(x.grad.numpy() - G[tag + '/gx']).max() < 1e-5
    y = pq.synthesis(torch.from_numpy(G[tag + '/analysis']))
    assert np.abs(y.numpy() - G[tag + '/synthesis']).max() < 1e-6
    with pytest.raises(AssertionError):
        PQMF(taps=61)


def test_pqmf_reconstruction_property():
    """synthesis(analysis(x)) reproduces x delayed by one sample (taps even, P = taps / 2 padding on both sides); with the
    reference's default prototype (cutoff 0.15, 4 bands) the reconstruction is approximate: correlation > 0.98 on white noise"""
    ha, hs = of.pqmf_filters()
    x = np.random.RandomState(0).randn(1, 1, 4096)
    y = of.pqmf_synthesis(of.pqmf_analysis(x, ha), hs)
    assert np.corrcoef(y[0, 0, 200:-200], x[0, 0, 199:-201])[0, 1] > 0.98
    assert abs(np.corrcoef(y[0, 0, 200:-200], x[0, 0, 200:-200])[0, 1]) < 0.1


def test_dct_known_answers_and_mfcc_host_path():
    from pytorch_sound_amd.models.transforms import MelToMFCC, create_dct
    D = of.create_dct(13, 80)
    assert np.abs(D.T @ D - np.eye(13)).max() < 1e-12                       # orthonormal columns
    assert np.allclose(D[:, 0], 1.0 / np.sqrt(80))                          # first basis vector constant
    assert np.allclose(of.create_dct(4, 8, None)[:, 1], 2 * np.cos(np.pi / 8 * (np.arange(8) + 0.5)))
    assert np.abs(create_dct(13, 80).numpy() - D).max() < 1e-7
    m = MelToMFCC(13, 80)
    assert m.dct_mat.shape == (13, 80) and set(m.state_dict()) == {'dct_mat'}
    mel = torch.randn(2, 80, 17)
    assert np.abs(m(mel).numpy() - of.mel_to_mfcc(mel.numpy(), 13)).max() < 1e-5
    with pytest.raises(AssertionError):
        m(torch.randn(80, 17))


def test_spectrogram_masker_host():
    from pytorch_sound_amd.models.transforms import SpectrogramMasker
    sm = SpectrogramMasker(win_length=8, hop_length=4)
    mask = torch.zeros(2, 32)
    mask[0, :10] = 1
    mask[1, :] = 1
    out = sm(mask)
    # frames = (32 + 8 - 8) / 4 + 1 = 9; clip 0: frames touching a valid sample (or the ones-padded left edge) are 1
    assert out.shape == (2, 9) and out[1].tolist() == [1.0] * 9
    assert out[0].tolist() == [1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0]


def test_learnable_stft_matches_reference():
    """LearnableSTFT (transforms.py:104-203): buffers / parameters, transform, inverse and the gradients wrt both trainable
    bases against the imported reference (tests/golden/lstft.npz)."""
    from pytorch_sound_amd.models.transforms import LearnableSTFT
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'lstft.npz'))
    m = LearnableSTFT(256, 64, 200)
    assert sorted(m.state_dict().keys()) == list(g['state_keys'])
    assert isinstance(m.forward_basis, torch.nn.Parameter) and isinstance(m.inverse_basis, torch.nn.Parameter)
    mag, phase = m.transform(torch.from_numpy(g['wav']))
    rec = m.inverse(mag, phase)
    (mag.sum() + rec.pow(2).sum()).backward()
    assert np.abs(mag.detach().numpy() - g['mag']).max() < 2e-5 * np.abs(g['mag']).max()
    strong = g['mag'] > 1e-3 * g['mag'].max()
    d = np.angle(np.exp(1j * (phase.numpy() - g['phase'])))
    assert np.abs(d[strong]).max() < 1e-3
    assert np.abs(rec.detach().numpy() - g['rec']).max() < 1e-4 * np.abs(g['rec']).max()
    rows = [0, 1, 64, 129, 200, 257]
    gf, gi = m.forward_basis.grad.numpy()[rows], m.inverse_basis.grad.numpy()[rows]
    assert np.abs(gf - g['g_forward_basis_rows']).max() < 1e-3 * max(np.abs(g['g_forward_basis_rows']).max(), 1e-6)
    assert np.abs(gi - g['g_inverse_basis_rows']).max() < 1e-3 * max(np.abs(g['g_inverse_basis_rows']).max(), 1e-6)
    frozen = LearnableSTFT(256, 64, trainable_inverse=False, trainable_forward=False)
    assert not list(frozen.parameters()) and len(frozen.state_dict()) == 3


def test_htk_filterbank_known_answers():
    from pytorch_sound_amd.utils.mel import mel_filterbank_htk
    fb = of.mel_filterbank_htk(22050, 1024, 80, 0.0, 8000.0)
    assert fb.shape == (80, 513) and fb.min() >= 0.0 and fb.max() <= 1.0
    hz = np.linspace(0, 11025, 513)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)            # noqa: E731
    centres = 700.0 * (10.0 ** (np.linspace(mel(0.0), mel(8000.0), 82) / 2595.0) - 1.0)[1:-1]
    peak = hz[fb.argmax(1)]
    assert np.abs(peak - centres).max() <= 11025 / 512              # every triangle peaks at its centre frequency (bin resolution)
    assert fb[:, hz > 8000.0 + 22].max() == 0.0                     # nothing above fmax
    inner = (hz > centres[0]) & (hz < centres[-1])
    assert np.allclose(fb[:, inner].sum(0), 1.0, atol=1e-9)         # neighbouring unit-peak triangles sum to one in between
    assert np.abs(mel_filterbank_htk(22050, 1024, 80, 0.0, 8000.0) - fb).max() < 1e-6
