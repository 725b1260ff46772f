"""Oracle (CPU restatement, numpy float64) of the small linear transforms next to the mel path in
pytorch_sound/models/transforms.py: the DCT of MelToMFCC / MFCC (:419-459) and the pseudo-QMF bank (:462-560).

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.  The PQMF restatement is pinned by tests/golden/filters.npz (generated
by importing the reference).  "Parity unpinned" for the DCT matrix VALUES: the reference takes them from
torchaudio.functional.create_dct (torchaudio is neither in the reference tree nor installed); restated here from the
published DCT-II definition and pinned by known answers (orthonormal rows, first row constant) in tests/.
"""
import numpy as np


def create_dct(n_mfcc, n_mels, norm='ortho'):
    """torchaudio.functional.create_dct: DCT-II basis, (n_mels, n_mfcc); norm None -> x2, 'ortho' -> orthonormal."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    dct = np.cos(np.pi / n_mels * (n + 0.5) * k)               # (n_mfcc, n_mels)
    if norm is None:
        dct *= 2.0
    else:
        assert norm == 'ortho'
        dct[0] *= 1.0 / np.sqrt(2.0)
        dct *= np.sqrt(2.0 / n_mels)
    return dct.T


def mel_to_mfcc(mel, n_mfcc, norm='ortho'):
    """transforms.py:430-432: matmul(dct_mat (n_mfcc, mel), mel_spec (N, mel, T))"""
    mel = np.asarray(mel, np.float64)
    return np.matmul(create_dct(n_mfcc, mel.shape[1], norm).T, mel)


def design_prototype_filter(taps=62, cutoff_ratio=0.15, beta=9.0):
    """transforms.py:462-489: windowed-sinc prototype (taps + 1 coefficients), kaiser(beta)."""
    assert taps % 2 == 0 and 0.0 < cutoff_ratio < 1.0
    n = np.arange(taps + 1) - 0.5 * taps
    with np.errstate(invalid='ignore', divide='ignore'):
        h = np.sin(np.pi * cutoff_ratio * n) / (np.pi * n)
    h[taps // 2] = cutoff_ratio
    return h * np.kaiser(taps + 1, beta)


def pqmf_filters(subbands=4, taps=62, cutoff_ratio=0.15, beta=9.0):
    """transforms.py:509-522: cosine-modulated analysis / synthesis banks, (subbands, taps + 1) each."""
    h = design_prototype_filter(taps, cutoff_ratio, beta)
    n = np.arange(taps + 1) - (taps - 1) / 2
    ha, hs = np.zeros((subbands, taps + 1)), np.zeros((subbands, taps + 1))
    for k in range(subbands):
        ph = (2 * k + 1) * (np.pi / (2 * subbands)) * n
        ha[k] = 2 * h * np.cos(ph + (-1) ** k * np.pi / 4)
        hs[k] = 2 * h * np.cos(ph - (-1) ** k * np.pi / 4)
    return ha, hs


def pqmf_analysis(x, ha):
    """transforms.py:534-542: (B, 1, T) -> (B, subbands, T // subbands)."""
    x = np.asarray(x, np.float64)
    B, _, T = x.shape
    S, nt = ha.shape
    P = (nt - 1) // 2
    xp = np.pad(x[:, 0], ((0, 0), (P, P)))
    M = T // S
    out = np.zeros((B, S, M))
    for j in range(nt):
        out += ha[None, :, j, None] * xp[:, None, j:j + M * S:S][:, :, :M]
    return out


def pqmf_synthesis(x, hs):
    """transforms.py:544-553: (B, subbands, M) -> (B, 1, M * subbands)."""
    x = np.asarray(x, np.float64)
    B, S, M = x.shape
    nt = hs.shape[1]
    P = (nt - 1) // 2
    up = np.zeros((B, S, M * S + 2 * P))
    up[:, :, P:P + M * S:S] = S * x
    y = np.zeros((B, M * S))
    for j in range(nt):
        y += (hs[None, :, j, None] * up[:, :, j:j + M * S]).sum(1)
    return y[:, None, :]


def mel_filterbank_htk(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """torchaudio.functional.melscale_fbanks(n_fft // 2 + 1, fmin, fmax, n_mels, sr, norm=None, mel_scale='htk') transposed to
    (n_mels, n_freqs): triangles of unit peak between HTK-mel-equidistant corner frequencies (float64).  Values unpinned by the
    reference (torchaudio absent): known answers in tests/test_filters_golden.py."""
    fmax = float(sr // 2) if fmax is None else float(fmax)
    freqs = np.linspace(0.0, sr // 2, n_fft // 2 + 1)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)          # noqa: E731
    pts = 700.0 * (10.0 ** (np.linspace(mel(float(fmin)), mel(fmax), n_mels + 2) / 2595.0) - 1.0)
    fb = np.zeros((n_mels, freqs.size))
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        fb[m] = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)))
    return fb
