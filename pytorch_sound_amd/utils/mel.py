"""Slaney-scale, Slaney-area-normalised triangular mel filterbank.

The reference obtains this matrix from a third-party call, ``librosa.filters.mel(sr, n_fft, n_mels,
fmin, fmax)`` of librosa==0.8.0 (pytorch_sound/models/transforms.py:220, :339-341,
interface/hifi_gan.py:42), which is not part of the reference tree and not installed here.  This is
the product's own (vectorised) implementation of that published algorithm with the 0.8.0 defaults
htk=False, norm='slaney', dtype=float32; the filter VALUES are "parity unpinned" (DESIGN.md).
"""
import numpy as np

_F_SP = 200.0 / 3.0            # Hz per mel in the linear region
_BREAK_HZ = 1000.0
_BREAK_MEL = _BREAK_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(hz):
    hz = np.atleast_1d(np.asarray(hz, dtype=np.float64))
    lin = hz / _F_SP
    with np.errstate(divide='ignore', invalid='ignore'):
        log = _BREAK_MEL + np.log(np.maximum(hz, 1e-300) / _BREAK_HZ) / _LOGSTEP
    return np.where(hz >= _BREAK_HZ, log, lin)


def mel_to_hz(mel):
    mel = np.atleast_1d(np.asarray(mel, dtype=np.float64))
    return np.where(mel >= _BREAK_MEL, _BREAK_HZ * np.exp(_LOGSTEP * (mel - _BREAK_MEL)), _F_SP * mel)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None) -> np.ndarray:
    """(n_mels, n_fft//2 + 1) float32."""
    fmax = float(sr) / 2 if fmax is None else fmax
    n_mels = int(n_mels)
    n_bins = 1 + n_fft // 2
    bin_hz = np.linspace(0.0, float(sr) / 2, n_bins)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin)[0], hz_to_mel(fmax)[0], n_mels + 2))   # band edges in Hz
    width = np.diff(edges)
    rise = (bin_hz[None, :] - edges[:-2, None]) / width[:-1, None]      # 0 at left edge, 1 at centre
    fall = (edges[2:, None] - bin_hz[None, :]) / width[1:, None]        # 1 at centre, 0 at right edge
    tri = np.maximum(0.0, np.minimum(rise, fall)).astype(np.float32)    # librosa stores float32 here
    area_norm = 2.0 / (edges[2:] - edges[:-2])
    return (tri.astype(np.float64) * area_norm[:, None]).astype(np.float32)


def mel_filterbank_htk(sr, n_fft, n_mels, fmin=0.0, fmax=None) -> np.ndarray:
    """(n_mels, n_fft // 2 + 1) float32 triangular filters on the HTK mel scale, peak 1, no area normalisation - what
    torchaudio.transforms.MelSpectrogram builds by default (melscale_fbanks(norm=None, mel_scale='htk')), used by the
    reference's LogMelSpectrogramTorchAudio (transforms.py:369-394)."""
    fmax = float(sr // 2) if fmax is None else float(fmax)
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0.0, sr // 2, n_freqs)
    to_mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)       # noqa: E731
    to_hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)      # noqa: E731
    f_pts = to_hz(np.linspace(to_mel(float(fmin)), to_mel(fmax), n_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]                  # (n_freqs, n_mels + 2)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).T.astype(np.float32)
