"""Host formulation of the STFT family for CPU tensors.

The reference's transform modules are device-agnostic (`InterfaceHifiGAN(device='cpu')` is its default,
interface/hifi_gan.py:92; BASELINE configs[0] is "feature extraction only, batch=4 on CPU").  A tensor that lives on the
host therefore takes the plain torch formulation below - the same arithmetic the reference states (transforms.py:53-101,
231-244, 297-319, 351-366; interface/hifi_gan.py:46-63), differentiable through torch's own autograd.

This is NOT a fallback of the GPU path: a HIP tensor never reaches this module (the modules dispatch on `tensor.is_cuda`, and
`pytorch_sound_amd.kernels` raises when libpsnd_hip.so is missing or a geometry is not covered), and nothing here is used by
bench.py's timed region.  It does not import `oracle/` (test infrastructure).
"""
import math

import torch
import torch.nn.functional as F

FRAMING_CENTER, FRAMING_HIFIGAN, FRAMING_NONE = 0, 1, 2
LOG_NONE, LOG_E, LOG_10 = 0, 1, 2


def _check_host(t, name):
    if t.is_cuda:
        raise RuntimeError('%s is a HIP tensor: the host formulation only takes CPU tensors (the GPU path is pytorch_sound_amd.kernels)' % name)


def frames(wav, n_fft, hop, framing):
    """(N, T) -> reflect-padded (N, T + 2 pad) ready for center=False framing.  pad = n/2 (transforms.py:55-60, torch.stft
    center=True), (n - hop)/2 (transforms.py:352-353, interface/hifi_gan.py:48-49) or none."""
    pad = {FRAMING_CENTER: n_fft // 2, FRAMING_HIFIGAN: (n_fft - hop) // 2, FRAMING_NONE: 0}[framing]
    if pad == 0:
        return wav
    return F.pad(wav.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1)


def stft_complex(wav, n_fft, hop, window, framing=FRAMING_CENTER):
    """one-sided DFT of the windowed frames, (N, K, F) complex64.  `window` has n_fft taps (already centre-padded)."""
    _check_host(wav, 'wav')
    if wav.dim() != 2:
        raise RuntimeError('expected a (N, T) waveform batch, got shape %s' % (tuple(wav.shape),))
    x = frames(wav, n_fft, hop, framing)
    return torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=window.to(x.dtype), center=False, normalized=False,
                      onesided=True, return_complex=True)


def stft_mag_phase(wav, n_fft, hop, window, framing=FRAMING_CENTER, mag_eps=0.0, want_phase=True, detach_phase=True):
    """STFT.transform (transforms.py:53-69): sqrt(re^2 + im^2 [+ eps]) and atan2(im, re) (of `.data` when detach_phase)."""
    c = stft_complex(wav, n_fft, hop, window, framing)
    re, im = c.real, c.imag
    mag = torch.sqrt(re ** 2 + im ** 2 + mag_eps) if mag_eps else torch.sqrt(re ** 2 + im ** 2)
    if not want_phase:
        return mag, None
    phase = torch.atan2(im.detach(), re.detach()) if detach_phase else torch.atan2(im, re)
    return mag, phase


def istft(magnitude, phase, n_fft, hop, window, eps=1e-9):
    """STFT.inverse (transforms.py:71-101).  The reference's `inverse_basis` is pinv((n/h) [cos; -sin]) . window: the rows of the
    stacked matrix are orthogonal, so one frame of its conv_transpose1d is (h/n) window irfft(mag e^{i phase}) (imaginary parts of
    DC / Nyquist do not reach the signal - irfft drops them too).  Overlap-add, divide by the squared-window envelope + eps, scale by
    n/h, trim n/2 on both sides: (F - 1) hop samples.  eps = 0 is torch.istft's convention (transforms.py:313-319)."""
    _check_host(magnitude, 'magnitude')
    N, K, Fr = magnitude.shape
    if K != n_fft // 2 + 1 or phase.shape != magnitude.shape:
        raise RuntimeError('istft: expected (N, %d, F) magnitude and phase, got %s / %s'
                           % (n_fft // 2 + 1, tuple(magnitude.shape), tuple(phase.shape)))
    w = window.to(magnitude.dtype)
    fr = torch.fft.irfft(torch.polar(magnitude, phase), n=n_fft, dim=1) * w.view(1, -1, 1)       # (N, n, F)
    L = n_fft + hop * (Fr - 1)
    ola = F.fold(fr, (1, L), (1, n_fft), stride=(1, hop)).reshape(N, L)
    env = F.fold((w * w).view(1, -1, 1).expand(1, n_fft, Fr).contiguous(), (1, L), (1, n_fft), stride=(1, hop)).reshape(L)
    p = n_fft // 2
    den = env + eps
    if eps == 0:                        # the envelope vanishes only inside the trimmed edges
        den = torch.where(den > 0, den, torch.ones_like(den))
    return (ola / den)[:, p:L - p]


def mel_log(mag, mel_filter, log_kind=LOG_E, log_offset=0.0, pre_clamp_min=None, clamp_lo=None, clamp_hi=None):
    """clamp(log(max(W @ mag, pre) + off), lo, hi) - transforms.py:235-243 / :364-365 / interface/hifi_gan.py:58-61."""
    mel = torch.matmul(mel_filter.to(mag.dtype), mag)
    if pre_clamp_min is not None:
        mel = torch.clamp(mel, min=pre_clamp_min)
    if log_kind == LOG_E:
        mel = torch.log(mel + log_offset) if log_offset else torch.log(mel)
    elif log_kind == LOG_10:
        mel = torch.log10(mel + log_offset) if log_offset else torch.log10(mel)
    if clamp_lo is not None:
        mel = mel.clamp_min(clamp_lo)
    if clamp_hi is not None:
        mel = mel.clamp_max(clamp_hi)
    return mel


def db_to_ln(db):
    return math.log(math.pow(10.0, db / 10.0))
