"""Signal transforms with the constructor signatures, buffer names and return conventions of
pytorch_sound/models/transforms.py, computed by the gfx950 kernels of libpsnd_hip.so.

Reference            here                                   kernel(s)
STFT.transform       STFT.transform -> (mag, phase)         psnd_stft_fwd / psnd_stft_bwd
STFT.inverse         STFT.inverse                           psnd_istft
LogMelSpectrogram    LogMelSpectrogram.forward              psnd_stft_fwd + psnd_mel_fwd (+ bwd)
STFTTorchAudio       forward -> (re, im); transform         psnd_stft_fwd(re,im) / psnd_stft_bwd
Audio2Mel            forward (N,1,T) -> log10 mel           psnd_stft_fwd(HIFIGAN) + psnd_mel_fwd
"""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.utils.mel import mel_filterbank


def periodic_window(name: str, win_length: int) -> np.ndarray:
    """scipy.signal.get_window(name, win_length, fftbins=True) (transforms.py:30) in float64."""
    if name == 'hann':
        m = np.arange(win_length, dtype=np.float64)
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * m / win_length)
    from scipy.signal import get_window
    return get_window(name, win_length, fftbins=True)


def centre_pad(w: np.ndarray, size: int) -> np.ndarray:
    """librosa.util.pad_center as used at transforms.py:31 (left pad (size - len)//2)."""
    out = np.zeros(size, dtype=w.dtype)
    lpad = (size - w.shape[-1]) // 2
    out[lpad:lpad + w.shape[-1]] = w
    return out


class _PlanCache:
    """device-resident read-only tables (psnd_*_plan_build) keyed by what they were built from."""

    def __init__(self):
        self._plans = {}

    def get(self, key, device, builder):
        k = (key, str(device))
        plan = self._plans.get(k)
        if plan is None:
            if len(self._plans) > 8:
                self._plans.clear()
            plan = builder().to(device)
            self._plans[k] = plan
        return plan


def _as_2d(wav: torch.Tensor) -> torch.Tensor:
    if wav.dim() != 2:
        raise RuntimeError('expected a (N, T) waveform batch, got shape %s' % (tuple(wav.shape),))
    return wav


class STFT(nn.Module):
    """Drop-in for transforms.py:13-101.  ``filter_length`` is the FFT size; a shorter ``win_length``
    is zero-centre-padded.  ``square_window`` is a real buffer; ``forward_basis`` / ``inverse_basis``
    (the reference's 2 x 4.2 MB dense matrices at n=1024) are not needed by the kernels: they are
    accepted and ignored on ``load_state_dict`` and re-created in closed form by ``state_dict`` when
    ``emit_reference_buffers`` is set, so checkpoints travel both ways."""

    emit_reference_buffers = False

    def __init__(self, filter_length: int = 1024, hop_length: int = 512, win_length: int = None,
                 window: str = 'hann'):
        super().__init__()
        self.filter_length = filter_length
        self.hop_length = hop_length
        self.win_length = win_length if win_length else filter_length
        self.window = window
        self.pad_amount = self.filter_length // 2
        assert filter_length >= self.win_length
        w = centre_pad(periodic_window(window, self.win_length), filter_length).astype(np.float32)
        self._window_np = w
        self.register_buffer('square_window', torch.from_numpy(w) ** 2)
        self._plans = _PlanCache()
        self._register_state_dict_hook(STFT._emit_reference_keys)

    # ---- kernels -------------------------------------------------------------------------------
    def _plan(self, device):
        return self._plans.get('stft', device, lambda: K.stft_plan(self.filter_length, self._window_np))

    def transform(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(N,T) -> magnitude, phase, each (N, filter_length//2+1, T//hop+1).  phase carries no
        gradient (the reference takes atan2 of ``.data``, transforms.py:69)."""
        wav = _as_2d(wav)
        mag, phase = K.StftMagPhase.apply(wav, self._plan(wav.device), self.filter_length, self.hop_length,
                                          K.FRAMING_CENTER, 0.0, True)
        return mag, phase

    def magnitude(self, wav: torch.Tensor) -> torch.Tensor:
        """transform()[0] without computing the phase nobody asked for."""
        wav = _as_2d(wav)
        return K.StftMagPhase.apply(wav, self._plan(wav.device), self.filter_length, self.hop_length,
                                    K.FRAMING_CENTER, 0.0, False)[0]

    def inverse(self, magnitude: torch.Tensor, phase: torch.Tensor, eps: float = 1e-9) -> torch.Tensor:
        # square_window = window ** 2 is the reference's buffer; the window itself comes from the plan's source
        win = torch.from_numpy(self._window_np).to(magnitude.device)
        return K.istft(magnitude, phase, self.filter_length, self.hop_length, self._plan(magnitude.device), eps, win)

    def forward(self, wav: torch.Tensor) -> torch.Tensor:  # the reference defines no forward
        raise NotImplementedError

    # ---- checkpoint compatibility ----------------------------------------------------------------
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for name in ('forward_basis', 'inverse_basis'):
            state_dict.pop(prefix + name, None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @staticmethod
    def _emit_reference_keys(module, state_dict, prefix, local_metadata):
        if not module.emit_reference_buffers:
            return
        fb, ib = reference_bases(module.filter_length, module.hop_length, module._window_np)
        state_dict[prefix + 'forward_basis'] = torch.from_numpy(fb)
        state_dict[prefix + 'inverse_basis'] = torch.from_numpy(ib)


def reference_bases(n: int, hop: int, window: np.ndarray):
    """The reference's ``forward_basis`` (2K,1,n) and ``inverse_basis`` (2K,1,n) buffers
    (transforms.py:35-51) in closed form: the rows of the stacked [cos; -sin] matrix are mutually
    orthogonal, so its pseudo-inverse is the transpose with rows scaled by 1/||row||^2 (rows that
    are identically zero - the sine rows of bins 0 and n/2 - stay zero)."""
    k = np.arange(n // 2 + 1)[:, None]
    m = np.arange(n)[None, :]
    ang = 2.0 * np.pi * k * m / n
    basis = np.vstack([np.cos(ang), -np.sin(ang)])
    norm2 = (basis * basis).sum(axis=1, keepdims=True)
    pinv_t = np.divide(basis, norm2, out=np.zeros_like(basis), where=norm2 > 1e-9) * (hop / n)
    w = window.astype(np.float32)
    fwd = basis.astype(np.float32)[:, None, :] * w
    inv = pinv_t.astype(np.float32)[:, None, :] * w
    return fwd, inv


class LogMelSpectrogram(nn.Module):
    """Drop-in for transforms.py:206-244: natural-log mel of the conv-STFT magnitude, clamped to
    [ln 10^(min_db/10), ln 10^(max_db/10)].  Quirks kept on purpose: the inner STFT uses
    ``filter_length=win_length`` (``n_fft`` only sizes the mel filter, :217-220) and the clamps are
    gated by truthiness, so ``min_db=0`` / ``max_db=0`` disable them (:222-229, :240-243)."""

    def __init__(self, sample_rate: int, mel_size: int, n_fft: int, win_length: int, hop_length: int,
                 min_db: float = None, max_db: float = None, mel_min: float = 0., mel_max: float = None):
        super().__init__()
        self.mel_size = mel_size
        self.stft = STFT(filter_length=win_length, hop_length=hop_length)
        self.register_buffer('mel_filter', torch.from_numpy(
            mel_filterbank(sample_rate, n_fft, mel_size, fmin=mel_min, fmax=mel_max)))
        self.min_db = np.log(np.power(10, min_db / 10)) if min_db else None
        self.max_db = np.log(np.power(10, max_db / 10)) if max_db else None
        self._plans = _PlanCache()

    def _mel_plan(self):
        mf = self.mel_filter
        key = ('mel', mf._version, mf.data_ptr(), tuple(mf.shape))
        return self._plans.get(key, mf.device, lambda: K.mel_plan(mf.detach().cpu().numpy()))

    def forward(self, wav: torch.Tensor, log_offset: float = 1e-6) -> torch.Tensor:
        st = self.stft
        if (st.filter_length // 2 + 1 == self.mel_filter.shape[1]
                and K.logmel_fused_ok(wav, st.filter_length, st.hop_length)):
            w2 = _as_2d(wav)                           # one kernel, the magnitude never leaves the chip
            return K.logmel_forward(w2, st.filter_length, st.hop_length, st._plan(w2.device), self._mel_plan(),
                                    self.mel_size, K.FRAMING_CENTER, 0.0, K.LOG_E, float(log_offset), None,
                                    self.min_db if self.min_db else None, self.max_db if self.max_db else None)
        mag = self.stft.magnitude(wav)
        if mag.shape[1] != self.mel_filter.shape[1]:
            raise RuntimeError('mel_filter has %d bins but the STFT (filter_length=win_length) produced %d: '
                               'n_fft must equal win_length, as in the reference'
                               % (self.mel_filter.shape[1], mag.shape[1]))
        return K.MelLog.apply(mag, self._mel_plan(), self.mel_size, K.LOG_E, float(log_offset), None,
                              self.min_db if self.min_db else None, self.max_db if self.max_db else None)


class STFTTorchAudio(nn.Module):
    """Drop-in for transforms.py:271-319 (the torch.stft wrapper that models/sound.py uses for losses).
    forward -> (real, imag), transform -> (magnitude, phase); here the phase IS differentiable
    (transforms.py:311), so it is formed from the kernel's (re, im) with autograd-visible ops."""

    def __init__(self, filter_length: int = 1024, hop_length: int = 512, win_length: int = None, n_fft: int = None,
                 window: str = 'hann'):
        super().__init__()
        self.filter_length = filter_length
        self.hop_length = hop_length
        self.win_length = win_length if win_length else filter_length
        if window != 'hann':
            raise NotImplementedError('{} is not implemented ! Use hann'.format(window))
        self.register_buffer('window', torch.hann_window(self.win_length))
        self.n_fft = n_fft if n_fft else self.win_length
        self._plans = _PlanCache()

    def _plan(self, device):
        w = self.window
        key = ('stft', w._version, w.data_ptr())
        return self._plans.get(key, device, lambda: K.stft_plan(
            self.n_fft, centre_pad(w.detach().cpu().numpy().astype(np.float32), self.n_fft)))

    def forward(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        wav = _as_2d(wav)
        return K.StftReIm.apply(wav, self._plan(wav.device), self.n_fft, self.hop_length, K.FRAMING_CENTER)

    def transform(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        re, im = self.forward(wav)
        return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)

    def inverse(self, magnitude: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
        spec = torch.complex(magnitude * torch.cos(phase), magnitude * torch.sin(phase))
        return torch.istft(spec, self.n_fft, self.hop_length, self.win_length, self.window)


class _HifiGanMel(nn.Module):
    """shared body of the two (n-h)/2-padded, center=False mel front ends."""

    def _setup(self, n_fft, hop, win_length, mel_np, window_buf, filter_buf):
        self.n_fft, self.hop_length, self.win_length = n_fft, hop, win_length
        self.register_buffer(filter_buf, torch.from_numpy(mel_np).float())
        self.register_buffer(window_buf, torch.hann_window(win_length).float())
        self._filter_buf, self._window_buf = filter_buf, window_buf
        self._plans = _PlanCache()

    def _stft_plan(self, device):
        w = getattr(self, self._window_buf)
        key = ('stft', w._version, w.data_ptr())
        return self._plans.get(key, device, lambda: K.stft_plan(
            self.n_fft, centre_pad(w.detach().cpu().numpy().astype(np.float32), self.n_fft)))

    def _mel_plan(self):
        mf = getattr(self, self._filter_buf)
        key = ('mel', mf._version, mf.data_ptr(), tuple(mf.shape))
        return self._plans.get(key, mf.device, lambda: K.mel_plan(mf.detach().cpu().numpy()))

    def _logmel(self, wav2d, framing, mag_eps, log_kind):
        if K.logmel_fused_ok(wav2d, self.n_fft, self.hop_length):
            M = getattr(self, self._filter_buf).shape[0]
            return K.logmel_forward(wav2d, self.n_fft, self.hop_length, self._stft_plan(wav2d.device), self._mel_plan(), M,
                                    framing, mag_eps, log_kind, 0.0, 1e-5, None, None)
        mag = K.StftMagPhase.apply(wav2d, self._stft_plan(wav2d.device), self.n_fft, self.hop_length,
                                   framing, mag_eps, False)[0]
        M = getattr(self, self._filter_buf).shape[0]
        return K.MelLog.apply(mag, self._mel_plan(), M, log_kind, 0.0, 1e-5, None, None)


class Audio2Mel(_HifiGanMel):
    """Drop-in for transforms.py:322-366 (MelGAN front end): input (N,1,T), reflect-pad (n-h)/2,
    center=False, log10(clamp(mel, 1e-5)); ``mel_fmax=None`` means sr/2."""

    def __init__(self, n_fft: int = 1024, hop_length: int = 256, win_length: int = 1024, sampling_rate: int = 22050,
                 n_mel_channels: int = 80, mel_fmin: float = 0.0, mel_fmax: Optional[float] = None):
        super().__init__()
        self._setup(n_fft, hop_length, win_length,
                    mel_filterbank(sampling_rate, n_fft, n_mel_channels, mel_fmin, mel_fmax), 'window', 'mel_basis')
        self.sampling_rate = sampling_rate
        self.n_mel_channels = n_mel_channels

    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        if audio.dim() != 3 or audio.shape[1] != 1:
            raise RuntimeError('Audio2Mel expects (N, 1, T), got %s' % (tuple(audio.shape),))
        return self._logmel(audio.squeeze(1), K.FRAMING_HIFIGAN, 0.0, K.LOG_10)
