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

import math
import numpy as np
import torch
import torch.nn as nn

from pytorch_sound_amd import dense as D
from pytorch_sound_amd import host as H
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
    accepted and ignored on ``load_state_dict`` and re-created in closed form by ``state_dict``, so checkpoints
    travel both ways (a checkpoint saved here passes a strict load in the reference).  ``emit_reference_buffers``:
    None (default) = emit for filter_length <= 1024 (2 x 4.2 MB, what the reference itself saves), skip above
    (2 x 67 MB at 4096: set True to get them); True / False force it."""

    emit_reference_buffers = None

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
        if not wav.is_cuda:                                  # host tensors: the reference's formulation in torch ops (host.py)
            return H.stft_mag_phase(wav, self.filter_length, self.hop_length, self._host_window(), H.FRAMING_CENTER)
        if not D.is_fast_size(self.filter_length):           # e.g. 800 / 1200 / 2400: the dense-basis GEMM on the matrix cores (dense.py)
            return D.stft_mag_phase(wav, self.filter_length, self.hop_length, self._window_np, self.pad_amount)
        from pytorch_sound_amd import deferred
        if (deferred.ENABLED and self._lazy_transform and wav.dtype == torch.float32
                and not (torch.is_grad_enabled() and wav.requires_grad)):
            # no gradient wanted: the magnitude from the bin-fastest kernel (psnd_stft_mag_nfk: 0.62 of the HBM roofline against 0.48 for
            # the frame-fastest store) as a deferred tensor that stands for the reference's (N, K, F) - consumers of this library take it as
            # it is, any other use transposes it once; the phase is computed when something uses it (deferred.py).  Adaptive: the first
            # time a magnitude of this module is transposed or a phase is used, `_lazy_transform` goes off and the module computes both
            # at once in the reference's layout from then on (set it back to True to probe again).
            nfk = K.stft_mag_nfk(wav, self.filter_length, self.hop_length, self._plan(wav.device), K.FRAMING_CENTER, 0.0)
            mag = deferred.mag_nfk(nfk, self)
            return mag, deferred.Deferred(deferred.Phase(wav, self, mag.shape))
        return self._transform_now(wav)

    _lazy_transform = True

    def _transform_now(self, wav):
        return K.StftMagPhase.apply(wav, self._plan(wav.device), self.filter_length, self.hop_length, K.FRAMING_CENTER, 0.0, True)

    def magnitude(self, wav: torch.Tensor) -> torch.Tensor:
        """transform()[0] without computing the phase nobody asked for."""
        wav = _as_2d(wav)
        if not wav.is_cuda:
            return H.stft_mag_phase(wav, self.filter_length, self.hop_length, self._host_window(), H.FRAMING_CENTER,
                                    want_phase=False)[0]
        if not D.is_fast_size(self.filter_length):
            return D.stft_mag_phase(wav, self.filter_length, self.hop_length, self._window_np, self.pad_amount, want_phase=False)[0]
        return K.StftMagPhase.apply(wav, self._plan(wav.device), self.filter_length, self.hop_length,
                                    K.FRAMING_CENTER, 0.0, False)[0]

    def _host_window(self):
        return torch.from_numpy(self._window_np)

    def inverse(self, magnitude: torch.Tensor, phase: torch.Tensor, eps: float = 1e-9) -> torch.Tensor:
        if not magnitude.is_cuda:
            return H.istft(magnitude, phase, self.filter_length, self.hop_length, self._host_window(), eps)
        if not D.is_fast_size(self.filter_length):
            return D.istft(magnitude, phase, self.filter_length, self.hop_length, self._window_np, eps)
        # square_window = window ** 2 is the reference's buffer; the window itself comes from the plan's source (cached per device:
        # the overlap-add envelope of IStft.backward is keyed on this tensor)
        win = self._plans.get('win', magnitude.device, lambda: torch.from_numpy(self._window_np))
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
        emit = module.emit_reference_buffers
        if emit is None:
            emit = module.filter_length <= 1024
        if not emit:
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
        from pytorch_sound_amd import deferred
        deferred.MEL_MODULES.add(self)        # matmul(self.mel_filter, <deferred estimate>) is recognised (deferred.py)
        self.min_db = np.log(np.power(10, min_db / 10)) if min_db else None
        self.max_db = np.log(np.power(10, max_db / 10)) if max_db else None
        self._plans = _PlanCache()

    def _mel_plan(self):
        mf = self.mel_filter
        key = ('mel', mf._version, mf.data_ptr(), tuple(mf.shape))
        return self._plans.get(key, mf.device, lambda: K.mel_plan(mf.detach().cpu().numpy()))

    def forward(self, wav: torch.Tensor, log_offset: float = 1e-6) -> torch.Tensor:
        st = self.stft
        if not wav.is_cuda:
            mag = st.magnitude(wav)
            return H.mel_log(mag, self.mel_filter, H.LOG_E, float(log_offset), None,
                             self.min_db if self.min_db else None, self.max_db if self.max_db else None)
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
    forward -> (real, imag), transform -> (magnitude, phase), inverse; the phase IS differentiable here (transforms.py:311)."""

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
        if not wav.is_cuda:
            c = H.stft_complex(wav, self.n_fft, self.hop_length, self._host_window(), H.FRAMING_CENTER)
            return c.real, c.imag
        return K.StftReIm.apply(wav, self._plan(wav.device), self.n_fft, self.hop_length, K.FRAMING_CENTER)

    def _host_window(self):
        w = self.window.detach().cpu().float()
        lp = (self.n_fft - w.numel()) // 2
        return torch.nn.functional.pad(w, (lp, self.n_fft - w.numel() - lp))

    def transform(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """magnitude and (differentiable) phase out of one launch; the phase gradient goes through psnd_polar_bwd."""
        wav = _as_2d(wav)
        if not wav.is_cuda:
            return H.stft_mag_phase(wav, self.n_fft, self.hop_length, self._host_window(), H.FRAMING_CENTER, detach_phase=False)
        return K.StftPolar.apply(wav, self._plan(wav.device), self.n_fft, self.hop_length, K.FRAMING_CENTER)

    def inverse(self, magnitude: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
        """torch.istft(mag e^{i phase}, n_fft, hop, win_length, window) (transforms.py:313-319): overlap-add of the windowed inverse
        frames divided by the squared-window envelope, n_fft/2 trimmed on both sides, (F-1)*hop samples - psnd_istft with eps = 0
        (STFT.inverse adds its 1e-9 to the envelope, torch.istft divides plainly)."""
        if not magnitude.is_cuda:
            return H.istft(magnitude, phase, self.n_fft, self.hop_length, self._host_window(), 0.0)
        w = self.window
        win = self._plans.get(('win', w._version, w.data_ptr()), magnitude.device, lambda: torch.from_numpy(
            centre_pad(w.detach().cpu().numpy().astype(np.float32), self.n_fft)))
        return K.istft(magnitude, phase, self.n_fft, self.hop_length, self._plan(magnitude.device), 0.0, win)


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
        if not wav2d.is_cuda:
            w = getattr(self, self._window_buf).detach().cpu().float()
            lp = (self.n_fft - w.numel()) // 2
            w = torch.nn.functional.pad(w, (lp, self.n_fft - w.numel() - lp))
            mag = H.stft_mag_phase(_as_2d(wav2d), self.n_fft, self.hop_length, w, framing, mag_eps, want_phase=False)[0]
            return H.mel_log(mag, getattr(self, self._filter_buf), log_kind, 0.0, 1e-5, None, None)
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


#
# MFCC (transforms.py:419-459)
#
def create_dct(n_mfcc: int, n_mels: int, norm: str = 'ortho') -> torch.Tensor:
    """DCT-II basis (n_mels, n_mfcc) as torchaudio.functional.create_dct builds it (norm None: x2, 'ortho': orthonormal)."""
    n = torch.arange(float(n_mels), dtype=torch.float64)
    k = torch.arange(float(n_mfcc), dtype=torch.float64).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    if norm is None:
        dct *= 2.0
    else:
        assert norm == 'ortho'
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().float()


class MelToMFCC(nn.Module):
    """Mel-frequency cepstrum coefficients from a (log-)mel spectrogram (transforms.py:419-432): dct_mat @ mel_spec.  On a
    HIP device the product runs on the fp32 matrix-core kernel of the mel projection (psnd_mel_fwd / psnd_mel_bwd, no log)."""

    def __init__(self, n_mfcc: int, mel_size: int, norm: str = 'ortho'):
        super().__init__()
        self.n_mfcc = n_mfcc
        self.register_buffer('dct_mat', create_dct(n_mfcc, mel_size, norm).transpose(0, 1).contiguous())
        self._plans = _PlanCache()

    def forward(self, mel_spec: torch.Tensor) -> torch.Tensor:
        assert len(mel_spec.size()) == 3
        if mel_spec.is_cuda:
            d = self.dct_mat
            plan = self._plans.get(('dct', d._version, d.data_ptr()), mel_spec.device,
                                   lambda: K.mel_plan(d.detach().float().cpu().numpy()))
            y = K.MelLog.apply(mel_spec.float().contiguous(), plan, self.n_mfcc, K.LOG_NONE, 0.0, None, None, None)
            return y if y.dtype == mel_spec.dtype else y.to(mel_spec.dtype)
        return torch.matmul(self.dct_mat, mel_spec)


class MFCC(nn.Module):
    """MFCC of a waveform (transforms.py:435-459): LogMelSpectrogram, then the DCT.  The reference asserts a 3-D input
    and hands it to a front end that only takes (N, T); here (N, 1, T) is accepted and squeezed."""

    def __init__(self, sample_rate: int, mel_size: int, n_fft: int, win_length: int, n_mfcc: int, hop_length: int,
                 min_db: float, max_db: float, mel_min: float = 0., mel_max: float = None, norm: str = 'ortho'):
        super().__init__()
        self.n_mfcc = n_mfcc
        self.mel_func = LogMelSpectrogram(sample_rate, mel_size, n_fft, win_length, hop_length, min_db, max_db, mel_min, mel_max)
        self.to_mfcc = MelToMFCC(n_mfcc, mel_size, norm)

    @property
    def dct_mat(self):
        return self.to_mfcc.dct_mat

    def forward(self, wav: torch.Tensor) -> torch.Tensor:
        assert len(wav.size()) == 3
        return self.to_mfcc(self.mel_func(wav.squeeze(1)))


class SpectrogramMasker(nn.Module):
    """wave-level mask -> frame-level mask (transforms.py:397-416): a frame is valid when any of its samples is - the mean of
    the window (a conv with constant 1 / win_length), rounded up.  The reference builds its conv on the GPU at
    construction; here it follows the input's device."""

    def __init__(self, win_length: int, hop_length: int):
        super().__init__()
        self.win_length = win_length
        self.conv = nn.Conv1d(1, 1, self.win_length, stride=hop_length, padding=0, bias=False)
        torch.nn.init.constant_(self.conv.weight, 1. / self.win_length)

    def forward(self, wav_mask: torch.Tensor) -> torch.Tensor:
        if wav_mask.is_cuda:
            # gfx950: psnd_frame_mask (span of 16 frames in LDS, window sums, exact division + ceil) - no library convolution for a
            # HIP tensor.  The conv weight stays a parameter (state_dict parity); it is the constant 1 / win_length by construction.
            from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check, PsndError
            if wav_mask.dim() != 2:
                raise RuntimeError('SpectrogramMasker: (N, T) wave-level mask expected, got %s' % (tuple(wav_mask.shape),))
            m = wav_mask.detach().float().contiguous()
            N, T = m.shape
            hop = int(self.conv.stride[0])
            F_ = int(lib().psnd_frame_mask_frames(T, self.win_length, hop))
            out = torch.empty((N, F_), dtype=torch.float32, device=m.device)
            with torch.cuda.device(m.device):
                check(lib().psnd_frame_mask(ptr(m), N, T, self.win_length, hop, ptr(out), stream_ptr(m.device)), 'psnd_frame_mask')
            return out
        with torch.no_grad():
            if self.conv.weight.device != wav_mask.device:
                self.conv.to(wav_mask.device)
            wav_mask = torch.nn.functional.pad(wav_mask, [0, self.win_length // 2], value=0.)
            wav_mask = torch.nn.functional.pad(wav_mask, [self.win_length // 2, 0], value=1.)
            mel_mask = self.conv(wav_mask.float().unsqueeze(1)).squeeze(1)
            mel_mask = torch.ceil(mel_mask)
        return mel_mask


#
# Pseudo-QMF bank (transforms.py:462-560)
#
def design_prototype_filter(taps=62, cutoff_ratio=0.15, beta=9.0):
    """Kaiser-windowed sinc prototype of the cosine-modulated bank, taps + 1 coefficients (transforms.py:462-489)."""
    assert taps % 2 == 0, "The number of taps mush be even number."
    assert 0.0 < cutoff_ratio < 1.0, "Cutoff ratio must be > 0.0 and < 1.0."
    n = np.arange(taps + 1) - 0.5 * taps
    with np.errstate(invalid='ignore', divide='ignore'):
        h = np.sin(np.pi * cutoff_ratio * n) / (np.pi * n)
    h[taps // 2] = cutoff_ratio                              # the 0/0 at the centre tap
    return h * np.kaiser(taps + 1, beta)


class PQMF(torch.nn.Module):
    """Near-perfect-reconstruction pseudo-QMF bank (transforms.py:492-560): ``analysis`` (B, 1, T) -> (B, subbands,
    T // subbands), ``synthesis`` back.  Buffers ``analysis_filter`` / ``synthesis_filter`` / ``updown_filter`` as in the
    reference.  On a HIP device both directions are polyphase kernels (psnd_pqmf_analysis / psnd_pqmf_synthesis: only the
    kept outputs are computed, no zero-stuffed intermediate) with autograd; CPU tensors take the reference's conv formulation."""

    def __init__(self, subbands=4, taps=62, cutoff_ratio=0.15, beta=9.0):
        super().__init__()
        h = design_prototype_filter(taps, cutoff_ratio, beta)
        n = np.arange(taps + 1) - ((taps - 1) / 2)
        bank = np.zeros((2, subbands, len(h)))
        for k in range(subbands):
            phase = (2 * k + 1) * (np.pi / (2 * subbands)) * n
            bank[0, k] = 2 * h * np.cos(phase + (-1) ** k * np.pi / 4)
            bank[1, k] = 2 * h * np.cos(phase - (-1) ** k * np.pi / 4)
        self.register_buffer('analysis_filter', torch.from_numpy(bank[0]).float().unsqueeze(1))
        self.register_buffer('synthesis_filter', torch.from_numpy(bank[1]).float().unsqueeze(0))
        updown = torch.zeros((subbands, subbands, subbands)).float()
        for k in range(subbands):
            updown[k, k, 0] = 1.0
        self.register_buffer('updown_filter', updown)
        self.subbands, self.taps = subbands, taps
        self.pad_fn = torch.nn.ConstantPad1d(taps // 2, 0.0)

    def _hip(self, x, channels):
        """a HIP tensor always takes psnd_pqmf_* (cast to fp32 if need be); a filter bank the kernel does not cover raises"""
        if not x.is_cuda:
            return False
        if self.subbands > 16 or self.taps > 255:
            raise K.PsndError('PQMF(subbands=%d, taps=%d): psnd_pqmf_* covers up to 16 bands and 255 taps; there is no library '
                              'path for a HIP tensor' % (self.subbands, self.taps))
        if x.dim() != 3 or x.size(1) != channels or not x.is_floating_point():
            raise RuntimeError('PQMF expects a floating-point (N, %d, T) tensor, got %s %s' % (channels, x.dtype, tuple(x.shape)))
        return True

    def analysis(self, x):
        if self._hip(x, 1):
            y = K.PqmfAnalysis.apply(x.squeeze(1).float(), self.analysis_filter.squeeze(1).float(), self.subbands, self.taps)
            return y if y.dtype == x.dtype else y.to(x.dtype)
        x = torch.nn.functional.conv1d(self.pad_fn(x), self.analysis_filter)
        return torch.nn.functional.conv1d(x, self.updown_filter, stride=self.subbands)

    def synthesis(self, x):
        if self._hip(x, self.subbands):
            y = K.PqmfSynthesis.apply(x.float(), self.synthesis_filter.squeeze(0).float(), self.subbands, self.taps).unsqueeze(1)
            return y if y.dtype == x.dtype else y.to(x.dtype)
        x = torch.nn.functional.conv_transpose1d(x, self.updown_filter * self.subbands, stride=self.subbands)
        return torch.nn.functional.conv1d(self.pad_fn(x), self.synthesis_filter)


class LearnableSTFT(nn.Module):
    """STFT whose analysis / synthesis bases are trainable (transforms.py:104-203, marked experimental there): the same dense
    [cos; -sin] bases as ``STFT`` but WITHOUT the window folded in (``fft_window`` is a buffer, multiplied in at every call), as
    ``nn.Parameter``s when ``trainable_*``.  With trained bases the transform is no longer an FFT, so it runs as the strided
    (transposed) convolution it is defined as - library kernels, not the FFT path."""

    def __init__(self, filter_length: int = 1024, hop_length: int = 512, win_length: int = None, window: str = 'hann',
                 trainable_inverse: bool = True, trainable_forward: bool = True):
        super().__init__()
        self.filter_length = filter_length
        self.hop_length = hop_length
        self.win_length = win_length if win_length else filter_length
        self.window = window
        self.pad_amount = self.filter_length // 2
        assert filter_length >= self.win_length
        w = centre_pad(periodic_window(window, self.win_length), filter_length)
        self.register_buffer('fft_window', torch.from_numpy(w).float())
        fwd, inv = reference_bases(filter_length, hop_length, np.ones(filter_length))          # bases without the window
        for name, basis, trainable in (('forward_basis', fwd, trainable_forward), ('inverse_basis', inv, trainable_inverse)):
            t = torch.from_numpy(np.ascontiguousarray(basis)).float()
            if trainable:
                setattr(self, name, nn.Parameter(t))
            else:
                self.register_buffer(name, t)

    def transform(self, wav: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        x = torch.nn.functional.pad(wav.unsqueeze(1), (self.pad_amount, self.pad_amount), mode='reflect')
        spec = torch.nn.functional.conv1d(x, self.forward_basis * self.fft_window, stride=self.hop_length)
        re, im = spec.chunk(2, 1)
        return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im.data, re.data)

    def inverse(self, magnitude: torch.Tensor, phase: torch.Tensor, eps: float = 1e-9) -> torch.Tensor:
        spec = torch.cat([magnitude * torch.cos(phase), magnitude * torch.sin(phase)], dim=1)
        y = torch.nn.functional.conv_transpose1d(spec, self.inverse_basis * self.fft_window, stride=self.hop_length)
        # squared-window overlap-add envelope, as STFT.inverse
        env = torch.nn.functional.conv_transpose1d(torch.ones(1, 1, spec.size(-1), dtype=y.dtype, device=y.device),
                                                   (self.fft_window ** 2).view(1, 1, -1), stride=self.hop_length)
        y = y / (env.squeeze() + eps) * (self.filter_length / self.hop_length)
        return y[..., self.pad_amount:-self.pad_amount].squeeze(1)


class LogMelSpectrogramTorchAudio(nn.Module):
    """Drop-in for transforms.py:369-394: torchaudio's MelSpectrogram (POWER spectrogram of torch.stft with a hann window of
    ``win_length`` centre-padded to ``n_fft``, HTK mel triangles of unit peak) -> ln(mel + log_offset) -> clamp to the dB range.
    torchaudio is not needed: the STFT is psnd_stft_fwd/bwd, the filterbank is restated (utils/mel.py:mel_filterbank_htk) and
    runs on the mel kernel."""

    def __init__(self, sample_rate: int, mel_size: int, n_fft: int, win_length: int, hop_length: int, min_db: float,
                 max_db: float, mel_min: float = 0., mel_max: float = None):
        super().__init__()
        from pytorch_sound_amd.utils.mel import mel_filterbank_htk
        self.mel_size = mel_size
        self.min_db = np.log(np.power(10, min_db / 10))
        self.max_db = np.log(np.power(10, max_db / 10))
        self.stft = STFTTorchAudio(win_length, hop_length, win_length, n_fft)
        self.register_buffer('mel_filter', torch.from_numpy(mel_filterbank_htk(sample_rate, n_fft, mel_size, mel_min, mel_max)))
        self._plans = _PlanCache()

    def forward(self, wav: torch.Tensor, log_offset: float = 1e-6) -> torch.Tensor:
        st = self.stft
        wav = _as_2d(wav)
        if not wav.is_cuda:
            mag = H.stft_mag_phase(wav, st.n_fft, st.hop_length, st._host_window(), H.FRAMING_CENTER, want_phase=False)[0]
            return H.mel_log(mag * mag, self.mel_filter, H.LOG_E, float(log_offset), None, float(self.min_db), float(self.max_db))
        mag = K.StftMagPhase.apply(wav, st._plan(wav.device), st.n_fft, st.hop_length, K.FRAMING_CENTER, 0.0, False)[0]
        mf = self.mel_filter
        plan = self._plans.get(('mel', mf._version, mf.data_ptr()), mf.device, lambda: K.mel_plan(mf.detach().cpu().numpy()))
        return K.MelLog.apply(mag * mag, plan, self.mel_size, K.LOG_E, float(log_offset), None, float(self.min_db), float(self.max_db))
