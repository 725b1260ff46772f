"""STFT for a `filter_length` that is not a power of two, on HIP tensors.

The reference's STFT (transforms.py:19-69) takes ANY filter_length: its transform is a strided conv1d with a dense (2K, n) DFT basis.  The
register-FFT kernels of this library (csrc/psnd_stft*.hip) cover powers of two; a HIP tensor with another size (800, 1200, 2400 ...) takes the
reference's own formulation on this library's exact-fp32 matrix-core GEMM (psnd_linear1x1_*: v_mfma_f32_32x32x2_f32, fp32 in / fp32
accumulate) - frames (reflect pad + unfold: data movement) x windowed [cos; -sin] basis, then sqrt / atan2 - and its autograd: no library
convolution / fft / bmm on a HIP tensor, and no silent fallback: without libpsnd_hip.so the GEMM raises.  Cost: 2 * 2K * n flop per frame
(what the reference pays at every size) instead of ~2.5 n log2 n - the power-of-two kernels stay the fast path.

The inverse (transforms.py:71-101) is the matching dense synthesis: frames = W_inv [mag cos(phase); mag sin(phase)] with the one-sided
inverse-DFT matrix, windowed, overlap-added (F.fold) and divided by the squared-window envelope + eps (host.istft states the conventions).
Any filter_length >= 2: K = int(n / 2 + 1) bins as transforms.py:34 cuts them (an odd size has no Nyquist bin).
Long batches are unfolded and multiplied in chunks of clips (CHUNK_BYTES of frames at a time)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

_CACHE = {}
CHUNK_BYTES = 1 << 30        # unfolded frames materialised at a time (the unfold is n / hop x the input)


def _bases(n, window_np, device):
    key = (n, window_np.tobytes(), str(device))
    got = _CACHE.get(key)
    if got is None:
        k = np.arange(n // 2 + 1)[:, None].astype(np.float64)
        m = np.arange(n)[None, :].astype(np.float64)
        ang = 2.0 * np.pi * k * m / n
        w = window_np.astype(np.float64)[None, :]
        fwd = np.vstack([np.cos(ang), -np.sin(ang)]) * w                                   # (2K, n): re = sum w x cos, im = -sum w x sin
        c = np.full((n // 2 + 1, 1), 2.0)
        c[0, 0] = 1.0
        if n % 2 == 0:
            c[-1, 0] = 1.0                                                                  # the Nyquist bin (an odd size has none: K = (n + 1) / 2)
        inv = np.hstack([(c * np.cos(ang)).T, (-c * np.sin(ang)).T]) / n                   # (n, 2K): x[m] = sum_k c_k (Re cos - Im sin) / n
        got = (torch.from_numpy(fwd.astype(np.float32)).to(device), torch.from_numpy(inv.astype(np.float32)).to(device),
               torch.from_numpy(window_np.astype(np.float32)).to(device))
        while len(_CACHE) >= 8:                                                            # least recently used first (dict order: see below)
            _CACHE.pop(next(iter(_CACHE)))
    else:
        _CACHE.pop(key)
    _CACHE[key] = got                                                                       # most recently used last
    return got


def _frames(wav, n, hop, pad):
    x = F.pad(wav.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1) if pad else wav
    return x.unfold(1, n, hop).permute(0, 2, 1).contiguous()                               # (N, n, F)


def stft_mag_phase(wav, n, hop, window_np, pad, mag_eps=0.0, want_phase=True, detach_phase=True):
    """(N, T) HIP tensor -> magnitude, phase (N, n / 2 + 1, F), F = (T + 2 pad - n) // hop + 1"""
    from . import kernels as K
    fwd, _, _ = _bases(n, window_np, wav.device)
    wav = wav.float()
    Fr = (wav.shape[1] + 2 * pad - n) // hop + 1
    per_clip = 4 * n * max(Fr, 1)                                                          # bytes of unfolded frames per clip: n / hop x the clip
    step = max(1, CHUNK_BYTES // per_clip)
    if wav.shape[0] <= step:
        y = K.Linear1x1.apply(_frames(wav, n, hop, pad), fwd, None, False)                 # exact-fp32 MFMA GEMM
    else:                                                                                  # chunks of clips: the frames of one chunk at a time
        y = torch.cat([K.Linear1x1.apply(_frames(wav[i:i + step], n, hop, pad), fwd, None, False) for i in range(0, wav.shape[0], step)])
    Kb = n // 2 + 1
    re, im = y[:, :Kb], y[:, Kb:]
    mag = torch.sqrt(re * re + im * im + mag_eps) if mag_eps else torch.sqrt(re * re + im * im)
    if not want_phase:
        return mag, None
    phase = torch.atan2(im.detach(), re.detach()) if detach_phase else torch.atan2(im, re)
    return mag, phase


def istft(magnitude, phase, n, hop, window_np, eps=1e-9):
    """magnitude, phase (N, n / 2 + 1, F) HIP tensors -> (N, (F - 1) hop) samples (transforms.py:71-101; host.istft)"""
    from . import kernels as K
    _, inv, w = _bases(n, window_np, magnitude.device)
    N, Kb, Fr = magnitude.shape
    m = magnitude.float()
    z = torch.cat([m * torch.cos(phase.float()), m * torch.sin(phase.float())], dim=1)     # (N, 2K, F)
    fr = K.Linear1x1.apply(z, inv, None, False) * w.view(1, -1, 1)                          # (N, n, F): windowed frames
    L = n + hop * (Fr - 1)
    ola = F.fold(fr, (1, L), (1, n), stride=(1, hop)).reshape(N, L)
    env = F.fold((w * w).view(1, -1, 1).expand(1, n, Fr).contiguous(), (1, L), (1, n), stride=(1, hop)).reshape(L)
    p = n // 2
    den = env + eps
    if eps == 0:
        den = torch.where(den > 0, den, torch.ones_like(den))
    return (ola / den)[:, p:L - p]


def is_fast_size(n):
    """sizes the register-FFT kernels take (psnd_stft_plan_bytes > 0): powers of two in [16, 8192]"""
    return n >= 16 and n <= 8192 and (n & (n - 1)) == 0
