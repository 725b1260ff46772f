"""Oracle (CPU restatement) of the reference's STFT -> magnitude -> mel pipeline.

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.  Plain numpy; every function
cites the reference lines (relative to /root/reference) it restates.

Two flavours of every floating-point routine:
  * ``*_ref32``  - follows the reference's own arithmetic (dense windowed-DFT
    basis rounded to float32, float32 products) and is what the golden vectors
    generated from the imported reference are compared against;
  * ``*_f64``    - the same mathematics in float64 (FFT based), used as the
    "truth" the HIP kernels are measured against with a stated tolerance.
Frame indexing is integer arithmetic and has exactly one flavour (bit-exact).
"""
import numpy as np

CENTER = 0   # STFT.transform / torch.stft(center=True): pad n_fft//2, reflect
NOPAD = 2    # no padding (kernel-internal: backward of the inverse STFT)
HIFIGAN = 1  # Audio2Mel / interface MelSpectrogram: pad (n_fft-hop)//2, reflect, center=False


# ----------------------------------------------------------------------------
# integer part: framing (bit-exact contract)
# ----------------------------------------------------------------------------
def pad_amount(n_fft: int, hop: int, framing: int) -> int:
    """pytorch_sound/models/transforms.py:25 (n/2) and :352 / interface/hifi_gan.py:37 ((n-h)/2)."""
    if framing == NOPAD:
        return 0
    return n_fft // 2 if framing == CENTER else (n_fft - hop) // 2


def frame_count(T: int, n_fft: int, hop: int, framing: int = CENTER) -> int:
    """Number of frames of a strided, un-padded conv over the reflect-padded signal.

    transforms.py:55-66 (F.pad + F.conv1d stride=hop, padding=0) and :353-360
    (torch.stft center=False on the manually padded signal).
    """
    p = pad_amount(n_fft, hop, framing)
    L = T + 2 * p
    if L < n_fft:
        return 0
    return (L - n_fft) // hop + 1


def reflect_index(i, T: int):
    """F.pad(mode='reflect') index map: x[-i] = x[i], x[T-1+i] = x[T-1-i] (no edge repeat).

    Valid for -T < i < 2T-1 (torch itself requires pad < T)."""
    i = np.asarray(i)
    i = np.where(i < 0, -i, i)
    i = np.where(i >= T, 2 * (T - 1) - i, i)
    return i


def frame_sample_index(f, m, T: int, n_fft: int, hop: int, framing: int = CENTER):
    """Original-sample index read by tap ``m`` of frame ``f``."""
    p = pad_amount(n_fft, hop, framing)
    return reflect_index(np.asarray(f) * hop - p + np.asarray(m), T)


def frames(wav: np.ndarray, n_fft: int, hop: int, framing: int = CENTER) -> np.ndarray:
    """(N,T) -> (N,F,n_fft) gather using the integer map above."""
    wav = np.asarray(wav)
    N, T = wav.shape
    F = frame_count(T, n_fft, hop, framing)
    idx = frame_sample_index(np.arange(F)[:, None], np.arange(n_fft)[None, :], T, n_fft, hop, framing)
    return wav[:, idx]


# ----------------------------------------------------------------------------
# window / basis
# ----------------------------------------------------------------------------
def hann_periodic(win_length: int) -> np.ndarray:
    """scipy.signal.get_window('hann', win, fftbins=True) == torch.hann_window(win)
    (transforms.py:30, :287).  float64."""
    m = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * m / win_length)


def pad_center(w: np.ndarray, size: int) -> np.ndarray:
    """librosa.util.pad_center as used at transforms.py:31: left pad (size-len)//2."""
    n = w.shape[-1]
    lpad = (size - n) // 2
    out = np.zeros(size, dtype=w.dtype)
    out[lpad:lpad + n] = w
    return out


def analysis_window(n_fft: int, win_length: int = None) -> np.ndarray:
    """float32 window exactly as the reference builds it (transforms.py:30-32)."""
    win_length = win_length or n_fft
    return pad_center(hann_periodic(win_length), n_fft).astype(np.float32)


def forward_basis_ref32(n_fft: int, win_length: int = None) -> np.ndarray:
    """(2K, n_fft) float32 = [Re;Im](fft(eye(n)))[:K] -> float32, times float32 window
    (transforms.py:35-45).  Im rows are -sin."""
    K = n_fft // 2 + 1
    fb = np.fft.fft(np.eye(n_fft))
    fb = np.vstack([np.real(fb[:K]), np.imag(fb[:K])]).astype(np.float32)
    return fb * analysis_window(n_fft, win_length)[None, :]


# ----------------------------------------------------------------------------
# STFT.transform (a3) / STFTTorchAudio (a5)
# ----------------------------------------------------------------------------
def stft_reim_ref32(wav, n_fft, hop, win_length=None, framing=CENTER):
    """float32 dense-DFT restatement of transforms.py:53-66. returns re, im (N,K,F)."""
    fr = frames(np.asarray(wav, np.float32), n_fft, hop, framing)           # N,F,n
    basis = forward_basis_ref32(n_fft, win_length)                           # 2K,n
    out = np.matmul(fr, basis.T)                                             # N,F,2K  (float32)
    out = np.transpose(out, (0, 2, 1))
    K = n_fft // 2 + 1
    return np.ascontiguousarray(out[:, :K]), np.ascontiguousarray(out[:, K:])


def stft_transform_ref32(wav, n_fft, hop, win_length=None, framing=CENTER):
    """mag, phase as transforms.py:67-69 (sqrt(re^2+im^2) - no eps; atan2(im, re))."""
    re, im = stft_reim_ref32(wav, n_fft, hop, win_length, framing)
    return np.sqrt(re * re + im * im), np.arctan2(im, re)


def stft_reim_f64(wav, n_fft, hop, win_length=None, framing=CENTER, window=None):
    """float64 truth: frame gather (integer map) * float32-rounded window -> rfft.

    The window is the float32 one the reference stores (so the only difference
    from ref32 is accumulation precision).  ``window`` overrides (n_fft taps)."""
    fr = frames(np.asarray(wav, np.float64), n_fft, hop, framing)
    w = analysis_window(n_fft, win_length).astype(np.float64) if window is None else np.asarray(window, np.float64)
    X = np.fft.rfft(fr * w[None, None, :], axis=-1)                          # N,F,K
    X = np.transpose(X, (0, 2, 1))
    return np.ascontiguousarray(X.real), np.ascontiguousarray(X.imag)


def stft_mag_f64(wav, n_fft, hop, win_length=None, framing=CENTER, window=None, eps=0.0):
    re, im = stft_reim_f64(wav, n_fft, hop, win_length, framing, window)
    return np.sqrt(re * re + im * im + eps)


def stft_mag_bwd_f64(gmag, wav, n_fft, hop, win_length=None, framing=CENTER, window=None, eps=0.0):
    """d(sum(gmag*mag))/d(wav) in float64 - adjoint of frame gather o window o rDFT o abs.

    What autograd computes through transforms.py:55-69 (sqrt has no eps: a bin that
    is exactly 0 yields NaN there; here 0/0 -> 0 is NOT applied, NaN propagates the same
    way only if mag==0 exactly)."""
    wav = np.asarray(wav, np.float64)
    N, T = wav.shape
    F = frame_count(T, n_fft, hop, framing)
    re, im = stft_reim_f64(wav, n_fft, hop, win_length, framing, window)
    mag = np.sqrt(re * re + im * im + eps)
    g = np.asarray(gmag, np.float64) / mag
    gre, gim = g * re, g * im                                                # N,K,F
    return stft_reim_bwd_f64(gre, gim, T, n_fft, hop, win_length, framing, window)


def stft_reim_bwd_f64(gre, gim, T, n_fft, hop, win_length=None, framing=CENTER, window=None):
    """adjoint of (wav -> re, im): gwav[t] = sum_{f,m: idx(f,m)=t} w[m] * sum_k (gre cos - gim sin)."""
    gre = np.asarray(gre, np.float64)
    gim = np.asarray(gim, np.float64)
    N, K, F = gre.shape
    w = analysis_window(n_fft, win_length).astype(np.float64) if window is None else np.asarray(window, np.float64)
    # adjoint of the one-sided DFT: halve the interior bins and use the unnormalised c2r
    G = (gre + 1j * gim).transpose(0, 2, 1).copy()                            # N,F,K
    G[..., 0] = G[..., 0].real
    if n_fft % 2 == 0:
        G[..., 1:K - 1] *= 0.5
        G[..., K - 1] = G[..., K - 1].real                                    # the Nyquist bin
    else:
        G[..., 1:K] *= 0.5                                                    # an odd size has no Nyquist bin (K = (n + 1) / 2: transforms.py:34)
    gfr = np.fft.irfft(G, n=n_fft, axis=-1) * n_fft * w[None, None, :]       # N,F,n
    idx = frame_sample_index(np.arange(F)[:, None], np.arange(n_fft)[None, :], T, n_fft, hop, framing)
    gw = np.zeros((N, T), np.float64)
    for b in range(N):
        np.add.at(gw[b], idx.ravel(), gfr[b].ravel())
    return gw


# ----------------------------------------------------------------------------
# mel filterbank: librosa==0.8.0 librosa.filters.mel restated (third party, unpinned)
# ----------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        big = f >= min_log_hz
        mels = mels.copy()
        mels[big] = min_log_mel + np.log(f[big] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if m.ndim:
        big = m >= min_log_mel
        freqs = freqs.copy()
        freqs[big] = min_log_hz * np.exp(logstep * (m[big] - min_log_mel))
    elif m >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (m - min_log_mel))
    return freqs


def mel_frequencies(n_mels, fmin, fmax):
    return _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels))


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with 0.8.0 defaults
    (htk=False, norm='slaney', dtype=float32).  Call sites: transforms.py:220,
    :339-341, interface/hifi_gan.py:42.  Returns (n_mels, 1+n_fft//2) float32."""
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    K = int(1 + n_fft // 2)
    weights = np.zeros((n_mels, K), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, K, endpoint=True)
    mel_f = mel_frequencies(n_mels + 2, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))      # float64 -> float32 store
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]                               # float64 multiply, float32 store
    return weights


# ----------------------------------------------------------------------------
# LogMelSpectrogram (a4), Audio2Mel / interface MelSpectrogram (a6)
# ----------------------------------------------------------------------------
def db_to_ln(db):
    """np.log(np.power(10, db / 10)) - transforms.py:223,227."""
    return np.log(np.power(10, db / 10))


def logmel_from_mag(mag, mel_filter, log_offset=1e-6, min_db=None, max_db=None, dtype=np.float32):
    """transforms.py:235-243.  NOTE truthiness gating (`if min_db:`): 0/None disables."""
    mel = np.matmul(np.asarray(mel_filter, dtype), np.asarray(mag, dtype))
    mel = np.log(mel + dtype(log_offset))
    if min_db:
        mel = np.maximum(mel, dtype(db_to_ln(min_db)))
    if max_db:
        mel = np.minimum(mel, dtype(db_to_ln(max_db)))
    return mel


def logmel_ref32(wav, sample_rate, mel_size, n_fft, win_length, hop, min_db=None, max_db=None,
                 mel_min=0.0, mel_max=None, log_offset=1e-6, mel_filter=None):
    """LogMelSpectrogram.forward.  The inner STFT is built with filter_length=win_length
    (transforms.py:217) - n_fft only sizes the mel filter."""
    mag, _ = stft_transform_ref32(wav, win_length, hop, None, CENTER)
    if mel_filter is None:
        mel_filter = mel_filterbank(sample_rate, n_fft, mel_size, mel_min, mel_max)
    return logmel_from_mag(mag, mel_filter, log_offset, min_db, max_db, np.float32)


def logmel_f64(wav, sample_rate, mel_size, n_fft, win_length, hop, min_db=None, max_db=None,
               mel_min=0.0, mel_max=None, log_offset=1e-6, mel_filter=None):
    mag = stft_mag_f64(wav, win_length, hop, None, CENTER)
    if mel_filter is None:
        mel_filter = mel_filterbank(sample_rate, n_fft, mel_size, mel_min, mel_max)
    return logmel_from_mag(mag, mel_filter, log_offset, min_db, max_db, np.float64)


def hifigan_mel_f64(wav, mel_filter, n_fft=1024, hop=256, win_length=1024, mag_eps=0.0,
                    clamp_min=1e-5, log10=False):
    """Audio2Mel.forward (transforms.py:351-366: mag_eps=0, log10) and
    interface MelSpectrogram.forward (interface/hifi_gan.py:46-63: mag_eps=1e-9, ln)."""
    mag = stft_mag_f64(wav, n_fft, hop, win_length, HIFIGAN, eps=mag_eps)
    mel = np.matmul(np.asarray(mel_filter, np.float64), mag)
    mel = np.maximum(mel, clamp_min)
    return np.log10(mel) if log10 else np.log(mel)


# ----------------------------------------------------------------------------
# STFT.inverse (next row f1) - overlap-add with the reference's pinv basis, in f64
# ----------------------------------------------------------------------------
def istft_f64(mag, phase, n_fft, hop, win_length=None, eps=1e-9):
    """Mathematical content of transforms.py:71-101.

    inverse_basis = pinv((n/h) * fourier_basis).T * window; for the full-rank one-sided
    stacked [cos; -sin] basis the pseudo-inverse synthesis of a *consistent* spectrum
    equals (h/n) * irDFT; the reference then divides by the squared-window overlap-add
    envelope (+eps), multiplies by n/h and trims n/2 each side.  For spectra that are
    not the DFT of a real frame the pinv is still the least-squares map; we therefore
    build the pinv explicitly for small n and use the closed form otherwise (the
    closed form is exact: rows of the stacked basis are orthogonal with known norms).
    """
    mag = np.asarray(mag, np.float64)
    phase = np.asarray(phase, np.float64)
    N, K, F = mag.shape
    w = analysis_window(n_fft, win_length).astype(np.float64)
    re = mag * np.cos(phase)
    im = mag * np.sin(phase)
    # closed-form pinv of B = (n/h) * [C; S] (C: K x n cosines, S: K x n (-sin)):
    #   B B^T is diagonal: (n/h)^2 * n/2 for interior rows, (n/h)^2 * n for k=0,n/2 cos rows,
    #   and 0 for the k=0,n/2 sine rows (those rows are identically zero).
    scale = (hop / n_fft)
    m = np.arange(n_fft)
    k = np.arange(K)
    ang = 2 * np.pi * np.outer(k, m) / n_fft
    C = np.cos(ang)
    S = -np.sin(ang)
    cn = np.full(K, 2.0 / n_fft)
    cn[0] = cn[K - 1] = 1.0 / n_fft
    sn = np.full(K, 2.0 / n_fft)
    sn[0] = sn[K - 1] = 0.0
    # frame[m] = scale * sum_k (cn re C + sn im S)
    fr = scale * (np.einsum('nkf,k,km->nfm', re, cn, C) + np.einsum('nkf,k,km->nfm', im, sn, S))
    fr = fr * w[None, None, :]
    L = (F - 1) * hop + n_fft
    out = np.zeros((N, L))
    env = np.zeros(L)
    for f in range(F):
        out[:, f * hop:f * hop + n_fft] += fr[:, f]
        env[f * hop:f * hop + n_fft] += w * w
    with np.errstate(divide='ignore', invalid='ignore'):     # eps = 0 (torch.istft): the envelope is zero only inside the trimmed n/2 edges
        out = out / (env + eps) * (n_fft / hop)
    p = n_fft // 2
    return out[:, p:L - p]
