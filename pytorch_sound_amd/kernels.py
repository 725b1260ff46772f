"""Thin torch-tensor front end of the C ABI (include/psnd.h).

PyTorch is plumbing here: it owns device memory and streams; every number is produced
by libpsnd_hip.so.  No function in this module has a CPU or eager fallback - a CPU tensor
or a missing library raises.
"""
import math
import os
from pytorch_sound_amd import _switches as _sw
import numpy as np
import torch

from . import _lib
from ._lib import PsndError  # noqa: F401
from ._lib import lib, check, ptr, stream_ptr, FRAMING_CENTER, FRAMING_HIFIGAN, FRAMING_NONE, LOG_NONE, LOG_E, LOG_10  # noqa: F401

_INF = float('inf')

# bench.py sets this to a list to collect (start, end) HIP event pairs recorded on the launch stream directly
# around every psnd_stft_fwd call (magnitude-only launches), i.e. without the Python work around it.
STFT_FWD_EVENTS = None
STFT_NFK_EVENTS = None          # the same for psnd_stft_mag_nfk


def _need_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PsndError('%s must be a CUDA(HIP) tensor: the MI355X path has no CPU fallback' % name)
    if t.dtype != torch.float32:
        raise _lib.PsndError('%s must be float32, got %s' % (name, t.dtype))


def plan_tensor(plan_np):
    """host plan (numpy uint8) -> torch uint8 tensor (CPU); callers register it as a buffer so
    Module.to(device) carries it to the GPU."""
    return torch.from_numpy(np.ascontiguousarray(plan_np))


def stft_plan(n_fft, window):
    return plan_tensor(_lib.build_stft_plan(n_fft, window))


def mel_plan(mel_filter):
    return plan_tensor(_lib.build_mel_plan(mel_filter))


def frame_count(T, n_fft, hop, framing=FRAMING_CENTER):
    return _lib.frame_count(T, n_fft, hop, framing)


def stft_forward(wav, n_fft, hop, plan, framing=FRAMING_CENTER, mag_eps=0.0,
                 want_mag=True, want_phase=False, want_reim=False, out_mag=None):
    """wav (N,T) fp32 cuda -> dict of requested (N,K,F) tensors.  out_mag: a caller-owned (N,K,F) fp32 tensor the magnitude is written
    into (feature extraction into persistent buffers, Trainer.static_prepare)."""
    _need_cuda(wav, 'wav')
    if wav.dim() != 2:
        raise _lib.PsndError('wav must be (N, T), got %s' % (tuple(wav.shape),))
    if plan.device != wav.device:
        raise _lib.PsndError('stft plan lives on %s but wav on %s (move the module with .to())' % (plan.device, wav.device))
    wav = wav.contiguous()
    N, T = wav.shape
    F = frame_count(T, n_fft, hop, framing)
    K = n_fft // 2 + 1
    mk = lambda: torch.empty((N, K, F), dtype=torch.float32, device=wav.device)  # noqa: E731
    if out_mag is not None and (not want_mag or tuple(out_mag.shape) != (N, K, F) or out_mag.dtype != torch.float32
                                or out_mag.device != wav.device or not out_mag.is_contiguous()):
        raise _lib.PsndError('stft_forward: out_mag must be a contiguous fp32 (%d, %d, %d) tensor on %s' % (N, K, F, wav.device))
    mag = (out_mag if out_mag is not None else mk()) if want_mag else None
    phase = mk() if want_phase else None
    re = mk() if want_reim else None
    im = mk() if want_reim else None
    ev = STFT_FWD_EVENTS if not torch.cuda.is_current_stream_capturing() else None   # no timing events inside a hipGraph capture
    with torch.cuda.device(wav.device):
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib().psnd_stft_fwd(ptr(wav), N, T, n_fft, hop, framing, ptr(plan), float(mag_eps),
                                  ptr(mag), ptr(phase), ptr(re), ptr(im), stream_ptr(wav.device)), 'psnd_stft_fwd')
        if ev is not None:
            e1.record()
            ev.append((e0, e1, N))
    return {'mag': mag, 'phase': phase, 're': re, 'im': im}


def stft_mag_nfk(wav, n_fft, hop, plan, framing=FRAMING_CENTER, mag_eps=0.0, out=None):
    """psnd_stft_mag_nfk: wav (N,T) fp32 -> magnitude (N, F, K), bin axis fastest (`.transpose(1, 2)` is the reference's (N, K, F)
    as a view).  For consumers inside this library that take the layout (mel kernel, channels-last conv stack); no autograd."""
    _need_cuda(wav, 'wav')
    if wav.dim() != 2:
        raise _lib.PsndError('wav must be (N, T), got %s' % (tuple(wav.shape),))
    if plan.device != wav.device:
        raise _lib.PsndError('stft plan lives on %s but wav on %s (move the module with .to())' % (plan.device, wav.device))
    wav = wav.detach().contiguous()
    N, T = wav.shape
    F = frame_count(T, n_fft, hop, framing)
    K = n_fft // 2 + 1
    if out is None:
        out = torch.empty((N, F, K), dtype=torch.float32, device=wav.device)
    elif tuple(out.shape) != (N, F, K) or out.dtype != torch.float32 or out.device != wav.device or not out.is_contiguous():
        raise _lib.PsndError('stft_mag_nfk: out must be a contiguous fp32 (%d, %d, %d) tensor on %s' % (N, F, K, wav.device))
    ev = STFT_NFK_EVENTS if not torch.cuda.is_current_stream_capturing() else None
    with torch.cuda.device(wav.device):
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib().psnd_stft_mag_nfk(ptr(wav), N, T, n_fft, hop, framing, ptr(plan), float(mag_eps), ptr(out), stream_ptr(wav.device)),
              'psnd_stft_mag_nfk')
        if ev is not None:
            e1.record()
            ev.append((e0, e1, N))
    return out


def stft_backward(wav, n_fft, hop, plan, framing=FRAMING_CENTER, mag_eps=0.0, gmag=None, gre=None, gim=None):
    _need_cuda(wav, 'wav')
    wav = wav.contiguous()
    N, T = wav.shape
    gs = [None if g is None else g.contiguous() for g in (gmag, gre, gim)]
    for g in gs:
        if g is not None:
            _need_cuda(g, 'grad')
    gwav = torch.empty_like(wav)
    with torch.cuda.device(wav.device):
        check(lib().psnd_stft_bwd(ptr(wav), N, T, n_fft, hop, framing, ptr(plan), float(mag_eps),
                                  ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gwav), stream_ptr(wav.device)),
              'psnd_stft_bwd')
    return gwav


def _clamp_args(clamp_lo, clamp_hi, pre_clamp_min):
    lo = -_INF if clamp_lo is None else float(clamp_lo)
    hi = _INF if clamp_hi is None else float(clamp_hi)
    pre = -1.0 if pre_clamp_min is None else float(pre_clamp_min)
    return lo, hi, pre


def mel_forward(mag, mel_plan_t, M, log_kind=LOG_E, log_offset=0.0, pre_clamp_min=None,
                clamp_lo=None, clamp_hi=None, want_lin=False, out=None):
    _need_cuda(mag, 'mag')
    mag = mag.contiguous()
    N, K, F = mag.shape
    lo, hi, pre = _clamp_args(clamp_lo, clamp_hi, pre_clamp_min)
    if out is not None and (tuple(out.shape) != (N, M, F) or out.dtype != torch.float32 or out.device != mag.device
                            or not out.is_contiguous()):
        raise _lib.PsndError('mel_forward: out must be a contiguous fp32 (%d, %d, %d) tensor on %s' % (N, M, F, mag.device))
    if out is None:
        out = torch.empty((N, M, F), dtype=torch.float32, device=mag.device)
    lin = torch.empty_like(out) if want_lin else None
    with torch.cuda.device(mag.device):
        check(lib().psnd_mel_fwd(ptr(mag), N, F, M, K, ptr(mel_plan_t), log_kind, float(log_offset), pre, lo, hi,
                                 ptr(out), ptr(lin), stream_ptr(mag.device)), 'psnd_mel_fwd')
    return out, lin


def mel_forward_nfk(mag_nfk, mel_plan_t, M, log_kind=LOG_E, log_offset=0.0, pre_clamp_min=None, clamp_lo=None, clamp_hi=None,
                    want_lin=False, out=None):
    """psnd_mel_fwd_nfk: the mel projection of a BIN-FASTEST magnitude (N, F, K) (stft_mag_nfk); the result is (N, M, F) as mel_forward's"""
    _need_cuda(mag_nfk, 'mag_nfk')
    mag_nfk = mag_nfk.contiguous()
    N, F, K = mag_nfk.shape
    lo, hi, pre = _clamp_args(clamp_lo, clamp_hi, pre_clamp_min)
    if out is not None and (tuple(out.shape) != (N, M, F) or out.dtype != torch.float32 or out.device != mag_nfk.device
                            or not out.is_contiguous()):
        raise _lib.PsndError('mel_forward_nfk: out must be a contiguous fp32 (%d, %d, %d) tensor on %s' % (N, M, F, mag_nfk.device))
    if out is None:
        out = torch.empty((N, M, F), dtype=torch.float32, device=mag_nfk.device)
    lin = torch.empty_like(out) if want_lin else None
    with torch.cuda.device(mag_nfk.device):
        check(lib().psnd_mel_fwd_nfk(ptr(mag_nfk), N, F, M, K, ptr(mel_plan_t), log_kind, float(log_offset), pre, lo, hi,
                                     ptr(out), ptr(lin), stream_ptr(mag_nfk.device)), 'psnd_mel_fwd_nfk')
    return out, lin


def mel_backward_nfk(gout, mel_lin, mel_plan_t, K, log_kind=LOG_E, log_offset=0.0, pre_clamp_min=None, clamp_lo=None, clamp_hi=None):
    """psnd_mel_bwd_nfk: gradient of mel_forward_nfk w.r.t. the magnitude, (N, F, K)"""
    _need_cuda(gout, 'gout')
    gout = gout.contiguous()
    N, M, F = gout.shape
    lo, hi, pre = _clamp_args(clamp_lo, clamp_hi, pre_clamp_min)
    gmag = torch.empty((N, F, K), dtype=torch.float32, device=gout.device)
    with torch.cuda.device(gout.device):
        check(lib().psnd_mel_bwd_nfk(ptr(gout), ptr(mel_lin), N, F, M, K, ptr(mel_plan_t), log_kind, float(log_offset), pre, lo, hi,
                                     ptr(gmag), stream_ptr(gout.device)), 'psnd_mel_bwd_nfk')
    return gmag


class MelLogNfk(torch.autograd.Function):
    """MelLog for a bin-fastest magnitude: (N, F, K) -> (N, M, F)"""

    @staticmethod
    def forward(ctx, mag_nfk, plan, M, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi):
        need_grad = ctx.needs_input_grad[0]
        out, lin = mel_forward_nfk(mag_nfk, plan, M, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, want_lin=need_grad)
        if need_grad:
            ctx.save_for_backward(lin, plan)
        ctx.cfg = (mag_nfk.shape[2], log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi)
        return out

    @staticmethod
    def backward(ctx, gout):
        lin, plan = ctx.saved_tensors
        K, log_kind, log_offset, pre, lo, hi = ctx.cfg
        return mel_backward_nfk(gout, lin, plan, K, log_kind, log_offset, pre, lo, hi), None, None, None, None, None, None, None


def logmel_fused_ok(wav, n_fft, hop):
    """the fused wav -> log-mel kernel covers the span-staged tile size and needs no gradient (forward only)"""
    return (n_fft == 1024 and 0 < hop <= 256 and hop % 4 == 0 and wav.is_cuda
            and not (torch.is_grad_enabled() and wav.requires_grad))


def logmel_forward(wav, n_fft, hop, stft_plan_t, mel_plan_t, M, framing=FRAMING_CENTER, mag_eps=0.0, log_kind=LOG_E,
                   log_offset=0.0, pre_clamp_min=None, clamp_lo=None, clamp_hi=None):
    """psnd_logmel_fwd: (N,T) waveform -> (N,M,F) log-mel, magnitude kept on chip"""
    _need_cuda(wav, 'wav')
    wav = wav.detach().contiguous()
    N, T = wav.shape
    F = frame_count(T, n_fft, hop, framing)
    lo, hi, pre = _clamp_args(clamp_lo, clamp_hi, pre_clamp_min)
    out = torch.empty((N, M, max(F, 0)), dtype=torch.float32, device=wav.device)
    with torch.cuda.device(wav.device):
        check(lib().psnd_logmel_fwd(ptr(wav), N, T, n_fft, hop, framing, ptr(stft_plan_t), float(mag_eps), M,
                                    ptr(mel_plan_t), log_kind, float(log_offset), pre, lo, hi, ptr(out),
                                    stream_ptr(wav.device)), 'psnd_logmel_fwd')
    return out


def mel_backward(gout, mel_lin, mel_plan_t, K, log_kind=LOG_E, log_offset=0.0, pre_clamp_min=None,
                 clamp_lo=None, clamp_hi=None):
    _need_cuda(gout, 'gout')
    gout = gout.contiguous()
    N, M, F = gout.shape
    lo, hi, pre = _clamp_args(clamp_lo, clamp_hi, pre_clamp_min)
    gmag = torch.empty((N, K, F), dtype=torch.float32, device=gout.device)
    with torch.cuda.device(gout.device):
        check(lib().psnd_mel_bwd(ptr(gout), ptr(mel_lin), N, F, M, K, ptr(mel_plan_t), log_kind, float(log_offset),
                                 pre, lo, hi, ptr(gmag), stream_ptr(gout.device)), 'psnd_mel_bwd')
    return gmag


# ---------------------------------------------------------------------------------------------
# autograd glue
# ---------------------------------------------------------------------------------------------
class StftMagPhase(torch.autograd.Function):
    """(mag, phase) of STFT.transform (transforms.py:53-69).  phase is detached there
    (atan2 of .data) -> marked non-differentiable here."""

    @staticmethod
    def forward(ctx, wav, plan, n_fft, hop, framing, mag_eps, want_phase):
        o = stft_forward(wav, n_fft, hop, plan, framing, mag_eps, True, want_phase, False)
        ctx.save_for_backward(wav, plan)
        ctx.cfg = (n_fft, hop, framing, mag_eps)
        if want_phase:
            ctx.mark_non_differentiable(o['phase'])
            return o['mag'], o['phase']
        return o['mag'], None

    @staticmethod
    def backward(ctx, gmag, _gphase):
        wav, plan = ctx.saved_tensors
        n_fft, hop, framing, mag_eps = ctx.cfg
        gw = stft_backward(wav, n_fft, hop, plan, framing, mag_eps, gmag=gmag)
        return gw, None, None, None, None, None, None


class StftReIm(torch.autograd.Function):
    """(re, im) of torch.stft as STFTTorchAudio.forward uses it (transforms.py:297-303); both
    outputs are differentiable (transform() there does NOT detach the phase, :311)."""

    @staticmethod
    def forward(ctx, wav, plan, n_fft, hop, framing):
        o = stft_forward(wav, n_fft, hop, plan, framing, 0.0, False, False, True)
        ctx.save_for_backward(wav, plan)
        ctx.cfg = (n_fft, hop, framing)
        return o['re'], o['im']

    @staticmethod
    def backward(ctx, gre, gim):
        wav, plan = ctx.saved_tensors
        n_fft, hop, framing = ctx.cfg
        if gre is None:
            gre = torch.zeros_like(gim)
        if gim is None:
            gim = torch.zeros_like(gre)
        gw = stft_backward(wav, n_fft, hop, plan, framing, 0.0, gre=gre, gim=gim)
        return gw, None, None, None, None


class StftPolar(torch.autograd.Function):
    """(magnitude, phase) of STFTTorchAudio.transform (transforms.py:305-311): both outputs come out of ONE psnd_stft_fwd launch and
    both are differentiable (the reference takes atan2 of the live tensors there).  Backward: a magnitude-only gradient takes the
    tuned magnitude adjoint; with a phase gradient, psnd_polar_bwd turns (g_mag, g_phase) into (g_re, g_im) in one pass and the
    (re, im) adjoint carries on."""

    @staticmethod
    def forward(ctx, wav, plan, n_fft, hop, framing):
        o = stft_forward(wav, n_fft, hop, plan, framing, 0.0, True, True, False)
        ctx.save_for_backward(wav, plan, o['mag'], o['phase'])
        ctx.cfg = (n_fft, hop, framing)
        return o['mag'], o['phase']

    @staticmethod
    def backward(ctx, gmag, gphase):
        wav, plan, mag, phase = ctx.saved_tensors
        n_fft, hop, framing = ctx.cfg
        if gphase is None:
            return stft_backward(wav, n_fft, hop, plan, framing, 0.0, gmag=gmag), None, None, None, None
        gphase = gphase.contiguous()
        gmag = None if gmag is None else gmag.contiguous()
        gre, gim = torch.empty_like(mag), torch.empty_like(mag)
        with torch.cuda.device(mag.device):
            check(lib().psnd_polar_bwd(ptr(gmag), ptr(gphase), ptr(mag), ptr(phase), mag.numel(), ptr(gre), ptr(gim),
                                       stream_ptr(mag.device)), 'psnd_polar_bwd')
        return stft_backward(wav, n_fft, hop, plan, framing, 0.0, gre=gre, gim=gim), None, None, None, None


class MelLog(torch.autograd.Function):
    """clamp(log(max(W @ mag, pre) + off), lo, hi) - transforms.py:235-243 / :364-365 /
    interface/hifi_gan.py:58-61."""

    @staticmethod
    def forward(ctx, mag, plan, M, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi):
        need_grad = ctx.needs_input_grad[0]
        out, lin = mel_forward(mag, plan, M, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, want_lin=need_grad)
        if need_grad:
            ctx.save_for_backward(lin, plan)
        ctx.cfg = (mag.shape[1], log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi)
        return out

    @staticmethod
    def backward(ctx, gout):
        lin, plan = ctx.saved_tensors
        K, log_kind, log_offset, pre, lo, hi = ctx.cfg
        gmag = mel_backward(gout, lin, plan, K, log_kind, log_offset, pre, lo, hi)
        return gmag, None, None, None, None, None, None, None


def db_to_ln(db):
    """np.log(np.power(10, db / 10)) of transforms.py:223,227"""
    return math.log(math.pow(10.0, db / 10.0))


def istft_forward(magnitude, phase, n_fft, hop, plan, eps=1e-9):
    _need_cuda(magnitude, 'magnitude')
    _need_cuda(phase, 'phase')
    magnitude, phase = magnitude.contiguous(), phase.contiguous()
    N, Kb, F = magnitude.shape
    if Kb != n_fft // 2 + 1 or phase.shape != magnitude.shape:
        raise _lib.PsndError('istft: expected (N, %d, F) magnitude and phase, got %s / %s'
                             % (n_fft // 2 + 1, tuple(magnitude.shape), tuple(phase.shape)))
    out = torch.empty((N, max((F - 1) * hop, 0)), dtype=torch.float32, device=magnitude.device)
    with torch.cuda.device(magnitude.device):
        check(lib().psnd_istft(ptr(magnitude), ptr(phase), N, F, n_fft, hop, ptr(plan), float(eps), ptr(out),
                               stream_ptr(magnitude.device)), 'psnd_istft')
    return out


_OLA_ENV = {}


def _ola_envelope(window, n_fft, hop, F):
    """sum_f window^2[t - f hop] over F frames, length (F - 1) hop + n_fft - the overlap-add envelope STFT.inverse divides by
    (transforms.py:90-99 builds it with a conv_transpose1d of ones; here ceil(n_fft / hop) shifted block adds, cached per window / geometry:
    no library convolution for a HIP tensor)."""
    import weakref
    key = (id(window), window._version, n_fft, hop, F)
    hit = _OLA_ENV.get(key)
    env = hit[1] if hit is not None and hit[0]() is window else None      # the id of a freed tensor may be reused: the weak reference tells
    if env is None:
        if len(_OLA_ENV) > 32:
            _OLA_ENV.clear()
        # deterministic: the window squared cut into R = ceil(n_fft / hop) pieces of `hop` samples; piece r of every frame lands on the hop
        # blocks r .. r + F - 1, added piece by piece in a fixed order (an index_add_ of F shifted copies adds with float atomics)
        w2 = (window.detach().float() * window.detach().float()).reshape(-1)
        R = (n_fft + hop - 1) // hop
        w2 = torch.nn.functional.pad(w2, (0, R * hop - n_fft)).view(R, hop)
        blocks = torch.zeros((F - 1 + R, hop), dtype=torch.float32, device=window.device)
        for r in range(R):
            blocks[r:r + F] += w2[r]
        env = blocks.reshape(-1)[:(F - 1) * hop + n_fft].contiguous()
        _OLA_ENV[key] = (weakref.ref(window), env)
    return env


class IStft(torch.autograd.Function):
    """STFT.inverse (transforms.py:71-101).  Backward: the map is linear in G = s_k * mag * e^{i phase}
    (s = 1/n for DC/Nyquist, 2/n otherwise; their imaginary parts do not reach the signal), and its adjoint is a
    plain windowed forward DFT (no padding) of the envelope-scaled output gradient - psnd_stft_fwd again."""

    @staticmethod
    def forward(ctx, magnitude, phase, window, plan, n_fft, hop, eps):
        out = istft_forward(magnitude, phase, n_fft, hop, plan, eps)
        ctx.save_for_backward(magnitude, phase, window, plan)
        ctx.cfg = (n_fft, hop, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        magnitude, phase, window, plan = ctx.saved_tensors
        n_fft, hop, eps = ctx.cfg
        N, Kb, F = magnitude.shape
        env = _ola_envelope(window, n_fft, hop, F)
        p = n_fft // 2
        gp = torch.nn.functional.pad(g.contiguous(), (p, p))
        den = env + eps
        # eps = 0 (torch.istft's convention): the envelope vanishes only inside the trimmed n/2 edges, where the gradient is zero too
        gp = torch.where(den > 0, gp / den, torch.zeros_like(gp))
        o = stft_forward(gp, n_fft, hop, plan, FRAMING_NONE, 0.0, False, False, True)
        s = torch.full((Kb,), 2.0 / n_fft, device=g.device)
        s[0] = s[-1] = 1.0 / n_fft
        are, aim = o['re'] * s.view(1, -1, 1), o['im'] * s.view(1, -1, 1)
        aim[:, 0] = 0
        aim[:, -1] = 0
        cs, sn = torch.cos(phase), torch.sin(phase)
        g_mag = cs * are + sn * aim
        g_phase = magnitude * (cs * aim - sn * are)
        return g_mag, g_phase, None, None, None, None, None


def istft(magnitude, phase, n_fft, hop, plan, eps=1e-9, window=None):
    """STFT.inverse - differentiable when `window` (n_fft taps, on the device) is given."""
    if window is not None and (magnitude.requires_grad or phase.requires_grad):
        return IStft.apply(magnitude, phase, window, plan, n_fft, hop, eps)
    return istft_forward(magnitude, phase, n_fft, hop, plan, eps)


# ---------------------------------------------------------------------------------------------
# transformer blocks (models/modules.py): GroupNorm(1, C)(x + res) and the masked softmax over keys
# ---------------------------------------------------------------------------------------------
RESIDUAL_LINKS = _sw.lab('PSND_RESIDUAL_LINKS', '1') == '1'      # 0: autograd accumulates the two gradients of a block's input (A/B)


class ResidualLink:
    """One residual connection y = norm(f(x) + x) whose branch f starts with a Linear1x1 on the same x (MultiHeadAttention,
    PointwiseFeedForward): autograd would add the two gradients of x in a pass of its own (three tensor sweeps); instead the norm's
    backward hands its gradient for x to the link and returns none, and the Linear1x1's backward - which runs later in the same pass and
    consumes it - writes W^T gy + that gradient (psnd_linear1x1_bwd_acc).  The projection registers itself when its forward was recorded
    with x requiring a gradient; without a registered consumer the norm returns its gradient as usual."""

    def __init__(self):
        self.consumer = False      # a Linear1x1 node that will produce a gradient for x holds this link
        self.g = None

    def offer(self, g) -> bool:
        if not self.consumer or not RESIDUAL_LINKS:
            return False
        self.g = g
        return True

    def take(self):
        g, self.g = self.g, None
        return g


RELU_LINKS = _sw.lab('PSND_RELU_LINKS', '1') == '1'              # 0: each GEMM behind the ReLU masks its operand itself (A/B)


PAD_HIDDEN_ROWS = _sw.lab('PSND_FFN_PAD_ROWS', '1') == '1'       # 0: the bf16 hidden tensor's rows are T frames long (A/B)
HIDDEN_BF16 = _sw.lab('PSND_FFN_HIDDEN_BF16', '1') == '1'      # 0: the feed-forward pair's hidden tensor stays fp32 under autocast (A/B)


class ReluLink:
    """Conv1d -> ReLU -> Conv1d (PointwiseFeedForward, modules.py:93-95) as two Linear1x1 nodes: the first one's backward would read
    its incoming gradient together with the ReLU's output as a mask in both of its GEMMs and its bias sum (two 169 MB tensors each at 32
    clips x 1292 frames x 1024 channels).  With a link the SECOND node, which produces that gradient and holds the ReLU's output as its
    input, zeroes it where the ReLU was off in its GEMM's epilogue (psnd_linear1x1_bwd_ex, gx_mask), says so here, and the first node
    reads the gradient alone.  The values are the same either way (a masked element is an exact zero in both)."""

    def __init__(self):
        self.consumer = False      # the second node was recorded on the first one's output and will produce a gradient for it
        self.y = None              # (address, shape) of the ReLU's output as the first node returned it: the second node checks that ITS
                                   # input is that tensor (no reference: the output's grad_fn holds the link)
        self.masked = False        # set by the second node's backward: the gradient arrives masked

    def take(self) -> bool:
        m, self.masked = self.masked, False
        return m


class GroupNorm1(torch.autograd.Function):
    """y = GroupNorm(1, C)(x + res) [-> ReLU]: statistics over (C x T) per sample (modules.py:58, :114-116)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, relu, link=None):
        _need_cuda(x, 'x')
        x = x.contiguous()
        res = None if res is None else res.contiguous()
        ctx.link = link if (link is not None and res is not None) else None
        N, C, T = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((N, 2), dtype=torch.float32, device=x.device)
        ws = torch.empty((N, 2 * C), dtype=torch.float64, device=x.device)         # PSND_GN_WS_DOUBLES(C) per sample: a pair of sums per row
        g32, b32 = gamma.detach().contiguous(), beta.detach().contiguous()
        with torch.cuda.device(x.device):
            check(lib().psnd_groupnorm1_fwd(ptr(x), ptr(res), ptr(g32), ptr(b32), N, C, T, float(eps), int(relu), ptr(y),
                                            ptr(stats), ptr(ws), stream_ptr(x.device)), 'psnd_groupnorm1_fwd')
        ctx.relu, ctx.has_res = bool(relu), res is not None
        ctx.save_for_backward(x, res, g32, y if relu else None, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, res, g32, y, stats = ctx.saved_tensors
        gy = gy.contiguous()
        N, C, T = x.shape
        gx = torch.empty_like(x)
        gg = torch.empty(C, dtype=torch.float32, device=x.device)
        gb = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty((N, 2 * C), dtype=torch.float64, device=x.device)         # PSND_GN_WS_DOUBLES(C) per sample: a pair of sums per row
        with torch.cuda.device(x.device):
            check(lib().psnd_groupnorm1_bwd(ptr(gy), ptr(x), ptr(res), ptr(g32), ptr(y), ptr(stats), N, C, T, int(ctx.relu),
                                            ptr(gx), ptr(gg), ptr(gb), ptr(ws), stream_ptr(x.device)), 'psnd_groupnorm1_bwd')
        if ctx.link is not None and ctx.needs_input_grad[1] and ctx.link.offer(gx):
            # the residual's gradient travels with the link: the first projection of the branch adds it in its input-gradient GEMM
            return gx, None, gg, gb, None, None, None
        return gx, (gx if ctx.has_res else None), gg, gb, None, None, None


class SoftmaxKeys(torch.autograd.Function):
    """att = softmax over keys (dim 1) of scale * scores, key-padded rows -> -inf before, query-padded columns -> 0
    after (modules.py:66-76).  In place on `scores` when it is a non-leaf temporary (the bmm output), else on a copy."""

    @staticmethod
    def forward(ctx, scores, mask_u8, scale):
        _need_cuda(scores, 'scores')
        if scores.is_contiguous() and not (scores.is_leaf and scores.requires_grad):
            att = scores                               # the bmm output is a temporary: softmax in place, no extra pass
            ctx.mark_dirty(scores)
        else:
            att = scores.contiguous().clone()
        B, T, _ = att.shape
        with torch.cuda.device(att.device):
            check(lib().psnd_softmax_keys_fwd(ptr(att), ptr(mask_u8), B, T, float(scale), stream_ptr(att.device)),
                  'psnd_softmax_keys_fwd')
        ctx.scale = float(scale)
        ctx.save_for_backward(att)
        return att

    @staticmethod
    def backward(ctx, gatt):
        (att,) = ctx.saved_tensors
        gatt = gatt.contiguous()
        B, T, _ = att.shape
        gs = torch.empty_like(att)
        with torch.cuda.device(att.device):
            check(lib().psnd_softmax_keys_bwd(ptr(att), ptr(gatt), B, T, ctx.scale, ptr(gs), stream_ptr(att.device)),
                  'psnd_softmax_keys_bwd')
        return gs, None, None


# ---------------------------------------------------------------------------------------------
# models/sound.py: PreEmphasis and multi_stft_loss
# ---------------------------------------------------------------------------------------------
class Linear1x1(torch.autograd.Function):
    """y = W x + b per time step (a 1x1 Conv1d, modules.py:21-22, 93-95) on the matrix-core GEMM (psnd_linear1x1_*): exact fp32
    products, or bf16 operands / fp32 accumulation with `bf16=True`; optionally with the ReLU that follows it fused (its backward
    masks by y > 0).  x: (N, Cin, T), w: (Cout, Cin) or (Cout, Cin, 1); output and gradients fp32."""

    @staticmethod
    def forward(ctx, x, w, bias, relu, bf16=False, link=None, relu_link=None, out_h=False, t_len=None, pad_h=False):
        # t_len: the frames of a row in use when x is a bf16 tensor with padded rows (N, Cin, ld_h >= t_len); pad_h: the bf16 output gets such
        # rows (ld_h = the next multiple of 64 frames: every row starts on a 128-byte line) - a tensor between two nodes of this class only
        # out_h (with bf16 operands): the output is STORED as bf16 - the hidden tensor of Conv1d -> ReLU -> Conv1d under autocast; the node
        # behind it takes it as it is (x.dtype == bfloat16).  The products round their operands to bf16 anyway: same values, half the bytes.
        x_h = x.dtype == torch.bfloat16
        if not (x_h and isinstance(x, torch.Tensor) and x.is_cuda):
            _need_cuda(x, 'input')
        if (x_h or out_h) and not bf16:
            raise PsndError('Linear1x1: bf16 storage comes with bf16 operands (torch.autocast) only')
        if x_h and out_h:
            raise PsndError('Linear1x1: bf16 storage on one side of a projection only')
        if x_h:
            link = None
        ctx.link = None
        if link is not None and ctx.needs_input_grad[0]:
            ctx.link, link.consumer = link, True
        # relu_link: this node is the one BEFORE the ReLU (relu=True: it publishes its output) or the one AFTER it (its input is that
        # output, and a gradient for it is wanted)
        ctx.relu_out = ctx.relu_in = None
        if relu_link is not None and RELU_LINKS:
            if relu:
                ctx.relu_out = relu_link
            elif relu_link.y == (x.data_ptr(), tuple(x.shape)) and x.is_contiguous() and ctx.needs_input_grad[0]:
                ctx.relu_in, relu_link.consumer = relu_link, True
        x = x.contiguous()
        w2 = w.reshape(w.shape[0], w.shape[1]).contiguous()
        N, Cin, T = x.shape
        ld = 0
        if t_len is not None and int(t_len) != T:
            if not x_h or int(t_len) > T or int(t_len) <= 0:
                raise PsndError('Linear1x1: t_len=%s with an input %s (padded rows are a bf16 tensor\'s)' % (t_len, tuple(x.shape)))
            ld, T = T, int(t_len)
        Cout = w2.shape[0]
        if w2.shape[1] != Cin:
            raise PsndError('Linear1x1: weight %s does not fit %d input channels' % (tuple(w.shape), Cin))
        b = None if bias is None else bias.contiguous()
        if out_h and pad_h and T % 64:
            ld = (T + 63) // 64 * 64
        y = torch.empty((N, Cout, ld if (out_h and ld) else T), dtype=torch.bfloat16 if out_h else torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib().psnd_linear1x1_fwd_ex(ptr(x), ptr(w2), ptr(b), N, Cin, Cout, T, int(bool(relu)), int(bool(bf16)), 1 if x_h else (2 if out_h else 0),
                                              ld, ptr(y), stream_ptr(x.device)), 'psnd_linear1x1_fwd')
        ctx.relu, ctx.has_bias, ctx.wshape, ctx.bf16 = bool(relu), bias is not None, tuple(w.shape), bool(bf16)
        ctx.x_h, ctx.out_h, ctx.t_len = x_h, bool(out_h), T
        ctx.params = (w, bias)
        from . import cl
        cl.note_param_use(ctx, w, bias)
        ctx.save_for_backward(x, w2, y if relu else None)
        if ctx.relu_out is not None:
            ctx.relu_out.y = (y.data_ptr(), tuple(y.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w2, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.relu_out is not None and ctx.relu_out.take():
            y = None                                                    # the gradient arrives masked (ReluLink)
        T = ctx.t_len
        if gy.dtype != torch.float32 and (y is not None or ctx.x_h or not ctx.bf16):
            # a bf16 gradient that still needs this node's mask, or bf16 on both sides: outside the feed-forward pair's plan - in fp32
            gy = gy[..., :T].float()
            if y is not None:
                gy, y = gy * (y[..., :T] > 0), None
            gy = gy.contiguous()
        io_h = 1 if gy.dtype == torch.bfloat16 else (2 if ctx.x_h else 0)
        ld = (gy.shape[-1] if io_h == 1 else x.shape[-1]) if io_h else 0
        if ld == T:
            ld = 0
        xmask = x if ctx.relu_in is not None else None
        N, Cin = x.shape[0], x.shape[1]
        Cout = w2.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dev = x.device
        addend = ctx.link.take() if ctx.link is not None else None      # the residual branch's gradient for x (ResidualLink)
        if addend is not None and (addend.shape != x.shape or addend.dtype != torch.float32 or not addend.is_contiguous()):
            raise PsndError('Linear1x1: residual gradient %s does not fit the input %s' % (tuple(addend.shape), tuple(x.shape)))
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty_like(w2) if need_w else None
        gb = torch.empty(Cout, dtype=torch.float32, device=dev) if need_b else None
        part = None
        if need_w:
            slabs = int(lib().psnd_linear1x1_wgrad_slabs(N, Cin, Cout, T))
            part = torch.empty((slabs, Cout, Cin), dtype=torch.float32, device=dev)
        # The parameter side (weight-gradient GEMM, slab sum, bias row sum: needed only by the optimizer) on the parameter stream - inside
        # the step graph a branch next to the input-gradient chain that goes on, joined at the end of the backward pass (cl.py,
        # BRANCH_PARAM_GRADS).  Only when these gradients are WRITTEN (leaf parameters without a .grad: autograd takes the tensors without
        # a launch) and nothing else shares the hardware queues with the step.
        from . import cl
        side = None
        if (need_x and (need_w or need_b) and cl.BRANCH_PARAM_GRADS and cl.AUTO_SECTIONS
                and all(q is None or (q.is_leaf and q.grad is None and cl.single_use(q)) for q in ctx.params)):
            side = cl.param_stream(dev)
            if cl.GRAD_SINK is not None:       # a reducer's bucket waits for the stream that WRITES these gradients (round 5)
                cl.GRAD_SINK.note_producer(ctx.params, side)
        with torch.cuda.device(dev):
            if side is None:
                check(lib().psnd_linear1x1_bwd_ex(ptr(gy), ptr(y), ptr(x), ptr(w2), N, Cin, Cout, T, int(ctx.bf16), io_h, ld, ptr(addend), ptr(xmask), ptr(gx),
                                                  ptr(gw), ptr(part), ptr(gb), stream_ptr(dev)), 'psnd_linear1x1_bwd')
            else:
                main = torch.cuda.current_stream(dev)
                side.wait_stream(main)                          # gy is complete on the main stream
                check(lib().psnd_linear1x1_bwd_ex(ptr(gy), ptr(y), ptr(x), ptr(w2), N, Cin, Cout, T, int(ctx.bf16), io_h, ld, ptr(addend), ptr(xmask), ptr(gx),
                                                  None, None, None, stream_ptr(dev)), 'psnd_linear1x1_bwd')
                with torch.cuda.stream(side):
                    check(lib().psnd_linear1x1_bwd_ex(ptr(gy), ptr(y), ptr(x), ptr(w2), N, Cin, Cout, T, int(ctx.bf16), io_h, ld, None, None, None, ptr(gw),
                                                      ptr(part), ptr(gb), stream_ptr(dev)), 'psnd_linear1x1_bwd')
                for t in (gy, y, x, w2, gw, part, gb):          # main-stream blocks the side stream reads / writes
                    if t is not None:
                        t.record_stream(side)
                cl._join_side_at_end_of_backward(dev, side)
        if xmask is not None and need_x:
            ctx.relu_in.masked = True
        cl.consume_param_use(ctx)
        return gx, (None if gw is None else gw.view(ctx.wshape)), gb, None, None, None, None, None, None, None


class Im2Col(torch.autograd.Function):
    """the unfolded input of a 1-d convolution, col[n][ci * k + j][t] = act(x[n][ci][(t * stride + j * dil - pad) / up]) (psnd_im2col_f32; zero
    outside the signal and between the samples of a zero-spread input), act = leaky-relu with `slope` (1.0: none); backward psnd_col2im_f32"""

    @staticmethod
    def forward(ctx, x, k, dil, pad, stride, up, To, slope):
        _need_cuda(x, 'input')
        x = x.contiguous()
        N, C, T = x.shape
        col = torch.empty((N, C * k, To), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib().psnd_im2col_f32(ptr(x), N, C, T, k, dil, pad, stride, up, To, float(slope), ptr(col), stream_ptr(x.device)), 'psnd_im2col_f32')
        ctx.geom = (int(k), int(dil), int(pad), int(stride), int(up), int(To), float(slope))
        ctx.save_for_backward(x if slope != 1.0 else None)
        ctx.xshape = (N, C, T)
        return col

    @staticmethod
    def backward(ctx, gcol):
        (x,) = ctx.saved_tensors
        k, dil, pad, stride, up, To, slope = ctx.geom
        N, C, T = ctx.xshape
        gcol = gcol.contiguous()
        gx = torch.empty((N, C, T), dtype=torch.float32, device=gcol.device)
        with torch.cuda.device(gcol.device):
            check(lib().psnd_col2im_f32(ptr(gcol), ptr(x), N, C, T, k, dil, pad, stride, up, To, slope, ptr(gx), stream_ptr(gcol.device)), 'psnd_col2im_f32')
        return gx, None, None, None, None, None, None, None


def conv1d_f32(x, w, bias, padding=0, dilation=1, pre_slope=1.0):
    """F.conv1d(leaky_relu(x, pre_slope), w, bias, 1, padding, dilation) in exact fp32 on the matrix cores - the precision of the reference's
    conv stack (hifi_gan.py:32-147): unfold (psnd_im2col_f32, the activation applied while gathering) + psnd_linear1x1_* (v_mfma_f32_32x32x2_f32),
    forward and both gradients.  x (N, Cin, T) fp32 HIP, w (Cout, Cin, k)."""
    Cout, Cin, k = w.shape
    T = x.shape[2]
    To = T + 2 * padding - dilation * (k - 1)
    if To <= 0:
        raise PsndError('conv1d_f32: no output samples (T=%d, k=%d, dilation=%d, padding=%d)' % (T, k, dilation, padding))
    if k == 1 and padding == 0 and pre_slope == 1.0:
        return Linear1x1.apply(x, w, bias, False, False)
    col = Im2Col.apply(x, k, dilation, padding, 1, 1, To, pre_slope)
    return Linear1x1.apply(col, w.reshape(Cout, Cin * k), bias, False, False)


def conv_transpose1d_f32(x, w, bias, stride, padding, pre_slope=1.0):
    """F.conv_transpose1d(leaky_relu(x, pre_slope), w, bias, stride, padding) in exact fp32: a convolution with the flipped taps over the
    zero-spread input (the zeros are never stored: psnd_im2col_f32 writes them into the unfolded rows), (T - 1) * stride - 2 * padding + k
    output samples (hifi_gan.py:107-110).  w (Cin, Cout, k) as nn.ConvTranspose1d keeps it."""
    Cin, Cout, k = w.shape
    T = x.shape[2]
    To = (T - 1) * stride - 2 * padding + k
    if k - 1 - padding < 0 or To <= 0:
        raise PsndError('conv_transpose1d_f32: padding %d beyond k - 1 = %d' % (padding, k - 1))
    weq = w.flip(2).permute(1, 0, 2).reshape(Cout, Cin * k)
    col = Im2Col.apply(x, k, 1, k - 1 - padding, 1, stride, To, pre_slope)
    return Linear1x1.apply(col, weq, bias, False, False)


class PosEnc(torch.autograd.Function):
    """PositionalEncoding.forward (modules.py:143-145): x * sqrt(C) + pe[..., :T] as one pass (psnd_posenc); backward g * sqrt(C)."""

    @staticmethod
    def forward(ctx, x, pe, scale):
        _need_cuda(x, 'input')
        x = x.contiguous()
        N, C, T = x.shape
        pe2 = pe.reshape(pe.shape[-2], pe.shape[-1])
        if pe2.shape[0] != C or pe2.shape[1] < T or not pe2.is_contiguous() or pe2.dtype != torch.float32 or pe2.device != x.device:
            raise PsndError('PosEnc: table %s does not cover a (%d, %d, %d) input' % (tuple(pe.shape), N, C, T))
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib().psnd_posenc(ptr(x), ptr(pe2), float(scale), N, C, T, pe2.shape[1], ptr(y), stream_ptr(x.device)), 'psnd_posenc')
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, C, T = g.shape
        gx = torch.empty_like(g)
        with torch.cuda.device(g.device):
            check(lib().psnd_posenc(ptr(g), None, ctx.scale, N, C, T, T, ptr(gx), stream_ptr(g.device)), 'psnd_posenc')
        return gx, None, None


# measured, off: the two attention gradient kernels on two streams - config-4 block 2.24 -> 2.36 ms (both kernels fill the chip at 2-3 waves
# per SIMD: next to each other each runs slower than the overlap returns; tools/r04/ab_c4.sh)
ATTN_BWD_TWO_STREAMS = _sw.lab('PSND_ATTN_BWD_TWO_STREAMS', '0') == '1'


class AttentionKVQ(torch.autograd.Function):
    """MultiHeadAttention.scale_dot_att over all heads at once (modules.py:38-48, 61-79), straight from the fused projection:
    kvq (N, 3C, T) in the reference's chunk order K | V | Q, heads folded head-major -> out (N, C, T) (heads unfolded, ready for
    the output projection) and att (H*N, T_key, T_query) when `want_att`.  psnd_mha_fwd / psnd_mha_bwd: the score tensor stays
    on the chip; exact fp32 products, or (`bf16`, what the module passes under torch.autocast(bfloat16)) bf16 operands with fp32
    scores / statistics / accumulation.  mask_u8: (N, T), 1 = padding."""

    @staticmethod
    def forward(ctx, kvq, mask_u8, heads, want_att, bf16=False):
        # kvq.dtype == bfloat16 (with bf16 operands, without the attention tensor): the projection stored it that way (Linear1x1 out_h, round 6)
        # - the kernels round the operands to bf16 when they load them anyway - and the gradient goes back the same way
        kvq_h = kvq.dtype == torch.bfloat16
        if kvq_h and (not bf16 or want_att):
            raise PsndError('AttentionKVQ: a bf16-stored kvq comes with bf16 operands (torch.autocast) and without the attention tensor')
        if not (kvq_h and isinstance(kvq, torch.Tensor) and kvq.is_cuda):
            _need_cuda(kvq, 'kvq')
        kvq = kvq.contiguous()
        N, C3, T = kvq.shape
        C = C3 // 3
        dev = kvq.device
        out = torch.empty((N, C, T), dtype=torch.bfloat16 if kvq_h else torch.float32, device=dev)      # (its consumer, the output projection, rounds it to bf16 anyway)
        att = torch.empty((heads * N, T, T), dtype=torch.float32, device=dev) if want_att else None
        stats = torch.empty((heads * N, T, 2), dtype=torch.float32, device=dev)
        m = None if mask_u8 is None else mask_u8.contiguous()
        with torch.cuda.device(dev):
            check(lib().psnd_mha_fwd(ptr(kvq), ptr(m), N, heads, C, T, ptr(out), ptr(att), ptr(stats), 2 if kvq_h else int(bool(bf16)),
                                     stream_ptr(dev)), 'psnd_mha_fwd')
        ctx.heads, ctx.bf16 = heads, (2 if kvq_h else int(bool(bf16)))
        ctx.set_materialize_grads(False)             # an unused `att` must not turn into an (H*N, T, T) tensor of zeros in backward
        ctx.save_for_backward(kvq, m, out, att, stats)
        if att is None:
            att = out.new_empty(0)
            ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, gout, gatt):
        kvq, m, out, att, stats = ctx.saved_tensors
        N, C3, T = kvq.shape
        C, H = C3 // 3, ctx.heads
        dev = kvq.device
        if gout is None:
            gout = torch.zeros_like(out)
        gout = gout.contiguous()
        if gatt is not None and att is None:
            gatt = None
        if gatt is not None:
            gatt = gatt.contiguous()
        if gout.dtype != out.dtype:
            gout = gout.to(out.dtype)
        delta = torch.empty((H * N, T), dtype=torch.float32, device=dev)
        gkvq = torch.empty_like(kvq)
        from . import cl
        args = (ptr(kvq), ptr(m), ptr(out), ptr(att), ptr(stats), ptr(gout), ptr(gatt), N, H, C, T, ptr(delta), ptr(gkvq), ctx.bf16)
        with torch.cuda.device(dev):
            if ATTN_BWD_TWO_STREAMS and cl.AUTO_SECTIONS and ctx.bf16 != 2:      # (bf16 = 2 runs as a whole: its query kernel forms delta)
                # the key / value and the query gradient kernels need `delta` only and write disjoint rows of gkvq: two streams (inside
                # the step graph: two branches) behind the delta launch, joined before the projection's backward reads gkvq
                main, side = torch.cuda.current_stream(dev), cl.branch_streams(dev, 1)[0]
                check(lib().psnd_mha_bwd_parts(*args, 1, stream_ptr(dev)), 'psnd_mha_bwd')
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    check(lib().psnd_mha_bwd_parts(*args, 4, stream_ptr(dev)), 'psnd_mha_bwd')
                check(lib().psnd_mha_bwd_parts(*args, 2, stream_ptr(dev)), 'psnd_mha_bwd')
                main.wait_stream(side)
            else:
                check(lib().psnd_mha_bwd(*args, stream_ptr(dev)), 'psnd_mha_bwd')
        return gkvq, None, None, None, None


class PreEmphasisFn(torch.autograd.Function):
    """y[t] = x[t] - coef * x[t-1] with one reflect-padded sample on the left (sound.py:76-81)."""

    @staticmethod
    def forward(ctx, x, coef):
        _need_cuda(x, 'input')
        x = x.contiguous()
        T = x.shape[-1]
        N = x.numel() // max(T, 1)
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib().psnd_preemphasis_fwd(ptr(x), N, T, float(coef), ptr(y), stream_ptr(x.device)), 'psnd_preemphasis_fwd')
        ctx.coef = float(coef)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        T = gy.shape[-1]
        N = gy.numel() // max(T, 1)
        gx = torch.empty_like(gy)
        with torch.cuda.device(gy.device):
            check(lib().psnd_preemphasis_bwd(ptr(gy), N, T, ctx.coef, ptr(gx), stream_ptr(gy.device)), 'psnd_preemphasis_bwd')
        return gx, None


def _msl_fused(n_fft, hop):
    """psnd_stft_bwd_msl takes this resolution (PSND_MSL_FUSED=0: the two-launch path, for A/B runs and tests)."""
    import os
    return _sw.lab('PSND_MSL_FUSED', '1') != '0' and bool(lib().psnd_stft_bwd_msl_supported(int(n_fft), int(hop)))


class MultiStftLossFn(torch.autograd.Function):
    """(loss, sc_loss, mag_loss) of sound.py:106-133 as ONE autograd node: per resolution two psnd_stft_fwd launches
    (pred, target) and one partial-sum pass, one combining launch for all resolutions; backward: per resolution one
    gradient pass over the magnitudes and psnd_stft_bwd, the waveform gradients of the resolutions added up.
    `cfgs`: tuple of (n_fft, hop) per resolution, `plans`: their device plans (window centre-padded to n_fft)."""

    @staticmethod
    def forward(ctx, pred, target, eps, cfgs, *plans):
        import ctypes
        _need_cuda(pred, 'pred')
        _need_cuda(target, 'target')
        if pred.shape != target.shape or pred.dim() != 2:
            raise _lib.PsndError('multi_stft_loss: pred %s and target %s must be equal (N, T) shapes'
                                 % (tuple(pred.shape), tuple(target.shape)))
        pred, target = pred.contiguous(), target.contiguous()
        N = pred.shape[0]
        L = len(cfgs)
        dev = pred.device
        mags, parts, kfs, blocks = [], [], [], []
        # training case (only the prediction needs a gradient): the prediction's magnitudes never reach HBM - psnd_stft_fwd_msl leaves
        # the three sums, psnd_stft_bwd_msl recomputes |X| in the backward
        only_pred = not ctx.needs_input_grad[1]
        with torch.cuda.device(dev):
            s = stream_ptr(dev)
            for (n_fft, hop), plan in zip(cfgs, plans):
                t = stft_forward(target, n_fft, hop, plan)['mag']
                KF = t.shape[1] * t.shape[2]
                B = int(lib().psnd_stft_fwd_msl_blocks(pred.shape[1], int(n_fft), int(hop))) if only_pred and _msl_fused(n_fft, hop) else 0
                if B > 0:
                    p = None
                    part = torch.empty((N, B, 3), dtype=torch.float64, device=dev)
                    check(lib().psnd_stft_fwd_msl(ptr(pred), N, pred.shape[1], int(n_fft), int(hop), ptr(plan), 0.0, ptr(t), float(eps),
                                                  ptr(part), s), 'psnd_stft_fwd_msl')
                else:
                    p = stft_forward(pred, n_fft, hop, plan)['mag']
                    B = int(lib().psnd_stft_loss_blocks(KF))
                    part = torch.empty((N, B, 3), dtype=torch.float64, device=dev)
                    check(lib().psnd_stft_loss_partial(ptr(p), ptr(t), N, KF, float(eps), ptr(part), s), 'psnd_stft_loss_partial')
                mags.append((p, t))
                parts.append(part)
                kfs.append(KF)
                blocks.append(B)
            norms = torch.empty((L, N, 2), dtype=torch.float32, device=dev)
            out = torch.empty(3, dtype=torch.float32, device=dev)
            parr = (ctypes.c_void_p * L)(*[ptr(q) for q in parts])
            karr = (ctypes.c_int64 * L)(*kfs)
            barr = (ctypes.c_int64 * L)(*blocks)
            check(lib().psnd_stft_loss_final_blocks(parr, karr, barr, L, N, ptr(norms), ptr(out), s), 'psnd_stft_loss_final_blocks')
        ctx.cfgs, ctx.eps, ctx.kfs = cfgs, float(eps), kfs
        ctx.has_p = [pt[0] is not None for pt in mags]
        ctx.save_for_backward(pred, target, norms, *[m if m is not None else norms for pt in mags for m in pt], *plans)
        return out

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        L = len(ctx.cfgs)
        pred, target, norms = saved[:3]
        mags, plans = saved[3:3 + 2 * L], saved[3 + 2 * L:]
        N = pred.shape[0]
        need_p, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g = g.contiguous().float()
        gpred = gtarget = None
        with torch.cuda.device(pred.device):
            s = stream_ptr(pred.device)
            for i, (n_fft, hop) in enumerate(ctx.cfgs):
                p, t = mags[2 * i], mags[2 * i + 1]
                if not ctx.has_p[i] or (need_p and not need_t and _msl_fused(n_fft, hop)):
                    # one launch: the adjoint STFT recomputes |X| and forms the loss gradient in registers (psnd_stft_bwd_msl)
                    # ... and adds it to the gradient of the resolutions before it (no memset, no add launch)
                    acc = gpred is not None
                    if not acc:
                        gpred = torch.empty_like(pred)
                    check(lib().psnd_stft_bwd_msl(ptr(pred), N, pred.shape[1], n_fft, hop, FRAMING_CENTER, ptr(plans[i]), 0.0, ptr(t),
                                                  ptr(norms[i]), ptr(g), L, ctx.eps, int(acc), ptr(gpred), s), 'psnd_stft_bwd_msl')
                    continue
                gp = torch.empty_like(p) if need_p else None
                gt = torch.empty_like(t) if need_t else None
                check(lib().psnd_stft_loss_bwd(ptr(p), ptr(t), N, ctx.kfs[i], ctx.eps, ptr(norms[i]), ptr(g), L, ptr(gp), ptr(gt), s),
                      'psnd_stft_loss_bwd')
                if need_p:
                    gw = stft_backward(pred, n_fft, hop, plans[i], gmag=gp)
                    gpred = gw if gpred is None else gpred.add_(gw)
                if need_t:
                    gw = stft_backward(target, n_fft, hop, plans[i], gmag=gt)
                    gtarget = gw if gtarget is None else gtarget.add_(gw)
        return (gpred, gtarget, None, None) + (None,) * L


# ---------------------------------------------------------------------------------------------
# PQMF (models/transforms.py:492-560): polyphase analysis / synthesis, each the other's adjoint with reversed taps
# ---------------------------------------------------------------------------------------------
def _pqmf(op, x, filt, subbands, taps, flip, scale, t_out=None):
    _need_cuda(x, 'input')
    x, filt = x.contiguous(), filt.contiguous()
    dev = x.device
    with torch.cuda.device(dev):
        if op == 'analysis':                        # (B, T) -> (B, S, T // S)
            B, T = x.shape
            out = torch.empty((B, subbands, T // subbands), dtype=torch.float32, device=dev)
            check(lib().psnd_pqmf_analysis(ptr(x), ptr(filt), B, T, subbands, taps, int(flip), float(scale), ptr(out), stream_ptr(dev)),
                  'psnd_pqmf_analysis')
        else:                                       # (B, S, M) -> (B, M * S)
            B, S, M = x.shape
            t_out = M * S if t_out is None else t_out
            out = torch.empty((B, t_out), dtype=torch.float32, device=dev)
            check(lib().psnd_pqmf_synthesis(ptr(x), ptr(filt), B, M, t_out, subbands, taps, int(flip), float(scale), ptr(out), stream_ptr(dev)),
                  'psnd_pqmf_synthesis')
    return out


class PqmfAnalysis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, filt, subbands, taps):
        ctx.save_for_backward(filt)
        ctx.cfg = (subbands, taps, x.shape[1])
        return _pqmf('analysis', x, filt, subbands, taps, 0, 1.0)

    @staticmethod
    def backward(ctx, g):
        (filt,) = ctx.saved_tensors
        S, taps, T = ctx.cfg
        return _pqmf('synthesis', g.contiguous(), filt, S, taps, 1, 1.0, t_out=T), None, None, None   # tail samples reach the last frames


class PqmfSynthesis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, filt, subbands, taps):
        ctx.save_for_backward(filt)
        ctx.cfg = (subbands, taps)
        return _pqmf('synthesis', x, filt, subbands, taps, 0, float(subbands))

    @staticmethod
    def backward(ctx, g):
        (filt,) = ctx.saved_tensors
        S, taps = ctx.cfg
        return _pqmf('analysis', g.contiguous(), filt, S, taps, 1, float(S)), None, None, None


class L1Loss(torch.autograd.Function):
    """F.l1_loss(a, b) (mean): one pass + a tiny combine forward, one pass backward (torch: sub, abs, mean / sign, scale, neg)."""

    @staticmethod
    def forward(ctx, a, b):
        _need_cuda(a, 'input')
        _need_cuda(b, 'target')
        if a.shape != b.shape:
            raise _lib.PsndError('l1_loss: shapes %s and %s differ' % (tuple(a.shape), tuple(b.shape)))
        a, b = a.contiguous(), b.contiguous()
        n = a.numel()
        part = torch.empty(int(lib().psnd_l1_loss_blocks(n)), dtype=torch.float64, device=a.device)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            check(lib().psnd_l1_loss_fwd(ptr(a), ptr(b), n, ptr(part), ptr(out), stream_ptr(a.device)), 'psnd_l1_loss_fwd')
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        g = g.contiguous().float()
        with torch.cuda.device(a.device):
            check(lib().psnd_l1_loss_bwd(ptr(a), ptr(b), a.numel(), ptr(g), ptr(ga), ptr(gb), stream_ptr(a.device)), 'psnd_l1_loss_bwd')
        return ga, gb


class L1LossSum(torch.autograd.Function):
    """sum_i w_i * F.l1_loss(a_i, b_i) as ONE scalar node (psnd_l1_loss_sum_fwd: one pass per term + one combine; backward one pass per
    term with the weight folded in) - a recipe's `l1(a, b) + 0.5 * l1(c, d)` without the scalar multiply / add launches."""

    @staticmethod
    def forward(ctx, weights, *tensors):
        import ctypes
        k = len(weights)
        if k < 1 or k > 4 or len(tensors) != 2 * k:
            raise _lib.PsndError('l1_loss_sum: 1 .. 4 (input, target) pairs with one weight each')
        ab = []
        for i in range(k):
            a, b = tensors[2 * i], tensors[2 * i + 1]
            _need_cuda(a, 'input')
            _need_cuda(b, 'target')
            if a.shape != b.shape:
                raise _lib.PsndError('l1_loss_sum: shapes %s and %s differ' % (tuple(a.shape), tuple(b.shape)))
            ab += [a.contiguous(), b.contiguous()]
        dev = ab[0].device
        ns = [ab[2 * i].numel() for i in range(k)]
        part = torch.empty(sum(int(lib().psnd_l1_loss_blocks(n)) for n in ns), dtype=torch.float64, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        pa = (ctypes.c_void_p * k)(*[ab[2 * i].data_ptr() for i in range(k)])
        pb = (ctypes.c_void_p * k)(*[ab[2 * i + 1].data_ptr() for i in range(k)])
        pn = (ctypes.c_int64 * k)(*ns)
        pw = (ctypes.c_double * k)(*[float(w) for w in weights])
        with torch.cuda.device(dev):
            check(lib().psnd_l1_loss_sum_fwd(pa, pb, pn, pw, k, ptr(part), ptr(out), stream_ptr(dev)), 'psnd_l1_loss_sum_fwd')
        ctx.weights = tuple(float(w) for w in weights)
        ctx.save_for_backward(*ab)
        return out

    @staticmethod
    def backward(ctx, g):
        ab = ctx.saved_tensors
        g = g.contiguous().float()
        grads = []
        with torch.cuda.device(g.device):
            for i, w in enumerate(ctx.weights):
                a, b = ab[2 * i], ab[2 * i + 1]
                ga = torch.empty_like(a) if ctx.needs_input_grad[1 + 2 * i] else None
                gb = torch.empty_like(b) if ctx.needs_input_grad[2 + 2 * i] else None
                if ga is not None or gb is not None:
                    check(lib().psnd_l1_loss_bwd_w(ptr(a), ptr(b), a.numel(), ptr(g), w, ptr(ga), ptr(gb), stream_ptr(g.device)),
                          'psnd_l1_loss_bwd_w')
                grads += [ga, gb]
        return (None,) + tuple(grads)


class MaskedL1(torch.autograd.Function):
    """sum |a - b| w / (C sum w) over (N, C, T) tensors with a per-frame weight w (N, T): the loss of a padded-batch recipe (clips of
    different lengths padded to the batch maximum, data/dataset.py:196-250) - psnd_masked_l1_fwd / _bwd instead of abs / mul / two sums /
    div and their autograd twins"""

    @staticmethod
    def forward(ctx, a, b, w):
        _need_cuda(a, 'input')
        _need_cuda(b, 'target')
        _need_cuda(w, 'frame weight')
        a, b, w = a.contiguous(), b.contiguous(), w.contiguous()
        N, C, T = a.shape
        if b.shape != a.shape or tuple(w.shape) != (N, T):
            raise _lib.PsndError('masked_l1_loss: input %s, target %s, frame weight %s' % (tuple(a.shape), tuple(b.shape), tuple(w.shape)))
        part = torch.empty(2 * int(lib().psnd_masked_l1_blocks(a.numel())), dtype=torch.float64, device=a.device)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        inv = torch.empty(1, dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            check(lib().psnd_masked_l1_fwd(ptr(a), ptr(b), ptr(w), N, C, T, ptr(part), ptr(out), ptr(inv), stream_ptr(a.device)), 'psnd_masked_l1_fwd')
        ctx.save_for_backward(a, b, w, inv)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, w, inv = ctx.saved_tensors
        N, C, T = a.shape
        g = g.contiguous().float()
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if ga is None and gb is None:
            return None, None, None
        with torch.cuda.device(a.device):
            check(lib().psnd_masked_l1_bwd(ptr(a), ptr(b), ptr(w), N, C, T, ptr(g), ptr(inv), ptr(ga), ptr(gb), stream_ptr(a.device)), 'psnd_masked_l1_bwd')
        return ga, gb, None


def masked_l1_loss(input, target, frame_weight):
    """sum |input - target| * frame_weight[:, None, :] / (C * frame_weight.sum()) on fp32 HIP tensors (N, C, T) / (N, T)"""
    return MaskedL1.apply(input, target, frame_weight)


def l1_loss_sum(pairs, weights):
    """sum_i weights[i] * F.l1_loss(*pairs[i]) on fp32 HIP tensors, one autograd node"""
    flat = [t for p in pairs for t in p]
    return L1LossSum.apply(tuple(weights), *flat)


def l1_loss(input, target):
    """drop-in for torch.nn.functional.l1_loss(input, target) (reduction 'mean') on fp32 HIP tensors"""
    return L1Loss.apply(input, target)
