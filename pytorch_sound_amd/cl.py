"""Channels-last ("CL") execution of the Conv1d stacks on the gfx950 implicit-GEMM kernel
(csrc/psnd_conv.hip, psnd_conv1d_cl).

A CL activation is a bf16 tensor (N, Lp, Cp): rows HP..HP+L of every clip hold the data, the other rows
are zero (they ARE the conv zero padding), channels are zero-padded to a multiple of 32.  Module
parameters stay exactly the reference's (weight_g / weight_v / bias per conv); this file only changes how
``leaky_relu -> conv -> bias -> (+ residual)`` of hifi_gan.py:56-62 / 84-88 is executed:

    fused_conv(xa, conv, res, want_raw, want_act):   y = conv(xa) + bias (+ res);  ya = leaky_relu(y)

entirely on hand-written kernels: weight prep (weight norm + bf16 packs), implicit-GEMM conv (forward and input
gradient), split-K weight gradient with all taps / bias gradient fused, weight-norm backward.
"""
import ctypes
import os
from pytorch_sound_amd import _switches as _sw
import weakref

import torch

from . import _lib
from ._lib import lib, check, ptr, stream_ptr

ALIGN_C = 32


def round_up(x, m):
    return (x + m - 1) // m * m


class CLShape:
    """geometry shared by every CL buffer of one forward pass"""

    def __init__(self, N, L, HP):
        self.N, self.L, self.HP = int(N), int(L), int(HP)
        self.Lp = round_up(self.L + 2 * self.HP, 8)

    @property
    def R(self):
        return self.N * self.Lp


def _need(t, dtype):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise _lib.PsndError('CL kernels need contiguous %s CUDA tensors (got %s, %s, contiguous=%s)'
                             % (dtype, t.device, t.dtype, t.is_contiguous()))


class ToCL(torch.autograd.Function):
    """(N, C, T) fp32 -> CL bf16 (optionally through log1p); backward is the inverse gather."""

    @staticmethod
    def forward(ctx, x, shape, preop):
        x = x.contiguous()
        _need(x, torch.float32)
        N, C, T = x.shape
        Cp = round_up(C, ALIGN_C)
        out = torch.empty((N, shape.Lp, Cp), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            check(lib().psnd_to_cl(ptr(x), N, C, T, shape.Lp, shape.HP, Cp, int(preop), ptr(out), stream_ptr(x.device)),
                  'psnd_to_cl')
        ctx.shape, ctx.C, ctx.T, ctx.preop = shape, C, T, preop
        if preop:
            ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gx = from_cl_raw(g, ctx.C, ctx.T, ctx.shape)
        if ctx.preop == 1:
            (x,) = ctx.saved_tensors
            gx = gx / (1.0 + x)
        return gx, None, None


def from_cl_raw(buf, C, T, shape):
    _need(buf, torch.bfloat16)
    N, Lp, Cp = buf.shape
    out = torch.empty((N, C, T), dtype=torch.float32, device=buf.device)
    with torch.cuda.device(buf.device):
        check(lib().psnd_from_cl(ptr(buf), N, C, T, Lp, shape.HP, Cp, ptr(out), stream_ptr(buf.device)), 'psnd_from_cl')
    return out


class FromCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf, C, T, shape):
        ctx.shape, ctx.Cp = shape, buf.shape[2]
        return from_cl_raw(buf.contiguous(), C, T, shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, C, T = g.shape
        out = torch.empty((N, ctx.shape.Lp, ctx.Cp), dtype=torch.bfloat16, device=g.device)
        with torch.cuda.device(g.device):
            check(lib().psnd_to_cl(ptr(g), N, C, T, ctx.shape.Lp, ctx.shape.HP, ctx.Cp, 0, ptr(out), stream_ptr(g.device)),
                  'psnd_to_cl')
        return out, None, None, None


class FromCLTanh(torch.autograd.Function):
    """tanh(from_cl(buf)): the generator's output non-linearity (hifi_gan.py:134-135) inside the layout change (psnd_from_cl_tanh); backward
    to_cl(g * (1 - out^2)) in one pass (psnd_to_cl_tanh_bwd) - no library tanh / tanh_backward launches on a HIP tensor"""

    @staticmethod
    def forward(ctx, buf, C, T, shape):
        buf = buf.contiguous()
        _need(buf, torch.bfloat16)
        N, Lp, Cp = buf.shape
        out = torch.empty((N, C, T), dtype=torch.float32, device=buf.device)
        with torch.cuda.device(buf.device):
            check(lib().psnd_from_cl_tanh(ptr(buf), N, C, T, Lp, shape.HP, Cp, ptr(out), stream_ptr(buf.device)), 'psnd_from_cl_tanh')
        ctx.shape, ctx.Cp = shape, Cp
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = g.contiguous().float()
        N, C, T = g.shape
        gx = torch.empty((N, ctx.shape.Lp, ctx.Cp), dtype=torch.bfloat16, device=g.device)
        with torch.cuda.device(g.device):
            check(lib().psnd_to_cl_tanh_bwd(ptr(g), ptr(out), N, C, T, ctx.shape.Lp, ctx.shape.HP, ctx.Cp, ptr(gx), stream_ptr(g.device)),
                  'psnd_to_cl_tanh_bwd')
        return gx, None, None, None


class MaskHeadCL(torch.autograd.Function):
    """est = sigmoid(from_cl(y)) * mag  - the head of a spectrogram-masking model (one pass instead of layout change + sigmoid +
    multiply; backward one pass instead of three).  mag: (N, C, T) fp32, treated as a constant (no gradient)."""

    @staticmethod
    def forward(ctx, y, mag, shape):
        _need(y, torch.bfloat16)
        _need(mag, torch.float32)
        y, mag = y.contiguous(), mag.contiguous()
        N, C, T = mag.shape
        est = torch.empty_like(mag)
        with torch.cuda.device(y.device):
            check(lib().psnd_mask_head_fwd(ptr(y), ptr(mag), N, C, T, shape.Lp, shape.HP, y.shape[2], ptr(est), stream_ptr(y.device)),
                  'psnd_mask_head_fwd')
        ctx.shape = shape
        ctx.save_for_backward(y, mag)
        return est

    @staticmethod
    def backward(ctx, g):
        y, mag = ctx.saved_tensors
        g = g.contiguous().float()
        N, C, T = mag.shape
        gy = torch.empty_like(y)
        with torch.cuda.device(y.device):
            check(lib().psnd_mask_head_bwd(ptr(g), ptr(mag), ptr(y), N, C, T, ctx.shape.Lp, ctx.shape.HP, y.shape[2], ptr(gy),
                                           stream_ptr(y.device)), 'psnd_mask_head_bwd')
        return gy, None, None


# the NaN flag (`loss != loss`, trainer.py:205) of the fused loss nodes below: written by the launch that forms the loss; the caller that receives
# the loss tensor attaches it as `loss.psnd_nan_flag`, which Trainer._nan_flag takes instead of a launch of its own
LAST_LOSS_NAN_FLAG = [None]
# where that flag should be written when somebody has a place for it: a data-parallel reducer registers the spare slot behind its last gradient
# bucket (distributed.FlatGradReducer: the flag rides along with the gradients) - FlatGradReducer.set_flag then has nothing to copy
NAN_FLAG_DEST = [None]


def _nan_flag_tensor(dev):
    dest = NAN_FLAG_DEST[0]
    if dest is not None and dest.device == dev and dest.dtype == torch.float32 and dest.numel() == 1:
        return dest.view(())
    return torch.empty((), dtype=torch.float32, device=dev)


class MaskHeadSpectralL1CL(torch.autograd.Function):
    """w1 * F.l1_loss(est, mag_ref) + w2 * F.l1_loss(log_mel(est), mel_ref) with est = sigmoid(from_cl(y)) * mag - a masking recipe's
    loss as ONE autograd node over four launches forward (mask head with the first term's partial sums, mel projection with the second
    term's, one combine) and two backward (the mel adjoint forms sign(log_mel - ref) on operand load; the mask head's backward adds
    sign(est - ref) itself).  The separate formulation (MaskHeadCL + MelLog + l1_loss_sum) takes five launches forward and five
    backward and materialises log_mel(est), two gradient tensors and their sum.  Returns (loss, est); est carries no gradient of its own
    (it is there for logging / metrics)."""

    @staticmethod
    def forward(ctx, y, mag, mag_ref, mel_ref, mel_plan, shape, M, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, w1, w2):
        import ctypes
        from .kernels import _clamp_args
        ctx.set_materialize_grads(False)             # `est` (no gradient of its own) must not cost a zero fill of its size in backward
        _need(y, torch.bfloat16)
        for t in (mag, mag_ref, mel_ref):
            _need(t, torch.float32)
        N, C, T = mag.shape
        if mag_ref.shape != mag.shape or tuple(mel_ref.shape) != (N, M, T):
            raise _lib.PsndError('mask_head_spectral_l1: mag_ref %s / mel_ref %s do not match mag %s and %d mel bands'
                                 % (tuple(mag_ref.shape), tuple(mel_ref.shape), tuple(mag.shape), M))
        dev = y.device
        lo, hi, pre = _clamp_args(clamp_lo, clamp_hi, pre_clamp_min)
        est = torch.empty_like(mag)
        lin = torch.empty((N, M, T), dtype=torch.float32, device=dev)
        nb1 = int(lib().psnd_mask_head_l1_blocks(N, T, y.shape[2]))
        nb2 = int(lib().psnd_mel_l1_blocks(N, T, M))
        part = torch.empty(nb1 + nb2, dtype=torch.float64, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        st = stream_ptr(dev)
        with torch.cuda.device(dev):
            check(lib().psnd_mask_head_l1_fwd(ptr(y), ptr(mag), ptr(mag_ref), N, C, T, shape.Lp, shape.HP, y.shape[2], ptr(est),
                                              ptr(part), st), 'psnd_mask_head_l1_fwd')
            p2 = ctypes.c_void_p(part.data_ptr() + 8 * nb1)
            check(lib().psnd_mel_l1_fwd(ptr(est), N, T, M, C, ptr(mel_plan), log_kind, float(log_offset), pre, lo, hi, ptr(mel_ref),
                                        ptr(lin), p2, st), 'psnd_mel_l1_fwd')
            parts = (ctypes.c_void_p * 2)(part.data_ptr(), part.data_ptr() + 8 * nb1)
            nbs = (ctypes.c_int64 * 2)(nb1, nb2)
            sc = (ctypes.c_double * 2)(float(w1) / mag.numel(), float(w2) / mel_ref.numel())
            nan_flag = _nan_flag_tensor(dev)
            check(lib().psnd_l1_loss_combine(parts, nbs, sc, 2, ptr(out), ptr(nan_flag), st), 'psnd_l1_loss_combine')
        LAST_LOSS_NAN_FLAG[0] = nan_flag
        ctx.cfg = (shape, M, log_kind, float(log_offset), pre, lo, hi, float(w1) / mag.numel(), float(w2) / mel_ref.numel())
        ctx.save_for_backward(y, mag, mag_ref, mel_ref, mel_plan, est, lin)
        ctx.mark_non_differentiable(est)
        return out, est

    @staticmethod
    def backward(ctx, g, _gest):
        y, mag, mag_ref, mel_ref, mel_plan, est, lin = ctx.saved_tensors
        shape, M, log_kind, log_offset, pre, lo, hi, c1, c2 = ctx.cfg
        N, C, T = mag.shape
        if g is None:
            return (None,) * 14
        g = g.contiguous().float()
        gest = torch.empty_like(mag)
        gy = torch.empty_like(y)
        st = stream_ptr(y.device)
        with torch.cuda.device(y.device):
            check(lib().psnd_mel_l1_bwd(ptr(mel_ref), ptr(lin), ptr(g), c2, N, T, M, C, ptr(mel_plan), log_kind, log_offset, pre, lo, hi,
                                        ptr(gest), st), 'psnd_mel_l1_bwd')
            check(lib().psnd_mask_head_l1_bwd(ptr(gest), ptr(mag), ptr(y), ptr(est), ptr(mag_ref), ptr(g), c1, N, C, T, shape.Lp, shape.HP,
                                              y.shape[2], ptr(gy), st), 'psnd_mask_head_l1_bwd')
        return (gy,) + (None,) * 13


def to_cl_nfk(x_nfk, shape, preop=0):
    """(N, F, K) fp32 (bin-fastest: kernels.stft_mag_nfk) -> CL bf16, optionally through log1p - a plain stream, (N, F, K) being the
    channels-last order already (psnd_to_cl_nfk).  No gradient: the features are inputs."""
    _need(x_nfk, torch.float32)
    if x_nfk.requires_grad and torch.is_grad_enabled():
        raise _lib.PsndError('to_cl_nfk carries no gradient (feature tensors are inputs); use ToCL on an (N, K, F) tensor')
    N, F, K = x_nfk.shape
    if F != shape.L or N != shape.N:
        raise _lib.PsndError('to_cl_nfk: tensor %s does not match the CL geometry (N=%d, L=%d)' % (tuple(x_nfk.shape), shape.N, shape.L))
    Cp = round_up(K, ALIGN_C)
    out = torch.empty((N, shape.Lp, Cp), dtype=torch.bfloat16, device=x_nfk.device)
    with torch.cuda.device(x_nfk.device):
        check(lib().psnd_to_cl_nfk(ptr(x_nfk), N, K, F, shape.Lp, shape.HP, Cp, int(preop), ptr(out), stream_ptr(x_nfk.device)),
              'psnd_to_cl_nfk')
    return out


class MaskHeadSpectralL1NFK(torch.autograd.Function):
    """MaskHeadSpectralL1CL with every spectrogram-sized tensor BIN-FASTEST, (N, F, K): mag / mag_ref / est are (N, F, K) (mel_ref stays
    (N, M, F)).  The mask head and its backward are plain streams over the channels-last rows (no 32 x 32 transposes through LDS), the mel
    kernels read / write whole 16-byte pieces of a frame's spectrum.  Same value, same gradient."""

    @staticmethod
    def forward(ctx, y, mag, mag_ref, mel_ref, mel_plan, shape, M, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, w1, w2):
        from .kernels import _clamp_args
        ctx.set_materialize_grads(False)
        _need(y, torch.bfloat16)
        for t in (mag, mag_ref, mel_ref):
            _need(t, torch.float32)
        N, F, K = mag.shape
        if mag_ref.shape != mag.shape or tuple(mel_ref.shape) != (N, M, F):
            raise _lib.PsndError('mask_head_spectral_l1 (nfk): mag_ref %s / mel_ref %s do not match mag %s and %d mel bands'
                                 % (tuple(mag_ref.shape), tuple(mel_ref.shape), tuple(mag.shape), M))
        dev = y.device
        lo, hi, pre = _clamp_args(clamp_lo, clamp_hi, pre_clamp_min)
        est = torch.empty_like(mag)
        lin = torch.empty((N, M, F), dtype=torch.float32, device=dev)
        nb1 = int(lib().psnd_mask_head_l1_blocks_nfk(N, F, K))
        nb2 = int(lib().psnd_mel_l1_blocks(N, F, M))
        part = torch.empty(nb1 + nb2, dtype=torch.float64, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        st = stream_ptr(dev)
        with torch.cuda.device(dev):
            check(lib().psnd_mask_head_l1_fwd_nfk(ptr(y), ptr(mag), ptr(mag_ref), N, K, F, shape.Lp, shape.HP, y.shape[2], ptr(est),
                                                  ptr(part), st), 'psnd_mask_head_l1_fwd_nfk')
            p2 = ctypes.c_void_p(part.data_ptr() + 8 * nb1)
            check(lib().psnd_mel_l1_fwd_nfk(ptr(est), N, F, M, K, ptr(mel_plan), log_kind, float(log_offset), pre, lo, hi, ptr(mel_ref),
                                            ptr(lin), p2, st), 'psnd_mel_l1_fwd_nfk')
            parts = (ctypes.c_void_p * 2)(part.data_ptr(), part.data_ptr() + 8 * nb1)
            nbs = (ctypes.c_int64 * 2)(nb1, nb2)
            sc = (ctypes.c_double * 2)(float(w1) / mag.numel(), float(w2) / mel_ref.numel())
            nan_flag = _nan_flag_tensor(dev)
            check(lib().psnd_l1_loss_combine(parts, nbs, sc, 2, ptr(out), ptr(nan_flag), st), 'psnd_l1_loss_combine')
        LAST_LOSS_NAN_FLAG[0] = nan_flag
        ctx.cfg = (shape, M, log_kind, float(log_offset), pre, lo, hi, float(w1) / mag.numel(), float(w2) / mel_ref.numel())
        ctx.save_for_backward(y, mag, mag_ref, mel_ref, mel_plan, est, lin)
        ctx.mark_non_differentiable(est)
        return out, est

    @staticmethod
    def backward(ctx, g, _gest):
        y, mag, mag_ref, mel_ref, mel_plan, est, lin = ctx.saved_tensors
        shape, M, log_kind, log_offset, pre, lo, hi, c1, c2 = ctx.cfg
        N, F, K = mag.shape
        if g is None:
            return (None,) * 14
        g = g.contiguous().float()
        gest = torch.empty_like(mag)
        gy = torch.empty_like(y)
        st = stream_ptr(y.device)
        with torch.cuda.device(y.device):
            check(lib().psnd_mel_l1_bwd_nfk(ptr(mel_ref), ptr(lin), ptr(g), c2, N, F, M, K, ptr(mel_plan), log_kind, log_offset, pre, lo, hi,
                                            ptr(gest), st), 'psnd_mel_l1_bwd_nfk')
            check(lib().psnd_mask_head_l1_bwd_nfk(ptr(gest), ptr(mag), ptr(y), ptr(est), ptr(mag_ref), ptr(g), c1, N, K, F, shape.Lp,
                                                  shape.HP, y.shape[2], ptr(gy), st), 'psnd_mask_head_l1_bwd_nfk')
        return (gy,) + (None,) * 13


def _launch_conv(A, A2, AM, a2_slope, W, bias, res, mask_src, shape, Ca, Cb, k, off0, dstep, act_slope, mask_slope,
                 want_raw, want_act, a_eff_out=None):
    dev = W.device
    raw = torch.empty((shape.N, shape.Lp, Cb), dtype=torch.bfloat16, device=dev) if want_raw else None
    act = torch.empty((shape.N, shape.Lp, Cb), dtype=torch.bfloat16, device=dev) if want_act else None
    with torch.cuda.device(dev):
        check(lib().psnd_conv1d_cl(ptr(A), ptr(A2), ptr(AM), float(a2_slope), ptr(W), ptr(bias), ptr(res), ptr(mask_src),
                                   shape.N, shape.Lp, shape.L, shape.HP, Ca, Cb, k, off0, dstep, float(act_slope),
                                   float(mask_slope), ptr(raw), ptr(act), ptr(a_eff_out), stream_ptr(dev)), 'psnd_conv1d_cl')
    return raw, act


_JOIN_PENDING = {}


def _join_side_at_end_of_backward(dev, side):
    """the stream this is called on (the running node's stream = the stream of its forward, in the flows of this library the stream
    `backward()` was called on) waits for `side` when the running backward pass ends.
    The target is taken HERE, not in the callback: autograd runs its final callbacks on whichever thread finishes the pass - possibly
    the device's worker thread, whose current stream is the default stream, not the caller's (inside a capture that would pull the
    legacy stream into the capture instead of joining the branch)."""
    target = torch.cuda.current_stream(dev)
    pending = _JOIN_PENDING.setdefault(dev.index, [])
    if not any(t == target and sd == side for t, sd in pending):
        pending.append((target, side))
    # (a callback per call, not per pass: the first one to run joins everything registered so far, the others find nothing - a pass that
    #  died half-way cannot leave a stale "already queued" mark behind)

    def join():
        for t, sd in _JOIN_PENDING.pop(dev.index, []):
            t.wait_stream(sd)

    torch.autograd.Variable._execution_engine.queue_callback(join)

# ---- batch sections on parallel streams --------------------------------------------------------------------------------------
# At the config-2 size every conv launch is a latency chain: ~350 workgroups that all start together, wait for their
# first stage together, pull their weight fragments through the 64 B/clk vector-memory path together and run their epilogues together.
# The clips of a batch are independent, so a conv chain can be walked as PSND_CL_SECTIONS independent chains over contiguous groups of
# clips (views of the same CL buffers: no copies), each on its own stream: the chains drift apart and one section's load phase
# overlaps another's wait / epilogue.  Inside a hipGraph capture the sections are parallel branches (ONE fork and ONE join per chain
# and direction; a fork / join per conv - round 2's weight-gradient side stream - lost 3 % against one stream).
_SECTION_STREAMS = {}
# Extra streams inside the step graph (batch sections when asked for, the split backward's weight-gradient streams) are only used while
# nothing else shares the hardware queues with the step: HIP multiplexes streams onto a few hardware queues, and a host->device prefetch
# stream that lands on the queue of such a graph branch waits behind its kernels (measured with two sections: prefetch_copy step 1.21 ->
# 2.9 ms in one run of three).  The Trainer clears the flag when it starts a prefetch stream or a gradient reducer.
AUTO_SECTIONS = True
# 'body': a conv chain (the separator's conv_pre + blocks + conv_post) is ONE autograd node with ONE weight-norm backward launch at its
# end - fewest launches, but every parameter gradient then appears at the very end of the backward.  'block': one node (and one
# weight-norm backward) per residual block - the gradients arrive block by block, which is what lets the gradient all-reduce of a
# data-parallel run overlap the backward (Trainer switches to it when a FlatGradReducer over more than one rank is active).
NODE_GRANULARITY = 'body'

# Data-parallel gradient hand-over from INSIDE a conv-chain node's backward (distributed.FlatGradReducer registers itself here while a
# data-parallel step runs): with a sink, the weight gradients of a chain are produced in CHUNKS - behind every input-gradient launch
# of the chain (one per ResBlock1) the weight gradients of the convs that launch completed, their weight-norm backward, and the
# hand-over of those parameters - so the all-reduce of the late blocks' buckets runs under the backward of the early blocks.  The
# weight-norm backward writes (g_v, g_g, g_bias) straight into the reducer's flat buckets (`dest`), no copy.
#   sink.dest(param, numel) -> fp32 tensor of `numel` elements to write the gradient into, or None (not this parameter / already
#                              delivered this step: the node then returns the gradient and autograd accumulates it)
#   sink.deliver(params)    -> these parameters' gradients are in place (enqueued on the current stream)
GRAD_SINK = None

# ---- how many autograd nodes will produce a gradient for a parameter in the backward pass that is running ---------------------------------
# A node may hand a parameter's gradient over early (GRAD_SINK.dest / deliver) or produce it on a side stream (BRANCH_PARAM_GRADS) only when
# it is the parameter's ONLY producer in this backward: with a second node (a module applied twice, shared weights) autograd accumulates
# the two contributions - into a slot whose bucket may already be on its way through the all-reduce, or on the main stream with no
# dependency on the side stream.  The count is kept PER AUTOGRAD GRAPH, not per process: every node that uses one
# of these shortcuts notes its parameters in forward; a note lives as long as its node does (weak reference: a graph that is dropped takes
# its notes with it) and until the node's backward has run (`consume_param_use`).  Two forwards before one backward therefore count two
# however many unrelated backward passes run in between; forward / backward / forward (a GAN step) counts one each time.
#   * a parameter seen with more than one live note during a backward stays multi-use for the rest of THAT backward (graph-task id), so the
#     last of its nodes - whose siblings have been consumed by then - does not see itself alone;
#   * a note that arrives WHILE a backward is running (checkpoint recomputation: the same module may be recomputed once per checkpoint
#     segment, one at a time) marks the parameter multi-use for good - those models keep autograd's own accumulation.
_PARAM_NOTES = {}                # id(parameter) -> {id(node ctx), ...} of live, not yet consumed nodes
_PARAM_MULTI_TASK = {}           # id(parameter) -> graph-task id of the backward that saw it with several producers
_PARAM_FORCED_MULTI = set()      # id(parameter): noted from inside a backward pass


def _graph_task_id():
    return torch._C._current_graph_task_id()


def _drop_notes(cid, pids):
    for pid in pids:
        live = _PARAM_NOTES.get(pid)
        if live is not None:
            live.discard(cid)
            if not live:
                del _PARAM_NOTES[pid]


def note_param_use(ctx, *params):
    """called from an autograd.Function's forward (grad mode is off in there: ctx.needs_input_grad tells a recorded pass from inference)"""
    if not any(ctx.needs_input_grad):
        return
    pids = tuple(id(p) for p in params if p is not None)
    if not pids:
        return
    cid = id(ctx)
    in_backward = _graph_task_id() != -1
    for pid in pids:
        _PARAM_NOTES.setdefault(pid, set()).add(cid)
        if in_backward:
            _PARAM_FORCED_MULTI.add(pid)
    ctx._psnd_noted = pids
    weakref.finalize(ctx, _drop_notes, cid, pids)


def single_use(p) -> bool:
    """true iff the node whose backward is running is the only live producer of a gradient for `p` in this backward pass
    (parameters nobody noted: false)"""
    if p is None:
        return True
    pid = id(p)
    if pid in _PARAM_FORCED_MULTI:
        return False
    task = _graph_task_id()
    if task != -1 and _PARAM_MULTI_TASK.get(pid) == task:
        return False
    n = len(_PARAM_NOTES.get(pid, ()))
    if n > 1 and task != -1:
        _PARAM_MULTI_TASK[pid] = task
    return n == 1


def consume_param_use(ctx):
    """end of a noting node's backward: its parameters' gradients have been produced (a second backward through a retained graph finds
    no note: not single-use, autograd accumulates)"""
    pids = getattr(ctx, '_psnd_noted', None)
    if pids:
        task = _graph_task_id()
        for pid in pids:                                 # siblings of this backward must keep seeing several producers
            if len(_PARAM_NOTES.get(pid, ())) > 1 and task != -1:
                _PARAM_MULTI_TASK[pid] = task
        _drop_notes(id(ctx), pids)


def reset_param_uses():
    _PARAM_NOTES.clear()
    _PARAM_MULTI_TASK.clear()
    _PARAM_FORCED_MULTI.clear()


# the hand-over points: the body's input-gradient launches (one per ResBlock1: 4 at config 2) are cut into this many chunks; every
# chunk costs an extra weight-gradient / weight-norm-backward launch pair and evicts the next input-gradient launch's operands from the
# L2 (38 -> 50 us for that launch), so not every block gets one: 3 chunks release 4 of the config-2 model's 6 buckets early
HANDOVER_CHUNKS = 3


def _sections(dev, N, rows):
    """PSND_CL_SECTIONS = n runs a chain as n batch sections; default 1.  Two sections took the config-2 step from 1.239 to 1.194 ms
    while the weight-gradient role was a 40 k-cycle chain per launch; with the transposing-read weight gradient (26 k cycles) one and
    two sections measure the same (1.014 / 1.017 ms), so the simpler graph is the default."""
    import os
    e = _sw.lab('PSND_CL_SECTIONS', 'auto')
    n = 1 if e == 'auto' else int(e)
    if n <= 1 or N % n != 0 or N // n < 1 or (e == 'auto' and not AUTO_SECTIONS):
        return 1, []
    pool = _SECTION_STREAMS.setdefault(dev.index, [])
    while len(pool) < n - 1:
        pool.append(torch.cuda.Stream(device=dev))
    return n, pool[:n - 1]


def _run_sections(dev, nsec, sides, fn):
    """fn(h) enqueues section h (0 <= h < nsec) on the CURRENT stream; section 0 stays on the caller's stream"""
    if nsec == 1:
        fn(0)
        return
    main = torch.cuda.current_stream(dev)
    for sd in sides:
        sd.wait_stream(main)
    fn(0)
    for h, sd in enumerate(sides, 1):
        with torch.cuda.stream(sd):
            fn(h)
    for sd in sides:
        main.wait_stream(sd)


# the parameter-side launches of a transposed conv's backward on a side stream / graph branch (ConvTransposeCL.backward)
BRANCH_PARAM_GRADS = _sw.lab('PSND_BRANCH_PARAM_GRADS', '1') == '1'
_BRANCH_STREAMS = {}
_PARAM_STREAM = {}


def param_stream(dev):
    """the stream of the parameter-side launches (BRANCH_PARAM_GRADS)"""
    s = _PARAM_STREAM.get(dev.index)
    if s is None:
        s = _PARAM_STREAM[dev.index] = torch.cuda.Stream(device=dev)
    return s



def branch_streams(dev, n):
    """n side streams for independent branches of a model graph (the resblocks of a HiFi-GAN stage)"""
    pool = _BRANCH_STREAMS.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


_WGRAD_STREAMS = {}


def _wgrad_streams(dev):
    """side streams of the split backward (PSND_CL_BWD_SPLIT=1, off by default): the input-gradient chain of a conv stack - one
    psnd_conv1d_cl_pair launch per residual pair - on the caller's stream, the weight gradients (needed only by the optimizer,
    independent of each other) on PSND_CL_WGRAD_STREAMS parallel streams / graph branches.  Measured on the config-2 step (MI355X,
    round 2): 1.22 ms with 2 side streams, 1.34 with 3, 1.26 with 4 against 1.13 ms for the paired input-/weight-gradient launches
    on one stream - the branches overlap for only a fifth of their time (tools/overlap.py) and every kernel runs slower next to
    another one (wgrad 20 -> 26 us, the masked pair launch 22 us): kept as an A/B switch and for the parity tests."""
    import os
    if _sw.lab('PSND_CL_BWD_SPLIT', '0') != '1' or not _pair_enabled() or not AUTO_SECTIONS:
        return []
    n = int(_sw.lab('PSND_CL_WGRAD_STREAMS', '3'))
    pool = _WGRAD_STREAMS.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def _pair_enabled():
    import os
    return _sw.lab('PSND_CL_PAIR', '1') != '0'


def _chain_pairs_bwd(plan, rows):
    """the same for the input-gradient launches ('pairb') of the batched backward: consecutive pairs, each reading the gradient the one
    before wrote, as one masked psnd_conv1d_cl_chain launch"""
    import os
    mx = int(_sw.lab('PSND_CL_CHAIN_BWD', _sw.lab('PSND_CL_CHAIN', '3')))
    if mx < 2 or rows > 8192:
        return plan
    out = []
    for e in plan:
        if e[0] == 'pairb' and e[10] == 256 and e[9] is e[1]:
            c = out[-1] if out and out[-1][0] == 'chainb' else None
            if c is not None and len(c[1]) < mx and c[1][-1][16] is e[1] and c[1][-1][11] == e[11]:
                taps = [t for q_ in c[1] + [e] for t in q_[12:16]]
                if lib().psnd_conv1d_cl_chain_rows(e[10], e[11], len(c[1]) + 1, (ctypes.c_int * len(taps))(*taps)) >= 32:
                    c[1].append(e)
                    continue
            out.append(['chainb', [e]])
        else:
            out.append(e)
    return [c[1][0] if c[0] == 'chainb' and len(c[1]) == 1 else c for c in out]


def _chain_pairs(plan, rows):
    """Consecutive 'pair' launches of a forward plan -> 'chain' launches (psnd_conv1d_cl_chain: a workgroup carries its rows through up
    to PSND_CL_CHAIN pairs - default 3, a ResBlock1 - on the chip; 0 switches it off).  Only where a launch is a latency chain of few
    workgroups (<= 8192 rows: the config-2 size); a pair joins the chain in front of it when it reads that chain's outputs."""
    import os
    mx = int(_sw.lab('PSND_CL_CHAIN', '3'))
    if mx < 2 or rows > 8192:
        return plan
    out = []
    for e in plan:
        if (e[0] == 'pair' and e[9] == 256 and e[8] is not None and e[16] is not None and e[17] is not None
                and 0.0 <= e[4] <= 1.0 and 0.0 <= e[15] <= 1.0):
            c = out[-1] if out and out[-1][0] == 'chain' else None
            if c is not None and len(c[5]) < mx and c[5][-1][17] is e[1] and c[5][-1][16] is e[8] and c[3] == e[9] and c[4] == e[10]:
                taps = [t for q_ in c[5] + [e] for t in q_[11:15]]
                if lib().psnd_conv1d_cl_chain_rows(e[9], e[10], len(c[5]) + 1, (ctypes.c_int * len(taps))(*taps)) >= 32:
                    c[5].append(e)
                    continue
            out.append(['chain', e[1], e[8], e[9], e[10], [e]])
        else:
            out.append(e)
    # a chain of one is the pair launch it came from
    return [c[5][0] if c[0] == 'chain' and len(c[5]) == 1 else c for c in out]


def _sec(t, h, nsec):
    """clips [h N/nsec, (h+1) N/nsec) of a (N, ...) buffer (a contiguous view), None stays None"""
    if t is None or nsec == 1:
        return t
    n = t.shape[0] // nsec
    return t[h * n:(h + 1) * n]


class FusedConvCL(torch.autograd.Function):
    """y = conv1d(xa; weight-normed w, dilation) + bias (+ res); ya = leaky_relu(y, act_slope).
    xa is an ALREADY activated CL buffer.  Returns (y or None, ya or None).

    Launches: forward = weight prep (norm + two bf16 packs) + conv; backward = input-gradient conv (the
    gradient combine g_raw + g_act*leaky'(y) happens while the operand is staged) + weight gradient (all
    taps, bias gradient and the residual's gradient in one kernel) + weight-norm backward."""

    @staticmethod
    def forward(ctx, xa, weight_v, weight_g, bias, res, shape, dil, want_raw, want_act, act_slope, prepped=None, pad=None):
        ctx.set_materialize_grads(False)     # an unused output hands None to backward, not a zero-filled tensor
        _need(xa, torch.bfloat16)
        Cout, Cin, k = weight_v.shape
        Ca, Cb = xa.shape[2], round_up(Cout, ALIGN_C)
        if Ca != round_up(Cin, ALIGN_C):
            raise _lib.PsndError('CL conv: buffer has %d channels, weight expects %d' % (Ca, Cin))
        if pad is None:
            pad = (k * dil - dil) // 2           # "same" convolution; an explicit pad gives taps at -pad .. -pad + (k-1)*dil
        dev = xa.device
        v32, g32 = weight_v.detach().contiguous(), weight_g.detach().contiguous()
        if prepped is not None:                      # packs made by prep_all() for the whole model in one launch
            wf, wb, bp = prepped
        else:
            wf = torch.empty((k, Cb, Ca), dtype=torch.bfloat16, device=dev)      # [j][co][ci]
            wb = torch.empty((k, Ca, Cb), dtype=torch.bfloat16, device=dev)      # [j][ci][co]
            bp = torch.empty(Cb, dtype=torch.float32, device=dev)
            b32 = None if bias is None else bias.detach().contiguous()
            with torch.cuda.device(dev):
                check(lib().psnd_conv1d_prep(ptr(v32), ptr(g32), ptr(b32), Cout, Cin, k, Cb, Ca, ptr(wf), ptr(wb), ptr(bp),
                                             stream_ptr(dev)), 'psnd_conv1d_prep')
        raw, act = _launch_conv(xa, None, None, 1.0, wf, bp, res, None, shape, Ca, Cb, k, -pad, dil, act_slope, 1.0,
                                want_raw, want_act)
        ctx.shape, ctx.dil, ctx.k, ctx.pad, ctx.act_slope = shape, dil, k, pad, act_slope
        ctx.dims = (Cout, Cin, Ca, Cb)
        ctx.has_res, ctx.has_bias = res is not None, bias is not None
        ctx.save_for_backward(xa, v32, g32, wb, act if want_act else None)
        return raw, act

    @staticmethod
    def backward(ctx, g_raw, g_act):
        xa, v32, g32, wb, act = ctx.saved_tensors
        shape, dil, k, pad = ctx.shape, ctx.dil, ctx.k, ctx.pad
        Cout, Cin, Ca, Cb = ctx.dims
        dev = xa.device
        g_raw = None if g_raw is None else g_raw.contiguous()
        g_act = None if g_act is None else g_act.contiguous()
        S = lib().psnd_conv1d_cl_wgrad_splits(shape.N, shape.Lp, Ca, Cb, k)
        am = act if g_act is not None else None
        need_gout = ctx.has_res and (g_act is not None)
        g_out = torch.empty((shape.N, shape.Lp, Cb), dtype=torch.bfloat16, device=dev) if need_gout else None
        # input gradient + weight-gradient slabs in ONE launch (conv_bwd_pair_kernel: both read the same incoming gradient,
        # each alone only part-fills the chip), then the weight-norm backward over the slabs
        gw = torch.empty((S, k, Cb, Ca), dtype=torch.float32, device=dev)
        gbp = torch.empty((S, Cb), dtype=torch.float32, device=dev)
        gb = torch.empty(Cb, dtype=torch.float32, device=dev)
        gv = torch.empty_like(v32)
        gg = torch.empty_like(g32)
        gx = torch.empty((shape.N, shape.Lp, Ca), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            check(lib().psnd_conv1d_cl_bwd(ptr(g_raw), ptr(g_act), ptr(am), float(ctx.act_slope), ptr(wb), ptr(xa), shape.N,
                                           shape.Lp, shape.L, shape.HP, Ca, Cb, k, pad, dil, ptr(gx), ptr(g_out), None, 1.0, None,
                                           ptr(gw), ptr(gbp), stream_ptr(dev)), 'psnd_conv1d_cl_bwd')
            check(lib().psnd_conv1d_wnorm_bwd(ptr(gw), ptr(gbp), S, ptr(v32), ptr(g32), Cout, Cin, k, Cb, Ca, ptr(gv),
                                              ptr(gg), ptr(gb), stream_ptr(dev)), 'psnd_conv1d_wnorm_bwd')
        g_res = None
        if ctx.has_res:
            g_res = g_out if need_gout else g_raw
        g_bias = gb[:Cout] if ctx.has_bias else None
        return gx, gv, gg, g_bias, g_res, None, None, None, None, None, None, None


def conv_transpose_cl(xa, up, shape, act_slope=0.1):
    """ConvTranspose1d of hifi_gan.py:118-121 (stride s, kernel k, padding p) on CL buffers, without leaving the layout:
    the input rows are spread s apart in a zeroed buffer and a plain k-tap convolution with the flipped, transposed
    weight (taps at -(k-1-p) .. +p) runs on the implicit-GEMM kernel - s-1 of every s products are zeros, which costs
    ~17 % extra FLOPs on the whole generator and saves the layout round trip and the library call.  The weight norm of
    the transposed weight (norm over dim 0 = input channels) is taken by torch; the kernel's own weight-norm
    parametrisation is fed g = ||w|| so that it reproduces w, and autograd adds the derivative through that norm.
    xa: activated CL input (N, Lp, Cin_p).  Returns (raw, leaky_relu(raw, act_slope), CLShape of the output)."""
    s, k, p = up.stride, up.weight_v.shape[2] if hasattr(up, 'weight_v') else up.weight.shape[2], up.padding
    N, T, HP = shape.N, shape.L, shape.HP
    if k - 1 - p < 0 or max(k - 1 - p, p) > HP:
        raise _lib.PsndError('conv_transpose_cl: k = %d, padding = %d needs a halo of %d rows (have %d)' % (k, p, max(k - 1 - p, p), HP))
    out_shape = CLShape(N, (T - 1) * s - 2 * p + k, HP)       # nn.ConvTranspose1d's output length (T * s for k - 2 p = s)
    xu = torch.zeros((N, out_shape.Lp, xa.shape[2]), dtype=xa.dtype, device=xa.device)
    xu[:, HP:HP + (T - 1) * s + 1:s, :] = xa[:, HP:HP + T, :]
    w_c = up.effective_weight().permute(1, 0, 2).flip(2).contiguous()            # (Cout, Cin, k) of the equivalent conv
    g = w_c.flatten(1).norm(dim=1).view(-1, 1, 1)
    raw, act = FusedConvCL.apply(xu, w_c, g, up.bias, None, out_shape, 1, True, True, act_slope, None, k - 1 - p)
    return raw, act, out_shape


def _up_covers(shape, out_shape, u, p):
    """true iff the polyphase forward kernel writes EVERY row of the high-resolution buffer: low row l writes the high rows
    HPO - p + (l - HP) u .. + u - 1 that exist (conv_cl_body, up_role 1), so the first low row must reach row 0 and the last one the end"""
    first = out_shape.HP - p - shape.HP * u
    last = out_shape.HP - p + (shape.Lp - 1 - shape.HP) * u + u - 1
    return first <= 0 and last >= out_shape.Lp - 1


class ConvTransposeCL(torch.autograd.Function):
    """ConvTranspose1d(k = 2 * stride, padding p) of hifi_gan.py:109 / 118-121 in polyphase form on the CL kernels
    (psnd_convtr1d_*): raw = y, act = leaky_relu(y, act_slope); the activations never leave the CL layout and no zero is
    multiplied.  weight_v: (Cin, Cout, K), weight_g: (Cin, 1, 1) - weight norm over dim 0, as torch's for nn.ConvTranspose1d."""

    @staticmethod
    def forward(ctx, xa, weight_v, weight_g, bias, shape, out_shape, stride, padding, act_slope, prepped=None):
        ctx.set_materialize_grads(False)     # an unused output hands None to backward, not a zero-filled tensor
        _need(xa, torch.bfloat16)
        Cin, Cout, K = weight_v.shape
        if K != 2 * stride or padding > stride:
            raise _lib.PsndError('ConvTransposeCL: kernel size %d, stride %d, padding %d (needs K = 2 * stride, padding <= stride)'
                                 % (K, stride, padding))
        Cip, Cr = xa.shape[2], round_up(Cout, ALIGN_C)
        if Cip != round_up(Cin, ALIGN_C) or out_shape.L != shape.L * stride or out_shape.N != shape.N:
            raise _lib.PsndError('ConvTransposeCL: buffer / geometry mismatch')
        dev = xa.device
        v32, g32 = weight_v.detach().contiguous(), weight_g.detach().contiguous()
        b32 = None if bias is None else bias.detach().contiguous()
        if prepped is not None:                      # packs made by prep_all_convtr() for all upsamplers in one launch
            wf, wb, bp = prepped
        else:
            wf = torch.empty(2 * stride * Cr * Cip, dtype=torch.bfloat16, device=dev)
            wb = torch.empty(2 * stride * Cr * Cip, dtype=torch.bfloat16, device=dev)
            bp = torch.empty(stride * Cr, dtype=torch.float32, device=dev)
        # the kernel writes every high-resolution row some low-resolution row maps to (zeros outside the clip); when those reach both ends
        # of the buffer - every HiFi-GAN upsampler: the same halo on both sides of the stride - nothing has to be zeroed first
        alloc = torch.empty if _up_covers(shape, out_shape, stride, padding) else torch.zeros
        both = alloc((2, shape.N, out_shape.Lp, Cr), dtype=torch.bfloat16, device=dev)
        raw, act = both[0], both[1]
        with torch.cuda.device(dev):
            st = stream_ptr(dev)
            if prepped is None:
                check(lib().psnd_convtr1d_prep(ptr(v32), ptr(g32), ptr(b32), Cin, Cout, K, stride, Cr, Cip, ptr(wf), ptr(wb), ptr(bp), st),
                      'psnd_convtr1d_prep')
            check(lib().psnd_convtr1d_cl_fwd(ptr(xa), ptr(wf), ptr(bp), shape.N, shape.Lp, shape.L, shape.HP, Cip, Cr, stride, padding,
                                             out_shape.Lp, out_shape.HP, float(act_slope), ptr(raw), ptr(act), st), 'psnd_convtr1d_cl_fwd')
        ctx.geo = (shape, out_shape, stride, padding, float(act_slope), Cin, Cout, K, Cip, Cr)
        ctx.has_bias = bias is not None
        ctx.params = (weight_v, weight_g, bias)
        note_param_use(ctx, weight_v, weight_g, bias)
        ctx.save_for_backward(xa, v32, g32, wb, act)
        return raw, act

    @staticmethod
    def backward(ctx, g_raw, g_act):
        xa, v32, g32, wb, act = ctx.saved_tensors
        shape, out_shape, stride, padding, slope, Cin, Cout, K, Cip, Cr = ctx.geo
        dev = xa.device
        g_raw = None if g_raw is None else g_raw.contiguous()
        g_act = None if g_act is None else g_act.contiguous()
        if g_raw is None and g_act is None:
            raise _lib.PsndError('ConvTransposeCL backward: no incoming gradient')
        S = lib().psnd_convtr1d_cl_wgrad_splits(shape.N, shape.Lp, Cip, Cr, stride)
        gx = torch.empty((shape.N, shape.Lp, Cip), dtype=torch.bfloat16, device=dev)
        # the combined gradient: written for every group of `stride` rows that lies inside the buffer (all the weight gradient reads); when
        # those include every clip row the bias gradient sums the clip rows only and nothing has to be zeroed first
        clip_rows = out_shape.HP >= stride
        g_eff = None
        if g_act is not None:
            g_eff = (torch.empty if clip_rows else torch.zeros)((shape.N, out_shape.Lp, Cr), dtype=torch.bfloat16, device=dev)
        gw = torch.empty((S, 2, Cip, stride * Cr), dtype=torch.float32, device=dev)
        gv, gg = torch.empty_like(v32), torch.empty_like(g32)
        # Parameter-side branch (weight gradient, weight-norm backward, bias gradient: needed only by the optimizer) on a side stream -
        # inside the step graph a parallel branch next to the stage's resblock backward that follows, joined once at the end of the
        # backward pass.  Only when these gradients are WRITTEN (no .grad yet: autograd then takes the tensors without a launch) and
        # nothing else shares the hardware queues with the step (AUTO_SECTIONS).
        side = None
        if (BRANCH_PARAM_GRADS and AUTO_SECTIONS
                and all(q is None or (q.is_leaf and q.grad is None and single_use(q)) for q in ctx.params)):
            side = param_stream(dev)
            if GRAD_SINK is not None:          # a reducer's bucket waits for the stream that WRITES these gradients (round 5)
                GRAD_SINK.note_producer(ctx.params, side)
        main = torch.cuda.current_stream(dev)

        def param_side(st):
            check(lib().psnd_convtr1d_cl_bwd(ptr(g_raw), ptr(g_act), ptr(act if g_act is not None else None), slope, ptr(wb), ptr(xa),
                                             shape.N, shape.Lp, shape.L, shape.HP, Cip, Cr, stride, padding, out_shape.Lp, out_shape.HP,
                                             None, ptr(g_eff), ptr(gw), st), 'psnd_convtr1d_cl_bwd')
            check(lib().psnd_convtr1d_wnorm_bwd(ptr(gw), S, ptr(v32), ptr(g32), Cin, Cout, K, stride, Cr, Cip, ptr(gv), ptr(gg), st),
                  'psnd_convtr1d_wnorm_bwd')
            if ctx.has_bias:   # column sums of the combined gradient (halo rows are zero): psnd_cl_colsum, fp32, fixed order
                gsrc = (g_eff if g_eff is not None else g_raw).view(-1, Cr)
                if Cr > 256:   # (wider than the kernel's column groups: not a HiFi-GAN upsampler)
                    return torch.sum(gsrc, 0, dtype=torch.float32)[:Cout]
                rows = gsrc.shape[0]
                part = torch.empty(lib().psnd_cl_colsum_splits(rows, Cr) * Cr, dtype=torch.float32, device=dev)
                gb_full = torch.empty(Cr, dtype=torch.float32, device=dev)
                win = (out_shape.Lp, out_shape.HP, out_shape.HP + out_shape.L) if (g_eff is not None and clip_rows) else (0, 0, 0)
                check(lib().psnd_cl_colsum(ptr(gsrc), rows, Cr, win[0], win[1], win[2], ptr(part), ptr(gb_full), st), 'psnd_cl_colsum')
                return gb_full[:Cout]
            return None

        with torch.cuda.device(dev):
            check(lib().psnd_convtr1d_cl_bwd(ptr(g_raw), ptr(g_act), ptr(act if g_act is not None else None), slope, ptr(wb), ptr(xa),
                                             shape.N, shape.Lp, shape.L, shape.HP, Cip, Cr, stride, padding, out_shape.Lp, out_shape.HP,
                                             ptr(gx), ptr(g_eff), None, stream_ptr(dev)), 'psnd_convtr1d_cl_bwd')
            if side is None:
                g_bias = param_side(stream_ptr(dev))
            else:
                side.wait_stream(main)                       # the combined gradient is complete on the main stream
                with torch.cuda.stream(side):
                    gw = torch.empty_like(gw)
                    gv, gg = torch.empty_like(gv), torch.empty_like(gg)
                    g_bias = param_side(stream_ptr(dev))
                for t in (g_raw, g_act, g_eff, act, xa, v32, g32, wb):     # allocated on the main stream, still read by the side stream
                    if t is not None:
                        t.record_stream(side)
                _join_side_at_end_of_backward(dev, side)
        consume_param_use(ctx)
        return gx, gv, gg, g_bias, None, None, None, None, None, None


def prep_all_convtr(owner, ups):
    """weight norm (over dim 0) + both bf16 packs + replicated bias of every transposed conv in `ups` (WNConvTranspose1d modules with
    k = 2 * stride) in ONE launch (psnd_convtr1d_prep_multi); buffers and the descriptor table cached on `owner`.
    Returns {id(up): (wf, wb, bp)} for ConvTransposeCL.apply(..., prepped)."""
    import struct
    key = tuple((u.weight_v.data_ptr(), u.weight_g.data_ptr(), 0 if u.bias is None else u.bias.data_ptr(), tuple(u.weight_v.shape), u.stride)
                for u in ups)
    cache = getattr(owner, '_cl_prep_convtr_cache', None)
    if cache is None or cache['key'] != key:
        dev = ups[0].weight_v.device
        packs, recs, blk0 = {}, [], 0
        for u in ups:
            Cin, Cout, K = u.weight_v.shape
            Cip, Cr = round_up(Cin, ALIGN_C), round_up(Cout, ALIGN_C)
            wf = torch.zeros(2 * u.stride * Cr * Cip, dtype=torch.bfloat16, device=dev)
            wb = torch.zeros(2 * u.stride * Cr * Cip, dtype=torch.bfloat16, device=dev)
            bp = torch.zeros(u.stride * Cr, dtype=torch.float32, device=dev)
            packs[id(u)] = (wf, wb, bp)
            recs.append(struct.pack('<6Q8i', u.weight_v.data_ptr(), u.weight_g.data_ptr(), 0 if u.bias is None else u.bias.data_ptr(),
                                    wf.data_ptr(), wb.data_ptr(), bp.data_ptr(), Cin, Cout, K, u.stride, Cr, Cip, blk0, 0))
            blk0 += Cin
        table = torch.frombuffer(bytearray(b''.join(recs)), dtype=torch.uint8).to(dev)
        cache = {'key': key, 'packs': packs, 'table': table, 'n': len(ups), 'blocks': blk0, 'dev': dev}
        owner._cl_prep_convtr_cache = cache
    for u in ups:
        if not (u.weight_v.is_contiguous() and u.weight_g.is_contiguous() and u.weight_v.dtype == torch.float32):
            raise _lib.PsndError('prep_all_convtr: fp32 contiguous weight_v / weight_g expected')
    with torch.cuda.device(cache['dev']):
        check(lib().psnd_convtr1d_prep_multi(ptr(cache['table']), cache['n'], cache['blocks'], stream_ptr(cache['dev'])),
              'psnd_convtr1d_prep_multi')
    return cache['packs']


class FanOutCL(torch.autograd.Function):
    """(x_raw, x_act) of an upsampler -> `count` aliases of each, one pair per resblock of the stage (hifi_gan.py:122-131: every resblock
    reads x; their outputs are averaged).  Forward is free (aliases); backward adds the gradients that come back - up to `count` per output -
    in ONE launch (psnd_cl_sum2, fp32 accumulation, one rounding) instead of autograd's 2 x (count - 1) library adds."""

    @staticmethod
    def forward(ctx, x_raw, x_act, count):
        ctx.set_materialize_grads(False)
        ctx.count = count
        ctx.meta = (x_raw.shape, x_raw.device)
        outs = []
        for _ in range(count):
            outs += [x_raw.detach(), x_act.detach()]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        g_raw = [g.contiguous() for g in gs[0::2] if g is not None]
        g_act = [g.contiguous() for g in gs[1::2] if g is not None]
        shape, dev = ctx.meta

        def single(lst):
            return lst[0] if len(lst) == 1 else None

        if len(g_raw) <= 1 and len(g_act) <= 1:
            return single(g_raw), single(g_act), None
        out_r = single(g_raw) if len(g_raw) <= 1 else torch.empty(shape, dtype=torch.bfloat16, device=dev)
        out_a = single(g_act) if len(g_act) <= 1 else torch.empty(shape, dtype=torch.bfloat16, device=dev)
        for g in g_raw + g_act:
            _need(g, torch.bfloat16)
        ra = g_raw if len(g_raw) > 1 else []
        aa = g_act if len(g_act) > 1 else []
        pa = (ctypes.c_void_p * max(1, len(ra)))(*[g.data_ptr() for g in ra])
        pb = (ctypes.c_void_p * max(1, len(aa)))(*[g.data_ptr() for g in aa])
        n = (ra or aa)[0].numel()
        with torch.cuda.device(dev):
            check(lib().psnd_cl_sum2(pa, len(ra), ptr(out_r) if ra else None, pb, len(aa), ptr(out_a) if aa else None, n, stream_ptr(dev)),
                  'psnd_cl_sum2')
        return out_r, out_a, None


class MeanActCL(torch.autograd.Function):
    """leaky_relu(mean of the stage's resblock outputs, slope) on CL buffers in one pass each way (hifi_gan.py:122-131:
    xs / num_kernels, then the leaky_relu in front of the next upsampler / of conv_post)"""

    @staticmethod
    def forward(ctx, slope, *rs):
        n = len(rs)
        if not 1 <= n <= 4:
            raise _lib.PsndError('MeanActCL: 1..4 branches, got %d' % n)
        rs = [r.contiguous() for r in rs]
        for r in rs:
            _need(r, torch.bfloat16)
        out = torch.empty_like(rs[0])
        ps = [ptr(r) for r in rs] + [None] * (4 - n)
        with torch.cuda.device(out.device):
            check(lib().psnd_cl_mean_act_fwd(ps[0], ps[1], ps[2], ps[3], n, float(slope), ptr(out), out.numel(), stream_ptr(out.device)),
                  'psnd_cl_mean_act_fwd')
        ctx.n, ctx.slope = n, float(slope)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = g.contiguous()
        gin = torch.empty_like(out)
        with torch.cuda.device(out.device):
            check(lib().psnd_cl_mean_act_bwd(ptr(g), ptr(out), ctx.n, ctx.slope, ptr(gin), out.numel(), stream_ptr(out.device)),
                  'psnd_cl_mean_act_bwd')
        return (None,) + (gin,) * ctx.n


class PrepPacks(dict):
    """{id(conv): (wf, wb, bp)} of prep_all().  `deferred`: set when the backward packs have NOT been written yet - the first backward
    node that needs one calls write_backward_packs() (once per forward)."""
    deferred = None

    def write_backward_packs(self):
        cache = self.deferred
        if cache is None or cache['wb_done']:
            return
        cache['wb_done'] = True
        with torch.cuda.device(cache['dev']):
            check(lib().psnd_conv1d_prep_multi(ptr(cache['table']), cache['n'], cache['blocks'], cache['max_row'], 2, stream_ptr(cache['dev'])),
                  'psnd_conv1d_prep_multi')


def prep_all(owner, convs, defer_backward_packs=False):
    """weight norm + both bf16 packs + padded bias of every conv in `convs` (WNConv1d modules) in ONE launch
    (psnd_conv1d_prep_multi).  The pack buffers and the device descriptor table are cached on `owner` and rebuilt only
    when a parameter moved; returns {id(conv): (wf, wb, bp)} for fused_conv(..., prepped=...).
    defer_backward_packs: only the forward packs + biases now; the [j][ci][co] packs of the input-gradient kernels are written by a second
    launch when the backward starts (PrepPacks.write_backward_packs, called by ResBlockCL.backward).  Measured on the config-2 step: packs
    written with 16-byte stores are read at full speed while fresh (the forward chains right behind the launch) but ~25 % slower 300 us
    later (input-gradient chains 38 -> 47 us each) - the per-channel kernel's 2-byte stores did not show that, at 35 us per launch."""
    import struct
    import os
    if _sw.lab('PSND_NO_PREP_ALL') == '1':      # A/B switch: one prep launch per conv
        return None
    # psnd_conv1d_prep_multi stages 8 rows of Cin * k bf16 values in LDS: wider rows (e.g. 1024 channels x 11 taps) take the per-conv prep
    # launch (fused_conv / ResBlockCL with prep = None) instead of failing with PSND_E_SHAPE
    if any(c.weight_v.shape[1] * c.weight_v.shape[2] > 10240 for c in convs):
        return None
    key = tuple((c.weight_v.data_ptr(), c.weight_g.data_ptr(), 0 if c.bias is None else c.bias.data_ptr(),
                 tuple(c.weight_v.shape)) for c in convs)
    cache = getattr(owner, '_cl_prep_cache', None)
    if cache is None or cache['key'] != key:
        dev = convs[0].weight_v.device
        packs, recs, blk0, max_row = PrepPacks(), [], 0, 0
        for c in convs:
            Cout, Cin, k = c.weight_v.shape
            Ca, Cb = round_up(Cin, ALIGN_C), round_up(Cout, ALIGN_C)
            wf = torch.zeros((k, Cb, Ca), dtype=torch.bfloat16, device=dev)
            wb = torch.zeros((k, Ca, Cb), dtype=torch.bfloat16, device=dev)
            bp = torch.zeros(Cb, dtype=torch.float32, device=dev)
            packs[id(c)] = (wf, wb, bp)
            recs.append(struct.pack('<6Q6i', c.weight_v.data_ptr(), c.weight_g.data_ptr(),
                                    0 if c.bias is None else c.bias.data_ptr(), wf.data_ptr(), wb.data_ptr(), bp.data_ptr(),
                                    Cout, Cin, k, Cb, Ca, blk0))
            blk0 += (Cout + 7) // 8                   # one workgroup per 8 output channels
            max_row = max(max_row, Cin * k)
        table = torch.frombuffer(bytearray(b''.join(recs)), dtype=torch.uint8).to(dev)
        cache = {'key': key, 'packs': packs, 'table': table, 'n': len(convs), 'blocks': blk0, 'max_row': max_row, 'dev': dev,
                 'wb_done': True}
        owner._cl_prep_cache = cache
    for c in convs:
        if not (c.weight_v.is_contiguous() and c.weight_g.is_contiguous() and c.weight_v.dtype == torch.float32):
            raise _lib.PsndError('prep_all: fp32 contiguous weight_v / weight_g expected')
    defer = bool(defer_backward_packs) and torch.is_grad_enabled() and _sw.lab('PSND_PREP_NO_DEFER') != '1'
    with torch.cuda.device(cache['dev']):
        check(lib().psnd_conv1d_prep_multi(ptr(cache['table']), cache['n'], cache['blocks'], cache['max_row'], 1 if defer else 3,
                                           stream_ptr(cache['dev'])), 'psnd_conv1d_prep_multi')
    cache['wb_done'] = not defer
    cache['packs'].deferred = cache if defer else None
    return cache['packs']


def fused_conv(xa, conv, shape, res=None, want_raw=False, want_act=True, act_slope=0.1, prep=None):
    """conv: a WNConv1d module (weight_v, weight_g, bias, dilation); prep: the dict returned by prep_all()."""
    return FusedConvCL.apply(xa, conv.weight_v, conv.weight_g, conv.bias, res, shape, conv.dilation, want_raw, want_act,
                             act_slope, None if prep is None else prep[id(conv)])


class ResBlockCL(torch.autograd.Function):
    """A whole ResBlock1 (pairs conv1 -> conv2 + residual, hifi_gan.py:55-63) or ResBlock2 (conv + residual per conv, :84-88)
    as ONE autograd node.  Same launches as a chain of FusedConvCL nodes in forward and for the paired input-/weight-gradient
    kernels in backward, but the weight-norm backward of ALL the block's convs is one launch (psnd_conv1d_wnorm_bwd_multi)
    instead of one per conv, and the gradient hand-over between the convs (raw / activated parts, residual branch) needs no
    autograd bookkeeping.  params: (weight_v, weight_g, bias) per conv, flattened; packs: per conv (wf, wb, bp) or None."""

    @staticmethod
    def forward(ctx, x, xa, shape, roles, dils, last_act_slope, want_raw, packs, *params):
        """roles[i]:  'c1' / 'c2' the two convs of a ResBlock1 pair, 'r2' a ResBlock2 conv (conv + residual), 'head' a plain conv in
        front of the chain (input xa, raw + activated output start the residual stream; x is None), 'tail' a plain conv behind it
        (activated input, raw output only - returned as the first output, the second is None)."""
        ctx.set_materialize_grads(False)     # an unused output hands None to backward, not a zero-filled tensor
        _need(xa, torch.bfloat16)
        dev = xa.device
        n = len(dils)
        ctx.pack_owner = getattr(packs, 'owner', None)       # PrepPacks with deferred backward packs (prep_all)
        steps, saved, plan = [], [], []
        cur_x, cur_xa, pending = x, xa, None
        for i in range(n):
            wv, wg, b = params[3 * i:3 * i + 3]
            Cout, Cin, k = wv.shape
            role = roles[i]
            src = pending if role == 'c2' else cur_xa
            Ca, Cb = src.shape[2], round_up(Cout, ALIGN_C)
            if Ca != round_up(Cin, ALIGN_C) or (role in ('c2', 'r2') and Cb != cur_x.shape[2]):
                raise _lib.PsndError('CL conv chain: %s conv %d -> %d channels on a %d-channel buffer' % (role, Cin, Cout, Ca))
            dil = dils[i]
            pad = (k * dil - dil) // 2
            v32, g32 = wv.detach().contiguous(), wg.detach().contiguous()
            if packs is not None and packs[i] is not None:
                wf, wb, bp = packs[i]
            else:
                wf = torch.empty((k, Cb, Ca), dtype=torch.bfloat16, device=dev)
                wb = torch.empty((k, Ca, Cb), dtype=torch.bfloat16, device=dev)
                bp = torch.empty(Cb, dtype=torch.float32, device=dev)
                b32 = None if b is None else b.detach().contiguous()
                with torch.cuda.device(dev):
                    check(lib().psnd_conv1d_prep(ptr(v32), ptr(g32), ptr(b32), Cout, Cin, k, Cb, Ca, ptr(wf), ptr(wb), ptr(bp),
                                                 stream_ptr(dev)), 'psnd_conv1d_prep')
            last = i == n - 1
            inp, slope, has_res = src, (last_act_slope if last else 0.1), role in ('c2', 'r2')
            mk = lambda: torch.empty((shape.N, shape.Lp, Cb), dtype=torch.bfloat16, device=dev)   # noqa: E731
            if role == 'c1':                               # activated output only, no residual
                raw, act, res, oslope = None, mk(), None, slope
                pending = act
            elif role == 'tail':                           # raw output only
                raw, act, res, oslope = mk(), None, None, 1.0
                cur_x, cur_xa = raw, None
            else:                                          # 'head' (no residual) / 'c2' / 'r2': raw + activated output
                raw, act, res, oslope = (mk() if (not last) or want_raw else None), mk(), (cur_x if has_res else None), slope
                cur_x, cur_xa = raw, act
            if (role == 'c2' and _pair_enabled() and plan and plan[-1][0] == 'conv' and steps[-1][10] == 'c1' and plan[-1][5] == Ca == Cb
                    and plan[-1][7] == k and lib().psnd_conv1d_cl_pair_supported(Ca, k, plan[-1][8], plan[-1][9], -pad, dil)):
                # conv1 -> conv2 of a ResBlock1 pair as ONE launch (psnd_conv1d_cl_pair): the first conv's output stays in LDS
                _, inp1, wf1, bp1, _, _, _, _, off1, dil1, slope1, _, mid = plan.pop()
                plan.append(('pair', inp1, wf1, bp1, slope1, mid, wf, bp, res, Ca, k, off1, dil1, -pad, dil, oslope, raw, act))
            else:
                plan.append(('conv', inp, wf, bp, res, Ca, Cb, k, -pad, dil, oslope, raw, act))
            steps.append((Cout, Cin, k, Ca, Cb, dil, pad, slope, has_res, b is not None, role))
            saved += [inp, act if act is not None else inp, wb, v32, g32]
        nsec, sides = _sections(dev, shape.N, shape.N * shape.Lp)
        Nh = shape.N // nsec
        plan = _chain_pairs(plan, Nh * shape.Lp)

        def run(h):
            st = stream_ptr(dev)
            q = lambda t: ptr(_sec(t, h, nsec))                # noqa: E731
            for e in plan:
                if e[0] == 'chain':
                    _, inp1, res0, C, k, pairs = e
                    arr = (_lib.ChainPair * len(pairs))()
                    for d, (_, _, wf1, bp1, slope1, mid, wf2, bp2, _, _, _, off1, dil1, off2, dil2, oslope, raw, act) in zip(arr, pairs):
                        d.W1, d.bias1, d.act1_slope, d.mid_out = wf1.data_ptr(), bp1.data_ptr(), float(slope1), q(mid)
                        d.W2, d.bias2, d.off1, d.dstep1, d.off2, d.dstep2 = wf2.data_ptr(), bp2.data_ptr(), off1, dil1, off2, dil2
                        # the residual stream BETWEEN the pairs of a chain never leaves the chip (nothing reads it back: the backward
                        # needs the activated tensors only); the last pair's is the next launch's
                        d.act2_slope, d.out_raw, d.out_act = float(oslope), (q(raw) if raw is pairs[-1][16] else None), q(act)
                    check(lib().psnd_conv1d_cl_chain(q(inp1), q(res0), ctypes.addressof(arr), len(pairs), Nh, shape.Lp, shape.L, shape.HP,
                                                     C, k, st), 'psnd_conv1d_cl_chain')
                    continue
                if e[0] == 'pair':
                    _, inp1, wf1, bp1, slope1, mid, wf2, bp2, res, C, k, off1, dil1, off2, dil2, oslope, raw, act = e
                    check(lib().psnd_conv1d_cl_pair(q(inp1), ptr(wf1), ptr(bp1), None, 1.0, float(slope1), q(mid), ptr(wf2), ptr(bp2),
                                                    None, 1.0, q(res), Nh, shape.Lp, shape.L, shape.HP, C, k, off1, dil1, off2, dil2,
                                                    float(oslope), q(raw), q(act), st), 'psnd_conv1d_cl_pair')
                    continue
                _, inp, wf, bp, res, Ca, Cb, k, off0, dil, oslope, raw, act = e
                check(lib().psnd_conv1d_cl(q(inp), None, None, 1.0, ptr(wf), ptr(bp), q(res), None,
                                           Nh, shape.Lp, shape.L, shape.HP, Ca, Cb, k, off0, dil, float(oslope), 1.0,
                                           q(raw), q(act), None, st), 'psnd_conv1d_cl')

        with torch.cuda.device(dev):
            _run_sections(dev, nsec, sides, run)
        ctx.steps, ctx.shape = steps, shape
        ctx.param_refs = params                    # the Parameter objects (GRAD_SINK looks its buckets up by them)
        note_param_use(ctx, *params)
        ctx.save_for_backward(*saved)
        return cur_x, cur_xa

    @staticmethod
    def backward(ctx, g_raw, g_act):
        import os
        import struct
        import ctypes
        saved, steps, shape = ctx.saved_tensors, ctx.steps, ctx.shape
        dev = saved[0].device
        n = len(steps)
        if ctx.pack_owner is not None:
            ctx.pack_owner.write_backward_packs()          # the [j][ci][co] packs: written now, read while fresh
        g_raw = None if g_raw is None else g_raw.contiguous()
        g_act = None if g_act is None else g_act.contiguous()
        grads = [None] * (3 * n)
        descs, keep = [None] * n, []
        res_pending = None
        sink = GRAD_SINK if (GRAD_SINK is not None and GRAD_SINK.sink_active()) else None
        prefs = getattr(ctx, 'param_refs', None)
        early = {}                             # conv index -> parameters whose gradient goes straight into the sink's buffers

        def grad_out(i, v32, g32, Cb):
            """(g_v, g_g, g_bias[Cb]) of conv i: the sink's flat-bucket slots when there is one, else fresh tensors"""
            gv = gg = gb = None
            if sink is not None and prefs is not None:
                pv, pg, pb = prefs[3 * i:3 * i + 3]
                gv, gg = sink.dest(pv, pv.numel()), sink.dest(pg, pg.numel())
                gb = sink.dest(pb, Cb) if pb is not None else None
                early[i] = [(0, pv) if gv is not None else None, (1, pg) if gg is not None else None,
                            (2, pb) if gb is not None else None]
            gv = torch.empty_like(v32) if gv is None else gv.view_as(v32)
            gg = torch.empty_like(g32) if gg is None else gg.view_as(g32)
            gb = torch.empty(Cb, dtype=torch.float32, device=dev) if gb is None else gb
            return gv, gg, gb

        nsec, sides = _sections(dev, shape.N, shape.N * shape.Lp)
        wstreams = _wgrad_streams(dev) if shape.N * shape.Lp <= 8192 else []
        if wstreams:
            nsec, sides = 1, []                # split backward: whole-batch launches, the concurrency comes from the wgrad streams
        Nh = shape.N // nsec
        plan = []                              # launches, walked once per batch section (_run_sections)
        skip = -1                              # conv already handled as the first conv of a fused input-gradient pair
        # batched backward (default; PSND_CL_BWD_BATCH=0: one psnd_conv1d_cl_pair_bwd launch per pair): the input-gradient chain runs alone (one psnd_conv1d_cl_pair launch per residual pair,
        # masks and mirrored taps), the weight gradients of all its convs follow in ONE launch (psnd_conv1d_cl_wgrad_multi)
        batch = (nsec == 1 and not wstreams and _pair_enabled() and _sw.lab('PSND_CL_BWD_BATCH', '1') == '1'
                 and shape.N * shape.Lp <= 8192)
        wbatch = []
        own_launch = []                        # (plan entry, conv): convs whose weight-gradient slabs are written by that entry itself
        flush_after = {}                       # id(input-gradient tensor a pair launch writes) -> len(wbatch) once that launch has run
        pair_bwd = (nsec == 1 and not wstreams and not batch and _pair_enabled() and _sw.lab('PSND_CL_PAIR_BWD', '1') != '0'
                    and shape.N * shape.Lp <= 8192)
        lag = None                             # weight gradient of a pair's first conv, carried to the next pair launch
        with torch.cuda.device(dev):
            # The gradient a conv receives is  g = g_raw + g_act * leaky'(own activated output).  Only the block's LAST conv gets
            # the two parts from outside and combines them on load; every earlier conv's g is formed in the EPILOGUE of the input-
            # gradient role that produces g_act (mask by the activation, add the residual branch's gradient), so its own backward
            # reads ONE plain tensor in both roles - and that tensor also is the gradient handed on along the residual stream.
            g_comb = None                      # combined incoming gradient of conv i (None for the block's last conv)
            res_pending = None                 # pairs: gradient on the residual stream behind the pair being walked
            def slabs(i, G1, G2, am, inp, S=None):
                Cout, Cin, k, Ca, Cb, dil, pad, slope = steps[i][:8]
                v32, g32 = saved[5 * i + 3], saved[5 * i + 4]
                if S is None:
                    S = lib().psnd_conv1d_cl_wgrad_splits(Nh, shape.Lp, Ca, Cb, k)
                gw = torch.empty((nsec * S, k, Cb, Ca), dtype=torch.float32, device=dev)    # slabs of section h: [h S, (h+1) S)
                gbp = torch.empty((nsec * S, Cb), dtype=torch.float32, device=dev)
                gv, gg, gb = grad_out(i, v32, g32, Cb)
                descs[i] = struct.pack('<7Q6i', gw.data_ptr(), gbp.data_ptr(), v32.data_ptr(), g32.data_ptr(), gv.data_ptr(),
                                       gg.data_ptr(), gb.data_ptr(), nsec * S, Cout, Cin, k, Cb, Ca)
                keep.extend([gw, gbp, G1, G2])
                grads[3 * i], grads[3 * i + 1] = gv, gg
                grads[3 * i + 2] = gb[:Cout] if steps[i][9] else None
                return ('w', G1, G2, am, slope, inp, Ca, Cb, k, -pad, dil, gw, gbp, S)

            for i in range(n - 1, -1, -1):
                if i == skip:
                    continue
                Cout, Cin, k, Ca, Cb, dil, pad, slope, has_res, has_bias, role = steps[i]
                inp, act, wb, v32, g32 = saved[5 * i:5 * i + 5]
                if (pair_bwd and role == 'c2' and i >= 2 and steps[i - 1][10] == 'c1' and g_comb is not None and Ca == Cb
                        and steps[i - 1][3] == steps[i - 1][4] == Ca and steps[i - 1][2] == k
                        and lib().psnd_conv1d_cl_pair_bwd_supported(Ca, k, pad, dil, steps[i - 1][6], steps[i - 1][5])):
                    # the whole pair's backward as ONE launch (psnd_conv1d_cl_pair_bwd): input gradients of conv2 and conv1 chained on
                    # chip, conv2's weight gradient, and the weight gradient of the conv1 of the pair handled BEFORE (its gradient is
                    # that launch's g_h); this pair's conv1 waits for the next such launch (or the flush after the loop)
                    inp1, wb1 = saved[5 * (i - 1)], saved[5 * (i - 1) + 2]
                    d1, pad1 = steps[i - 1][5], steps[i - 1][6]
                    G = g_comb
                    g_h = torch.empty((shape.N, shape.Lp, Ca), dtype=torch.bfloat16, device=dev)
                    gx = torch.empty((shape.N, shape.Lp, Ca), dtype=torch.bfloat16, device=dev)
                    S2 = lib().psnd_conv1d_cl_pair_bwd_splits(shape.N, shape.Lp, Ca, k)
                    wa = slabs(i, G, None, None, inp, S2)
                    plan.append(('pb', G, wb, inp, float(steps[i - 1][7]), g_h, wb1, inp1, float(steps[i - 2][7]), Ca, k, pad, dil, pad1, d1, gx,
                                 wa[11], wa[12], lag))
                    wl = slabs(i - 1, g_h, None, None, inp1, S2)
                    lag = (g_h, inp1, -pad1, d1, wl[11], wl[12], Ca, k)
                    keep.extend([g_h, gx])
                    res_pending = G
                    g_comb = gx
                    skip = i - 1
                    continue
                if ((wstreams or batch) and role == 'c2' and i >= 2 and steps[i - 1][10] == 'c1' and g_comb is not None and Ca == Cb
                        and steps[i - 1][3] == steps[i - 1][4] == Ca and steps[i - 1][2] == k
                        and k == 3 and Ca in (128, 256)          # (the masked form of the pair kernel: the 3-tap instances only)
                        and lib().psnd_conv1d_cl_pair_supported(Ca, k, pad, -dil, steps[i - 1][6], -steps[i - 1][5])):
                    # input gradients of conv2 and conv1 of a residual pair as ONE launch on this stream (psnd_conv1d_cl_pair with
                    # the transposed packs, mirrored taps and the leaky' masks); their weight gradients go to the side streams, or
                    # (batch) into the one launch behind the chain
                    inp1, wb1 = saved[5 * (i - 1)], saved[5 * (i - 1) + 2]
                    d1, pad1 = steps[i - 1][5], steps[i - 1][6]
                    G = g_comb
                    g_h = torch.empty((shape.N, shape.Lp, Ca), dtype=torch.bfloat16, device=dev)      # gradient wrt conv1's output
                    gx = torch.empty((shape.N, shape.Lp, Ca), dtype=torch.bfloat16, device=dev)
                    if batch:
                        wbatch.append((i, G, inp, -pad, dil))
                    else:
                        plan.append(('side',) + slabs(i, G, None, None, inp))                          # ready before the pair launch
                    plan.append(('pairb', G, wb, inp, float(steps[i - 1][7]), g_h, wb1, inp1, float(steps[i - 2][7]), G, Ca, k,
                                 pad, -dil, pad1, -d1, gx))
                    if batch:
                        wbatch.append((i - 1, g_h, inp1, -pad1, d1))
                        flush_after[id(gx)] = len(wbatch)
                    else:
                        plan.append(('side',) + slabs(i - 1, g_h, None, None, inp1))
                    keep.extend([g_h, gx])
                    res_pending = G
                    g_comb = gx
                    skip = i - 1
                    continue
                if g_comb is None:
                    if g_raw is None and g_act is None:
                        raise _lib.PsndError('CL conv chain backward: no incoming gradient')
                    G1, G2 = g_raw, g_act
                    am = act if g_act is not None else None
                    need_gout = has_res and g_act is not None
                else:
                    G1, G2, am, need_gout = g_comb, None, None, False
                S = lib().psnd_conv1d_cl_wgrad_splits(Nh, shape.Lp, Ca, Cb, k)
                gw = torch.empty((nsec * S, k, Cb, Ca), dtype=torch.float32, device=dev)    # slabs of section h: [h S, (h+1) S)
                gbp = torch.empty((nsec * S, Cb), dtype=torch.float32, device=dev)
                gv, gg, gb = None, None, None                  # (taken below, once it is known that this conv keeps its own launch)
                g_out = torch.empty((shape.N, shape.Lp, Cb), dtype=torch.bfloat16, device=dev) if need_gout else None
                g_here = g_out if need_gout else G1          # this conv's combined gradient (= what its residual input receives)
                # convs outside the residual pairs (a model's head / tail) keep their own launches: with them the one weight-gradient launch
                # grows by as much as their launches took (PSND_CL_BWD_BATCH_ENDS=1: 81 -> 122 us against 23 + 7 us saved at config 2)
                wb_ok = batch and G2 is None and not need_gout and _sw.lab('PSND_CL_BWD_BATCH_ENDS', '0') == '1'
                if i == 0 and not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
                    # the chain's input needs no gradient (features): weight gradient only
                    gx = None
                    if wb_ok:                                  # batched backward: with all the others, in the one launch behind the chain
                        wbatch.append((i, G1, inp, -pad, dil))
                        keep.append(G1)
                        if i > 0:
                            g_comb = gx
                        g_raw = g_act = None
                        continue
                    plan.append((('side',) if wstreams else ()) + ('w', G1, G2, am, slope, inp, Ca, Cb, k, -pad, dil, gw, gbp, S))
                    g_raw = g_act = None
                else:
                    gx = torch.empty((shape.N, shape.Lp, Ca), dtype=torch.bfloat16, device=dev)
                    # epilogue of the input gradient: this conv's input `inp` is the previous conv's activated output
                    if i == 0:
                        ep_mask, ep_res = None, None           # leaves the chain: (g_x, g_xa) are returned apart
                    elif role == 'c1':                         # -> conv2 / head before it: + the residual stream's gradient
                        ep_mask, ep_res = inp, res_pending
                    elif role == 'r2':                         # ResBlock2: every conv carries the residual
                        ep_mask, ep_res = inp, g_here
                    else:                                      # 'c2' -> its conv1, 'tail' -> the last conv of the stack
                        ep_mask, ep_res = inp, None
                    if wb_ok:                                  # input gradient alone here; the weight gradient joins the batch
                        plan.append(('b', G1, G2, am, slope, wb, inp, Ca, Cb, k, pad, dil, gx, g_out, ep_mask,
                                     float(steps[i - 1][7] if i > 0 else 1.0), ep_res, None, None, S))
                        wbatch.append((i, G1, inp, -pad, dil))
                        keep += [G1]
                        if role == 'c2':
                            res_pending = g_here
                        if i > 0:
                            g_comb = gx
                        elif role == 'head':
                            g_raw, g_act = None, gx
                        else:
                            g_raw, g_act = (res_pending if role == 'c1' else g_here), gx
                        continue
                    plan.append(('b', G1, G2, am, slope, wb, inp, Ca, Cb, k, pad, dil, gx, g_out, ep_mask,
                                 float(steps[i - 1][7] if i > 0 else 1.0), ep_res, gw, gbp, S))
                gv, gg, gb = grad_out(i, v32, g32, Cb)
                descs[i] = struct.pack('<7Q6i', gw.data_ptr(), gbp.data_ptr(), v32.data_ptr(), g32.data_ptr(), gv.data_ptr(),
                                       gg.data_ptr(), gb.data_ptr(), nsec * S, Cout, Cin, k, Cb, Ca)
                own_launch.append((plan[-1], i))               # its slabs exist once that plan entry has run
                keep += [gw, gbp, G1, G2, g_out]
                grads[3 * i], grads[3 * i + 1] = gv, gg
                grads[3 * i + 2] = gb[:Cout] if has_bias else None
                if role == 'c2':
                    res_pending = g_here
                if i > 0:
                    g_comb = gx
                elif gx is None:
                    pass
                elif role == 'head':                           # gradient wrt the chain's input buffer
                    g_raw, g_act = None, gx
                else:                                          # gradients wrt the block's inputs (x, xa)
                    g_raw, g_act = (res_pending if role == 'c1' else g_here), gx

            if lag is not None:                # the chain's first pair: its conv1 weight gradient has no later pair launch to ride on
                plan.append(('pbflush', lag))
            if batch:
                plan = _chain_pairs_bwd(plan, shape.N * shape.Lp)
            used = []

            def run(h, entries=None):
                st = stream_ptr(dev)
                q = lambda t: ptr(_sec(t, h, nsec))            # noqa: E731
                main = torch.cuda.current_stream(dev)
                for e in (plan if entries is None else entries):
                    if e[0] == 'side':                         # a weight gradient on the next side stream, behind what is enqueued so far
                        sd = wstreams[len(used) % len(wstreams)]
                        used.append(sd)
                        ev = torch.cuda.Event()
                        ev.record(main)
                        sd.wait_event(ev)
                        _, _, G1, G2, am, slope, inp, Ca, Cb, k, off0, dil, gw, gbp, S = e
                        with torch.cuda.stream(sd):
                            check(lib().psnd_conv1d_cl_wgrad(ptr(G1), ptr(G2), ptr(am), float(slope), ptr(inp), shape.N, shape.Lp, Ca, Cb, k,
                                                             off0, dil, ptr(gw), ptr(gbp), None, stream_ptr(dev)), 'psnd_conv1d_cl_wgrad')
                        continue
                    if e[0] == 'pb':
                        _, G, wb2, m1, m1s, g_h, wb1, m2, m2s, C, k, pad2, dil2, pad1, d1, gx, gwa, gba, lg = e
                        lb = lg if lg is not None else (None, None, 0, 0, None, None, C, k)
                        check(lib().psnd_conv1d_cl_pair_bwd(ptr(G), ptr(wb2), ptr(m1), m1s, ptr(g_h), ptr(wb1), ptr(m2), m2s, ptr(G), shape.N,
                                                            shape.Lp, shape.L, shape.HP, C, k, pad2, dil2, pad1, d1, ptr(gx), ptr(m1), ptr(gwa),
                                                            ptr(gba), ptr(lb[0]), ptr(lb[1]), lb[2], lb[3], ptr(lb[4]), ptr(lb[5]), st),
                              'psnd_conv1d_cl_pair_bwd')
                        continue
                    if e[0] == 'pbflush':
                        lg = e[1]
                        check(lib().psnd_conv1d_cl_pair_bwd(None, None, None, 1.0, None, None, None, 1.0, None, shape.N, shape.Lp, shape.L,
                                                            shape.HP, lg[6], lg[7], 0, 1, 0, 1, None, None, None, None, ptr(lg[0]), ptr(lg[1]),
                                                            lg[2], lg[3], ptr(lg[4]), ptr(lg[5]), st), 'psnd_conv1d_cl_pair_bwd')
                        continue
                    if e[0] == 'chainb':
                        pairs = e[1]
                        arr = (_lib.ChainPair * len(pairs))()
                        for d, (_, G, wb2, m1, m1s, g_h, wb1, m2, m2s, res, C, k, off1, ds1, off2, ds2, gx) in zip(arr, pairs):
                            d.W1, d.bias1, d.act1_slope, d.mid_out = wb2.data_ptr(), None, 1.0, g_h.data_ptr()
                            d.W2, d.bias2, d.off1, d.dstep1, d.off2, d.dstep2 = wb1.data_ptr(), None, off1, ds1, off2, ds2
                            d.act2_slope, d.out_raw, d.out_act = 1.0, gx.data_ptr(), None
                            d.M1, d.M2, d.m1_slope, d.m2_slope = m1.data_ptr(), m2.data_ptr(), m1s, m2s
                        G0 = pairs[0][1]
                        check(lib().psnd_conv1d_cl_chain(ptr(G0), ptr(G0), ctypes.addressof(arr), len(pairs), shape.N, shape.Lp, shape.L,
                                                         shape.HP, pairs[0][10], pairs[0][11], st), 'psnd_conv1d_cl_chain')
                        continue
                    if e[0] == 'pairb':
                        _, G, wb2, m1, m1s, g_h, wb1, m2, m2s, res, C, k, off1, ds1, off2, ds2, gx = e
                        check(lib().psnd_conv1d_cl_pair(ptr(G), ptr(wb2), None, ptr(m1), m1s, 1.0, ptr(g_h), ptr(wb1), None, ptr(m2), m2s,
                                                        ptr(res), shape.N, shape.Lp, shape.L, shape.HP, C, k, off1, ds1, off2, ds2, 1.0,
                                                        ptr(gx), None, st), 'psnd_conv1d_cl_pair')
                        continue
                    if e[0] == 'w':
                        _, G1, G2, am, slope, inp, Ca, Cb, k, off0, dil, gw, gbp, S = e
                        check(lib().psnd_conv1d_cl_wgrad(q(G1), q(G2), q(am), float(slope), q(inp), Nh, shape.Lp, Ca, Cb, k, off0, dil,
                                                         ptr(gw[h * S:(h + 1) * S]), ptr(gbp[h * S:(h + 1) * S]), None, st),
                              'psnd_conv1d_cl_wgrad')
                    else:
                        _, G1, G2, am, slope, wb, inp, Ca, Cb, k, pad, dil, gx, g_out, ep_mask, ep_slope, ep_res, gw, gbp, S = e
                        check(lib().psnd_conv1d_cl_bwd(q(G1), q(G2), q(am), float(slope), ptr(wb), q(inp), Nh, shape.Lp, shape.L,
                                                       shape.HP, Ca, Cb, k, pad, dil, q(gx), q(g_out), q(ep_mask), ep_slope, q(ep_res),
                                                       None if gw is None else ptr(gw[h * S:(h + 1) * S]),
                                                       None if gbp is None else ptr(gbp[h * S:(h + 1) * S]), st),
                              'psnd_conv1d_cl_bwd')

            st = stream_ptr(dev)
            handed = set()

            def wgrad_batch(part):
                """the weight gradients of these convs (wbatch entries) as psnd_conv1d_cl_wgrad_multi launches of <= 32"""
                for j0 in range(0, len(part), 32):
                    sub = part[j0:j0 + 32]
                    # row ranges: what suits the most frequent shape of the launch (the body's 256 -> 256 convs); the others take the same
                    shp = [(steps[ci][3], steps[ci][4], steps[ci][2]) for ci, _, _, _, _ in sub]
                    main_shape = max(set(shp), key=shp.count)
                    # (a chunk of the hand-over: row ranges as if it were at least half of all the convs - the weight-norm backward
                    #  has to add every range's slab up)
                    cnt = max(shp.count(main_shape), len(wbatch) // 2) if len(part) < len(wbatch) else shp.count(main_shape)
                    Sb = lib().psnd_conv1d_cl_wgrad_multi_splits(shape.N, shape.Lp, main_shape[0], main_shape[1], main_shape[2], cnt)
                    arr = (_lib.WgradDesc * len(sub))()
                    for d, (ci, G, inp, off0, dstep) in zip(arr, sub):
                        w = slabs(ci, G, None, None, inp, Sb)
                        d.g, d.xa, d.gw_part, d.gbias_part, d.off0, d.dstep = G.data_ptr(), inp.data_ptr(), w[11].data_ptr(), w[12].data_ptr(), off0, dstep
                        d.Ca, d.Cb, d.k, d.splits = steps[ci][3], steps[ci][4], steps[ci][2], Sb
                    check(lib().psnd_conv1d_cl_wgrad_multi(ctypes.addressof(arr), len(sub), shape.N, shape.Lp, st), 'psnd_conv1d_cl_wgrad_multi')

            def finish(convs):
                """weight-norm backward of these convs (their slabs are complete on this stream), then the hand-over of what went straight
                into the sink's buffers"""
                ds = [descs[ci] for ci in convs]
                for j0 in range(0, len(ds), 32):                 # PSND_WNORM_MAX descriptors per launch
                    chunk = ds[j0:j0 + 32]
                    buf = ctypes.create_string_buffer(b''.join(chunk))
                    check(lib().psnd_conv1d_wnorm_bwd_multi(buf, len(chunk), st), 'psnd_conv1d_wnorm_bwd_multi')
                out = []
                for ci in convs:
                    for ent in early.get(ci, ()):
                        if ent is not None:
                            grads[3 * ci + ent[0]] = None        # in place already: nothing for autograd to accumulate
                            out.append(ent[1])
                    handed.add(ci)
                if out:
                    sink.deliver(out)

            if sink is not None and batch and nsec == 1 and not wstreams:
                # chunked: behind every input-gradient launch the weight gradients of the convs it completed, their weight-norm
                # backward and their hand-over - the reducer's buckets fill (and leave) block by block, last block first
                done_w = 0

                def chunk(part, ready):
                    if part:
                        wgrad_batch(part)
                    finish(ready)

                launches = [e for e in plan if (e[1][-1] if e[0] == 'chainb' else e)[0] == 'pairb']
                L, C = len(launches), max(1, int(HANDOVER_CHUNKS))
                cut = {id(launches[min(L - 1, max(0, round((j + 1) * L / C) - 1))]) for j in range(C - 1)} if L else set()
                ready = []
                for e in plan:
                    run(0, [e])
                    ready += [ci for pe, ci in own_launch if pe is e]
                    last = e[1][-1] if e[0] == 'chainb' else e
                    upto = flush_after.get(id(last[16])) if (last[0] == 'pairb' and id(e) in cut) else None
                    if upto is None:
                        continue                                  # no hand-over point behind this launch (own-launch convs wait for the next)
                    part = []
                    if upto is not None and upto > done_w:
                        part = wbatch[done_w:upto]
                        ready += [ci for ci, _, _, _, _ in part]
                        done_w = upto
                    if ready:
                        chunk(part, ready)
                        ready = []
                part = wbatch[done_w:]
                rest = [ci for ci in range(n) if ci not in handed and descs[ci] is not None and ci not in [c for c, _, _, _, _ in part]]
                rest = [ci for ci in rest if ci not in ready]
                if part or rest or ready:
                    chunk(part, ready + rest + [ci for ci, _, _, _, _ in part])
            else:
                _run_sections(dev, nsec, sides, run)
                for sd in set(used):
                    torch.cuda.current_stream(dev).wait_stream(sd)
                # (measured, not kept: these two on the parameter stream as ConvTransposeCL.backward does - config-3 step 3.04 -> 3.10-3.15 ms,
                #  config-2 step 0.666 -> 0.669 ms: the resblocks' weight-norm backward launches queue up behind each other on that stream.
                #  Likewise at the config-2 size: the input layout change next to the weight prep + the backward packs next to the loss
                #  node's backward, 0.655-0.660 -> 0.668-0.679 ms and one 3.3 ms block in 18 - tools/r04/ab_c2.sh)
                wgrad_batch(wbatch)
                finish([ci for ci in range(n) if descs[ci] is not None])
        consume_param_use(ctx)
        return (g_raw, g_act, None, None, None, None, None, None) + tuple(grads)


def _block_node(convs, x, xa, shape, pairs, last_act_slope, want_raw, prep):
    params = []
    for c in convs:
        params += [c.weight_v, c.weight_g, c.bias]
    packs = None if prep is None else _PackList(prep[id(c)] for c in convs)
    if packs is not None:
        packs.owner = prep if isinstance(prep, PrepPacks) else None
    roles = pairs if isinstance(pairs, tuple) else (('c1', 'c2') * (len(convs) // 2) if pairs else ('r2',) * len(convs))
    return ResBlockCL.apply(x, xa, shape, roles, tuple(c.dilation for c in convs), last_act_slope, want_raw, packs, *params)


class _PackList(list):
    owner = None                               # the PrepPacks the entries come from (deferred backward packs)


def _use_block_node(convs, x):
    import os
    return (_sw.lab('PSND_NO_BLOCK_NODE') != '1' and x is not None and len(convs) <= 64
            and all(c.weight_v.shape[0] == c.weight_v.shape[1] for c in convs))


def conv_body_cl(head, blocks, tail, x0, shape, prep=None):
    """head conv -> ResBlock1 blocks -> tail conv (the separator's whole body: conv_pre, blocks, conv_post) as ONE autograd node.
    x0: activated CL input of the head conv.  Returns the tail conv's raw output."""
    import os
    stack = [c for b in blocks for pair in zip(b.convs1, b.convs2) for c in pair]
    if (_sw.lab('PSND_NO_BODY_NODE') == '1' or _sw.lab('PSND_NO_BLOCK_NODE') == '1' or not _stack_enabled()
            or NODE_GRANULARITY == 'block'):
        if isinstance(prep, PrepPacks):
            prep.write_backward_packs()                    # per-conv nodes keep their backward pack from the forward
        x, xa = fused_conv(x0, head, shape, None, True, True, 0.1, prep)
        x, xa = resblock1_stack_cl(blocks, x, xa, shape, prep=prep)
        return fused_conv(xa, tail, shape, None, True, False, prep=prep)[0]
    convs = [head] + stack + [tail]
    roles = ('head',) + ('c1', 'c2') * (len(stack) // 2) + ('tail',)
    return _block_node(convs, None, x0, shape, roles, 0.1, True, prep)[0]


def resblock1_cl(block, x, xa, shape, last_act_slope=0.1, want_raw=True, prep=None):
    """ResBlock1 (hifi_gan.py:55-63) on CL buffers.  x: residual stream (raw), xa = leaky_relu(x, 0.1).
    Returns (x_out raw or None, leaky_relu(x_out, last_act_slope))."""
    n = len(block.convs1)
    convs = [c for pair in zip(block.convs1, block.convs2) for c in pair]
    if _use_block_node(convs, x):
        return _block_node(convs, x, xa, shape, True, last_act_slope, want_raw, prep)
    for i, (c1, c2) in enumerate(zip(block.convs1, block.convs2)):
        _, ta = fused_conv(xa, c1, shape, None, False, True, 0.1, prep)
        last = i == n - 1
        x, xa = fused_conv(ta, c2, shape, x, (not last) or want_raw, True, last_act_slope if last else 0.1, prep)
    return x, xa


def resblock1_stack_cl(blocks, x, xa, shape, last_act_slope=0.1, want_raw=True, prep=None):
    """consecutive ResBlock1 blocks on one residual stream (the separator's body) as ONE autograd node: the chain of
    (conv1, conv2 + residual) pairs simply continues across the block boundaries."""
    convs = [c for b in blocks for pair in zip(b.convs1, b.convs2) for c in pair]
    if _use_block_node(convs, x) and _stack_enabled():
        return _block_node(convs, x, xa, shape, True, last_act_slope, want_raw, prep)
    for i, b in enumerate(blocks):
        last = i == len(blocks) - 1
        x, xa = resblock1_cl(b, x, xa, shape, last_act_slope if last else 0.1, want_raw or not last, prep)
    return x, xa


def _stack_enabled():
    import os
    return _sw.lab('PSND_NO_BLOCK_STACK') != '1' and NODE_GRANULARITY != 'block'       # one node per block instead


def resblock2_cl(block, x, xa, shape, last_act_slope=0.1, want_raw=True, prep=None):
    """ResBlock2 (hifi_gan.py:84-88) on CL buffers: x = conv(leaky_relu(x)) + x per conv."""
    n = len(block.convs)
    if _use_block_node(list(block.convs), x):
        return _block_node(list(block.convs), x, xa, shape, False, last_act_slope, want_raw, prep)
    for i, c in enumerate(block.convs):
        last = i == n - 1
        x, xa = fused_conv(xa, c, shape, x, (not last) or want_raw, True, last_act_slope if last else 0.1, prep)
    return x, xa
