"""Deferred tensors: the reference's call pattern on the fused kernels.

A user of pytorch_sound writes a masking recipe's step with torch ops on the model's output (README usage, trainer.py:58-73; the log-mel of a
magnitude is the reference's own three lines, transforms.py:235-243):

    est = model(mag_mix)                                                   # sigmoid(logits) * mag_mix
    mel_est = torch.log(torch.matmul(fe.mel_filter, est) + 1e-6).clamp(fe.min_db, fe.max_db)
    loss = F.l1_loss(est, mag_ref) + 0.5 * F.l1_loss(mel_est, mel_ref)

As written that is a library GEMM, ~15 elementwise / reduction launches forward and as many backward around tensors (`mel_est`, two gradient
tensors, their sum) that the fused node cl.MaskHeadSpectralL1CL never materialises (BENCH_r05 dropin_step: 0.82 ms against 0.67 ms for the same step
on the fused entry points).  Here the model hands out its estimate as a `Deferred` tensor - a torch.Tensor subclass that RECORDS the handful of
ops of that pattern instead of running them (`__torch_function__`):

    Est                      the model's estimate, not yet formed (logits in the channels-last layout + the mixture magnitude)
    matmul(mel_filter, Est)  -> MelLin     (mel_filter: the buffer of a LogMelSpectrogram of this package)
    MelLin + eps             -> MelLin     torch.log(MelLin) -> MelLog     MelLog.clamp(lo, hi) -> LogMel
    F.l1_loss(Est | LogMel, target)        -> Term       c * Term, Term + Term -> Sum

and resolves them when a real tensor is needed: `resolve()` (Trainer calls it on what `forward` returns), or ANY other torch function applied
to a Deferred - which first materialises its operands with the plain kernels (mask head, kernels.MelLog, kernels.l1_loss) and then runs as
written.  A Sum of one magnitude term and one log-mel term of the SAME estimate resolves to the fused node; every other shape resolves term by
term.  Nothing is approximated and nothing is skipped: an unrecognised use costs the unfused launches, never a different result.

The same mechanism carries the LAYOUT (round 6): `STFT.transform` on a HIP waveform hands out its magnitude as a `MagNFK` - the bin-fastest
(N, F, K) result of the kernel that runs at 0.62 of the HBM roofline (psnd_stft_mag_nfk; the reference's frame-fastest (N, K, F) is written at
0.48) - reporting the reference's shape (N, K, F).  The consumers of this library take it as it is (the separator's way into the channels-last
layout, the fused loss as target, the mel kernels); any other use transposes it once into a real (N, K, F) tensor.  The phase, which the
reference computes with every transform and its own LogMelSpectrogram discards (transforms.py:232), is a `Phase` node: computed from the waveform
on first use (an in-place write to the waveform in between is detected by its version counter and raises).
"""
import weakref

import torch
import torch.nn.functional as F

ENABLED = True            # False: models hand out plain tensors (the A/B of tests/test_gpu_deferred.py)
MEL_MODULES = weakref.WeakSet()      # LogMelSpectrogram instances: `mel_filter` buffers that matmul(mel_filter, Est) recognises


# ---- expression nodes ---------------------------------------------------------------------------------------------------------------
class _Node:
    shape = ()
    device = None
    requires_grad = False
    _real = None

    def real(self):
        if self._real is None:
            self._real = self._materialize()
        return self._real


class MagNFK(_Node):
    """an STFT magnitude held bin-fastest: `nfk` (N, F, K) fp32, no gradient; stands for the reference's (N, K, F) tensor"""

    def __init__(self, nfk, owner=None):
        self.nfk, self.owner = nfk, owner
        N, Fr, K = nfk.shape
        self.shape, self.device, self.requires_grad = (N, K, Fr), nfk.device, False

    def _materialize(self):
        # a consumer outside the library: one transposing pass - and the module that produced it stops deferring (its next transforms write
        # the reference's layout directly: a consumer that wants (N, K, F) should not pay for two passes every step)
        if self.owner is not None:
            self.owner._lazy_transform = False
        return self.nfk.transpose(1, 2).contiguous()


class Phase(_Node):
    """atan2(im, re) of STFT.transform (detached in the reference, transforms.py:69), not yet computed"""

    def __init__(self, wav, module, shape):
        self.wav, self.version, self.module = wav, wav._version, module
        self.shape, self.device, self.requires_grad = tuple(shape), wav.device, False

    def _materialize(self):
        if self.wav._version != self.version:
            raise RuntimeError('the waveform handed to STFT.transform was modified in place before the first use of the phase it returned '
                               '(the phase is computed on first use: pytorch_sound_amd/deferred.py; deferred.ENABLED = False computes it at once)')
        self.module._lazy_transform = False                  # the phase IS used: this module's next transforms compute both at once
        return self.module._transform_now(self.wav)[1]


class Est(_Node):
    """sigmoid(from_cl(y)) * mag of a masking model (ConvSeparator): y = logits, channels-last bf16; mag (N, K, F) fp32 - or, `mag_node` given,
    the bin-fastest magnitude of a MagNFK (then y was formed from it on the plain-stream way in, and the fused loss runs bin-fastest too)"""

    def __init__(self, y, mag, shape, mag_node=None):
        self.y, self.mag, self.cl_shape, self.mag_node = y, mag, shape, mag_node
        self.shape = tuple(mag.shape) if mag_node is None else mag_node.shape
        self.device, self.requires_grad = y.device, y.requires_grad

    def _materialize(self):
        from . import cl
        mag = self.mag if self.mag_node is None else self.mag_node.real()
        return cl.MaskHeadCL.apply(self.y, mag, self.cl_shape)


class MelLin(_Node):
    """mel_filter @ est (+ eps), log taken or not"""

    def __init__(self, est, module, eps=0.0, logged=False):
        self.est, self.module, self.eps, self.logged = est, module, eps, logged
        N, _, T = est.shape
        self.shape, self.device, self.requires_grad = (N, module.mel_filter.shape[0], T), est.device, est.requires_grad

    def _materialize(self):
        x = torch.matmul(self.module.mel_filter, self.est.real())
        if self.eps:
            x = x + self.eps
        return torch.log(x) if self.logged else x


class LogMel(_Node):
    """clamp(log(mel_filter @ est + eps), lo, hi): LogMelSpectrogram's own arithmetic on a magnitude (transforms.py:235-243)"""

    def __init__(self, lin, lo, hi):
        self.est, self.module, self.eps, self.lo, self.hi = lin.est, lin.module, lin.eps, lo, hi
        self.shape, self.device, self.requires_grad = lin.shape, lin.device, lin.requires_grad

    def _materialize(self):
        m = self.module
        if isinstance(self.est, MagNFK):                                       # the log-mel of a bin-fastest magnitude: the (N, F, K) mel kernel
            from . import kernels as K
            return K.MelLogNfk.apply(self.est.nfk, m._mel_plan(), m.mel_filter.shape[0], K.LOG_E, float(self.eps), None, self.lo, self.hi)
        x = self.est.real()
        if not x.is_cuda:                                                     # host tensors: the ops as written
            y = torch.log(torch.matmul(m.mel_filter, x) + self.eps)
            return y if self.lo is None and self.hi is None else y.clamp(self.lo, self.hi)
        from . import kernels as K
        return K.MelLog.apply(x, m._mel_plan(), m.mel_filter.shape[0], K.LOG_E, float(self.eps), None, self.lo, self.hi)


class Sum(_Node):
    """sum_i w_i * F.l1_loss(node_i, target_i) - a scalar"""

    def __init__(self, terms):
        self.terms = terms                               # [(weight, node, target)]
        self.device = terms[0][1].device
        self.requires_grad = any(t[1].requires_grad for t in terms)

    def _fused(self):
        if len(self.terms) != 2:
            return None
        mag_t = [t for t in self.terms if isinstance(t[1], Est)]
        mel_t = [t for t in self.terms if isinstance(t[1], LogMel)]
        if len(mag_t) != 1 or len(mel_t) != 1 or mel_t[0][1].est is not mag_t[0][1]:
            return None
        (w1, est, mag_ref), (w2, lm, mel_ref) = mag_t[0], mel_t[0]
        from . import cl, kernels as K
        m = lm.module
        plain = lambda t: isinstance(t, torch.Tensor) and not isinstance(t, Deferred) and t.dtype == torch.float32 and t.is_cuda and not t.requires_grad  # noqa: E731
        if est._real is not None or lm._real is not None or not plain(mel_ref) or tuple(mel_ref.shape) != lm.shape:
            return None
        ref_node = _node(mag_ref, MagNFK)
        if est.mag_node is not None and ref_node is not None and ref_node.shape == est.shape and ref_node._real is None:
            # bin-fastest end to end: mixture magnitude, target and estimate are (N, F, K)
            loss, est_nfk = cl.MaskHeadSpectralL1NFK.apply(est.y, est.mag_node.nfk, ref_node.nfk, mel_ref.contiguous(), m._mel_plan(), est.cl_shape,
                                                           m.mel_filter.shape[0], K.LOG_E, float(lm.eps), None, lm.lo, lm.hi, float(w1), float(w2))
            est_real = est_nfk.detach().transpose(1, 2)    # the reference's (N, K, F) as a view
        else:
            mag = est.mag if est.mag_node is None else est.mag_node.real()
            ref = resolve(mag_ref)
            if not (plain(mag) and plain(ref) and tuple(ref.shape) == est.shape):
                return None
            loss, est_real = cl.MaskHeadSpectralL1CL.apply(est.y, mag.contiguous(), ref.contiguous(), mel_ref.contiguous(), m._mel_plan(), est.cl_shape,
                                                           m.mel_filter.shape[0], K.LOG_E, float(lm.eps), None, lm.lo, lm.hi, float(w1), float(w2))
            est_real = est_real.detach()
        loss.psnd_nan_flag = cl.LAST_LOSS_NAN_FLAG[0]      # isnan(loss), written by the launch that formed the loss (Trainer._nan_flag)
        est._real = est_real                               # for logging / metrics: carries no gradient of its own (the loss node holds it)
        return loss

    def _materialize(self):
        out = self._fused()
        if out is not None:
            return out
        from . import kernels as K
        tot = None
        for w, node, target in self.terms:
            x, target = node.real(), resolve(target)
            t = K.l1_loss(x, target) if (x.is_cuda and x.dtype == torch.float32 and target.dtype == torch.float32 and x.shape == target.shape) \
                else F.l1_loss(x, target)
            t = t if w == 1.0 else t * w
            tot = t if tot is None else tot + t
        return tot


# ---- the tensor subclass ------------------------------------------------------------------------------------------------------------
def _is_num(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


class Deferred(torch.Tensor):
    @staticmethod
    def __new__(cls, node):
        t = torch.Tensor._make_wrapper_subclass(cls, node.shape, dtype=torch.float32, device=node.device, requires_grad=False)
        t._node = node
        return t

    def __repr__(self):
        return 'Deferred(%s, shape=%s)' % (type(self._node).__name__, tuple(self._node.shape))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # every use is resolved at the torch-function level (below): an op that reaches the dispatcher with an unresolved operand came in
        # around it - fail loudly rather than compute without the autograd history the real tensor carries
        raise RuntimeError('deferred tensor reached the dispatcher unresolved in %s: call pytorch_sound_amd.deferred.resolve() on it first' % (func,))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, '__name__', '')
        if name == '__get__' or func in _METADATA:           # shape / dtype / device / dim() ...: the wrapper's own metadata
            prop = getattr(getattr(func, '__self__', None), '__name__', '')
            if func in _METADATA or prop in _META_PROPS:
                if prop == 'requires_grad':
                    return args[0]._node.requires_grad
                with torch._C.DisableTorchFunctionSubclass():
                    return func(*args, **kwargs)
        h = _HANDLERS.get(func)
        if h is not None:
            out = h(*args, **kwargs)
            if out is not NotImplemented:
                return out
        args, kwargs = torch.utils._pytree.tree_map(resolve, (args, kwargs))
        return func(*args, **kwargs)


_META_PROPS = {'shape', 'dtype', 'device', 'ndim', 'is_cuda', 'layout', 'is_sparse', 'is_quantized', 'is_meta', 'requires_grad', 'is_leaf', 'names',
               'is_cpu', 'is_mkldnn', 'is_nested'}
_METADATA = {torch.Tensor.size, torch.Tensor.dim, torch.Tensor.numel, torch.Tensor.ndimension, torch.Tensor.nelement, torch.Tensor.is_floating_point,
             torch.Tensor.is_complex, torch.Tensor.element_size, torch.Tensor.__len__, torch.Tensor.get_device}


def resolve(x):
    """the real tensor behind a Deferred (anything else is returned as it is)"""
    return x._node.real() if isinstance(x, Deferred) else x


def est(y, mag, shape, mag_node=None):
    """what a masking model returns for est = sigmoid(from_cl(y)) * mag when ENABLED (models/separator.py)"""
    return Deferred(Est(y, mag, shape, mag_node))


def mag_nfk(nfk, owner=None):
    """STFT.transform's magnitude, held bin-fastest (N, F, K), standing for the reference's (N, K, F)"""
    return Deferred(MagNFK(nfk, owner))


def nfk_of(x):
    """the MagNFK node behind `x` if it is an un-transposed lazy magnitude, else None"""
    n = _node(x, MagNFK)
    return n if n is not None and n._real is None else None


def _node(x, kinds):
    return x._node if isinstance(x, Deferred) and isinstance(x._node, kinds) else None


def _h_float(x, *a, **k):
    return x if not a and not k else NotImplemented


def _h_to(x, *a, **k):
    if not k and len(a) == 1 and a[0] is torch.float32:
        return x
    return NotImplemented


def _h_matmul(a, b, *rest, **k):
    e = _node(b, (Est, MagNFK))
    if e is None or rest or k or isinstance(a, Deferred) or not isinstance(a, torch.Tensor) or a.dim() != 2:
        return NotImplemented
    for m in MEL_MODULES:
        if m.mel_filter is a and a.device == e.device and a.dtype == torch.float32 and a.shape[1] == e.shape[1]:
            return Deferred(MelLin(e, m))
    return NotImplemented


def _h_add(a, b, *rest, **k):
    if rest or (k and (set(k) != {'alpha'} or k['alpha'] != 1)):
        return NotImplemented
    for x, c in ((a, b), (b, a)):
        lin = _node(x, MelLin)
        if lin is not None and not lin.logged and _is_num(c):
            return Deferred(MelLin(lin.est, lin.module, lin.eps + float(c), False))
    sa, sb = _node(a, Sum), _node(b, Sum)
    if sa is not None and sb is not None:
        return Deferred(Sum(sa.terms + sb.terms))
    return NotImplemented


def _h_log(x, *a, **k):
    lin = _node(x, MelLin)
    if lin is None or lin.logged or a or k:
        return NotImplemented
    return Deferred(MelLin(lin.est, lin.module, lin.eps, True))


def _h_clamp(x, min=None, max=None, **k):
    lin = _node(x, MelLin)
    if lin is None or not lin.logged or k or not all(v is None or _is_num(v) for v in (min, max)):
        return NotImplemented
    return Deferred(LogMel(lin, None if min is None else float(min), None if max is None else float(max)))


def _h_l1(input, target, *a, **k):
    n = _node(input, (Est, LogMel))
    if n is None or a or not isinstance(target, torch.Tensor) or (isinstance(target, Deferred) and _node(target, MagNFK) is None):
        return NotImplemented
    if any(k.get(key) is not None for key in ('size_average', 'reduce')) or k.get('reduction', 'mean') != 'mean' or k.get('weight') is not None:
        return NotImplemented
    if tuple(target.shape) != tuple(n.shape):
        return NotImplemented
    return Deferred(Sum([(1.0, n, target)]))


def _h_mul(a, b, *rest, **k):
    if rest or k:
        return NotImplemented
    for x, c in ((a, b), (b, a)):
        s = _node(x, Sum)
        if s is not None and _is_num(c):
            return Deferred(Sum([(w * float(c), n, t) for w, n, t in s.terms]))
    return NotImplemented


def _h_div(a, b, *rest, **k):
    s = _node(a, Sum)
    if s is None or rest or k or not _is_num(b) or b == 0:
        return NotImplemented
    return Deferred(Sum([(w / float(b), n, t) for w, n, t in s.terms]))


_HANDLERS = {
    torch.Tensor.float: _h_float, torch.Tensor.to: _h_to, torch.Tensor.contiguous: _h_float,
    torch.matmul: _h_matmul, torch.Tensor.matmul: _h_matmul, torch.Tensor.__matmul__: _h_matmul, torch.Tensor.__rmatmul__: lambda b, a: _h_matmul(a, b),
    torch.add: _h_add, torch.Tensor.add: _h_add, torch.Tensor.__add__: _h_add, torch.Tensor.__radd__: _h_add,
    torch.log: _h_log, torch.Tensor.log: _h_log,
    torch.clamp: _h_clamp, torch.Tensor.clamp: _h_clamp,
    F.l1_loss: _h_l1,
    torch.mul: _h_mul, torch.Tensor.mul: _h_mul, torch.Tensor.__mul__: _h_mul, torch.Tensor.__rmul__: _h_mul,
    torch.div: _h_div, torch.Tensor.div: _h_div, torch.Tensor.__truediv__: _h_div, torch.true_divide: _h_div,
}


# ---- custom autograd Functions --------------------------------------------------------------------------------------------------------
# torch.autograd.Function.apply does not consult __torch_function__: handed a Deferred it would run `forward` on the wrapper - the numbers
# right (every op inside resolves it) but WITHOUT a gradient edge, the wrapper itself having no history.  Every apply therefore resolves
# deferred operands first (also for Functions defined outside this package: a silent loss of gradients is not an acceptable way to fail).
def _patch_function_apply():
    base = torch.autograd.Function
    if getattr(base.apply, '_psnd_resolves_deferred', False):
        return
    orig = base.apply.__func__

    def apply(cls, *args, **kwargs):
        if any(isinstance(a, Deferred) for a in args):
            args = tuple(resolve(a) for a in args)
        if kwargs and any(isinstance(a, Deferred) for a in kwargs.values()):
            kwargs = {k: resolve(v) for k, v in kwargs.items()}
        return orig(cls, *args, **kwargs)

    apply._psnd_resolves_deferred = True
    apply.__doc__ = orig.__doc__
    base.apply = classmethod(apply)


_patch_function_apply()
