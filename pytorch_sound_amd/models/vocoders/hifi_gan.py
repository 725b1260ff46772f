"""HiFi-GAN generator: the Conv1d / ConvTranspose1d stack of pytorch_sound/models/vocoders/hifi_gan.py.

Checkpoint layout is the reference's (every conv carries ``weight_g`` / ``weight_v`` [/ ``bias``], as
the old-style ``weight_norm`` hook produces; shipped ``hifi_gan_v2.pt`` loads unchanged).  Semantics
pinned by tests/golden/hifigan.npz:
  * weight norm over dim 0: w = g * v / ||v||, norm per output channel for Conv1d and per INPUT channel
    for ConvTranspose1d (its dim 0), recomputed every forward
  * ResBlock1: 3 x [lrelu(0.1) -> dilated conv -> lrelu(0.1) -> conv -> + x]  (hifi_gan.py:55-63)
  * ResBlock2: 2 x [lrelu(0.1) -> dilated conv -> + x]                         (hifi_gan.py:83-89)
  * Generator: conv_pre(80 -> C, k7) ; per stage lrelu(0.1) -> ConvTranspose1d(k, stride u,
    pad (k-u)//2) -> mean of the stage's resblocks ; then leaky_relu with the DEFAULT slope 0.01
    (hifi_gan.py:134), conv_post(C -> 1, k7), tanh.
"""
from argparse import Namespace

import os
from pytorch_sound_amd import _switches as _sw
import torch
import torch.nn as nn
import torch.nn.functional as F

from pytorch_sound_amd.models import register_model, register_model_architecture

LRELU_SLOPE = 0.1

# The plain (N, C, T) formulation below runs its convolutions on HIP fp32 tensors as exact-fp32 matrix-core GEMMs over the unfolded input
# (kernels.conv1d_f32 / conv_transpose1d_f32: psnd_im2col_f32 + psnd_linear1x1_*, the activation in front of a conv applied while
# unfolding) - the arithmetic of the reference's stack (hifi_gan.py:32-147 is fp32).  LIBRARY_CONVS = True (Generator.use_cl = False sets it
# for its call) keeps F.conv1d / F.conv_transpose1d there: the A/B against the library formulation.  CPU tensors always take the torch ops.
LIBRARY_CONVS = False


def _native_f32(x) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and not LIBRARY_CONVS


def wants_bf16(precision, x) -> bool:
    """which arithmetic a HIP input of a conv stack gets: 'bf16' / 'fp32' as named; 'auto': bf16 conv operands (fp32 accumulation, the
    channels-last kernels) under torch.autocast or for a bf16 / fp16 input, the reference's fp32 convolutions for an fp32 input outside it"""
    if precision == 'bf16':
        return True
    if precision == 'fp32':
        return False
    if precision != 'auto':
        raise ValueError("precision must be 'auto', 'bf16' or 'fp32', got %r" % (precision,))
    return x.dtype in (torch.bfloat16, torch.float16) or torch.is_autocast_enabled('cuda')


def _pre_act(x, slope):
    return x if slope is None else F.leaky_relu(x, slope)


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    return int((kernel_size * dilation - dilation) / 2)


class _WNConvBase(nn.Module):
    """Conv with an explicit weight-norm parametrisation (parameters ``weight_g``, ``weight_v``)."""

    def _init_params(self, weight_shape, bias_features, std):
        v = torch.empty(weight_shape)
        if std is None:
            # nn.Conv1d default init (kaiming_uniform(a=sqrt(5))) for layers the reference leaves alone
            nn.init.kaiming_uniform_(v, a=5 ** 0.5)
        else:
            v.normal_(0.0, std)
        self.weight_v = nn.Parameter(v)
        self.weight_g = nn.Parameter(self._norm(v).detach().clone())
        fan_in = weight_shape[1] * weight_shape[2]
        bound = 1.0 / fan_in ** 0.5 if fan_in > 0 else 0.0
        self.bias = nn.Parameter(torch.empty(bias_features).uniform_(-bound, bound))
        self._folded = False

    @staticmethod
    def _norm(v):
        return v.flatten(1).norm(dim=1).view(-1, 1, 1)

    def effective_weight(self):
        if self._folded:
            return self.weight
        return self.weight_v * (self.weight_g / self._norm(self.weight_v))

    def remove_weight_norm(self):
        """fold g * v/||v|| into a plain ``weight`` parameter (inference; hifi_gan.py:65-69)."""
        if self._folded:
            return
        w = self.effective_weight().detach()
        del self.weight_g
        del self.weight_v
        self.weight = nn.Parameter(w)
        self._folded = True
        # the gfx950 conv kernels keep working on the folded weight: (v, g) = (w, ||w||) reproduces w.  Non-persistent buffers -
        # the state dict holds `weight` / `bias` only, as the reference's does after remove_weight_norm.  They are DERIVED from
        # `weight` and re-derived whenever it changed (sync_folded: load_state_dict / .to() / in-place edits after the fold)
        self.register_buffer('weight_v', self.weight.detach().clone(), persistent=False)
        self.register_buffer('weight_g', self._norm(w).clone(), persistent=False)
        self._fold_key = (self.weight.data_ptr(), self.weight._version)

    def sync_folded(self):
        """folded module: make (weight_v, weight_g) = (weight, ||weight||) again if `weight` was replaced or written since
        the last call (a folded checkpoint loaded after remove_weight_norm, as interface/hifi_gan.py:66-80 allows).  In place:
        the buffers' addresses - the key of the prepared bf16 packs - do not move."""
        if not self._folded:
            return
        key = (self.weight.data_ptr(), self.weight._version)
        if key == self._fold_key and self.weight_v.device == self.weight.device:
            return
        with torch.no_grad():
            if self.weight_v.device != self.weight.device or self.weight_v.shape != self.weight.shape:
                self.weight_v = self.weight.detach().clone()
                self.weight_g = self._norm(self.weight.detach()).clone()
            else:
                self.weight_v.copy_(self.weight.detach())
                self.weight_g.copy_(self._norm(self.weight.detach()))
        self._fold_key = key


class WNConv1d(_WNConvBase):
    def __init__(self, cin, cout, kernel_size, dilation=1, padding=0, init_std=None):
        super().__init__()
        self.dilation, self.padding = dilation, padding
        self._init_params((cout, cin, kernel_size), cout, init_std)

    def forward(self, x, pre_slope=None):
        """conv(leaky_relu(x, pre_slope)) (pre_slope None: conv(x))"""
        if _native_f32(x):
            from pytorch_sound_amd import kernels as K
            return K.conv1d_f32(x, self.effective_weight(), self.bias, self.padding, self.dilation, 1.0 if pre_slope is None else pre_slope)
        return F.conv1d(_pre_act(x, pre_slope), self.effective_weight(), self.bias, 1, self.padding, self.dilation)


class WNConvTranspose1d(_WNConvBase):
    def __init__(self, cin, cout, kernel_size, stride, padding, init_std=None):
        super().__init__()
        self.stride, self.padding = stride, padding
        self._init_params((cin, cout, kernel_size), cout, init_std)
        # nn.ConvTranspose1d computes its bias bound from weight.size(1) * k = cout * k
        bound = 1.0 / (cout * kernel_size) ** 0.5
        with torch.no_grad():
            self.bias.uniform_(-bound, bound)

    def forward(self, x, pre_slope=None):
        if _native_f32(x):
            from pytorch_sound_amd import kernels as K
            return K.conv_transpose1d_f32(x, self.effective_weight(), self.bias, self.stride, self.padding, 1.0 if pre_slope is None else pre_slope)
        return F.conv_transpose1d(_pre_act(x, pre_slope), self.effective_weight(), self.bias, self.stride, self.padding)


class ResBlock1(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h = h
        self.convs1 = nn.ModuleList([
            WNConv1d(channels, channels, kernel_size, d, get_padding(kernel_size, d), init_std=0.01) for d in dilation])
        self.convs2 = nn.ModuleList([
            WNConv1d(channels, channels, kernel_size, 1, get_padding(kernel_size, 1), init_std=0.01) for _ in dilation])

    def forward(self, x):
        for c1, c2 in zip(self.convs1, self.convs2):
            xt = c1(x, LRELU_SLOPE)
            xt = c2(xt, LRELU_SLOPE)
            x = xt + x
        return x

    def remove_weight_norm(self):
        for c in list(self.convs1) + list(self.convs2):
            c.remove_weight_norm()


class ResBlock2(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.h = h
        self.convs = nn.ModuleList([
            WNConv1d(channels, channels, kernel_size, d, get_padding(kernel_size, d), init_std=0.01) for d in dilation])

    def forward(self, x):
        for c in self.convs:
            x = c(x, LRELU_SLOPE) + x
        return x

    def remove_weight_norm(self):
        for c in self.convs:
            c.remove_weight_norm()


@register_model('hifi_gan')
class Generator(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        c0 = h.upsample_initial_channel
        self.conv_pre = WNConv1d(80, c0, 7, 1, 3)
        block_cls = ResBlock1 if h.resblock == '1' else ResBlock2
        self.ups = nn.ModuleList([
            WNConvTranspose1d(c0 // (2 ** i), c0 // (2 ** (i + 1)), k, u, (k - u) // 2, init_std=0.01)
            for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(block_cls(h, ch, k, d))
        self.conv_post = WNConv1d(ch, 1, 7, 1, 3, init_std=0.01)

    # Which arithmetic a HIP input gets (`precision`):
    #   'auto' (default)  bf16 conv operands / fp32 accumulation on the channels-last kernels (forward_cl) under torch.autocast or for a
    #                     bf16 / fp16 input - the user asked for reduced precision; an fp32 input outside autocast gets the reference's own
    #                     arithmetic: fp32 convolutions (exact-fp32 matrix-core GEMMs, kernels.conv1d_f32), no silent narrowing
    #   'bf16'            the channels-last bf16 kernels whatever the input (explicit opt-in: 16x the matrix rate)
    #   'fp32'            the fp32 convolutions whatever the context
    precision = 'auto'

    def _wants_bf16(self, x) -> bool:
        return wants_bf16(self.precision, x)

    def forward(self, x):
        global LIBRARY_CONVS
        if self._cl_ok(x):
            if self._wants_bf16(x):
                if x.dtype != torch.float32:             # the CL kernels read fp32 mels and write fp32 audio: cast, do not fall back
                    return self.forward_cl(x.float()).to(x.dtype)
                return self.forward_cl(x)
            with torch.autocast('cuda', enabled=False):
                return self._forward_plain(x.float()).to(x.dtype)
        if x.is_cuda:                                    # use_cl = False: the library formulation (A/B switch)
            prev, LIBRARY_CONVS = LIBRARY_CONVS, True
            try:
                return self._forward_plain(x)
            finally:
                LIBRARY_CONVS = prev
        return self._forward_plain(x)

    def _forward_plain(self, x):
        x = self.conv_pre(x)
        for i, up in enumerate(self.ups):
            x = up(x, LRELU_SLOPE)
            stage = self.resblocks[i * self.num_kernels:(i + 1) * self.num_kernels]
            xs = stage[0](x)
            for block in stage[1:]:
                xs = xs + block(x)
            x = xs / self.num_kernels
        return torch.tanh(self.conv_post(x, 0.01))       # F.leaky_relu's default slope, as the reference (hifi_gan.py:134)

    # ---- gfx950 path: the whole generator on the channels-last bf16 implicit-GEMM kernels (pytorch_sound_amd/cl.py):
    #      conv_pre, every ResBlock conv and conv_post with leaky-relu / bias / residual / weight norm fused, the ConvTranspose1d
    #      upsamplers in polyphase form (cl.ConvTransposeCL: k = 2 * stride, no zero multiplied), the mean over a stage's
    #      resblocks + the next activation in one pass (cl.MeanActCL).  fp32 in / fp32 out, fp32 accumulation, bf16 between convs;
    #      the activations never leave the CL layout.  Upsamplers the polyphase kernel does not cover (odd stride, k != 2 * stride) run
    #      as a convolution over zero-spread rows on the same conv kernels (cl.conv_transpose_cl) - chosen automatically (round 5);
    #      cl_upsample = 'kernel' forces that form (parity tests).
    use_cl = True
    cl_upsample = 'polyphase'
    cl_branches = _sw.lab('PSND_HIFIGAN_BRANCHES', '1') == '1'      # a stage's resblocks on parallel streams (0: one stream, the A/B of tools/r04/ab_branches.sh)
    _CL_MAX_REACH = 40          # tap reach (k-1)/2*dilation the conv kernel's A-tile ring is sized for (25; 40 for k <= 7)

    def _all_convs(self):
        out = [self.conv_pre, self.conv_post] + list(self.ups)
        for b in self.resblocks:
            out += self._block_convs(b)
        return out

    @staticmethod
    def _block_convs(block):
        return list(block.convs1) + list(block.convs2) if hasattr(block, 'convs1') else list(block.convs)

    def _cl_ok(self, x) -> bool:
        """False only for CPU tensors (and for `use_cl = False`, the explicit A/B switch to the library formulation); a HIP input
        whose model the kernels do not cover RAISES in forward_cl - no silent library path for a tensor that lives on the GPU."""
        if not (self.use_cl and x.is_cuda):
            return False
        if x.dim() != 3 or not x.is_floating_point():
            raise RuntimeError('hifi_gan Generator expects a floating-point (N, 80, T) mel batch, got %s %s' % (x.dtype, tuple(x.shape)))
        for c in self._all_convs():
            c.sync_folded()
        return True

    def _check_cl(self):
        from pytorch_sound_amd._lib import PsndError
        for block in self.resblocks:
            for c in self._block_convs(block):
                k = c.weight_v.shape[2]
                if c.padding > self._CL_MAX_REACH or (c.padding > 25 and k > 7) or k > 16:
                    raise PsndError('hifi_gan: conv with k=%d, dilation %d (tap reach %d) is beyond the gfx950 conv kernels '
                                    '(reach <= 25, or <= 40 with k <= 7); use_cl = False selects the library formulation'
                                    % (k, c.dilation, c.padding))
        for u in self.ups:
            k = u.weight_v.shape[2]
            if k - 1 - u.padding < 0 or getattr(u, 'output_padding', 0) not in (0, (0,)):
                raise PsndError('hifi_gan: ConvTranspose1d(k=%d, stride=%d, padding=%d, output_padding=%s) is beyond the gfx950 kernels; '
                                'use_cl = False selects the library formulation' % (k, u.stride, u.padding, getattr(u, 'output_padding', 0)))

    def _polyphase_ok(self):
        """the polyphase kernel (psnd_convtr1d_*) needs k = 2 * stride, an even stride and padding = stride / 2 (an odd stride makes the
        reference's output T * stride + 1 samples long - padding rounds down - which its row map does not produce)"""
        for u in self.ups:
            k = u.weight_v.shape[2]
            if k != 2 * u.stride or (k - u.stride) % 2 != 0 or 2 * u.padding != k - u.stride:
                return False
        return True

    def forward_cl(self, x):
        from pytorch_sound_amd import cl
        self._check_cl()
        if self.cl_upsample not in ('polyphase', 'kernel'):
            raise ValueError("hifi_gan Generator.cl_upsample must be 'polyphase' or 'kernel', got %r" % (self.cl_upsample,))
        if self.cl_upsample != 'polyphase' or not self._polyphase_ok():
            if self.cl_upsample == 'polyphase' and not getattr(self, '_zero_spread_logged', False):
                self._zero_spread_logged = True          # once per model: the form costs stride - 1 of every stride products on zeros
                import logging
                logging.getLogger('pytorch_sound_amd').info(
                    'hifi_gan: upsamplers %s are outside the polyphase kernel (k = 2 * stride, even stride, padding = stride / 2): '
                    'running them as zero-spread convolutions', [(u.weight_v.shape[2], u.stride, u.padding) for u in self.ups])
            return self._forward_cl_zero_spread(x)
        N, _, T = x.shape
        nst = len(self.ups)
        convs = [self.conv_pre, self.conv_post] + [c for b in self.resblocks for c in self._block_convs(b)]
        branches = (self.cl_branches and cl.AUTO_SECTIONS and x.is_cuda and _sw.lab('PSND_CL_SECTIONS', 'auto') in ('auto', '1'))
        main = torch.cuda.current_stream(x.device)
        prep = cl.prep_all(self, convs)                            # all weight-norm packs of the Conv1d layers: one launch
        prep_up = cl.prep_all_convtr(self, list(self.ups))         # ... and of the upsamplers: one more (next to the first on the parameter
        #                                                            stream: measured, no difference - 2.97-3.07 against 3.00-3.09 ms)
        # halo of a stage's buffers = the widest tap reach of the convs that read them
        halo = [max(self.conv_pre.padding, 1)]
        for i in range(nst):
            stage = self.resblocks[i * self.num_kernels:(i + 1) * self.num_kernels]
            h = max([c.padding for b in stage for c in self._block_convs(b)] + [self.ups[i].padding, 1])
            halo.append(max(h, self.conv_post.padding) if i + 1 == nst else h)
        shape = cl.CLShape(N, T, halo[0])
        _, xa = cl.fused_conv(cl.ToCL.apply(x, shape, 0), self.conv_pre, shape, None, False, True, LRELU_SLOPE, prep)
        for i, up in enumerate(self.ups):
            last = i + 1 == nst
            out_shape = cl.CLShape(N, T * up.stride, halo[i + 1])
            x_raw, x_act = cl.ConvTransposeCL.apply(xa, up.weight_v, up.weight_g, up.bias, shape, out_shape, up.stride, up.padding,
                                                    LRELU_SLOPE, prep_up[id(up)])
            shape, T = out_shape, out_shape.L
            stage = self.resblocks[i * self.num_kernels:(i + 1) * self.num_kernels]
            rs = []
            # every resblock of the stage reads (x_raw, x_act): aliases whose gradients come back summed by ONE launch (cl.FanOutCL)
            fan = cl.FanOutCL.apply(x_raw, x_act, len(stage)) if (len(stage) > 1 and torch.is_grad_enabled() and x_raw.requires_grad) \
                else (x_raw, x_act) * len(stage)
            def run_block(bi):
                block = stage[bi]
                fn = cl.resblock1_cl if hasattr(block, 'convs1') else cl.resblock2_cl
                return fn(block, fan[2 * bi], fan[2 * bi + 1], shape, prep=prep)[0]
            if branches and len(stage) > 1:
                # the resblocks of a stage are independent: one stream each (inside the step graph: parallel branches, ONE fork and
                # join per stage and direction - autograd runs a node's backward on the stream of its forward)
                sides = cl.branch_streams(x.device, len(stage) - 1)
                for sd in sides:
                    sd.wait_stream(main)
                rs = [None] * len(stage)
                rs[0] = run_block(0)          # (which block goes first / stays on the main stream measures the same: 3.01-3.06 ms either way)
                for bi, sd in enumerate(sides, 1):
                    with torch.cuda.stream(sd):
                        rs[bi] = run_block(bi)
                for sd in sides:
                    main.wait_stream(sd)
            else:
                rs = [run_block(bi) for bi in range(len(stage))]
            xa = cl.MeanActCL.apply(0.01 if last else LRELU_SLOPE, *rs)      # last: F.leaky_relu's default slope, as the reference
        y, _ = cl.fused_conv(xa, self.conv_post, shape, None, True, False, prep=prep)
        return cl.FromCLTanh.apply(y, 1, T, shape)

    def _forward_cl_zero_spread(self, x):
        """every ConvTranspose1d as a convolution over zero-spread rows on the CL conv kernels (cl.conv_transpose_cl: any kernel size,
        stride and padding; (T - 1) * stride - 2 * padding + k output samples as hifi_gan.py:107-110 gives) - the form for upsamplers the
        polyphase kernel does not cover; s - 1 of every s products are zeros"""
        from pytorch_sound_amd import cl
        N, _, T = x.shape
        halo = max([self.conv_pre.padding, self.conv_post.padding] + [
            c.padding for b in self.resblocks for c in self._block_convs(b)] + [
            max(u.weight_v.shape[2] - 1 - u.padding, u.padding) for u in self.ups])
        shape = cl.CLShape(N, T, halo)
        convs = [self.conv_pre, self.conv_post] + [c for b in self.resblocks for c in self._block_convs(b)]
        prep = cl.prep_all(self, convs)
        _, xa = cl.fused_conv(cl.ToCL.apply(x, shape, 0), self.conv_pre, shape, None, False, True, LRELU_SLOPE, prep)
        for i, up in enumerate(self.ups):
            last = i + 1 == len(self.ups)
            x_raw, x_act, shape = cl.conv_transpose_cl(xa, up, shape, LRELU_SLOPE)
            T = shape.L
            stage = self.resblocks[i * self.num_kernels:(i + 1) * self.num_kernels]
            rs = []
            for block in stage:
                fn = cl.resblock1_cl if hasattr(block, 'convs1') else cl.resblock2_cl
                rs.append(fn(block, x_raw, x_act, shape, prep=prep)[0])
            xa = cl.MeanActCL.apply(0.01 if last else LRELU_SLOPE, *rs)      # last: F.leaky_relu's default slope, as the reference
        y, _ = cl.fused_conv(xa, self.conv_post, shape, None, True, False, prep=prep)
        return cl.FromCLTanh.apply(y, 1, T, shape)

    def remove_weight_norm(self):
        for up in self.ups:
            up.remove_weight_norm()
        for block in self.resblocks:
            block.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()


@register_model_architecture('hifi_gan', 'hifi_gan_v1')
def hifi_gan_v1():
    return {'h': Namespace(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                           upsample_initial_channel=512, resblock_kernel_sizes=[3, 7, 11],
                           resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])}


@register_model_architecture('hifi_gan', 'hifi_gan_v2')
def hifi_gan_v2():
    return {'h': Namespace(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                           upsample_initial_channel=128, resblock_kernel_sizes=[3, 7, 11],
                           resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                           resblock_initial_channel=64)}


@register_model_architecture('hifi_gan', 'hifi_gan_v3')
def hifi_gan_v3():
    return {'h': Namespace(resblock='2', upsample_rates=[8, 8, 4], upsample_kernel_sizes=[16, 16, 8],
                           upsample_initial_channel=256, resblock_kernel_sizes=[3, 5, 7],
                           resblock_dilation_sizes=[[1, 2], [2, 6], [3, 12]])}
