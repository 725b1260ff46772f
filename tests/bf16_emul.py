"""Test-side model of the CL conv path's NUMERICS in plain torch (test infrastructure, never imported by the product).

The gfx950 conv kernels keep activations, incoming gradients and weights in bf16 and accumulate in fp32
(pytorch_sound_amd/cl.py, csrc/psnd_conv.hip).  Comparing them with an fp32 formulation therefore needs bounds of the
size of the accumulated bf16 rounding (4e-2 on a 20-conv generator, far looser on a single bias gradient) - bounds a
misplaced gradient slab could hide behind.  This module restates the SAME arithmetic with the rounding points of the
kernels (`q`: round to bf16 in the forward direction AND round the gradient in the backward direction, as a stored
CL tensor does), fp32 accumulation from torch's fp32 convolution on the rounded operands.  Against it the kernels agree
to a few bf16 ulps on a few elements, so the parity tests can use bounds 10-50x tighter than against fp32.

Follows pytorch_sound/models/vocoders/hifi_gan.py:55-63 (ResBlock1), :84-88 (ResBlock2), :118-136 (Generator.forward).
"""
import torch
import torch.nn.functional as F


class _Q(torch.autograd.Function):
    """a tensor stored as bf16 in both directions: value and gradient are rounded to nearest-even bf16"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _QF(torch.autograd.Function):
    """rounded on the way in only (the bf16 weight packs: their gradient is formed and kept in fp32)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def q(x):
    return _Q.apply(x)


def qf(x):
    return _QF.apply(x)


def conv(c, xa, res=None):
    """fp32 pre-rounding output of one fused conv: conv(xa; bf16(w)) + bias (+ res).  xa / res are bf16-valued."""
    y = F.conv1d(xa, qf(c.effective_weight()), c.bias, 1, c.padding, c.dilation)
    return y if res is None else y + res


def resblock1(block, x, xa, last_slope=0.1):
    n = len(block.convs1)
    for i, (c1, c2) in enumerate(zip(block.convs1, block.convs2)):
        ta = q(F.leaky_relu(conv(c1, xa), 0.1))
        v = conv(c2, ta, x)
        x, xa = q(v), q(F.leaky_relu(v, last_slope if i == n - 1 else 0.1))
    return x


def resblock2(block, x, xa, last_slope=0.1):
    n = len(block.convs)
    for i, c in enumerate(block.convs):
        v = conv(c, xa, x)
        x, xa = q(v), q(F.leaky_relu(v, last_slope if i == n - 1 else 0.1))
    return x


def generator(gen, x, upsample='library'):
    """Generator.forward_cl restated: same rounding points as pytorch_sound_amd/models/vocoders/hifi_gan.py:forward_cl."""
    xa = q(F.leaky_relu(conv(gen.conv_pre, q(x)), 0.1))
    h = xa
    for i, up in enumerate(gen.ups):
        last = i + 1 == len(gen.ups)
        if upsample == 'library':
            h = up(h)                                   # fp32 library transposed conv between the layout kernels
            x_raw, x_act = q(h), q(F.leaky_relu(h, 0.1))
        else:                                           # on the CL kernels: bf16 operands, fp32 accumulation, bf16 outputs
            v = F.conv_transpose1d(h, qf(up.effective_weight()), up.bias, up.stride, up.padding)
            x_raw, x_act = q(v), q(F.leaky_relu(v, 0.1))
        stage = gen.resblocks[i * gen.num_kernels:(i + 1) * gen.num_kernels]
        acc = None
        for block in stage:
            r = resblock1(block, x_raw, x_act) if hasattr(block, 'convs1') else resblock2(block, x_raw, x_act)
            acc = r if acc is None else acc + r
        h = F.leaky_relu(acc / gen.num_kernels, 0.01 if last else 0.1)
        if upsample != 'library':
            h = q(h)
    return torch.tanh(q(conv(gen.conv_post, q(h))))


def separator(model, mag):
    """ConvSeparator.forward_cl restated (pytorch_sound_amd/models/separator.py): log1p rounded into the CL layout, conv_pre, the
    ResBlock1 chain with its (raw, activated) tensor pair handed from block to block, conv_post stored as bf16, sigmoid(.) * mag in
    fp32.  Rounding points as psnd_conv1d_cl / psnd_conv1d_cl_pair (the pair's intermediate is rounded to bf16 in LDS)."""
    v = conv(model.conv_pre, q(torch.log1p(mag)))
    x, xa = q(v), q(F.leaky_relu(v, 0.1))
    for block in model.blocks:
        for c1, c2 in zip(block.convs1, block.convs2):
            ta = q(F.leaky_relu(conv(c1, xa), 0.1))
            v = conv(c2, ta, x)
            x, xa = q(v), q(F.leaky_relu(v, 0.1))
    return torch.sigmoid(q(conv(model.conv_post, xa))) * mag
