"""Config-3 scale parity of the conv stack (BASELINE.json configs[2]: LibriTTS mel + conv-vocoder training, 16 clips x 8192
samples, registered `hifi_gan_v1`): the full-width generator (C0 = 512, 7- and 11-tap blocks at 512 / 256 channels) and the
128-row workgroup-tile instances (`MT = 2`) of the forward and the paired backward kernels, which only launches of >= 1024
64-row tiles select - i.e. only the long stages of this configuration.

Three oracles, from loose to tight:
  * the imported reference's own outputs and gradients (tests/golden/hifigan.npz, tiny1 / tiny2), loaded on the GPU;
  * the module's fp32 torch formulation (pinned to the reference by tests/test_modules_golden.py on CPU): 4e-2 / 8e-2
    relative Frobenius - the accumulated bf16 rounding of ~20 convs;
  * tests/bf16_emul.py - the same arithmetic with the kernels' rounding points (bf16 operands, fp32 accumulation):
    outputs to 1e-3 (tiny nets: bit-exact) / 5e-3 (v1), every single parameter gradient to 3e-2 / 6e-2, and for one conv
    the fp32 weight-gradient / bias-gradient / weight-norm results to 2e-4 of max against exact arithmetic on the
    bf16-rounded operands.
"""
import ctypes
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bf16_emul as E
from test_modules_golden import TINY, sd_from

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _stats(reset=False):
    from pytorch_sound_amd._lib import lib
    out = (ctypes.c_int64 * 4)()
    lib().psnd_conv_stats(out, 1 if reset else 0)
    return list(out)


def _rescale(g):
    with torch.no_grad():                       # the 0.01-std initialisation gives vanishing activations: rescale
        for n, p in g.named_parameters():
            if n.endswith('weight_v'):
                p.mul_(10.0 if p.abs().max() < 0.1 else 1.0)


def _run(g, fn, x, w):
    g.zero_grad()
    xr = x.clone().requires_grad_(True)
    y = fn(xr)
    (y * w).sum().backward()
    return y.detach(), xr.grad.clone(), {n: p.grad.clone() for n, p in g.named_parameters()}


def _compare(got, want, tol_out, tol_gx, tol_all, tol_each, tag):
    (y, gx, gp), (yr, gxr, gpr) = got, want
    assert y.shape == yr.shape
    errs = {'out': _rel(y, yr), 'gx': _rel(gx, gxr)}
    names = sorted(gp)
    errs['all'] = _rel(torch.cat([gp[n].flatten() for n in names]), torch.cat([gpr[n].flatten() for n in names]))
    each = sorted(((_rel(gp[n], gpr[n]), n) for n in names if gpr[n].norm() > 0), reverse=True)
    print('%s: out %.2e gx %.2e params(all) %.2e worst %.2e (%s)' % (tag, errs['out'], errs['gx'], errs['all'], each[0][0], each[0][1]))
    assert errs['out'] <= tol_out, (tag, errs)
    assert errs['gx'] <= tol_gx, (tag, errs)
    assert errs['all'] <= tol_all, (tag, errs)
    assert each[0][0] <= tol_each, (tag, each[:3])


@pytest.mark.parametrize('upsample', ['polyphase', 'kernel'])
def test_hifi_gan_v1_config3_shape(upsample):
    """registered hifi_gan_v1, 16 clips x 32 frames -> 16 x 8192 samples (the config-3 training shape), forward + backward"""
    from pytorch_sound_amd.models import build_model
    import pytorch_sound_amd.models.vocoders.hifi_gan  # noqa: F401
    torch.manual_seed(11)
    g = build_model('hifi_gan_v1').cuda()
    _rescale(g)
    g.cl_upsample = upsample
    x = torch.randn(16, 80, 32, device='cuda')
    assert g._cl_ok(x)
    _stats(reset=True)
    g.use_cl = True
    got = None
    w = None
    with torch.no_grad():
        w = torch.randn(16, 1, 8192, device='cuda')
    got = _run(g, g, x, w)
    st = _stats()
    assert got[0].shape == (16, 1, 8192)
    # the long stages ran the 128-row tile instances forward, the short first stage the 64-row ones; the paired backward takes
    # 128-row input-gradient tiles at every size (the 64-row paired instances are forced in test_fused_conv_exact_on_rounded_operands)
    assert st[1] > 0 and st[3] > 0 and st[0] > 0, st
    g.use_cl = False
    ref32 = _run(g, g, x, w)
    emul = _run(g, lambda t: E.generator(g, t, 'kernel'), x, w)
    g.use_cl = True
    _compare(got, ref32, 4e-2, 8e-2, 8e-2, 1.0, 'v1 vs fp32')          # single tensors vs fp32: see the emulation bound below
    # measured: out 7e-4, all parameter gradients together 1.2e-3; the input gradient and single bias gradients 2.4e-2 .. 3.5e-2 -
    # outputs differ by one bf16 ulp on a few elements (fp32 summation order over K = 512 x 11), which flips leaky' masks
    _compare(got, emul, 5e-3, 5e-2, 5e-3, 6e-2, 'v1 vs bf16 emulation')


@pytest.mark.parametrize('name', ['tiny1', 'tiny2'])
@pytest.mark.parametrize('upsample', ['polyphase', 'kernel'])
def test_reference_golden_on_gpu(golden, name, upsample):
    """the imported reference's outputs and gradients (fp32, CPU) against the CL kernels on the GPU"""
    from pytorch_sound_amd.models.vocoders import hifi_gan
    gd = golden('hifigan')
    g = hifi_gan.Generator(TINY[name])
    g.load_state_dict(sd_from(gd, name + '/sd/'))
    g = g.cuda()
    g.cl_upsample = upsample
    x = torch.from_numpy(gd[name + '/x']).cuda()
    w = torch.from_numpy(gd[name + '/g']).cuda()
    assert g._cl_ok(x)
    got = _run(g, g, x, w)
    want = (torch.from_numpy(gd[name + '/y']).cuda(), torch.from_numpy(gd[name + '/gx']).cuda(),
            {n: torch.from_numpy(gd['%s/g/%s' % (name, n)]).cuda() for n, _ in g.named_parameters()})
    # fp32 reference vs bf16 kernels: accumulated rounding of the whole stack
    _compare(got, want, 4e-2, 1.5e-1, 1e-1, 0.5, name + ' vs reference golden')
    emul = _run(g, lambda t: E.generator(g, t, 'kernel'), x, w)
    _compare(got, emul, 1e-3, 2e-2, 2e-2, 3e-2, name + ' vs bf16 emulation')      # measured: output bit-exact, gradients 5e-3 .. 9e-3
    _compare(emul, want, 4e-2, 1.5e-1, 1e-1, 0.5, name + ' emulation vs reference golden')   # the emulation itself is the reference's function


SHAPES = [(64, 64, 3, 1, 50, 2), (96, 64, 3, 5, 173, 3), (64, 40, 7, 3, 61, 2), (513, 256, 3, 1, 173, 2),
          (256, 513, 3, 1, 45, 2), (32, 32, 11, 5, 200, 1), (128, 128, 11, 1, 300, 2), (512, 512, 7, 5, 40, 2),
          (64, 64, 7, 12, 80, 2), (32, 64, 5, 9, 50, 1)]        # tap reach 36 (hifi_gan_v3's k = 7, dilation 12): the 40-row instances


@pytest.mark.parametrize('Cin,Cout,k,dil,L,N', [(80, 512, 7, 1, 32, 16), (64, 64, 3, 1, 50, 2), (128, 128, 11, 5, 64, 2)])
@pytest.mark.parametrize('outs', ['act', 'raw'])
def test_single_output_conv_exact_on_rounded_operands(Cin, Cout, k, dil, L, N, outs):
    """the head / tail roles of a chain: activated-only output (conv_pre: the incoming gradient is g_act * leaky'(own output),
    formed while the operand is staged, with NO raw part) and raw-only output (conv_post), no residual"""
    from pytorch_sound_amd import cl
    from pytorch_sound_amd.models.vocoders.hifi_gan import WNConv1d
    dev = torch.device('cuda:0')
    torch.manual_seed(Cin + k)
    pad = (k * dil - dil) // 2
    conv = WNConv1d(Cin, Cout, k, dil, pad, init_std=0.05).to(dev)
    bf = lambda t: t.to(torch.bfloat16).double()                     # noqa: E731
    x = torch.randn(N, Cin, L, device=dev)
    gy = torch.randn(N, Cout, L, device=dev)
    shape = cl.CLShape(N, L, pad + 2)
    xc = x.clone().requires_grad_(True)
    yb, yab = cl.fused_conv(cl.ToCL.apply(xc, shape, 0), conv, shape, None, outs == 'raw', outs == 'act', 0.1)
    out = cl.FromCL.apply(yab if outs == 'act' else yb, Cout, L, shape)
    (out * gy).sum().backward()
    w32 = (conv.weight_v.detach() * (conv.weight_g.detach() / conv.weight_v.detach().flatten(1).norm(dim=1).view(-1, 1, 1)))
    wq, xq = bf(w32), bf(x)
    v = F.conv1d(xq, wq, conv.bias.detach().double(), 1, pad, dil)
    want = F.leaky_relu(v, 0.1) if outs == 'act' else v
    ulp = 2.0 ** -8
    tol_bf = lambda got, w_: bool(((got.double() - w_).abs() <= 1.01 * ulp * w_.abs() + 4e-4 * float(w_.abs().max())).all())  # noqa: E731
    assert tol_bf(out.detach(), want)
    g = bf(gy)
    if outs == 'act':          # formed in fp32 as the kernel does (slope 0.1f), rounded to bf16 once
        g = bf(gy.to(torch.bfloat16).float() * torch.where(out.detach() > 0, 1.0, 0.1).float())
    gx = torch.nn.grad.conv1d_input(xq.shape, wq, g, 1, pad, dil)
    err = float((xc.grad.double() - gx).norm() / gx.norm())
    assert tol_bf(xc.grad, gx), err
    gw = torch.nn.grad.conv1d_weight(xq, wq.shape, g, 1, pad, dil)
    v64, g64 = conv.weight_v.detach().double(), conv.weight_g.detach().double()
    nrm = v64.flatten(1).norm(dim=1).view(-1, 1, 1)
    vhat = v64 / nrm
    d = (gw * vhat).flatten(1).sum(1).view(-1, 1, 1)
    gv = (g64 / nrm) * (gw - vhat * d)
    close = lambda got, w_: float((got.double() - w_).abs().max()) <= 2e-4 * float(w_.abs().max())   # noqa: E731
    assert close(conv.bias.grad, g.sum((0, 2))) and close(conv.weight_g.grad, d) and close(conv.weight_v.grad, gv)


@pytest.mark.parametrize('mt', [1, 2])
@pytest.mark.parametrize('Cin,Cout,k,dil,L,N', SHAPES)
def test_fused_conv_exact_on_rounded_operands(Cin, Cout, k, dil, L, N, mt, monkeypatch, lab_lib):
    """one fused conv, forward + backward, against EXACT (float64) arithmetic on the bf16-rounded operands the kernels see:
    the bf16 outputs to one rounding (2^-8 relative per element; + 4e-4 of max absolute: the kernel's weight-norm scale
    g / ||v|| is summed in another order than torch's, so a handful of the 10^5 weights round to the neighbouring bf16), the
    fp32 results (weight / bias gradient slabs summed, weight-norm backward) to 2e-4 of max.  mt forces the 64- / 128-row tile instances
    (forward and paired backward) on these small shapes."""
    from pytorch_sound_amd import cl
    from pytorch_sound_amd.models.vocoders.hifi_gan import WNConv1d
    if mt:
        monkeypatch.setenv('PSND_CONV_MT', str(mt))
        monkeypatch.setenv('PSND_PAIR_MT', str(mt))
    dev = torch.device('cuda:0')
    torch.manual_seed(Cin + k)
    pad = (k * dil - dil) // 2
    conv = WNConv1d(Cin, Cout, k, dil, pad, init_std=0.05).to(dev)
    with torch.no_grad():
        conv.weight_g.mul_(1.0 + 0.3 * torch.rand_like(conv.weight_g))
    bf = lambda t: t.to(torch.bfloat16).double()                     # noqa: E731
    x = torch.randn(N, Cin, L, device=dev)
    r = torch.randn(N, Cout, L, device=dev)
    gy = torch.randn(N, Cout, L, device=dev)
    gya = torch.randn(N, Cout, L, device=dev)
    shape = cl.CLShape(N, L, pad + 1)
    xc = x.clone().requires_grad_(True)
    rc = r.clone().requires_grad_(True)
    _stats(reset=True)
    yb, yab = cl.fused_conv(cl.ToCL.apply(xc, shape, 0), conv, shape, cl.ToCL.apply(rc, shape, 0), True, True, 0.1)
    y2 = cl.FromCL.apply(yb, Cout, L, shape)
    ya2 = cl.FromCL.apply(yab, Cout, L, shape)
    ((y2 * gy).sum() + (ya2 * gya).sum()).backward()
    st = _stats()
    assert st[1 if mt == 2 else 0] > 0 and st[3 if mt == 2 else 2] > 0, st
    # exact arithmetic on the rounded operands
    v64, g64 = conv.weight_v.detach().double(), conv.weight_g.detach().double()
    nrm = v64.flatten(1).norm(dim=1).view(-1, 1, 1)
    w32 = (conv.weight_v.detach() * (conv.weight_g.detach() / conv.weight_v.detach().flatten(1).norm(dim=1).view(-1, 1, 1)))
    wq = bf(w32)
    xq, rq = bf(x), bf(r)
    v = F.conv1d(xq, wq, conv.bias.detach().double(), 1, pad, dil) + rq
    ulp = 2.0 ** -8
    tol_bf = lambda got, want: bool(((got.double() - want).abs() <= 1.01 * ulp * want.abs() + 4e-4 * float(want.abs().max())).all())  # noqa: E731
    assert tol_bf(y2.detach(), v)
    assert tol_bf(ya2.detach(), F.leaky_relu(v, 0.1))
    # incoming gradient as the kernels form it: bf16(gy) + bf16(gya) * leaky'(own activated output), rounded once
    gcomb = bf(bf(gy) + bf(gya) * torch.where(ya2.detach().double() > 0, 1.0, 0.1))
    assert tol_bf(rc.grad, gcomb)                                     # the residual branch receives exactly that tensor
    gx = torch.nn.grad.conv1d_input(xq.shape, wq, gcomb, 1, pad, dil)
    assert tol_bf(xc.grad, gx)
    gw = torch.nn.grad.conv1d_weight(xq, wq.shape, gcomb, 1, pad, dil)
    gb = gcomb.sum((0, 2))
    vhat = v64 / nrm
    d = (gw * vhat).flatten(1).sum(1).view(-1, 1, 1)
    gv = (g64 / nrm) * (gw - vhat * d)
    close = lambda got, want: float((got.double() - want).abs().max()) <= 2e-4 * float(want.abs().max())   # noqa: E731
    assert close(conv.bias.grad, gb), float((conv.bias.grad.double() - gb).abs().max() / gb.abs().max())
    assert close(conv.weight_g.grad, d)
    assert close(conv.weight_v.grad, gv)


@pytest.mark.parametrize('Cin,Cout,u,L,N', [(64, 32, 8, 32, 3), (512, 256, 8, 32, 16), (32, 16, 2, 100, 2), (128, 64, 4, 61, 2),
                                             (16, 8, 2, 24, 1)])
def test_polyphase_conv_transpose_exact_on_rounded_operands(Cin, Cout, u, L, N):
    """ConvTranspose1d(k = 2u, stride u, padding u/2) of hifi_gan.py:109 on psnd_convtr1d_*: outputs, input gradient, and the
    fp32 parameter gradients (weight norm over dim 0 = per input channel) against float64 arithmetic on the bf16-rounded
    operands.  bf16 outputs to one rounding (+ 4e-4 of max, see above), fp32 results to 2e-4 of max."""
    from pytorch_sound_amd import cl
    from pytorch_sound_amd.models.vocoders.hifi_gan import WNConvTranspose1d
    dev = torch.device('cuda:0')
    torch.manual_seed(Cin + u)
    K, pad = 2 * u, u // 2
    up = WNConvTranspose1d(Cin, Cout, K, u, pad, init_std=0.05).to(dev)
    with torch.no_grad():
        up.weight_g.mul_(1.0 + 0.3 * torch.rand_like(up.weight_g))
    bf = lambda t: t.to(torch.bfloat16).double()                     # noqa: E731
    x = torch.randn(N, Cin, L, device=dev)
    gy = torch.randn(N, Cout, L * u, device=dev)
    gya = torch.randn(N, Cout, L * u, device=dev)
    shape, out_shape = cl.CLShape(N, L, 3), cl.CLShape(N, L * u, 11)
    xc = x.clone().requires_grad_(True)
    # the output buffers and the combined gradient are NOT zeroed first (round 5): the kernels write every row that is read.  Freed blocks
    # of exactly their sizes, full of NaN, are what the allocator hands out next.
    Cr = cl.round_up(Cout, cl.ALIGN_C)
    junk = [torch.full(shp, float('nan'), dtype=torch.bfloat16, device=dev)
            for shp in ((2, N, out_shape.Lp, Cr), (N, out_shape.Lp, Cr)) for _ in range(4)]
    del junk
    raw, act = cl.ConvTransposeCL.apply(cl.ToCL.apply(xc, shape, 0), up.weight_v, up.weight_g, up.bias, shape, out_shape, u, pad, 0.1)
    assert bool(torch.isfinite(raw.float()).all()) and bool(torch.isfinite(act.float()).all())
    y2 = cl.FromCL.apply(raw, Cout, L * u, out_shape)
    ya2 = cl.FromCL.apply(act, Cout, L * u, out_shape)
    ((y2 * gy).sum() + (ya2 * gya).sum()).backward()
    # halo rows and padded channels of the outputs are zero
    for buf in (raw, act):
        b_ = buf.detach().float()
        assert float(b_[:, :out_shape.HP].abs().max()) == 0 and float(b_[:, out_shape.HP + L * u:].abs().max()) == 0
        if b_.shape[2] > Cout:
            assert float(b_[:, :, Cout:].abs().max()) == 0
    v32, g32 = up.weight_v.detach(), up.weight_g.detach()
    w32 = v32 * (g32 / v32.flatten(1).norm(dim=1).view(-1, 1, 1))
    wq, xq = bf(w32), bf(x)
    v = F.conv_transpose1d(xq, wq, up.bias.detach().double(), u, pad)
    ulp = 2.0 ** -8
    tol_bf = lambda got, want: bool(((got.double() - want).abs() <= 1.01 * ulp * want.abs() + 4e-4 * float(want.abs().max())).all())  # noqa: E731
    assert tol_bf(y2.detach(), v)
    assert tol_bf(ya2.detach(), F.leaky_relu(v, 0.1))
    gcomb = bf(gy.to(torch.bfloat16).float() + gya.to(torch.bfloat16).float() * torch.where(ya2.detach() > 0, 1.0, 0.1).float())
    gx = F.conv1d(gcomb, wq, None, u, pad)                            # adjoint of conv_transpose1d wrt its input
    assert tol_bf(xc.grad, gx), float((xc.grad.double() - gx).norm() / gx.norm())
    # parameter gradients in float64 through autograd on the rounded operands
    v64 = v32.double().requires_grad_(True)
    g64 = g32.double().requires_grad_(True)
    b64 = up.bias.detach().double().requires_grad_(True)
    w64 = v64 * (g64 / v64.flatten(1).norm(dim=1).view(-1, 1, 1))
    # the kernels see bf16(w): gradient wrt w taken at the rounded point, chained through the exact weight norm
    wq_leaf = wq.clone().requires_grad_(True)
    (F.conv_transpose1d(xq, wq_leaf, b64, u, pad) * gcomb).sum().backward()
    w64.backward(wq_leaf.grad)
    close = lambda got, want: float((got.double() - want).abs().max()) <= 2e-4 * float(want.abs().max())   # noqa: E731
    assert close(up.bias.grad, b64.grad), float((up.bias.grad.double() - b64.grad).abs().max() / b64.grad.abs().max())
    assert close(up.weight_g.grad, g64.grad)
    assert close(up.weight_v.grad, v64.grad)


def test_folded_checkpoint_loaded_after_weight_norm_removal():
    """build -> remove_weight_norm -> load_state_dict({weight, bias}) (accepted with strict=True): the CL kernels
    must run on the NEW weights, not on the (v, g) derived at fold time"""
    from pytorch_sound_amd.models import build_model
    import pytorch_sound_amd.models.vocoders.hifi_gan  # noqa: F401
    torch.manual_seed(3)
    src = build_model('hifi_gan_v2')
    _rescale(src)
    src.remove_weight_norm()
    dst = build_model('hifi_gan_v2')
    dst.remove_weight_norm()
    dst = dst.cuda()
    mel = torch.randn(2, 80, 16, device='cuda')
    with torch.no_grad():
        before = dst(mel)
        dst.load_state_dict(src.state_dict())
        assert dst._cl_ok(mel)
        got = dst(mel)
        dst.use_cl = False
        want = dst(mel)
    assert _rel(got, want) < 4e-2, _rel(got, want)
    assert _rel(before, want) > 0.5                                   # the two weight sets really differ
