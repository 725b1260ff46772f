"""HiFi-GAN Generator on the GPU: the channels-last bf16 conv path (Generator.forward_cl: conv_pre, every ResBlock,
conv_post on psnd_conv1d_cl*; ConvTranspose1d polyphase on psnd_convtr1d_*, or the two A/B variants) against the fp32 torch formulation of the same module
(which the CPU golden tests pin to the reference).  bf16 activations between ~20 convs: tolerance 4e-2 relative
Frobenius on the output, 8e-2 on gradients."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.detach().double() - b.detach().double()).norm() / (b.detach().double().norm() + 1e-30))


def _make(resblock, rates, ksz, c0, rk, rd):
    from pytorch_sound_amd.models.vocoders.hifi_gan import Generator
    torch.manual_seed(5)
    h = Namespace(resblock=resblock, upsample_rates=rates, upsample_kernel_sizes=ksz, upsample_initial_channel=c0,
                  resblock_kernel_sizes=rk, resblock_dilation_sizes=rd)
    g = Generator(h).cuda()
    with torch.no_grad():                       # the 0.01-std initialisation gives vanishing activations: rescale
        for n, p in g.named_parameters():
            if n.endswith('weight_v'):
                p.mul_(10.0 if p.abs().max() < 0.1 else 1.0)
    return g


@pytest.mark.parametrize('cfg', [
    ('1', [4, 2], [8, 4], 64, [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]]),
    ('2', [4, 4], [8, 8], 64, [3, 5], [[1, 2], [2, 6]]),
])
@pytest.mark.parametrize('upsample', ['polyphase', 'kernel'])
def test_generator_cl_matches_torch_path(cfg, upsample):
    g = _make(*cfg)
    g.cl_upsample = upsample
    x = torch.randn(3, 80, 24, device='cuda')
    assert g._cl_ok(x)
    xr = x.clone().requires_grad_(True)
    g.use_cl = False
    ref = g(xr)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    gref = {n: p.grad.clone() for n, p in g.named_parameters()}
    gx_ref = xr.grad.clone()
    g.zero_grad()
    g.use_cl = True
    xc = x.clone().requires_grad_(True)
    out = g(xc)
    assert out.shape == ref.shape and out.dtype == torch.float32
    (out * w).sum().backward()
    assert _rel(out, ref) <= 4e-2
    assert _rel(xc.grad, gx_ref) <= 8e-2
    # parameter gradients: all of them together to 8e-2; single tensors looser - a bias gradient is a sum of ~10^3
    # bf16-rounded terms of both signs, its relative error is that of a random walk against a small total
    names = [n for n, _ in g.named_parameters()]
    got = torch.cat([dict(g.named_parameters())[n].grad.flatten() for n in names])
    want = torch.cat([gref[n].flatten() for n in names])
    assert _rel(got, want) <= 8e-2
    worst = max((_rel(p.grad, gref[n]), n) for n, p in g.named_parameters() if gref[n].norm() > 0)
    assert worst[0] <= 0.3, worst


def test_generator_v3_runs_on_the_kernels():
    """hifi_gan_v3 (hifi_gan.py:196-205): ResBlock2, k = 7 with dilation 12 -> tap reach 36, beyond the 25 rows of the default
    A-tile ring: the 7-tap instances sized for 40 rows take it (round 1 dropped the whole generator to the library here)"""
    import ctypes
    from pytorch_sound_amd.models import build_model
    from pytorch_sound_amd._lib import lib
    import bf16_emul as E
    torch.manual_seed(2)
    g = build_model('hifi_gan_v3').cuda()
    with torch.no_grad():
        for n, p in g.named_parameters():
            if n.endswith('weight_v'):
                p.mul_(10.0 if p.abs().max() < 0.1 else 1.0)
    x = torch.randn(2, 80, 12, device='cuda')
    assert g._cl_ok(x)
    w = torch.randn(2, 1, 12 * 256, device='cuda')
    st = (ctypes.c_int64 * 4)()
    lib().psnd_conv_stats(st, 1)

    def run(fn):
        g.zero_grad()
        xr = x.clone().requires_grad_(True)
        y = fn(xr)
        (y * w).sum().backward()
        return y.detach(), xr.grad.clone(), torch.cat([p.grad.flatten() for p in g.parameters()])

    got = run(g)
    lib().psnd_conv_stats(st, 0)
    assert sum(st) > 0                                     # conv kernel launches happened (no library path)
    emul = run(lambda t: E.generator(g, t, 'kernel'))
    g.use_cl = False
    ref = run(g)
    g.use_cl = True
    assert _rel(got[0], ref[0]) <= 4e-2 and _rel(got[1], ref[1]) <= 1e-1 and _rel(got[2], ref[2]) <= 8e-2
    assert _rel(got[0], emul[0]) <= 5e-3 and _rel(got[1], emul[1]) <= 5e-2 and _rel(got[2], emul[2]) <= 2e-2


def test_unsupported_geometry_fails_loudly():
    """no silent library path on a GPU tensor: a conv the kernels cannot take raises"""
    from pytorch_sound_amd._lib import PsndError
    g = _make('2', [4, 4], [8, 8], 64, [9], [[1, 8]])       # k = 9, dilation 8: reach 32 > 25 and k > 7
    with pytest.raises(PsndError):
        g(torch.randn(1, 80, 8, device='cuda'))


@pytest.mark.parametrize('arch', ['hifi_gan_v1', 'hifi_gan_v2', 'hifi_gan_v3'])
def test_folded_generator_decodes_on_the_cl_kernels(arch):
    """InterfaceHifiGAN's decoder (interface/hifi_gan.py:66-117): weight norm removed, torch.no_grad - still on the gfx950 conv
    kernels ((v, g) = (w, ||w||)), equal to the fp32 torch path within the bf16 tolerance; state dict = weight / bias only."""
    from pytorch_sound_amd.models import build_model
    import pytorch_sound_amd.models.vocoders.hifi_gan  # noqa: F401
    torch.manual_seed(5)
    gen = build_model(arch)
    mel = torch.randn(2, 80, 24)
    gen.eval()
    with torch.no_grad():
        want = gen(mel)                                   # CPU, fp32, weight-normed
    gen.remove_weight_norm()
    assert all(k.endswith('.weight') or k.endswith('.bias') for k in gen.state_dict())
    with torch.no_grad():
        assert float((gen(mel) - want).abs().max()) < 1e-5       # folding itself changes nothing (CPU)
    gen = gen.to('cuda:0')
    on_cl = gen._cl_ok(mel.to('cuda:0'))
    assert on_cl
    with torch.no_grad():
        got = gen(mel.to('cuda:0')).cpu()
    assert got.shape == want.shape
    err = float((got - want).norm() / want.norm())
    # measured over seeds and the three architectures: 0.4e-3 .. 2.2e-3 relative L2 (bf16 activations between ~20 convs, fp32 accumulation)
    assert err < (5e-3 if on_cl else 1e-4), err


def test_fanout_sums_stage_gradients_in_one_launch():
    """cl.FanOutCL: the aliases an upsampler's outputs are handed to the resblocks of its stage as (hifi_gan.py:122-131) are the same
    memory, and the gradients that come back are added by psnd_cl_sum2 (fp32 accumulation, one bf16 rounding) - compared with the
    float64 sum; a single contribution passes through untouched."""
    from pytorch_sound_amd import cl
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    xr = torch.randn(2, 40, 64, device=dev).to(torch.bfloat16).requires_grad_(True)
    xa = torch.randn(2, 40, 64, device=dev).to(torch.bfloat16).requires_grad_(True)
    outs = cl.FanOutCL.apply(xr, xa, 3)
    assert len(outs) == 6 and all(o.data_ptr() in (xr.data_ptr(), xa.data_ptr()) for o in outs)
    gs = [torch.randn(2, 40, 64, device=dev).to(torch.bfloat16) for _ in range(6)]
    loss = sum((o.float() * g.float()).sum() for o, g in zip(outs, gs))
    loss.backward()
    want_r = (gs[0].double() + gs[2].double() + gs[4].double())
    want_a = (gs[1].double() + gs[3].double() + gs[5].double())
    assert float((xr.grad.double() - want_r).abs().max()) <= 2 ** -8 * float(want_r.abs().max())
    assert float((xa.grad.double() - want_a).abs().max()) <= 2 ** -8 * float(want_a.abs().max())
    # one contribution only (the other outputs unused): handed through as it is
    xr.grad = xa.grad = None
    outs = cl.FanOutCL.apply(xr, xa, 3)
    (outs[2].float() * gs[0].float()).sum().backward()
    assert torch.equal(xr.grad, gs[0]) and xa.grad is None


@pytest.mark.parametrize('N,Lp,C,lo,hi', [(3, 40, 32, 5, 33), (16, 306, 256, 25, 281), (2, 8200, 64, 25, 8175), (5, 64, 128, 0, 64), (1, 24, 8, 3, 3)])
def test_cl_colsum_window(N, Lp, C, lo, hi):
    """psnd_cl_colsum (bias gradient of a transposed conv, hifi_gan.py:107-110): fp32 column sums of a channels-last bf16 matrix over the
    rows [lo, hi) of every Lp-row clip buffer - rows outside the window are never read (NaN there must not show) - and over all rows with
    Lp = 0; against a float64 sum, and repeatable bit for bit"""
    from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
    dev = torch.device('cuda:0')
    torch.manual_seed(N + Lp + C)
    g = torch.randn(N, Lp, C, device=dev).to(torch.bfloat16)
    poisoned = g.clone()
    poisoned[:, :lo] = float('nan')
    poisoned[:, hi:] = float('nan')
    rows = N * Lp
    part = torch.empty(int(lib().psnd_cl_colsum_splits(rows, C)) * C, dtype=torch.float32, device=dev)
    out = torch.empty(C, dtype=torch.float32, device=dev)
    st = stream_ptr(dev)
    check(lib().psnd_cl_colsum(ptr(poisoned), rows, C, Lp, lo, hi, ptr(part), ptr(out), st), 'colsum')
    want = g[:, lo:hi].double().sum(dim=(0, 1))
    assert bool(torch.isfinite(out).all())
    assert float((out.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())) + 1e-4
    out2 = torch.empty_like(out)
    check(lib().psnd_cl_colsum(ptr(poisoned), rows, C, Lp, lo, hi, ptr(part), ptr(out2), st), 'colsum')
    assert torch.equal(out, out2)
    check(lib().psnd_cl_colsum(ptr(g), rows, C, 0, 0, 0, ptr(part), ptr(out), st), 'colsum')
    want_all = g.double().sum(dim=(0, 1))
    assert float((out.double() - want_all).abs().max()) <= 1e-5 * max(1.0, float(want_all.abs().max())) + 1e-4


def test_unknown_upsample_mode_raises():
    """'library' (removed in round 5) or a typo must not silently select the zero-spread form"""
    g = _make('1', [4, 2], [8, 4], 64, [3, 7, 11], [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    g.cl_upsample = 'library'
    with pytest.raises(ValueError, match='cl_upsample'):
        g(torch.randn(1, 80, 8, device='cuda'))
