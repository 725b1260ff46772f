"""HiFi-GAN Generator on the GPU: the channels-last bf16 conv path (Generator.forward_cl: conv_pre, every ResBlock,
conv_post on psnd_conv1d_cl*; ConvTranspose1d from the library) against the fp32 torch formulation of the same module
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
@pytest.mark.parametrize('upsample', ['library', 'kernel'])
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


def test_generator_v3_dilation_falls_back_to_library_path():
    from pytorch_sound_amd.models import build_model
    g = build_model('hifi_gan_v3').cuda()       # k=7, dilation 12: tap reach 36 > 25
    x = torch.randn(1, 80, 8, device='cuda')
    assert not g._cl_ok(x)
    assert g(x).shape == (1, 1, 8 * 256)


@pytest.mark.parametrize('arch', ['hifi_gan_v2', 'hifi_gan_v3'])
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
    assert on_cl == (arch != 'hifi_gan_v3')                # v3's dilated 7-tap convs reach 36 rows: beyond the staged tile, library path
    with torch.no_grad():
        got = gen(mel.to('cuda:0')).cpu()
    assert got.shape == want.shape
    err = float((got - want).norm() / want.norm())
    assert err < (4e-2 if on_cl else 1e-4), err
