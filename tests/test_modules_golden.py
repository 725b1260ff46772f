"""Attention / feed-forward / positional-encoding blocks and the HiFi-GAN stack against outputs AND
gradients of the imported reference (tools/gen_golden.py G5, G6).  These are floating-point blocks
built on library GEMM/conv calls: tolerance 2e-5 relative to the tensor's max (fp32 reassociation)."""
import numpy as np
import pytest
import torch
from argparse import Namespace

from pytorch_sound_amd.models import build_model
from pytorch_sound_amd.models.modules import MultiHeadAttention, PointwiseFeedForward, PositionalEncoding
from pytorch_sound_amd.models.vocoders import hifi_gan

RTOL = 2e-5


def close(a, b, rtol=RTOL):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    scale = max(np.abs(b).max(), 1e-6)
    return np.abs(a - b).max() <= rtol * scale


def sd_from(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize('tag', ['nomask', 'mask'])
def test_multi_head_attention(golden, tag):
    g = golden('modules')
    mha = MultiHeadAttention(16, 4, 0.0)
    mha.load_state_dict(sd_from(g, 'mha/sd/'))
    x = torch.from_numpy(g['mha/x']).requires_grad_(True)
    mask = torch.from_numpy(g['mha/mask']) if tag == 'mask' else None
    y, att = mha(x, mask)
    (y * torch.from_numpy(g['mha/g'])).sum().backward()
    assert close(y, g['mha/%s/y' % tag]) and close(att, g['mha/%s/att' % tag])
    assert att.shape == (4 * 3, 10, 10)
    assert close(x.grad, g['mha/%s/gx' % tag], 5e-5)
    for k, p in mha.named_parameters():
        assert close(p.grad, g['mha/%s/g/%s' % (tag, k)], 5e-5), k
    if tag == 'mask':
        a = att.detach().numpy()
        m = g['mha/mask']                                   # (N,T); batch index b = h*N + n
        for b in range(12):
            n = b % 3
            assert np.all(a[b][m[n], :] == 0) and np.all(a[b][:, m[n]] == 0)
            assert np.allclose(a[b][:, ~m[n]].sum(0), 1.0, atol=1e-6)


def test_pointwise_feed_forward_and_positional_encoding(golden):
    g = golden('modules')
    ffn = PointwiseFeedForward(16, 0.0)
    ffn.load_state_dict(sd_from(g, 'ffn/sd/'))
    x = torch.from_numpy(g['mha/x']).requires_grad_(True)
    y = ffn(x)
    (y * torch.from_numpy(g['mha/g'])).sum().backward()
    assert close(y, g['ffn/y']) and close(x.grad, g['ffn/gx'], 5e-5)
    for k, p in ffn.named_parameters():
        assert close(p.grad, g['ffn/g/' + k], 5e-5), k
    pe = PositionalEncoding(16, 32)
    assert np.array_equal(pe.pe.numpy(), g['pe/table'])
    assert close(pe(torch.from_numpy(g['mha/x'])), g['pe/y'], 1e-6)
    assert list(pe.state_dict()) == ['pe']


TINY = {
    'tiny1': Namespace(resblock='1', upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=32,
                       resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]]),
    'tiny2': Namespace(resblock='2', upsample_rates=[8, 4], upsample_kernel_sizes=[16, 8], upsample_initial_channel=16,
                       resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]]),
}


@pytest.mark.parametrize('name', ['tiny1', 'tiny2'])
def test_hifigan_generator_fwd_bwd(golden, name):
    g = golden('hifigan')
    gen = hifi_gan.Generator(TINY[name])
    missing = gen.load_state_dict(sd_from(g, name + '/sd/'))
    assert not missing.missing_keys and not missing.unexpected_keys
    x = torch.from_numpy(g[name + '/x']).requires_grad_(True)
    y = gen(x)
    (y * torch.from_numpy(g[name + '/g'])).sum().backward()
    assert close(y, g[name + '/y']) and close(x.grad, g[name + '/gx'], 1e-4)
    for k, p in gen.named_parameters():
        assert close(p.grad, g['%s/g/%s' % (name, k)], 1e-4), k
    # weight norm removal keeps the function
    with torch.no_grad():
        gen.remove_weight_norm()
        assert close(gen(torch.from_numpy(g[name + '/x'])), g[name + '/y'])
    assert not any(k.endswith('weight_g') for k in gen.state_dict())


@pytest.mark.parametrize('arch', ['hifi_gan_v1', 'hifi_gan_v2', 'hifi_gan_v3'])
def test_registered_archs_match_reference_layout(golden, arch):
    g = golden('hifigan')
    gen = build_model(arch)
    sd = gen.state_dict()
    assert sum(p.numel() for p in gen.parameters()) == int(g[arch + '/n_params'])
    assert sorted(sd.keys()) == list(g[arch + '/keys'])
    assert [str(tuple(v.shape)) for k, v in sorted(sd.items())] == list(g[arch + '/shapes'])
    with torch.no_grad():
        assert list(gen(torch.randn(1, 80, 4)).shape) == list(g[arch + '/out_shape'])


def test_shipped_v2_checkpoint_output(golden):
    """the reference's own asset (hifi_gan_v2.pt) evaluated by the imported reference: our generator
    reproduces the waveform when the same weights are present (fixture carries no weights - the test
    runs only where the asset is reachable, i.e. in the build container)."""
    import os
    ck = os.environ.get('PSND_V2_CKPT', '/root/reference/assets/vocoders/hifi_gan_v2.pt')
    if not os.path.exists(ck):
        pytest.skip('hifi_gan_v2.pt asset not present on this machine')
    g = golden('hifigan')
    gen = build_model('hifi_gan_v2')
    gen.load_state_dict(torch.load(ck, map_location='cpu', weights_only=False)['generator'])
    with torch.no_grad():
        y = gen(torch.from_numpy(g['v2ckpt/mel']))
    assert close(y, g['v2ckpt/y'], 1e-4)


def test_folded_buffers_follow_a_later_load():
    """remove_weight_norm() derives the (v, g) pair the gfx950 kernels read from `weight`; a folded checkpoint
    loaded AFTER the fold (strict=True accepts it) must refresh that pair (sync_folded, called by the CL path)."""
    torch.manual_seed(0)
    a, b = hifi_gan.Generator(TINY['tiny1']), hifi_gan.Generator(TINY['tiny1'])
    a.remove_weight_norm()
    b.remove_weight_norm()
    ptrs = [(c.weight_v.data_ptr(), c.weight_g.data_ptr()) for c in b._all_convs()]
    b.load_state_dict(a.state_dict())
    for c in b._all_convs():
        c.sync_folded()
    for ca, cb, pp in zip(a._all_convs(), b._all_convs(), ptrs):
        assert torch.equal(cb.weight_v, ca.weight.detach()) and torch.equal(cb.weight, ca.weight)
        assert torch.allclose(cb.weight_g, ca.weight.detach().flatten(1).norm(dim=1).view(-1, 1, 1))
        assert (cb.weight_v.data_ptr(), cb.weight_g.data_ptr()) == pp          # refreshed in place: pack cache keys stay valid
    x = torch.randn(1, 80, 5)
    with torch.no_grad():
        assert torch.equal(a(x), b(x))
