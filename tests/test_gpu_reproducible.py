"""Run-to-run reproducibility on the GPU: the kernels of the three model paths add up in fixed orders (slab sums, row sums, partial + final
loss sums, the GroupNorm's row pairs; no atomics), so two passes over the same data give the same bits (the transformer block:
tests/test_gpu_modules.py::test_transformer_block_gradients_are_reproducible)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _passes(net, args, n=3):
    runs = []
    for _ in range(n):
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = net(*args)
        y = y[0] if isinstance(y, (tuple, list)) else y
        loss = (y.float() ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        runs.append([loss.detach().clone()] + [p.grad.clone() for p in net.parameters() if p.grad is not None])
        for p in net.parameters():
            p.grad = None
    return runs


@pytest.mark.parametrize('name,shape', [('conv_separator_voicebank', (8, 513, 173)), ('hifi_gan_v1', (4, 80, 32))])
def test_model_gradients_are_the_same_bits_from_run_to_run(name, shape):
    from pytorch_sound_amd.models import build_model
    from pytorch_sound_amd.models import separator  # noqa: F401  (registers conv_separator_*)
    from pytorch_sound_amd.models.vocoders import hifi_gan  # noqa: F401
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = build_model(name).to(dev)
    x = torch.rand(*shape, device=dev) if name.startswith('conv_separator') else torch.randn(*shape, device=dev)
    runs = _passes(net, (x,))
    assert len(runs[0]) > 50
    for other in runs[1:]:
        for u, v in zip(runs[0], other):
            assert torch.equal(u, v)
