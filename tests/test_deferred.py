"""deferred.py on the host: the recorded pattern of the reference's loss recipe resolves to the same numbers and gradients as the ops run
one by one; any use outside the pattern materialises the operand and runs as written (CPU: no fused node, the term-by-term path)."""
import pytest
import torch
import torch.nn.functional as F

from pytorch_sound_amd import deferred as D
from pytorch_sound_amd.models.transforms import LogMelSpectrogram


class _HostEst(D.Est):
    """Est with a torch formulation of the mask head (the CL kernels need the GPU)"""

    def __init__(self, logits, mag):
        self.y, self.mag, self.cl_shape = logits, mag, None
        self.shape, self.device, self.requires_grad = tuple(mag.shape), mag.device, logits.requires_grad
        self.count = 0

    def _materialize(self):
        self.count += 1
        return torch.sigmoid(self.y) * self.mag


def _setup():
    torch.manual_seed(0)
    fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0, 8000)
    logits = torch.randn(2, 513, 9, requires_grad=True)
    mag, mag_ref = torch.rand(2, 513, 9) * 3, torch.rand(2, 513, 9) * 3
    mel_ref = torch.randn(2, 80, 9)
    return fe, logits, mag, mag_ref, mel_ref


def _recipe(fe, est, mag_ref, mel_ref):
    est = est.float()
    mel_est = torch.log(torch.matmul(fe.mel_filter, est) + 1e-6).clamp(fe.min_db, fe.max_db)
    return F.l1_loss(est, mag_ref) + 0.5 * F.l1_loss(mel_est, mel_ref)


def test_recipe_is_recorded_and_resolves_to_the_same_loss_and_gradient():
    fe, logits, mag, mag_ref, mel_ref = _setup()
    want = _recipe(fe, torch.sigmoid(logits) * mag, mag_ref, mel_ref)
    want.backward()
    gw = logits.grad.clone()
    logits.grad = None
    node = _HostEst(logits, mag)
    loss = _recipe(fe, D.Deferred(node), mag_ref, mel_ref)
    assert isinstance(loss, D.Deferred) and isinstance(loss._node, D.Sum) and node.count == 0       # nothing ran yet
    assert [round(w, 6) for w, _, _ in loss._node.terms] == [1.0, 0.5]
    assert isinstance(loss._node.terms[1][1], D.LogMel) and loss._node.terms[1][1].eps == 1e-6
    real = D.resolve(loss)
    assert not isinstance(real, D.Deferred) and node.count == 1
    real.backward()
    assert torch.allclose(real, want, rtol=1e-6, atol=1e-7)
    assert torch.allclose(logits.grad, gw, rtol=1e-5, atol=1e-7)


def test_metadata_does_not_materialise():
    fe, logits, mag, _, _ = _setup()
    node = _HostEst(logits, mag)
    e = D.Deferred(node)
    assert e.shape == (2, 513, 9) and e.dim() == 3 and e.dtype == torch.float32 and e.numel() == 2 * 513 * 9 and not e.is_cuda
    assert e.size(1) == 513 and len(e) == 2 and e.requires_grad and isinstance(e, torch.Tensor)
    m = torch.matmul(fe.mel_filter, e)
    assert isinstance(m, D.Deferred) and m.shape == (2, 80, 9)
    assert node.count == 0


def test_any_other_use_runs_as_written():
    fe, logits, mag, mag_ref, mel_ref = _setup()
    ref = torch.sigmoid(logits) * mag
    node = _HostEst(logits, mag)
    e = D.Deferred(node)
    assert torch.equal((e - mag_ref).abs().max(), (ref - mag_ref).abs().max())          # an op outside the pattern
    assert node.count == 1
    assert torch.equal(e[0, :3, :2], ref[0, :3, :2]) and torch.equal(e.detach().mean(), ref.detach().mean())
    assert torch.equal(F.mse_loss(e, mag_ref), F.mse_loss(ref, mag_ref))
    assert torch.equal(F.l1_loss(e, mag_ref, reduction='sum'), F.l1_loss(ref, mag_ref, reduction='sum'))       # other reduction: as written
    other = torch.rand(80, 513)
    assert torch.allclose(torch.matmul(other, e), torch.matmul(other, ref))                  # not a registered mel filter: as written
    assert torch.allclose(torch.log(torch.matmul(fe.mel_filter, e) + 1e-6), torch.log(torch.matmul(fe.mel_filter, ref) + 1e-6))
    x = torch.log(torch.matmul(fe.mel_filter, e) + 1e-6) * 2.0                              # a recorded prefix used outside the pattern
    assert torch.allclose(x, torch.log(torch.matmul(fe.mel_filter, ref) + 1e-6) * 2.0)
    assert node.count == 1                                                                  # the estimate was formed once
    s = F.l1_loss(e, mag_ref) / 4 + F.l1_loss(e, mag_ref) * 0.25
    assert isinstance(s, D.Deferred) and torch.allclose(D.resolve(s), F.l1_loss(ref, mag_ref) / 2)
    assert torch.allclose(F.l1_loss(e, mag_ref) + 1.0, F.l1_loss(ref, mag_ref) + 1.0)        # scalar + number: outside the pattern
    assert float(F.l1_loss(e, mag_ref)) == pytest.approx(float(F.l1_loss(ref, mag_ref)))


def test_trainer_resolves_what_forward_returns():
    from pytorch_sound_amd.trainer import Trainer, LogType
    fe, logits, mag, mag_ref, mel_ref = _setup()

    class T(Trainer):
        def forward(self, *inputs, is_logging=False):
            loss = _recipe(fe, D.Deferred(_HostEst(logits, mag)), mag_ref, mel_ref)
            return loss, {'loss': (loss, LogType.SCALAR), 'mag': (F.l1_loss(D.Deferred(_HostEst(logits, mag)), mag_ref), LogType.SCALAR)}

    t = T.__new__(T)
    loss, meta = t._forward_resolved()
    assert not isinstance(loss, D.Deferred) and meta['loss'][0] is loss and meta['loss'][1] == LogType.SCALAR
    assert not isinstance(meta['mag'][0], D.Deferred) and meta['mag'][0].dim() == 0


def test_custom_autograd_functions_get_the_real_tensor():
    """torch.autograd.Function.apply does not go through __torch_function__: a Deferred operand is resolved first, so the gradient edge exists"""
    fe, logits, mag, mag_ref, _ = _setup()

    class Double(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            assert not isinstance(x, D.Deferred)
            return x * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    e = D.Deferred(_HostEst(logits, mag))
    y = Double.apply(e)
    assert y.requires_grad and y.grad_fn is not None
    y.sum().backward()
    want = torch.autograd.grad((torch.sigmoid(logits) * mag * 2).sum(), logits)[0]
    assert torch.allclose(logits.grad, want)
