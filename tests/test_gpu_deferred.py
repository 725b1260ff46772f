"""deferred.py on the GPU: configs[1] written with the reference's API names only (the drop-in step of bench.py: STFT.transform, the model
under autocast, LogMelSpectrogram's three lines on `mel_filter`, F.l1_loss) resolves to the fused loss node - same loss, same parameter
gradients as the ops run one by one on plain tensors (deferred.ENABLED = False), with no library GEMM / reduction in the step."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(N=3, T=6000):
    from pytorch_sound_amd.models import build_model
    from pytorch_sound_amd.models import separator  # noqa: F401
    from pytorch_sound_amd.models.transforms import LogMelSpectrogram
    torch.manual_seed(7)
    dev = torch.device('cuda:0')
    fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0, 8000).to(dev)
    model = build_model('conv_separator_voicebank', {'channels': 64, 'num_blocks': 2}).to(dev)
    clean = (0.07 * torch.randn(N, T, device=dev)).clamp(-1, 1)
    noisy = (clean + 0.03 * torch.randn(N, T, device=dev)).clamp(-1, 1)
    return fe, model, noisy, clean


def _step(fe, model, noisy, clean):
    with torch.no_grad():
        mag_ref, _ = fe.stft.transform(clean)
        mel_ref = fe(clean)
    mag_mix, _ = fe.stft.transform(noisy)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        est = model(mag_mix)
    est = est.float()
    mel_est = torch.log(torch.matmul(fe.mel_filter, est) + 1e-6).clamp(fe.min_db, fe.max_db)
    loss = F.l1_loss(est, mag_ref) + 0.5 * F.l1_loss(mel_est, mel_ref)
    return loss, est


def test_dropin_recipe_resolves_to_the_fused_node():
    from pytorch_sound_amd import deferred as D
    from test_gpu_no_library_paths import forbid_library_ops
    fe, model, noisy, clean = _setup()
    D.ENABLED = False
    try:
        loss0, est0 = _step(fe, model, noisy, clean)
        assert not isinstance(loss0, D.Deferred)
        loss0.backward()
        g0 = {n: p.grad.clone() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
    finally:
        D.ENABLED = True
    with forbid_library_ops():                               # no GEMM / reduction library call: the whole recipe is recorded
        loss, est = _step(fe, model, noisy, clean)
        assert isinstance(loss, D.Deferred) and isinstance(loss._node, D.Sum) and isinstance(est, D.Deferred)
        real = D.resolve(loss)
        assert real.grad_fn is not None and 'MaskHeadSpectralL1CL' in type(real.grad_fn).__name__
        assert getattr(real, 'psnd_nan_flag', None) is not None
        real.backward()
    assert abs(float(real) - float(loss0)) <= 2e-6 * abs(float(loss0))
    assert torch.allclose(D.resolve(est), est0.detach(), rtol=0, atol=0)                  # the estimate itself: the same kernel arithmetic
    worst = 0.0
    for n, p in model.named_parameters():
        worst = max(worst, float((p.grad - g0[n]).norm() / g0[n].norm().clamp_min(1e-20)))
    assert worst <= 2e-3, worst                              # bf16 gradient tensors: the fused node rounds (g_mag + g_mel) once, the plain path each


def test_other_uses_of_the_estimate_still_work():
    from pytorch_sound_amd import deferred as D
    fe, model, noisy, clean = _setup(2, 4000)
    with torch.no_grad():
        mag_ref, _ = fe.stft.transform(clean)
    mag_mix, _ = fe.stft.transform(noisy)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        est = model(mag_mix)
    assert isinstance(est, D.Deferred) and est.shape == mag_mix.shape and est.is_cuda
    snr = 10 * torch.log10(mag_ref.pow(2).sum() / (est - mag_ref).pow(2).sum())          # a metric outside the pattern: forms the estimate
    assert torch.isfinite(snr)
    loss = F.mse_loss(est, mag_ref) + F.l1_loss(est, mag_ref)                             # mixed: mse as written, l1 recorded, the sum as written
    assert not isinstance(loss, D.Deferred)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        e2 = model(mag_mix)
    assert torch.equal(e2.cpu(), D.resolve(est).detach().cpu())


def test_trainer_graph_steps_on_the_recipe_train_like_the_plain_ops():
    """four captured steps of the drop-in Trainer with the deferred estimate against four with plain tensors: same loss trajectory"""
    import tempfile
    from pytorch_sound_amd import deferred as D, optim as poptim
    from pytorch_sound_amd.trainer import Trainer, LogType
    from pytorch_sound_amd.models import build_model

    def run(enabled):
        D.ENABLED = enabled
        try:
            fe, model, noisy, clean = _setup(2, 5000)
            both = torch.cat([noisy, clean])
            losses = []

            class Step(Trainer):
                def forward(self, both, is_logging=False):
                    n = both.shape[0] // 2
                    loss, _ = _step(fe, self.model, both[:n], both[n:])
                    return loss, {'loss': (loss, LogType.SCALAR)}

            pool = [(both,)]
            tr = Step(model, poptim.Adam(model.parameters(), lr=2e-4, betas=(0.8, 0.99)), pool, pool, max_step=10 ** 9, valid_max_step=1,
                      save_interval=10 ** 9, log_interval=10 ** 9, save_dir=tempfile.mkdtemp(prefix='psnd_def_'), seed=1)
            tr.graph_steps = True
            model.train()
            for i in range(tr.graph_warmup + 5):
                tr.train(i)
            torch.cuda.synchronize()
            with torch.no_grad():
                l, _ = tr._forward_resolved(both)
            return float(l)
        finally:
            D.ENABLED = True

    a, b = run(True), run(False)
    assert abs(a - b) <= 2e-3 * abs(b), (a, b)
