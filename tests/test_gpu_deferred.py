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
        assert real.grad_fn is not None and 'MaskHeadSpectralL1' in type(real.grad_fn).__name__       # (NFK: the lazy magnitudes of STFT.transform)
        assert getattr(real, 'psnd_nan_flag', None) is not None
        real.backward()
    assert abs(float(real) - float(loss0)) <= 2e-6 * abs(float(loss0))
    e_, e0 = D.resolve(est), est0.detach()                   # the estimate itself: the two ways into the channels-last layout round log1p(mag) to bf16 in different kernels
    assert float((e_ - e0).norm() / e0.norm()) <= 2e-3
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


def test_transform_hands_out_a_lazy_bin_fastest_magnitude():
    """STFT.transform without a gradient: the magnitude is a deferred (N, F, K) result standing for (N, K, F), the phase a deferred node; every
    plain use gives exactly what the eager transform gives"""
    from pytorch_sound_amd import deferred as D
    from pytorch_sound_amd.models.transforms import STFT
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    stft = STFT(1024, 256).to(dev)
    x = (0.1 * torch.randn(3, 9000, device=dev)).clamp(-1, 1)
    D.ENABLED = False
    try:
        mag0, ph0 = stft.transform(x)
    finally:
        D.ENABLED = True
    mag, ph = stft.transform(x)
    assert isinstance(mag, D.Deferred) and isinstance(mag._node, D.MagNFK) and isinstance(ph, D.Deferred)
    assert mag.shape == mag0.shape and ph.shape == ph0.shape and mag._node._real is None and ph._node._real is None
    assert float((mag + 0 - mag0).abs().max()) <= 2e-6 * float(mag0.abs().max()) and mag._node._real is not None     # the bin-fastest kernel, transposed once
    assert torch.equal(D.resolve(ph), ph0)
    assert not stft._lazy_transform                                                           # a magnitude was transposed, a phase used: this module
    m_e, p_e = stft.transform(x)                                                              # computes both at once from now on
    assert not isinstance(m_e, D.Deferred) and torch.equal(m_e, mag0) and torch.equal(p_e, ph0)
    rec = stft.inverse(mag, ph)                                                               # deferred operands into a kernel wrapper
    assert torch.allclose(rec, x[:, :rec.shape[1]], atol=2e-5)
    xg = x.clone().requires_grad_(True)                                                       # a gradient is wanted: the eager autograd path
    mg, _ = stft.transform(xg)
    assert not isinstance(mg, D.Deferred) and mg.grad_fn is not None
    stft._lazy_transform = True                                                               # probe again
    w = x.clone()
    m3, p3 = stft.transform(w)
    w.mul_(2.0)                                                                               # the waveform changes before the phase is used
    with pytest.raises(RuntimeError, match='modified in place'):
        D.resolve(p3)
    assert float((D.resolve(m3) - mag0).abs().max()) <= 2e-6 * float(mag0.abs().max())           # the magnitude was formed at the call


def test_dropin_recipe_runs_bin_fastest_end_to_end():
    """the recipe of test_dropin_recipe_resolves_to_the_fused_node with STFT.transform's lazy magnitudes: the (N, F, K) fused node, no transposed
    copy of any magnitude, same loss and gradients as the plain ops"""
    from pytorch_sound_amd import deferred as D
    fe, model, noisy, clean = _setup()
    D.ENABLED = False
    try:
        loss0, _ = _step(fe, model, noisy, clean)
        loss0.backward()
        g0 = {n: p.grad.clone() for n, p in model.named_parameters()}
        model.zero_grad(set_to_none=True)
    finally:
        D.ENABLED = True
    loss, est = _step(fe, model, noisy, clean)
    node = loss._node
    assert isinstance(node, D.Sum) and est._node.mag_node is not None
    real = D.resolve(loss)
    assert 'MaskHeadSpectralL1NFK' in type(real.grad_fn).__name__
    assert est._node.mag_node._real is None                                                  # never transposed
    real.backward()
    assert abs(float(real.detach()) - float(loss0.detach())) <= 2e-6 * abs(float(loss0.detach()))
    worst = max(float((p.grad - g0[n]).norm() / g0[n].norm().clamp_min(1e-20)) for n, p in model.named_parameters())
    assert worst <= 2e-3, worst
    e = D.resolve(est)
    assert e.shape == (noisy.shape[0], 513, est.shape[2]) and torch.isfinite(e).all()
