"""psnd_adam_step (pytorch_sound_amd.optim.Adam / AdamW) against torch.optim.Adam / AdamW run in float64 on the CPU with
the same gradients: parameters and both moments after 1, 2 and 25 steps (tolerance 2e-6 relative to each tensor's max:
fp32 kernel vs f64), odd sizes and unaligned views, weight decay of both kinds, the AMP found_inf / grad_scale protocol,
state_dict exchange with torch.optim.Adam."""
import pytest
import torch

from pytorch_sound_amd import optim as O

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
SHAPES = [(256, 256, 3), (256,), (1,), (7, 33), (5000,), (2049,), (1 << 20,)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, s in enumerate(SHAPES):
        t = torch.randn(*s, generator=g)
        if i == 3:      # an unaligned (4-byte offset) view as a leaf
            base = torch.zeros(t.numel() + 1)
            base[1:] = t.flatten()
            t = base[1:].view(*s)
        out.append(t)
    return out


def _run(cls_gpu, cls_ref, steps, **kw):
    host = [p.double().clone().requires_grad_(True) for p in _params(0)]
    dev = []
    for p in _params(0):
        if p.storage_offset():
            base = torch.zeros(p.numel() + 1, device=DEV)
            base[1:] = p.flatten().to(DEV)
            dev.append(base[1:].view(p.shape).requires_grad_(True))
        else:
            dev.append(p.to(DEV).requires_grad_(True))
    og, orf = cls_gpu(dev, **kw), cls_ref(host, **kw)
    for s in range(steps):
        grads = _params(100 + s)
        for p, q, g in zip(dev, host, grads):
            p.grad = g.to(DEV) if not p.storage_offset() else g.to(DEV).clone()
            q.grad = g.double()
        og.step()
        orf.step()
    return dev, host, og, orf


def _close(a, b, tol=2e-6):
    b = b.detach().cpu().double() if torch.is_tensor(b) else b
    a = a.detach().cpu().double()
    return float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize('steps', [1, 2, 25])
@pytest.mark.parametrize('kind,kw', [('adam', dict(lr=2e-4, betas=(0.8, 0.99))), ('adam', dict(lr=1e-3, weight_decay=0.1)),
                                     ('adamw', dict(lr=1e-3, weight_decay=0.05, eps=1e-6))])
def test_adam_matches_torch_f64(kind, kw, steps):
    gpu_cls, ref_cls = (O.Adam, torch.optim.Adam) if kind == 'adam' else (O.AdamW, torch.optim.AdamW)
    dev, host, og, orf = _run(gpu_cls, ref_cls, steps, **kw)
    for p, q in zip(dev, host):
        assert _close(p, q), p.shape
        assert _close(og.state[p]['exp_avg'], orf.state[q]['exp_avg']) and _close(og.state[p]['exp_avg_sq'], orf.state[q]['exp_avg_sq'])
        assert float(og.state[p]['step']) == steps


def test_found_inf_skips_update_and_step_count():
    dev, host, og, _ = _run(O.Adam, torch.optim.Adam, 2, lr=1e-3)
    before = [p.detach().clone() for p in dev]
    m_before = [og.state[p]['exp_avg'].clone() for p in dev]
    og.found_inf = torch.ones((), device=DEV)
    og.grad_scale = None
    og.step()
    for p, b, m in zip(dev, before, m_before):
        assert torch.equal(p, b) and torch.equal(og.state[p]['exp_avg'], m) and float(og.state[p]['step']) == 2
    og.found_inf = torch.zeros((), device=DEV)
    og.step()
    del og.found_inf, og.grad_scale
    assert all(float(og.state[p]['step']) == 3 for p in dev) and not torch.equal(dev[0], before[0])


def test_grad_scale_divides_gradients():
    a, _, oa, _ = _run(O.Adam, torch.optim.Adam, 1, lr=1e-3)
    b = [p.detach().clone().requires_grad_(True) for p in [q.to(DEV) for q in _params(0)]]
    ob = O.Adam(b, lr=1e-3)
    for p, g in zip(b, _params(100)):
        p.grad = (g * 1024.0).to(DEV)
    ob.grad_scale = torch.full((), 1024.0, device=DEV)
    ob.found_inf = torch.zeros((), device=DEV)
    ob.step()
    for p, q in zip(a, b):
        assert _close(q, p, 1e-6)


def test_state_dict_roundtrip_with_torch_adam():
    dev, host, og, _ = _run(O.Adam, torch.optim.Adam, 3, lr=2e-4, betas=(0.8, 0.99))
    import copy
    sd = copy.deepcopy(og.state_dict())       # as torch.save / torch.load would: load_state_dict itself keeps references
    # into torch's own (fused) Adam and one more step there == one more step here
    twins = [p.detach().clone().requires_grad_(True) for p in dev]
    ot = torch.optim.Adam(twins, lr=2e-4, betas=(0.8, 0.99), fused=True)
    ot.load_state_dict(sd)
    grads = _params(777)
    for p, q, g in zip(dev, twins, grads):
        p.grad = g.to(DEV).clone()
        q.grad = g.to(DEV).clone()
    og.step()
    ot.step()
    for p, q in zip(dev, twins):
        assert _close(p, q, 2e-6)
    # and back: a state dict written by a non-fused torch Adam (host step counts) loads here
    on = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in dev], lr=2e-4, betas=(0.8, 0.99))
    for q, g in zip(on.param_groups[0]['params'], grads):
        q.grad = g.to(DEV)
    on.step()
    o2 = O.Adam([p.detach().clone().requires_grad_(True) for p in dev], lr=2e-4, betas=(0.8, 0.99))
    o2.load_state_dict(copy.deepcopy(on.state_dict()))
    for q, g in zip(o2.param_groups[0]['params'], grads):
        q.grad = g.to(DEV)
    o2.step()
    assert all(float(o2.state[q]['step']) == 2 for q in o2.param_groups[0]['params'])


def test_rejects_what_it_cannot_do():
    from pytorch_sound_amd import _lib
    with pytest.raises(NotImplementedError):
        O.Adam([torch.zeros(3, device=DEV, requires_grad=True)], amsgrad=True)
    with pytest.raises(ValueError):
        O.Adam([torch.zeros(3, device=DEV, requires_grad=True)], betas=(1.0, 0.9))
    p = torch.zeros(3, requires_grad=True)
    p.grad = torch.ones(3)
    with pytest.raises(_lib.PsndError):
        O.Adam([p]).step()                                    # CPU parameter: no fallback
    assert _lib.lib().psnd_adam_step(None, 1, None, None, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, None, None, 0.0, None, None) == -1


@pytest.mark.parametrize('clip_value,max_norm,scale', [(0.5, 0.0, None), (0.0, 3.0, None), (0.7, 5.0, None), (0.7, 5.0, 2.0),
                                                       (0.0, 1e9, None)])
def test_fused_clip_matches_trainer_clip_grad(clip_value, max_norm, scale):
    """K18: `optimizer.fused_clip = (grad_clip, grad_norm)` = Trainer.clip_grad of the reference (trainer.py:184-191: per-parameter
    clamp, then torch.nn.utils.clip_grad_norm_) followed by the step, evaluated in float64 on the host: parameters, moments, the
    reported norm; the gradients in memory stay untouched.  Two parameter groups (the norm is global), a grad_scale (the data-
    parallel division by the world size) applied BEFORE the clamp, as averaging before clipping does."""
    host = [p.double().clone().requires_grad_(True) for p in _params(0)]
    dev = [p.to(DEV).clone().requires_grad_(True) for p in _params(0)]
    og = O.Adam([{'params': dev[:3]}, {'params': dev[3:], 'lr': 3e-3}], lr=1e-3)
    orf = torch.optim.Adam([{'params': host[:3]}, {'params': host[3:], 'lr': 3e-3}], lr=1e-3)
    for s in range(3):
        grads = _params(200 + s)
        for p, q, g in zip(dev, host, grads):
            p.grad = g.to(DEV).clone()
            q.grad = g.double() / (scale or 1.0)
        keep = [p.grad.clone() for p in dev]
        if clip_value:
            for q in host:
                q.grad = q.grad.clamp(-clip_value, clip_value)
        norm = torch.nn.utils.clip_grad_norm_(host, max_norm) if max_norm else None
        og.fused_clip = (clip_value, max_norm)
        if scale:
            og.grad_scale = torch.tensor(scale, device=DEV)
        og.step()
        del og.fused_clip
        if scale:
            del og.grad_scale
        orf.step()
        for p, k in zip(dev, keep):
            assert torch.equal(p.grad, k)                                   # p.grad is read, never rewritten
        if max_norm:
            assert abs(float(og.last_grad_norm) - float(norm)) <= 2e-6 * float(norm)
    for p, q in zip(dev, host):
        assert _close(p, q, 4e-6)
        assert _close(og.state[p]['exp_avg'], orf.state[q]['exp_avg'], 4e-6)
        assert _close(og.state[p]['exp_avg_sq'], orf.state[q]['exp_avg_sq'], 8e-6)


@pytest.mark.parametrize('clip_value,max_norm', [(0.0, 0.0), (0.5, 0.0), (0.5, 3.0), (0.0, 3.0)])
def test_nan_gradient_propagates_like_torch_clamp(clip_value, max_norm):
    """fminf / fmaxf return the non-NaN operand, torch.clamp + clip_grad_norm_ (trainer.py:184-191) propagate NaN.  A NaN
    gradient element with a finite loss must poison its parameter visibly (and, through the global norm, every parameter when
    grad_norm is on) - not turn into a maximal finite update."""
    host = [p.double().clone().requires_grad_(True) for p in _params(0)]
    dev = [p.to(DEV).clone().requires_grad_(True) for p in _params(0)]
    og, orf = O.Adam(dev, lr=1e-3), torch.optim.Adam(host, lr=1e-3)
    grads = _params(300)
    grads[0].view(-1)[5] = float('nan')
    for p, q, g in zip(dev, host, grads):
        p.grad = g.to(DEV).clone()
        q.grad = g.double()
    if clip_value:
        for q in host:
            q.grad = q.grad.clamp(-clip_value, clip_value)
    if max_norm:
        torch.nn.utils.clip_grad_norm_(host, max_norm)
    og.fused_clip = (clip_value, max_norm)
    og.step()
    orf.step()
    for p, q in zip(dev, host):
        a, b = p.detach().cpu(), q.detach()
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.isfinite(a[~torch.isnan(a)]).all()
    assert torch.isnan(dev[0].detach().view(-1)[5]).item()
    if max_norm:
        assert torch.isnan(og.last_grad_norm).item()


def test_trainer_folds_clip_grad_into_the_optimizer_step(tmp_path):
    """Trainer with grad_clip / grad_norm and the HIP optimizer: the device-skip path hands both to the optimizer (no per-parameter
    clamp, no torch norm kernels) - equal to the same training with the reference's eager clip_grad() and to float64 on the host;
    a Trainer subclass that overrides clip_grad() keeps being called."""
    from pytorch_sound_amd.trainer import Trainer, LogType

    class T(Trainer):
        def forward(self, x, y, is_logging=False):
            loss = torch.nn.functional.mse_loss(self.model(x), y)
            return loss, {'loss': (loss, LogType.SCALAR)}

    calls = []

    class TOverride(T):
        def clip_grad(self):
            calls.append(1)
            super().clip_grad()

    def net():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Conv1d(4, 16, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv1d(16, 2, 1))

    g = torch.Generator().manual_seed(5)
    data = [(torch.randn(8, 4, 32, generator=g), 3 * torch.randn(8, 2, 32, generator=g)) for _ in range(6)]
    out = {}
    for name, cls in (('fused', T), ('override', TOverride)):
        m = net().to(DEV)
        tr = cls(m, O.Adam(m.parameters(), lr=1e-2), data, data[:1], max_step=6, valid_max_step=1, save_interval=10 ** 6,
                 log_interval=10 ** 6, save_dir=str(tmp_path), save_prefix=name, grad_clip=0.05, grad_norm=0.1, seed=1)
        m.train()
        for i in range(1, 7):
            tr.step = i
            tr.train(i)
        torch.cuda.synchronize()
        out[name] = {k: v.double().cpu() for k, v in m.state_dict().items()}
    assert len(calls) == 6
    m = net().double()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    for x, y in data:
        opt.zero_grad()
        torch.nn.functional.mse_loss(m(x.double()), y.double()).backward()
        for p in m.parameters():
            p.grad = p.grad.clamp(-0.05, 0.05)
        torch.nn.utils.clip_grad_norm_(m.parameters(), 0.1)
        opt.step()
    for k, v in m.state_dict().items():
        assert (out['fused'][k] - v).abs().max() <= 2e-5 * max(1.0, float(v.abs().max())), k
        assert (out['override'][k] - v).abs().max() <= 2e-5 * max(1.0, float(v.abs().max())), k


def test_grad_pack_unpack_bf16_round_to_nearest_even():
    """psnd_grad_pack_bf16 / psnd_grad_unpack_bf16 (the bf16 wire of FlatGradReducer): pack = torch's fp32 -> bf16 conversion bit for bit
    (round to nearest even, NaN / inf / denormals included), with a scale; unpack is exact; n must be a multiple of 8"""
    from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check, PsndError
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(8 * 1237, generator=g) * torch.logspace(-30, 30, 8 * 1237)).to(dev)
    x[:8] = torch.tensor([0.0, -0.0, float('inf'), -float('inf'), float('nan'), 1.0039062, 1.0117188, 3.0e-41], device=dev)   # ties, a denormal
    out = torch.empty(x.numel(), dtype=torch.bfloat16, device=dev)
    back = torch.empty_like(x)
    st = stream_ptr(dev)
    for scale in (1.0, 0.125):
        check(lib().psnd_grad_pack_bf16(ptr(x), ptr(out), x.numel(), scale, st), 'pack')
        want = (x * scale).to(torch.bfloat16)
        same = (out.view(torch.int16) == want.view(torch.int16)) | (torch.isnan(out.float()) & torch.isnan(want.float()))
        assert bool(same.all()), int((~same).sum())
        check(lib().psnd_grad_unpack_bf16(ptr(out), ptr(back), x.numel(), 4.0, st), 'unpack')
        ref = want.float() * 4.0
        ok = (back == ref) | (torch.isnan(back) & torch.isnan(ref))
        assert bool(ok.all())
    with pytest.raises(PsndError):
        check(lib().psnd_grad_pack_bf16(ptr(x), ptr(out), 12, 1.0, st), 'pack')
