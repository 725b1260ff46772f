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
    assert _lib.lib().psnd_adam_step(None, 1, None, None, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, None, None, None) == -1
