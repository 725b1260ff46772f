"""Trainer.graph_steps = 'auto' (round 6) and the adoption of a stock torch.optim.Adam: a step whose forward() is pure tensor code is captured
without the caller doing anything and trains like the eager loop; a forward() that draws host-side random numbers, reads the step count
or synchronises with the host keeps running eagerly (the reason logged once) - never a frozen host-side decision."""
import random
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Conv1d(4, 16, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv1d(16, 2, 1)).cuda()


def _data(n=12):
    g = torch.Generator().manual_seed(5)
    return [(torch.randn(4, 4, 32, generator=g).cuda(), torch.randn(4, 2, 32, generator=g).cuda()) for _ in range(n)]


def _trainer(fwd, opt_cls=torch.optim.Adam, **attrs):
    from pytorch_sound_amd.trainer import Trainer, LogType

    class T(Trainer):
        def forward(self, x, y, is_logging=False):
            loss = fwd(self, x, y)
            return loss, {'loss': (loss, LogType.SCALAR)}

    net = _net()
    data = _data()
    tr = T(net, opt_cls(net.parameters(), lr=1e-2), data, data[:1], max_step=12, valid_max_step=1, save_interval=10 ** 6, log_interval=10 ** 6,
           save_dir=tempfile.mkdtemp(prefix='psnd_auto_'), seed=1)
    for k, v in attrs.items():
        setattr(tr, k, v)
    net.train()
    for i in range(1, 13):
        tr.step = i
        tr.train(i)
    torch.cuda.synchronize()
    return tr, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


def _captured(tr):
    return any('graph' in v for v in getattr(tr, '_graphs', {}).values())


def test_pure_forward_is_captured_by_default_and_trains_like_eager():
    from pytorch_sound_amd import optim as poptim
    plain = lambda self, x, y: F.mse_loss(self.model(x), y)            # noqa: E731
    tr, w = _trainer(plain)
    assert tr.graph_steps == 'auto' and tr._graph_auto_ok is True and _captured(tr)
    assert type(tr.optimizer) is poptim.Adam and tr._opt_adopted                           # the stock torch.optim.Adam, adopted in place
    tr0, w0 = _trainer(plain, graph_steps=False)
    assert not _captured(tr0)
    for k in w:
        assert float((w[k] - w0[k]).abs().max()) <= 1e-5 * max(1.0, float(w0[k].abs().max())), k   # replayed steps follow eager steps (torch's convolutions: the library may pick other algorithms under capture)
    ref = _net().double()                                                                   # and both follow torch.optim.Adam in float64
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for x, y in _data():
        opt.zero_grad()
        F.mse_loss(ref(x.double()), y.double()).backward()
        opt.step()
    for k, v in ref.state_dict().items():
        assert float((v.cpu() - w[k].double()).abs().max()) <= 2e-5 * max(1.0, float(v.abs().max())), k
    sd = tr.optimizer.state_dict()                                                          # the state stays torch.optim.Adam's
    fresh = torch.optim.Adam(_net().parameters(), lr=1e-2)
    fresh.load_state_dict(sd)
    assert set(next(iter(sd['state'].values()))) == {'step', 'exp_avg', 'exp_avg_sq'}


@pytest.mark.parametrize('kind', ['python_random', 'numpy_random', 'torch_cpu_random', 'reads_step', 'host_sync'])
def test_host_dependent_forward_stays_eager(kind):
    calls = []

    def fwd(self, x, y):
        calls.append(1)
        if kind == 'python_random':
            s = 1.0 + 0.01 * random.random()
        elif kind == 'numpy_random':
            s = 1.0 + 0.01 * np.random.rand()
        elif kind == 'torch_cpu_random':
            s = 1.0 + 0.01 * float(torch.rand(()))
        elif kind == 'reads_step':
            s = 1.0 + 0.001 * self.step
        else:
            s = 1.0
        loss = F.mse_loss(self.model(x), y) * s
        if kind == 'host_sync':
            loss = loss * (1.0 if float(loss) > 0 else 0.5)                                  # a host synchronisation: not capturable
        return loss

    tr, _ = _trainer(fwd)
    assert tr._graph_auto_ok is False and not _captured(tr)
    assert len(calls) >= 12                                                                 # forward() really ran every step


def test_explicit_switch_and_non_adoptable_optimizers():
    plain = lambda self, x, y: F.mse_loss(self.model(x), y)            # noqa: E731
    tr, _ = _trainer(plain, adopt_optimizer=False)
    assert type(tr.optimizer) is torch.optim.Adam and not _captured(tr)                     # not adopted: torch's own (non-fused) step, eager
    tr, _ = _trainer(plain, opt_cls=lambda p, lr: torch.optim.Adam(p, lr=lr, amsgrad=True))
    assert type(tr.optimizer) is torch.optim.Adam and not tr._opt_adopted                   # amsgrad: left alone
    tr, _ = _trainer(plain, opt_cls=lambda p, lr: torch.optim.SGD(p, lr=lr))
    assert type(tr.optimizer) is torch.optim.SGD


def test_resume_with_an_adopted_optimizer(tmp_path):
    """save() / load() round trip of a Trainer whose stock torch.optim.Adam was adopted: the resumed run continues the moments and step counts
    (a fresh stock optimizer loads the checkpoint, is adopted again) and ends where the uninterrupted run ends"""
    from pytorch_sound_amd.trainer import Trainer, LogType

    class T(Trainer):
        def forward(self, x, y, is_logging=False):
            loss = F.mse_loss(self.model(x), y)
            return loss, {'loss': (loss, LogType.SCALAR)}

    data = _data(10)

    def make(save_dir):
        net = _net()
        tr = T(net, torch.optim.Adam(net.parameters(), lr=1e-2), data, data[:1], max_step=10, valid_max_step=1, save_interval=10 ** 6,
               log_interval=10 ** 6, save_dir=str(save_dir), save_prefix='r', seed=1)
        tr.graph_steps = False                       # (eager: the resumed Trainer sees the batches in the same order)
        net.train()
        return tr, net

    def steps(tr, a, b):
        for i in range(a, b + 1):
            tr.step = i
            x, y = data[i - 1]
            tr.optimizer.zero_grad()
            loss, _ = tr._forward_resolved(x, y)
            loss.backward()
            tr._maybe_adopt_optimizer()
            tr.optimizer.step()

    tr, net = make(tmp_path / 'a')
    steps(tr, 1, 10)
    want = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    tr, net = make(tmp_path / 'b')
    steps(tr, 1, 5)
    tr.save(5)
    tr2, net2 = make(tmp_path / 'b')                 # loads the step-5 checkpoint in its constructor
    assert tr2.step == 5
    steps(tr2, 6, 10)
    assert tr2._opt_adopted
    for k, v in net2.state_dict().items():
        assert float((v.cpu() - want[k]).abs().max()) <= 1e-6 * max(1.0, float(want[k].abs().max())), k
