"""Trainer against the imported reference's 6-step CPU run (G7): weights after training (incl. the
NaN-skipped step, grad clamp + norm clip, scheduler), log lines, checkpoint files/keys, resume."""
import io
import logging
import os
import numpy as np
import torch

from pytorch_sound_amd.trainer import Trainer, LogType
from pytorch_sound_amd.utils.commons import LOGGER


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.l1 = torch.nn.Linear(8, 16)
        self.l2 = torch.nn.Linear(16, 1)

    def forward(self, x):
        return self.l2(torch.tanh(self.l1(x)))


class T(Trainer):
    poison = False

    def forward(self, x, y, is_logging=False):
        o = self.model(x)
        loss = torch.nn.functional.mse_loss(o, y)
        if self.poison and self.step == 3 and self.model.training:
            loss = loss * float('nan')
        return loss, {'loss': (loss.item(), LogType.SCALAR), 'mae': ((o - y).abs().mean().item(), LogType.SCALAR)}


def _data(g, split, n):
    return [(torch.from_numpy(g['%s/x%d' % (split, i)]), torch.from_numpy(g['%s/y%d' % (split, i)])) for i in range(n)]


def test_six_step_run_matches_reference(golden, tmp_path, monkeypatch):
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)     # keep this test on the CPU path
    g = golden('trainer')
    buf = io.StringIO()
    hdl = logging.StreamHandler(buf)
    hdl.setFormatter(logging.Formatter('%(message)s'))
    LOGGER.addHandler(hdl)
    try:
        net = Net()
        net.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('init/')})
        train, valid = _data(g, 'train', 5), _data(g, 'valid', 3)
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
        sch = torch.optim.lr_scheduler.StepLR(opt, 2, 0.5)
        T.poison = True
        tr = T(net, opt, train, valid, max_step=6, valid_max_step=3, save_interval=3, log_interval=2,
               save_dir=str(tmp_path), save_prefix='exp', grad_clip=0.5, grad_norm=1.0, scheduler=sch, seed=99)
        best = tr.run()
        assert abs(float(best) - float(g['run/best_valid'])) <= 1e-6
        for k, v in net.state_dict().items():
            assert np.abs(v.numpy() - g['final/' + k]).max() <= 1e-6, k
        files = sorted(os.path.relpath(os.path.join(r, f), str(tmp_path)) for r, _, fs in os.walk(tmp_path) for f in fs)
        assert [f for f in files if f.startswith('models')] == [f for f in g['run/files'] if f.startswith('models')]
        ck = torch.load(tmp_path / 'models' / 'exp' / 'Net' / 'step_000006.chkpt', weights_only=False)
        assert sorted(ck.keys()) == list(g['run/ckpt_keys'])
        assert sorted(ck['model'].keys()) == list(g['run/ckpt_model_keys'])
        assert ck['step'] == int(g['run/ckpt_step']) and ck['seed'] == int(g['run/ckpt_seed'])
        mine = ['%s|%.9g|%s' % s for s in tr.writer.scalars]
        ref = list(g['run/writer_scalars'])
        assert [m.split('|')[0::2] for m in mine] == [r.split('|')[0::2] for r in ref]
        assert np.allclose([float(m.split('|')[1]) for m in mine], [float(r.split('|')[1]) for r in ref], rtol=1e-5)

        # resume on the same directory continues from step 6; ctor seed wins over the stored one
        net2 = Net()
        opt2 = torch.optim.SGD(net2.parameters(), lr=0.05, momentum=0.9)
        sch2 = torch.optim.lr_scheduler.StepLR(opt2, 2, 0.5)
        T.poison = False
        tr2 = T(net2, opt2, train, valid, max_step=8, valid_max_step=3, save_interval=3, log_interval=2,
                save_dir=str(tmp_path), save_prefix='exp', grad_clip=0.5, grad_norm=1.0, scheduler=sch2, seed=5)
        assert tr2.step == int(g['resume/step']) == 6 and tr2.seed == int(g['resume/seed'])
        tr2.run()
        for k, v in net2.state_dict().items():
            assert np.abs(v.numpy() - g['resume_final/' + k]).max() <= 1e-6, k
    finally:
        LOGGER.removeHandler(hdl)
    lines = [ln for ln in buf.getvalue().splitlines() if 'checkpoint' not in ln and 'No any checkpoint' not in ln]
    assert lines == list(g['run/log_lines'])
