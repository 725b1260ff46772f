"""Trainer.graph_steps on the GPU: a run whose steps are replayed from a hipGraph must follow the eager run
(same seeds, same batches) - parameters after 12 steps equal to fp32 reassociation noise - and must skip a NaN step
on the device exactly like the eager path."""
import numpy as np
import tempfile

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _run(graph, steps=12, poison_step=None, static=False):
    from pytorch_sound_amd.trainer import Trainer, LogType
    from pytorch_sound_amd.models import build_model
    from pytorch_sound_amd.models import separator  # noqa: F401
    from pytorch_sound_amd.models.transforms import STFT
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    model = build_model('conv_separator_voicebank').to(dev)
    stft = STFT(1024, 256).to(dev)
    g = torch.Generator().manual_seed(3)
    pool = [(torch.randn(4, 8192, generator=g) * 0.1, torch.randn(4, 8192, generator=g) * 0.1) for _ in range(steps)]
    if poison_step is not None:                     # batch i is consumed by step i + 1
        bad = pool[poison_step - 1][0].clone()
        bad[0, 100] = float('nan')
        pool[poison_step - 1] = (bad, pool[poison_step - 1][1])

    from pytorch_sound_amd import kernels as K
    bufs = {}

    class T(Trainer):
        static_prepare = static

        def prepare(self, noisy, clean):
            with torch.no_grad():
                if not static:
                    return stft.magnitude(noisy), stft.magnitude(clean)
                outs = []
                for name, w in (('a', noisy), ('b', clean)):     # features written into persistent buffers (kernels.stft_forward out_mag=)
                    if name not in bufs:
                        bufs[name] = torch.empty((w.shape[0], 513, K.frame_count(w.shape[1], 1024, 256)), device=w.device)
                    outs.append(K.stft_forward(w, 1024, 256, stft._plan(w.device), out_mag=bufs[name])['mag'])
                return tuple(outs)

        def forward(self, mag_mix, mag_ref, is_logging=False):
            with torch.autocast('cuda', dtype=torch.bfloat16):
                est = self.model(mag_mix)
            loss = F.l1_loss(est.float(), mag_ref)
            return loss, {'loss': (loss, LogType.SCALAR)}

    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    tr = T(model, opt, pool, pool, max_step=10 ** 9, valid_max_step=1, save_interval=10 ** 9, log_interval=10 ** 9,
           save_dir=tempfile.mkdtemp(prefix='psnd_graph_'), seed=11)
    tr.graph_steps = graph
    model.train()
    for i in range(1, steps + 1):
        tr.step = i
        tr.train(i)
    torch.cuda.synchronize()
    tr._poll_nan_log(block=True)
    captured = len(getattr(tr, '_graphs', {}))
    return [p.detach().float().clone() for p in model.parameters()], captured, opt


def test_graph_steps_follow_eager_steps():
    eager, n0, _ = _run(False)
    graph, n1, _ = _run(True)
    assert n0 == 0 and n1 == 1
    for a, b in zip(eager, graph):
        scale = float(a.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-2 * scale     # bf16 model, Adam: a few ulp of bf16 after 12 steps


def test_static_prepare_reads_the_features_in_place():
    """Trainer.static_prepare: prepare() writes into persistent buffers (out_mag=) and returns them every step; the captured graph takes
    them as its inputs where they are - bit-identical parameters to the run that copies fresh tensors into graph-owned inputs."""
    a, n0, _ = _run(True)
    b, n1, _ = _run(True, static=True)
    assert n0 == 1 and n1 == 1
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_graph_step_skips_nan_on_device():
    params, n, opt = _run(True, steps=8, poison_step=7)        # step 7 is a replayed step with a NaN input
    assert n == 1
    assert all(bool(torch.isfinite(p).all()) for p in params)
    steps = {int(s['step'].item()) for s in opt.state.values() if 'step' in s}
    assert steps == {7}                                        # 8 steps, one skipped by the optimizer kernel itself


def test_prefetch_prepare_gives_the_same_steps():
    """Trainer.prefetch_prepare stages batch k+1 (copy + prepare()) on a side stream during step k: same losses, same
    parameters as the in-line order, in eager and in graph mode."""
    import tempfile
    from pytorch_sound_amd.trainer import Trainer, LogType
    dev = torch.device('cuda:0')

    def run(prefetch, graph):
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Conv1d(1, 8, 5, padding=2), torch.nn.Tanh(), torch.nn.Conv1d(8, 1, 5, padding=2)).to(dev)
        seen = []

        class T(Trainer):
            def prepare(self, x, y):
                return x * 2.0, y + 1.0                      # parameter-free preprocessing

            def forward(self, x, y, is_logging=False):
                loss = torch.nn.functional.mse_loss(self.model(x.unsqueeze(1)).squeeze(1), y)
                seen.append(loss.detach())
                return loss, {'loss': (loss, LogType.SCALAR)}

        g = torch.Generator().manual_seed(1)
        data = [(torch.randn(4, 256, generator=g), torch.randn(4, 256, generator=g)) for _ in range(8)]
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True)
        tr = T(model, opt, data, data, max_step=8, valid_max_step=1, save_interval=100, log_interval=100,
               save_dir=tempfile.mkdtemp(prefix='psnd_pf_'), seed=3)
        tr.prefetch_prepare, tr.graph_steps = prefetch, graph
        model.train()
        for i in range(1, 9):
            tr.step = i
            tr.train(i)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in model.parameters()], [float(v) for v in seen]

    for graph in (False, True):
        pa, la = run(False, graph)
        pb, lb = run(True, graph)
        assert la == lb or np.allclose(la, lb, rtol=1e-6), (la, lb)
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_graph_cache_is_bounded_and_recaptures():
    """variable-length batches: one captured graph per input signature, least recently used ones dropped beyond
    Trainer.graph_cache_size; a dropped shape is captured again when it comes back; same parameters as eager steps."""
    from pytorch_sound_amd.trainer import Trainer, LogType
    dev = torch.device('cuda:0')

    def run(graph):
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Conv1d(1, 4, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv1d(4, 1, 3, padding=1)).to(dev)

        class T(Trainer):
            def forward(self, x, y, is_logging=False):
                loss = torch.nn.functional.mse_loss(self.model(x.unsqueeze(1)).squeeze(1), y)
                return loss, {'loss': (loss, LogType.SCALAR)}

        g = torch.Generator().manual_seed(1)
        lens = [64, 96, 128]
        data = [(torch.randn(2, lens[i % 3], generator=g), torch.randn(2, lens[i % 3], generator=g)) for i in range(30)]
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True)
        tr = T(model, opt, data, data, max_step=30, valid_max_step=1, save_interval=100, log_interval=100,
               save_dir=tempfile.mkdtemp(prefix='psnd_gc_'), seed=3)
        tr.graph_steps, tr.graph_warmup, tr.graph_cache_size = graph, 1, 2
        model.train()
        for i in range(1, 31):
            tr.step = i
            tr.train(i)
        torch.cuda.synchronize()
        return tr, [p.detach().clone() for p in model.parameters()]

    tr, pg = run(True)
    captured = [k for k, v in tr._graphs.items() if 'graph' in v]
    assert 1 <= len(captured) <= 2 and len(tr._graphs) <= 3
    _, pe = run(False)
    for a, b in zip(pg, pe):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
