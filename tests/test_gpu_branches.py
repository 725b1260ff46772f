"""Round 4: independent parts of a step on parallel streams (graph branches).  The branches must not change a single bit: the same
launches on the same data, only their order in time differs.
  * the resblocks of a HiFi-GAN stage on cl.branch_streams (Generator.cl_branches),
  * the parameter-side backward of the transposed convs / the 1x1 projections on cl.param_stream (cl.BRANCH_PARAM_GRADS),
  * psnd_convtr1d_cl_bwd with either role alone, psnd_mha_bwd_parts against psnd_mha_bwd."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def _gen(seed=5):
    from pytorch_sound_amd.models.vocoders.hifi_gan import Generator
    torch.manual_seed(seed)
    h = Namespace(resblock='1', upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=64,
                  resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    g = Generator(h).cuda()
    g.precision = 'bf16'       # the channels-last kernels (with their graph branches) whatever the context: also inside spawned workers,
    #                            which do not see conftest's fixture (an fp32 input outside autocast would take the fp32 convolutions)
    with torch.no_grad():
        for n, p in g.named_parameters():
            if n.endswith('weight_v'):
                p.mul_(10.0 if p.abs().max() < 0.1 else 1.0)
    return g


def _run(g, x, w):
    g.zero_grad(set_to_none=True)
    xc = x.clone().requires_grad_(True)
    out = g(xc)
    (out * w).sum().backward()
    torch.cuda.synchronize()
    return out.detach().clone(), xc.grad.clone(), {n: p.grad.clone() for n, p in g.named_parameters()}


def test_generator_branches_are_bit_identical_to_one_stream(monkeypatch):
    from pytorch_sound_amd import cl
    g = _gen()
    x = torch.randn(3, 80, 24, device='cuda')
    w = torch.randn(3, 1, 24 * 8, device='cuda')
    monkeypatch.setattr(g, 'cl_branches', False, raising=False)
    monkeypatch.setattr(cl, 'BRANCH_PARAM_GRADS', False)
    o0, gx0, gp0 = _run(g, x, w)
    monkeypatch.setattr(g, 'cl_branches', True, raising=False)
    monkeypatch.setattr(cl, 'BRANCH_PARAM_GRADS', True)
    for _ in range(3):                                     # several passes: a missing join shows up as a changing result
        o1, gx1, gp1 = _run(g, x, w)
        assert torch.equal(o0, o1) and torch.equal(gx0, gx1)
        for n in gp0:
            assert torch.equal(gp0[n], gp1[n]), n


def test_generator_branches_inside_a_captured_step(monkeypatch):
    """forward + backward captured as one hipGraph with the branches on: replays reproduce the eager one-stream gradients"""
    from pytorch_sound_amd import cl
    g = _gen(7)
    x = torch.randn(2, 80, 16, device='cuda')
    w = torch.randn(2, 1, 16 * 8, device='cuda')
    monkeypatch.setattr(g, 'cl_branches', False, raising=False)
    monkeypatch.setattr(cl, 'BRANCH_PARAM_GRADS', False)
    o0, _, gp0 = _run(g, x, w)
    monkeypatch.setattr(g, 'cl_branches', True, raising=False)
    monkeypatch.setattr(cl, 'BRANCH_PARAM_GRADS', True)
    _run(g, x, w)                                          # warm-up: stream pools, pack caches
    g.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = g(x)
        (out * w).sum().backward()
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, o0)
        for n, p in g.named_parameters():
            assert torch.equal(p.grad, gp0[n]), n


def test_linear1x1_parameter_branch_is_bit_identical(monkeypatch):
    from pytorch_sound_amd import cl, kernels as K
    torch.manual_seed(0)
    lin = torch.nn.Conv1d(96, 160, 1).cuda()
    x = torch.randn(4, 96, 300, device='cuda')
    res = []
    for flag in (False, True, True):
        monkeypatch.setattr(cl, 'BRANCH_PARAM_GRADS', flag)
        lin.zero_grad(set_to_none=True)
        xc = x.clone().requires_grad_(True)
        y = K.Linear1x1.apply(xc, lin.weight, lin.bias, True, False)
        (y * y).sum().backward()
        torch.cuda.synchronize()
        res.append((y.detach().clone(), xc.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()))
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert torch.equal(a, b)
    # with a .grad already there autograd ACCUMULATES with a launch on the caller's stream: the branch must not be taken
    monkeypatch.setattr(cl, 'BRANCH_PARAM_GRADS', True)
    xc = x.clone().requires_grad_(True)
    y = K.Linear1x1.apply(xc, lin.weight, lin.bias, True, False)
    (y * y).sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(lin.weight.grad, 2 * res[0][2], rtol=1e-6, atol=0) and torch.allclose(lin.bias.grad, 2 * res[0][3], rtol=1e-6, atol=0)


def test_mha_bwd_parts_equal_the_whole(monkeypatch):
    from pytorch_sound_amd._lib import lib, check, ptr, stream_ptr
    torch.manual_seed(1)
    N, H, C, T = 3, 4, 128, 200
    dev = torch.device('cuda', 0)
    kvq = torch.randn(N, 3 * C, T, device=dev)
    out = torch.empty(N, C, T, device=dev)
    stats = torch.empty(H * N, T, 2, device=dev)
    gout = torch.randn(N, C, T, device=dev)
    for bf16 in (0, 1):
        check(lib().psnd_mha_fwd(ptr(kvq), None, N, H, C, T, ptr(out), None, ptr(stats), bf16, stream_ptr(dev)), 'fwd')
        d0, g0 = torch.empty(H * N, T, device=dev), torch.full_like(kvq, float('nan'))
        check(lib().psnd_mha_bwd(ptr(kvq), None, ptr(out), None, ptr(stats), ptr(gout), None, N, H, C, T, ptr(d0), ptr(g0), bf16, stream_ptr(dev)), 'bwd')
        d1, g1 = torch.empty(H * N, T, device=dev), torch.full_like(kvq, float('nan'))
        for part in (1, 4, 2):
            check(lib().psnd_mha_bwd_parts(ptr(kvq), None, ptr(out), None, ptr(stats), ptr(gout), None, N, H, C, T, ptr(d1), ptr(g1), bf16, part,
                                           stream_ptr(dev)), 'bwd parts')
        torch.cuda.synchronize()
        assert torch.equal(d0, d1) and torch.equal(g0, g1) and not torch.isnan(g1).any()
    assert lib().psnd_mha_bwd_parts(ptr(kvq), None, ptr(out), None, ptr(stats), ptr(gout), None, N, H, C, T, ptr(d1), ptr(g1), 0, 8, stream_ptr(dev)) != 0


def test_from_cl_tanh_and_its_backward():
    """psnd_from_cl_tanh / psnd_to_cl_tanh_bwd against tanh(from_cl) / to_cl(g * (1 - y^2)) formed by torch on the same buffers"""
    from pytorch_sound_amd import cl
    torch.manual_seed(3)
    N, C, T, HP = 5, 1, 301, 3
    shape = cl.CLShape(N, T, HP)
    buf = torch.zeros(N, shape.Lp, 32, dtype=torch.bfloat16, device='cuda')
    buf[:, HP:HP + T, :C] = (2.0 * torch.randn(N, T, C, device='cuda')).to(torch.bfloat16)
    b1 = buf.clone().requires_grad_(True)
    y = cl.FromCLTanh.apply(b1, C, T, shape)
    ref = torch.tanh(cl.from_cl_raw(buf, C, T, shape))
    assert float((y.detach() - ref).abs().max()) <= 2e-7                  # the same tanhf, up to the compiler's contraction
    g = torch.randn_like(y)
    y.backward(g)
    gref = torch.zeros_like(buf, dtype=torch.float32)
    gref[:, HP:HP + T, :C] = (g * (1 - y.detach() * y.detach())).transpose(1, 2)
    assert torch.equal(b1.grad, gref.to(torch.bfloat16))
    # psnd_to_cl keeps refusing pre-operations it does not know (2 is the internal tanh-backward mode and needs its operand)
    from pytorch_sound_amd._lib import lib, ptr, stream_ptr
    x = torch.randn(N, C, T, device='cuda')
    out = torch.empty_like(buf)
    assert lib().psnd_to_cl(ptr(x), N, C, T, shape.Lp, HP, 32, 2, ptr(out), stream_ptr(x.device)) != 0


def _reducer_worker(port, q):
    """a Trainer on a one-rank RCCL group (PSND_DDP_FORCE): captured steps of a small HiFi-GAN generator with the graph branches next to the
    gradient reducer (round 5) against the same steps with the branches switched off - gradients after one eager + one replayed step and
    parameters after 4 steps, bit for bit; no parameter zero-filled at a release point; every parameter arrived exactly once"""
    import faulthandler
    import os
    import sys
    import tempfile
    faulthandler.dump_traceback_later(350, exit=True)        # a hang in here must not outlive the test's timeout
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK='0', PSND_DDP_FORCE='1')
        import torch.distributed as dist
        from pytorch_sound_amd import cl, kernels as K, optim as poptim
        from pytorch_sound_amd.trainer import Trainer, LogType
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1)
        out = {}
        for branches in (False, True):
            os.environ['PSND_DDP_BRANCHES'] = '1' if branches else '0'
            for nstep in (2, 4):
                g = _gen(11)
                g.cl_branches = branches
                cl.BRANCH_PARAM_GRADS = branches

                class Step(Trainer):
                    def forward(self, x, y, is_logging=False):
                        loss = K.l1_loss(self.model(x), y)
                        return loss, {'loss': (loss, LogType.SCALAR)}

                gen = torch.Generator().manual_seed(3)
                data = [(torch.randn(2, 80, 16, generator=gen).cuda(), torch.randn(2, 1, 128, generator=gen).cuda()) for _ in range(4)]
                tr = Step(g, poptim.Adam(g.parameters(), lr=1e-3), data, data[:1], max_step=4, valid_max_step=1, save_interval=10 ** 6,
                          log_interval=10 ** 6, save_dir=tempfile.mkdtemp(prefix='psnd_br_'), save_prefix='b%d%d' % (branches, nstep), seed=1)
                assert tr._reducer is not None and tr._reducer.active
                tr.graph_steps, tr.graph_warmup = True, 1
                g.train()
                for i in range(1, nstep + 1):
                    tr.step = i
                    tr.train(i)
                torch.cuda.synchronize()
                modes = [v.get('ddp') for v in getattr(tr, '_graphs', {}).values() if 'graph' in v]
                red = tr._reducer
                out[(branches, nstep)] = ({k: p.grad.detach().float().cpu().numpy().copy() for k, p in g.named_parameters()} if nstep == 2 else
                                          {k: v.detach().float().cpu().numpy() for k, v in g.state_dict().items()},
                                          modes, bool(cl.AUTO_SECTIONS), list(red.zeroed_log), list(red.emit_log), [len(b['params']) for b in red.buckets])
                red.remove()
        dist.destroy_process_group()
        q.put(('ok', out))
    except Exception as e:                  # noqa: BLE001
        import traceback
        q.put(('err', repr(e) + traceback.format_exc()))
        raise


@pytest.mark.timeout(700)
def test_branches_next_to_a_gradient_reducer_are_bit_identical():
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_reducer_worker, args=(port, q))
    p.start()
    got = q.get(timeout=400)
    p.join(timeout=60)
    assert p.exitcode == 0 and got[0] == 'ok', got
    for nstep in (2, 4):
        (w0, m0, s0, z0, e0, n0), (w1, m1, s1, z1, e1, n1) = got[1][(False, nstep)], got[1][(True, nstep)]
        assert m0 == ['capture'] and m1 == ['capture'], (m0, m1)       # the all-reduce captured into the step graph both times
        assert s1 is True and s0 is False                               # the branches really were on next to the reducer / off
        assert z0 == [] and z1 == [], (z0[:3], z1[:3])                  # no used parameter was zero-filled at a release point ...
        need = list(np.cumsum(n1))
        assert all(a == n for (_, a), n in zip(e1, need)), (e1, need)   # ... every bucket left with exactly its parameters arrived, once each
        for k in w0:
            assert np.array_equal(w0[k], w1[k]), (nstep, k)


def test_branches_next_to_a_prefetch_copy_stream_train_the_same(tmp_path):
    """Round 5: graph branches stay on next to Trainer.prefetch_copy (the next batch's host -> device copy on a side stream).  A small
    HiFi-GAN generator trained for 6 captured steps from batches in PINNED HOST memory (prefetch_copy, branches on: cl.AUTO_SECTIONS stays
    True) ends at the same parameters, bit for bit, as the same steps from a device-resident pool - and as the prefetch run with the
    branches switched off (PSND_PREFETCH_BRANCHES=0)."""
    import os
    from pytorch_sound_amd import cl, kernels as K, optim as poptim
    from pytorch_sound_amd.trainer import Trainer, LogType
    dev = torch.device('cuda:0')
    g0 = torch.Generator().manual_seed(11)
    host = [(torch.randn(2, 80, 16, generator=g0), torch.randn(2, 1, 16 * 8, generator=g0)) for _ in range(6)]

    def run(where, env):
        old = os.environ.get('PSND_PREFETCH_BRANCHES')
        if env is None:
            os.environ.pop('PSND_PREFETCH_BRANCHES', None)
        else:
            os.environ['PSND_PREFETCH_BRANCHES'] = env
        try:
            gen = _gen(9)

            class T(Trainer):
                def forward(self, m, w, is_logging=False):
                    loss = K.l1_loss(self.model(m), w)
                    return loss, {'loss': (loss, LogType.SCALAR)}

            data = [tuple(t.pin_memory() for t in b) for b in host] if where == 'host' else [tuple(t.to(dev) for t in b) for b in host]
            tr = T(gen, poptim.Adam(gen.parameters(), lr=1e-3), data, data, max_step=6, valid_max_step=1, save_interval=100,
                   log_interval=100, save_dir=str(tmp_path / (where + str(env))), seed=3)
            tr.graph_steps = True
            tr.prefetch_copy = where == 'host'
            gen.train()
            for i in range(1, 7):
                tr.step = i
                tr.train(i)
            torch.cuda.synchronize()
            return {n: p.detach().clone() for n, p in gen.named_parameters()}, cl.AUTO_SECTIONS
        finally:
            if old is None:
                os.environ.pop('PSND_PREFETCH_BRANCHES', None)
            else:
                os.environ['PSND_PREFETCH_BRANCHES'] = old

    pd, sd = run('dev', None)
    ph, sh = run('host', None)
    p0, s0 = run('host', '0')
    assert sd is True and sh is True and s0 is False, (sd, sh, s0)
    for n in pd:
        assert torch.equal(pd[n], ph[n]), n
        assert torch.equal(pd[n], p0[n]), n
