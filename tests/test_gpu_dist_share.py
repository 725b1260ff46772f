"""The device-side data-parallel step (hipGraph replay, flat-bucket gradient reduction with the NaN flag riding along, division by
the world size inside the optimizer kernel, collective NaN skip) with TWO ranks sharing the one GPU of the test box over gloo
(PSND_DIST_SHARE_GPU=1; RCCL refuses two ranks on one device): ranks end bit-identical and equal to one process on the full batches."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Conv1d(4, 8, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv1d(8, 2, 1))


def _batches(n):
    g = torch.Generator().manual_seed(11)
    return [(torch.randn(4, 4, 16, generator=g), torch.randn(4, 2, 16, generator=g)) for _ in range(n)]


NAN_STEP, STEPS = 5, 9


def _worker(rank, world, port, tmp, q, graph, hip_adam):
    try:
        _run(rank, world, port, tmp, q, graph, hip_adam)
    except Exception as e:                                                # the parent must not wait for its timeout
        q.put((rank, repr(e)))
        raise


def _run(rank, world, port, tmp, q, graph, hip_adam):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank),
                      PSND_DIST_SHARE_GPU='1')
    import torch.distributed as dist
    from pytorch_sound_amd import distributed as pdist, optim as poptim
    from pytorch_sound_amd.trainer import Trainer, LogType
    assert pdist.init_from_env('nccl')                                    # share mode switches to gloo itself
    dev = torch.device('cuda:0')

    class T(Trainer):
        def forward(self, x, y, is_logging=False):
            loss = torch.nn.functional.mse_loss(self.model(x), y)
            return loss, {'loss': (loss, LogType.SCALAR)}

    net = _net().to(dev)
    data = _batches(STEPS)
    mine = [(x[rank * 2:rank * 2 + 2].clone(), y[rank * 2:rank * 2 + 2].clone()) for x, y in data]
    if rank == 1:
        mine[NAN_STEP - 1][0][0, 0, 0] = float('nan')                     # only rank 1 sees a NaN, through its data (replays included)
    opt = poptim.Adam(net.parameters(), lr=1e-2) if hip_adam else torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)
    tr = T(net, opt, mine, mine[:1], max_step=STEPS, valid_max_step=1, save_interval=10 ** 6, log_interval=10 ** 6,
           save_dir=tmp, save_prefix='dp', seed=3)
    tr.graph_steps, tr.graph_warmup = graph, 1
    net.train()
    for i in range(1, STEPS + 1):
        tr.step = i
        tr.train(i)
    torch.cuda.synchronize()
    q.put((rank, {k: v.cpu().numpy() for k, v in net.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize('graph,hip_adam', [(True, True), (False, True), (True, False)])
def test_two_ranks_on_one_gpu(tmp_path, graph, hip_adam):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q, graph, hip_adam)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in procs)
    assert all(isinstance(v, dict) for v in res.values()), res
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k                    # ranks bit-identical
    # one process, full batches, the NaN step skipped by everybody
    net = _net().double()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    data = _batches(STEPS)
    for step in range(1, STEPS + 1):
        if step == NAN_STEP:
            continue
        x, y = data[step - 1]
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(x.double()), y.double()).backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert np.abs(v.numpy() - res[0][k]).max() <= 2e-5 * max(1.0, np.abs(v.numpy()).max()), k
