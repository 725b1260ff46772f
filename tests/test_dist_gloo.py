"""world_size-2 gloo run of the data-parallel path on CPU: the flat-bucket reducer averages gradients
exactly like one process on the concatenated batch, ranks stay bit-identical, rank 0 alone saves, and
a NaN on ONE rank makes BOTH ranks skip the step (no deadlock)."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
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


def _worker(rank, world, port, tmp, q, deferred, comm=None):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    torch.cuda.is_available = lambda: False
    import torch.distributed as dist
    from pytorch_sound_amd import distributed as pdist
    from pytorch_sound_amd.trainer import Trainer, LogType
    assert pdist.init_from_env('gloo')

    class T(Trainer):
        def forward(self, x, y, is_logging=False):
            loss = torch.nn.functional.mse_loss(self.model(x), y)
            if self.step == 2 and pdist.rank() == 1 and self.model.training:
                loss = loss * float('nan')                        # only rank 1 sees a NaN
            return loss, {'loss': (loss.item(), LogType.SCALAR)}

    net = _net()
    if rank == 1:                                                 # broadcast must repair this
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    data = _batches(6)
    mine = [(x[rank * 2:rank * 2 + 2], y[rank * 2:rank * 2 + 2]) for x, y in data]     # shard each batch
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    if comm == 'bf16':
        T.ddp_comm_dtype = torch.bfloat16
    tr = T(net, opt, mine, mine[:2], max_step=4, valid_max_step=2, save_interval=2, log_interval=1,
           save_dir=tmp, save_prefix='dp', seed=3)
    assert tr._reducer.comm_dtype is (torch.bfloat16 if comm == 'bf16' else None)
    tr._reducer.deferred = deferred        # graph mode: no collective from the backward hooks, one reduction in finish()
    tr.run()
    q.put((rank, {k: v.numpy() for k, v in net.state_dict().items()}, float(tr.best_valid_loss)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('deferred', [False, True])
def test_two_rank_gloo_training(tmp_path, deferred):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q, deferred)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, sd, best = q.get(timeout=240)
        res[r] = (sd, best)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # ranks identical
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), k
    assert res[0][1] == res[1][1]
    # == one process on the full batches, skipping step 2 (the collectively skipped NaN step)
    net = _net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    data = _batches(6)
    for step in range(1, 5):
        x, y = data[(step - 1) % 6]
        if step == 2:
            continue
        opt.zero_grad()
        # mean over ranks of per-rank means == mean over the full batch (equal shard sizes)
        torch.nn.functional.mse_loss(net(x), y).backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert np.abs(v.numpy() - res[0][0][k]).max() <= 2e-6, k
    # rank 0 alone wrote checkpoints
    ck = sorted(os.listdir(tmp_path / 'models' / 'dp' / 'Sequential'))
    assert ck == ['step_000002.chkpt', 'step_000004.chkpt']


@pytest.mark.timeout(300)
@pytest.mark.parametrize('deferred', [False, True])
def test_two_rank_gloo_training_bf16_wire(tmp_path, deferred):
    """FlatGradReducer(comm_dtype=bfloat16): a bf16 image of every bucket crosses the wire.  The ranks stay BIT-identical (same collective,
    same result everywhere), the NaN step is still skipped by both, and the weights follow the fp32 run to bf16 rounding of the gradients."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q, deferred, 'bf16')) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, sd, best = q.get(timeout=240)
        res[r] = (sd, best)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in res[0][0]:
        assert np.array_equal(res[0][0][k], res[1][0][k]), k
    assert res[0][1] == res[1][1]
    net = _net()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    data = _batches(6)
    for step in range(1, 5):
        x, y = data[(step - 1) % 6]
        if step == 2:
            continue
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(x), y).backward()
        opt.step()
    worst = 0.0
    for k, v in net.state_dict().items():
        worst = max(worst, float(np.abs(v.numpy() - res[0][0][k]).max()))
    # three SGD steps of lr 0.1 on gradients of magnitude <= ~1, each rounded to 8 mantissa bits twice (per rank, and the sum)
    assert 0.0 < worst <= 3 * 0.1 * 2.0 ** -7, worst


def test_flat_reducer_load_grads_and_alignment():
    """graph mode hands the reducer gradients that live outside the buckets: they must land in the (256-byte aligned)
    bucket views, unused parameters must read as zero, and .grad must point into the bucket afterwards"""
    from pytorch_sound_amd.distributed import FlatGradReducer
    net = _net()
    red = FlatGradReducer(net, bucket_bytes=1 << 10)          # several buckets
    params = [p for p in net.parameters()]
    grads = {p: torch.full_like(p, float(i + 1)) for i, p in enumerate(params)}
    grads[params[1]] = None                                   # an unused parameter
    for b in red.buckets:
        b['flat'].fill_(7.0)                                  # stale content must not survive for the unused one
    red.load_grads(grads)
    for i, p in enumerate(params):
        assert p.grad.data_ptr() % 256 == red._bucket_of[p]['flat'].data_ptr() % 256
        lo, hi = red._bucket_of[p]['flat'].data_ptr(), red._bucket_of[p]['flat'].data_ptr() + red._bucket_of[p]['flat'].numel() * 4
        assert lo <= p.grad.data_ptr() < hi
        want = 0.0 if i == 1 else float(i + 1)
        assert bool((p.grad == want).all()), i
    red.finish()                                              # world 1: a no-op


def test_auto_bucket_size_gives_the_backward_something_to_overlap():
    """no bucket size given: about six buckets per model (whole MiB, 1..32 MiB) - a single 32 MiB bucket would hold all 22 MB of the
    config-2 separator's gradients, whose all-reduce could then only start after the last gradient of the backward"""
    from pytorch_sound_amd.distributed import FlatGradReducer
    from pytorch_sound_amd.models import build_model
    from pytorch_sound_amd.models import separator  # noqa: F401
    import pytorch_sound_amd.models.vocoders.hifi_gan  # noqa: F401
    mib = 1 << 20
    assert FlatGradReducer.auto_bucket_bytes(22 * mib) == 4 * mib
    assert FlatGradReducer.auto_bucket_bytes(56 * mib) == 10 * mib
    assert FlatGradReducer.auto_bucket_bytes(100) == 4 * mib and FlatGradReducer.auto_bucket_bytes(1 << 40) == 32 * mib     # floor 4 MiB (round 6: a bucket is a release point of the step graph)
    assert FlatGradReducer.auto_bucket_bytes(int(4.2 * mib)) == 4 * mib                                                      # the config-4 block: one or two buckets, not five
    sep = build_model('conv_separator_voicebank')
    red = FlatGradReducer(sep)
    total = sum(p.numel() * 4 for p in sep.parameters())
    assert red.bucket_bytes == FlatGradReducer.auto_bucket_bytes(total)
    assert 5 <= len(red.buckets) <= 8, len(red.buckets)
    # reverse registration order: the first bucket holds conv_post (the first gradients of the backward), the last conv_pre
    first = {id(p) for p in red.buckets[0]['params']}
    assert id(sep.conv_post.bias) in first and id(sep.conv_pre.weight_v) in {id(p) for p in red.buckets[-1]['params']}
    red.remove()


def _flag_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    torch.cuda.is_available = lambda: False
    from pytorch_sound_amd import distributed as pdist
    assert pdist.init_from_env('gloo')
    net = _net()
    red = pdist.FlatGradReducer(net)
    out = []
    for step, raise_on in enumerate((None, 1, 0)):                       # nobody, rank 1, rank 0 raises its flag
        red.zero_grad()
        red.set_flag(torch.tensor(1.0 if raise_on == rank else 0.0))
        for p in net.parameters():
            p.grad.copy_(torch.full_like(p, float(rank + 1)))
        for p in reversed(list(net.parameters())):                        # what the backward hooks do
            red._on_grad(p)
        red.finish(average=(step == 0))
        g = next(net.parameters()).grad
        out.append((float(red.flag), float(g.flatten()[0])))
    q.put((rank, out))


def test_nan_flag_rides_with_the_gradients_and_sum_mode():
    """FlatGradReducer.set_flag / .flag: the per-rank NaN flag is reduced inside the last gradient bucket (no collective of its
    own), flag > 0 on EVERY rank iff any rank raised it; finish(average=False) leaves the SUM (the optimizer kernel divides)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_flag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        (f0, g0), (f1, g1), (f2, g2) = res[rank]
        assert f0 == 0.0 and f1 > 0.0 and f2 > 0.0
        assert g0 == 1.5 and g1 == 3.0 and g2 == 3.0                     # mean of (1, 2), then sums


def _order_worker(rank, world, port, tmp, q):
    """(i) Trainer(seed=None) under DDP - the seed is drawn on rank 0 and broadcast as an exact int64;
    (ii) a parameter that gets no gradient on ONE rank only: the bucket collectives must still be issued in the same
    (fixed) order on both ranks"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    torch.cuda.is_available = lambda: False
    import torch.distributed as dist
    from pytorch_sound_amd import distributed as pdist
    from pytorch_sound_amd.trainer import Trainer, LogType
    assert pdist.init_from_env('gloo')

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(5)
            self.a = torch.nn.Linear(8, 8)
            self.b = torch.nn.Linear(8, 8)          # skipped by rank 1
            self.c = torch.nn.Linear(8, 1)

        def forward(self, x, use_b):
            h = torch.tanh(self.a(x))
            if use_b:
                h = h + self.b(h)
            return self.c(h)

    class T(Trainer):
        def forward(self, x, y, is_logging=False):
            loss = torch.nn.functional.mse_loss(self.model(x, pdist.rank() == 0), y)
            return loss, {'loss': (loss.item(), LogType.SCALAR)}

    net = Net()
    g = torch.Generator().manual_seed(100 + rank)
    data = [(torch.randn(4, 8, generator=g), torch.randn(4, 1, generator=g)) for _ in range(3)]
    np.random.seed(1000 + rank)                         # different host RNG state per rank: the drawn seeds differ
    tr = T(net, torch.optim.SGD(net.parameters(), lr=0.05), data, data[:1], max_step=3, valid_max_step=1, save_interval=10,
           log_interval=10, save_dir=tmp, save_prefix='ord', seed=None)
    tr._reducer.remove()
    tr._reducer = pdist.FlatGradReducer(net, bucket_bytes=128)      # one bucket per leaf module (a module's weight and bias stay together)
    tr.run()
    q.put((rank, int(tr.seed), list(tr._reducer.launch_log), len(tr._reducer.buckets),
           {k: v.numpy() for k, v in net.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_seed_broadcast_and_fixed_collective_order(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, seed, order, nb, sd = q.get(timeout=240)
        res[r] = (seed, order, nb, sd)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0] and res[0][0] > 0                  # exact, identical seed
    nb = res[0][2]
    assert nb >= 3
    assert res[0][1] == res[1][1] == list(range(nb)) * 3             # every step: buckets 0..nb-1 in order on both ranks
    for k in res[0][3]:
        assert np.array_equal(res[0][3][k], res[1][3][k]), k
