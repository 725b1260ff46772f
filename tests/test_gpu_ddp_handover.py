"""Gradient hand-over from INSIDE the conv body's backward (cl.GRAD_SINK): with a data-parallel reducer the
separator's body node produces its weight gradients block by block (weight-gradient launch + weight-norm backward behind every
input-gradient chain launch), writes them straight into the flat buckets and releases a bucket the moment it is complete - so the
all-reduce of the late blocks runs under the backward of the early ones.  Driven with a one-rank RCCL process group on the one GPU of
the box (PSND_DDP_FORCE=1: the sum over one rank is the identity), eager and as a captured step graph:
  * at least 4 of the 6 buckets leave before the backward has delivered all its gradients (emit_log);
  * the training equals the same steps without a reducer, and with the hand-over switched off (sink_enabled = False), to the fp32
    summation order of the weight-gradient slabs (the chunked launches split their rows differently), as far as Adam lets that show."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 5


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, tmp, q, mode, graph):
    try:
        sys.path.insert(0, ROOT)
        use_ddp = mode != 'none'
        os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK='0',
                          PSND_DDP_FORCE='1' if use_ddp else '0')
        os.environ.pop('PSND_DDP_GRAPH', None)
        import torch.distributed as dist
        from pytorch_sound_amd import kernels as K, optim as poptim
        from pytorch_sound_amd.models import build_model, separator  # noqa: F401
        from pytorch_sound_amd.models.transforms import LogMelSpectrogram
        from pytorch_sound_amd.trainer import Trainer, LogType
        torch.cuda.set_device(0)
        if use_ddp:
            dist.init_process_group('nccl', rank=0, world_size=1)
        dev = torch.device('cuda:0')
        fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0.0, 8000.0).to(dev)

        class T(Trainer):
            def forward(self, mag, ref, mel_ref, is_logging=False):
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    loss, _ = self.model.spectral_l1_loss(mag, ref, mel_ref, fe._mel_plan(), 80, 1.0, 0.5, 1e-6, fe.min_db, fe.max_db)
                return loss, {'loss': (loss, LogType.SCALAR)}

        torch.manual_seed(1234)
        net = build_model('conv_separator_voicebank').to(dev)
        g = torch.Generator().manual_seed(5)
        data = []
        for _ in range(STEPS):
            mag = (torch.rand(8, 513, 173, generator=g) * 4).to(dev)
            ref = (torch.rand(8, 513, 173, generator=g) * 4).to(dev)
            data.append((mag, ref, K.mel_forward(ref, fe._mel_plan(), 80, K.LOG_E, 1e-6, None, fe.min_db, fe.max_db)[0]))
        tr = T(net, poptim.Adam(net.parameters(), lr=2e-4, betas=(0.8, 0.99)), data, data[:1], max_step=STEPS, valid_max_step=1,
               save_interval=10 ** 6, log_interval=10 ** 6, save_dir=tmp, save_prefix=mode + str(int(graph)), seed=3)
        info = {}
        if use_ddp:
            assert tr._reducer is not None and tr._reducer.active
            tr._reducer.sink_enabled = mode == 'sink'
        tr.graph_steps, tr.graph_warmup = graph, 1
        net.train()
        for i in range(1, STEPS + 1):
            if use_ddp and i == STEPS and not graph:
                tr._reducer.handover_log.clear()               # the last (eager) step's hand-overs
            tr.step = i
            tr.train(i)
        torch.cuda.synchronize()
        if use_ddp:
            red = tr._reducer
            info = {'modes': [v.get('ddp') for v in getattr(tr, '_graphs', {}).values() if 'graph' in v],
                    'emit': list(getattr(red, 'emit_log', [])), 'handover': list(red.handover_log),
                    'nb': len(red.buckets), 'nparams': len(red.params), 'sizes': [len(b['params']) for b in red.buckets]}
            dist.destroy_process_group()
        q.put((mode, info, {k: v.float().cpu().numpy() for k, v in net.state_dict().items()}))
    except Exception as e:
        import traceback
        q.put((mode, repr(e) + traceback.format_exc()))
        raise


@pytest.mark.timeout(600)
@pytest.mark.parametrize('graph', [True, False])
def test_body_backward_hands_buckets_over_early(tmp_path, graph):
    ctx = mp.get_context('spawn')
    res = {}
    for mode in ('none', 'nosink', 'sink'):
        q = ctx.Queue()
        p = ctx.Process(target=_worker, args=(_port(), str(tmp_path), q, mode, graph))
        p.start()
        got = q.get(timeout=300)
        p.join(timeout=120)
        assert p.exitcode == 0 and len(got) == 3, got
        res[mode] = got
    info = res['sink'][1]
    if graph:
        assert info['modes'] == ['capture'], info
    nb, npar = info['nb'], info['nparams']
    assert nb == 6 and npar == 26 * 3, info
    # (bucket, number of the deliver() call - a chunk of the body node's backward - that released it; 0 = released by an autograd hook
    #  after the node had returned, or by finish())
    ho = info['handover'][-nb:]
    assert [i for i, _ in ho] == list(range(nb)), ho                   # fixed order
    early = [c for _, c in ho if c > 0]
    assert len(early) >= 4 and len(set(early)) >= 3, ho                # >= 4 of 6 buckets left from inside the body's backward, chunk by chunk
    assert ho[-1][1] >= max(early)                                     # the last bucket (block 1 + head) at the end
    # without the hand-over every bucket leaves after the node has returned (the round-3 behaviour this replaces)
    assert all(c == 0 for _, c in res['nosink'][1]['handover'][-nb:]), res['nosink'][1]['handover']
    if graph:
        emit = info['emit']
        need = list(np.cumsum(info['sizes']))
        assert all(a >= n for (_, a), n in zip(emit, need)), (emit, need)  # never before its own gradients
    worst_b = max(float(np.abs(res['nosink'][2][k] - res['none'][2][k]).max()) for k in res['none'][2])
    worst_c = max(float(np.abs(res['sink'][2][k] - res['none'][2][k]).max()) for k in res['none'][2])
    assert all(np.isfinite(v).all() for v in res['sink'][2].values())
    # Adam's first steps move every weight by ~lr whatever the gradient's size (g / sqrt(v)): a summation-order difference in a
    # near-zero gradient shows up as a fraction of a step - a fifth of the distance travelled is the bound (the gradients themselves
    # agree to 1e-5 relative: test_handover_gradients_equal_the_hooks_path)
    tol = 0.2 * 2e-4 * STEPS
    assert worst_b <= tol and worst_c <= tol, (worst_b, worst_c, tol)


def _grad_worker(port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK='0')
        import torch.distributed as dist
        from pytorch_sound_amd import kernels as K, distributed as pdist
        from pytorch_sound_amd.models import build_model, separator  # noqa: F401
        from pytorch_sound_amd.models.transforms import LogMelSpectrogram
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1)
        dev = torch.device('cuda:0')
        fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0.0, 8000.0).to(dev)
        torch.manual_seed(1234)
        net = build_model('conv_separator_voicebank').to(dev)
        g = torch.Generator().manual_seed(5)
        mag = (torch.rand(8, 513, 173, generator=g) * 4).to(dev)
        ref = (torch.rand(8, 513, 173, generator=g) * 4).to(dev)
        mel_ref = K.mel_forward(ref, fe._mel_plan(), 80, K.LOG_E, 1e-6, None, fe.min_db, fe.max_db)[0]
        red = pdist.FlatGradReducer(net, force=True)
        out, logs = {}, {}
        for mode in (False, True, True):
            red.sink_enabled = mode
            red.zero_grad()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                loss, _ = net.spectral_l1_loss(mag, ref, mel_ref, fe._mel_plan(), 80, 1.0, 0.5, 1e-6, fe.min_db, fe.max_db)
            loss.backward()
            red.finish()
            torch.cuda.synchronize()
            cur = {k: p.grad.clone() for k, p in net.named_parameters()}
            if mode in out:
                assert all(torch.equal(cur[k], out[mode][k]) for k in cur)         # deterministic
            out[mode], logs[mode] = cur, list(red.handover_log)[-len(red.buckets):]
        worst = max(float((out[True][k] - out[False][k]).abs().max() / out[False][k].abs().max().clamp_min(1e-30)) for k in out[True])
        dist.destroy_process_group()
        q.put(('ok', worst, logs))
    except Exception as e:
        import traceback
        q.put(('err', repr(e) + traceback.format_exc()))
        raise


@pytest.mark.timeout(300)
def test_handover_gradients_equal_the_hooks_path():
    """one eager backward with the hand-over on and off in the same process: every parameter gradient equal to 1e-5 of its largest
    entry (the chunked weight-gradient launches split their rows differently), repeatable bit for bit, the buckets released by
    successive chunks of the body node"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_grad_worker, args=(_port(), q))
    p.start()
    got = q.get(timeout=200)
    p.join(timeout=60)
    assert p.exitcode == 0 and got[0] == 'ok', got
    assert got[1] <= 1e-5, got[1]
    assert all(c == 0 for _, c in got[2][False])
    chunks = [c for _, c in got[2][True]]
    # cl.HANDOVER_CHUNKS = 3: bucket 0 behind the last block, buckets 1-3 behind the second, 4-5 at the end
    assert chunks == sorted(chunks) and len(set(chunks)) == 3 and chunks[0] > 0 and chunks[3] < chunks[4], chunks


def _shared_worker(port, q):
    """the body applied TWICE in one forward (two nodes produce gradients for every parameter): ADVICE r04 - the hand-over must stand back"""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK='0')
        import torch.distributed as dist
        from pytorch_sound_amd import kernels as K, distributed as pdist
        from pytorch_sound_amd.models import build_model, separator  # noqa: F401
        from pytorch_sound_amd.models.transforms import LogMelSpectrogram
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1)
        dev = torch.device('cuda:0')
        fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0.0, 8000.0).to(dev)
        torch.manual_seed(1234)
        net = build_model('conv_separator_voicebank').to(dev)
        g = torch.Generator().manual_seed(5)
        mags = [(torch.rand(4, 513, 173, generator=g) * 4).to(dev) for _ in range(2)]
        refs = [(torch.rand(4, 513, 173, generator=g) * 4).to(dev) for _ in range(2)]
        mels = [K.mel_forward(r, fe._mel_plan(), 80, K.LOG_E, 1e-6, None, fe.min_db, fe.max_db)[0] for r in refs]

        def losses():
            with torch.autocast('cuda', dtype=torch.bfloat16):
                return [net.spectral_l1_loss(m, r, l, fe._mel_plan(), 80, 1.0, 0.5, 1e-6, fe.min_db, fe.max_db)[0] for m, r, l in zip(mags, refs, mels)]

        # reference: the two uses in separate backward passes, no reducer, summed
        want = {k: torch.zeros_like(p) for k, p in net.named_parameters()}
        for i in range(2):
            net.zero_grad(set_to_none=True)
            losses()[i].backward()
            for k, p in net.named_parameters():
                want[k] += p.grad
        net.zero_grad(set_to_none=True)
        red = pdist.FlatGradReducer(net, force=True)
        out = {}
        for mode in (False, True):
            red.sink_enabled = mode
            red.zero_grad()
            a, b = losses()
            (a + b).backward()
            red.finish()
            torch.cuda.synchronize()
            out[mode] = {k: p.grad.clone() for k, p in net.named_parameters()}
            late = [c for _, c in list(red.handover_log)[-len(red.buckets):]]
            out[str(mode) + 'log'] = late
        worst = {m: max(float((out[m][k] - want[k]).abs().max() / want[k].abs().max().clamp_min(1e-30)) for k in want) for m in (False, True)}
        dist.destroy_process_group()
        q.put(('ok', worst, out['Truelog']))
    except Exception as e:
        import traceback
        q.put(('err', repr(e) + traceback.format_exc()))
        raise


@pytest.mark.timeout(300)
def test_shared_parameters_are_not_handed_over_early():
    """a model whose body runs twice in one forward: every parameter has two producing nodes, so no bucket may leave from inside a node's
    backward (chunk 0 = released by the post-accumulate hooks / finish()), and the gradients equal the sum of the two uses."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_shared_worker, args=(_port(), q))
    p.start()
    got = q.get(timeout=200)
    p.join(timeout=60)
    assert p.exitcode == 0 and got[0] == 'ok', got
    assert got[1][False] <= 2e-5 and got[1][True] <= 2e-5, got[1]
    assert all(c == 0 for c in got[2]), got[2]


def _bf16_wire_worker(port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK='0')
        import torch.distributed as dist
        from pytorch_sound_amd import kernels as K, distributed as pdist
        from pytorch_sound_amd.models import build_model, separator  # noqa: F401
        from pytorch_sound_amd.models.transforms import LogMelSpectrogram
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1)
        dev = torch.device('cuda:0')
        fe = LogMelSpectrogram(22050, 80, 1024, 1024, 256, -50, 30, 0.0, 8000.0).to(dev)
        torch.manual_seed(1234)
        net = build_model('conv_separator_voicebank').to(dev)
        g = torch.Generator().manual_seed(5)
        mag = (torch.rand(8, 513, 173, generator=g) * 4).to(dev)
        ref = (torch.rand(8, 513, 173, generator=g) * 4).to(dev)
        mel_ref = K.mel_forward(ref, fe._mel_plan(), 80, K.LOG_E, 1e-6, None, fe.min_db, fe.max_db)[0]
        out = {}
        for wire in (None, torch.bfloat16):
            red = pdist.FlatGradReducer(net, force=True, comm_dtype=wire)
            assert red.comm_dtype is wire
            for rep in range(2):
                red.zero_grad()
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    loss, _ = net.spectral_l1_loss(mag, ref, mel_ref, fe._mel_plan(), 80, 1.0, 0.5, 1e-6, fe.min_db, fe.max_db)
                loss.backward()
                red.finish()
                torch.cuda.synchronize()
            out[wire] = {k: p.grad.clone() for k, p in net.named_parameters()}
            red.remove()
            for p in net.parameters():
                p.grad = None
        # one rank: the sum over the ranks is the rank's own contribution, so the bf16 wire returns exactly bf16(fp32 gradient)
        bad = [k for k in out[None] if not torch.equal(out[torch.bfloat16][k], out[None][k].to(torch.bfloat16).float())]
        nz = sum(int((out[torch.bfloat16][k] != out[None][k]).sum()) for k in out[None])
        dist.destroy_process_group()
        q.put(('ok', bad, nz))
    except Exception as e:
        import traceback
        q.put(('err', repr(e) + traceback.format_exc()))
        raise


@pytest.mark.timeout(300)
def test_bf16_wire_returns_the_rounded_gradients():
    """FlatGradReducer(comm_dtype=bfloat16) over a one-rank RCCL group: psnd_grad_pack_bf16 -> all-reduce -> psnd_grad_unpack_bf16 on the
    release stream; every gradient the optimizer sees is the bf16 rounding (nearest even) of the fp32-wire gradient, bit for bit"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_bf16_wire_worker, args=(_port(), q))
    p.start()
    got = q.get(timeout=200)
    p.join(timeout=60)
    assert p.exitcode == 0 and got[0] == 'ok', got
    assert got[1] == [], got[1][:5]
    assert got[2] > 1000            # (and it did round something)
