"""which switch makes the reducer run differ: resblock branches / parameter-side branches / AUTO_SECTIONS alone"""
import faulthandler, os, sys, tempfile
faulthandler.dump_traceback_later(100, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29871', LOCAL_RANK='0', PSND_DDP_FORCE='1')
import numpy as np, torch, torch.distributed as dist
import test_gpu_branches as tb
from pytorch_sound_amd import cl, kernels as K, optim as poptim
from pytorch_sound_amd.trainer import Trainer, LogType
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
res = {}
for name, (ddpb, resb, parb) in {'off': (0, 0, 0), 'sections_only': (1, 0, 0), 'res': (1, 1, 0), 'par': (1, 0, 1), 'all': (1, 1, 1), 'off2': (0, 0, 0)}.items():
    os.environ['PSND_DDP_BRANCHES'] = str(ddpb)
    g = tb._gen(11); g.cl_branches = bool(resb); cl.BRANCH_PARAM_GRADS = bool(parb)
    class Step(Trainer):
        def forward(self, x, y, is_logging=False):
            loss = K.l1_loss(self.model(x), y)
            return loss, {'loss': (loss, LogType.SCALAR)}
    gen = torch.Generator().manual_seed(3)
    data = [(torch.randn(2, 80, 16, generator=gen).cuda(), torch.randn(2, 1, 128, generator=gen).cuda()) for _ in range(4)]
    tr = Step(g, poptim.Adam(g.parameters(), lr=1e-3), data, data[:1], max_step=4, valid_max_step=1, save_interval=10 ** 6,
              log_interval=10 ** 6, save_dir=tempfile.mkdtemp(prefix='psnd_br_'), save_prefix=name, seed=1)
    tr.graph_steps, tr.graph_warmup = True, 1
    g.train()
    for i in range(1, int(os.environ.get('NSTEP', '3')) + 1):
        tr.step = i; tr.train(i)
    torch.cuda.synchronize()
    res[name] = {k: p.grad.detach().float().cpu().numpy().copy() for k, p in g.named_parameters()}
    red = tr._reducer
    print(name, 'emit_log', red.emit_log, 'sizes', [len(b['params']) for b in red.buckets], 'zeroed', getattr(red, 'zeroed_log', None), flush=True)
    tr._reducer.remove()
for name in res:
    bad = {k: float(np.abs(res[name][k] - res['off'][k]).max() / max(np.abs(res['off'][k]).max(), 1e-30)) for k in res['off'] if not np.array_equal(res[name][k], res['off'][k])}
    bad = dict(sorted(bad.items(), key=lambda kv: -kv[1]))
    print(name, 'differs in', len(bad), 'tensors', [(k, round(v, 4), float(np.abs(res[name][k]).max()), float(np.abs(res['off'][k]).max())) for k, v in list(bad.items())[:3]])
dist.destroy_process_group()
