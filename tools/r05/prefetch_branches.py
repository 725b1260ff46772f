"""config 3 with the batches in pinned HOST memory: Trainer.prefetch_copy (the copy of the next batch on a side stream) next to the
graph branches of the generator (PSND_PREFETCH_BRANCHES=1 keeps them on) against branches off (the default next to a prefetch stream)
and against the device-resident pool.  python tools/r05/prefetch_branches.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = os.environ.get('MODE', 'host')
cfg = int(os.environ.get('CONFIG', '3'))
built = bench._config3_build(dev) if cfg == 3 else bench._config4_build(dev)
tr = built[0]
if mode == 'host':
    src = tr.train_dataset
    dev_pool = src if isinstance(src, (list, tuple)) else [next(src) for _ in range(4)]
    pool = [tuple(t.detach().cpu().pin_memory() if torch.is_tensor(t) else t for t in b) for b in dev_pool]
    tr.train_dataset = tr.repeat(pool)
    tr.prefetch_copy = True
ms, dist = bench._time_steps(tr, steps, 10)
from pytorch_sound_amd import cl
print('config', cfg, 'mode', mode, 'PSND_PREFETCH_BRANCHES', os.environ.get('PSND_PREFETCH_BRANCHES', '0'), 'AUTO_SECTIONS', cl.AUTO_SECTIONS,
      'ms/step %.3f' % ms, 'p50 %.3f p99 %.3f max %.3f' % (dist['p50'], dist['p99'], dist['max']))
