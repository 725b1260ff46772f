"""where do the slow steps of the EAGER drop-in leg come from (BENCH_r05 dropin_step.eager: p50 3.0 ms, max 124 ms)?  Logs every cyclic-GC pass with
its duration next to the per-step host times of bench._dropin_leg.  argv: steps [freeze]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
freeze = len(sys.argv) > 2 and sys.argv[2] == 'freeze'
log = []
t_start = {}


def cb(phase, info):
    if phase == 'start':
        t_start[info['generation']] = time.perf_counter()
    else:
        log.append((info['generation'], (time.perf_counter() - t_start[info['generation']]) * 1e3, info['collected']))


gc.callbacks.append(cb)
if freeze:
    from pytorch_sound_amd import trainer
    trainer.Trainer.gc_freeze = True
else:
    from pytorch_sound_amd import trainer
    trainer.Trainer.gc_freeze = False
dev = torch.device('cuda:0')
res = bench._dropin_leg(dev, steps=steps)
for m in ('eager', 'graph_steps'):
    print(m, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in res[m]['step_ms'].items() if k != 'note'}, 'ms/step %.3f' % res[m]['ms_per_step'])
g2 = [x for x in log if x[0] == 2]
print('gc passes: gen0 %d gen1 %d gen2 %d; gen2 durations ms: %s; longest gen0/1: %.2f ms' % (
    sum(1 for x in log if x[0] == 0), sum(1 for x in log if x[0] == 1), len(g2), ' '.join('%.1f' % x[1] for x in g2),
    max([x[1] for x in log if x[0] < 2] or [0])))
print('objects tracked:', len(gc.get_objects()), 'frozen:', gc.get_freeze_count())
