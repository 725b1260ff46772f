#!/usr/bin/env python
"""a few config-2 training steps exactly as bench.py sets them up (eager warm-up, one capture, some replays) and nothing else -
rocprofv3 --pmc target covering every kernel of the step:  tools/run_step_once.py [replays]"""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pytorch_sound_amd import optim as poptim
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
Trainer, model = bench.build_step(dev, amp=True, static=True, fused_loss=True)
opt = poptim.Adam(model.parameters(), lr=2e-4, betas=(0.8, 0.99))
pool = [bench.synth_batch(1234 + 1000 * i, bench.BATCH_PER_GPU, int(bench.SR * bench.CLIP_SECONDS), dev) for i in range(2)]
tr = Trainer(model, opt, pool, pool, max_step=10 ** 9, valid_max_step=1, save_interval=10 ** 9, log_interval=10 ** 9,
             save_dir=tempfile.mkdtemp(prefix='psnd_step_'), save_prefix='step', seed=1234)
tr.graph_steps = True
model.train()
for s in range(1, tr.graph_warmup + 2 + (int(sys.argv[1]) if len(sys.argv) > 1 else 4)):
    tr.step = s
    tr.train(s)
torch.cuda.synchronize()
print('done')
