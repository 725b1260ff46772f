"""host time of one config-2 step (enqueue only) against its GPU time: is the step loop host-bound?
python tools/r04/host_time.py [steps]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
Trainer, model = bench.build_step(dev, amp=True)
from pytorch_sound_amd import optim as poptim
opt = poptim.Adam(model.parameters(), lr=2e-4, betas=(0.8, 0.99))
T = int(bench.SR * bench.CLIP_SECONDS)
pool = [bench.synth_batch(1234 + 1000 * i, 32, T, dev) for i in range(8)]
tr = Trainer(model, opt, pool, pool, max_step=10 ** 9, valid_max_step=1, save_interval=10 ** 9, log_interval=10 ** 9,
             save_dir=tempfile.mkdtemp(prefix='psnd_ht_'), save_prefix='b', seed=1234)
tr.graph_steps = True
model.train()
s = 0
for _ in range(60):
    s += 1; tr.step = s; tr.train(s)
torch.cuda.synchronize()
# (a) enqueue time per step while the GPU queue is deep (no sync inside)
host = []
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    s += 1; tr.step = s; tr.train(s)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('steps %d: host enqueue %.1f us/step (median %.1f, p90 %.1f), wall incl. drain %.1f us/step' % (
    steps, t_enq / steps * 1e6, np.median(host) * 1e6, np.percentile(host, 90) * 1e6, t_all / steps * 1e6), flush=True)
# (b) parts of the host time
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    s += 1; tr.step = s; tr.train(s)
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats('cumulative').print_stats(28)
print(st.getvalue()[:6000])
