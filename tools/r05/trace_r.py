"""s_memtime phase table of stft_fwd_n4096r_kernel (build with -DPSND_R_TRACE: tools/r04/variant_q.sh rtrace psnd_stft_r -DPSND_R_TRACE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
dev = torch.device('cuda:0')
trace = torch.zeros((256, 16, 16), dtype=torch.int64, device=dev)
os.environ['PSND_R_TRACE_PTR'] = str(trace.data_ptr())
os.environ['PSND_R_TRACE_ITER'] = sys.argv[1] if len(sys.argv) > 1 else '4'
from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import periodic_window
plan = K.stft_plan(4096, periodic_window('hann', 4096).astype(np.float32)).to(dev)
x = 0.07 * torch.randn(32, 1323000, device=dev)
o = torch.empty((32, K.frame_count(1323000, 4096, 1024), 2049), device=dev)
for _ in range(3): K.stft_mag_nfk(x, 4096, 1024, plan, out=o)
torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.float64)
names = ['poll 4 chunks', 'samples + window 0 arrive', 'window, radix-2', 'swap', 'radix-32 #1', 'twiddle', 'publish (vmcnt)', 'lock', 'transpose', 'radix-32 #2',
         'split + magnitudes', 'slot wait + request', 'stores issued']
lt = t[:, 15, :].copy()
t = t[:, :15, :]
d = t[:, :, 1:14] - t[:, :, 0:13]
ok = (t[:, :, 13] > 0)
print('frames traced:', int(ok.sum()), ' frame (stamp 0 -> 13) mean %.0f cycles' % (t[:, :, 13] - t[:, :, 0])[ok].mean())
for i, n in enumerate(names):
    v = d[:, :, i][ok]
    print('%-28s mean %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f' % (n, v.mean(), np.median(v), np.percentile(v, 90), v.max()))


okl = lt[:, 4] > 0
for i, n in enumerate(['loader: slot poll', 'loader: request', 'loader: vmcnt wait', 'loader: publish']):
    v = (lt[:, i + 1] - lt[:, i])[okl]
    print('%-28s mean %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f' % (n, v.mean(), np.median(v), np.percentile(v, 90), v.max()))

full = trace.cpu().numpy().astype(np.float64)
t0 = full[:, :, 14]; t1 = full[:, :15, 15]
print('kernel entry spread over the chip: %.0f cycles; per workgroup entry -> last frame wave exit: mean %.0f  min %.0f max %.0f cycles' % (
    t0.max() - t0.min(), (t1.max(axis=1) - t0.min(axis=1)).mean(), (t1.max(axis=1) - t0.min(axis=1)).min(), (t1.max(axis=1) - t0.min(axis=1)).max()))
print('whole launch (first entry -> last exit): %.0f cycles' % (t1.max() - t0.min()))
print('frame waves: exit spread inside a workgroup (max - min): mean %.0f' % (t1.max(axis=1) - t1.min(axis=1)).mean())
it = int(os.environ['PSND_R_TRACE_ITER'])
fs = full[:, :15, 0]
print('start of traced frame (number %d of each wave) after workgroup entry: mean %.0f' % (it, (fs - t0.min(axis=1)[:, None])[fs > 0].mean()))
