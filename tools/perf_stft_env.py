#!/usr/bin/env python
"""time psnd_stft_fwd (mag, n=1024/256, N clips of 2 s) under the env knobs given on the command line
(each argument is a comma-separated list of KEY=VALUE; '-' = defaults)."""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from pytorch_sound_amd import kernels as K
    from pytorch_sound_amd._lib import lib, ptr, stream_ptr, check
    n = int(os.environ.get('NFFT', '1024'))
    h, N, T = n // 4, int(os.environ.get('NCLIPS', '1024')), int(os.environ.get('TLEN', '44100'))
    dev = torch.device('cuda:0')
    m = np.arange(n); w = (0.5 - 0.5*np.cos(2*np.pi*m/n)).astype(np.float32)
    wav = torch.randn(N, T, device=dev) * 0.07
    plan = K.stft_plan(n, w).to(dev)
    F = K.frame_count(T, n, h); Kb = n//2+1
    mag = torch.empty(N, Kb, F, device=dev)
    def run():
        check(lib().psnd_stft_fwd(ptr(wav), N, T, n, h, 0, ptr(plan), 0.0, ptr(mag), None, None, None, stream_ptr(dev)), 'stft')
    for _ in range(3): run()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20 * 1e-3)
    b = 4*N*T + 4*N*Kb*F
    print('%-40s %.1f us  %.0f GB/s' % (sys.argv[2], best*1e6, b/best/1e9), flush=True)
else:
    for spec in sys.argv[1:]:
        env = dict(os.environ)
        if spec != '-':
            for kv in spec.split(','):
                k, v = kv.split('='); env[k] = v
        subprocess.run([sys.executable, __file__, '--child', spec], env=env)
