"""python tools/r05/time_mel.py - psnd_mel_fwd on 1024 clips x 2 s (the bench's roofline_mel workload), sustained and best launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
dev = torch.device('cuda:0')
for _ in range(2):
    r = bench._mel_roofline(dev)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ('frac', 'launch_us', 'best_launch_us', 'burst8_launch_us', 'first_launch_us')})
