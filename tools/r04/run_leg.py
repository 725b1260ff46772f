"""one of bench.py's bounded single-GPU legs alone (a target for rocprofv3 --kernel-trace --stats): python tools/r04/run_leg.py config3|config4"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
leg = {'config3': bench._config3_leg, 'config4': bench._config4_leg}[sys.argv[1]]
print(json.dumps(leg(dev)))
