#!/usr/bin/env python
"""optimizer step on the config-2 separator's parameter set (5.5 M fp32 parameters in 78 tensors): psnd_adam_step against
torch.optim.Adam(fused=True); 28 bytes per parameter."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_sound_amd import optim as O  # noqa: E402
from pytorch_sound_amd.models import build_model  # noqa: E402
import pytorch_sound_amd.models.separator  # noqa: E402,F401

dev = torch.device('cuda:0')
for name, mk in (('separator (config 2)', lambda: build_model('conv_separator_voicebank')), ):
    try:
        model = mk().to(dev)
    except Exception as e:  # arch name differs: fall back to a same-sized parameter list
        print('build_model failed (%s): synthetic 26 x (256,256,3) + biases' % e)
        model = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(256, 256, 3)) for _ in range(26)]
                                       + [torch.nn.Parameter(torch.randn(256)) for _ in range(52)]).to(dev)
    params = [p for p in model.parameters() if p.requires_grad]
    n = sum(p.numel() for p in params)
    for p in params:
        p.grad = torch.randn_like(p)
    for oname, opt in (('psnd_adam_step', O.Adam(params, lr=2e-4, betas=(0.8, 0.99))),
                       ('torch fused Adam', torch.optim.Adam(params, lr=2e-4, betas=(0.8, 0.99), fused=True))):
        for _ in range(5):
            opt.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            opt.step()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print('%-22s %-18s %d tensors, %.2f M parameters: %.1f us/step = %.2f TB/s' % (name, oname, len(params), n / 1e6, us, 28 * n / us / 1e6))
