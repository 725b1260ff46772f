"""weight prep of the config-2 separator (26 convs, 5.5 M weights) in one launch: time per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_sound_amd import cl
from pytorch_sound_amd.models import build_model, separator  # noqa
dev = torch.device('cuda:0')
m = build_model('conv_separator_voicebank').to(dev)
convs = [c for c in m.modules() if hasattr(c, 'weight_v')]
print(len(convs), 'convs')
for _ in range(5): cl.prep_all(m, convs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): cl.prep_all(m, convs)
e1.record(); torch.cuda.synchronize()
print('prep_all: %.1f us per launch' % (e0.elapsed_time(e1) / 50 * 1e3))
