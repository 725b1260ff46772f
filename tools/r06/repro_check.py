"""two forward + backward passes of the separator (configs[1] model) and of hifi_gan_v1 on the same data: are loss and gradients the same bits?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_sound_amd.models import build_model
from pytorch_sound_amd.models import separator  # noqa: F401
from pytorch_sound_amd.models.vocoders import hifi_gan  # noqa: F401
dev = torch.device('cuda:0')


def check(name, net, args, reduce):
    runs = []
    for _ in range(3):
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = net(*args)
        loss = reduce(y)
        loss.backward()
        torch.cuda.synchronize()
        runs.append([loss.detach().clone()] + [p.grad.clone() for p in net.parameters() if p.grad is not None])
        for p in net.parameters():
            p.grad = None
    bad = [i for o in runs[1:] for i, (u, v) in enumerate(zip(runs[0], o)) if not torch.equal(u, v)]
    print(name, 'tensors', len(runs[0]), 'differing', sorted(set(bad))[:10], flush=True)


torch.manual_seed(0)
sep = build_model('conv_separator_voicebank').to(dev)
wav = torch.rand(8, 513, 173, device=dev)        # a magnitude spectrogram (N, K, F)
w = None


def red(y):
    y = y[0] if isinstance(y, (tuple, list)) else y
    return (y.float() ** 2).mean()


check('separator', sep, (wav,), red)
g = build_model('hifi_gan_v1').to(dev)
mel = torch.randn(4, 80, 32, device=dev)
check('hifi_gan_v1', g, (mel,), red)
