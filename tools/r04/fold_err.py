"""prints the relative L2 error of the folded HiFi-GAN generators on the CL kernels vs the fp32 host path (tolerance calibration)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_sound_amd.models import build_model
import pytorch_sound_amd.models.vocoders.hifi_gan  # noqa: F401
for arch in ('hifi_gan_v1', 'hifi_gan_v2', 'hifi_gan_v3'):
    for seed in (5, 6, 7):
        torch.manual_seed(seed)
        gen = build_model(arch)
        mel = torch.randn(2, 80, 24)
        gen.eval()
        with torch.no_grad():
            want = gen(mel)
        gen.remove_weight_norm()
        gen = gen.to('cuda:0')
        with torch.no_grad():
            got = gen(mel.to('cuda:0')).cpu()
        print(arch, seed, float((got - want).norm() / want.norm()), float((got - want).abs().max()), float(want.abs().max()))
