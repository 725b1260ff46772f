"""Magnitude-domain Conv1d source separator - the model BASELINE config 2 is quoted on.

The reference repository does not contain its "speech source-separation Conv1d model" (its README
only links an external project, SURVEY "fact 3"); this one is composed from the reference's own
building block - the HiFi-GAN ResBlock1 (k=3, dilations 1/3/5, leaky-relu 0.1, weight norm;
models/vocoders/hifi_gan.py:32-63) - and registered through the reference's registry:
    |X| (N,513,F) -> log1p -> Conv1d(513,C,3) -> B x ResBlock1(C) -> lrelu -> Conv1d(C,513,3) -> sigmoid
    -> mask * |X|.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from pytorch_sound_amd.models import register_model, register_model_architecture
from pytorch_sound_amd.models.vocoders.hifi_gan import WNConv1d, ResBlock1, LRELU_SLOPE


@register_model('conv_separator')
class ConvSeparator(nn.Module):
    def __init__(self, spec_size: int = 513, channels: int = 256, num_blocks: int = 4, kernel_size: int = 3,
                 dilation=(1, 3, 5)):
        super().__init__()
        self.conv_pre = WNConv1d(spec_size, channels, 3, 1, 1)
        self.blocks = nn.ModuleList([ResBlock1(None, channels, kernel_size, tuple(dilation)) for _ in range(num_blocks)])
        self.conv_post = WNConv1d(channels, spec_size, 3, 1, 1, init_std=0.01)

    def forward(self, mag: torch.Tensor) -> torch.Tensor:
        x = self.conv_pre(torch.log1p(mag))
        for block in self.blocks:
            x = block(x)
        mask = torch.sigmoid(self.conv_post(F.leaky_relu(x, LRELU_SLOPE)))
        return mask * mag


@register_model_architecture('conv_separator', 'conv_separator_voicebank')
def conv_separator_voicebank():
    return {'spec_size': 513, 'channels': 256, 'num_blocks': 4, 'kernel_size': 3, 'dilation': (1, 3, 5)}
