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
from pytorch_sound_amd.models.vocoders.hifi_gan import WNConv1d, ResBlock1, LRELU_SLOPE, wants_bf16


@register_model('conv_separator')
class ConvSeparator(nn.Module):
    def __init__(self, spec_size: int = 513, channels: int = 256, num_blocks: int = 4, kernel_size: int = 3,
                 dilation=(1, 3, 5)):
        super().__init__()
        self.conv_pre = WNConv1d(spec_size, channels, 3, 1, 1)
        self.blocks = nn.ModuleList([ResBlock1(None, channels, kernel_size, tuple(dilation)) for _ in range(num_blocks)])
        self.conv_post = WNConv1d(channels, spec_size, 3, 1, 1, init_std=0.01)

    # arithmetic of a HIP input, as for the HiFi-GAN generator (hifi_gan.wants_bf16): 'auto' = the channels-last bf16 kernels under
    # torch.autocast / for a bf16 input, fp32 convolutions (exact-fp32 matrix-core GEMMs, kernels.conv1d_f32) for an fp32 input outside it
    precision = 'auto'

    def forward(self, mag: torch.Tensor) -> torch.Tensor:
        from pytorch_sound_amd import deferred
        if mag.is_cuda and wants_bf16(self.precision, mag):
            node = deferred.nfk_of(mag) if deferred.ENABLED else None
            if node is not None:
                # STFT.transform's lazy bin-fastest magnitude: (N, F, K) is the channels-last order itself - the way in is a plain stream,
                # and the estimate stays deferred on that layout (the fused loss then runs bin-fastest end to end)
                y, shape = self.logits_cl(node.nfk, 'nfk')
                return deferred.est(y, None, shape, node)
            return self.forward_cl(deferred.resolve(mag))
        mag = deferred.resolve(mag)
        if mag.is_cuda:
            with torch.autocast('cuda', enabled=False):
                return self._forward_plain(mag.float()).to(mag.dtype)
        return self._forward_plain(mag)

    def _forward_plain(self, mag):
        x = self.conv_pre(torch.log1p(mag))
        for block in self.blocks:
            x = block(x)
        mask = torch.sigmoid(self.conv_post(x, LRELU_SLOPE))
        return mask * mag


    def logits_cl(self, mag: torch.Tensor, layout: str = 'nkf'):
        """the mask logits in the CL layout (bf16) and that layout's geometry: everything of forward_cl but the head.  layout 'nfk': `mag`
        is bin-fastest, (N, F, K) (kernels.stft_mag_nfk) - the channels-last order itself, so the way in is a plain stream"""
        from pytorch_sound_amd import cl
        if layout == 'nfk':
            N, T, C = mag.shape
        else:
            N, C, T = mag.shape
        halo = max(c.padding for b in self.blocks for c in list(b.convs1) + list(b.convs2))
        shape = cl.CLShape(N, T, max(halo, self.conv_pre.padding, self.conv_post.padding))
        convs = [self.conv_pre] + [c for b in self.blocks for c in list(b.convs1) + list(b.convs2)] + [self.conv_post]
        prep = cl.prep_all(self, convs, defer_backward_packs=True)                 # backward packs: written when the backward starts
        x0 = cl.to_cl_nfk(mag.float(), shape, 1) if layout == 'nfk' else cl.ToCL.apply(mag.float(), shape, 1)
        return cl.conv_body_cl(self.conv_pre, list(self.blocks), self.conv_post, x0, shape, prep), shape

    def spectral_l1_loss(self, mag: torch.Tensor, mag_ref: torch.Tensor, mel_ref: torch.Tensor, mel_plan: torch.Tensor, n_mels: int,
                         w_mag: float = 1.0, w_mel: float = 0.5, log_offset: float = 1e-6, clamp_lo=None, clamp_hi=None, layout: str = 'nkf'):
        """w_mag * l1(est, mag_ref) + w_mel * l1(log_mel(est), mel_ref) for est = forward(mag), as one fused node on a HIP tensor
        (cl.MaskHeadSpectralL1CL: the log-mel of the estimate and the two gradient tensors never exist); returns (loss, est).
        mel_plan: kernels.mel_plan of the mel filter; the log-mel is ln(mel + log_offset) clamped to [clamp_lo, clamp_hi] as
        LogMelSpectrogram does.  layout 'nfk': mag / mag_ref (and the returned est) are bin-fastest (N, F, K) tensors (kernels.stft_mag_nfk;
        `.transpose(1, 2)` is the reference's layout as a view); mel_ref stays (N, M, F)."""
        from pytorch_sound_amd import cl, kernels as K
        if not mag.is_cuda:
            raise RuntimeError('spectral_l1_loss is the fused HIP formulation; on the host compose the loss from forward()')
        if layout == 'nfk':
            y, shape = self.logits_cl(mag, 'nfk')
            loss, est = cl.MaskHeadSpectralL1NFK.apply(y, mag.float().contiguous(), mag_ref.float().contiguous(), mel_ref.float().contiguous(),
                                                       mel_plan, shape, n_mels, K.LOG_E, log_offset, None, clamp_lo, clamp_hi, w_mag, w_mel)
            loss.psnd_nan_flag = cl.LAST_LOSS_NAN_FLAG[0]          # isnan(loss), written by the launch that formed the loss
            return loss, est
        y, shape = self.logits_cl(mag)
        loss, est = cl.MaskHeadSpectralL1CL.apply(y, mag.float().contiguous(), mag_ref.float().contiguous(), mel_ref.float().contiguous(),
                                                  mel_plan, shape, n_mels, K.LOG_E, log_offset, None, clamp_lo, clamp_hi, w_mag, w_mel)
        loss.psnd_nan_flag = cl.LAST_LOSS_NAN_FLAG[0]
        return loss, est

    def forward_cl(self, mag: torch.Tensor) -> torch.Tensor:
        """the same function on the gfx950 implicit-GEMM conv kernel (channels-last bf16, fp32 accumulate):
        one launch per conv forward, leaky-relu / bias / residual fused, log1p fused into the layout change."""
        from pytorch_sound_amd import cl
        N, C, T = mag.shape
        halo = max(c.padding for b in self.blocks for c in list(b.convs1) + list(b.convs2))
        shape = cl.CLShape(N, T, max(halo, self.conv_pre.padding, self.conv_post.padding))
        convs = [self.conv_pre] + [c for b in self.blocks for c in list(b.convs1) + list(b.convs2)] + [self.conv_post]
        prep = cl.prep_all(self, convs, defer_backward_packs=True)                 # all weight-norm packs: one launch (+ one when the backward starts)
        x0 = cl.ToCL.apply(mag.float(), shape, 1)                                  # log1p(mag), CL bf16
        y = cl.conv_body_cl(self.conv_pre, list(self.blocks), self.conv_post, x0, shape, prep)     # 26 convs: one autograd node
        if mag.dtype == torch.float32 and not mag.requires_grad:
            from pytorch_sound_amd import deferred
            if deferred.ENABLED:
                # the estimate as a deferred tensor: torch ops of the reference's loss recipe on it (matmul with a LogMelSpectrogram's
                # mel_filter, log, clamp, F.l1_loss) are recorded and resolve to the fused loss node; any other use forms it first
                return deferred.est(y, mag.contiguous(), shape)
            return cl.MaskHeadCL.apply(y, mag, shape)                               # sigmoid(from_cl(y)) * mag, one pass
        logits = cl.FromCL.apply(y, C, T, shape)
        return torch.sigmoid(logits) * mag


@register_model_architecture('conv_separator', 'conv_separator_voicebank')
def conv_separator_voicebank():
    return {'spec_size': 513, 'channels': 256, 'num_blocks': 4, 'kernel_size': 3, 'dilation': (1, 3, 5)}
