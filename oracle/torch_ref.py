"""Oracle / CPU baseline: the reference's feature path restated with the SAME torch calls the
reference makes, so that it runs on the host cores of the GPU box (where /root/reference does not
exist).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by bench.py's `cpu_baseline`
leg ("kind": "port") and by tests as a second checker.  Equality with the imported reference is
established by tests/test_oracle_golden.py::test_torch_ref_* on the golden fixtures.

RefSTFT.transform          <- pytorch_sound/models/transforms.py:53-69 (reflect pad + conv1d with the
                              dense windowed-DFT basis + sqrt / atan2)
RefLogMel.forward          <- transforms.py:231-244 (matmul, log(.+1e-6), truthiness-gated clamps)
RefSTFTTorch.transform     <- transforms.py:297-311 (STFTTorchAudio: torch.stft(center, reflect, onesided) -> re, im -> magnitude,
                              phase), the library-FFT variant BASELINE.md asks to be timed separately next to the dense-DFT port
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import features as ofe


class RefSTFT(torch.nn.Module):
    def __init__(self, filter_length=1024, hop_length=512, win_length=None):
        super().__init__()
        self.filter_length, self.hop_length = filter_length, hop_length
        self.pad_amount = filter_length // 2
        basis = ofe.forward_basis_ref32(filter_length, win_length)            # (2K, n) float32
        self.register_buffer('forward_basis', torch.from_numpy(basis[:, None, :].copy()))

    def transform(self, wav):
        x = F.pad(wav.unsqueeze(1).unsqueeze(1), (self.pad_amount, self.pad_amount, 0, 0), mode='reflect').squeeze(1)
        y = F.conv1d(x, self.forward_basis, stride=self.hop_length, padding=0)
        re, im = y.chunk(2, 1)
        return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im.data, re.data)


class RefLogMel(torch.nn.Module):
    def __init__(self, sample_rate, mel_size, n_fft, win_length, hop_length, min_db=None, max_db=None,
                 mel_min=0., mel_max=None):
        super().__init__()
        self.stft = RefSTFT(win_length, hop_length)
        self.register_buffer('mel_filter', torch.from_numpy(ofe.mel_filterbank(sample_rate, n_fft, mel_size, mel_min, mel_max)))
        self.min_db = np.log(np.power(10, min_db / 10)) if min_db else None
        self.max_db = np.log(np.power(10, max_db / 10)) if max_db else None

    def mel_of_mag(self, mag, log_offset=1e-6):
        mel = torch.log(torch.matmul(self.mel_filter, mag) + log_offset)
        if self.min_db:
            mel = mel.clamp_min(self.min_db)
        if self.max_db:
            mel = mel.clamp_max(self.max_db)
        return mel

    def forward(self, wav, log_offset=1e-6):
        return self.mel_of_mag(self.stft.transform(wav)[0], log_offset)


class RefSTFTTorch(torch.nn.Module):
    """STFTTorchAudio.forward / transform (transforms.py:297-311) on this torch: `torch.stft(..., return_complex=True)` viewed
    as (re, im) is what the reference's torch-1.7 call returned."""

    def __init__(self, filter_length=1024, hop_length=512, win_length=None):
        super().__init__()
        self.n_fft, self.hop_length = filter_length, hop_length
        self.win_length = win_length if win_length else filter_length
        self.register_buffer('window', torch.hann_window(self.win_length))

    def transform(self, wav):
        s = torch.stft(wav, self.n_fft, self.hop_length, self.win_length, self.window, True, 'reflect', False, True,
                       return_complex=True)
        re, im = s.real, s.imag
        return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)


class RefLogMelTorch(RefLogMel):
    """RefLogMel with the torch.stft front end"""

    def __init__(self, sample_rate, mel_size, n_fft, win_length, hop_length, min_db=None, max_db=None, mel_min=0., mel_max=None):
        super().__init__(sample_rate, mel_size, n_fft, win_length, hop_length, min_db, max_db, mel_min, mel_max)
        self.stft = RefSTFTTorch(win_length, hop_length)
