"""Drop-in for pytorch_sound/models/sound.py: VolNormConv, PreEmphasis, InversePreEmphasis, build_stft_functions,
multi_stft_loss - same names, arguments and results.

On a HIP device PreEmphasis and multi_stft_loss run on libpsnd_hip.so (psnd_preemphasis_*, psnd_stft_fwd/bwd,
psnd_stft_loss_*); CPU tensors take the reference's torch formulation (host-side use: tests, data preparation).
"""
from typing import List, Tuple

import torch
import torch.nn.functional as F

from pytorch_sound_amd import kernels as K
from pytorch_sound_amd.models.transforms import STFTTorchAudio as STFT


class VolNormConv:
    """Windowed volume normalisation (sound.py:7-60): every hop-sized slice is divided by the standard deviation of
    the window starting there, scaled to ``target_db``; ``reverse`` undoes it with the remembered deviations.
    Host-side utility (python loop over hops on ``.data``, exactly the reference's slicing rules)."""

    def __init__(self, window_size: int, hop_size: int, target_db: float):
        self.window_size = window_size
        self.hop_size = hop_size
        self.target_db = target_db
        self.prev_wav_len = -1
        self.std_buffer = None

    def init_buffer(self, wav_len: int):
        self.prev_wav_len = wav_len
        self.std_buffer = torch.zeros((wav_len - self.window_size) // self.hop_size + 1)

    def _scale(self, std):
        return std / 10 ** (self.target_db / 10)

    def forward(self, wav: torch.Tensor) -> torch.Tensor:
        wav_len = wav.size(-1)
        self.init_buffer(wav_len)
        last = wav_len - self.window_size
        chunks = []
        for idx, start in enumerate(range(0, last, self.hop_size)):
            stop = start + self.hop_size if start < last - 1 else None        # the final slice runs to the end
            std = torch.std(wav.data[..., start:start + self.window_size])
            self.std_buffer[idx] = std
            chunks.append(wav.data[..., start:stop] / self._scale(std))
        return torch.cat(chunks, dim=-1)

    def reverse(self, wav: torch.Tensor) -> torch.Tensor:
        wav_len = wav.size(-1)
        assert self.prev_wav_len >= wav_len, '{} is smaller than {} !'.format(self.prev_wav_len, wav_len)
        last = wav_len - self.window_size
        chunks = []
        for idx, start in enumerate(range(0, last, self.hop_size)):
            stop = start + self.hop_size if start < last - self.hop_size else None
            chunks.append(wav.data[..., start:stop] * self._scale(self.std_buffer[idx]))
        return torch.cat(chunks, dim=-1)


class PreEmphasis(torch.nn.Module):
    """y[t] = x[t] - coef * x[t-1] on (N, 1, T), one reflect-padded sample on the left (sound.py:66-81).  The
    ``flipped_filter`` buffer is kept for state_dict compatibility."""

    def __init__(self, coef: float = 0.97):
        super().__init__()
        self.coef = coef
        self.register_buffer('flipped_filter', torch.FloatTensor([-self.coef, 1.]).unsqueeze(0).unsqueeze(0))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        assert len(input.size()) == 3, 'The number of dimensions of input tensor must be 3!'
        if input.is_cuda:                                   # a HIP tensor always takes psnd_preemphasis_* (fp32 kernel: cast in and out)
            if input.size(1) != 1 or not input.is_floating_point():
                raise RuntimeError('PreEmphasis expects a floating-point (N, 1, T) tensor, got %s %s' % (input.dtype, tuple(input.shape)))
            y = K.PreEmphasisFn.apply(input.float(), self.coef)
            return y if y.dtype == input.dtype else y.to(input.dtype)
        input = F.pad(input, (1, 0), 'reflect')
        return F.conv1d(input, self.flipped_filter)


class InversePreEmphasis(torch.nn.Module):
    """sound.py:84-99 verbatim in behaviour: a 1-unit ``torch.nn.RNN`` (default tanh non-linearity, as in the
    reference) with input weight 1 and recurrent weight ``coef`` - a sequential scan, inference-side only."""

    def __init__(self, coef: float = 0.97):
        super().__init__()
        self.coef = coef
        self.rnn = torch.nn.RNN(1, 1, 1, bias=False, batch_first=True)
        self.rnn.weight_ih_l0.data.fill_(1)
        self.rnn.weight_hh_l0.data.fill_(self.coef)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        x, _ = self.rnn(input.transpose(1, 2))
        return x.transpose(1, 2)


#
# Multi-resolution STFT loss
#
_STFT_CACHE = {}


def build_stft_functions(*params: Tuple[int, int, int]):
    """STFT modules for tuples (n_fft, window size, hop size) (sound.py:89-101: ``STFT(win, hop, win, fft)``).  The
    reference rebuilds them (and moves them to the GPU) on every loss call; here they are built once per tuple."""
    out = []
    for fft, win, hop in params:
        key = (int(fft), int(win), int(hop))
        if key not in _STFT_CACHE:
            if not _STFT_CACHE:
                print('Build Mel Functions ...')
            _STFT_CACHE[key] = STFT(win, hop, win, fft)
        out.append(_STFT_CACHE[key])
    return out


def multi_stft_loss(pred: torch.Tensor, target: torch.Tensor, stft_params: List[Tuple[int, int, int]], eps: float = 1e-5
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Multi-resolution STFT loss (sound.py:106-133).
    :param pred: predicted waveforms (N, T)
    :param target: target waveforms (N, T)
    :param stft_params: list of tuples (n_fft, window size, hop size)
    :param eps: added inside the logs
    :return: (loss = mean over resolutions of sc + mag, spectral-convergence loss, log-magnitude loss)
    """
    funcs = build_stft_functions(*stft_params)
    if pred.is_cuda:
        cfgs = tuple((f.n_fft, f.hop_length) for f in funcs)
        plans = [f._plan(pred.device) for f in funcs]
        out = K.MultiStftLossFn.apply(pred.float(), target.float(), eps, cfgs, *plans)
        return out[0], out[1], out[2]
    loss, sc_loss, mag_loss = 0., 0., 0.
    for f in funcs:                                  # host tensors: the reference's formulation on torch.stft
        win = f.window.to(pred.device)
        spec = lambda w: torch.stft(w, f.n_fft, f.hop_length, f.win_length, win, True, 'reflect', False, True,   # noqa: E731
                                    return_complex=True).abs()
        p_stft, t_stft = spec(pred), spec(target)
        n = t_stft.size(1) * t_stft.size(2)
        frob = lambda m: m.pow(2).sum((1, 2)).sqrt()                             # noqa: E731
        sc_loss_ = (frob(t_stft - p_stft) / frob(t_stft)).mean()
        mag_loss_ = (t_stft.add(eps).log() - p_stft.add(eps).log()).abs().sum((1, 2)).mean() / n
        loss += sc_loss_ + mag_loss_
        sc_loss += sc_loss_
        mag_loss += mag_loss_
    return loss / len(funcs), sc_loss / len(funcs), mag_loss / len(funcs)
