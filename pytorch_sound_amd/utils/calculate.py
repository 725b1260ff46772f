"""Element-wise helpers mirroring pytorch_sound/utils/calculate.py (adjacent to the hot path;
cheap enough that they stay as array expressions)."""
import math
from typing import Union

import numpy as np
import torch

from pytorch_sound_amd import settings

TensorOrArr = Union[torch.Tensor, np.ndarray]


def db2log(db):
    """dB (power) -> natural log: ln(10^(db/10)) (calculate.py:10-19); numpy path for ndarray/int,
    torch path otherwise."""
    if isinstance(db, (np.ndarray, int)):
        return np.log(np.power(10, db / 10))
    return torch.log(torch.pow(10, db / 10.))


def _mel_range():
    return db2log(settings.MIN_DB), db2log(settings.MAX_DB)


def unnorm_mel(x: TensorOrArr) -> TensorOrArr:
    """[-1, 1] -> log-mel in [ln MIN_DB, ln MAX_DB] (calculate.py:22-29)."""
    lo, hi = _mel_range()
    return ((x + 1) / 2) * (hi - lo) + lo


def norm_mel(x: TensorOrArr) -> TensorOrArr:
    """clip to [ln MIN_DB, ln MAX_DB] then map to [-1, 1] (calculate.py:32-43)."""
    lo, hi = _mel_range()
    x = x.clip(lo, hi) if type(x) == np.ndarray else x.clamp(lo, hi)
    return (x - lo) / (hi - lo) * 2 - 1


def volume_norm_log(x: np.ndarray, target_db: float = -11.5) -> np.ndarray:
    """scale so that std(x) == 10^(target_db/10) (calculate.py:46-53)."""
    return x / (np.std(x) / 10 ** (target_db / 10))


def volume_norm_log_torch(x: torch.Tensor, target_db: float = -11.5) -> torch.Tensor:
    return x / (torch.std(x) / 10 ** (target_db / 10))


def conv_same_padding(filter_size: int, stride: int, dilation: int, x: int = 44100) -> int:
    """'same' padding for a length-x input (calculate.py:66-70)."""
    eff = filter_size + (filter_size - 1) * (dilation - 1)
    return int(math.ceil(((x / stride - 1) * stride + eff - x) / 2))
