from typing import Any

import numpy as np
import torch


def to_device(tup: Any):
    """Lazy H2D of a batch (pytorch_sound/utils/tensor.py:6-15): a single tensor is wrapped into a
    1-tuple; returns a ``map`` object issuing ``x.cuda(non_blocking=True)`` per element, so the copies
    are enqueued when the Trainer unpacks it into ``forward(*batch)``.  Non-tensor items raise, as in
    the reference."""
    if not isinstance(tup, (tuple, list)):
        tup = (tup,)
    return map(lambda x: x.cuda(non_blocking=True), tup)


def to_numpy(gpu_tensor: torch.Tensor) -> np.ndarray:
    return gpu_tensor.detach().cpu().numpy()


def concat_complex(a: torch.Tensor, b: torch.Tensor, dim: int = 1) -> torch.Tensor:
    """[a_re, a_im] ++ [b_re, b_im] -> [a_re, b_re, a_im, b_im] along ``dim`` (utils/tensor.py:27-37)."""
    a_re, a_im = a.chunk(2, dim)
    b_re, b_im = b.chunk(2, dim)
    return torch.cat([a_re, b_re, a_im, b_im], dim=dim)
