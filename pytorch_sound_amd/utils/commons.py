import logging
from typing import Dict

import torch

__all__ = ['LOGGER', 'get_logger', 'log', 'get_loadable_checkpoint']

_FORMAT = '[%(asctime)s] [%(name)s] [%(levelname)s] %(message)s'


def get_logger(name: str) -> logging.Logger:
    """INFO-level stream logger with the reference's line format (utils/commons.py:25-41)."""
    logger = logging.getLogger(name)
    if not logger.handlers:
        logger.propagate = False
        logger.setLevel(logging.INFO)
        handler = logging.StreamHandler()
        handler.setLevel(logging.INFO)
        handler.setFormatter(logging.Formatter(_FORMAT))
        logger.addHandler(handler)
    return logger


LOGGER = get_logger('main')


def log(msg: str):
    LOGGER.info(msg)


def get_loadable_checkpoint(checkpoint: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Strip the ``module.`` that nn.DataParallel / DistributedDataParallel put in front of every
    key (utils/commons.py:55-66: keys that START with it lose every occurrence of the substring)."""
    return {(k.replace('module.', '') if k.startswith('module.') else k): v for k, v in checkpoint.items()}
