"""Model / architecture registry - the drop-in surface of pytorch_sound/models/__init__.py:9-85.

Four module-level dicts and three functions with the reference's exact semantics:

* ``register_model(name)``            class decorator; a second registration of ``name`` -> ValueError
* ``register_model_architecture(model_name, arch_name)``
                                      decorator for a zero-argument function returning the constructor
                                      kwargs; unknown model, duplicate arch or non-callable -> ValueError
* ``build_model(arch_name, extra_kwargs=None)``
                                      KeyError for an unknown arch; the arch's kwargs are filtered to the
                                      constructor's argument names; ``extra_kwargs`` may only override keys
                                      that survived that filter (others are silently ignored).
"""
from typing import Any, Callable, Dict, Optional

import torch.nn as nn

from pytorch_sound_amd.utils.training import parse_model_kwargs

MODEL_REGISTRY: Dict[str, type] = {}
ARCH_MODEL_REGISTRY: Dict[str, type] = {}
ARCH_MODEL_INV_REGISTRY: Dict[str, list] = {}
ARCH_CONFIG_REGISTRY: Dict[str, Callable[[], Dict[str, Any]]] = {}


def build_model(arch_name: str, extra_kwargs: Optional[Dict[str, Any]] = None) -> nn.Module:
    model_cls = ARCH_MODEL_REGISTRY[arch_name]                      # KeyError on purpose
    ctor_kwargs = parse_model_kwargs(model_cls, **ARCH_CONFIG_REGISTRY[arch_name]())
    for key, value in (extra_kwargs or {}).items():
        if key in ctor_kwargs:                                      # only already-present keys
            ctor_kwargs[key] = value
    return model_cls(**ctor_kwargs)


def register_model(name: str) -> Callable:
    def _decorate(cls):
        if name in MODEL_REGISTRY:
            raise ValueError('Cannot register duplicate model ({})'.format(name))
        MODEL_REGISTRY[name] = cls
        return cls
    return _decorate


def register_model_architecture(model_name: str, arch_name: str) -> Callable:
    def _decorate(fn):
        if model_name not in MODEL_REGISTRY:
            raise ValueError('Cannot register model architecture for unknown model type ({})'.format(model_name))
        if arch_name in ARCH_MODEL_REGISTRY:
            raise ValueError('Cannot register duplicate model architecture ({})'.format(arch_name))
        if not callable(fn):
            raise ValueError('Model architecture must be callable ({})'.format(arch_name))
        ARCH_MODEL_REGISTRY[arch_name] = MODEL_REGISTRY[model_name]
        ARCH_MODEL_INV_REGISTRY.setdefault(model_name, []).append(arch_name)
        ARCH_CONFIG_REGISTRY[arch_name] = fn
        return fn
    return _decorate
