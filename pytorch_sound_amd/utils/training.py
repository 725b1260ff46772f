import inspect
from typing import Any, Dict


def parse_model_kwargs(model_cls, **kwargs) -> Dict[str, Any]:
    """Keep only the entries of ``kwargs`` that name a positional/keyword argument of
    ``model_cls.__init__`` (pytorch_sound/utils/training.py:6-14: membership in
    ``inspect.getfullargspec(cls).args``, so **kw-only and **kwargs names do not count)."""
    accepted = set(inspect.getfullargspec(model_cls).args)
    return {name: value for name, value in kwargs.items() if name in accepted}
