"""Registry semantics of pytorch_sound/models/__init__.py (G8)."""
import pytest
import torch.nn as nn

import pytorch_sound_amd.models as M
from pytorch_sound_amd.models import build_model, register_model, register_model_architecture
from pytorch_sound_amd.utils.training import parse_model_kwargs


@pytest.fixture()
def clean():
    saved = [dict(d) for d in (M.MODEL_REGISTRY, M.ARCH_MODEL_REGISTRY, M.ARCH_MODEL_INV_REGISTRY, M.ARCH_CONFIG_REGISTRY)]
    yield
    for d, s in zip((M.MODEL_REGISTRY, M.ARCH_MODEL_REGISTRY, M.ARCH_MODEL_INV_REGISTRY, M.ARCH_CONFIG_REGISTRY), saved):
        d.clear()
        d.update(s)


def test_register_and_build(clean):
    @register_model('toy')
    class Toy(nn.Module):
        def __init__(self, a, b=2, *, kwonly=5):
            super().__init__()
            self.a, self.b, self.kwonly = a, b, kwonly

    @register_model_architecture('toy', 'toy_small')
    def toy_small():
        return {'a': 1, 'b': 3, 'not_an_arg': 9, 'kwonly': 7}

    m = build_model('toy_small')
    assert (m.a, m.b, m.kwonly) == (1, 3, 5)                       # kw-only names are not in getfullargspec().args
    m = build_model('toy_small', {'b': 10, 'c': 99, 'not_an_arg': 1})
    assert (m.a, m.b) == (1, 10)                                   # only keys that survived the filter are overridable
    assert M.ARCH_MODEL_INV_REGISTRY['toy'] == ['toy_small']
    assert M.ARCH_MODEL_REGISTRY['toy_small'] is Toy
    assert parse_model_kwargs(Toy, a=1, z=2, self=3) == {'a': 1, 'self': 3}


def test_errors(clean):
    @register_model('toy')
    class Toy(nn.Module):
        pass

    with pytest.raises(ValueError, match='duplicate model'):
        register_model('toy')(Toy)
    with pytest.raises(ValueError, match='unknown model type'):
        register_model_architecture('nope', 'x')(lambda: {})
    register_model_architecture('toy', 'toy_a')(lambda: {})
    with pytest.raises(ValueError, match='duplicate model architecture'):
        register_model_architecture('toy', 'toy_a')(lambda: {})
    with pytest.raises(ValueError, match='must be callable'):
        register_model_architecture('toy', 'toy_b')({'a': 1})
    with pytest.raises(KeyError):
        build_model('never_registered')


def test_alias_package_is_same_objects():
    import pytorch_sound.models as PM
    from pytorch_sound.trainer import Trainer as T1
    from pytorch_sound_amd.trainer import Trainer as T2
    assert PM.MODEL_REGISTRY is M.MODEL_REGISTRY and T1 is T2
    from pytorch_sound import settings
    assert (settings.SAMPLE_RATE, settings.N_FFT, settings.WIN_LENGTH, settings.HOP_LENGTH, settings.SPEC_SIZE,
            settings.MEL_SIZE, settings.MEL_MIN, settings.MEL_MAX, settings.MIN_DB, settings.MAX_DB) == \
        (22050, 1024, 1024, 256, 513, 80, 0, 8000, -50, 30)
    assert settings.HOP_STRIDE == 4 and settings.VN_DB == -11.5 and settings.MULAW_BINS == 256
