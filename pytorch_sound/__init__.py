"""Drop-in name: ``import pytorch_sound...`` resolves to the MI355X-native implementation.

Every ``pytorch_sound.X`` is THE SAME module object as ``pytorch_sound_amd.X`` (one registry, one
Trainer class), provided through a meta-path alias finder - no code lives here.  Only the parts of
the reference on the accelerated path exist; anything else raises
ModuleNotFoundError, as it should.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_SRC, _DST = 'pytorch_sound', 'pytorch_sound_amd'


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self._target = target

    def create_module(self, spec):
        return importlib.import_module(self._target)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_SRC + '.'):
            return None
        real = _DST + fullname[len(_SRC):]
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name == real:
                return None
            raise
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=hasattr(mod, '__path__'))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

_impl = importlib.import_module(_DST)
__version__ = _impl.__version__
__path__ = []          # submodules come from the finder only
