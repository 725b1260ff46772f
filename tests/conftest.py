import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# the host side reads its A/B switches (PSND_CL_*, PSND_NO_*, PSND_DDP_GRAPH ...: pytorch_sound_amd/_switches.py) only in a lab environment; the
# tests flip them (monkeypatch.setenv / os.environ, child processes inherit) - unflipped, every switch has its product default
os.environ.setdefault('PSND_LAB', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'native_precision: the HiFi-GAN generator picks its arithmetic by the default rule (fp32 outside autocast)')


def pytest_collection_modifyitems(config, items):
    """GPU tests never silently pass on a box without a GPU: they are skipped only when the
    run did not ask for them; under `-m gpu` a missing device or library is a hard failure."""
    import torch
    if config.pluginmanager.hasplugin('timeout'):
        # a GPU test that hangs (a missing stream join, a wedged queue) ends the run with the stacks of all threads after 4 minutes (the slowest test takes 26 s) instead
        # of sitting there until the caller's own limit; 'thread': a blocked hipStreamSynchronize does not return to a signal handler
        for it in items:
            if 'gpu' in it.keywords and it.get_closest_marker('timeout') is None:
                it.add_marker(pytest.mark.timeout(240, method='thread'))
    if torch.cuda.is_available():
        return
    asked = 'gpu' in (config.getoption('-m') or '') and 'not gpu' not in (config.getoption('-m') or '')
    if asked:
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


def seeded_wav(seed, N, T, sr=22050):
    """same recipe as tools/gen_golden.py / BASELINE.md synthetic input"""
    g = np.random.RandomState(seed)
    t = np.arange(T) / sr
    w = 0.0708 * g.randn(N, T) + 0.1 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t + 0.3)
    return np.clip(w, -1, 1).astype(np.float32)


def _refresh_switches():
    """A LAB build of the library (libpsnd_hip_lab.so, fixture `lab_lib`) looks its PSND_* A/B switches up once per call site: tests that flip
    them inside one process have them read again.  The product library has no switches - nothing to refresh."""
    try:
        from pytorch_sound_amd import _lib
        _lib.refresh_switches()
    except Exception:      # noqa: BLE001 - CPU-only collection without the library
        pass


@pytest.fixture
def lab_lib():
    """route the test's calls through the lab build of the library (-DPSND_LAB: the dispatchers' PSND_* switches select kernel instances that
    the product library only takes at other sizes).  Built by __graft_entry__.build() / `python -m pytorch_sound_amd._build --lab`."""
    from pytorch_sound_amd import _lib
    with _lib.use_library(_lib.LAB_LIB_PATH) as h:
        assert hasattr(h, 'psnd_env_refresh')
        _lib.refresh_switches()
        yield h
        _lib.refresh_switches()


@pytest.fixture(autouse=True)
def _fresh_switches():
    _refresh_switches()
    yield
    _refresh_switches()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch, with the library's cached switches refreshed after every environment change"""
    set_, del_ = monkeypatch.setenv, monkeypatch.delenv

    def setenv(*a, **k):
        set_(*a, **k)
        _refresh_switches()

    def delenv(*a, **k):
        del_(*a, **k)
        _refresh_switches()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    return monkeypatch


@pytest.fixture(autouse=True)
def _generator_precision(request):
    """The HiFi-GAN generator gives an fp32 HIP tensor outside autocast the reference's fp32 convolutions (Generator.precision = 'auto',
    round 6).  The tests written for the channels-last bf16 kernels hand it fp32 tensors without autocast: they opt into those kernels
    explicitly (precision = 'bf16'); tests marked `native_precision` run with the default rule."""
    try:
        from pytorch_sound_amd.models.vocoders.hifi_gan import Generator
    except Exception:      # noqa: BLE001
        yield
        return
    from pytorch_sound_amd.models.separator import ConvSeparator
    old = Generator.precision, ConvSeparator.precision
    Generator.precision = ConvSeparator.precision = 'auto' if request.node.get_closest_marker('native_precision') else 'bf16'
    yield
    Generator.precision, ConvSeparator.precision = old
