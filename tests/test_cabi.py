"""libpsnd_hip.so: loads without a GPU, exports every symbol include/psnd.h declares, host-side
entry points (integer framing contract, plan builders, argument validation) behave."""
import ctypes
import os
import re
import numpy as np
import pytest

from conftest import ROOT
from oracle import features as ofe


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'psnd.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'#ifdef PSND_LAB.*?#endif', '', txt, flags=re.S)         # lab-only entry points (libpsnd_hip_lab.so)
    return sorted(set(re.findall(r'\b(psnd_[a-z0-9_]+)\s*\(', txt)))


@pytest.fixture(scope='module')
def L():
    from pytorch_sound_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build(verbose=False)
    return _lib


def test_exports_every_declared_symbol(L):
    h = ctypes.CDLL(L.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(h, n), 'include/psnd.h declares %s but the library does not export it' % n
    assert sorted(L.SIGNATURES) == names, 'ctypes table and header disagree'
    assert not hasattr(h, 'psnd_env_refresh'), 'the product library exports the lab build\'s switch refresh'
    assert L.lib().psnd_version() >= 100


def test_frame_contract_matches_oracle(L):
    lib = L.lib()
    for T, n, h in [(44100, 1024, 256), (8192, 1024, 256), (1323000, 4096, 1024), (700, 256, 64), (2049, 1024, 255),
                    (513, 1024, 256), (100, 1024, 256)]:
        for framing in (0, 1):
            F = ofe.frame_count(T, n, h, framing)
            assert lib.psnd_frame_count(T, n, h, framing) == F
            if T <= ofe.pad_amount(n, h, framing):
                continue
            for f in sorted({0, 1, 2, F // 2, max(F - 2, 0), max(F - 1, 0)}):
                ms = np.array([0, 1, n // 2 - 1, n // 2, n - 1])
                want = ofe.frame_sample_index(f, ms, T, n, h, framing)
                got = [lib.psnd_frame_sample_index(f, int(m), T, n, h, framing) for m in ms]
                assert list(want) == got


def test_plans_build_on_host(L):
    for n in (64, 256, 512, 1024, 2048, 4096):
        w = ofe.analysis_window(n)
        plan = L.build_stft_plan(n, w)
        assert plan.nbytes == L.lib().psnd_stft_plan_bytes(n) > 0
        pf = plan.view(np.float32)
        raw = pf[:n] if n in (64, 4096) else pf[-2 * n:-n]            # generic sizes lead with it, tuned ones append a permuted copy
        assert np.array_equal(raw, w)                                # the raw window is in every plan
    assert L.lib().psnd_stft_plan_bytes(1000) == 0                   # not a power of two
    W = ofe.mel_filterbank(22050, 1024, 80, 0, 8000)
    mp = L.build_mel_plan(W).view(np.int32)
    assert list(mp[:6]) == [80, 513, 5, 129, 33, 20]
    bands = mp[8:8 + 10].reshape(5, 2)
    nz = [np.nonzero(W[16 * t:16 * t + 16].sum(0))[0] for t in range(5)]
    for t in range(5):
        assert bands[t, 0] == nz[t].min() // 4 and bands[t, 1] == nz[t].max() // 4 + 1
    assert (bands[:, 1] - bands[:, 0]).sum() < 0.25 * 5 * 129        # band sparsity is real


def test_argument_validation_without_gpu(L):
    lib = L.lib()
    assert lib.psnd_stft_fwd(None, 1, 100, 1024, 256, 0, None, 0.0, None, None, None, None, None) == -1
    assert b'null' in lib.psnd_last_error()
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.psnd_stft_fwd(p, 1, 100, 1024, 256, 0, p, 0.0, p, None, None, None, None) == -2     # T <= pad
    assert lib.psnd_stft_fwd(p, 1, 5000, 1000, 256, 0, p, 0.0, p, None, None, None, None) == -4    # n_fft unsupported
    assert lib.psnd_stft_fwd(p, 1, 5000, 1024, 256, 7, p, 0.0, p, None, None, None, None) == -1    # framing enum
    assert lib.psnd_mel_fwd(None, 1, 1, 80, 513, None, 1, 0.0, -1.0, 0.0, 0.0, None, None, None) == -1
    assert lib.psnd_stft_bwd(p, 1, 5000, 1024, 256, 0, p, 0.0, None, None, None, p, None) == -1    # no gradient source


def test_argument_validation_of_the_wider_entry_points(L):
    """every entry point added behind the feature path rejects bad arguments before touching the device (0 work sizes are OK)"""
    lib = L.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.psnd_preemphasis_fwd(p, 1, 1, 0.97, p, None) == -2                      # reflect pad needs T >= 2
    assert lib.psnd_preemphasis_fwd(None, 1, 8, 0.97, p, None) == -1
    assert lib.psnd_stft_loss_blocks(0) == 0 and lib.psnd_stft_loss_blocks(8192) == 1 and lib.psnd_stft_loss_blocks(8193) == 2
    assert lib.psnd_stft_loss_partial(p, p, 0, 10, 1e-5, p, None) == 0               # empty batch: nothing to do
    assert lib.psnd_stft_loss_final(p, p, 9, 1, p, p, None) == -1                    # more than 8 resolutions
    assert lib.psnd_l1_loss_blocks(16384) == 1 and lib.psnd_l1_loss_blocks(16385) == 2
    assert lib.psnd_l1_loss_fwd(p, p, 0, p, p, None) == -2
    assert lib.psnd_pad_collate(p, p, p, 0, 100, p, None, None) == 0
    assert lib.psnd_pad_collate(p, p, p, 70000, 100, p, None, None) == -2
    assert lib.psnd_pqmf_analysis(p, p, 1, 64, 4, 61, 0, 1.0, p, None) == -2           # odd tap count
    assert lib.psnd_pqmf_analysis(p, p, 1, 64, 17, 62, 0, 1.0, p, None) == -2          # too many subbands
    assert lib.psnd_pqmf_synthesis(p, p, 0, 16, 64, 4, 62, 0, 4.0, p, None) == 0
    assert lib.psnd_adam_step(p, 1, p, p, 1, 1e-3, 1.0, 0.999, 1e-8, 0.0, 0, None, None, p, 0.0, None, None) == -1    # beta1 = 1
    assert lib.psnd_adam_step(p, 1, p, p, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, None, p, -1.0, None, None) == -1   # clip_value < 0
    assert lib.psnd_grad_sumsq(p, 1, p, p, 1, 0.0, None, 0, 1.0, None, p, p, None) == -1                              # no scratch
    assert lib.psnd_grad_sumsq(p, 1, p, p, 1, 0.0, None, 0, -1.0, p, p, p, None) == -1                                # max_norm < 0
    assert lib.psnd_polar_bwd(None, None, p, p, 4, p, p, None) == -1                                                   # no gradient given
    assert lib.psnd_adam_step(p, 0, p, p, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, None, p, 0.0, None, None) == 0
    assert lib.psnd_adam_chunk() == 2048 and lib.psnd_adam_table_bytes() == 48
    assert lib.psnd_conv1d_wnorm_bwd_multi(p, 0, None) == -1 and lib.psnd_conv1d_wnorm_bwd_multi(p, 33, None) == -1
    assert lib.psnd_mask_head_fwd(None, p, 1, 8, 8, 8, 0, 32, p, None) == -1
    assert lib.psnd_conv1d_cl(p, None, None, 0.0, p, None, None, None, 1, 16, 8, 4, 32, 40, 3, -1, 1, 0.1, 1.0, p, None, None, None) == -2   # Cb % 32


def test_product_has_no_fallback(L, monkeypatch):
    """a CPU tensor or a missing library must raise, never compute on the host."""
    import torch
    from pytorch_sound_amd import kernels
    with pytest.raises(L.PsndError):
        kernels.stft_forward(torch.zeros(1, 4096), 1024, 256, torch.zeros(8, dtype=torch.uint8))
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', '/nonexistent/libpsnd_hip.so')
    with pytest.raises(L.PsndError):
        L.lib()
