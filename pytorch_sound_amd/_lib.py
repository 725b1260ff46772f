"""ctypes binding of libpsnd_hip.so (include/psnd.h).  The product path has NO fallback:
if the library is missing or a call fails, an exception is raised."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PSND_LIB', os.path.join(_HERE, 'libpsnd_hip.so'))   # PSND_LIB: A/B builds of the same ABI

PSND_OK = 0
FRAMING_CENTER = 0
FRAMING_HIFIGAN = 1
FRAMING_NONE = 2
LOG_NONE, LOG_E, LOG_10 = 0, 1, 2

_c = ctypes
_P = _c.c_void_p
_I64 = _c.c_int64
_INT = _c.c_int
_F = _c.c_float
_D = _c.c_double

# name -> (restype, argtypes); mirrors include/psnd.h one to one
SIGNATURES = {
    'psnd_version': (_INT, []),
    'psnd_last_error': (_c.c_char_p, []),
    'psnd_event_create': (_P, []),
    'psnd_event_destroy': (_INT, [_P]),
    'psnd_event_record_external': (_INT, [_P, _P]),
    'psnd_stream_wait_event': (_INT, [_P, _P]),
    'psnd_event_external_supported': (_INT, []),
    'psnd_frame_count': (_I64, [_I64, _INT, _INT, _INT]),
    'psnd_frame_sample_index': (_I64, [_I64, _INT, _I64, _INT, _INT, _INT]),
    'psnd_stft_plan_bytes': (_c.c_size_t, [_INT]),
    'psnd_stft_plan_build': (_INT, [_INT, _P, _P]),
    'psnd_stft_fwd': (_INT, [_P, _I64, _I64, _INT, _INT, _INT, _P, _F, _P, _P, _P, _P, _P]),
    'psnd_stft_mag_nfk': (_INT, [_P, _I64, _I64, _INT, _INT, _INT, _P, _F, _P, _P]),
    'psnd_stft_bwd': (_INT, [_P, _I64, _I64, _INT, _INT, _INT, _P, _F, _P, _P, _P, _P, _P]),
    'psnd_istft': (_INT, [_P, _P, _I64, _I64, _INT, _INT, _P, _F, _P, _P]),
    'psnd_mel_plan_bytes': (_c.c_size_t, [_INT, _INT]),
    'psnd_mel_plan_build': (_INT, [_INT, _INT, _P, _P]),
    'psnd_mel_fwd': (_INT, [_P, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P, _P]),
    'psnd_mel_bwd': (_INT, [_P, _P, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P]),
    'psnd_logmel_fwd': (_INT, [_P, _I64, _I64, _INT, _INT, _INT, _P, _F, _INT, _P, _INT, _F, _F, _F, _F, _P, _P]),
    'psnd_conv1d_cl': (_INT, [_P, _P, _P, _F, _P, _P, _P, _P, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _F, _F, _P, _P, _P, _P]),
    'psnd_conv1d_cl_pair_supported': (_INT, [_INT, _INT, _INT, _INT, _INT, _INT]),
    'psnd_conv1d_cl_pair': (_INT, [_P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _F, _P, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _F, _P, _P, _P]),
    'psnd_conv1d_cl_wgrad': (_INT, [_P, _P, _P, _F, _P, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P]),
    'psnd_conv1d_prep': (_INT, [_P, _P, _P, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P]),
    'psnd_conv1d_wnorm_bwd': (_INT, [_P, _P, _INT, _P, _P, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P]),
    'psnd_conv1d_cl_bwd': (_INT, [_P, _P, _P, _F, _P, _P, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _F, _P, _P, _P, _P]),
    'psnd_conv1d_cl_wgrad_multi_splits': (_INT, [_I64, _INT, _INT, _INT, _INT, _INT]),
    'psnd_conv1d_cl_wgrad_multi': (_INT, [_P, _INT, _I64, _INT, _P]),
    'psnd_nan_flag': (_INT, [_P, _I64, _P, _P]),
    'psnd_conv1d_cl_chain_rows': (_INT, [_INT, _INT, _INT, _P]),
    'psnd_conv1d_cl_chain_plan': (_INT, [_INT, _INT, _INT, _P, _I64, _P]),
    'psnd_conv1d_cl_chain': (_INT, [_P, _P, _P, _INT, _I64, _INT, _INT, _INT, _INT, _INT, _P]),
    'psnd_conv_chain_stats': (None, [_P]),
    'psnd_conv1d_cl_pair_bwd_supported': (_INT, [_INT, _INT, _INT, _INT, _INT, _INT]),
    'psnd_conv1d_cl_pair_bwd_splits': (_INT, [_I64, _INT, _INT, _INT]),
    'psnd_conv1d_cl_pair_bwd': (_INT, [_P, _P, _P, _F, _P, _P, _P, _F, _P, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P,
                                       _P, _P, _INT, _INT, _P, _P, _P]),
    'psnd_conv1d_wnorm_bwd_multi': (_INT, [_P, _INT, _P]),
    'psnd_conv1d_prep_multi': (_INT, [_P, _INT, _INT, _INT, _INT, _P]),
    'psnd_conv1d_cl_wgrad_splits': (_INT, [_I64, _INT, _INT, _INT, _INT]),
    'psnd_mask_head_l1_blocks': (_I64, [_I64, _I64, _INT]),
    'psnd_masked_l1_blocks': (_I64, [_I64]),
    'psnd_masked_l1_fwd': (_INT, [_P, _P, _P, _I64, _INT, _I64, _P, _P, _P, _P]),
    'psnd_masked_l1_bwd': (_INT, [_P, _P, _P, _I64, _INT, _I64, _P, _P, _P, _P, _P]),
    'psnd_to_cl_nfk': (_INT, [_P, _I64, _INT, _I64, _INT, _INT, _INT, _INT, _P, _P]),
    'psnd_mask_head_l1_blocks_nfk': (_I64, [_I64, _I64, _INT]),
    'psnd_mask_head_l1_fwd_nfk': (_INT, [_P, _P, _P, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P, _P]),
    'psnd_mask_head_l1_bwd_nfk': (_INT, [_P, _P, _P, _P, _P, _P, _F, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P]),
    'psnd_mel_fwd_nfk': (_INT, [_P, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P, _P]),
    'psnd_mel_bwd_nfk': (_INT, [_P, _P, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P]),
    'psnd_mel_l1_fwd_nfk': (_INT, [_P, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P, _P, _P]),
    'psnd_mel_l1_bwd_nfk': (_INT, [_P, _P, _P, _F, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P]),
    'psnd_mask_head_l1_fwd': (_INT, [_P, _P, _P, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P, _P]),
    'psnd_mask_head_l1_bwd': (_INT, [_P, _P, _P, _P, _P, _P, _F, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P]),
    'psnd_mel_l1_blocks': (_I64, [_I64, _I64, _INT]),
    'psnd_mel_l1_fwd': (_INT, [_P, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P, _P, _P]),
    'psnd_mel_l1_bwd': (_INT, [_P, _P, _P, _F, _I64, _I64, _INT, _INT, _P, _INT, _F, _F, _F, _F, _P, _P]),
    'psnd_l1_loss_combine': (_INT, [_P, _P, _P, _INT, _P, _P, _P]),
    'psnd_conv_stats': (_INT, [_P, _INT]),
    'psnd_conv_pair_stats': (_INT, [_P, _INT]),
    'psnd_convtr1d_prep_multi': (_INT, [_P, _INT, _INT, _P]),
    'psnd_convtr1d_prep': (_INT, [_P, _P, _P, _INT, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P]),
    'psnd_convtr1d_cl_fwd': (_INT, [_P, _P, _P, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _F, _P, _P, _P]),
    'psnd_convtr1d_cl_wgrad_splits': (_INT, [_I64, _INT, _INT, _INT, _INT]),
    'psnd_convtr1d_cl_bwd': (_INT, [_P, _P, _P, _F, _P, _P, _I64, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _INT, _P, _P, _P, _P]),
    'psnd_convtr1d_wnorm_bwd': (_INT, [_P, _INT, _P, _P, _INT, _INT, _INT, _INT, _INT, _INT, _P, _P, _P]),
    'psnd_cl_mean_act_fwd': (_INT, [_P, _P, _P, _P, _INT, _F, _P, _I64, _P]),
    'psnd_cl_mean_act_bwd': (_INT, [_P, _P, _INT, _F, _P, _I64, _P]),
    'psnd_cl_sum2': (_INT, [_P, _INT, _P, _P, _INT, _P, _I64, _P]),
    'psnd_grad_pack_bf16': (_INT, [_P, _P, _I64, _F, _P]),
    'psnd_grad_unpack_bf16': (_INT, [_P, _P, _I64, _F, _P]),
    'psnd_cl_colsum_splits': (_INT, [_I64, _INT]),
    'psnd_cl_colsum': (_INT, [_P, _I64, _INT, _INT, _INT, _INT, _P, _P, _P]),
    'psnd_posenc': (_INT, [_P, _P, _F, _I64, _INT, _I64, _I64, _P, _P]),
    'psnd_groupnorm1_fwd': (_INT, [_P, _P, _P, _P, _I64, _INT, _I64, _F, _INT, _P, _P, _P, _P]),
    'psnd_groupnorm1_bwd': (_INT, [_P, _P, _P, _P, _P, _P, _I64, _INT, _I64, _INT, _P, _P, _P, _P, _P]),
    'psnd_linear1x1_fwd': (_INT, [_P, _P, _P, _I64, _INT, _INT, _I64, _INT, _INT, _P, _P]),
    'psnd_linear1x1_wgrad_slabs': (_I64, [_I64, _INT, _INT, _I64]),
    'psnd_linear1x1_bwd': (_INT, [_P, _P, _P, _P, _I64, _INT, _INT, _I64, _INT, _P, _P, _P, _P, _P]),
    'psnd_linear1x1_bwd_acc': (_INT, [_P, _P, _P, _P, _I64, _INT, _INT, _I64, _INT, _P, _P, _P, _P, _P, _P]),
    'psnd_linear1x1_bwd_ex': (_INT, [_P, _P, _P, _P, _I64, _INT, _INT, _I64, _INT, _INT, _I64, _P, _P, _P, _P, _P, _P, _P]),
    'psnd_linear1x1_fwd_ex': (_INT, [_P, _P, _P, _I64, _INT, _INT, _I64, _INT, _INT, _INT, _I64, _P, _P]),
    'psnd_mha_fwd': (_INT, [_P, _P, _I64, _INT, _INT, _I64, _P, _P, _P, _INT, _P]),
    'psnd_mha_bwd': (_INT, [_P, _P, _P, _P, _P, _P, _P, _I64, _INT, _INT, _I64, _P, _P, _INT, _P]),
    'psnd_mha_bwd_parts': (_INT, [_P, _P, _P, _P, _P, _P, _P, _I64, _INT, _INT, _I64, _P, _P, _INT, _INT, _P]),
    'psnd_softmax_keys_fwd': (_INT, [_P, _P, _I64, _I64, _F, _P]),
    'psnd_softmax_keys_bwd': (_INT, [_P, _P, _I64, _I64, _F, _P, _P]),
    'psnd_polar_bwd': (_INT, [_P, _P, _P, _P, _I64, _P, _P, _P]),
    'psnd_preemphasis_fwd': (_INT, [_P, _I64, _I64, _F, _P, _P]),
    'psnd_preemphasis_bwd': (_INT, [_P, _I64, _I64, _F, _P, _P]),
    'psnd_stft_loss_blocks': (_I64, [_I64]),
    'psnd_stft_loss_partial': (_INT, [_P, _P, _I64, _I64, _F, _P, _P]),
    'psnd_stft_loss_final': (_INT, [_P, _P, _INT, _I64, _P, _P, _P]),
    'psnd_stft_loss_bwd': (_INT, [_P, _P, _I64, _I64, _F, _P, _P, _INT, _P, _P, _P]),
    'psnd_stft_bwd_msl_supported': (_INT, [_INT, _INT]),
    'psnd_stft_fwd_msl_blocks': (_I64, [_I64, _INT, _INT]),
    'psnd_stft_fwd_msl': (_INT, [_P, _I64, _I64, _INT, _INT, _P, _F, _P, _F, _P, _P]),
    'psnd_stft_loss_final_blocks': (_INT, [_P, _P, _P, _INT, _I64, _P, _P, _P]),
    'psnd_stft_bwd_msl': (_INT, [_P, _I64, _I64, _INT, _INT, _INT, _P, _F, _P, _P, _P, _INT, _F, _INT, _P, _P]),
    'psnd_frame_mask_frames': (_I64, [_I64, _INT, _INT]),
    'psnd_frame_mask': (_INT, [_P, _I64, _I64, _INT, _INT, _P, _P]),
    'psnd_pad_collate': (_INT, [_P, _P, _P, _I64, _I64, _P, _P, _P]),
    'psnd_im2col_f32': (_INT, [_P, _I64, _INT, _I64, _INT, _INT, _INT, _INT, _INT, _I64, _F, _P, _P]),
    'psnd_col2im_f32': (_INT, [_P, _P, _I64, _INT, _I64, _INT, _INT, _INT, _INT, _INT, _I64, _F, _P, _P]),
    'psnd_adam_chunk': (_I64, []),
    'psnd_adam_table_bytes': (_I64, []),
    'psnd_adam_step': (_INT, [_P, _INT, _P, _P, _I64, _D, _D, _D, _D, _D, _INT, _P, _P, _P, _F, _P, _P]),
    'psnd_adam_step_logged': (_INT, [_P, _INT, _P, _P, _I64, _D, _D, _D, _D, _D, _INT, _P, _P, _P, _F, _P, _P, _INT, _INT, _P]),
    'psnd_grad_sumsq': (_INT, [_P, _INT, _P, _P, _I64, _F, _P, _INT, _F, _P, _P, _P, _P]),
    'psnd_mask_head_fwd': (_INT, [_P, _P, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P]),
    'psnd_mask_head_bwd': (_INT, [_P, _P, _P, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P]),
    'psnd_pqmf_analysis': (_INT, [_P, _P, _I64, _I64, _INT, _INT, _INT, _F, _P, _P]),
    'psnd_pqmf_synthesis': (_INT, [_P, _P, _I64, _I64, _I64, _INT, _INT, _INT, _F, _P, _P]),
    'psnd_l1_loss_blocks': (_I64, [_I64]),
    'psnd_l1_loss_fwd': (_INT, [_P, _P, _I64, _P, _P, _P]),
    'psnd_l1_loss_bwd': (_INT, [_P, _P, _I64, _P, _P, _P, _P]),
    'psnd_l1_loss_sum_fwd': (_INT, [_P, _P, _P, _P, _INT, _P, _P, _P]),
    'psnd_l1_loss_bwd_w': (_INT, [_P, _P, _I64, _P, _D, _P, _P, _P]),
    'psnd_to_cl': (_INT, [_P, _I64, _INT, _I64, _INT, _INT, _INT, _INT, _P, _P]),
    'psnd_from_cl': (_INT, [_P, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P]),
    'psnd_from_cl_tanh': (_INT, [_P, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P]),
    'psnd_to_cl_tanh_bwd': (_INT, [_P, _P, _I64, _INT, _I64, _INT, _INT, _INT, _P, _P]),
}

_lib = None


class PsndError(RuntimeError):
    pass


class WgradDesc(ctypes.Structure):
    """psnd_wgrad_desc of include/psnd.h"""
    _fields_ = [('g', _P), ('xa', _P), ('gw_part', _P), ('gbias_part', _P), ('off0', _INT), ('dstep', _INT),
                ('Ca', _INT), ('Cb', _INT), ('k', _INT), ('splits', _INT)]


class ChainPair(ctypes.Structure):
    """psnd_chain_pair of include/psnd.h"""
    _fields_ = [('W1', _P), ('bias1', _P), ('act1_slope', _F), ('mid_out', _P), ('W2', _P), ('bias2', _P),
                ('off1', _INT), ('dstep1', _INT), ('off2', _INT), ('dstep2', _INT), ('act2_slope', _F), ('out_raw', _P), ('out_act', _P),
                ('M1', _P), ('M2', _P), ('m1_slope', _F), ('m2_slope', _F)]


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP extension is absent."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64: it must be in the process BEFORE our library is
        # dlopen'ed, otherwise the dynamic linker binds us to a second HIP runtime (/opt/rocm) that
        # never sees torch's device context ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise PsndError('libpsnd_hip.so not found at %s - run `python -m pytorch_sound_amd._build` '
                            '(there is no CPU / eager fallback)' % LIB_PATH)
        _lib = _load(LIB_PATH)
    return _lib


def _load(path):
    h = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in LAB_SIGNATURES.items():     # entry points of a -DPSND_LAB build only
        fn = getattr(h, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    return h


LAB_LIB_PATH = os.path.join(_HERE, 'libpsnd_hip_lab.so')     # `python -m pytorch_sound_amd._build --lab`
LAB_SIGNATURES = {'psnd_env_refresh': (None, [])}


def is_lab() -> bool:
    """True when the loaded library is a lab build (reads PSND_* A/B switches from the environment)"""
    return hasattr(lib(), 'psnd_env_refresh')


def refresh_switches():
    """lab build: have the PSND_* switches looked up again after the environment changed; product build: nothing to refresh"""
    if _lib is not None and hasattr(_lib, 'psnd_env_refresh'):
        _lib.psnd_env_refresh()


class use_library:
    """context manager: route every call of this process through another build of the same ABI (the lab library in parity tests of kernel
    instances, tools/ A/B runs) and back.  Plans and tensors are plain memory: they carry over."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        lib()
        if not os.path.exists(self.path):
            raise PsndError('%s not found - run `python -m pytorch_sound_amd._build --lab`' % self.path)
        self.prev, _lib = _lib, _load(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def check(rc, what):
    if rc != PSND_OK:
        msg = lib().psnd_last_error()
        raise PsndError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))


def ptr(t):
    """device (or host) pointer of a contiguous torch tensor, or NULL for None."""
    if t is None:
        return None
    return _P(t.data_ptr())


def np_ptr(a):
    return _P(a.ctypes.data)


def stream_ptr(device):
    import torch
    return _P(torch.cuda.current_stream(device).cuda_stream)


def frame_count(T, n_fft, hop, framing=FRAMING_CENTER):
    return int(lib().psnd_frame_count(int(T), int(n_fft), int(hop), int(framing)))


def build_stft_plan(n_fft, window):
    """window: array-like of n_fft taps (already centre-padded).  Returns a uint8 numpy plan."""
    w = np.ascontiguousarray(np.asarray(window, dtype=np.float32))
    if w.shape != (n_fft,):
        raise PsndError('stft plan: window must have n_fft=%d taps, got %s' % (n_fft, w.shape))
    nb = lib().psnd_stft_plan_bytes(int(n_fft))
    if nb == 0:
        raise PsndError('stft plan: n_fft=%d unsupported (power of two in [16, 8192])' % n_fft)
    plan = np.zeros(nb, dtype=np.uint8)
    check(lib().psnd_stft_plan_build(int(n_fft), np_ptr(w), np_ptr(plan)), 'psnd_stft_plan_build')
    return plan


def build_mel_plan(mel_filter):
    W = np.ascontiguousarray(np.asarray(mel_filter, dtype=np.float32))
    M, K = W.shape
    nb = lib().psnd_mel_plan_bytes(M, K)
    plan = np.zeros(nb, dtype=np.uint8)
    check(lib().psnd_mel_plan_build(M, K, np_ptr(W), np_ptr(plan)), 'psnd_mel_plan_build')
    return plan
