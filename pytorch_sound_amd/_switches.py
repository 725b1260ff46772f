"""Environment switches of the Python host side.

PRODUCT switches - documented behaviour, always read:
    PSND_LIB             path of the library build to load (_lib.py; default: libpsnd_hip.so next to the package)
    PSND_VOCODER_DIR     directory of the HiFi-GAN checkpoints (interface/hifi_gan.py)
    PSND_DDP_COMM        'bf16': gradient buckets cross the wire as bf16 (distributed.FlatGradReducer / Trainer.ddp_comm_dtype)
    PSND_DDP_FORCE       '1': keep the gradient reducer on a one-rank process group (the one-GPU tests and bench legs of the DDP path)
    PSND_DIST_SHARE_GPU  '1': every rank on device 0 over gloo (the world-2 tests on a one-GPU box)

LAB switches - A/B choices of launch groupings, stream branches and node granularities that the parity tests and tools/ flip (PSND_CL_*,
PSND_NO_*, PSND_HIFIGAN_BRANCHES, PSND_DDP_GRAPH / _RELEASE / _BRANCHES, PSND_PREFETCH_BRANCHES, PSND_MSL_FUSED, ...): read ONLY when PSND_LAB=1 is set in
the environment; without it every one of them takes its measured default, whatever the environment holds.  (The C side has the same split:
libpsnd_hip.so reads no environment variable at all, libpsnd_hip_lab.so - `_build --lab` - does.)
"""
import os


def lab_on() -> bool:
    return os.environ.get('PSND_LAB') == '1'


def lab(name, default=None):
    """the value of lab switch `name` (PSND_LAB=1 only), else `default`"""
    if os.environ.get('PSND_LAB') != '1':
        return default
    return os.environ.get(name, default)
