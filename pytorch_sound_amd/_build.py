"""Build libpsnd_hip.so in-tree with hipcc for gfx950 (no torch headers, plain C ABI).

`python -m pytorch_sound_amd._build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU.  Objects are cached under csrc/build/ by source mtime.

`--lab` builds libpsnd_hip_lab.so from the same sources with -DPSND_LAB: the PSND_* environment switches of the dispatchers (kernel-instance
choices, ablations), psnd_env_refresh() - what tools/ and the parity tests of kernel instances load (tests/conftest.py `lab_lib`).  The
product library has no environment switch.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
BUILD = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libpsnd_hip.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libpsnd_hip.so cannot be built')


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'psnd.h'))
    return max(os.path.getmtime(h) for h in hdrs)


LAB_LIB = os.path.join(HERE, 'libpsnd_hip_lab.so')


def _compile(src, hipcc, hdr_mtime, verbose, lab=False):
    obj = os.path.join(BUILD, os.path.basename(src)[:-4] + ('.lab.o' if lab else '.o'))
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_mtime):
        return obj
    cmd = [hipcc] + FLAGS + (['-DPSND_LAB'] if lab else []) + ['-c', src, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
    return obj


def build(verbose=True, force=False, lab=False):
    """lab: the -DPSND_LAB build (libpsnd_hip_lab.so) instead of the product library"""
    LIB = LAB_LIB if lab else globals()['LIB']
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    hdr_mtime = _deps_mtime()
    if force:
        for f in os.listdir(BUILD):
            if f.endswith('.lab.o') == lab:
                os.remove(os.path.join(BUILD, f))
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hipcc, hdr_mtime, verbose, lab), srcs))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, lab='--lab' in sys.argv))
