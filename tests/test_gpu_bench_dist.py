"""bench.py under torch.distributed.run (the driver's multi-GPU launch line), on the one GPU of the test box: two ranks share the
device over gloo (PSND_DIST_SHARE_GPU=1 - RCCL refuses two ranks on one device).  Checks the contract of the JSON line: one
line, rank 0 only, n_gpus 2, global batch 64 (weak scaling: 32 clips per rank), a finite whole-job value consistent with
ms_per_step, no cpu_baseline at N > 1."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_bench_two_ranks_under_torch_distributed_run():
    env = dict(os.environ, PSND_DIST_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                              # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert out['config']['global_batch'] == 64 and out['config']['parallelism'] == 'dp2'
    assert out['scaling'] == 'weak' and out['unit'] == 'audio-s/s' and out['higher_is_better'] is True
    assert math.isfinite(out['value']) and out['value'] > 0
    # whole-job aggregate: 2 ranks x 32 clips x 2 s per step
    assert abs(out['value'] - 2 * 32 * 2.0 / (out['ms_per_step'] * 1e-3)) <= 1e-6 * out['value']
    assert 'cpu_baseline' not in out
    assert out['roofline']['bound'] == 'hbm' and 0 < out['roofline']['frac'] < 1
    assert len(out['timing']['blocks_ms_per_step']) >= 5


@pytest.mark.timeout(600)
@pytest.mark.parametrize('extra', [[], ['--config', '3'], ['--config', '5']])
def test_plain_command_starts_its_own_ranks(extra):
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run (the shape of the driver's one-GPU command): bench.py re-executes
    itself under torch.distributed.run and rank 0's one JSON line comes through."""
    env = dict(os.environ, PSND_DIST_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert math.isfinite(out['value']) and out['value'] > 0
