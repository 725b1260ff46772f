mkdir -p gpurun_out/r05
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
run() { env "$@" MASTER_PORT=$((29600 + RANDOM % 300)) python bench.py --config ${CFG:-3} --steps 40 --warmup 5 --settle 10 --force-ddp 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print('ok  ', 'config ${CFG:-3}', '$*', round(d['ms_per_step'],3), d['reducer'])
except Exception: print('FAIL', 'config ${CFG:-3}', '$*')
"; }
run X=1
run PSND_BRANCH_PARAM_GRADS=0
run PSND_HIFIGAN_BRANCHES=0
run PSND_DDP_BRANCHES=0
run PSND_DDP_RELEASE=current PSND_DDP_BRANCHES=0
run PSND_DDP_GRAPH=deferred
CFG=4 run X=1
CFG=4 run PSND_DDP_BRANCHES=0
CFG=4 run PSND_DDP_GRAPH=deferred
