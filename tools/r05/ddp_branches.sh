# one-rank RCCL runs of configs 3 / 4: graph branches next to the reducer (round 5) against branches off (round 4) and no reducer
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
for c in 3 4; do
python bench.py --config $c --steps 60 --warmup 5 --settle 20 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c no reducer       ', round(d['ms_per_step'],3), d['reducer'])"
python bench.py --config $c --steps 60 --warmup 5 --settle 20 --force-ddp 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c reducer+branches ', round(d['ms_per_step'],3), d['reducer'])"
PSND_DDP_BRANCHES=0 python bench.py --config $c --steps 60 --warmup 5 --settle 20 --force-ddp 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c reducer, no branch', round(d['ms_per_step'],3), d['reducer'])"
done
