# config-3 step against the weight-gradient workgroup target of the paired backward launches (PSND_WGRAD_BLOCKS, default 192; narrow 32 -> 32 layers: PSND_WGRAD_BLOCKS_NARROW)
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
r() { python bench.py --leg config3 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
echo base $(r) $(r)
for b in 48 64 96 112 128 144; do echo WGRAD_BLOCKS=$b $(PSND_WGRAD_BLOCKS=$b r) $(PSND_WGRAD_BLOCKS=$b r); done
for b in 64 96 128; do echo BLOCKS=96 NARROW=$b $(PSND_WGRAD_BLOCKS=96 PSND_WGRAD_BLOCKS_NARROW=$b r); done
echo base $(r)
