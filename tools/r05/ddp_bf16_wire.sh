# one-rank RCCL runs: fp32 buckets on the wire against their bf16 image (FlatGradReducer comm_dtype, PSND_DDP_COMM=bf16): what the two extra
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
# launches per bucket (pack / unpack on the release stream) cost on one GPU - the bytes they save only show on >= 2 devices
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/r05; mkdir -p $OUT
for rep in 1 2; do
for w in fp32 bf16; do
  PSND_DDP_COMM=$w python bench.py --steps 60 --warmup 10 --no-legs --cpu-seconds 0 --force-ddp 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 2 wire $w', round(d['ms_per_step'],4))"
  for c in 3 4; do
  PSND_DDP_COMM=$w python bench.py --config $c --steps 60 --warmup 5 --settle 20 --force-ddp 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c wire $w', round(d['ms_per_step'],3), d['reducer'])"
  done
done
done | tee $OUT/ddp_bf16_wire.txt
