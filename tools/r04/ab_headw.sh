for v in 1 0 1 0 1 0; do
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
  PSND_HEAD_W_SIDE=$v python bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-legs 2>&1 | grep -o '"blocks_ms_per_step": [^]]*' | head -1 | sed "s/^/config2 head_w_side=$v /"
done
PSND_HEAD_W_SIDE=1 timeout 300 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -2
