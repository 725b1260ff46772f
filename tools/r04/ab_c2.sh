# A/B: config-2 step with the input layout change next to the weight prep and the backward packs next to the loss backward (PSND_BRANCH_PARAM_GRADS)
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
for v in 1 0 1 0 1 0; do
  PSND_BRANCH_PARAM_GRADS=$v python bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-legs 2>&1 | grep -o '"blocks_ms_per_step": [^]]*' | head -1 | sed "s/^/config2 param_side=$v /"
done
timeout 500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trainer_graph.py -x -q 2>&1 | tail -3
