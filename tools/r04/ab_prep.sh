cd /tmp && export TMPDIR=/tmp
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
for v in 0 1; do
  rm -rf /tmp/prof
  export PSND_PREP_NO_DEFER=$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-legs > /tmp/b.log 2>&1
  echo "== no_defer $v"
  python $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/prof 30 | grep -E "prep_multi|chain_kernel<256, 2, true|bwd_pair|step:" | awk '{print $2, $4, $5, $6, $7}' | cut -c1-90
  python $GRAFT_REPO_ROOT/tools/r04/benchline.py $v < /tmp/b.log
done
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_conv.py tests/test_gpu_nfk_path.py tests/test_gpu_trainer_graph.py -x -q 2>&1 | tail -3
