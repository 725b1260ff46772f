# spread of the legs with the graph branches over repeated runs (an outlier = a branch behind another stream's work on a hardware queue)
for i in 1 2 3 4 5 6; do
  python tools/r04/run_leg.py config3 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/config3 /"
  python tools/r04/run_leg.py config4 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/config4 /"
done
python bench.py --force-ddp --steps 30 --warmup 10 --cpu-seconds 0 --no-legs 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/config2 one-rank ddp /"
