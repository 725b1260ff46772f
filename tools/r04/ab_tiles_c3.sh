# config 3 with the graph branches: do the row-tile choices made for launches that have the chip to themselves still hold?
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
for v in "" "PSND_CONV_MT3=0" "PSND_CONV_MT=1" "PSND_CONV_MT=2" "PSND_PAIR_MT=1" "PSND_PAIR_MT=2" ""; do
  env $v python tools/r04/run_leg.py config3 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/config3 [$v] /"
done
