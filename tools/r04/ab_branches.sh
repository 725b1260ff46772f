# A/B: hifi_gan_v1 config-3 step with a stage's resblocks on parallel streams (PSND_HIFIGAN_BRANCHES=1), with / without batch sections
for v in "1 auto" "1 2" "1 auto" "1 2" "0 2"; do
  set -- $v
  PSND_HIFIGAN_BRANCHES=$1 PSND_CL_SECTIONS=$2 python tools/r04/run_leg.py config3 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/branches=$1 sections=$2 /"
done
