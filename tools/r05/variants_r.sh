#!/bin/sh
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
# ablation builds of psnd_stft_r.hip (PSND_R_ABL bits) -> tools/mb/variants/libpsnd_rabl<bits>.so
set -e
cd "$(dirname "$0")/../.."
python -m pytorch_sound_amd._build > /dev/null
for b in "$@"; do sh tools/r04/variant_q.sh rabl$b psnd_stft_r -DPSND_R_ABL=$b; done
