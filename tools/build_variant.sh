#!/bin/sh
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
# build a variant of libpsnd_hip.so with extra compiler flags: tools/build_variant.sh <name> <flags...>
# -> tools/mb/variants/libpsnd_<name>.so  (select it with PSND_LIB=...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/mb/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$root/include "$@" $root/pytorch_sound_amd/csrc/*.hip -o $root/tools/mb/variants/libpsnd_$name.so
echo $root/tools/mb/variants/libpsnd_$name.so
