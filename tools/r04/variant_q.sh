#!/bin/sh
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
# A/B builds of ONE source file: tools/r04/variant_q.sh <name> <file.hip> <flags...> -> tools/mb/variants/libpsnd_<name>.so (PSND_LIB=...)
# (the other objects are the ones of the regular build, pytorch_sound_amd/csrc/build)
set -e
name=$1; src=$2; shift; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/tools/mb/variants
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $root/pytorch_sound_amd/csrc/$base.hip -o /tmp/var_$name.o
objs=$(ls $root/pytorch_sound_amd/csrc/build/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/mb/variants/libpsnd_$name.so $objs /tmp/var_$name.o
echo $root/tools/mb/variants/libpsnd_$name.so
