V=tools/mb/variants
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
timeout 120 python tools/r05/check_ring.py 2>&1 | grep -v amdgpu.ids | head -8
for i in 1 2; do
python tools/r05/time_ring.py
PSND_ABLATE=2 python tools/r05/time_ring.py
PSND_STFT4096_NORING=1 python tools/r05/time_ring.py
done
for b in "$@"; do PSND_LIB=$V/libpsnd_rabl$b.so python tools/r05/time_ring.py; done
