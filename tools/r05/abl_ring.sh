V=tools/mb/variants
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
python tools/r05/time_ring.py
PSND_ABLATE=2 python tools/r05/time_ring.py
PSND_STFT4096_NORING=1 python tools/r05/time_ring.py
PSND_STFT4096_NORING=1 PSND_ABLATE=2 python tools/r05/time_ring.py
for b in 1 2 4 8 16 32 64 3 7 15 31; do PSND_LIB=$V/libpsnd_rabl$b.so python tools/r05/time_ring.py; done
PSND_ABLATE=2 PSND_LIB=$V/libpsnd_rabl31.so python tools/r05/time_ring.py
python tools/r05/time_ring.py
