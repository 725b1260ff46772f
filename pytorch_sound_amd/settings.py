"""Default audio / STFT parameters.

Values of pytorch_sound/settings.py:9-32 of the reference (the text-vocabulary half of that file
is outside the accelerated path).  Change them here, as the reference's README instructs.
"""
import multiprocessing

# --- audio / STFT ---
SAMPLE_RATE: int = 22050
N_FFT: int = 1024
WIN_LENGTH: int = 1024
HOP_LENGTH: int = 256
HOP_STRIDE: int = WIN_LENGTH // HOP_LENGTH      # frames overlapping one window
SPEC_SIZE: int = WIN_LENGTH // 2 + 1            # one-sided spectrum bins
MEL_SIZE: int = 80
MFCC_SIZE: int = 40
MEL_MIN: int = 0                                # Hz
MEL_MAX: int = 8000                             # Hz
MIN_DB: int = -50
MAX_DB: int = 30
VN_DB: float = -11.5                            # volume-normalisation target
MULAW_BINS: int = 256

# --- clip-length filters used by the data side (seconds, multiplied by the sample rate) ---
MIN_WAV_RATE: int = 2
MAX_WAV_RATE: int = 15
MIN_TXT_RATE: float = 0

NUM_WORKERS: int = multiprocessing.cpu_count() // 2
