#include "psnd_stft_q.h"
bool psnd_stft1024q_ok(long long, long long, int, int) { return false; }
int psnd_stft1024q_launch(const float *, const float *, float *, long long, long long, long long, int, int, float, int, hipStream_t) { return PSND_E_UNSUPPORTED; }
