// psnd_stft_q.h - interface between psnd_stft.hip (entry points, plan) and psnd_stft_q.hip (n_fft = 1024, bin-fastest output:
// a wave owns four frames).
#pragma once
#include "psnd_common.h"

bool psnd_stft1024q_ok(long long T, long long F, int hop, int pad);
int psnd_stft1024q_launch(const float *wav, const float *plan, float *mag_nfk, long long N, long long T, long long F, int hop, int pad,
                          float mag_eps, int ablate, hipStream_t stream);
