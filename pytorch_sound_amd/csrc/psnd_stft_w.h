// psnd_stft_w.h - interface between psnd_stft.hip (entry points, plan) and psnd_stft_w.hip (the wave-per-frame 4096 kernel).
#pragma once
#include "psnd_common.h"

// plan(4096) = [win[4096] | wA[256][16] | twA[256][8](re,im) | twB[16][16](re,im) | vk[1025](re,im), padded]   (psnd_stft.hip)
//              | tw[32][32](re,im) = W_1024^(lam q1) | cL[64](re,im) = W_2048^(lane & 31) (-i)^(lane >> 5)        (this kernel)
constexpr int kW4096VkOff = 3 * 4096 + 512;
constexpr int kW4096TwOff = 3 * 4096 + 512 + 2052;
constexpr int kW4096ClOff = kW4096TwOff + 2048;
constexpr int kW4096PlanFloats = kW4096ClOff + 128;

void psnd_stft4096w_plan_fill(float *plan);                      // host: the two tables above
bool psnd_stft4096w_ok(long long T, long long F, int hop, int pad);
int psnd_stft4096w_launch(const float *wav, const float *plan, float *mag, long long N, long long T, long long F, int hop, int pad,
                          float mag_eps, int ablate, int nfk /* 1: output (N, F, K) */, hipStream_t stream);

// psnd_stft_r.hip - hop = 1024, (N, F, K): the same transform fed from a workgroup-shared LDS sample ring (LDS-DMA, every sample fetched once)
bool psnd_stft4096r_ok(long long T, long long F, int hop, int pad);
int psnd_stft4096r_launch(const float *wav, const float *plan, float *mag_nfk, long long N, long long T, long long F, int pad, float mag_eps,
                          int ablate, hipStream_t stream);
