// psnd_pqmf.hip - the pseudo-QMF bank of pytorch_sound/models/transforms.py:492-560 (multi-band vocoders) on gfx950.
// The reference runs analysis as a (subbands, 1, taps+1) conv1d over the zero-padded signal followed by a strided identity
// conv that throws S-1 of every S outputs away, and synthesis as a zero-stuffing transposed conv followed by a dense conv
// - S times the useful arithmetic and two extra round trips each way.  Here both are polyphase:
//   analysis :  out[b][k][m] = scale * sum_j H[k][j] x[b][m S + j - P]                      (P = taps / 2, zeros outside)
//   synthesis:  y[b][t]      = scale * sum_k sum_{j : (t + j - P) % S == 0} G[k][j] x[b][k][(t + j - P) / S]
// and each is the other's adjoint with the filter reversed (flip = 1), which is how the backward passes run.
// HBM-bound: reads T, writes T floats per clip; the signal span of a workgroup and the filters sit in LDS.
#include "psnd_common.h"

namespace {

constexpr int PQ_MAXS = 16, PQ_MAXTAPS = 255;      // subbands, taps (filters have taps + 1 coefficients)

__global__ __launch_bounds__(256) void pqmf_analysis_kernel(const float *x, const float *h, int S, int taps, long long T, long long M,
                                                            int flip, float scale, float *out) {
    extern __shared__ float sm[];
    float *sh = sm;                                   // S * (taps + 1) coefficients
    float *sx = sm + S * (taps + 1);                  // span: 256 * S + taps samples
    const int b = blockIdx.y, nt = taps + 1, P = taps / 2;
    const long long m0 = (long long)blockIdx.x * 256;
    for (int i = threadIdx.x; i < S * nt; i += 256) {
        const int k = i / nt, j = i - k * nt;
        sh[i] = h[k * nt + (flip ? taps - j : j)];
    }
    const long long g0 = m0 * S - P;
    const int span = 256 * S + taps;
    const float *xb = x + (size_t)b * T;
    for (int i = threadIdx.x; i < span; i += 256) {
        const long long g = g0 + i;
        sx[i] = (g >= 0 && g < T) ? xb[g] : 0.f;
    }
    __syncthreads();
    const long long m = m0 + threadIdx.x;
    if (m >= M) return;
    const float *px = sx + threadIdx.x * S;
    for (int k = 0; k < S; ++k) {
        float acc = 0.f;
        const float *ph = sh + k * nt;
        for (int j = 0; j < nt; ++j) acc = __builtin_fmaf(ph[j], px[j], acc);
        out[((size_t)b * S + k) * M + m] = scale * acc;
    }
}

__global__ __launch_bounds__(256) void pqmf_synthesis_kernel(const float *x, const float *g, int S, int taps, long long M, long long T,
                                                             int flip, float scale, float *y) {
    extern __shared__ float sm[];
    float *sg = sm;                                   // S * (taps + 1)
    float *sx = sm + S * (taps + 1);                  // per subband: the input samples this workgroup's outputs touch
    const int b = blockIdx.y, nt = taps + 1, P = taps / 2;
    const long long t0 = (long long)blockIdx.x * 256;
    for (int i = threadIdx.x; i < S * nt; i += 256) {
        const int k = i / nt, j = i - k * nt;
        sg[i] = g[k * nt + (flip ? taps - j : j)];
    }
    // inputs: m in [floor((t0 - P) / S), floor((t0 + 255 + taps - P) / S)]
    long long mlo = t0 - P;
    mlo = mlo >= 0 ? mlo / S : -((-mlo + S - 1) / S);
    const int nm = (256 + taps) / S + 2;
    for (int i = threadIdx.x; i < S * nm; i += 256) {
        const int k = i / nm, q = i - k * nm;
        const long long m = mlo + q;
        sx[i] = (m >= 0 && m < M) ? x[((size_t)b * S + k) * M + m] : 0.f;
    }
    __syncthreads();
    const long long t = t0 + threadIdx.x;
    if (t >= T) return;
    // first tap j0 >= 0 with (t + j0 - P) % S == 0
    long long r = (t - P) % S;
    if (r < 0) r += S;
    const int j0 = (int)((S - r) % S);
    const long long mfirst = (t + j0 - P) >= 0 ? (t + j0 - P) / S : -((-(t + j0 - P) + S - 1) / S);
    const int q0 = (int)(mfirst - mlo);
    float acc = 0.f;
    for (int k = 0; k < S; ++k) {
        const float *pg = sg + k * nt, *px = sx + k * nm + q0;
        int q = 0;
        for (int j = j0; j < nt; j += S, ++q) acc = __builtin_fmaf(pg[j], px[q], acc);
    }
    y[(size_t)b * T + t] = scale * acc;
}

}  // namespace

extern "C" int psnd_pqmf_analysis(const float *x, const float *filt, int64_t B, int64_t T, int subbands, int taps, int flip, float scale,
                                  float *out, void *stream) {
    if (!x || !filt || !out) PSND_FAIL(PSND_E_ARG, "pqmf_analysis: null pointer");
    if (subbands < 1 || subbands > PQ_MAXS || taps < 2 || taps > PQ_MAXTAPS || (taps & 1)) PSND_FAIL(PSND_E_SHAPE, "pqmf_analysis: subbands=%d taps=%d", subbands, taps);
    if (B < 0 || B > 65535 || T < 0) PSND_FAIL(PSND_E_SHAPE, "pqmf_analysis: B=%lld T=%lld", (long long)B, (long long)T);
    const int64_t M = T / subbands;
    if (B == 0 || M == 0) return PSND_OK;
    const size_t lds = sizeof(float) * ((size_t)subbands * (taps + 1) + 256 * (size_t)subbands + taps);
    hipLaunchKernelGGL(pqmf_analysis_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)B), dim3(256), lds, static_cast<hipStream_t>(stream),
                       x, filt, subbands, taps, (long long)T, (long long)M, flip, scale, out);
    PSND_CHECK_LAUNCH("pqmf_analysis");
    return PSND_OK;
}

extern "C" int psnd_pqmf_synthesis(const float *x, const float *filt, int64_t B, int64_t M, int64_t T_out, int subbands, int taps, int flip,
                                   float scale, float *y, void *stream) {
    if (!x || !filt || !y) PSND_FAIL(PSND_E_ARG, "pqmf_synthesis: null pointer");
    if (subbands < 1 || subbands > PQ_MAXS || taps < 2 || taps > PQ_MAXTAPS || (taps & 1)) PSND_FAIL(PSND_E_SHAPE, "pqmf_synthesis: subbands=%d taps=%d", subbands, taps);
    if (B < 0 || B > 65535 || M < 0 || T_out < 0) PSND_FAIL(PSND_E_SHAPE, "pqmf_synthesis: B=%lld M=%lld T_out=%lld", (long long)B, (long long)M, (long long)T_out);
    if (B == 0 || T_out == 0) return PSND_OK;
    const int64_t T = T_out;
    const size_t lds = sizeof(float) * ((size_t)subbands * (taps + 1) + (size_t)subbands * ((256 + taps) / subbands + 2));
    hipLaunchKernelGGL(pqmf_synthesis_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)B), dim3(256), lds, static_cast<hipStream_t>(stream),
                       x, filt, subbands, taps, (long long)M, (long long)T, flip, scale, y);
    PSND_CHECK_LAUNCH("pqmf_synthesis");
    return PSND_OK;
}
