// psnd_loss.hip - pytorch_sound/models/sound.py on gfx950: PreEmphasis (sound.py:66-81) and the reductions of
// multi_stft_loss (sound.py:106-133).  All HBM-bound fp32 streams; the STFT magnitudes themselves come from
// psnd_stft_fwd / go back through psnd_stft_bwd.
//
// multi_stft_loss per resolution i on magnitudes p, t of shape (N, K, F):
//     sc_i  = mean_n ||t_n - p_n||_F / ||t_n||_F                         (sound.py:123)
//     mag_i = mean_n sum |log(t_n + eps) - log(p_n + eps)| / (K F)       (sound.py:124)
//     returns (sum_i (sc_i + mag_i) / L, sum_i sc_i / L, sum_i mag_i / L)
// The reference spends ~12 elementwise / reduction launches per resolution on this (each a full HBM round trip over
// both magnitudes).  Here: ONE pass per resolution reads p and t once and leaves three partial sums per workgroup
// (double, no atomics, fixed summation order => bit-reproducible), one single-workgroup launch combines every
// resolution into the three scalars, and ONE pass per resolution writes the gradient wrt both magnitudes.
#include "psnd_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

constexpr int LB = 256;                     // threads per workgroup
constexpr int LCHUNK = 8192;                // elements of one clip handled by one workgroup (32 KB of each operand)

// part[(n * B + b) * 3 + {0,1,2}] = sum (t-p)^2, sum t^2, sum |log(t+eps) - log(p+eps)|  over chunk b of clip n
__global__ __launch_bounds__(LB) void loss_partial_kernel(const float *p, const float *t, long long KF, float eps, double *part) {
    const int b = blockIdx.x, n = blockIdx.y, B = gridDim.x;
    const long long e0 = (long long)b * LCHUNK, e1 = min(e0 + LCHUNK, KF);
    const float *pp = p + (size_t)n * KF, *tt = t + (size_t)n * KF;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    // rows of different clips start at arbitrary 4-byte offsets (KF is odd for every power-of-two n_fft): peel to
    // a 16-byte boundary of the CLIP pointer, then float4 loads
    const long long mis = (4 - (((uintptr_t)(pp + e0) >> 2) & 3)) & 3;
    const bool same = ((((uintptr_t)(pp + e0)) ^ ((uintptr_t)(tt + e0))) & 15) == 0;
    auto term = [&](float pv, float tv) __attribute__((always_inline)) {
        const float d = tv - pv;
        s1 = __builtin_fmaf(d, d, s1);
        s2 = __builtin_fmaf(tv, tv, s2);
        s3 += fabsf(__logf(tv + eps) - __logf(pv + eps));
    };
    if (same) {
        const long long a0 = min(e0 + mis, e1);
        for (long long e = e0 + threadIdx.x; e < a0; e += LB) term(pp[e], tt[e]);
        const long long nv = (e1 - a0) >> 2;
        const float4 *p4 = reinterpret_cast<const float4 *>(pp + a0), *t4 = reinterpret_cast<const float4 *>(tt + a0);
        for (long long v = threadIdx.x; v < nv; v += LB) {
            const float4 a = p4[v], c = t4[v];
            term(a.x, c.x), term(a.y, c.y), term(a.z, c.z), term(a.w, c.w);
        }
        for (long long e = a0 + 4 * nv + threadIdx.x; e < e1; e += LB) term(pp[e], tt[e]);
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += LB) term(pp[e], tt[e]);
    }
    const double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2), d3 = wave_sum_d((double)s3);
    __shared__ double red[3 * (LB / 64)];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = d1, red[LB / 64 + w] = d2, red[2 * (LB / 64) + w] = d3;
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int i = 0; i < LB / 64; ++i) s += red[threadIdx.x * (LB / 64) + i];
        part[((size_t)n * B + b) * 3 + threadIdx.x] = s;
    }
}

constexpr int MAXRES = 8;
struct FinalArgs {
    const double *part[MAXRES];
    int B[MAXRES];
    double inv_kf[MAXRES];
    int L, N;
};

// one workgroup: norms[(i * N + n) * 2 + {0,1}] = ||t - p||, ||t||;  out = {loss, sc_loss, mag_loss}
__global__ __launch_bounds__(256) void loss_final_kernel(FinalArgs a, float *norms, float *out) {
    __shared__ double red[2][4];
    double sc_all = 0, mag_all = 0;
    for (int i = 0; i < a.L; ++i) {
        double sc = 0, mg = 0;
        for (int n = threadIdx.x; n < a.N; n += 256) {
            const double *q = a.part[i] + (size_t)n * a.B[i] * 3;
            double s1 = 0, s2 = 0, s3 = 0;
            for (int b = 0; b < a.B[i]; ++b) s1 += q[3 * b], s2 += q[3 * b + 1], s3 += q[3 * b + 2];
            const double nd = sqrt(s1), nt = sqrt(s2);
            norms[((size_t)i * a.N + n) * 2] = (float)nd;
            norms[((size_t)i * a.N + n) * 2 + 1] = (float)nt;
            sc += nd / nt;
            mg += s3;
        }
        sc = wave_sum_d(sc), mg = wave_sum_d(mg);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = sc, red[1][threadIdx.x >> 6] = mg;
        __syncthreads();
        sc_all += (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / a.N;
        mag_all += (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / a.N * a.inv_kf[i];
    }
    if (threadIdx.x == 0) {
        out[0] = (float)((sc_all + mag_all) / a.L);
        out[1] = (float)(sc_all / a.L);
        out[2] = (float)(mag_all / a.L);
    }
}

// gp = c_sc (p - t) / (||d|| ||t||) + c_mag sign(log(p+eps) - log(t+eps)) / (p + eps)
// gt = c_sc (-(p - t) / (||d|| ||t||) - ||d|| t / ||t||^3) - c_mag sign(..) / (t + eps)
// c_sc = (g[0] + g[1]) / (L N),  c_mag = (g[0] + g[2]) / (L N K F);  g = upstream gradient of (loss, sc_loss, mag_loss)
__global__ __launch_bounds__(LB) void loss_bwd_kernel(const float *p, const float *t, long long KF, float eps, const float *norms,
                                                      const float *g, float invLN, float invLNKF, float *gp, float *gt) {
    const int n = blockIdx.y;
    const float nd = norms[2 * n], nt = norms[2 * n + 1];
    const float c_sc = (g[0] + g[1]) * invLN, c_mag = (g[0] + g[2]) * invLNKF;
    const float k1 = c_sc / (nd * nt), k2 = c_sc * nd / (nt * nt * nt);
    const size_t base = (size_t)n * KF;
    const long long e0 = (long long)blockIdx.x * LCHUNK, e1 = min(e0 + LCHUNK, KF);
    for (long long e = e0 + threadIdx.x; e < e1; e += LB) {
        const float pv = p[base + e], tv = t[base + e];
        const float d = pv - tv;
        const float lp = __logf(pv + eps) - __logf(tv + eps);
        const float sg = lp > 0.f ? c_mag : (lp < 0.f ? -c_mag : 0.f);
        if (gp) gp[base + e] = k1 * d + sg / (pv + eps);
        if (gt) gt[base + e] = -k1 * d - k2 * tv - sg / (tv + eps);
    }
}

// y[n][t] = x[n][t] - coef * x[n][t-1],  x[n][-1] := x[n][1]   (F.pad(input, (1, 0), 'reflect') + conv1d [-coef, 1])
__global__ __launch_bounds__(256) void preemph_fwd_kernel(const float *x, long long T, float coef, float *y) {
    const size_t base = (size_t)blockIdx.y * T;
    const long long t = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long tt = t + 256 * u;
        if (tt < T) y[base + tt] = x[base + tt] - coef * x[base + (tt == 0 ? 1 : tt - 1)];
    }
}
// adjoint: gx[t] = gy[t] - coef * gy[t+1] (t + 1 < T);  gx[1] -= coef * gy[0]
__global__ __launch_bounds__(256) void preemph_bwd_kernel(const float *gy, long long T, float coef, float *gx) {
    const size_t base = (size_t)blockIdx.y * T;
    const long long t = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long tt = t + 256 * u;
        if (tt < T) {
            float v = gy[base + tt];
            if (tt + 1 < T) v -= coef * gy[base + tt + 1];
            if (tt == 1) v -= coef * gy[base];
            gx[base + tt] = v;
        }
    }
}

// ---- mean absolute error (F.l1_loss, reduction 'mean') as two launches forward, one backward ---------------------------
constexpr int L1CH = 16384;             // elements per workgroup
__global__ __launch_bounds__(256) void l1_partial_kernel(const float *a, const float *b, long long n, double *part) {
    const long long e0 = (long long)blockIdx.x * L1CH, e1 = min(e0 + L1CH, n);
    float s = 0.f;
    if ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) {
        const long long v1 = e0 + ((e1 - e0) & ~3ll);
        for (long long e = e0 + 4 * threadIdx.x; e < v1; e += 1024) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(a + e), y = *reinterpret_cast<const f32x4 *>(b + e);
            s += fabsf(x.x - y.x) + fabsf(x.y - y.y) + fabsf(x.z - y.z) + fabsf(x.w - y.w);
        }
        for (long long e = v1 + threadIdx.x; e < e1; e += 256) s += fabsf(a[e] - b[e]);
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) s += fabsf(a[e] - b[e]);
    }
    const double d = wave_sum_d((double)s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void l1_final_kernel(const double *part, int nb, double inv_n, float *out) {
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    s = wave_sum_d(s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) * inv_n);
}
// out[0] = sum_i w_i * mean_i: the partial sums of up to 4 L1 terms folded by one workgroup (a recipe's `l1(a, b) + 0.5 * l1(c, d)` without
// the scalar multiply / add launches in between)
struct L1Terms {
    const double *part[4];
    int nb[4];
    double scale[4];            // w_i / n_i
    int terms;
};
// one workgroup of 1024 threads, four partials per thread in flight (256 threads walking 3264 + 480 partials one load at a time were
// fifteen dependent L2 round trips: 6.4 us for a kernel that adds 30 KB)
__global__ __launch_bounds__(1024) void l1_final_multi_kernel(L1Terms t, float *out, float *nan_flag) {
    double s = 0;
    for (int k = 0; k < t.terms; ++k) {
        double sk = 0;
        const int nb = t.nb[k];
        for (int i = threadIdx.x; i < nb; i += 4096) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = t.part[k][min(i + 1024 * u, nb - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) sk += i + 1024 * u < nb ? v[u] : 0.0;
        }
        s += sk * t.scale[k];
    }
    s = wave_sum_d(s);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0;
        for (int w = 0; w < 16; ++w) a += red[w];
        out[0] = (float)a;
        if (nan_flag) nan_flag[0] = a != a ? 1.f : 0.f;       // Trainer's `loss != loss` (trainer.py:205), formed where the loss is
    }
}
// ga = g * sign(a - b) / n,  gb = -ga   (either may be NULL); g: device scalar
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float *a, const float *b, long long n, const float *g, float inv_n, float *ga,
                                                     float *gb) {
    const float c = g[0] * inv_n;
    for (long long e = (long long)blockIdx.x * 1024 + threadIdx.x; e < min((long long)(blockIdx.x + 1) * 1024, n); e += 256) {
        const float d = a[e] - b[e];
        const float v = d > 0.f ? c : (d < 0.f ? -c : 0.f);
        if (ga) ga[e] = v;
        if (gb) gb[e] = -v;
    }
}

}  // namespace

extern "C" int64_t psnd_stft_loss_blocks(int64_t KF) { return KF <= 0 ? 0 : (KF + LCHUNK - 1) / LCHUNK; }

extern "C" int psnd_stft_loss_partial(const float *p_mag, const float *t_mag, int64_t N, int64_t KF, float eps, double *part,
                                      void *stream) {
    if (!p_mag || !t_mag || !part) PSND_FAIL(PSND_E_ARG, "stft_loss_partial: null pointer");
    if (N < 0 || KF <= 0 || N > 65535) PSND_FAIL(PSND_E_SHAPE, "stft_loss_partial: N=%lld KF=%lld", (long long)N, (long long)KF);
    if (N == 0) return PSND_OK;
    const int64_t B = psnd_stft_loss_blocks(KF);
    if (B > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "stft_loss_partial: KF=%lld too large", (long long)KF);
    hipLaunchKernelGGL(loss_partial_kernel, dim3((unsigned)B, (unsigned)N), dim3(LB), 0, static_cast<hipStream_t>(stream), p_mag, t_mag,
                       (long long)KF, eps, part);
    PSND_CHECK_LAUNCH("stft_loss_partial");
    return PSND_OK;
}

// blocks[i] = partial-sum entries per clip of resolution i (psnd_stft_loss_blocks(KF[i]) for psnd_stft_loss_partial,
// psnd_stft_fwd_msl_blocks for the sums the STFT kernel leaves); blocks == NULL: all from psnd_stft_loss_partial
extern "C" int psnd_stft_loss_final_blocks(const double *const *parts, const int64_t *KF, const int64_t *blocks, int L, int64_t N,
                                           float *norms, float *out3, void *stream) {
    if (!parts || !KF || !norms || !out3) PSND_FAIL(PSND_E_ARG, "stft_loss_final: null pointer");
    if (L <= 0 || L > MAXRES) PSND_FAIL(PSND_E_ARG, "stft_loss_final: %d resolutions (1..%d)", L, MAXRES);
    if (N <= 0 || N > 65535) PSND_FAIL(PSND_E_SHAPE, "stft_loss_final: N=%lld", (long long)N);
    FinalArgs a;
    a.L = L, a.N = (int)N;
    for (int i = 0; i < L; ++i) {
        if (!parts[i] || KF[i] <= 0) PSND_FAIL(PSND_E_ARG, "stft_loss_final: resolution %d: null partials / KF=%lld", i, (long long)KF[i]);
        const int64_t B = blocks ? blocks[i] : psnd_stft_loss_blocks(KF[i]);
        if (B <= 0 || B > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "stft_loss_final: resolution %d: %lld partial sums per clip", i, (long long)B);
        a.part[i] = parts[i];
        a.B[i] = (int)B;
        a.inv_kf[i] = 1.0 / (double)KF[i];
    }
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a, norms, out3);
    PSND_CHECK_LAUNCH("stft_loss_final");
    return PSND_OK;
}

extern "C" int psnd_stft_loss_final(const double *const *parts, const int64_t *KF, int L, int64_t N, float *norms, float *out3,
                                    void *stream) {
    return psnd_stft_loss_final_blocks(parts, KF, nullptr, L, N, norms, out3, stream);
}

extern "C" int psnd_stft_loss_bwd(const float *p_mag, const float *t_mag, int64_t N, int64_t KF, float eps, const float *norms,
                                  const float *g3, int L, float *gp, float *gt, void *stream) {
    if (!p_mag || !t_mag || !norms || !g3 || (!gp && !gt)) PSND_FAIL(PSND_E_ARG, "stft_loss_bwd: null pointer");
    if (N < 0 || KF <= 0 || N > 65535 || L <= 0) PSND_FAIL(PSND_E_SHAPE, "stft_loss_bwd: N=%lld KF=%lld L=%d", (long long)N, (long long)KF, L);
    if (N == 0) return PSND_OK;
    const int64_t B = psnd_stft_loss_blocks(KF);
    const double ln = (double)L * (double)N;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)B, (unsigned)N), dim3(LB), 0, static_cast<hipStream_t>(stream), p_mag, t_mag,
                       (long long)KF, eps, norms, g3, (float)(1.0 / ln), (float)(1.0 / (ln * (double)KF)), gp, gt);
    PSND_CHECK_LAUNCH("stft_loss_bwd");
    return PSND_OK;
}

extern "C" int psnd_preemphasis_fwd(const float *x, int64_t N, int64_t T, float coef, float *y, void *stream) {
    if (!x || !y) PSND_FAIL(PSND_E_ARG, "preemphasis_fwd: null pointer");
    if (N < 0 || N > 65535) PSND_FAIL(PSND_E_SHAPE, "preemphasis_fwd: N=%lld", (long long)N);
    if (T < 2) PSND_FAIL(PSND_E_SHAPE, "preemphasis_fwd: reflect padding of 1 needs T >= 2 (T=%lld)", (long long)T);
    if (N == 0) return PSND_OK;
    hipLaunchKernelGGL(preemph_fwd_kernel, dim3((unsigned)((T + 1023) / 1024), (unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, (long long)T, coef, y);
    PSND_CHECK_LAUNCH("preemphasis_fwd");
    return PSND_OK;
}

extern "C" int psnd_preemphasis_bwd(const float *gy, int64_t N, int64_t T, float coef, float *gx, void *stream) {
    if (!gy || !gx) PSND_FAIL(PSND_E_ARG, "preemphasis_bwd: null pointer");
    if (N < 0 || N > 65535 || T < 2) PSND_FAIL(PSND_E_SHAPE, "preemphasis_bwd: N=%lld T=%lld", (long long)N, (long long)T);
    if (N == 0) return PSND_OK;
    hipLaunchKernelGGL(preemph_bwd_kernel, dim3((unsigned)((T + 1023) / 1024), (unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       gy, (long long)T, coef, gx);
    PSND_CHECK_LAUNCH("preemphasis_bwd");
    return PSND_OK;
}

extern "C" int64_t psnd_l1_loss_blocks(int64_t n) { return n <= 0 ? 0 : (n + L1CH - 1) / L1CH; }

extern "C" int psnd_l1_loss_fwd(const float *a, const float *b, int64_t n, double *part, float *out, void *stream) {
    if (!a || !b || !part || !out) PSND_FAIL(PSND_E_ARG, "l1_loss_fwd: null pointer");
    if (n <= 0 || psnd_l1_loss_blocks(n) > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "l1_loss_fwd: n=%lld", (long long)n);
    const int nb = (int)psnd_l1_loss_blocks(n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, (long long)n, part);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, s, part, nb, 1.0 / (double)n, out);
    PSND_CHECK_LAUNCH("l1_loss_fwd");
    return PSND_OK;
}

extern "C" int psnd_l1_loss_bwd(const float *a, const float *b, int64_t n, const float *g, float *ga, float *gb, void *stream) {
    return psnd_l1_loss_bwd_w(a, b, n, g, 1.0, ga, gb, stream);
}

extern "C" int psnd_l1_loss_bwd_w(const float *a, const float *b, int64_t n, const float *g, double weight, float *ga, float *gb, void *stream) {
    if (!a || !b || !g || (!ga && !gb)) PSND_FAIL(PSND_E_ARG, "l1_loss_bwd: null pointer");
    if (n <= 0 || (n + 1023) / 1024 > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "l1_loss_bwd: n=%lld", (long long)n);
    hipLaunchKernelGGL(l1_bwd_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, (long long)n, g,
                       (float)(weight / (double)n), ga, gb);
    PSND_CHECK_LAUNCH("l1_loss_bwd");
    return PSND_OK;
}

// ---- masked L1 over (N, C, T) tensors with a per-frame weight m (N, T) - the padded-batch recipes' loss (variable-length clips,
//      data/dataset.py:196-250 pads to the batch maximum): out = sum_{n,c,t} |a - b| m[n,t] / (C sum m).  Forward: one partial pair
//      (sum |a - b| m, sum m of the block's frames for c = 0) per workgroup, one combine; backward ga = g sign(a - b) m / (C sum m).
namespace {
constexpr int ML1CH = 4096;      // elements per workgroup
__global__ __launch_bounds__(256) void masked_l1_partial_kernel(const float *a, const float *b, const float *m, long long CT, int T, long long n,
                                                                 double *part) {
    const long long e0 = (long long)blockIdx.x * ML1CH;
    float acc = 0.f, macc = 0.f;
    for (long long e = e0 + threadIdx.x; e < e0 + ML1CH && e < n; e += 256) {
        const long long clip = e / CT, r = e - clip * CT;
        const int t = (int)(r % T);
        const float w = m[clip * T + t];
        acc += __builtin_fabsf(a[e] - b[e]) * w;
        if (r < T) macc += w;                                  // every frame's weight once (channel 0)
    }
    double d = (double)acc, dm = (double)macc;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) d += __shfl_xor(d, k, 64), dm += __shfl_xor(dm, k, 64);
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d, red[4 + (threadIdx.x >> 6)] = dm;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * (size_t)blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        part[2 * (size_t)blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
    }
}
__global__ __launch_bounds__(256) void masked_l1_final_kernel(const double *part, int nb, int C, float *out, float *inv_den) {
    double s = 0.0, sm = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[2 * i], sm += part[2 * i + 1];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k, 64), sm += __shfl_xor(sm, k, 64);
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s, red[4 + (threadIdx.x >> 6)] = sm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double den = (red[4] + red[5] + red[6] + red[7]) * (double)C;
        out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / den);
        inv_den[0] = (float)(1.0 / den);
    }
}
__global__ __launch_bounds__(256) void masked_l1_bwd_kernel(const float *a, const float *b, const float *m, long long CT, int T, long long n,
                                                             const float *g, const float *inv_den, float *ga, float *gb) {
    const float sc = g[0] * inv_den[0];
    for (long long e = (long long)blockIdx.x * 1024 + threadIdx.x; e < (long long)(blockIdx.x + 1) * 1024 && e < n; e += 256) {
        const long long clip = e / CT, r = e - clip * CT;
        const float d = a[e] - b[e];
        const float v = sc * m[clip * T + (int)(r % T)] * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        if (ga) ga[e] = v;
        if (gb) gb[e] = -v;
    }
}
}  // namespace
extern "C" int64_t psnd_masked_l1_blocks(int64_t n) { return n <= 0 ? 0 : (n + ML1CH - 1) / ML1CH; }
extern "C" int psnd_masked_l1_fwd(const float *a, const float *b, const float *frame_weight, int64_t N, int C, int64_t T, double *part,
                                  float *out, float *inv_den, void *stream) {
    if (!a || !b || !frame_weight || !part || !out || !inv_den) PSND_FAIL(PSND_E_ARG, "masked_l1_fwd: null pointer");
    const int64_t n = N * C * T;
    if (N <= 0 || C <= 0 || T <= 0 || T > 0x7fffffff || psnd_masked_l1_blocks(n) > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "masked_l1_fwd: N=%lld C=%d T=%lld", (long long)N, C, (long long)T);
    const int nb = (int)psnd_masked_l1_blocks(n);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(masked_l1_partial_kernel, dim3(nb), dim3(256), 0, s, a, b, frame_weight, (long long)C * T, (int)T, (long long)n, part);
    hipLaunchKernelGGL(masked_l1_final_kernel, dim3(1), dim3(256), 0, s, part, nb, C, out, inv_den);
    PSND_CHECK_LAUNCH("masked_l1_fwd");
    return PSND_OK;
}
extern "C" int psnd_masked_l1_bwd(const float *a, const float *b, const float *frame_weight, int64_t N, int C, int64_t T, const float *g,
                                  const float *inv_den, float *ga, float *gb, void *stream) {
    if (!a || !b || !frame_weight || !g || !inv_den || (!ga && !gb)) PSND_FAIL(PSND_E_ARG, "masked_l1_bwd: null pointer");
    const int64_t n = N * C * T;
    if (N <= 0 || C <= 0 || T <= 0 || T > 0x7fffffff || (n + 1023) / 1024 > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "masked_l1_bwd: N=%lld C=%d T=%lld", (long long)N, C, (long long)T);
    hipLaunchKernelGGL(masked_l1_bwd_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, frame_weight,
                       (long long)C * T, (int)T, (long long)n, g, inv_den, ga, gb);
    PSND_CHECK_LAUNCH("masked_l1_bwd");
    return PSND_OK;
}

// flag[0] = 1 if any of x[0 .. n) is NaN else 0 (one workgroup; n is a loss value or a handful of them): the trainer's device-side
// "loss != loss" (trainer.py:205) as ONE launch instead of torch.isnan + a dtype cast
namespace {
__global__ __launch_bounds__(64) void nan_flag_kernel(const float *x, long long n, float *flag) {
    bool bad = false;
    for (long long i = threadIdx.x; i < n; i += 64) bad |= x[i] != x[i];
    const unsigned long long any = __builtin_amdgcn_ballot_w64(bad);
    if (threadIdx.x == 0) flag[0] = any ? 1.f : 0.f;
}
}  // namespace
extern "C" int psnd_nan_flag(const float *x, int64_t n, float *flag, void *stream) {
    if (!x || !flag) PSND_FAIL(PSND_E_ARG, "nan_flag: null pointer");
    if (n < 0) PSND_FAIL(PSND_E_SHAPE, "nan_flag: n=%lld", (long long)n);
    hipLaunchKernelGGL(nan_flag_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), x, (long long)n, flag);
    PSND_CHECK_LAUNCH("nan_flag");
    return PSND_OK;
}

// out[0] = sum_i scale[i] * sum(parts[i][0 .. nb[i])): folds the partial sums that fused producers leave (psnd_mask_head_l1_fwd,
// psnd_mel_l1_fwd, psnd_l1_loss_fwd's own partials) into one loss value; scale[i] = weight_i / numel_i; 1 .. 4 terms
extern "C" int psnd_l1_loss_combine(const double *const *parts, const int64_t *nb, const double *scale, int terms, float *out, float *nan_flag, void *stream) {
    if (!parts || !nb || !scale || !out) PSND_FAIL(PSND_E_ARG, "l1_loss_combine: null pointer");
    if (terms < 1 || terms > 4) PSND_FAIL(PSND_E_SHAPE, "l1_loss_combine: 1 .. 4 terms, got %d", terms);
    L1Terms t;
    t.terms = terms;
    for (int k = 0; k < terms; ++k) {
        if (!parts[k] || nb[k] <= 0 || nb[k] > 0x7fffffff) PSND_FAIL(PSND_E_ARG, "l1_loss_combine: term %d: null partials / nb=%lld", k, (long long)nb[k]);
        t.part[k] = parts[k], t.nb[k] = (int)nb[k], t.scale[k] = scale[k];
    }
    hipLaunchKernelGGL(l1_final_multi_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), t, out, nan_flag);
    PSND_CHECK_LAUNCH("l1_loss_combine");
    return PSND_OK;
}

extern "C" int psnd_l1_loss_sum_fwd(const float *const *a, const float *const *b, const int64_t *n, const double *w, int terms, double *part,
                                    float *out, void *stream) {
    if (!a || !b || !n || !w || !part || !out) PSND_FAIL(PSND_E_ARG, "l1_loss_sum_fwd: null pointer");
    if (terms < 1 || terms > 4) PSND_FAIL(PSND_E_SHAPE, "l1_loss_sum_fwd: 1 .. 4 terms, got %d", terms);
    hipStream_t s = static_cast<hipStream_t>(stream);
    L1Terms t;
    t.terms = terms;
    double *pp = part;
    for (int k = 0; k < 4; ++k) t.part[k] = nullptr, t.nb[k] = 0, t.scale[k] = 0.0;
    for (int k = 0; k < terms; ++k) {
        if (!a[k] || !b[k]) PSND_FAIL(PSND_E_ARG, "l1_loss_sum_fwd: null tensor in term %d", k);
        if (n[k] <= 0 || psnd_l1_loss_blocks(n[k]) > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "l1_loss_sum_fwd: n=%lld", (long long)n[k]);
        const int nb = (int)psnd_l1_loss_blocks(n[k]);
        hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, s, a[k], b[k], (long long)n[k], pp);
        t.part[k] = pp, t.nb[k] = nb, t.scale[k] = w[k] / (double)n[k];
        pp += nb;
    }
    hipLaunchKernelGGL(l1_final_multi_kernel, dim3(1), dim3(1024), 0, s, t, out, static_cast<float *>(nullptr));
    PSND_CHECK_LAUNCH("l1_loss_sum_fwd");
    return PSND_OK;
}
