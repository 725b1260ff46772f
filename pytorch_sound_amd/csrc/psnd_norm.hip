// psnd_norm.hip - the non-GEMM parts of pytorch_sound/models/modules.py on gfx950:
//   * GroupNorm(1, C)(x + residual) [+ ReLU]  (modules.py:30,58 / :98,114-116): statistics over (C x T) jointly per
//     sample, per-channel affine - forward and backward;
//   * the masked softmax over KEYS of MultiHeadAttention.scale_dot_att (modules.py:66-76): scale, -inf on padded
//     key rows, softmax along dim 1 of the (B, T_key, T_query) score tensor, zero on padded query columns -
//     forward (in place) and backward.
// All HBM-bound, fp32, (N, C, T) / (B, T, T) layouts with the last axis contiguous.
#include "psnd_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

// rows are walked with 16-byte accesses when their length and every pointer involved allow it
__device__ __forceinline__ bool row_is_vec4(const void *a, const void *b, const void *c, long long T) {
    return (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// GroupNorm(1, C): one workgroup per (channel row chunk, sample); rows are contiguous T floats.
// pass A: ws[n][c] = {sum, sum of squares} of s = x + res over the row (double: E[s^2] - mean^2 is formed in double).  One pair PER ROW,
// written, not accumulated (round 6): no atomics, no memset launch in front, and every reader adds the C pairs of a sample up in the same
// order - the statistics are the same bits from run to run.  (Rounds 3-5 accumulated into 16 slots per sample with double atomics.)
// pass B: y = (s - mean) * rstd * gamma[c] + beta[c] (ReLU optional); stats[n] = {mean, rstd}
// ------------------------------------------------------------------------------------------------------------
// the C pairs of sample n, optionally weighted by w[c], added up by the whole workgroup in a fixed order (the same in every workgroup)
__device__ __forceinline__ void gn_ws_sum(const double *ws, const float *w, int n, int C, double &a, double &b) {
    double pa = 0.0, pb = 0.0;
    for (int i = threadIdx.x; i < C; i += 256) {
        const double wi = w ? (double)w[i] : 1.0;
        pa += wi * ws[2 * ((size_t)n * C + i)], pb += wi * ws[2 * ((size_t)n * C + i) + 1];
    }
    pa = wave_sum(pa), pb = wave_sum(pb);
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pa, red[4 + (threadIdx.x >> 6)] = pb;
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]), b = (red[4] + red[5]) + (red[6] + red[7]);
}
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *x, const float *res, int C, long long T, double *ws) {
    const int c = blockIdx.x, n = blockIdx.y;
    const size_t base = ((size_t)n * C + c) * T;
    float s1 = 0.f, s2 = 0.f;
    if (row_is_vec4(x + base, res ? res + base : nullptr, nullptr, T)) {           // 16-byte accesses (rows of T % 4 == 0 floats)
        const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x + base), *r4 = res ? reinterpret_cast<const f32x4 *>(res + base) : nullptr;
        for (long long t = threadIdx.x; t < (T >> 2); t += 256) {
            f32x4 v = x4[t];
            if (r4) v += r4[t];
#pragma unroll
            for (int j = 0; j < 4; ++j) s1 += v[j], s2 = __builtin_fmaf(v[j], v[j], s2);
        }
    } else {
        for (long long t = threadIdx.x; t < T; t += 256) {
            float v = x[base + t];
            if (res) v += res[base + t];
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);
        }
    }
    double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d1, red[4 + (threadIdx.x >> 6)] = d2;
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[2 * ((size_t)n * C + c)] = red[0] + red[1] + red[2] + red[3];
        ws[2 * ((size_t)n * C + c) + 1] = red[4] + red[5] + red[6] + red[7];
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float *x, const float *res, const float *gamma, const float *beta,
                                                       int C, long long T, float eps, int relu, const double *ws, float *y,
                                                       float *stats) {
    const int c = blockIdx.x, n = blockIdx.y;
    const double M = (double)C * (double)T;
    double w1, w2;
    gn_ws_sum(ws, nullptr, n, C, w1, w2);
    const double mean = w1 / M;
    double var = w2 / M - mean * mean;
    if (var < 0) var = 0;
    const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (c == 0 && threadIdx.x == 0) stats[2 * n] = mu, stats[2 * n + 1] = rstd;
    const float g = gamma[c] * rstd, b = beta[c] - mu * g;
    const size_t base = ((size_t)n * C + c) * T;
    if (row_is_vec4(x + base, res ? res + base : nullptr, y + base, T)) {
        const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x + base), *r4 = res ? reinterpret_cast<const f32x4 *>(res + base) : nullptr;
        f32x4 *y4 = reinterpret_cast<f32x4 *>(y + base);
        for (long long t = threadIdx.x; t < (T >> 2); t += 256) {
            f32x4 v = x4[t];
            if (r4) v += r4[t];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = __builtin_fmaf(v[j], g, b);
                if (relu) v[j] = fmaxf(v[j], 0.f);
            }
            y4[t] = v;
        }
        return;
    }
    for (long long t = threadIdx.x; t < T; t += 256) {
        float v = x[base + t];
        if (res) v += res[base + t];
        v = __builtin_fmaf(v, g, b);
        if (relu) v = fmaxf(v, 0.f);
        y[base + t] = v;
    }
}

// backward pass A: per row (n, c):  a = sum_t gy',  b = sum_t gy' * xhat   (gy' = gy * [y > 0] under ReLU)
//   ws[n][c] = {a, b}  (written per row; pass B adds gamma[c] * {a, b} up per sample, and its workgroups of sample 0 add the rows of all
//   samples up into gbeta[c] / ggamma[c] - no atomics, no zeroing launch: round 6)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float *gy, const float *x, const float *res, const float *gamma,
                                                            const float *y, const float *stats, int C, long long T, int relu,
                                                            double *ws, float *ggamma, float *gbeta) {
    const int c = blockIdx.x, n = blockIdx.y;
    const float mu = stats[2 * n], rstd = stats[2 * n + 1];
    const size_t base = ((size_t)n * C + c) * T;
    float a = 0.f, b = 0.f;
    if (row_is_vec4(gy + base, x + base, res ? res + base : nullptr, T) && row_is_vec4(relu ? y + base : nullptr, nullptr, nullptr, T)) {
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gy + base), *x4 = reinterpret_cast<const f32x4 *>(x + base);
        const f32x4 *r4 = res ? reinterpret_cast<const f32x4 *>(res + base) : nullptr, *y4 = relu ? reinterpret_cast<const f32x4 *>(y + base) : nullptr;
        for (long long t = threadIdx.x; t < (T >> 2); t += 256) {
            f32x4 g = g4[t], v = x4[t];
            if (r4) v += r4[t];
            if (y4) {
                const f32x4 yy = y4[t];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!(yy[j] > 0.f)) g[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) a += g[j], b = __builtin_fmaf(g[j], (v[j] - mu) * rstd, b);
        }
    } else {
        for (long long t = threadIdx.x; t < T; t += 256) {
            float g = gy[base + t];
            if (relu && !(y[base + t] > 0.f)) g = 0.f;
            float v = x[base + t];
            if (res) v += res[base + t];
            a += g;
            b = __builtin_fmaf(g, (v - mu) * rstd, b);
        }
    }
    a = wave_sum(a), b = wave_sum(b);
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a, red[4 + (threadIdx.x >> 6)] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ta = red[0] + red[1] + red[2] + red[3], tb = red[4] + red[5] + red[6] + red[7];
        ws[2 * ((size_t)n * C + c)] = (double)ta;
        ws[2 * ((size_t)n * C + c) + 1] = (double)tb;
    }
}

// backward pass B: gx = rstd * (gamma gy' - S1/M - xhat * S2/M)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float *gy, const float *x, const float *res, const float *gamma,
                                                           const float *y, const float *stats, int C, long long T, int relu,
                                                           const double *ws, float *gx, int N, float *ggamma, float *gbeta) {
    const int c = blockIdx.x, n = blockIdx.y;
    const double M = (double)C * (double)T;
    const float mu = stats[2 * n], rstd = stats[2 * n + 1];
    if (n == 0) {                                // the parameter gradients of channel c: its rows of all samples, in sample order
        double ga = 0.0, gb = 0.0;
        for (int i = threadIdx.x; i < N; i += 256) ga += ws[2 * ((size_t)i * C + c)], gb += ws[2 * ((size_t)i * C + c) + 1];
        ga = wave_sum(ga), gb = wave_sum(gb);
        __shared__ double pred[8];
        if ((threadIdx.x & 63) == 0) pred[threadIdx.x >> 6] = ga, pred[4 + (threadIdx.x >> 6)] = gb;
        __syncthreads();
        if (threadIdx.x == 0) gbeta[c] = (float)((pred[0] + pred[1]) + (pred[2] + pred[3])), ggamma[c] = (float)((pred[4] + pred[5]) + (pred[6] + pred[7]));
    }
    double w1, w2;
    gn_ws_sum(ws, gamma, n, C, w1, w2);
    const float m1 = (float)(w1 / M), m2 = (float)(w2 / M);
    const float gc = gamma[c];
    const size_t base = ((size_t)n * C + c) * T;
    if (row_is_vec4(gy + base, x + base, res ? res + base : nullptr, T) && row_is_vec4(relu ? y + base : nullptr, gx + base, nullptr, T)) {
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gy + base), *x4 = reinterpret_cast<const f32x4 *>(x + base);
        const f32x4 *r4 = res ? reinterpret_cast<const f32x4 *>(res + base) : nullptr, *y4 = relu ? reinterpret_cast<const f32x4 *>(y + base) : nullptr;
        f32x4 *o4 = reinterpret_cast<f32x4 *>(gx + base);
        for (long long t = threadIdx.x; t < (T >> 2); t += 256) {
            f32x4 g = g4[t], v = x4[t], o;
            if (r4) v += r4[t];
            if (y4) {
                const f32x4 yy = y4[t];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!(yy[j] > 0.f)) g[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rstd * (g[j] * gc - m1 - (v[j] - mu) * rstd * m2);
            o4[t] = o;
        }
        return;
    }
    for (long long t = threadIdx.x; t < T; t += 256) {
        float g = gy[base + t];
        if (relu && !(y[base + t] > 0.f)) g = 0.f;
        float v = x[base + t];
        if (res) v += res[base + t];
        const float xh = (v - mu) * rstd;
        gx[base + t] = rstd * (g * gc - m1 - xh * m2);
    }
}

// ------------------------------------------------------------------------------------------------------------
// softmax over keys.  s: (B, Tk, Tq), query axis contiguous.  One workgroup = 64 query columns x all key rows;
// thread (row group rg = tid / 64, column c = tid % 64): online (max, sum) over its rows, combined through LDS.
//   fwd:  a = softmax_tk( scale * s[tk][tq] with key-padded rows -> -inf ), query-padded columns -> 0   (in place)
//   bwd:  gs = scale * a * (ga - sum_tk ga * a)     (masked positions have a = 0 -> gs = 0, as in the reference where
//          the masks are written through .data)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_keys_fwd_kernel(float *s, const unsigned char *mask, long long T, float scale) {
    const int b = blockIdx.y;
    const long long tq = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    float *sb = s + (size_t)b * T * T;
    const unsigned char *mb = mask ? mask + (size_t)b * T : nullptr;
    float mx = -INFINITY, sum = 0.f;
    if (tq < T) {
        for (long long tk = rg; tk < T; tk += 4) {
            if (mb && mb[tk]) continue;
            const float v = sb[tk * T + tq] * scale;
            if (v > mx) {
                sum = sum * __expf(mx - v) + 1.f;
                mx = v;
            } else {
                sum += __expf(v - mx);
            }
        }
    }
    __shared__ float smx[4][64], ssum[4][64];
    smx[rg][threadIdx.x & 63] = mx;
    ssum[rg][threadIdx.x & 63] = sum;
    __syncthreads();
    const int cc = threadIdx.x & 63;
    float gm = fmaxf(fmaxf(smx[0][cc], smx[1][cc]), fmaxf(smx[2][cc], smx[3][cc]));
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ssum[i][cc] > 0.f) gs += ssum[i][cc] * __expf(smx[i][cc] - gm);
    if (tq >= T) return;
    const bool qpad = mb && mb[tq];
    const float inv = 1.f / gs;     // every key padded -> 0/0 = NaN, exactly what softmax of an all -inf column gives
    for (long long tk = rg; tk < T; tk += 4) {
        float a = 0.f;
        if (!qpad && !(mb && mb[tk])) a = __expf(sb[tk * T + tq] * scale - gm) * inv;
        else if (!qpad && gs == 0.f) a = NAN;
        sb[tk * T + tq] = a;
    }
}

__global__ __launch_bounds__(256) void softmax_keys_bwd_kernel(const float *a, const float *ga, long long T, float scale, float *gs) {
    const int b = blockIdx.y;
    const long long tq = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    const size_t base = (size_t)b * T * T;
    float dot = 0.f;
    if (tq < T)
        for (long long tk = rg; tk < T; tk += 4) dot = __builtin_fmaf(ga[base + tk * T + tq], a[base + tk * T + tq], dot);
    __shared__ float sd[4][64];
    sd[rg][threadIdx.x & 63] = dot;
    __syncthreads();
    const int cc = threadIdx.x & 63;
    const float d = sd[0][cc] + sd[1][cc] + sd[2][cc] + sd[3][cc];
    if (tq >= T) return;
    for (long long tk = rg; tk < T; tk += 4) {
        const size_t o = base + tk * T + tq;
        gs[o] = scale * a[o] * (ga[o] - d);
    }
}

// ---- the same two kernels for T % 4 == 0 (rows 16-byte aligned): 16-byte accesses, 4 columns per thread ---------------------
// One workgroup = 64 query columns x all key rows again, but thread (row group rg = tid / 16 of 16, column quad cq = tid % 16)
// moves float4s, four rows of a row group in flight per iteration; the online (max, sum) update is branch-free.  The scalar
// kernels above ran at 1.6 TB/s of their 3 (fwd) / 5 (bwd) passes: one 4-byte load per lane and iteration, a data-dependent
// branch in the chain.
__device__ __forceinline__ void online_update(float v, float &mx, float &sum) {
    const float m2 = fmaxf(mx, v);
    sum = sum * __expf(mx - m2) + __expf(v - m2);            // mx = -inf, v finite: exp(-inf) = 0;  both -inf never happens (v finite)
    mx = m2;
}

__global__ __launch_bounds__(256) void softmax_keys_fwd4_kernel(float *s, const unsigned char *mask, long long T, float scale) {
    const int b = blockIdx.y;
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const long long tq = (long long)blockIdx.x * 64 + 4 * cq;
    float *sb = s + (size_t)b * T * T;
    const unsigned char *mb = mask ? mask + (size_t)b * T : nullptr;
    const bool live = tq < T;                                  // T % 4 == 0: a quad is inside or outside as a whole
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (long long tk0 = rg; tk0 < T; tk0 += 64) {
            f32x4 v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long tk = tk0 + 16 * u;
                ok[u] = tk < T && !(mb && mb[tk]);
                v[u] = ok[u] ? *reinterpret_cast<const f32x4 *>(sb + tk * T + tq) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) online_update(v[u][j] * scale, mx[j], sum[j]);
                }
        }
    }
    __shared__ float smx[16][64], ssum[16][64];
#pragma unroll
    for (int j = 0; j < 4; ++j) smx[rg][4 * cq + j] = mx[j], ssum[rg][4 * cq + j] = sum[j];
    __syncthreads();
    float gm[4], inv[4];
    bool none[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cc = 4 * cq + j;
        float m = -INFINITY;
        for (int i = 0; i < 16; ++i) m = fmaxf(m, smx[i][cc]);
        float g = 0.f;
        for (int i = 0; i < 16; ++i)
            if (ssum[i][cc] > 0.f) g += ssum[i][cc] * __expf(smx[i][cc] - m);
        gm[j] = m, inv[j] = 1.f / g, none[j] = g == 0.f;
    }
    if (!live) return;
    bool qpad[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qpad[j] = mb && mb[tq + j];
    for (long long tk0 = rg; tk0 < T; tk0 += 64) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long tk = tk0 + 16 * u;
            v[u] = tk < T ? *reinterpret_cast<const f32x4 *>(sb + tk * T + tq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long tk = tk0 + 16 * u;
            if (tk >= T) continue;
            const bool kpad = mb && mb[tk];
            f32x4 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = 0.f;
                if (!qpad[j] && !kpad) r = __expf(v[u][j] * scale - gm[j]) * inv[j];
                else if (!qpad[j] && none[j]) r = NAN;         // every key padded: softmax of an all -inf column
                a[j] = r;
            }
            *reinterpret_cast<f32x4 *>(sb + tk * T + tq) = a;
        }
    }
}

__global__ __launch_bounds__(256) void softmax_keys_bwd4_kernel(const float *a, const float *ga, long long T, float scale, float *gs) {
    const int b = blockIdx.y;
    const int cq = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const long long tq = (long long)blockIdx.x * 64 + 4 * cq;
    const size_t base = (size_t)b * T * T;
    const bool live = tq < T;
    f32x4 dot = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (long long tk0 = rg; tk0 < T; tk0 += 64) {
            f32x4 x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long tk = tk0 + 16 * u;
                const bool ok = tk < T;
                x[u] = ok ? *reinterpret_cast<const f32x4 *>(a + base + tk * T + tq) : f32x4{0.f, 0.f, 0.f, 0.f};
                y[u] = ok ? *reinterpret_cast<const f32x4 *>(ga + base + tk * T + tq) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) dot += x[u] * y[u];
        }
    }
    __shared__ float sd[16][64];
#pragma unroll
    for (int j = 0; j < 4; ++j) sd[rg][4 * cq + j] = dot[j];
    __syncthreads();
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] += sd[i][4 * cq + j];
    if (!live) return;
    for (long long tk0 = rg; tk0 < T; tk0 += 64) {
        f32x4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long tk = tk0 + 16 * u;
            const bool ok = tk < T;
            x[u] = ok ? *reinterpret_cast<const f32x4 *>(a + base + tk * T + tq) : f32x4{0.f, 0.f, 0.f, 0.f};
            y[u] = ok ? *reinterpret_cast<const f32x4 *>(ga + base + tk * T + tq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long tk = tk0 + 16 * u;
            if (tk < T) *reinterpret_cast<f32x4 *>(gs + base + tk * T + tq) = scale * x[u] * (y[u] - d);
        }
    }
}

}  // namespace

extern "C" int psnd_groupnorm1_fwd(const float *x, const float *res, const float *gamma, const float *beta, int64_t N, int C,
                                   int64_t T, float eps, int relu, float *y, float *stats, double *ws, void *stream) {
    if (!x || !gamma || !beta || !y || !stats || !ws) PSND_FAIL(PSND_E_ARG, "groupnorm1_fwd: null pointer");
    if (N < 0 || C <= 0 || T <= 0 || N > 65535) PSND_FAIL(PSND_E_SHAPE, "groupnorm1_fwd: N=%lld C=%d T=%lld", (long long)N, C, (long long)T);
    if (N == 0) return PSND_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(C, (unsigned)N), dim3(256), 0, s, x, res, C, (long long)T, ws);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(C, (unsigned)N), dim3(256), 0, s, x, res, gamma, beta, C, (long long)T, eps, relu, ws, y, stats);
    PSND_CHECK_LAUNCH("groupnorm1_fwd");
    return PSND_OK;
}

extern "C" int psnd_groupnorm1_bwd(const float *gy, const float *x, const float *res, const float *gamma, const float *y,
                                   const float *stats, int64_t N, int C, int64_t T, int relu, float *gx, float *ggamma,
                                   float *gbeta, double *ws, void *stream) {
    if (!gy || !x || !gamma || !stats || !gx || !ggamma || !gbeta || !ws || (relu && !y)) PSND_FAIL(PSND_E_ARG, "groupnorm1_bwd: null pointer");
    if (N < 0 || C <= 0 || T <= 0 || N > 65535) PSND_FAIL(PSND_E_SHAPE, "groupnorm1_bwd: bad shape");
    if (N == 0) return PSND_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(C, (unsigned)N), dim3(256), 0, s, gy, x, res, gamma, y, stats, C, (long long)T, relu, ws,
                       ggamma, gbeta);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(C, (unsigned)N), dim3(256), 0, s, gy, x, res, gamma, y, stats, C, (long long)T, relu, ws, gx, (int)N,
                       ggamma, gbeta);
    PSND_CHECK_LAUNCH("groupnorm1_bwd");
    return PSND_OK;
}

extern "C" int psnd_softmax_keys_fwd(float *scores, const uint8_t *mask, int64_t B, int64_t T, float scale, void *stream) {
    if (!scores) PSND_FAIL(PSND_E_ARG, "softmax_keys_fwd: null pointer");
    if (B < 0 || T <= 0 || B > 65535) PSND_FAIL(PSND_E_SHAPE, "softmax_keys_fwd: B=%lld T=%lld", (long long)B, (long long)T);
    if (B == 0) return PSND_OK;
    const bool v4 = T % 4 == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0 && PSND_ENV("PSND_SOFTMAX_SCALAR") == nullptr;
    hipLaunchKernelGGL(v4 ? softmax_keys_fwd4_kernel : softmax_keys_fwd_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), scores, mask, (long long)T, scale);
    PSND_CHECK_LAUNCH("softmax_keys_fwd");
    return PSND_OK;
}

extern "C" int psnd_softmax_keys_bwd(const float *att, const float *gatt, int64_t B, int64_t T, float scale, float *gscores,
                                     void *stream) {
    if (!att || !gatt || !gscores) PSND_FAIL(PSND_E_ARG, "softmax_keys_bwd: null pointer");
    if (B < 0 || T <= 0 || B > 65535) PSND_FAIL(PSND_E_SHAPE, "softmax_keys_bwd: bad shape");
    if (B == 0) return PSND_OK;
    const bool v4 = T % 4 == 0 && ((reinterpret_cast<uintptr_t>(att) | reinterpret_cast<uintptr_t>(gatt) | reinterpret_cast<uintptr_t>(gscores)) & 15) == 0 &&
                    PSND_ENV("PSND_SOFTMAX_SCALAR") == nullptr;
    hipLaunchKernelGGL(v4 ? softmax_keys_bwd4_kernel : softmax_keys_bwd_kernel, dim3((unsigned)((T + 63) / 64), (unsigned)B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), att, gatt, (long long)T, scale, gscores);
    PSND_CHECK_LAUNCH("softmax_keys_bwd");
    return PSND_OK;
}

// ---- PositionalEncoding.forward (modules.py:119-145): y[n][c][t] = x[n][c][t] * scale + pe[c][t] in ONE pass (the torch formulation is a
// scalar multiply + a broadcast add: 13 + 50 us at 32 x 256 x 1292); backward gx = g * scale.  pe: (C, pe_len) rows, pe_len >= T.
namespace {
__global__ __launch_bounds__(256) void posenc_kernel(const float *x, const float *pe, float scale, int C, long long T, long long pe_len, float *y,
                                                     long long rows) {
    // one row (n, c) per blockIdx.y stride; threads walk t
    for (long long row = blockIdx.y; row < rows; row += gridDim.y) {
        const float *xr = x + row * T, *pr = pe ? pe + (row % C) * pe_len : nullptr;
        float *yr = y + row * T;
        for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < T; t += (long long)gridDim.x * 256)
            yr[t] = pr ? __fadd_rn(__fmul_rn(xr[t], scale), pr[t]) : xr[t] * scale;      // (product rounded first, as the torch formulation)
    }
}
// ... four frames per thread (T % 4 == 0, 16-byte aligned rows: 1292 frames, a 2048-frame table), four 16-byte loads in flight per thread: the
// one-element form moved 84 MB in 32 us (49 k workgroups of one element per thread, the sixth of a row's six workgroups 12 threads wide)
__global__ __launch_bounds__(256) void posenc4_kernel(const float *x, const float *pe, float scale, int C, long long Q /* T / 4 */, long long peq, float *y,
                                                      long long total /* rows * Q */) {
#pragma clang fp contract(off)      // the product is rounded before the add, as the torch formulation's two kernels round it (bit-equal: tests)
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x), *p4 = reinterpret_cast<const f32x4 *>(pe);
    f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
    for (long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x); i0 < total; i0 += (long long)gridDim.x * 256 * 4) {
        f32x4 v[4], q[4];
        long long idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            idx[u] = i0 + (long long)u * gridDim.x * 256;
            const long long i = idx[u] < total ? idx[u] : total - 1;
            v[u] = x4[i];
            if (pe) {
                const long long row = i / Q;
                q[u] = p4[(row % C) * peq + (i - row * Q)];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (idx[u] >= total) continue;
            f32x4 o;
            if (pe) {
                const f32x4 m = v[u] * scale;
                o = m + q[u];
            } else {
                o.x = v[u].x * scale, o.y = v[u].y * scale, o.z = v[u].z * scale, o.w = v[u].w * scale;
            }
            y4[idx[u]] = o;
        }
    }
}
}  // namespace

extern "C" int psnd_posenc(const float *x, const float *pe, float scale, int64_t N, int C, int64_t T, int64_t pe_len, float *y, void *stream) {
    if (!x || !y) PSND_FAIL(PSND_E_ARG, "posenc: null pointer");
    if (N < 0 || C <= 0 || T <= 0 || (pe && pe_len < T)) PSND_FAIL(PSND_E_SHAPE, "posenc: N=%lld C=%d T=%lld pe_len=%lld", (long long)N, C, (long long)T, (long long)pe_len);
    if (N == 0) return PSND_OK;
    const long long rows = (long long)N * C;
    if (T % 4 == 0 && (!pe || pe_len % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(pe)) & 15) == 0) {
        const long long total = rows * (T / 4);
        long long blocks = (total + 4 * 256 - 1) / (4 * 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(posenc4_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, pe, scale, C, (long long)(T / 4),
                           (long long)(pe_len / 4), y, total);
        PSND_CHECK_LAUNCH("posenc");
        return PSND_OK;
    }
    const unsigned gx = (unsigned)((T + 255) / 256 > 8 ? 8 : (T + 255) / 256), gy = (unsigned)(rows > 16384 ? 16384 : rows);
    hipLaunchKernelGGL(posenc_kernel, dim3(gx, gy), dim3(256), 0, static_cast<hipStream_t>(stream), x, pe, scale, C, (long long)T, (long long)pe_len, y, rows);
    PSND_CHECK_LAUNCH("posenc");
    return PSND_OK;
}
