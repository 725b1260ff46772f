// psnd_optim.hip - the optimizer step of Trainer.train (pytorch_sound/trainer.py:215-216, `self.optimizer.step()` with the
// Adam the reference's recipes construct) as ONE launch over every parameter tensor of the model.
// torch's fused multi-tensor Adam walks 64 Ki-element chunks (~160 workgroups for the 5.5 M-parameter separator: fewer
// workgroups than CUs, 1.1 TB/s); here a chunk is 2048 elements, ~2.7 k workgroups, 16-byte accesses.  HBM-bound:
// 28 bytes per parameter (read p, g, m, v; write p, m, v).
// The AMP `found_inf` protocol is honoured on the device (a non-zero flag skips the update AND the step count), so the
// Trainer's NaN skip needs no host synchronisation.
#include "psnd_common.h"
#include <math.h>

namespace {

struct AdamTensor {         // one parameter tensor (device table, 48 bytes)
    float *p;
    const float *g;
    float *m;
    float *v;
    float *step;            // this tensor's step count (float, as torch keeps it for fused / capturable optimizers)
    long long numel;
};

constexpr int ACH = 2048;   // elements per workgroup (256 threads x 2 float4)

// step += 1 and the two bias-correction factors of every tensor, in double like torch's kernels (1 - b^step cancels badly
// in float for the first steps): corr[2 i] = 1 / (1 - b1^step), corr[2 i + 1] = 1 / sqrt(1 - b2^step)
// flag_log (may be null): a host-visible (pinned, device-mapped) int ring {flag, seq} per slot.  The first thread leaves the step's skip flag
// there and, behind a system-scope fence, the sequence number the host waits for - the trainer learns about a skipped (NaN) step without
// a device-to-host copy and an event behind every optimizer launch (4 us of blit kernel + ~10 us of idle queue per config-2 step).
__global__ __launch_bounds__(64) void adam_tick_kernel(const AdamTensor *tab, int n, double b1, double b2, const float *found_inf, float *corr,
                                                        int *flag_log, int log_slot, int log_seq) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (flag_log && i == 0) {
        volatile int *slot = flag_log + 2 * log_slot;
        slot[0] = (found_inf && *found_inf != 0.f) ? 1 : 0;
        __threadfence_system();
        slot[1] = log_seq;
    }
    if (i >= n) return;
    if (found_inf && *found_inf != 0.f) return;
    const float step = *tab[i].step + 1.f;
    *tab[i].step = step;
    corr[2 * i] = (float)(1.0 / (1.0 - pow(b1, (double)step)));
    corr[2 * i + 1] = (float)(1.0 / sqrt(1.0 - pow(b2, (double)step)));
}

// chunk_tensor[b], chunk_off[b]: tensor index and first element of workgroup b's chunk
template <bool DECOUPLED>
__global__ __launch_bounds__(256) void adam_kernel(const AdamTensor *tab, const int *chunk_tensor, const long long *chunk_off, float lr,
                                                   float b1, float b2, float omb1, float omb2, float eps, float wd, const float *found_inf,
                                                   const float *grad_scale, const float *corr, float clip_value, const float *clip_coef) {
    if (found_inf && *found_inf != 0.f) return;
    const int ti = chunk_tensor[blockIdx.x];
    const AdamTensor T = tab[ti];
    const long long e0 = chunk_off[blockIdx.x];
    const long long e1 = min(e0 + ACH, T.numel);
    const float step_size = lr * corr[2 * ti], inv_sqrt_bc2 = corr[2 * ti + 1];      // from adam_tick_kernel
    const float gs = grad_scale ? 1.f / *grad_scale : 1.f;
    // Trainer.clip_grad (trainer.py:184-191) folded in: clamp to +-clip_value, then the global-norm factor of psnd_grad_sumsq
    // torch's clamp propagates NaN (fminf / fmaxf return the non-NaN operand): a NaN gradient must stay NaN and poison the parameter
    // visibly, as `p.grad.clamp(...)` + clip_grad_norm_ do in the reference.  No clamp at all when clip_value == 0.
    const bool do_clamp = clip_value > 0.f;
    const float cv = clip_value, cc = clip_coef ? *clip_coef : 1.f;
    auto upd = [&](float &p, float g, float &m, float &v) __attribute__((always_inline)) {
        g *= gs;
        if (do_clamp) g = (g != g) ? g : __builtin_fminf(__builtin_fmaxf(g, -cv), cv);
        g *= cc;
        if constexpr (DECOUPLED) p -= lr * wd * p;
        else g = __builtin_fmaf(wd, p, g);
        m = __builtin_fmaf(b1, m, omb1 * g);
        v = __builtin_fmaf(b2, v, omb2 * g * g);
        const float denom = __builtin_fmaf(__builtin_sqrtf(v), inv_sqrt_bc2, eps);
        p -= step_size * (m / denom);
    };
    const bool al = ((((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v) & 15) == 0);
    if (al) {
        for (long long e = e0 + 4 * threadIdx.x; e < e1; e += 1024) {
            if (e + 3 < e1) {
                f32x4 p = *reinterpret_cast<f32x4 *>(T.p + e), m = *reinterpret_cast<f32x4 *>(T.m + e), v = *reinterpret_cast<f32x4 *>(T.v + e);
                const f32x4 g = *reinterpret_cast<const f32x4 *>(T.g + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pj = p[j], mj = m[j], vj = v[j];
                    upd(pj, g[j], mj, vj);
                    p[j] = pj, m[j] = mj, v[j] = vj;
                }
                *reinterpret_cast<f32x4 *>(T.p + e) = p;
                *reinterpret_cast<f32x4 *>(T.m + e) = m;
                *reinterpret_cast<f32x4 *>(T.v + e) = v;
            } else {
                for (long long q = e; q < e1; ++q) upd(T.p[q], T.g[q], T.m[q], T.v[q]);
            }
        }
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) upd(T.p[e], T.g[e], T.m[e], T.v[e]);
    }
}

// ---- global gradient norm of Trainer.clip_grad (trainer.py:184-191: per-parameter clamp, then torch's clip_grad_norm_) over the
// same table / work list: partial[b] = sum over chunk b of clamp(g / grad_scale, +-clip_value)^2 (double), then ONE workgroup
// adds the partials in index order (deterministic) to *sumsq and writes coef[0] = min(1, max_norm / (sqrt(sumsq) + 1e-6)) (torch's
// clip_coef_clamped), coef[1] = the norm.  HBM-bound: 4 bytes per parameter.
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const AdamTensor *tab, const int *chunk_tensor, const long long *chunk_off,
                                                          float clip_value, const float *grad_scale, double *partial) {
    const AdamTensor T = tab[chunk_tensor[blockIdx.x]];
    const long long e0 = chunk_off[blockIdx.x];
    const long long e1 = min(e0 + ACH, T.numel);
    const float gs = grad_scale ? 1.f / *grad_scale : 1.f;
    const bool do_clamp = clip_value > 0.f;
    const float cv = clip_value;
    float acc = 0.f;
    auto add = [&](float g) __attribute__((always_inline)) {
        g *= gs;
        if (do_clamp) g = (g != g) ? g : __builtin_fminf(__builtin_fmaxf(g, -cv), cv);      // NaN stays NaN (torch.clamp semantics)
        acc = __builtin_fmaf(g, g, acc);
    };
    if ((((uintptr_t)T.g) & 15) == 0) {
        for (long long e = e0 + 4 * threadIdx.x; e < e1; e += 1024) {
            if (e + 3 < e1) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(T.g + e);
                add(g[0]), add(g[1]), add(g[2]), add(g[3]);
            } else {
                for (long long q = e; q < e1; ++q) add(T.g[q]);
            }
        }
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) add(T.g[e]);
    }
    __shared__ double red[256];
    red[threadIdx.x] = (double)acc;                       // <= 8 terms per thread in float, the tree in double
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void grad_sumsq_final_kernel(const double *partial, long long n, int accumulate, float max_norm, double *sumsq,
                                                                float *coef) {
    __shared__ double red[256];
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double tot = red[0] + (accumulate ? *sumsq : 0.0);
        *sumsq = tot;
        const float norm = (float)sqrt(tot);
        const float c = max_norm / (norm + 1e-6f);
        coef[0] = max_norm > 0.f ? ((c != c) ? c : fminf(c, 1.f)) : 1.f;      // a NaN norm poisons every gradient, as torch's clamp(max=1) does
        coef[1] = norm;
    }
}

}  // namespace

extern "C" int psnd_grad_sumsq(const void *table, int n_tensors, const int *chunk_tensor, const int64_t *chunk_off, int64_t n_chunks,
                               float clip_value, const float *grad_scale, int accumulate, float max_norm, double *partial, double *sumsq,
                               float *coef, void *stream) {
    if (!table || !chunk_tensor || !chunk_off || !partial || !sumsq || !coef) PSND_FAIL(PSND_E_ARG, "grad_sumsq: null pointer");
    if (n_tensors <= 0 || n_chunks <= 0 || n_chunks > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "grad_sumsq: n_tensors=%d n_chunks=%lld", n_tensors, (long long)n_chunks);
    if (!(clip_value >= 0.f) || !(max_norm >= 0.f)) PSND_FAIL(PSND_E_ARG, "grad_sumsq: clip_value=%g max_norm=%g", (double)clip_value, (double)max_norm);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, static_cast<const AdamTensor *>(table), chunk_tensor,
                       reinterpret_cast<const long long *>(chunk_off), clip_value, grad_scale, partial);
    hipLaunchKernelGGL(grad_sumsq_final_kernel, dim3(1), dim3(256), 0, s, partial, (long long)n_chunks, accumulate ? 1 : 0, max_norm, sumsq, coef);
    PSND_CHECK_LAUNCH("grad_sumsq");
    return PSND_OK;
}

extern "C" int64_t psnd_adam_chunk(void) { return ACH; }
extern "C" int64_t psnd_adam_table_bytes(void) { return (int64_t)sizeof(AdamTensor); }

extern "C" int psnd_adam_step(const void *table, int n_tensors, const int *chunk_tensor, const int64_t *chunk_off, int64_t n_chunks,
                              double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled,
                              const float *found_inf, const float *grad_scale, float *corr, float clip_value, const float *clip_coef,
                              void *stream) {
    return psnd_adam_step_logged(table, n_tensors, chunk_tensor, chunk_off, n_chunks, lr, beta1, beta2, eps, weight_decay, decoupled, found_inf,
                                 grad_scale, corr, clip_value, clip_coef, nullptr, 0, 0, stream);
}

extern "C" int psnd_adam_step_logged(const void *table, int n_tensors, const int *chunk_tensor, const int64_t *chunk_off, int64_t n_chunks,
                                     double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled,
                                     const float *found_inf, const float *grad_scale, float *corr, float clip_value, const float *clip_coef,
                                     int *flag_log, int log_slot, int log_seq, void *stream) {
    if (flag_log && log_slot < 0) PSND_FAIL(PSND_E_ARG, "adam_step: log_slot=%d", log_slot);
    if (!table || !chunk_tensor || !chunk_off || !corr) PSND_FAIL(PSND_E_ARG, "adam_step: null pointer");
    if (!(clip_value >= 0.f)) PSND_FAIL(PSND_E_ARG, "adam_step: clip_value=%g", (double)clip_value);
    if (n_tensors < 0 || n_chunks < 0 || n_chunks > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "adam_step: n_tensors=%d n_chunks=%lld", n_tensors, (long long)n_chunks);
    if (!(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.) || !(lr >= 0.) || !(weight_decay >= 0.))
        PSND_FAIL(PSND_E_ARG, "adam_step: lr=%g betas=(%g, %g) eps=%g weight_decay=%g", lr, beta1, beta2, eps, weight_decay);
    if (n_tensors == 0 || n_chunks == 0) return PSND_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const AdamTensor *tab = static_cast<const AdamTensor *>(table);
    hipLaunchKernelGGL(adam_tick_kernel, dim3((n_tensors + 63) / 64), dim3(64), 0, s, tab, n_tensors, beta1, beta2, found_inf, corr, flag_log, log_slot, log_seq);
    if (decoupled)
        hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)n_chunks), dim3(256), 0, s, tab, chunk_tensor, reinterpret_cast<const long long *>(chunk_off),
                           (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, found_inf, grad_scale, corr, clip_value, clip_coef);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)n_chunks), dim3(256), 0, s, tab, chunk_tensor, reinterpret_cast<const long long *>(chunk_off),
                           (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, found_inf, grad_scale, corr, clip_value, clip_coef);
    PSND_CHECK_LAUNCH("adam_step");
    return PSND_OK;
}

// ---- bf16 communication buffers of the data-parallel gradient all-reduce (distributed.FlatGradReducer, comm_dtype = bf16) ---------------
// The reducer's flat buckets stay fp32 (the gradients the backward kernels write and the optimizer reads); what crosses xGMI is a bf16
// image of a bucket: pack (fp32 -> bf16, round to nearest even) -> all-reduce -> unpack (bf16 -> fp32, exact).  Half the bytes per link;
// n is a multiple of 8 (bucket slots are 64-float aligned).
namespace {
typedef __bf16 hwbf16x2_o __attribute__((ext_vector_type(2)));
typedef float f32x2_o __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void grad_pack_bf16_kernel(const float *src, unsigned *dst, long long n8, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(src + 8 * i), b = *reinterpret_cast<const f32x4 *>(src + 8 * i + 4);
    const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, b.x * scale, b.y * scale, b.z * scale, b.w * scale};
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x2_o t = {v[2 * e], v[2 * e + 1]};
        w[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, hwbf16x2_o));
    }
    *reinterpret_cast<uint4 *>(dst + 4 * i) = make_uint4(w[0], w[1], w[2], w[3]);
}
__global__ __launch_bounds__(256) void grad_unpack_bf16_kernel(const unsigned *src, float *dst, long long n8, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 q = *reinterpret_cast<const uint4 *>(src + 4 * i);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __builtin_bit_cast(float, w[e] << 16) * scale;
        v[2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xffff0000u) * scale;
    }
    const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4 *>(dst + 8 * i) = a;
    *reinterpret_cast<f32x4 *>(dst + 8 * i + 4) = b;
}
}  // namespace

extern "C" int psnd_grad_pack_bf16(const float *src, void *dst, int64_t n, float scale, void *stream) {
    if (!src || !dst) PSND_FAIL(PSND_E_ARG, "grad_pack_bf16: null pointer");
    if (n < 0 || n % 8 != 0) PSND_FAIL(PSND_E_SHAPE, "grad_pack_bf16: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return PSND_OK;
    hipLaunchKernelGGL(grad_pack_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       static_cast<unsigned *>(dst), (long long)(n / 8), scale);
    PSND_CHECK_LAUNCH("grad_pack_bf16");
    return PSND_OK;
}
extern "C" int psnd_grad_unpack_bf16(const void *src, float *dst, int64_t n, float scale, void *stream) {
    if (!src || !dst) PSND_FAIL(PSND_E_ARG, "grad_unpack_bf16: null pointer");
    if (n < 0 || n % 8 != 0) PSND_FAIL(PSND_E_SHAPE, "grad_unpack_bf16: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return PSND_OK;
    hipLaunchKernelGGL(grad_unpack_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned *>(src), dst, (long long)(n / 8), scale);
    PSND_CHECK_LAUNCH("grad_unpack_bf16");
    return PSND_OK;
}
