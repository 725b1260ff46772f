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
__global__ __launch_bounds__(64) void adam_tick_kernel(const AdamTensor *tab, int n, double b1, double b2, const float *found_inf, float *corr) {
    const int i = blockIdx.x * 64 + threadIdx.x;
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
                                                   const float *grad_scale, const float *corr) {
    if (found_inf && *found_inf != 0.f) return;
    const int ti = chunk_tensor[blockIdx.x];
    const AdamTensor T = tab[ti];
    const long long e0 = chunk_off[blockIdx.x];
    const long long e1 = min(e0 + ACH, T.numel);
    const float step_size = lr * corr[2 * ti], inv_sqrt_bc2 = corr[2 * ti + 1];      // from adam_tick_kernel
    const float gs = grad_scale ? 1.f / *grad_scale : 1.f;
    auto upd = [&](float &p, float g, float &m, float &v) __attribute__((always_inline)) {
        g *= gs;
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

}  // namespace

extern "C" int64_t psnd_adam_chunk(void) { return ACH; }
extern "C" int64_t psnd_adam_table_bytes(void) { return (int64_t)sizeof(AdamTensor); }

extern "C" int psnd_adam_step(const void *table, int n_tensors, const int *chunk_tensor, const int64_t *chunk_off, int64_t n_chunks,
                              double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled,
                              const float *found_inf, const float *grad_scale, float *corr, void *stream) {
    if (!table || !chunk_tensor || !chunk_off || !corr) PSND_FAIL(PSND_E_ARG, "adam_step: null pointer");
    if (n_tensors < 0 || n_chunks < 0 || n_chunks > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "adam_step: n_tensors=%d n_chunks=%lld", n_tensors, (long long)n_chunks);
    if (!(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.) || !(lr >= 0.) || !(weight_decay >= 0.))
        PSND_FAIL(PSND_E_ARG, "adam_step: lr=%g betas=(%g, %g) eps=%g weight_decay=%g", lr, beta1, beta2, eps, weight_decay);
    if (n_tensors == 0 || n_chunks == 0) return PSND_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const AdamTensor *tab = static_cast<const AdamTensor *>(table);
    hipLaunchKernelGGL(adam_tick_kernel, dim3((n_tensors + 63) / 64), dim3(64), 0, s, tab, n_tensors, beta1, beta2, found_inf, corr);
    if (decoupled)
        hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)n_chunks), dim3(256), 0, s, tab, chunk_tensor, reinterpret_cast<const long long *>(chunk_off),
                           (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, found_inf, grad_scale, corr);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)n_chunks), dim3(256), 0, s, tab, chunk_tensor, reinterpret_cast<const long long *>(chunk_off),
                           (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, found_inf, grad_scale, corr);
    PSND_CHECK_LAUNCH("adam_step");
    return PSND_OK;
}
