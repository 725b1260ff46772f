// psnd_data.hip - the device half of pytorch_sound/data/dataset.py:196-250 (SpeechDataLoader.pad_collate_fn on the audio
// column): the host ships the batch's clips back to back (sum of lengths floats, one pinned H2D copy - the padding never
// crosses PCIe), and one launch lays them out as the zero-padded (N, Tmax) batch plus the validity mask the reference
// builds with np.ones_like + zero padding (dataset.py:70-71, 88-89).  HBM-bound: reads sum(len) floats, writes N * Tmax
// (x2 with the mask).
#include "psnd_common.h"

namespace {

// out[n][t] = t < lens[n] ? flat[offs[n] + t] : 0;   mask[n][t] = t < lens[n]
__global__ __launch_bounds__(256) void pad_collate_kernel(const float *flat, const long long *offs, const long long *lens, long long Tmax,
                                                          float *out, float *mask) {
    const int n = blockIdx.y;
    const long long len = min(lens[n], Tmax);
    const float *src = flat + offs[n];
    float *dst = out + (size_t)n * Tmax;
    float *msk = mask ? mask + (size_t)n * Tmax : nullptr;
    const long long t0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (t0 >= Tmax) return;
    // 16-byte stores when the row allows it (Tmax % 4 == 0 keeps every row aligned); the source offset is arbitrary
    if ((Tmax & 3) == 0) {
        f32x4 v, m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = t0 + j < len;
            v[j] = in ? src[t0 + j] : 0.f;
            m[j] = in ? 1.f : 0.f;
        }
        *reinterpret_cast<f32x4 *>(dst + t0) = v;
        if (msk) *reinterpret_cast<f32x4 *>(msk + t0) = m;
    } else {
        for (int j = 0; j < 4 && t0 + j < Tmax; ++j) {
            const bool in = t0 + j < len;
            dst[t0 + j] = in ? src[t0 + j] : 0.f;
            if (msk) msk[t0 + j] = in ? 1.f : 0.f;
        }
    }
}

}  // namespace

extern "C" int psnd_pad_collate(const float *flat, const int64_t *offs, const int64_t *lens, int64_t N, int64_t Tmax, float *out,
                                float *mask, void *stream) {
    if (!flat || !offs || !lens || !out) PSND_FAIL(PSND_E_ARG, "pad_collate: null pointer");
    if (N < 0 || N > 65535 || Tmax < 0) PSND_FAIL(PSND_E_SHAPE, "pad_collate: N=%lld Tmax=%lld", (long long)N, (long long)Tmax);
    if (N == 0 || Tmax == 0) return PSND_OK;
    const int64_t bx = (Tmax + 1023) / 1024;
    if (bx > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "pad_collate: Tmax=%lld too large", (long long)Tmax);
    hipLaunchKernelGGL(pad_collate_kernel, dim3((unsigned)bx, (unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream), flat,
                       reinterpret_cast<const long long *>(offs), reinterpret_cast<const long long *>(lens), (long long)Tmax, out, mask);
    PSND_CHECK_LAUNCH("pad_collate");
    return PSND_OK;
}
