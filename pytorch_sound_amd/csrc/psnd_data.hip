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

// SpectrogramMasker.forward (transforms.py:409-416): out[n][f] = ceil( mean of the padded mask over frame f ), the padded mask being
// win/2 ones, the wave-level mask, win/2 zeros; frame f covers padded samples [f hop, f hop + win).  A workgroup stages the span of
// FM frames in LDS once (coalesced) and every wave sums its frames from there; sum / win is an exact division, so a fully valid frame
// gives exactly 1 whatever win is.
constexpr int FM = 16;
__global__ __launch_bounds__(256) void frame_mask_kernel(const float *mask, long long T, int win, int hop, long long F, float *out) {
    extern __shared__ float s_span[];
    const int n = blockIdx.y;
    const long long f0 = (long long)blockIdx.x * FM;
    const int nf = (int)min((long long)FM, F - f0);
    const int span = (nf - 1) * hop + win;
    const long long p0 = f0 * hop - win / 2;                  // mask index of the span's first padded sample
    const float *m = mask + (size_t)n * T;
    for (int i = threadIdx.x; i < span; i += 256) {
        const long long t = p0 + i;
        s_span[i] = t < 0 ? 1.f : (t < T ? m[t] : 0.f);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int f = wave; f < nf; f += 4) {
        float a = 0.f;
        for (int i = lane; i < win; i += 64) a += s_span[f * hop + i];
        for (int k = 32; k >= 1; k >>= 1) a += __shfl_xor(a, k, 64);
        if (lane == 0) out[(size_t)n * F + f0 + f] = ceilf(a / (float)win);
    }
}

}  // namespace

extern "C" int64_t psnd_frame_mask_frames(int64_t T, int win, int hop) {
    if (T < 0 || win <= 0 || hop <= 0) return 0;
    const int64_t Tp = T + 2 * (int64_t)(win / 2);
    return Tp < win ? 0 : (Tp - win) / hop + 1;
}

extern "C" int psnd_frame_mask(const float *mask, int64_t N, int64_t T, int win, int hop, float *out, void *stream) {
    if (!mask || !out) PSND_FAIL(PSND_E_ARG, "frame_mask: null pointer");
    if (N < 0 || N > 65535 || T < 0 || win <= 0 || hop <= 0) PSND_FAIL(PSND_E_SHAPE, "frame_mask: N=%lld T=%lld win=%d hop=%d", (long long)N, (long long)T, win, hop);
    const int64_t F = psnd_frame_mask_frames(T, win, hop);
    if (N == 0 || F == 0) return PSND_OK;
    const size_t lds = ((size_t)(FM - 1) * hop + win) * sizeof(float);
    if (lds > 150 * 1024) PSND_FAIL(PSND_E_UNSUPPORTED, "frame_mask: win=%d hop=%d: a span of %d frames does not fit the LDS", win, hop, FM);
    const int64_t bx = (F + FM - 1) / FM;
    if (bx > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "frame_mask: F=%lld too large", (long long)F);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(frame_mask_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "frame_mask: set LDS size: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(frame_mask_kernel, dim3((unsigned)bx, (unsigned)N), dim3(256), lds, static_cast<hipStream_t>(stream), mask, (long long)T, win,
                       hop, (long long)F, out);
    PSND_CHECK_LAUNCH("frame_mask");
    return PSND_OK;
}

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
