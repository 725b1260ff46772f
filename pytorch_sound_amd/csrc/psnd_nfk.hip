// psnd_nfk.hip - the spectrogram side of a masking model when the magnitudes are kept BIN-FASTEST, (N, F, K) (psnd_stft_mag_nfk):
//   psnd_to_cl_nfk              (N, F, K) fp32 -> channels-last bf16 (N, Lp, Cp), optionally through log1p
//   psnd_mask_head_l1_fwd_nfk   est = sigmoid(y) * mag (+ the partial sums of F.l1_loss(est, ref))
//   psnd_mask_head_l1_bwd_nfk   gy = (gest + coef g sign(est - ref)) * mag * s (1 - s)
// (N, F, K) IS the channels-last order: frame t of clip n is row n * Lp + HP + t of the CL matrix and bin c its channel c.  The
// (N, K, F) versions of these passes (to_cl_kernel / from_cl_kernel, psnd_conv.hip) transpose 32 x 32 tiles through LDS and move 4 B
// per lane and load; here every pass is a plain stream: a thread owns 8 consecutive channels of one row - 32 B of every fp32
// operand, 16 B of the bf16 one - and all its loads are in flight together (range-checked buffer loads, no branches).
//
// They take the place of log1p + layout change in front of the conv stack and of `torch.sigmoid(conv_post(..)) * mag` behind it
// (the separator composed from pytorch_sound/models/vocoders/hifi_gan.py:32-69 blocks, models/separator.py).  Bound: HBM.
#include "psnd_common.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hwbf16x2));
}

// 8 consecutive floats of row `row` of an (N F, K) tensor starting at channel c0: channels >= K read as zero.  K is odd in practice
// (513), so a row starts at any 4-byte boundary: two 16-byte buffer loads when all 8 channels exist, else element loads (the last
// chunk of a row only).
struct Row8 {
    float v[8];
};
__device__ __forceinline__ Row8 load8(__amdgpu_buffer_rsrc_t r, unsigned row_byte, int c0, int K, bool row_ok) {
    Row8 o;
    constexpr unsigned OOB = 0xffffffffu;
    if (c0 + 8 <= K) {                                    // (uniform for all but the last chunk of a row)
        const unsigned b = row_ok ? row_byte + 4u * (unsigned)c0 : OOB;
        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)b, 0, 0));
        const f32x4 c = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(row_ok ? b + 16u : OOB), 0, 0));
        o.v[0] = a.x, o.v[1] = a.y, o.v[2] = a.z, o.v[3] = a.w, o.v[4] = c.x, o.v[5] = c.y, o.v[6] = c.z, o.v[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o.v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((row_ok && c0 + j < K) ? row_byte + 4u * (unsigned)(c0 + j) : OOB), 0, 0));
    }
    return o;
}
__device__ __forceinline__ void store8(__amdgpu_buffer_rsrc_t r, unsigned row_byte, int c0, int K, const float (&v)[8]) {
    if (c0 + 8 <= K) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), r, (int)(row_byte + 4u * (unsigned)c0), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), r, (int)(row_byte + 4u * (unsigned)c0 + 16u), 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < K) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j]), r, (int)(row_byte + 4u * (unsigned)(c0 + j)), 0, 0);
    }
}

struct NfkParams {
    const float *x;          // to_cl: the (N, F, K) input;   mask head: mag
    const float *ref;        // mask head: the L1 target (N, F, K) or null
    const float *est_in;     // mask head backward: est of the forward
    const float *gest;       // mask head backward: gradient on est from elsewhere (may be null)
    const float *g;          // mask head backward: device scalar, gradient of the loss value
    const bf16_t *y;         // mask head: the logits, CL (N, Lp, Cp)
    bf16_t *out_cl;          // to_cl / mask head backward: CL output
    float *est;              // mask head forward: (N, F, K)
    double *part;            // mask head forward: one partial sum of |est - ref| per workgroup
    long long items;         // to_cl / backward: N Lp (Cp / 8);  forward: N F ceil(K / 8)
    int N, F, K, Lp, HP, Cp, preop;
    unsigned tensor_bytes;   // N F K 4 (< 2^32: one buffer descriptor covers a whole (N, F, K) tensor)
    float coef;
};

// item = (row of the CL matrix, chunk of 8 channels): every row of the output is written, halo rows and padded channels as zeros
template <bool MASKBWD>
__global__ __launch_bounds__(256) void to_cl_nfk_kernel(NfkParams p) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= p.items) return;
    const int cpr = p.Cp >> 3;
    const long long row = it / cpr;
    const int c0 = (int)(it - row * cpr) * 8;
    const int n = (int)(row / p.Lp), l = (int)(row - (long long)n * p.Lp), t = l - p.HP;
    const bool ok = t >= 0 && t < p.F;
    // ONE descriptor per tensor (wave-uniform: no waterfall loops around the loads); 32-bit byte offsets, checked at launch
    const unsigned rb = (unsigned)(((size_t)n * p.F + (size_t)(ok ? t : 0)) * p.K * 4);
    const int tb = (int)p.tensor_bytes;
    const __amdgpu_buffer_rsrc_t rx = make_uniform_rsrc(p.x, tb);
    float w[8];
    if constexpr (!MASKBWD) {
        const Row8 a = load8(rx, rb, c0, p.K, ok);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = a.v[j];
            if (p.preop == 1) {               // log1p for a bf16 result: the corrected log(1 + w) (to_cl_kernel, psnd_conv.hip)
                const float u = 1.f + v;
                v = u == 1.f ? v : __logf(u) * __fdividef(v, u - 1.f);
            }
            w[j] = v;
        }
    } else {
        const __amdgpu_buffer_rsrc_t re = make_uniform_rsrc(p.est_in, tb), rr = make_uniform_rsrc(p.ref, tb);
        const __amdgpu_buffer_rsrc_t rg = make_uniform_rsrc(p.gest ? p.gest : p.x, p.gest ? tb : 0);
        const Row8 m = load8(rx, rb, c0, p.K, ok), e = load8(re, rb, c0, p.K, ok), r = load8(rr, rb, c0, p.K, ok), ge = load8(rg, rb, c0, p.K, ok);
        const u32x4 yq = *reinterpret_cast<const u32x4 *>(p.y + (size_t)row * p.Cp + c0);       // 8 logits of this row (16-byte aligned: Cp % 8 == 0)
        const float l1c = p.coef * p.g[0];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned q = yq[j >> 1];
            const float yv = (j & 1) ? bf_hi(q) : bf_lo(q);
            const float sg = 1.f / (1.f + __expf(-yv));
            const float d = e.v[j] - r.v[j];
            const float gg = ge.v[j] + l1c * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            w[j] = gg * m.v[j] * sg * (1.f - sg);     // zero where mag reads zero: halo rows, padded channels
        }
    }
    const u32x4 o = {pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3]), pack_bf16(w[4], w[5]), pack_bf16(w[6], w[7])};
    *reinterpret_cast<u32x4 *>(p.out_cl + (size_t)row * p.Cp + c0) = o;
}

// item = (frame row of (N F), chunk of 8 bins)
__global__ __launch_bounds__(256) void mask_head_nfk_kernel(NfkParams p) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cpr = (p.K + 7) >> 3;
    float acc = 0.f;
    if (it < p.items) {
        const long long row = it / cpr;
        const int c0 = (int)(it - row * cpr) * 8;
        const int n = (int)(row / p.F), t = (int)(row - (long long)n * p.F);
        const unsigned rb = (unsigned)((size_t)row * p.K * 4);
        const int tb = (int)p.tensor_bytes;
        const __amdgpu_buffer_rsrc_t rm = make_uniform_rsrc(p.x, tb), rr = make_uniform_rsrc(p.ref ? p.ref : p.x, p.ref ? tb : 0);
        const __amdgpu_buffer_rsrc_t ro = make_uniform_rsrc(p.est, tb);
        const Row8 m = load8(rm, rb, c0, p.K, true), r = load8(rr, rb, c0, p.K, true);
        // the logits of the frame: CL row n Lp + HP + t, channels c0 .. c0 + 7 (Cp >= round_up(K, 8): inside the row)
        const u32x4 yq = *reinterpret_cast<const u32x4 *>(p.y + ((size_t)n * p.Lp + p.HP + t) * p.Cp + c0);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned q = yq[j >> 1];
            const float yv = (j & 1) ? bf_hi(q) : bf_lo(q);
            v[j] = m.v[j] / (1.f + __expf(-yv));
            if (c0 + j < p.K) acc += __builtin_fabsf(v[j] - r.v[j]);
        }
        store8(ro, rb, c0, p.K, v);
    }
    if (p.part) {
        double d = (double)acc;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
        __shared__ double red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) p.part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

int check_shape(const char *what, int64_t N, int K, int64_t F, int Lp, int HP, int Cp) {
    if (N < 0 || K <= 0 || F <= 0 || HP < 0 || Lp < F + 2 * HP || Cp < ((K + 7) & ~7) || Cp % 8 != 0 || (size_t)N * (size_t)K * (size_t)F * 4 >= ((size_t)1 << 32) - 64 ||
        (size_t)Lp * Cp * 2 >= ((size_t)1 << 31) || N * (int64_t)Lp * (Cp / 8) >= ((int64_t)1 << 39))
        PSND_FAIL(PSND_E_SHAPE, "%s: N=%lld K=%d F=%lld Lp=%d HP=%d Cp=%d", what, (long long)N, K, (long long)F, Lp, HP, Cp);
    return PSND_OK;
}

}  // namespace

extern "C" int psnd_to_cl_nfk(const float *x_nfk, int64_t N, int K, int64_t F, int Lp, int HP, int Cp, int preop, void *out, void *stream) {
    if (!x_nfk || !out) PSND_FAIL(PSND_E_ARG, "to_cl_nfk: null pointer");
    if (preop < 0 || preop > 1) PSND_FAIL(PSND_E_ARG, "to_cl_nfk: preop=%d", preop);
    if (int rc = check_shape("to_cl_nfk", N, K, F, Lp, HP, Cp)) return rc;
    if (N == 0) return PSND_OK;
    NfkParams p = {};
    p.x = x_nfk, p.out_cl = static_cast<bf16_t *>(out), p.N = (int)N, p.F = (int)F, p.K = K, p.Lp = Lp, p.HP = HP, p.Cp = Cp, p.preop = preop;
    p.items = N * (int64_t)Lp * (Cp / 8);
    p.tensor_bytes = (unsigned)((size_t)N * F * K * 4);
    hipLaunchKernelGGL(to_cl_nfk_kernel<false>, dim3((unsigned)((p.items + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    PSND_CHECK_LAUNCH("to_cl_nfk");
    return PSND_OK;
}

extern "C" int64_t psnd_mask_head_l1_blocks_nfk(int64_t N, int64_t F, int K) {
    if (N <= 0 || F <= 0 || K <= 0) return 0;
    return (N * F * ((K + 7) / 8) + 255) / 256;
}

extern "C" int psnd_mask_head_l1_fwd_nfk(const void *y, const float *mag_nfk, const float *ref_nfk, int64_t N, int K, int64_t F, int Lp, int HP,
                                         int Cp, float *est_nfk, double *part, void *stream) {
    if (!y || !mag_nfk || !est_nfk) PSND_FAIL(PSND_E_ARG, "mask_head_l1_fwd_nfk: null pointer");
    if ((ref_nfk == nullptr) != (part == nullptr)) PSND_FAIL(PSND_E_ARG, "mask_head_l1_fwd_nfk: ref and part go together");
    if (int rc = check_shape("mask_head_l1_fwd_nfk", N, K, F, Lp, HP, Cp)) return rc;
    if (N == 0) return PSND_OK;
    NfkParams p = {};
    p.x = mag_nfk, p.ref = ref_nfk, p.y = static_cast<const bf16_t *>(y), p.est = est_nfk, p.part = part;
    p.N = (int)N, p.F = (int)F, p.K = K, p.Lp = Lp, p.HP = HP, p.Cp = Cp;
    p.items = N * F * ((K + 7) / 8);
    p.tensor_bytes = (unsigned)((size_t)N * F * K * 4);
    hipLaunchKernelGGL(mask_head_nfk_kernel, dim3((unsigned)((p.items + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    PSND_CHECK_LAUNCH("mask_head_l1_fwd_nfk");
    return PSND_OK;
}

extern "C" int psnd_mask_head_l1_bwd_nfk(const float *gest_nfk, const float *mag_nfk, const void *y, const float *est_nfk, const float *ref_nfk,
                                         const float *g, float coef, int64_t N, int K, int64_t F, int Lp, int HP, int Cp, void *gy, void *stream) {
    if (!mag_nfk || !y || !est_nfk || !ref_nfk || !g || !gy) PSND_FAIL(PSND_E_ARG, "mask_head_l1_bwd_nfk: null pointer");
    if (int rc = check_shape("mask_head_l1_bwd_nfk", N, K, F, Lp, HP, Cp)) return rc;
    if (N == 0) return PSND_OK;
    NfkParams p = {};
    p.x = mag_nfk, p.ref = ref_nfk, p.est_in = est_nfk, p.gest = gest_nfk, p.g = g, p.coef = coef, p.y = static_cast<const bf16_t *>(y);
    p.out_cl = static_cast<bf16_t *>(gy), p.N = (int)N, p.F = (int)F, p.K = K, p.Lp = Lp, p.HP = HP, p.Cp = Cp;
    p.items = N * (int64_t)Lp * (Cp / 8);
    p.tensor_bytes = (unsigned)((size_t)N * F * K * 4);
    hipLaunchKernelGGL(to_cl_nfk_kernel<true>, dim3((unsigned)((p.items + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    PSND_CHECK_LAUNCH("mask_head_l1_bwd_nfk");
    return PSND_OK;
}
