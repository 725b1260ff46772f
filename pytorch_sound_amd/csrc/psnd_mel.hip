// psnd_mel.hip - mel projection + log + clamp (and its backward) on the fp32 matrix cores.
//
// Replaces `torch.matmul(mel_filter, mag)`, `torch.log(mel + off)`, clamp_min/clamp_max of
// LogMelSpectrogram.forward (pytorch_sound/models/transforms.py:235-243), the matmul + log10 /
// ln(clamp) of Audio2Mel (:364-365) and interface MelSpectrogram (interface/hifi_gan.py:58-61).
//
// OUT[r][f] = sum_c Wp[r][c] * IN[c][f] per clip, v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
//   forward : r = mel band (M), c = frequency bin (K), IN = magnitude
//   backward: r = frequency bin (K), c = mel band (M), IN = gout * dlog(mel_lin) (built on load)
// The filterbank is triangular (each row is non-zero on a short run of bins), so the plan stores,
// per 16-row tile, the [lo, hi) range of 4-column steps that hold any non-zero weight and the
// kernel only walks that band: ~6.6x fewer MFMAs than the dense product at 80 x 513, which is
// what moves the stage from the fp32-MFMA roof (157 TF) back under the HBM roof.
//
// One wave = one (clip, 64-frame tile, 16-row tile).  B operand: lane (kk = l>>4, fq = l&15) loads
// 16 B = 4 consecutive frames of row c0+kk, so a wave instruction reads 4 rows x 256 contiguous
// bytes; component j of that float4 feeds MFMA j, whose output column (l&15) is therefore frame
// 4*(l&15)+j: each lane ends up with 4 consecutive frames per output row -> 16-B stores.
#include "psnd_common.h"
#include <math.h>
#include <string.h>
#include <vector>

namespace {

// plan: int32 header[HDR] then float weights.
//   hdr[0]=M hdr[1]=K hdr[2]=MT hdr[3]=KS hdr[4]=KT hdr[5]=MS hdr[6]=fwd_w_off hdr[7]=bwd_w_off (floats from plan start)
//   hdr[8 + 2*t + {0,1}]            fwd band [lo,hi) in 4-bin steps for mel tile t   (t < MT)
//   hdr[8 + 2*MT + 2*t + {0,1}]     bwd band [lo,hi) in 4-mel steps for bin tile t   (t < KT)
//   fwd weights: Wf[(t*KS + s)*64 + lane] = W[16 t + (lane&15)][4 s + (lane>>4)]
//   bwd weights: Wb[(t*MS + s)*64 + lane] = W[4 s + (lane>>4)][16 t + (lane&15)]
inline int hdr_ints(int MT, int KT) { return (8 + 2 * MT + 2 * KT + 63) & ~63; }

struct MelParams {
    const float *in0;   // fwd: mag ; bwd: gout
    const float *in1;   // bwd: mel_lin
    const int *plan;
    float *out;         // fwd: log-mel ; bwd: gmag
    float *lin;         // fwd: optional linear mel
    long long F;
    int N, R, Cc;       // rows (out), cols (reduction)
    int RT, CS;         // row tiles, column steps
    int band_off;       // offset into hdr of this direction's band table
    int w_off;          // float offset of this direction's weights
    int nft;            // 64-frame tiles per clip
    int log_kind;
    float log_offset, pre_clamp_min, clamp_lo, clamp_hi;
    // fused F.l1_loss(log_mel, ref) (psnd_mel_l1_fwd / _bwd): forward adds sum |y - ref| of the wave to l1_part[wave] (double) and writes only
    // the linear mel; backward forms the incoming gradient coef * g[0] * sign(y - ref) on operand load (in0 = ref, in1 = mel_lin)
    const float *l1_ref;
    double *l1_part;
    const float *l1_g;
    float l1_coef;
    int wn_off, CS16;   // NFK forward: float offset of the 16-bin-group weights, groups per row tile
    int seg_off;        // forward, read-once kernel: int offset of the segment table (mel_layout)
};

// derivative of the forward epilogue wrt mel (0 where any clamp is active; matches autograd of
// torch.clamp(min=): gradient passes where input >= min / <= max).
__device__ __forceinline__ float log_grad(float mel, int kind, float off, float pre, float lo, float hi) {
    float pass = 1.f;
    float v = mel;
    if (pre >= 0.f) {
        if (v < pre) pass = 0.f;
        v = fmaxf(v, pre);
    }
    v += off;
    float y = v, d = 1.f;
    if (kind == PSND_LOG_E) {
        y = logf(v);
        d = 1.f / v;
    } else if (kind == PSND_LOG_10) {
        y = log10f(v);
        d = 0.43429448190325182765f / v;
    }
    if (y < lo || y > hi) pass = 0.f;
    return pass * d;
}

// gradient of  coef * sum |y - ref|  wrt the linear mel: sign(y - ref) * dy/dmel, y = the forward epilogue of mel (0 where a clamp is active)
__device__ __forceinline__ float l1_log_grad(float mel, float ref, int kind, float off, float pre, float lo, float hi) {
    float pass = 1.f, v = mel;
    if (pre >= 0.f) {
        if (v < pre) pass = 0.f;
        v = fmaxf(v, pre);
    }
    v += off;
    float y = v, d = 1.f;
    if (kind == PSND_LOG_E) {
        y = logf(v);
        d = 1.f / v;
    } else if (kind == PSND_LOG_10) {
        y = log10f(v);
        d = 0.43429448190325182765f / v;
    }
    if (y < lo || y > hi) pass = 0.f;
    y = fminf(fmaxf(y, lo), hi);
    const float sg = y > ref ? 1.f : (y < ref ? -1.f : 0.f);
    return sg * pass * d;
}

// B operand of one k-step: 4 consecutive frames of reduction row c, as a BRANCH-FREE buffer load (an out-of-range row, or a lane whose
// frames lie past the clip, reads zeros through the descriptor's range check).  A lane whose 4 frames straddle the end of the row
// reads the first frames of the next row - harmless: MFMA output columns are independent and frames >= F are never stored.
// (The first version tested `f + 3 < F` per lane with a scalar tail: every load sat in its own basic block, hipcc waited for each
// one where it was issued, and a wave walked its band with ONE load in flight - 14.5 us per launch at 32 AND at 64 clips, whatever
// the batch size MU.)
__device__ __forceinline__ f32x4 load_b_raw(__amdgpu_buffer_rsrc_t r, int c, int Cc, long long f, long long F) {
    const unsigned off = (c < Cc && f < F) ? (unsigned)(((size_t)c * F + f) * sizeof(float)) : 0xffffffffu;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

// NFK: the magnitude side of the product is bin-fastest, (N, F, K) (psnd_stft_mag_nfk) - forward: the INPUT, backward: the OUTPUT.
//   forward   a lane loads 16 B = 4 consecutive BINS 16 S + 4 kk .. + 3 of ONE frame (f + g for accumulator g); component j feeds the MFMA
//             whose A operand holds W[.][16 S + 4 kk + j] (a second weight table in 16-bin groups, plan[wn_off ..)): the same 16 MFMAs
//             per 16 bins and 4 loads of 16 B as the frame-fastest walk.  The last group of a row (K = 513: one bin) takes element loads.
//   backward  a lane owns 4 consecutive bins of a frame per accumulator: one 16-byte store each.
template <bool BWD, bool NFK = false>
__global__ __launch_bounds__(256) void mel_kernel(MelParams p) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long per_clip = (long long)p.nft * p.RT;
    if (wid >= per_clip * p.N) return;
    const int clip = (int)(wid / per_clip);
    const int rem = (int)(wid - (long long)clip * per_clip);
    const int ft = rem / p.RT, rt = rem - ft * p.RT;
    const long long F = p.F;
    const long long f = (long long)ft * 64 + 4 * (lane & 15);
    const int kk = lane >> 4;

    const int lo = p.plan[p.band_off + 2 * rt], hi = p.plan[p.band_off + 2 * rt + 1];
    const float *W = reinterpret_cast<const float *>(p.plan) + p.w_off + (size_t)rt * p.CS * 64 + lane;
    const int in_bytes = (int)((size_t)p.Cc * F * sizeof(float));
    const __amdgpu_buffer_rsrc_t r0 = make_uniform_rsrc(p.in0 + (size_t)clip * p.Cc * F, in_bytes);
    const __amdgpu_buffer_rsrc_t r1 = make_uniform_rsrc(BWD ? p.in1 + (size_t)clip * p.Cc * F : p.in0, BWD ? in_bytes : 0);
    const bool l1b = BWD && p.l1_g != nullptr;                                  // uniform: in0 is the L1 target
    const float l1c = l1b ? p.l1_coef * p.l1_g[0] : 0.f;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    // the band is walked in batches of MU k-steps with all loads of a batch in flight
#ifndef PSND_MEL_MU
#define PSND_MEL_MU 8
#endif
    constexpr int MU = PSND_MEL_MU;
    if constexpr (!BWD && NFK) {
        const int K = p.Cc;
        const f32x4 *Wn = reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p.plan) + p.wn_off) + (size_t)rt * p.CS16 * 64 + lane;
        const int lo16 = lo >> 2, hi16 = (hi + 3) >> 2;
        constexpr unsigned OOB = 0xffffffffu;
        // byte offset of bin 4 kk of frame f + g inside the clip, or out of range for a frame past the clip
        unsigned fb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) fb[g] = (f + g < F) ? (unsigned)((((size_t)(f + g)) * K + 4 * kk) * sizeof(float)) : OOB;
        for (int S0 = lo16; S0 < hi16; S0 += 2) {
            f32x4 a[2], b[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int S = S0 + u;
                const bool in = S < hi16;
                a[u] = in ? Wn[(size_t)S * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
                if (16 * S + 16 <= K || !in) {                                  // (uniform) whole group inside the row
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        b[u][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, (int)((in && fb[g] != OOB) ? fb[g] + 64u * (unsigned)S : OOB), 0, 0));
                } else {                                                        // the row ends inside the group: element by element
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            b[u][g][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                r0, (int)((fb[g] != OOB && 16 * S + 4 * kk + j < K) ? fb[g] + 64u * (unsigned)S + 4u * j : OOB), 0, 0));
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][0][j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][1][j], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][2][j], acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][3][j], acc3, 0, 0, 0);
                }
        }
    } else
    for (int s0 = lo; s0 < hi; s0 += MU) {
        float a[MU];
        f32x4 b[MU], m[BWD ? MU : 1];
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            const int c = s0 + u < hi ? 4 * (s0 + u) + kk : p.Cc;       // past the band: an out-of-range row (zeros)
            a[u] = W[(size_t)min(s0 + u, hi - 1) * 64];
            b[u] = load_b_raw(r0, c, p.Cc, f, F);
            if constexpr (BWD) m[u] = load_b_raw(r1, c, p.Cc, f, F);
        }
        if constexpr (BWD) {
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                if (l1b) {
                    b[u].x = l1c * l1_log_grad(m[u].x, b[u].x, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                    b[u].y = l1c * l1_log_grad(m[u].y, b[u].y, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                    b[u].z = l1c * l1_log_grad(m[u].z, b[u].z, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                    b[u].w = l1c * l1_log_grad(m[u].w, b[u].w, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                } else {
                    b[u].x *= log_grad(m[u].x, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                    b[u].y *= log_grad(m[u].y, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                    b[u].z *= log_grad(m[u].z, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                    b[u].w *= log_grad(m[u].w, p.log_kind, p.log_offset, p.pre_clamp_min, p.clamp_lo, p.clamp_hi);
                }
                // a row past the band / the matrix contributes nothing even where the gradient of zero is not zero
                if (!(s0 + u < hi && 4 * (s0 + u) + kk < p.Cc)) b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < MU; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].z, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u].w, acc3, 0, 0, 0);
        }
    }
    // D layout: col = lane&15 (-> frames f..f+3 across acc0..3), row = 4*(lane>>4) + reg
    const size_t obase = (size_t)clip * p.R * F;
    float l1acc = 0.f;
    if constexpr (BWD && NFK) {
        // D: row = bin 16 rt + 4 kk + reg, column = frame f + j of accumulator j -> (N, F, K): 4 consecutive bins of one frame per store
        const int bin0 = 16 * rt + 4 * kk;
        const f32x4 accs[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (f + j >= F || bin0 >= p.R) continue;
            float *dst = p.out + obase + (size_t)(f + j) * p.R + bin0;
            if (bin0 + 3 < p.R) {
                *reinterpret_cast<f32x4_u *>(dst) = accs[j];
            } else {
                for (int r = 0; r < 4; ++r)
                    if (bin0 + r < p.R) dst[r] = accs[j][r];
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt + 4 * kk + r;
        if (row >= p.R) continue;
        f32x4 v = {acc0[r], acc1[r], acc2[r], acc3[r]};
        f32x4 y = v;
        if constexpr (!BWD) {
            y.x = fminf(fmaxf(log_apply(v.x, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
            y.y = fminf(fmaxf(log_apply(v.y, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
            y.z = fminf(fmaxf(log_apply(v.z, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
            y.w = fminf(fmaxf(log_apply(v.w, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
        }
        const size_t o = obase + (size_t)row * F + f;
        if constexpr (!BWD) {
            if (p.l1_ref) {                                                // fused L1 against ref: the log-mel itself is not written
                if (f + 3 < F) {
                    const f32x4 rf = *reinterpret_cast<const f32x4_u *>(p.l1_ref + o);
                    l1acc += fabsf(y.x - rf.x) + fabsf(y.y - rf.y) + fabsf(y.z - rf.z) + fabsf(y.w - rf.w);
                    *reinterpret_cast<f32x4_u *>(p.lin + o) = v;
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (f + j < F) {
                            l1acc += fabsf(y[j] - p.l1_ref[o + j]);
                            p.lin[o + j] = v[j];
                        }
                }
                continue;
            }
        }
        if (f + 3 < F) {
            *reinterpret_cast<f32x4_u *>(p.out + o) = y;
            if (!BWD && p.lin) *reinterpret_cast<f32x4_u *>(p.lin + o) = v;
        } else {
            for (int j = 0; j < 4; ++j)
                if (f + j < F) {
                    p.out[o + j] = y[j];
                    if (!BWD && p.lin) p.lin[o + j] = v[j];
                }
        }
    }
    if constexpr (!BWD) {
        if (p.l1_part) {
            double d = (double)l1acc;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
            if (lane == 0) p.l1_part[wid] = d;
        }
    }
}

// ---- forward, every magnitude read ONCE (round 6) ---------------------------------------------------------------------------------------
// mel_kernel<false> gives every 16-mel tile a wave of its own, which walks the tile's band of bins: neighbouring tiles overlap in bins
// and a tile's 64-frame segments of a 692-byte row share their first and last lines with the neighbouring frame tiles of OTHER workgroups,
// i.e. other XCDs' L2s - 438.8 MB fetched for the 363.5 MB of a 1024-clip batch (profiles/stft_pmc.json).  Here a wave owns ALL the mel
// tiles of its (clip, 64-frame tile): it walks the union of the bands once, feeding every k-step to the one or two tiles whose triangles
// cover it (the plan's range table: per tile the k-steps it takes alone and those it shares with the next tile), and the frame tiles of a clip sit in one
// workgroup (shared lines meet in that CU's L1).  Bins above the last triangle (fmax < sr / 2) are never read.  The accumulators of MTMAX
// tiles stay in registers (16 VGPRs per tile); the MFMA count is the band-sparse one of mel_kernel.  A filter matrix whose bands overlap in
// more than two tiles (the dense DCT of MelToMFCC) has every tile walk its whole band alone, as in mel_kernel; more than kOnceTiles
// tiles (M > 128) take mel_kernel.
constexpr int kOnceTiles = 8;

template <int MTMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MTMAX <= 5 ? 4 : 2))) void mel_fwd_once_kernel(MelParams p) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)p.nft * p.N) return;
    const int clip = (int)(wid / p.nft);
    const int ft = (int)(wid - (long long)clip * p.nft);
    const long long F = p.F;
    const long long f = (long long)ft * 64 + 4 * (lane & 15);
    const int kk = lane >> 4;
    const __amdgpu_buffer_rsrc_t r0 = make_uniform_rsrc(p.in0 + (size_t)clip * p.Cc * F, (int)((size_t)p.Cc * F * sizeof(float)));
    const float *Wbase = reinterpret_cast<const float *>(p.plan) + p.w_off + lane;
    const int *seg = p.plan + p.seg_off;

    f32x4 acc[MTMAX][4];
#pragma unroll
    for (int t = 0; t < MTMAX; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int MU = 4;                      // k-steps per batch: all their loads in flight (4 KiB per wave, 16 waves per CU)
    auto run = [&](auto tc, auto twoc, int sb, int se) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
        constexpr bool TWO = decltype(twoc)::value;
        const float *W0 = Wbase + (size_t)T * p.CS * 64, *W1 = Wbase + (size_t)(T + 1) * p.CS * 64;
        for (int s0 = sb; s0 < se; s0 += MU) {
            float a0[MU], a1[MU];
            f32x4 b[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int sc = min(s0 + u, se - 1);
                a0[u] = W0[(size_t)sc * 64];
                if constexpr (TWO) a1[u] = W1[(size_t)sc * 64];
                b[u] = load_b_raw(r0, s0 + u < se ? 4 * (s0 + u) + kk : p.Cc, p.Cc, f, F);       // past the segment: zeros
            }
#pragma unroll
            for (int u = 0; u < MU; ++u) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[T][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b[u][g], acc[T][g], 0, 0, 0);
                    if constexpr (TWO) acc[T + 1][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b[u][g], acc[T + 1][g], 0, 0, 0);
                }
            }
        }
    };
    // tile T alone over [seg[4T], seg[4T+1]), then tiles T and T + 1 together over [seg[4T+2], seg[4T+3]): the k-steps of every tile in
    // ascending order, as mel_kernel walks them (same MFMA sequence per accumulator: bit-identical results)
    static_for<0, MTMAX>([&](auto tc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
        if (T < p.RT) {                                                                            // wave-uniform (scalar loads)
            run(tc, std::false_type{}, seg[4 * T], seg[4 * T + 1]);
            if constexpr (T + 1 < MTMAX) run(tc, std::true_type{}, seg[4 * T + 2], seg[4 * T + 3]);
        }
    });
    // D layout: col = lane&15 (-> frames f..f+3 across accumulators 0..3), row = 4*(lane>>4) + reg
    const size_t obase = (size_t)clip * p.R * F;
    static_for<0, MTMAX>([&](auto tc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
        float l1acc = 0.f;
        if (T < p.RT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * T + 4 * kk + r;
                if (row >= p.R) continue;
                const f32x4 v = {acc[T][0][r], acc[T][1][r], acc[T][2][r], acc[T][3][r]};
                f32x4 y;
                y.x = fminf(fmaxf(log_apply(v.x, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
                y.y = fminf(fmaxf(log_apply(v.y, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
                y.z = fminf(fmaxf(log_apply(v.z, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
                y.w = fminf(fmaxf(log_apply(v.w, p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
                const size_t o = obase + (size_t)row * F + f;
                if (p.l1_ref) {                                            // fused L1 against ref: the log-mel itself is not written
                    if (f + 3 < F) {
                        const f32x4 rf = *reinterpret_cast<const f32x4_u *>(p.l1_ref + o);
                        l1acc += fabsf(y.x - rf.x) + fabsf(y.y - rf.y) + fabsf(y.z - rf.z) + fabsf(y.w - rf.w);
                        *reinterpret_cast<f32x4_u *>(p.lin + o) = v;
                    } else {
                        for (int j = 0; j < 4; ++j)
                            if (f + j < F) {
                                l1acc += fabsf(y[j] - p.l1_ref[o + j]);
                                p.lin[o + j] = v[j];
                            }
                    }
                    continue;
                }
                if (f + 3 < F) {
                    *reinterpret_cast<f32x4_u *>(p.out + o) = y;
                    if (p.lin) *reinterpret_cast<f32x4_u *>(p.lin + o) = v;
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (f + j < F) {
                            p.out[o + j] = y[j];
                            if (p.lin) p.lin[o + j] = v[j];
                        }
                }
            }
            if (p.l1_part) {                  // the partial sums keep mel_kernel's order: one per (clip, frame tile, mel tile)
                double d = (double)l1acc;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
                if (lane == 0) p.l1_part[wid * p.RT + T] = d;
            }
        }
    });
}

struct MelHostPlan {
    int M, K, MT, KS, KT, MS, hdr, fw_off, bw_off, wn_off, KS16, seg_off;
    size_t total_bytes;
};
MelHostPlan mel_layout(int M, int K) {
    MelHostPlan h;
    h.M = M, h.K = K;
    h.MT = (M + 15) / 16, h.KS = (K + 3) / 4;
    h.KT = (K + 15) / 16, h.MS = (M + 3) / 4;
    h.hdr = hdr_ints(h.MT, h.KT);
    h.fw_off = h.hdr;
    h.bw_off = h.fw_off + h.MT * h.KS * 64;
    // forward weights once more, in 16-bin groups for the bin-fastest operand: Wn[((t*KS16 + S)*64 + lane)*4 + j] = W[16 t + (lane&15)][16 S + 4 (lane>>4) + j]
    h.KS16 = (K + 15) / 16;
    h.wn_off = h.bw_off + h.KT * h.MS * 64;
    // range table of the read-once forward (mel_fwd_once_kernel), four ints per mel tile t: [begin, end) of the k-steps tile t takes alone,
    // [begin, end) of those it shares with tile t + 1
    h.seg_off = h.wn_off + h.MT * h.KS16 * 256;
    h.total_bytes = sizeof(float) * ((size_t)h.seg_off + 4 * h.MT + 4);
    return h;
}

}  // namespace

extern "C" size_t psnd_mel_plan_bytes(int M, int K) {
    if (M <= 0 || K <= 0) return 0;
    return mel_layout(M, K).total_bytes;
}

extern "C" int psnd_mel_plan_build(int M, int K, const float *W, void *plan_host) {
    if (!W || !plan_host || M <= 0 || K <= 0) PSND_FAIL(PSND_E_ARG, "mel_plan_build: bad arguments");
    const MelHostPlan h = mel_layout(M, K);
    memset(plan_host, 0, h.total_bytes);
    int *hdr = static_cast<int *>(plan_host);
    float *fl = static_cast<float *>(plan_host);
    hdr[0] = M, hdr[1] = K, hdr[2] = h.MT, hdr[3] = h.KS, hdr[4] = h.KT, hdr[5] = h.MS;
    hdr[6] = h.fw_off, hdr[7] = h.bw_off;
    auto w = [&](int m, int k) -> float { return (m < M && k < K) ? W[(size_t)m * K + k] : 0.f; };
    for (int t = 0; t < h.MT; ++t) {
        int lo = h.KS, hi = 0;
        for (int s = 0; s < h.KS; ++s) {
            bool any = false;
            for (int lane = 0; lane < 64; ++lane) {
                const float v = w(16 * t + (lane & 15), 4 * s + (lane >> 4));
                fl[h.fw_off + ((size_t)t * h.KS + s) * 64 + lane] = v;
                any |= (v != 0.f);   // NaN != 0 is true: NaN weights stay inside the band
            }
            if (any) {
                if (s < lo) lo = s;
                hi = s + 1;
            }
        }
        if (hi <= lo) lo = hi = 0;
        hdr[8 + 2 * t] = lo, hdr[8 + 2 * t + 1] = hi;
    }
    for (int t = 0; t < h.KT; ++t) {
        int lo = h.MS, hi = 0;
        for (int s = 0; s < h.MS; ++s) {
            bool any = false;
            for (int lane = 0; lane < 64; ++lane) {
                const float v = w(4 * s + (lane >> 4), 16 * t + (lane & 15));
                fl[h.bw_off + ((size_t)t * h.MS + s) * 64 + lane] = v;
                any |= (v != 0.f);
            }
            if (any) {
                if (s < lo) lo = s;
                hi = s + 1;
            }
        }
        if (hi <= lo) lo = hi = 0;
        hdr[8 + 2 * h.MT + 2 * t] = lo, hdr[8 + 2 * h.MT + 2 * t + 1] = hi;
    }
    for (int t = 0; t < h.MT; ++t)
        for (int S = 0; S < h.KS16; ++S)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    fl[h.wn_off + (((size_t)t * h.KS16 + S) * 64 + lane) * 4 + j] = w(16 * t + (lane & 15), 16 * S + 4 * (lane >> 4) + j);
    // ranges of the read-once forward: tile t alone, then tiles t and t + 1 together.  Banded = the bands start and end in tile order and no
    // k-step lies in three of them; any other matrix (the dense DCT of MelToMFCC): every tile walks its whole band alone.
    {
        int *sg = hdr + h.seg_off;
        auto lo = [&](int t) { return hdr[8 + 2 * t]; };
        auto hi = [&](int t) { return hdr[8 + 2 * t + 1]; };
        bool banded = true;
        for (int t = 0; t + 1 < h.MT; ++t) banded &= lo(t) <= lo(t + 1) && hi(t) <= hi(t + 1) && hi(t) > lo(t) && hi(t + 1) > lo(t + 1);
        for (int t = 0; t + 2 < h.MT; ++t) banded &= hi(t) <= lo(t + 2);
        for (int t = 0; t < h.MT; ++t) {
            int s1b = lo(t), s1e = hi(t), s2b = 0, s2e = 0;
            if (banded) {
                if (t > 0 && hi(t - 1) > s1b) s1b = hi(t - 1);                 // [lo(t), hi(t-1)) was walked with tile t - 1
                if (t + 1 < h.MT && lo(t + 1) < s1e) s2b = lo(t + 1), s2e = s1e, s1e = s2b;
                if (s1e < s1b) s1e = s1b;
            }
            sg[4 * t] = s1b, sg[4 * t + 1] = s1e, sg[4 * t + 2] = s2b, sg[4 * t + 3] = s2e;
        }
    }
    return PSND_OK;
}

static int mel_launch(bool bwd, const float *in0, const float *in1, int64_t N, int64_t F, int M, int K,
                      const void *plan, int log_kind, float log_offset, float pre, float lo, float hi,
                      float *out, float *lin, void *stream, const float *l1_ref = nullptr, double *l1_part = nullptr,
                      const float *l1_g = nullptr, float l1_coef = 0.f, bool nfk = false) {
    if (!in0 || !plan || (!out && !l1_ref) || (bwd && !in1)) PSND_FAIL(PSND_E_ARG, "mel: null pointer");
    if (M <= 0 || K <= 0 || N < 0 || F < 0) PSND_FAIL(PSND_E_SHAPE, "mel: M=%d K=%d N=%lld F=%lld", M, K, (long long)N, (long long)F);
    if (log_kind < PSND_LOG_NONE || log_kind > PSND_LOG_10) PSND_FAIL(PSND_E_ARG, "mel: log_kind=%d", log_kind);
    if (N == 0 || F == 0) return PSND_OK;
    if ((size_t)(M > K ? M : K) * (size_t)F * sizeof(float) >= ((size_t)1 << 31))
        PSND_FAIL(PSND_E_SHAPE, "mel: a clip of %lld frames exceeds the 2 GB buffer range of one clip's operand", (long long)F);
    const MelHostPlan h = mel_layout(M, K);
    MelParams p;
    p.in0 = in0, p.in1 = in1, p.plan = static_cast<const int *>(plan), p.out = out, p.lin = lin;
    p.F = F, p.N = (int)N;
    p.log_kind = log_kind, p.log_offset = log_offset, p.pre_clamp_min = pre, p.clamp_lo = lo, p.clamp_hi = hi;
    p.nft = (int)((F + 63) / 64);
    p.l1_ref = l1_ref, p.l1_part = l1_part, p.l1_g = l1_g, p.l1_coef = l1_coef;
    p.wn_off = h.wn_off, p.CS16 = h.KS16, p.seg_off = h.seg_off;
    if (!bwd) {
        p.R = M, p.Cc = K, p.RT = h.MT, p.CS = h.KS, p.band_off = 8, p.w_off = h.fw_off;
    } else {
        p.R = K, p.Cc = M, p.RT = h.KT, p.CS = h.MS, p.band_off = 8 + 2 * h.MT, p.w_off = h.bw_off;
    }
    const bool once = !bwd && !nfk && h.MT <= kOnceTiles;          // forward on (N, K, F): every magnitude read once
    const long long waves = (long long)N * p.nft * p.RT;
    const long long blocks = (waves + 3) / 4;
    if (blocks >= (1ll << 31)) PSND_FAIL(PSND_E_SHAPE, "mel: grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nfk) {
        if (!bwd) hipLaunchKernelGGL((mel_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((mel_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    } else if (!bwd && once) {
        const long long ob = ((long long)N * p.nft + 3) / 4;
        if (h.MT <= 5) hipLaunchKernelGGL(mel_fwd_once_kernel<5>, dim3((unsigned)ob), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(mel_fwd_once_kernel<8>, dim3((unsigned)ob), dim3(256), 0, s, p);
    } else if (!bwd) hipLaunchKernelGGL(mel_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(mel_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    PSND_CHECK_LAUNCH(bwd ? "mel_bwd" : "mel_fwd");
    return PSND_OK;
}

extern "C" int psnd_mel_fwd(const float *mag, int64_t N, int64_t F, int M, int K, const void *mel_plan,
                            int log_kind, float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi,
                            float *out, float *mel_lin, void *stream) {
    return mel_launch(false, mag, nullptr, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo,
                      clamp_hi, out, mel_lin, stream);
}

extern "C" int psnd_mel_bwd(const float *gout, const float *mel_lin, int64_t N, int64_t F, int M, int K,
                            const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min,
                            float clamp_lo, float clamp_hi, float *gmag, void *stream) {
    return mel_launch(true, gout, mel_lin, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo,
                      clamp_hi, gmag, nullptr, stream);
}

// ---- F.l1_loss(log_mel(mag), ref) without the log-mel tensor: forward writes the linear mel (for the backward) and one partial sum of
//      |y - ref| per wave (psnd_mel_l1_blocks of them, double; fold them with psnd_l1_loss_combine); backward forms the incoming gradient
//      coef * g[0] * sign(y - ref) * dy/dmel while loading its operand (g: device scalar, the gradient of the loss value).
extern "C" int64_t psnd_mel_l1_blocks(int64_t N, int64_t F, int M) {
    if (N <= 0 || F <= 0 || M <= 0) return 0;
    return N * ((F + 63) / 64) * ((M + 15) / 16);
}
extern "C" int psnd_mel_l1_fwd(const float *mag, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind, float log_offset,
                               float pre_clamp_min, float clamp_lo, float clamp_hi, const float *ref, float *mel_lin, double *part,
                               void *stream) {
    if (!ref || !mel_lin || !part) PSND_FAIL(PSND_E_ARG, "mel_l1_fwd: null ref / mel_lin / part");
    return mel_launch(false, mag, nullptr, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, nullptr, mel_lin,
                      stream, ref, part);
}
extern "C" int psnd_mel_l1_bwd(const float *ref, const float *mel_lin, const float *g, float coef, int64_t N, int64_t F, int M, int K,
                               const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi,
                               float *gmag, void *stream) {
    if (!g) PSND_FAIL(PSND_E_ARG, "mel_l1_bwd: null g");
    return mel_launch(true, ref, mel_lin, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, gmag, nullptr,
                      stream, nullptr, nullptr, g, coef);
}

// ---- the same four with the magnitude side bin-fastest, (N, F, K) (psnd_stft_mag_nfk): forward INPUT mag_nfk, backward OUTPUT gmag_nfk; the
//      mel side stays (N, M, F) as transforms.py:235 returns it.
extern "C" int psnd_mel_fwd_nfk(const float *mag_nfk, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind, float log_offset,
                                float pre_clamp_min, float clamp_lo, float clamp_hi, float *out, float *mel_lin, void *stream) {
    return mel_launch(false, mag_nfk, nullptr, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, out, mel_lin, stream,
                      nullptr, nullptr, nullptr, 0.f, true);
}
extern "C" int psnd_mel_bwd_nfk(const float *gout, const float *mel_lin, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind,
                                float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi, float *gmag_nfk, void *stream) {
    return mel_launch(true, gout, mel_lin, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, gmag_nfk, nullptr, stream,
                      nullptr, nullptr, nullptr, 0.f, true);
}
extern "C" int psnd_mel_l1_fwd_nfk(const float *mag_nfk, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind, float log_offset,
                                   float pre_clamp_min, float clamp_lo, float clamp_hi, const float *ref, float *mel_lin, double *part,
                                   void *stream) {
    if (!ref || !mel_lin || !part) PSND_FAIL(PSND_E_ARG, "mel_l1_fwd_nfk: null ref / mel_lin / part");
    return mel_launch(false, mag_nfk, nullptr, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, nullptr, mel_lin,
                      stream, ref, part, nullptr, 0.f, true);
}
extern "C" int psnd_mel_l1_bwd_nfk(const float *ref, const float *mel_lin, const float *g, float coef, int64_t N, int64_t F, int M, int K,
                                   const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi,
                                   float *gmag_nfk, void *stream) {
    if (!g) PSND_FAIL(PSND_E_ARG, "mel_l1_bwd_nfk: null g");
    return mel_launch(true, ref, mel_lin, N, F, M, K, mel_plan, log_kind, log_offset, pre_clamp_min, clamp_lo, clamp_hi, gmag_nfk, nullptr,
                      stream, nullptr, nullptr, g, coef, true);
}
