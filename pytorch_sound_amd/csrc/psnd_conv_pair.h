// psnd_conv_pair.h - device body of the two-chained-convs kernel (see psnd_conv_pair.hip), shared with the paired backward launch of
// psnd_conv.hip where it is the input-gradient role of a residual pair.
#pragma once
#include "psnd_common.h"
#include <stdlib.h>

namespace {
namespace pairk {
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hwbf16x2));
}

struct PairParams {
    const bf16_t *A;            // (R, C) input of the first conv
    const bf16_t *W1, *W2;      // [k][C][C] fragment-ordered packs
    const float *bias1, *bias2; // C or null
    const bf16_t *M1, *M2;      // (R, C) or null: v *= (M > 0 ? 1 : slope)
    const bf16_t *res;          // (R, C) or null, added to the second conv's output
    bf16_t *mid_out;            // (R, C) or null
    bf16_t *out_raw, *out_act;  // (R, C), either may be null
    long long R;
    int Lp, L, HP;
    int off1, dstep1, h1;       // taps of the first conv: rows off1 + t * dstep1 (t < KT, the kernel's tap count), reach h1
    int off2, dstep2, h2;
    float m1_slope, m2_slope, act1_slope, act2_slope;
    long long *trace;          // PSND_PAIR_TRACE_PTR (tools/trace_pair.py): 8 s_memtime stamps per workgroup, or null
};

#define PAIR_STAMP(i_)                                                                  \
    do {                                                                                \
        if (p.trace && threadIdx.x == 0) p.trace[(size_t)tile * 8 + (i_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#ifndef PSND_PAIR_RU
#define PSND_PAIR_RU 12
#endif
constexpr unsigned OOB = 0x80000000u;
constexpr int RU8 = PSND_PAIR_RU;          // B units (one tap of one k-step) in flight per wave, 8-wave workgroups
constexpr int HMAXP = 8;        // largest tap reach of either conv of the 3-tap instances (dilation <= 8)
constexpr int HMAXW = 25;       // ... of the 7- / 11-tap instances (hifi_gan_v1 / v2: k = 11, dilation 5)

__device__ __forceinline__ uint4 ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

// NW waves = CGW column groups x RG row groups, a wave owning NBW column blocks of 32 for MB row blocks of 32:
//   NW = 8 (the stand-alone kernel, 512 threads): C = 256 -> 8 x 1 groups, one column block per wave, both row blocks (every B fragment feeds
//           MR MFMAs); C = 128 -> 4 x 2.  Two waves per SIMD on purpose: a wave issues in order and a 1 KB buffer load occupies its issue
//           stream for ~60 cycles (MI355X guide) - with ONE wave per SIMD (the first version: 256 threads, 2 x 2 accumulators per wave) a
//           unit of 4 MFMAs + 2 loads + 2 ds_reads took 272 cycles instead of 128; the second wave's MFMAs run under them.
//   NW = 4 (256 threads): the input-gradient role inside the paired backward launch of psnd_conv.hip, next to weight-gradient workgroups
//           that supply the other waves of a SIMD; two column blocks per wave at C = 256.
// MR: 32-row blocks of `mid` per workgroup (1: launches that would otherwise leave most CUs without a workgroup - the per-CU store
// path, ~12 B/clk, and the load issue slots are what a workgroup waits for, so more CUs is what helps; 2: every B fragment feeds
// two MFMAs).  RU: ring depth in units (a multiple of KT that divides KT C / 16).  KT: taps of both convs (3; round 5: 7 and 11, the
// other two resblock kernels of a HiFi-GAN stage), HMX: the largest tap reach the tile bookkeeping is sized for.
template <int C, int MR, bool HASM1, int NW, int RU, int KT = 3, int HMX = HMAXP>
__device__ __forceinline__ void conv_pair_body(const PairParams &p, const int tile, bf16_t *smem) {
    constexpr int NT = 64 * NW;
    constexpr int RS = C + 8, PCS = C / 8, KSTEPS = C / 16, UNITS = KT * KSTEPS;
    constexpr int MROWS = 32 * MR, CGT = C / 32, NBW = CGT > NW ? CGT / NW : 1, CG = CGT / NBW, RG = NW / CG, MB = MR / RG;
    static_assert(CG * RG == NW && MB * RG == MR && MB >= 1 && NBW * CG == CGT, "NW waves tile MROWS rows x C columns");
    constexpr int NAU = ((MROWS + 2 * HMX) * PCS + NT - 1) / NT;
    static_assert(UNITS % RU == 0 && RU % KT == 0, "the B ring turns whole");   // (RU % 3 != 0: see the end of a turn in run_conv)
    __shared__ unsigned char s_in1[MROWS + 2 * HMX], s_in2[MROWS];      // row inside its clip (mid rows / output rows)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kg = lane >> 5;
    const int cg = wave % CG, rg = wave / CG;
    const int col0 = cg * NBW * 32 + li;                             // this lane's first output channel (both convs); + 32 nb
    const int TS = MROWS - 2 * p.h2;
    const long long r0 = (long long)tile * TS, m0 = r0 - p.h2, a0 = m0 - p.h1;
    const int rowsA = MROWS + 2 * p.h1, rowsM = MROWS + 2 * p.h2;
    bf16_t *sA = smem, *sM = smem + rowsA * RS;
    const unsigned t_bytes = (unsigned)((size_t)p.R * C * sizeof(bf16_t)), w_bytes = (unsigned)((unsigned)KT * C * C * sizeof(bf16_t));
    const __amdgpu_buffer_rsrc_t rA = make_uniform_rsrc(p.A, (int)t_bytes);
    const __amdgpu_buffer_rsrc_t rW1 = make_uniform_rsrc(p.W1, (int)w_bytes), rW2 = make_uniform_rsrc(p.W2, (int)w_bytes);
    PAIR_STAMP(0);

    // ---- everything this workgroup reads before its first MFMA is requested now: the input tile, the first B units, bias, mask of mid
    uint4 ra[NAU];
    const int nA = rowsA * PCS;
#pragma unroll
    for (int u = 0; u < NAU; ++u) {
        const int idx = tid + NT * u, rr = idx / PCS, pc = idx % PCS;
        const long long r = a0 + rr;
        ra[u] = ld16(rA, (idx < nA && r >= 0 && r < p.R) ? (unsigned)(((size_t)r * C + 8 * pc) * sizeof(bf16_t)) : OOB);
    }
    uint4 rb[RU][NBW];
    const unsigned fwave = (unsigned)(cg * NBW) * (unsigned)KSTEPS * 1024u + (unsigned)lane * 16u;
    constexpr unsigned FTAP = (unsigned)(C / 32) * (unsigned)KSTEPS * 1024u;
    // unit u = (k-step u / KT, tap u % KT); the ring is RU units = RU / KT k-steps long (12 units = 4 k-steps at 3 taps), so slot s always
    // holds tap s % KT of k-step (RU / KT) it + s / KT:
    // every address is a per-lane base plus a compile-time / scalar offset - no per-unit address arithmetic beside the MFMAs
    auto fetch_b = [&](auto slotc, __amdgpu_buffer_rsrc_t rW, int it) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value, tap = slot % KT, ksl = slot / KT;
        const int ks = (RU / KT) * it + ksl;                                  // uniform
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
            rb[slot][nb] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(ks < KSTEPS ? fwave : OOB),
                                                                                            (int)((unsigned)tap * FTAP + (unsigned)(nb * KSTEPS + ks) * 1024u), 0));
    };
    static_for<0, RU>([&](auto uc) __attribute__((always_inline)) { fetch_b(uc, rW1, 0); });
    float b1[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) b1[nb] = p.bias1 ? p.bias1[col0 + 32 * nb] : 0.f;
    // mask of the first conv's output, at this lane's accumulator elements (row rho(rg_, kg) of block m, column col)
    unsigned short mk[HASM1 ? MB : 1][NBW][16];
    if constexpr (HASM1) {
        const __amdgpu_buffer_rsrc_t rM = make_uniform_rsrc(p.M1, (int)t_bytes);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long long r = m0 + (rg * MB + m) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
                    mk[m][nb][q] = __builtin_amdgcn_raw_buffer_load_b16(rM, (r >= 0 && r < p.R) ? (unsigned)(((size_t)r * C + col0 + 32 * nb) * 2) : OOB, 0, 0);
                }
    }
    if constexpr (MROWS + 2 * HMX <= 128 && 128 + MROWS <= NT) {     // (the 3-tap instances: one row per thread, two thread ranges)
        if (tid < rowsM) {
            const long long r = m0 + tid;
            const int l = (int)(((r % p.Lp) + p.Lp) % p.Lp);
            s_in1[tid] = (r >= 0 && r < p.R && l >= p.HP && l < p.HP + p.L) ? 1 : 0;
        } else if (tid >= 128 && tid < 128 + MROWS) {
            const int i = tid - 128;
            const long long r = r0 + i;
            const int l = (int)(r % p.Lp);
            s_in2[i] = (i < TS && r < p.R && l >= p.HP && l < p.HP + p.L) ? 1 : 0;
        }
    } else {
        for (int i = tid; i < rowsM; i += NT) {
            const long long r = m0 + i;
            const int l = (int)(((r % p.Lp) + p.Lp) % p.Lp);
            s_in1[i] = (r >= 0 && r < p.R && l >= p.HP && l < p.HP + p.L) ? 1 : 0;
        }
        for (int i = tid; i < MROWS; i += NT) {
            const long long r = r0 + i;
            const int l = (int)(r % p.Lp);
            s_in2[i] = (i < TS && r < p.R && l >= p.HP && l < p.HP + p.L) ? 1 : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < NAU; ++u) {
        const int idx = tid + NT * u;
        if (idx < nA) *reinterpret_cast<uint4 *>(sA + (idx / PCS) * RS + 8 * (idx % PCS)) = ra[u];
    }
    for (int idx = tid; idx < 2 * p.h2 * PCS; idx += NT)              // rows of mid past the 64 computed ones: read by the second
        *reinterpret_cast<uint4 *>(sM + (MROWS + idx / PCS) * RS + 8 * (idx % PCS)) = make_uint4(0, 0, 0, 0);   // conv's discarded rows only
    __syncthreads();
    PAIR_STAMP(1);

    f32x16 acc[MB][NBW];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][nb][i] = 0.f;
    };
    // one conv over the tile in `src` (row i of the output <-> row i + hh + tap offset of src); the ring holds units u .. u + RU - 1
    auto run_conv = [&](const bf16_t *src, int hh, int off0, int dstep, __amdgpu_buffer_rsrc_t rW) __attribute__((always_inline)) {
        const bf16_t *ab[KT];                                           // A fragment of (tap, k-step 0) for this lane's first row block
#pragma unroll
        for (int t = 0; t < KT; ++t) ab[t] = src + (rg * MB * 32 + li + hh + off0 + t * dstep) * RS + 8 * kg;
        auto afrag = [&](auto uc, int it, bf16x8 (&x)[MB]) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value, tap = u % KT, ksl = u / KT;  // u may run past the turn: k-step (RU / KT) it + ksl all the same
            const bf16_t *pa = ab[tap] + 16 * ((RU / KT) * it + ksl);
#pragma unroll
            for (int m = 0; m < MB; ++m) x[m] = *reinterpret_cast<const bf16x8 *>(pa + m * 32 * RS);
        };
        // A fragments run AD = 2 units ahead of their MFMAs in a ring of three (the LDS latency of ~100 cycles is longer than one unit)
        bf16x8 xr[3][MB];
        afrag(std::integral_constant<int, 0>{}, 0, xr[0]);
        afrag(std::integral_constant<int, 1>{}, 0, xr[1]);
        // Not unrolled, and every unit fenced: a refill is consumed one ring turn later.  Left to itself the scheduler sinks the loads
        // next to their use to save registers (fully unrolled: vmcnt(1) in front of every MFMA pair, 33 us per launch) or batches all
        // the refills of a turn at its end.
#pragma unroll 1
        for (int it = 0; it < UNITS / RU; ++it) {
            static_for<0, RU>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value;
                afrag(std::integral_constant<int, u + 2>{}, it, xr[(u + 2) % 3]);     // past the last k-step: stale bytes of the tile, unused
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const bf16x8 b = __builtin_bit_cast(bf16x8, rb[u][nb]);
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr[u % 3][m], b, acc[m][nb], 0, 0, 0);
                }
                fetch_b(uc, rW, it + 1);
                __builtin_amdgcn_sched_group_barrier(0x100, MB, 0);         // A fragments two units ahead (LDS) ...
                __builtin_amdgcn_sched_group_barrier(0x008, MB * NBW, 0);   // ... this unit's MFMAs ...
                __builtin_amdgcn_sched_group_barrier(0x020, NBW, 0);        // ... then the refill of this ring slot
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (RU % 3 != 0) {
                // the A ring of three turns with the units: a turn of 14 (7 taps) or 11 units leaves the fragments of the next turn's
                // units 0 / 1 in slots RU % 3 / (RU + 1) % 3 - moved to where that turn reads them (2 MB register quads per turn)
                bf16x8 t0[MB], t1[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) t0[m] = xr[RU % 3][m], t1[m] = xr[(RU + 1) % 3][m];
#pragma unroll
                for (int m = 0; m < MB; ++m) xr[0][m] = t0[m], xr[1][m] = t1[m];
            }
        }
    };

    // ---- first conv -> mid (bf16) in LDS
    zero_acc();
    run_conv(sA, p.h1, p.off1, p.dstep1, rW1);
    PAIR_STAMP(2);
    static_for<0, RU>([&](auto uc) __attribute__((always_inline)) { fetch_b(uc, rW2, 0); });   // land during the epilogue
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (rg * MB + m) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
                float v = acc[m][nb][q] + b1[nb];
                if constexpr (HASM1) v *= bf2f(mk[m][nb][q]) > 0.f ? 1.f : p.m1_slope;
                v = v > 0.f ? v : v * p.act1_slope;
                if (!s_in1[i]) v = 0.f;
                sM[i * RS + col0 + 32 * nb] = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu);
            }
    PAIR_STAMP(6);
    // what the store epilogue will need from memory, requested before the second conv so that it has landed by then
    constexpr int NFIN = MROWS * PCS / NT;
    static_assert(NFIN >= 1 && NFIN * NT == MROWS * PCS && NT % PCS == 0, "store epilogue tiling");
    uint4 qres[NFIN], qm2[NFIN];
    f32x4 qb0 = {0.f, 0.f, 0.f, 0.f}, qb1 = {0.f, 0.f, 0.f, 0.f};
    {
        const __amdgpu_buffer_rsrc_t rR = make_uniform_rsrc(p.res ? p.res : p.A, p.res ? (int)t_bytes : 0);
        const __amdgpu_buffer_rsrc_t rM2 = make_uniform_rsrc(p.M2 ? p.M2 : p.A, p.M2 ? (int)t_bytes : 0);
#pragma unroll
        for (int u = 0; u < NFIN; ++u) {
            const int idx = tid + NT * u, i = idx / PCS, c8 = 8 * (idx % PCS);
            const long long r = r0 + i;
            const unsigned o = (i < TS && r < p.R) ? (unsigned)(((size_t)r * C + c8) * 2) : OOB;
            qres[u] = ld16(rR, o);
            qm2[u] = ld16(rM2, o);
        }
        if (p.bias2) {
            const int c8 = 8 * (tid % PCS);                             // NT % PCS == 0: the same 8 channels in every round
            qb0 = *reinterpret_cast<const f32x4 *>(p.bias2 + c8), qb1 = *reinterpret_cast<const f32x4 *>(p.bias2 + c8 + 4);
        }
    }
    __syncthreads();
    PAIR_STAMP(7);
    if (p.mid_out) {                                                   // the rows this workgroup owns, 16 bytes per store
        const __amdgpu_buffer_rsrc_t rO = make_uniform_rsrc(p.mid_out, (int)t_bytes);
#pragma unroll
        for (int u = 0; u < NFIN; ++u) {
            const int idx = tid + NT * u, i = idx / PCS, pc = idx % PCS;
            const long long r = r0 + i;
            const uint4 v = *reinterpret_cast<const uint4 *>(sM + (p.h2 + i) * RS + 8 * pc);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rO,
                                                   (int)((i < TS && r < p.R) ? (unsigned)(((size_t)r * C + 8 * pc) * 2) : OOB), 0, 0);
        }
    }

    // ---- second conv on mid
    PAIR_STAMP(3);
    zero_acc();
    run_conv(sM, p.h2, p.off2, p.dstep2, rW2);
    PAIR_STAMP(4);

    // ---- epilogue through LDS (fp32 tile over the whole dynamic area): every thread finishes 8 consecutive channels of a row
    constexpr int OS = C + 8;
    float *sO = reinterpret_cast<float *>(smem);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (rg * MB + m) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
                sO[i * OS + col0 + 32 * nb] = acc[m][nb][q];
            }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NFIN; ++u) {
        const int idx = tid + NT * u, i = idx / PCS, c8 = 8 * (idx % PCS);
        const long long r = r0 + i;
        if (i >= TS || r >= p.R) continue;
        const size_t o = (size_t)r * C + c8;
        float v[8];
        if (s_in2[i]) {
            const f32x4 c0 = *reinterpret_cast<const f32x4 *>(sO + i * OS + c8), c1 = *reinterpret_cast<const f32x4 *>(sO + i * OS + c8 + 4);
            v[0] = c0.x + qb0.x, v[1] = c0.y + qb0.y, v[2] = c0.z + qb0.z, v[3] = c0.w + qb0.w;
            v[4] = c1.x + qb1.x, v[5] = c1.y + qb1.y, v[6] = c1.z + qb1.z, v[7] = c1.w + qb1.w;
            if (p.M2) {
                const unsigned *pq = reinterpret_cast<const unsigned *>(&qm2[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] *= bf2f((bf16_t)(pq[e] & 0xffff)) > 0.f ? 1.f : p.m2_slope;
                    v[2 * e + 1] *= bf2f((bf16_t)(pq[e] >> 16)) > 0.f ? 1.f : p.m2_slope;
                }
            }
            if (p.res) {
                const unsigned *pq = reinterpret_cast<const unsigned *>(&qres[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += bf2f((bf16_t)(pq[e] & 0xffff));
                    v[2 * e + 1] += bf2f((bf16_t)(pq[e] >> 16));
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        if (p.out_raw) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack_bf16(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<uint4 *>(p.out_raw + o) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (p.out_act) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = v[2 * e], x1 = v[2 * e + 1];
                w[e] = pack_bf16(x0 > 0.f ? x0 : x0 * p.act2_slope, x1 > 0.f ? x1 : x1 * p.act2_slope);
            }
            *reinterpret_cast<uint4 *>(p.out_act + o) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    PAIR_STAMP(5);
}

inline int reachk(int off0, int dstep, int k) {
    int h = 0;
    for (int t = 0; t < k; ++t) {
        const int o = off0 + t * dstep;
        h = (o < 0 ? -o : o) > h ? (o < 0 ? -o : o) : h;
    }
    return h;
}
inline int reach3(int off0, int dstep) { return reachk(off0, dstep, 3); }

}  // namespace pairk

}  // namespace
