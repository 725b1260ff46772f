// psnd_conv_chain.hip - a CHAIN of residual pairs (`leaky -> conv(d) -> leaky -> conv(1) -> + x`, hifi_gan.py:56-62: the three pairs of a
// ResBlock1) in ONE launch, for gfx950.  Forward only (no masks); per pair the same arithmetic, in the same order and with the same
// rounding points, as psnd_conv1d_cl_pair - the results are bit-identical to a sequence of pair launches.
//
// Why: at the config-2 size a pair launch is 13 us back to back of which a workgroup lives 8.7 us (psnd_conv_pair.hip); the rest is the
// launch boundary - the slowest workgroup, the drain of its stores, the first-touch latency of the next launch's input tile across the
// XCDs.  Twelve such launches are the forward of the separator's body.  Here a workgroup carries its rows through all pairs of the chain:
//   * FIXED tile coordinates: tile row j is global row g0 + j at every stage, the workgroup computes rows [0, 64) of every conv, and
//     what a conv cannot know about rows outside the tile simply makes its outermost rows wrong - the valid range shrinks by the tap
//     reach per conv (the first conv reads a real halo from memory and loses nothing).  A ResBlock1 (reaches 1,1, 3,1, 5,1) leaves
//     rows [11, 53): the workgroup OWNS those 42 rows and writes them, at every stage (they are valid at every stage); the recomputed
//     rims cost 1.52 x the MFMAs, which this size has to spare.
//   * the activated output of a pair goes straight from the accumulators (bias, residual, leaky) into the LDS tile the next pair's
//     first conv reads; the residual stream (rounded to bf16 exactly as the tensor a pair launch writes) stays in registers in the
//     accumulator layout.  No transposing fp32 epilogue through LDS, no global round trip between pairs; the tensors the backward needs
//     (mid, raw, activated of every pair) are copied out of LDS with 16-byte stores while the next conv runs.
//   * 64-row tiles: every B fragment (1 KB per wave and unit, straight from L2 in the fragment-ordered packs) feeds two MFMAs, so the
//     weight stream (2.3 MB per workgroup and ResBlock through the CU's vector-memory path) and the matrix pipe are balanced.
// LDS: act tile [80][264] + mid tile [80][264] + raw tile [64][264] bf16 = 118 KB: one 512-thread workgroup per CU.
#include "psnd_conv_pair.h"

#include <atomic>
extern std::atomic<long long> g_conv_pair_stats[4];     // psnd_conv.hip
std::atomic<long long> g_conv_chain_stats[2];            // chain launches, pairs carried by them

namespace {
using namespace pairk;

constexpr int CHAIN_MAX = 4;
struct ChainPair {
    const bf16_t *W1, *W2;
    const float *bias1, *bias2;
    bf16_t *mid_out, *out_raw, *out_act;
    int off1, dstep1, off2, dstep2;
    float act1_slope, act2_slope;
    const bf16_t *M1, *M2;      // BWD: leaky' masks of the two convs' outputs (v *= M > 0 ? 1 : slope)
    float m1_slope, m2_slope;
};
struct ChainParams {
    const bf16_t *A, *res;      // (R, C): activated input of the first pair, the residual stream in front of it
    long long R;
    int Lp, L, HP;
    int n_pairs, lo, ts;        // owned tile rows [lo, lo + ts)
    ChainPair d[CHAIN_MAX];
    long long *trace;           // PSND_PAIR_TRACE_PTR (tools/trace_chain.py): 16 s_memtime stamps per WAVE, or null
};
#define CHAIN_STAMP(i_)                                                                  \
    do {                                                                                 \
        if (p.trace && (threadIdx.x & 63) == 0) p.trace[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

template <int C, int MR, bool BWD>
__global__ __launch_bounds__(512, 1) void conv_chain_kernel(ChainParams p) {
    constexpr int NT = 512, RU = RU8;
#ifndef PSND_CHAIN_AR
#define PSND_CHAIN_AR 3
#endif
    constexpr int AR = PSND_CHAIN_AR;          // A-fragment ring (a divisor of RU)
    static_assert(RU % AR == 0, "static ring slots");
    constexpr int RS = C + 8, PCS = C / 8, KSTEPS = C / 16, UNITS = 3 * KSTEPS;
    constexpr int MROWS = 32 * MR, MB = MR, BR = MROWS + 2 * HMAXP;
    static_assert(C == 256, "eight waves = eight column blocks of 32");
    static_assert(UNITS % RU == 0 && RU % 3 == 0, "the B ring turns whole");
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    __shared__ unsigned char s_in[MROWS];
    // Hand-over between the convs of the chain WITHOUT a workgroup barrier (-DPSND_CHAIN_FLAGS=1; measured, OFF by default): s_flag[g] counts the epilogues
    // the waves of group g (0: waves 0-3 = channels 0-127 of a tile, 1: waves 4-7 = channels 128-255) have finished.  A conv's loop may
    // start on k-steps 0-7 as soon as group 0 has written the tile's first half and needs group 1 only before k-step 8.  Why: a SIMD
    // issues its older wave first, so waves 0-3 finish every loop ~3 k cycles before waves 4-7; behind a barrier the late waves' epilogue
    // (1.2-2 k cycles per conv) ran with the matrix pipe idle - with the flags the early waves are already multiplying the next conv's first
    // half.  Result (tools/trace_chain.py, bit-identical outputs): 65.7 k cycles per workgroup against 64.9 k with barriers - the late
    // waves' chain (loop + epilogue) is what a SIMD's matrix pipe and VALU can do for its two waves together (6.1 k MFMA + ~3 k VALU
    // cycles per conv), the early waves running ahead only compete with them; with the gate INSIDE the unrolled loop 70 k.
    __shared__ unsigned s_flag[2];
#ifndef PSND_CHAIN_FLAGS
#define PSND_CHAIN_FLAGS 0
#endif
    constexpr bool FLAGS = PSND_CHAIN_FLAGS != 0;
#ifndef PSND_CHAIN_EPRIO
#define PSND_CHAIN_EPRIO 2
#endif
    if (threadIdx.x < 2) s_flag[threadIdx.x] = 0;
    auto wait_flag = [&](int g, unsigned need) __attribute__((always_inline)) {
        if (!FLAGS || need == 0) return;
        volatile unsigned *f = &s_flag[g];
        int spins = 0;
        while (*f < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 24)) __builtin_trap();       // a lost hand-over must fail loudly, not hang the queue
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto signal_flag = [&]() __attribute__((always_inline)) {
        if (!FLAGS) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the LDS writes of this wave's epilogue are ordered in front of the count
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&s_flag[threadIdx.x >> 8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    bf16_t *sXa = smem, *sM = smem + BR * RS, *sXr = smem + 2 * BR * RS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kg = lane >> 5;
    const int col0 = wave * 32 + li;
    const long long g0 = (long long)blockIdx.x * p.ts - p.lo;
    const unsigned t_bytes = (unsigned)((size_t)p.R * C * sizeof(bf16_t)), w_bytes = (unsigned)(3u * C * C * sizeof(bf16_t));
    const __amdgpu_buffer_rsrc_t rA = make_uniform_rsrc(p.A, (int)t_bytes), rR = make_uniform_rsrc(p.res, (int)t_bytes);
    CHAIN_STAMP(0);

    // ---- everything read before the first MFMA is requested now: the activated tile with its halo, the raw tile, the first B units
    constexpr int NAU = (BR * PCS + NT - 1) / NT, NRU = MROWS * PCS / NT;
    static_assert(NRU * NT == MROWS * PCS, "raw tile tiling");
    uint4 ra[NAU], rr[BWD ? 1 : NRU];
#pragma unroll
    for (int u = 0; u < NAU; ++u) {
        const int idx = tid + NT * u, row = idx / PCS, pc = idx % PCS;
        const long long r = g0 - HMAXP + row;
        ra[u] = ld16(rA, (idx < BR * PCS && r >= 0 && r < p.R) ? (unsigned)(((size_t)r * C + 8 * pc) * sizeof(bf16_t)) : OOB);
    }
    if constexpr (!BWD) {
#pragma unroll
        for (int u = 0; u < NRU; ++u) {
            const int idx = tid + NT * u, row = idx / PCS, pc = idx % PCS;
            const long long r = g0 + row;
            rr[u] = ld16(rR, (r >= 0 && r < p.R) ? (unsigned)(((size_t)r * C + 8 * pc) * sizeof(bf16_t)) : OOB);
        }
    }
    // BWD - the leaky' masks (bf16 activations of the forward, (R, C)) become BIT tiles in LDS: sBits[buf][row][piece] = one byte, bit e <->
    // channel 8 piece + e of tile row `row` has M > 0.  A thread fetches four 16-byte pieces (coalesced: 32 pieces = one row per half
    // wave) and turns each into its byte with integer compares on the packed words.  The mask of conv c + 1 is requested in front of
    // conv c's loop and committed (bytes -> LDS) in conv c's epilogue, into the buffer conv c's epilogue is not reading.
    // (First version: lane = row, ballots = column words: four loads of 64 different cache lines per thread and mask competed with the
    // weight stream for the CU's vector-memory path - 45 us per launch against 35 us for the unmasked chain.)
    unsigned char *sBits = reinterpret_cast<unsigned char *>(smem + 2 * BR * RS);
    constexpr int NMU = MROWS * PCS / NT;
    uint4 mq[BWD ? NMU : 1];
    auto mask_fetch = [&](const bf16_t *M) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rM = make_uniform_rsrc(M, (int)t_bytes);
#pragma unroll
        for (int u = 0; u < NMU; ++u) {
            const int idx = tid + NT * u, row = idx / PCS, pc = idx % PCS;
            const long long r = g0 + row;
            mq[u] = ld16(rM, (r >= 0 && r < p.R) ? (unsigned)(((size_t)r * C + 8 * pc) * sizeof(bf16_t)) : OOB);
        }
    };
    auto mask_commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NMU; ++u) {
            const unsigned *w = reinterpret_cast<const unsigned *>(&mq[u]);
            unsigned byte = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {       // a bf16 is > 0 iff its sign is clear and it is not zero (NaN payloads aside)
                byte |= ((int)(w[e] << 16) > 0 ? 1u : 0u) << (2 * e);
                byte |= ((int)w[e] > 0xffff ? 1u : 0u) << (2 * e + 1);
            }
            sBits[buf * (MROWS * PCS) + tid + NT * u] = (unsigned char)byte;
        }
    };
    if constexpr (BWD) mask_fetch(p.d[0].M1);
    uint4 rb[RU];
    const unsigned fwave = (unsigned)wave * (unsigned)KSTEPS * 1024u + (unsigned)lane * 16u;
    constexpr unsigned FTAP = (unsigned)(C / 32) * (unsigned)KSTEPS * 1024u;
    auto fetch_b = [&](auto slotc, __amdgpu_buffer_rsrc_t rW, int it) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value, tap = slot % 3, ksl = slot / 3;
        const int ks = (RU / 3) * it + ksl;                                   // uniform
        rb[slot] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(ks < KSTEPS ? fwave : OOB),
                                                                                    (int)((unsigned)tap * FTAP + (unsigned)ks * 1024u), 0));
    };
    {
        const __amdgpu_buffer_rsrc_t rW = make_uniform_rsrc(p.d[0].W1, (int)w_bytes);
        static_for<0, RU>([&](auto uc) __attribute__((always_inline)) { fetch_b(uc, rW, 0); });
    }
    if (tid < MROWS) {
        const long long r = g0 + tid;
        const int l = (int)(((r % p.Lp) + p.Lp) % p.Lp);
        s_in[tid] = (r >= 0 && r < p.R && l >= p.HP && l < p.HP + p.L) ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < NAU; ++u) {
        const int idx = tid + NT * u;
        if (idx < BR * PCS) *reinterpret_cast<uint4 *>(sXa + (idx / PCS) * RS + 8 * (idx % PCS)) = ra[u];
    }
    if constexpr (!BWD) {
#pragma unroll
        for (int u = 0; u < NRU; ++u) {
            const int idx = tid + NT * u;
            *reinterpret_cast<uint4 *>(sXr + (idx / PCS) * RS + 8 * (idx % PCS)) = rr[u];
        }
    } else {
        mask_commit(0);
    }
    // the rims of the mid tile are never computed: zero once (they reach only rows outside the owned range)
    for (int idx = tid; idx < 2 * HMAXP * PCS; idx += NT) {
        const int row = idx / PCS < HMAXP ? idx / PCS : MROWS + idx / PCS;
        *reinterpret_cast<uint4 *>(sM + row * RS + 8 * (idx % PCS)) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    CHAIN_STAMP(1);
    // the residual stream at this lane's accumulator elements, as the bf16 values a pair launch would read back
    float xres[MB][16];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = m * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
            xres[m][q] = bf2f(BWD ? sXa[(HMAXP + i) * RS + col0] : sXr[i * RS + col0]);     // BWD: the residual IS the input gradient
        }

    f32x16 acc[MB];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    };
    // one conv over the haloed tile `src` (tile row i <-> buffer row i + HMAXP); the unit order is that of conv_pair_body::run_conv
    // rWn: the pack of the NEXT conv - the last ring turn refills with its first units instead of out-of-range zeros, so the next loop
    // starts with a full ring and no wave spends ~700 cycles issuing twelve 1 KB loads between two loops
    auto run_conv = [&](const bf16_t *src, int off0, int dstep, __amdgpu_buffer_rsrc_t rW, __amdgpu_buffer_rsrc_t rWn, unsigned need) __attribute__((always_inline)) {
        wait_flag(0, need);                                  // channels 0-127 of the source tile are written
        const bf16_t *ab[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) ab[t] = src + (li + HMAXP + off0 + t * dstep) * RS + 8 * kg;
        auto afrag = [&](auto uc, int it, bf16x8 (&x)[MB]) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value, tap = u % 3, ksl = u / 3;
            const bf16_t *pa = ab[tap] + 16 * ((RU / 3) * it + ksl);
#pragma unroll
            for (int m = 0; m < MB; ++m) x[m] = *reinterpret_cast<const bf16x8 *>(pa + m * 32 * RS);
        };
        // A fragments run AD units ahead of their MFMAs in a ring of AD + 1
        bf16x8 xr[AR][MB];
        // the ring loop over turns [it0, it1); nothing conditional inside (a uniform test in the unrolled body splits its basic block
        // and costs ~1.8 k cycles per conv: measured with the group-1 gate placed inside the loop)
        auto turns = [&](int it0, int it1) __attribute__((always_inline)) {
            static_for<0, AR - 1>([&](auto uc) __attribute__((always_inline)) { afrag(uc, it0, xr[decltype(uc)::value]); });
#pragma unroll 1
            for (int it = it0; it < it1; ++it) {
                static_for<0, RU>([&](auto uc) __attribute__((always_inline)) {
                    constexpr int u = decltype(uc)::value;
                    afrag(std::integral_constant<int, u + AR - 1>{}, it, xr[(u + AR - 1) % AR]);
                    const bf16x8 b = __builtin_bit_cast(bf16x8, rb[u]);
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xr[u % AR][m], b, acc[m], 0, 0, 0);
                    fetch_b(uc, it + 1 < UNITS / RU ? rW : rWn, it + 1 < UNITS / RU ? it + 1 : 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, MB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        };
        if constexpr (FLAGS) {
            // k-steps 0-7 (channels 0-127), then - once group 1 has written its half - k-steps 8-15.  The A prefetch of the first
            // half's last units reads two fragments of the second half early: they are read again behind the gate.
            turns(0, UNITS / RU / 2);
            wait_flag(1, need);
            turns(UNITS / RU / 2, UNITS / RU);
        } else {
            turns(0, UNITS / RU);
        }
    };
    // owned rows of a tile in LDS -> memory, 16 bytes per store.  Done by waves 0-3 only: a SIMD issues its OLDER wave first, so the
    // matrix pipe serves waves 0-3 until their loop ends and waves 4-7 finish ~3 k cycles later (tools/trace_chain.py) - the copy costs
    // the early waves nothing and would sit on the critical path of the late ones.
    auto copy_out = [&](const bf16_t *src, int row_off, bf16_t *dst) __attribute__((always_inline)) {
        if (wave >= 4) return;
        const __amdgpu_buffer_rsrc_t rO = make_uniform_rsrc(dst, (int)t_bytes);
        for (int idx = tid; idx < p.ts * PCS; idx += NT / 2) {
            const int i = p.lo + idx / PCS, pc = idx % PCS;
            const long long r = g0 + i;
            const uint4 v = *reinterpret_cast<const uint4 *>(src + (row_off + i) * RS + 8 * pc);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rO,
                                                   (int)(r < p.R ? (unsigned)(((size_t)r * C + 8 * pc) * 2) : OOB), 0, 0);
        }
    };

    // rows of this lane's accumulator elements that lie inside a clip, as AND masks on the packed bf16 pair of rows (i, i + 1):
    // mw[m][q / 2] for accumulator elements q, q + 1 of row block m
    unsigned mw[MB][8];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const int i = m * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
            mw[m][q / 2] = (s_in[i] ? 0xffffu : 0u) | (s_in[i + 1] ? 0xffff0000u : 0u);
        }
    // The epilogues are branch-free and read nothing but registers: the first version tested `v > 0` with the slope read from the kernel
    // arguments inside the branch and s_in[row] from LDS per element - 32 dependent (scalar load -> wait -> multiply) chains per wave,
    // 9 k cycles per epilogue against 5 k for the conv loop in front of it (tools/trace_chain.py).  leaky(v) = max(v, slope v) for
    // 0 <= slope <= 1 (the launcher checks): the same values as the select of psnd_conv1d_cl_pair, two instructions less per element.
    // the descriptor and the bias values of pair pp + 1 are requested while pair pp runs (a scalar-cache miss and two global loads in
    // front of every first conv otherwise)
    bf16_t *prev_raw = nullptr, *prev_act = nullptr;
    ChainPair d = p.d[0];
    float b1 = d.bias1 ? d.bias1[col0] : 0.f, b2 = d.bias2 ? d.bias2[col0] : 0.f;
    // BWD: v *= (M > 0 ? 1 : slope) from the bit tile `buf`: the element's byte, its bit shifted to the sign and spread, selects between
    // v and slope * v
    const int bsh = 31 - (col0 & 7);
    auto masked = [&](float v, float slope, int buf, int row) __attribute__((always_inline)) {
        const unsigned byte = sBits[buf * (MROWS * PCS) + row * PCS + (col0 >> 3)];
        const unsigned sel = (unsigned)(((int)(byte << bsh)) >> 31);
        return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v) & sel) | (__builtin_bit_cast(unsigned, v * slope) & ~sel));
    };
#pragma unroll 1
    for (int pp = 0; pp < p.n_pairs; ++pp) {
        const ChainPair dn = p.d[pp + 1 < p.n_pairs ? pp + 1 : pp];
        const float b1n = dn.bias1 ? dn.bias1[col0] : 0.f, b2n = dn.bias2 ? dn.bias2[col0] : 0.f;
        const float sl1 = d.act1_slope, sl2 = d.act2_slope, ms1 = d.m1_slope, ms2 = d.m2_slope;
        const __amdgpu_buffer_rsrc_t rW1 = make_uniform_rsrc(d.W1, (int)w_bytes), rW2 = make_uniform_rsrc(d.W2, (int)w_bytes);
        const __amdgpu_buffer_rsrc_t rWn = make_uniform_rsrc(dn.W1, pp + 1 < p.n_pairs ? (int)w_bytes : 0);
        // ---- first conv -> mid (bf16) in LDS
        if constexpr (BWD) mask_fetch(d.M2);                 // for the second conv's epilogue
        zero_acc();
        run_conv(sXa, d.off1, d.dstep1, rW1, rW2, 8u * pp);
        if constexpr (FLAGS) __builtin_amdgcn_s_setprio(PSND_CHAIN_EPRIO);
        CHAIN_STAMP(2 + 4 * pp);
        // Copy-outs are issued BEHIND a conv loop, not in front of it: the vector-memory counter is in order, so a loop whose first
        // ring refill waits behind 5 stores waits for their write acknowledgements (~2 k cycles, tools/trace_chain.py).  The tiles they
        // read stay intact until the epilogue after the NEXT loop.
        if (pp > 0) {
            if (prev_raw) copy_out(BWD ? sXa : sXr, BWD ? HMAXP : 0, prev_raw);
            if (!BWD && prev_act) copy_out(sXa, HMAXP, prev_act);
        }
        if constexpr (BWD) {
            // conv 2 pp reads buffer 0, conv 2 pp + 1 buffer 1
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const int rho = (q & 3) + 8 * (q >> 2), i = m * 32 + rho + 4 * kg;
                    const float v0 = masked(acc[m][q] + b1, ms1, 0, i), v1 = masked(acc[m][q + 1] + b1, ms1, 0, i + 1);
                    const unsigned w = pack_bf16(v0, v1) & mw[m][q / 2];
                    sM[(HMAXP + i) * RS + col0] = (bf16_t)(w & 0xffffu);
                    sM[(HMAXP + i + 1) * RS + col0] = (bf16_t)(w >> 16);
                }
            mask_commit(1);
        } else {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const int i = m * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;          // rows i, i + 1
                    const float v0 = acc[m][q] + b1, v1 = acc[m][q + 1] + b1;
                    const unsigned w = pack_bf16(fmaxf(v0, v0 * sl1), fmaxf(v1, v1 * sl1)) & mw[m][q / 2];
                    sM[(HMAXP + i) * RS + col0] = (bf16_t)(w & 0xffffu);
                    sM[(HMAXP + i + 1) * RS + col0] = (bf16_t)(w >> 16);
                }
        }
        if (pp == 1) CHAIN_STAMP(14);
        if constexpr (FLAGS) {
            signal_flag();
            __builtin_amdgcn_s_setprio(0);
        } else {
            __syncthreads();
        }
        CHAIN_STAMP(3 + 4 * pp);
        // ---- second conv on mid; its output replaces the activated tile and the residual stream
        if constexpr (BWD) mask_fetch(dn.M1);                // for the next pair's first conv (the last pair: fetched, never read)
        zero_acc();
        run_conv(sM, d.off2, d.dstep2, rW2, rWn, 8u * pp + 4u);
        if constexpr (FLAGS) __builtin_amdgcn_s_setprio(PSND_CHAIN_EPRIO);
        CHAIN_STAMP(4 + 4 * pp);
        if (d.mid_out) copy_out(sM, HMAXP, d.mid_out);
        if constexpr (BWD) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const int rho = (q & 3) + 8 * (q >> 2), i = m * 32 + rho + 4 * kg;
                    const float v0 = masked(acc[m][q] + b2, ms2, 1, i) + xres[m][q];
                    const float v1 = masked(acc[m][q + 1] + b2, ms2, 1, i + 1) + xres[m][q + 1];
                    const unsigned w = pack_bf16(v0, v1) & mw[m][q / 2];
                    xres[m][q] = __builtin_bit_cast(float, w << 16), xres[m][q + 1] = __builtin_bit_cast(float, w & 0xffff0000u);
                    sXa[(HMAXP + i) * RS + col0] = (bf16_t)(w & 0xffffu);
                    sXa[(HMAXP + i + 1) * RS + col0] = (bf16_t)(w >> 16);
                }
            mask_commit(0);
        } else {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const int i = m * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
                    const float v0 = acc[m][q] + b2 + xres[m][q], v1 = acc[m][q + 1] + b2 + xres[m][q + 1];
                    const unsigned w = pack_bf16(v0, v1) & mw[m][q / 2];
                    xres[m][q] = __builtin_bit_cast(float, w << 16), xres[m][q + 1] = __builtin_bit_cast(float, w & 0xffff0000u);
                    sXr[i * RS + col0] = (bf16_t)(w & 0xffffu);
                    sXr[(i + 1) * RS + col0] = (bf16_t)(w >> 16);
                    const unsigned wa = pack_bf16(fmaxf(v0, v0 * sl2), fmaxf(v1, v1 * sl2)) & mw[m][q / 2];
                    sXa[(HMAXP + i) * RS + col0] = (bf16_t)(wa & 0xffffu);
                    sXa[(HMAXP + i + 1) * RS + col0] = (bf16_t)(wa >> 16);
                }
        }
        if (pp == 1) CHAIN_STAMP(15);
        if constexpr (FLAGS) {
            signal_flag();
            __builtin_amdgcn_s_setprio(0);
        } else {
            __syncthreads();
        }
        CHAIN_STAMP(5 + 4 * pp);
        prev_raw = d.out_raw, prev_act = d.out_act;
        d = dn, b1 = b1n, b2 = b2n;
    }
    if constexpr (FLAGS) __syncthreads();                  // every wave's last epilogue is in the tiles
    if (prev_raw) copy_out(BWD ? sXa : sXr, BWD ? HMAXP : 0, prev_raw);
    if (!BWD && prev_act) copy_out(sXa, HMAXP, prev_act);
}

int chain_loss(const int *taps, int n_pairs) {          // rows lost per side: every conv's reach but the first's
    int loss = 0;
    for (int i = 0; i < n_pairs; ++i) loss += reach3(taps[4 * i], taps[4 * i + 1]) + reach3(taps[4 * i + 2], taps[4 * i + 3]);
    return loss - reach3(taps[0], taps[1]);
}
}  // namespace

extern "C" int psnd_conv1d_cl_chain_rows(int C, int k, int n_pairs, const int *taps) {
    if (k != 3 || C != 256 || n_pairs < 1 || n_pairs > CHAIN_MAX || !taps) return 0;
    for (int i = 0; i < 2 * n_pairs; ++i)
        if (reach3(taps[2 * i], taps[2 * i + 1]) > HMAXP) return 0;
    const int ts = 64 - 2 * chain_loss(taps, n_pairs);
    return ts >= 16 ? ts : 0;
}

// Row tile of a launch over R rows: 64 computed rows per workgroup (mr = 2); PSND_CHAIN_MR=1 selects 32 (A/B runs).  Returns the owned
// rows, 0 = unsupported.
extern "C" int psnd_conv1d_cl_chain_plan(int C, int k, int n_pairs, const int *taps, int64_t R, int *mr_out) {
    const int ts2 = psnd_conv1d_cl_chain_rows(C, k, n_pairs, taps);
    if (!ts2) return 0;
    const int ts1 = ts2 - 32;
    // measured back to back at 32 x 173 frames: one pair on 32-row tiles 14.2 us (the pair kernel: 14.1), pairs (5, 1) 24.7 us, pairs
    // (1, 3) on 64-row tiles 26.7 us, a whole block (1, 3, 5) on 64-row tiles 38.8 us = 12.9 us per pair - the best per pair; 32-row tiles
    // are therefore only taken on request (a launch pays ~6 us beside its workgroups' lifetime, whatever the tile)
    (void)R;
    int mr = 2;
    if (const char *e = PSND_ENV("PSND_CHAIN_MR")) {
        const int f = atoi(e);
        if (f == 2 || (f == 1 && ts1 >= 8)) mr = f;
    }
    if (mr_out) *mr_out = mr;
    return mr == 1 ? ts1 : ts2;
}

extern "C" int psnd_conv1d_cl_chain(const void *A, const void *res, const psnd_chain_pair *pairs, int n_pairs, int64_t N, int Lp, int L, int HP,
                                    int C, int k, void *stream) {
    if (!A || !res || !pairs) PSND_FAIL(PSND_E_ARG, "conv1d_cl_chain: null pointer");
    const bool bwd = n_pairs >= 1 && pairs[0].M1 != nullptr;
    if (N < 0 || Lp <= 0 || L <= 0 || HP < 0 || Lp < L + HP) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_chain: N=%lld Lp=%d L=%d HP=%d", (long long)N, Lp, L, HP);
    if (n_pairs < 1 || n_pairs > CHAIN_MAX) PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_chain: %d pairs (1 ... %d)", n_pairs, CHAIN_MAX);
    int taps[4 * CHAIN_MAX];
    for (int i = 0; i < n_pairs; ++i) {
        taps[4 * i] = pairs[i].off1, taps[4 * i + 1] = pairs[i].dstep1, taps[4 * i + 2] = pairs[i].off2, taps[4 * i + 3] = pairs[i].dstep2;
        if (!pairs[i].W1 || !pairs[i].W2) PSND_FAIL(PSND_E_ARG, "conv1d_cl_chain: pair %d without weights", i);
        if (bwd != (pairs[i].M1 != nullptr) || bwd != (pairs[i].M2 != nullptr))
            PSND_FAIL(PSND_E_ARG, "conv1d_cl_chain: pair %d: masks on every conv of the chain (the input-gradient form) or on none", i);
        if (bwd && (res != A || pairs[i].act1_slope != 1.f || pairs[i].act2_slope != 1.f || pairs[i].out_act))
            PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_chain: pair %d: the masked (input-gradient) form has res == A, no activation, no out_act", i);
        if (!(pairs[i].act1_slope >= 0.f && pairs[i].act1_slope <= 1.f && pairs[i].act2_slope >= 0.f && pairs[i].act2_slope <= 1.f))
            PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_chain: activation slopes %g, %g of pair %d outside [0, 1]", pairs[i].act1_slope, pairs[i].act2_slope, i);
    }
    int mr = 2;
    const int ts = psnd_conv1d_cl_chain_plan(C, k, n_pairs, taps, N * (int64_t)Lp, &mr);
    if (!ts) PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_chain: C=%d k=%d, %d pairs: only k = 3, C = 256, reach <= %d, >= 16 rows left of 64", C, k, n_pairs, HMAXP);
    if (N == 0) return PSND_OK;
    if ((size_t)N * Lp * C * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_chain: operand larger than 2 GB");
    ChainParams p;
    p.A = static_cast<const bf16_t *>(A), p.res = static_cast<const bf16_t *>(res);
    p.R = N * (int64_t)Lp, p.Lp = Lp, p.L = L, p.HP = HP;
    p.n_pairs = n_pairs, p.ts = ts, p.lo = (32 * mr - ts) / 2;
    for (int i = 0; i < CHAIN_MAX; ++i) {
        const psnd_chain_pair &s = pairs[i < n_pairs ? i : n_pairs - 1];
        ChainPair &d = p.d[i];
        d.W1 = static_cast<const bf16_t *>(s.W1), d.W2 = static_cast<const bf16_t *>(s.W2), d.bias1 = s.bias1, d.bias2 = s.bias2;
        d.mid_out = static_cast<bf16_t *>(s.mid_out), d.out_raw = static_cast<bf16_t *>(s.out_raw), d.out_act = static_cast<bf16_t *>(s.out_act);
        d.off1 = s.off1, d.dstep1 = s.dstep1, d.off2 = s.off2, d.dstep2 = s.dstep2, d.act1_slope = s.act1_slope, d.act2_slope = s.act2_slope;
        d.M1 = static_cast<const bf16_t *>(s.M1), d.M2 = static_cast<const bf16_t *>(s.M2), d.m1_slope = s.m1_slope, d.m2_slope = s.m2_slope;
    }
    p.trace = nullptr;
#ifdef PSND_TRACE      // tools/trace_chain.py builds: a device pointer the kernel writes s_memtime stamps to - never in the product build
    {
        const char *tp = PSND_ENV("PSND_PAIR_TRACE_PTR");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
    }
#endif
    const int64_t tiles = (p.R + ts - 1) / ts;
    constexpr int RS = 256 + 8;
    // forward: act + mid tiles with rims and the raw tile; input-gradient form: the two tiles and two 2 KB bit tiles of the masks
    const int mrows = 32 * mr;
    const size_t lds = bwd ? (size_t)2 * (mrows + 2 * HMAXP) * RS * 2 + 2 * 256 * 8 : (size_t)(2 * (mrows + 2 * HMAXP) + mrows) * RS * 2;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kern) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, p);
    };
    if (bwd && mr == 2) launch(conv_chain_kernel<256, 2, true>);
    else if (bwd) launch(conv_chain_kernel<256, 1, true>);
    else if (mr == 2) launch(conv_chain_kernel<256, 2, false>);
    else launch(conv_chain_kernel<256, 1, false>);
    PSND_CHECK_LAUNCH("conv1d_cl_chain");
    g_conv_chain_stats[0]++;
    g_conv_chain_stats[1] += n_pairs;
    return PSND_OK;
}

extern "C" void psnd_conv_chain_stats(long long *out2) {
    out2[0] = g_conv_chain_stats[0].load(), out2[1] = g_conv_chain_stats[1].load();
}
