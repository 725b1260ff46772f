// psnd_stft_w.hip - n_fft = 4096 forward STFT, magnitude output, third generation: ONE WAVE PER FRAME, 16 frames per workgroup.
//
// Replaces STFT.transform (pytorch_sound/models/transforms.py:53-69) for BASELINE config 5 (44.1 kHz, 4096 / 1024, 30 s clips).
//
// Why a third kernel.  The output is (N, K, F) with the FRAME axis fastest (the reference's conv1d layout), so a workgroup that
// owns f frames writes 4 f bytes per bin row.  stft_fwd_n4096b_kernel (psnd_stft.hip) keeps the frames of a tile in LDS between its
// passes (66 KB for 4 frames) and therefore writes 16-byte runs: 2049 partial lines per tile, which merge in the L2 only when the 8
// neighbouring workgroups write at the same moment - measured 1.94 x the algorithmic write bytes, 35 % of the write requests leaving
// the L2 as 32-byte partials, 243 us for 508 MB (0.26 of 8 TB/s) against 141 us with the stores ablated; forcing the neighbours to
// rendezvous cleans the traffic (1.07 x) but costs more than it returns (profiles/r03_stft4096b_rendezvous_experiment.txt).
// A 64-byte run needs 16 frames of a bin row in ONE store instruction, and 16 frames of spectrum (131 KB) do not fit next to any
// LDS-resident FFT of 4 or more frames.  Here the FFT lives in REGISTERS instead:
//
//   * a wave owns one frame: 2048 packed complex points = 32 per lane (64 VGPRs); 16 waves = 16 consecutive frames of a clip =
//     one workgroup of 1024 threads per CU, 4 waves per SIMD, <= 128 VGPRs;
//   * no workgroup barrier inside the transform - the only exchanges are wave-private (a v_permlane32_swap round, one 32 x 32
//     transpose per half-wave through an 8.4 KB LDS buffer of the wave, 32 ds_bpermute for the real-FFT partners) - so the 16
//     waves drift apart and their load / butterfly / LDS phases overlap freely;
//   * the magnitudes of the 16 frames meet in an LDS staging tile [16 frames][2049 bins] (131 KB, aliased onto the exchange buffers,
//     which are idle by then) and leave as 64-byte runs: 16 rows x 64 B per store instruction;
//   * the samples of the NEXT tile are requested into the dead data registers right after the split, i.e. ahead of this tile's
//     stores in the in-order vector-memory queue, and travel during the staging / store phases.
//
// Transform of one frame (z[n] = 0.5 w (x[2n], x[2n+1]), n < 2048; Z = FFT_2048(z); lane = (lam = lane & 31, g = lane >> 5)):
//   radix-2, in lane   lane (lam, g) holds n = lam + 32 (a + 16 g) and n + 1024, a < 16:  u = z[n] + z[n+1024],
//                      v = (z[n] - z[n+1024]) W_2048^n  (= d W_64^a cL, cL = W_2048^lam (-i)^g a per-lane constant)
//   v_permlane32_swap  lanes < 32 collect all u (-> even bins), lanes >= 32 all v (-> odd bins): y[a'] = seq[lam + 32 a'], a' < 32
//   radix-32, in lane  Y[q1] = sum_a' y[a'] W_32^(a' q1), times W_1024^(lam q1) (table in LDS)
//   transpose          per half-wave, through the wave's LDS buffer (rows q1, pitch 33): lane m receives Y'[lam = 0..31][q1 = m]
//   radix-32, in lane  Zh[m + 32 q2], i.e. bin k = 2 (m + 32 q2) + h = 64 q2 + c,  c = 2 m + h
//   real split         X[k], X[C - k] from Z[k], Z[C - k]: the partner lane ((32 - m) % 32 | 31 - m) sends its upper 16 values
//                      (ds_bpermute); a lane then owns bins 64 j + c (j < 16: rows 0..1023) and their mirrors 2048 - 64 j - c
//                      (rows 1025..2048); lane 0 pairs inside itself and owns bins 0 | 2048 and 1024
//
// Bound: HBM (4 hop + 4 K = 12 292 B per frame), see DESIGN.md 4.1 for the measured fraction.
#include "psnd_pk.h"
#include "psnd_stft_pass.h"
#include "psnd_stft_w.h"
#include <stdlib.h>

#ifndef PSND_W_NFK_AUX
#define PSND_W_NFK_AUX 2     // cache-policy bits of the (N, F, K) stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  nt: 188 against 205 us (config 5, same box)
#endif
#ifndef PSND_W_NFK_SYNC
#define PSND_W_NFK_SYNC 0
#endif
#ifndef PSND_W_SKIP
#define PSND_W_SKIP 0        // register-pressure bisection only: bit k leaves stage k of the transform out
#endif

namespace {
using namespace psnd_stft;

constexpr int kC = 2048, kNFFT = 4096, kK = 2049;
constexpr int kFrames = 16;                       // frames per workgroup = waves per workgroup
constexpr int kStgP = 2052;                       // staging pitch per frame (floats, 2049 bins): 4 P = 16 (mod 32) keeps the flush reads conflict-free
constexpr int kXP = 33;                           // exchange row pitch (complex values)
constexpr int kXaFloats = 32 * kXP * 2;           // one wave's exchange buffer (one half-wave at a time)
constexpr int kLdsFloats = 4096 + 2048 + kFrames * kXaFloats + 260;      // window | inter-pass twiddles | exchange / staging | cL, v_c per lane
static_assert(kFrames * kXaFloats >= kFrames * kStgP, "the staging tile [16 frames][2049 bins] must fit the exchange area");
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");

struct WParams {
    const float *wav;
    const float *plan;
    float *mag;
    long long T, F;
    int hop, pad, ntile, total_tiles;
    float mag_eps;
    int ablate;                                   // debug (PSND_ABLATE): 2 = no global stores
    int stagger;                                  // s_sleep units between the four waves of a SIMD at the top of a transform
#ifdef PSND_W_DEBUG
    float *dbg;                                   // tools/dbg_w.py: z of wave DBGW after every stage, [stage][lane][32][2]
    long long *trace;                             // tools/trace_w.py: s_memtime stamps [block][wave][16] of tile iteration trace_iter
    int trace_iter;
#endif
};
#ifdef PSND_W_DEBUG
#define PSND_W_DUMP(st_)                                                                                        \
    do {                                                                                                       \
        if (p.dbg && blockIdx.x == 0 && w == 3 && tile == tw.first)                                            \
            for (int i_ = 0; i_ < 32; ++i_) {                                                                  \
                p.dbg[(((st_) * 64 + lane) * 32 + i_) * 2] = z[i_].x;                                          \
                p.dbg[(((st_) * 64 + lane) * 32 + i_) * 2 + 1] = z[i_].y;                                      \
            }                                                                                                  \
    } while (0)
#define PSND_W_STAMP(i_)                                                                                       \
    do {                                                                                                       \
        if (p.trace && lane == 0 && titer == p.trace_iter)                                                     \
            p.trace[((size_t)blockIdx.x * 16 + w) * 16 + (i_)] = __builtin_amdgcn_s_memtime();                  \
    } while (0)
#else
#define PSND_W_DUMP(st_)
#define PSND_W_STAMP(i_)
#endif

// v * W_N^J (forward sign), compile-time twiddle
template <int J, int N>
__device__ __forceinline__ v2f cmul_ct(v2f t) {
    if constexpr (J % N == 0) {
        return t;
    } else if constexpr (4 * J == N) {            // -i
        return pk::swp(t) * v2f{1.f, -1.f};
    } else if constexpr (8 * J == N) {            // (1 - i) / sqrt 2
        constexpr float r = (float)ct::cos2pi(1, 8);
        return pk::fma(pk::swp(t), v2f{1.f, -1.f}, t) * v2f{r, r};
    } else if constexpr (8 * J == 3 * N) {        // (-1 - i) / sqrt 2
        constexpr float r = (float)ct::cos2pi(1, 8);
        return pk::fma(pk::swp(t), v2f{-1.f, 1.f}, t) * v2f{-r, -r};
    } else {
        constexpr float c = (float)ct::cos2pi(J, N), s = (float)ct::sin2pi(J, N);
        return pk::fma(pk::swp(t), v2f{s, -s}, t * v2f{c, c});
    }
}

__device__ __forceinline__ int stg_pi(int r) { return r ^ ((r >> 5) & 1); }

__device__ __forceinline__ float bperm(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// NFK = true: the bin-fastest output (N, F, K) of psnd_stft_mag_nfk - a frame's spectrum is ONE contiguous 8196-byte run, so every
// wave stores its own frame straight from registers (a v_permlane32_swap pairs the even / odd bins of the two half-waves: 8-byte
// stores, 256 contiguous bytes per half-wave and instruction): no staging tile, no workgroup barrier inside the tile loop - the 16
// waves of the workgroup share nothing but the tables.
template <bool ALIGNED4, bool NFK = false>
__global__ __launch_bounds__(1024, 1) void stft_fwd_n4096w_kernel(WParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_win = smem;                          // [32 loads][64 lanes] of (0.5 w[2n], 0.5 w[2n+1]) in the order a lane loads its samples
    float *s_tw = smem + 4096;                    // W_1024^(lam q1) as [q1][lam] (re, im)
    float *s_xa = smem + 6144;                    // 16 exchange buffers; later the staging tile [16][kStgP]
    float *s_cl = smem + 6144 + kFrames * kXaFloats;      // per lane: cL = W_2048^lam (-i)^g, then v_c = -i W_4096^c (c = 2 lam + g)
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);                            // wave = frame of the tile (scalar)
    const float *plan = p.plan;

    // ---- tables --------------------------------------------------------------------------------------------------------
    for (int e = t; e < 2048; e += 1024) {
        const int i = e >> 6, ln = e & 63;
        const int n = (ln & 31) + 32 * ((i & 15) + 16 * (ln >> 5)) + 1024 * (i >> 4);
        const f32x2 wv = *reinterpret_cast<const f32x2 *>(plan + 2 * n);             // plan[0 .. 4096) = the raw window
        *reinterpret_cast<f32x2 *>(s_win + 2 * e) = f32x2{0.5f * wv.x, 0.5f * wv.y};
    }
    if (t < 512) reinterpret_cast<f32x4 *>(s_tw)[t] = reinterpret_cast<const f32x4 *>(plan + kW4096TwOff)[t];
    if (t < 64) {
        *reinterpret_cast<v2f *>(s_cl + 2 * t) = *reinterpret_cast<const v2f *>(plan + kW4096ClOff + 2 * t);
        const int c = 2 * (t & 31) + (t >> 5);
        *reinterpret_cast<v2f *>(s_cl + 128 + 2 * t) = *reinterpret_cast<const v2f *>(plan + kW4096VkOff + 2 * c);
        if (t == 0) *reinterpret_cast<v2f *>(s_cl + 256) = *reinterpret_cast<const v2f *>(plan + kW4096VkOff + 2 * 1024);   // v_(C/2)
    }
    float *xa = s_xa + w * kXaFloats;
    // Per-lane addresses are NOT kept across the tile loop: each phase derives its own from a lane id laundered through an empty asm
    // (the compiler cannot hoist them), so that besides the frame itself only `lane` lives across the transform.  Hoisted, they are
    // ~25 more registers than the 128 there are, i.e. scratch spills - and a scratch reload issued behind a global store waits for
    // that store to be acknowledged (in-order vmcnt): measured 2-3 k cycles each in the store phase.
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // the lane id, two instructions: never kept
        asm volatile("" : "+v"(ln));
        return ln;
    };

    const TileWalk tw = tile_walk(p.total_tiles);
    const int hop = p.hop;
    const long long F = p.F;
    v2f z[32];                                    // the frame: first the samples (z[a] = n, z[16 + a] = n + 1024), then the transform in place

    // the wave's frame of a tile: 0 = none (past the last frame), 1 = interior (plain loads), 2 = clip edge (reflect indexing)
    auto frame_kind = [&](int tile_, long long &s0, const float *&x_) __attribute__((always_inline)) {
        const int clip_ = tile_ / p.ntile;
        const long long f = (long long)(tile_ - clip_ * p.ntile) * kFrames + w;
        x_ = p.wav + (size_t)clip_ * p.T;
        s0 = f * hop - p.pad;
        if (f >= F) return 0;
        return (s0 >= 0 && s0 + kNFFT <= p.T) ? 1 : 2;
    };
    auto request_interior = [&](const float *x_, long long s0) __attribute__((always_inline)) {
        const int ln = fresh_lane();
        const float *px = x_ + s0 + 2 * ((ln & 31) + 512 * (ln >> 5));               // sample 2 n at a = 0
        static_for<0, 16>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            z[a] = *reinterpret_cast<const f32x2 *>(px + 64 * a);
            z[16 + a] = *reinterpret_cast<const f32x2 *>(px + 64 * a + 2048);
        });
    };
    // clip edge (two frames at either end of a clip): the reflect-resolved samples bounce through the wave's exchange buffer, half a
    // frame at a time - a compact loop instead of 64 index computations in registers.  Only called while the buffer is the wave's own.
    auto request_edge = [&](const float *x_, long long s0) __attribute__((always_inline)) {
        const int ln = fresh_lane(), lam_ = ln & 31, g_ = ln >> 5;
        const int Ti = (int)p.T, b0 = (int)s0 + ln;
#pragma unroll 1
        for (int i = 0; i < 32; ++i) xa[i * 64 + ln] = x_[reflect_idx32(b0 + 64 * i, Ti)];
        static_for<0, 16>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            z[a] = *reinterpret_cast<const v2f *>(xa + 2 * (lam_ + 512 * g_) + 64 * a);
        });
#pragma unroll 1
        for (int i = 0; i < 32; ++i) xa[i * 64 + ln] = x_[reflect_idx32(b0 + 2048 + 64 * i, Ti)];
        static_for<0, 16>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            z[16 + a] = *reinterpret_cast<const v2f *>(xa + 2 * (lam_ + 512 * g_) + 64 * a);
        });
    };

    int kind = 0;
    {
        long long s0;
        const float *x_;
        if (tw.first < tw.end) {
            kind = frame_kind(tw.first, s0, x_);
            if (kind == 1) request_interior(x_, s0);
            else if (kind == 2) request_edge(x_, s0);
        }
    }
    __syncthreads();                              // tables visible

    int titer = -1;
    (void)titer;
    for (int tile = tw.first; tile < tw.end; tile += tw.step) {
        ++titer;
        PSND_W_STAMP(0);
        const int clip = tile / p.ntile;
        const int f0 = (tile - clip * p.ntile) * kFrames;
        float mlo[16], mhi[16], mext = 0.f;
        const bool had = kind != 0;
        if (p.stagger) {
            // the 16 waves leave the last barrier together and would walk the same instruction stream in lock step - all of them in the
            // butterflies, then all of them in the LDS - so the four waves of a SIMD (w, w + 4, w + 8, w + 12) start a quarter phase apart
            for (int i = 0; i < (w >> 2) * p.stagger; ++i) __builtin_amdgcn_s_sleep(8);
        }
        if (had) {
#if !(PSND_W_SKIP & 1)
            // ---- window, radix-2 (decimation in frequency) in lane ---------------------------------------------------------
            const int ln1 = fresh_lane();
            const v2f *winl = reinterpret_cast<const v2f *>(s_win) + ln1;
            const v2f cL = *reinterpret_cast<const v2f *>(s_cl + 2 * ln1);
            static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                static_for<0, 4>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = decltype(cc)::value * 4 + decltype(ac)::value;
                    const v2f lo = z[a] * winl[a * 64];
                    const v2f hi = z[16 + a] * winl[(16 + a) * 64];
                    z[a] = lo + hi;
                    z[16 + a] = pk::cmul(cmul_ct<a, 64>(lo - hi), cL);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
#endif
            PSND_W_DUMP(0);
#if !(PSND_W_SKIP & 2)
            // ---- lanes < 32 take every u, lanes >= 32 every v ---------------------------------------------------------------
            static_for<0, 16>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                // (the builtin, fed from the halves of 64-bit register pairs, came back with x == y on hipcc 7.2: explicit instruction;
                //  s_nop: two wait states between a VALU write of an operand and the swap)
                float ux = z[a].x, uy = z[a].y, vx = z[16 + a].x, vy = z[16 + a].y;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(ux), "+v"(vx), "+v"(uy), "+v"(vy));
                z[a] = v2f{ux, uy};
                z[16 + a] = v2f{vx, vy};
            });
#endif
            PSND_W_DUMP(1);
            PSND_W_STAMP(1);
#if !(PSND_W_SKIP & 4)
            // ---- first radix-32 + inter-pass twiddle ---------------------------------------------------------------------------
            pk::fft<32>(z);                       // Y[q1] in slot bitrev(q1)
            const int ln2 = fresh_lane();
            const v2f *twl = reinterpret_cast<const v2f *>(s_tw) + (ln2 & 31);
            static_for<0, 8>([&](auto cc) __attribute__((always_inline)) {
                static_for<0, 4>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q1 = decltype(cc)::value * 4 + decltype(qc)::value, s = ct::bitrev(q1, 5);
                    if constexpr (q1 != 0) z[s] = pk::cmul(z[s], twl[q1 * 32]);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
#endif
            PSND_W_DUMP(2);
            PSND_W_STAMP(2);
#if !(PSND_W_SKIP & 8)
            // ---- 32 x 32 transpose per half-wave through the wave's own buffer (LDS operations of one wave execute in order) -----
            // The reads are explicit instructions with the destination tied to z[]: as plain loads inside the divergent branch they
            // would be new values merged with the other half's registers by a phi - 64 more live VGPRs than the 128 there are.
            const int lam3 = ln2 & 31, g3 = ln2 >> 5;
            v2f *xr = reinterpret_cast<v2f *>(xa);
            const unsigned rd_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(xr + lam3 * kXP));
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (g3 == hh) {
                    static_for<0, 32>([&](auto sc) __attribute__((always_inline)) {
                        constexpr int s = decltype(sc)::value, q1 = ct::bitrev(s, 5);
                        xr[q1 * kXP + lam3] = z[s];
                    });
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (writes issued; the reads below are not tracked by the compiler)
#define PSND_W_RD(l2_) asm volatile("ds_read_b64 %0, %1 offset:%2" : "+v"(z[l2_]) : "v"(rd_addr), "n"((l2_) * 8))
#define PSND_W_RD4(b_) PSND_W_RD(b_); PSND_W_RD(b_ + 1); PSND_W_RD(b_ + 2); PSND_W_RD(b_ + 3)
                    PSND_W_RD4(0); PSND_W_RD4(4); PSND_W_RD4(8); PSND_W_RD4(12); PSND_W_RD4(16); PSND_W_RD4(20); PSND_W_RD4(24); PSND_W_RD4(28);
#undef PSND_W_RD4
#undef PSND_W_RD
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]),
                                   "+v"(z[9]), "+v"(z[10]), "+v"(z[11]), "+v"(z[12]), "+v"(z[13]), "+v"(z[14]), "+v"(z[15]));
                    asm volatile("" : "+v"(z[16]), "+v"(z[17]), "+v"(z[18]), "+v"(z[19]), "+v"(z[20]), "+v"(z[21]), "+v"(z[22]), "+v"(z[23]),
                                 "+v"(z[24]), "+v"(z[25]), "+v"(z[26]), "+v"(z[27]), "+v"(z[28]), "+v"(z[29]), "+v"(z[30]), "+v"(z[31]));
                }
            }
#endif
            PSND_W_DUMP(3);
            PSND_W_STAMP(3);
#if !(PSND_W_SKIP & 16)
            // ---- second radix-32: slot bitrev(q2) holds Zh[lam + 32 q2] = Z[64 q2 + c] ----------------------------------------------
            pk::fft<32>(z);
#endif
            PSND_W_DUMP(4);
            PSND_W_STAMP(4);
#if !(PSND_W_SKIP & 32)
            // ---- real-FFT split: own lower 16 (q2 = j) against the partner lane's upper 16 (q2 = 31 - j); lane 0 pairs q2 with 32 - q2
            //      inside itself (and owns the self-paired bin C/2).  Four pairs at a time: the partners' values are short-lived.
            const int ln4 = fresh_lane();
            const bool special = ln4 == 0;
            const v2f vL = *reinterpret_cast<const v2f *>(s_cl + 128 + 2 * ln4);      // v_c = -i W_4096^c
            const int paddr = ((ln4 >> 5) == 0 ? ((32 - ln4) & 31) : (95 - ln4)) * 4;  // partner lane of the split (byte address): (32 - lam) % 32 | 32 + (31 - lam)
            static_for<0, 8>([&](auto cc) __attribute__((always_inline)) {
                v2f zb[2];
                static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(cc)::value * 2 + decltype(jc)::value;
                    const v2f snd = z[ct::bitrev(31 - j, 5)];
                    const v2f own = z[ct::bitrev(j == 0 ? 0 : 32 - j, 5)];
                    const v2f got = v2f{bperm(paddr, snd.x), bperm(paddr, snd.y)};
                    zb[decltype(jc)::value] = special ? own : got;
                });
                static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(cc)::value * 2 + decltype(jc)::value;
                    const v2f za = z[ct::bitrev(j, 5)], zp = zb[decltype(jc)::value];
                    const v2f s = pk::fma(zp, v2f{1.f, -1.f}, za);                   // za + conj(zb)
                    const v2f d = pk::fma(zp, v2f{-1.f, 1.f}, za);                   // za - conj(zb)
                    const v2f e = pk::cmul(cmul_ct<j, 64>(d), vL);                   // d v_k,  v_k = v_c W_64^j
                    const v2f xk = s + e, xc = s - e;
                    mlo[j] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.x, xk.x, __builtin_fmaf(xk.y, xk.y, p.mag_eps)));
                    mhi[j] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xc.x, xc.x, __builtin_fmaf(xc.y, xc.y, p.mag_eps)));
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            {   // bin C/2 = 1024 (q2 = 16 of lane 0, self-paired): evaluated by every lane (no divergent branch inside the transform), kept by lane 0
                const v2f vMid = *reinterpret_cast<const v2f *>(s_cl + 256);
                const v2f mid = z[ct::bitrev(16, 5)];
                v2f xk, xc;
                rfft_pair_pk(mid, mid, vMid, xk, xc);
                mext = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.x, xk.x, __builtin_fmaf(xk.y, xk.y, p.mag_eps)));
            }
#endif
        }
        PSND_W_STAMP(5);
        if constexpr (NFK) {
            long long ns0 = 0;
            const float *nx = nullptr;
            const int nkind = tile + tw.step < tw.end ? frame_kind(tile + tw.step, ns0, nx) : 0;
            if (nkind == 1) request_interior(nx, ns0);            // ahead of this frame's stores in the in-order vector-memory queue
            if (had && !(PSND_ABL(p, 2))) {
                const int ln = fresh_lane();
                const int lam_ = ln & 31, g_ = ln >> 5;
                // frame f0 + w of the clip: K contiguous floats (wave-uniform base, range-checked by the descriptor)
                const __amdgpu_buffer_rsrc_t ro = make_uniform_rsrc(p.mag + ((size_t)clip * (size_t)F + (size_t)(f0 + w)) * kK, kK * 4);
                const int vlo = (64 * g_ + 2 * lam_) * 4;                      // bins 64 (j + g) + 2 lam, + 1
                const int vhi = (2047 - 64 * 14 - 64 * g_ - 2 * lam_) * 4;     // bins 2047 - 64 (j + g) - 2 lam, + 1   (j even: + 256 (14 - j) bytes)
                static_for<0, 8>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = 2 * decltype(jc)::value;
                    // lanes < 32 hold c = 2 lam, lanes >= 32 c = 2 lam + 1: after the swap a lane of the lower half owns the bin PAIR of
                    // row j, a lane of the upper half the pair of row j + 1
                    float a0 = mlo[j], a1 = mlo[j + 1], b0 = mhi[j], b1 = mhi[j + 1];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2f{a0, a1}), ro, vlo, 256 * j, PSND_W_NFK_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2f{b1, b0}), ro, vhi, 256 * (14 - j), PSND_W_NFK_AUX);
                });
                if (ln == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mext), ro, 1024 * 4, 0, 0);
            }
            if (nkind == 2) request_edge(nx, ns0);                // the exchange buffer is the wave's own at all times
            kind = nkind;
#if PSND_W_NFK_SYNC
            // the 16 waves of a workgroup read overlapping samples (hop = n / 4: every sample belongs to 4 frames): started together, the
            // later readers find them in the L2; drifting apart they go back to HBM (FETCH_SIZE 1.57 x the algorithmic reads without this)
            __syncthreads();
#endif
            continue;
        }
        // addresses of the staging / store phases (see fresh_lane)
        const int ln = fresh_lane();
        const bool special = ln == 0;
        const int cl2 = 2 * (ln & 31) + (ln >> 5);
        float *srow = s_xa + w * kStgP;               // this wave's row of the staging tile
        // staging positions: row r sits at pi(r) = r ^ bit 5 of r (bank spread); adding multiples of 64 commutes with pi
        float *slo = srow + stg_pi(cl2);                                             // row c;            row 64 j + c at slo[64 j]
        float *shi = srow + stg_pi(2048 - cl2) - 64 * 15;                            // row 2048 - c - 64 j at shi[64 (15 - j)]
        const int rr = ln & 15, fq = ln >> 4;                                        // flush: row within a group of 16, quad of frames
        const float *fsrc0 = s_xa + (4 * fq) * kStgP + rr, *fsrc1 = s_xa + (4 * fq) * kStgP + (rr ^ 1);
        const unsigned fdst = (unsigned)rr * (unsigned)p.F + 4u * (unsigned)fq;      // element offset of the lane inside a 16-row store
        // the wave's frame of the next tile
        long long ns0 = 0;
        const float *nx = nullptr;
        const int nkind = tile + tw.step < tw.end ? frame_kind(tile + tw.step, ns0, nx) : 0;

        float *oclip = p.mag + (size_t)clip * kK * (size_t)F + f0;                   // this clip's spectrogram at the tile's first frame (uniform)
        const bool nostore = PSND_ABL(p, 2);
        const int iF = (int)F;
        auto flush = [&](int nrows, int row_base) __attribute__((always_inline)) {
            // staging rows [0, nrows) -> bins row_base + r: every store instruction writes 16 rows x 64 bytes.  r = r0 + 16 it + rr:
            // bit 5 of r is it >> 1, so pi(r) = r0 + 16 it + (it < 2 ? rr : rr ^ 1) - two base addresses and immediates
            for (int r0 = 64 * w; r0 < nrows; r0 += 64 * kFrames) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (r0 + 16 * it + rr < nrows) {
                        const float *src = (it < 2 ? fsrc0 : fsrc1) + r0 + 16 * it;
                        const float v0 = src[0], v1 = src[kStgP], v2 = src[2 * kStgP], v3 = src[3 * kStgP];
                        float *dst = oclip + (size_t)(row_base + r0 + 16 * it) * (size_t)F + fdst;
                        if (!nostore) {
                            if constexpr (ALIGNED4) {
                                if (f0 + 4 * fq < iF) *reinterpret_cast<f32x4 *>(dst) = f32x4{v0, v1, v2, v3};
                            } else {
                                if (f0 + 4 * fq < iF) dst[0] = v0;
                                if (f0 + 4 * fq + 1 < iF) dst[1] = v1;
                                if (f0 + 4 * fq + 2 < iF) dst[2] = v2;
                                if (f0 + 4 * fq + 3 < iF) dst[3] = v3;
                            }
                        }
                    }
                }
            }
        };
        // The next tile's samples are requested HERE - as soon as this wave's transform is done, into the data registers, which are dead
        // until the next transform, and ahead of this tile's stores in the in-order vector-memory queue.  A CU can only keep so many
        // line fetches in flight (16 waves x 32 loads took 13-20 k cycles to ISSUE when all waves asked together behind the barrier);
        // asked for as each wave finishes, they travel under the other waves' butterflies.
        if (nkind == 1) request_interior(nx, ns0);
        __syncthreads();                          // every wave is through with its exchange buffer: the area becomes the staging tile
        PSND_W_STAMP(6);
        if (had) {
#pragma unroll
            for (int j = 0; j < 16; ++j) slo[64 * j] = mlo[j];                        // bin 64 j + c at pi(64 j + c) = 64 j + pi(c)
#pragma unroll
            for (int j = 0; j < 16; ++j) shi[64 * (15 - j)] = mhi[j];                 // bin 2048 - 64 j - c
            if (special) srow[1024] = mext;                                          // bin 1024 (pi(1024) = 1024)
        }
        PSND_W_STAMP(7);
        __syncthreads();
        PSND_W_STAMP(8);
        flush(kK, 0);                             // all 2049 bins: the whole spectrum of the 16 frames fits the exchange area (131 KB)
        PSND_W_STAMP(9);
        __syncthreads();                          // staging reads done: the exchange buffers are free again
        PSND_W_STAMP(14);
        if (nkind == 2) request_edge(nx, ns0);    // (rare) needs the wave's exchange buffer
        kind = nkind;
    }
}

}  // namespace

// plan tables of this kernel, behind the tables of the earlier 4096 kernels (psnd_stft_plan_build)
void psnd_stft4096w_plan_fill(float *plan) {
    const double two_pi = 6.283185307179586476925286766559;
    float *tw = plan + kW4096TwOff, *cl = plan + kW4096ClOff;
    for (int q1 = 0; q1 < 32; ++q1)
        for (int lam = 0; lam < 32; ++lam) {
            const double th = two_pi * (double)(lam * q1) / 1024.0;
            tw[2 * (q1 * 32 + lam)] = (float)cos(th);
            tw[2 * (q1 * 32 + lam) + 1] = (float)(-sin(th));
        }
    for (int lane = 0; lane < 64; ++lane) {
        const double th = two_pi * (double)(lane & 31) / 2048.0;
        const double c = cos(th), s = -sin(th);                                      // W_2048^lam
        if (lane < 32) cl[2 * lane] = (float)c, cl[2 * lane + 1] = (float)s;
        else cl[2 * lane] = (float)s, cl[2 * lane + 1] = (float)(-c);                // times -i
    }
}

bool psnd_stft4096w_ok(long long T, long long F, int hop, int pad) {
    (void)T;
    // even sample offsets (8-byte loads); 32-bit element offsets inside a clip's spectrogram
    return hop % 2 == 0 && pad % 2 == 0 && F > 0 && (long long)kK * F < (1ll << 31);
}

int psnd_stft4096w_launch(const float *wav, const float *plan, float *mag, long long N, long long T, long long F, int hop, int pad,
                          float mag_eps, int ablate, int nfk, hipStream_t stream) {
    WParams p;
    p.wav = wav, p.plan = plan, p.mag = mag, p.T = T, p.F = F, p.hop = hop, p.pad = pad, p.mag_eps = mag_eps, p.ablate = ablate;
    {
        const char *e = PSND_ENV("PSND_STFT4096_STAGGER");
        p.stagger = psnd_env_int(e, 0, 0, 64);
    }
#ifdef PSND_W_DEBUG
    {
        const char *e = PSND_ENV("PSND_W_DBG_PTR");
        p.dbg = e ? reinterpret_cast<float *>(strtoull(e, nullptr, 0)) : nullptr;
        const char *tp = PSND_ENV("PSND_W_TRACE_PTR"), *ti = PSND_ENV("PSND_W_TRACE_ITER");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
        p.trace_iter = ti ? atoi(ti) : 3;
    }
#endif
    const long long ntile = (F + kFrames - 1) / kFrames;
    if (ntile * N >= (1ll << 31)) PSND_FAIL(PSND_E_SHAPE, "stft_fwd(n4096w): too many tiles");
    p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
    int grid = p.total_tiles < 256 ? p.total_tiles : 256;                            // one persistent workgroup per CU
    if (const char *e = PSND_ENV("PSND_STFT4096_GRID")) grid = psnd_env_int(e, grid, 1, 65535);
    grid = (grid + 7) & ~7;
    constexpr size_t lds = sizeof(float) * kLdsFloats;
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_fwd(n4096w): set LDS size: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, stream, p);
        return PSND_OK;
    };
    const int rc = nfk ? launch(stft_fwd_n4096w_kernel<false, true>)
                       : ((F % 4 == 0) ? launch(stft_fwd_n4096w_kernel<true>) : launch(stft_fwd_n4096w_kernel<false>));
    if (rc != PSND_OK) return rc;
    PSND_CHECK_LAUNCH("stft_fwd(n4096w)");
    return PSND_OK;
}
