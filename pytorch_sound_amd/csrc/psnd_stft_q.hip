// psnd_stft_q.hip - n_fft = 1024 forward STFT, magnitude output in the BIN-FASTEST layout (N, F, K) of psnd_stft_mag_nfk:
// ONE WAVE OWNS FOUR CONSECUTIVE FRAMES from its samples to its stores.
//
// Replaces STFT.transform (pytorch_sound/models/transforms.py:53-69) where the consumer is a kernel of this library (mel projection,
// channels-last conv stack, spectral losses), which take either layout.
//
// Why another n = 1024 kernel.  stft_fwd_n1024_kernel (psnd_stft.hip) writes the reference's frame-fastest (N, K, F): its pass 2 spreads
// the bin rows of a 16-frame tile over the four waves of a workgroup so that a store instruction covers 16 frames x 4 bins = 64-byte
// runs - which needs the exchange between the passes to cross waves, i.e. five workgroup barriers per tile, 39 KB of LDS per 4 waves and
// one tile per workgroup (prologue latency exposed: profiles/r03_stft1024_phase_table.txt - a quarter of a workgroup's life is the load
// prologue, a quarter the half exchange with its barriers).  With the bin axis fastest a frame's spectrum is ONE contiguous 2052-byte
// run, so the lanes that hold a frame's bins can store it alone, whatever wave they are in:
//
//   * wave = 4 frames x 16 lanes in BOTH passes: pass 1 lane (frame, l) holds z[l + 16 a], a < 32 (window, radix-32 in registers,
//     inter-pass twiddle), pass 2 lane (frame, qq) holds the rows qq and 32 - qq (two radix-16, real-FFT split: post_emit_pk of the
//     span-staged kernel) - the exchange between the passes never leaves the wave: a 9.2 KB LDS buffer of its own, two half rounds,
//     LDS operations of one wave execute in order => NO workgroup barrier anywhere in the tile loop;
//   * the wave's span of samples (3 hop + 1024 = 7 KB) goes straight into the same LDS buffer by LDS-DMA one quad AHEAD (requested before
//     the current quad's stores in the in-order vector-memory queue; round 6);
//   * 16 waves = one persistent 1024-thread workgroup per CU share nothing but the tables (10.8 KB, loaded once per workgroup instead
//     of once per tile); they drift apart, so loads, butterflies, LDS rounds and stores of different waves overlap freely;
//   * stores: the quad's magnitudes are staged in the wave's buffer frame by frame (pitch 2112 B: the four lane groups of a staging write on
//     disjoint banks) and leave as 16-byte pieces - a store instruction writes 1 KiB of one frame's spectrum, nine instructions per quad
//     fill one contiguous 8.2 KB region that no other wave touches.
//
// Bound: HBM, 4 hop + 4 K = 3076 B per frame at hop 256 (DESIGN.md 4.1 for the measured fraction).
#include "psnd_pk.h"
#include "psnd_stft_pass.h"
#include "psnd_stft_emit.h"
#include "psnd_stft_q.h"
#include <stdlib.h>

#ifndef PSND_Q_STORE_AUX
#define PSND_Q_STORE_AUX 2    // cache-policy bits of the output stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  nt: whole lines written once, never
                              // re-read by this kernel - measured 148 against 162 us (same box, 1024 clips x 2 s); 17 / 18 / 19: 166 / 159 / 161
#endif
#ifndef PSND_Q_LOAD_MOD
#define PSND_Q_LOAD_MOD ""    // cache-policy modifiers of the span transfers (A/B builds: " nt", " sc1", " sc0 sc1")
#endif
#ifndef PSND_Q_PRIO_MEM
#define PSND_Q_PRIO_MEM 2     // wave priority of a quad's memory phase (read-out of the staged magnitudes, next span's transfer, stores) - round 6: every raised setting measures 1.5-2 % faster than none (0 / 0), same-box interleaved medians of 480 launches: 115.2 (0/0) 113.9 (0/2) 113.0 (0/1) 113.3 (3/1) 112.9 (3/2) 112.7 (2/1) us
#endif
#ifndef PSND_Q_PRIO_X
#define PSND_Q_PRIO_X 1       // ... of the exchange between the passes (LDS round trips)
#endif
#define Q_PRIO(n_) do { if ((PSND_Q_PRIO_MEM) != 0 || (PSND_Q_PRIO_X) != 0) __builtin_amdgcn_s_setprio(n_); } while (0)
#ifdef PSND_Q_NOSB            // A/B builds (tools/r04/variant_q.sh): no scheduling fences between the phases
#define Q_SB()
#else
#define Q_SB() __builtin_amdgcn_sched_barrier(0)
#endif

namespace {
using namespace psnd_stft;

constexpr int kR1 = 32, kL = 16, kC = 512, kNFFT = 1024, kK = 513;
constexpr int kWaves = 16;                         // waves per workgroup: 4 per SIMD at <= 128 VGPRs, one workgroup per CU
constexpr int kRow = 2 * kR1 + 4;                  // pitch of the window / twiddle tables (plan layout, psnd_stft_plan.h)
constexpr int kVkp = (2 * (kC / 2 + 1) + 3) & ~3;
constexpr int kTab = 2 * kL * kRow + kVkp;         // floats of tables in LDS: wt[16][68] | tw[16][68] | vk[257](re, im)
constexpr int kRP = 2 * kL + 4;                    // exchange row pitch (floats): 16 complex + 4; row * 9 mod 16 distinct -> b128 reads conflict-free
constexpr int kXF = 16 * kRP;                      // one frame's half exchange (16 rows); 576 = 0 (mod 64): the 4 frames of a lane group differ in rows only
constexpr int kXW = 4 * kXF;                       // a wave's buffer: 2304 floats = 9216 B (also its span of samples)
constexpr int kLdsFloats = kTab + kWaves * kXW;
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");
constexpr int kFPitch = 2112;                      // bytes between two frames' staged magnitudes (2052 used): 528 words = 16 (mod 64) banks - the four 16-lane
                                                   // groups of a staging write hit disjoint banks (a pitch of 513 words put them one bank apart: 4-way conflicts) -
                                                   // and a multiple of 16 bytes: the read-out stays on aligned ds_read_b128
static_assert(4 * kFPitch <= kXW * 4, "padded staging fits the wave's buffer");
constexpr int kStoresPerQuad = 9;                  // store instructions a wave issues per quad (behind the next span's transfer): 8 x 16 bytes per lane + the frames' last words
constexpr int kSPVMax = 9;                         // 16-byte span pieces per lane: 64 x 9 x 4 = 2304 samples (hop 256: 7)

struct QParams {
    const float *wav;
    const float *plan;
    float *mag;
    long long T, F;
    int hop, pad, nq, total_groups;                // nq = quads per clip; a group = 16 consecutive quads (one per wave)
    long long total_quads;
    float mag_eps;
    int ablate;                                    // debug (PSND_ABLATE): 2 = no global stores
};

template <int OFF>
__device__ __forceinline__ void lds_rd64(v2f &dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_rd128(f32x4 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// wait until at most N (<= 15: the counter's width) LDS operations of this wave are outstanding; `v` is what the wait makes available
// (its uses are ordered behind the wait).  The "memory" clobber keeps hipcc's own LDS accesses on their side of it, so that the counts
// written at the call sites - which include the stores hipcc emits - hold.
template <int N, class V>
__device__ __forceinline__ void lds_wait(V &v) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}
template <class V>
__device__ __forceinline__ void lds_touch(V &v) {              // orders the uses of v behind the asm statements in front of it
    asm volatile("" : "+v"(v));
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return static_cast<unsigned>(reinterpret_cast<uintptr_t>(p)); }

// the split twiddles of a lane in registers (post_emit_pk_vk with hooks): those of evaluation pairs 0 .. 3 and of the middle bin are
// requested before the second pass's transforms (first()), those of pairs 4 .. 7 when the split begins - behind them come the eight
// staging stores of pairs 0 .. 3 (at least four instructions: hipcc may fuse two into a ds_write2_b32), so one wait in front of pair 4
// that leaves four operations outstanding has them all
struct VkRegs {
    static constexpr bool kHooks = true;
    v2f a_[kL / 2], b_[kL / 2], mid_;
    unsigned va, vb;
    __device__ __forceinline__ void first(unsigned vk_lds) {
        static_for<0, kL / 4>([&](auto pc) __attribute__((always_inline)) {
            constexpr int pp = decltype(pc)::value;
            lds_rd64<8 * kR1 * pp>(a_[pp], va);
            lds_rd64<8 * kR1 * pp>(b_[pp], vb);
        });
        lds_rd64<8 * kR1 * (kL / 2)>(mid_, vk_lds);
    }
    __device__ __forceinline__ void first_arrived() {           // (call behind a wait that covers first())
        static_for<0, kL / 4>([&](auto pc) __attribute__((always_inline)) { lds_touch(a_[decltype(pc)::value]), lds_touch(b_[decltype(pc)::value]); });
        lds_touch(mid_);
    }
    __device__ __forceinline__ void begin() {
        static_for<kL / 4, kL / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int pp = decltype(pc)::value;
            lds_rd64<8 * kR1 * pp>(a_[pp], va);
            lds_rd64<8 * kR1 * pp>(b_[pp], vb);
        });
    }
    template <class PC>
    __device__ __forceinline__ void before(PC) {
        if constexpr (PC::value == kL / 4) {
            lds_wait<4>(a_[kL / 4]);
            static_for<kL / 4, kL / 2>([&](auto pc) __attribute__((always_inline)) { lds_touch(a_[decltype(pc)::value]), lds_touch(b_[decltype(pc)::value]); });
        }
    }
    template <class PC>
    __device__ __forceinline__ v2f a(PC) const { return a_[PC::value]; }
    template <class PC>
    __device__ __forceinline__ v2f b(PC) const { return b_[PC::value]; }
    __device__ __forceinline__ v2f mid() const { return mid_; }
};

// magnitude writer into the wave's LDS buffer: the quad's four spectra as they will lie in memory (frame * K + bin), post_emit_pk's
// byte offsets are used as they are
struct EmitStage {
    static constexpr bool kPairMag = true;
    float *buf;
    float eps;
    __device__ __forceinline__ v2f pair_mag(v2f za, v2f zb, v2f v) const {
        const v2f sq = rfft_pair_sq(za, zb, v, v2f{eps, eps});
        return v2f{__builtin_amdgcn_sqrtf(sq.x), __builtin_amdgcn_sqrtf(sq.y)};
    }
    template <bool CONJ>
    __device__ __forceinline__ OutVal make(v2f x) const {
        OutVal o;
        const v2f sq = pk::fma(x, x, v2f{eps, 0.f});
        o.m = __builtin_amdgcn_sqrtf(sq.x + sq.y);
        return o;
    }
    __device__ __forceinline__ void store(int voff, int soff, const OutVal &o) const {
        *reinterpret_cast<float *>(reinterpret_cast<char *>(buf) + voff + soff) = o.m;
    }
};

// HOP256: hop == 256 (settings.py:13, every BASELINE config at this size) - the span is stored with a skew of 32 floats per 256
// samples, so that the two frames of a 32-lane group (256 samples apart = the same 32 of 64 banks) read disjoint banks; the tap
// offsets stay compile-time immediates.  Other hops: plain span (partial two-way conflicts on the tap reads).
//
// LDS schedule of a quad (round 6).  A wave's time is a chain of LDS round trips - taps, window, inter-pass twiddles, rows, split
// twiddles, staged magnitudes - with four waves per SIMD to cover them.  hipcc issues a table read next to its use and answers it with
// lgkmcnt(0), which also waits for the LDS stores in front of it: ~40 exposed round trips per quad.  Here every table read is an explicit
// instruction issued one phase AHEAD of its use, with counted waits (LDS operations of a wave complete in order):
//   taps + window pieces of the first two butterfly groups (one wait) | window groups 2, 3 behind groups 0, 1 | twiddles of the first
//   exchange half before the last three radix-32 stages | twiddles of the second half in the middle of the first half's stores | rows A |
//   rows B and all 17 split twiddles before the rows' transforms | staged magnitudes out in one sweep.
// 119-122 -> 112 us for 1024 clips x 2 s in interleaved runs on one box, bit-identical output.
template <bool HOP256>
__global__ __launch_bounds__(1024, 1) void stft_fwd_n1024q_kernel(QParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_wt = smem, *s_tw = smem + kL * kRow, *s_vk = smem + 2 * kL * kRow;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    float *xw = smem + kTab + w * kXW;             // this wave's buffer: span of samples, then the exchange halves, then the magnitudes
    // lane = (frame of the quad fi = lane >> 4, l = lane & 15): pass 1 lane l of the frame, pass 2 row pair qq = l.
    // Per-lane addresses are NOT kept across the loop: every phase derives its own from a lane id laundered through an empty asm (hipcc
    // would hoist ~15 address registers out of the loop, past the 128 there are - and a scratch reload issued behind a global store
    // waits for that store's acknowledgement: one in-order vmcnt, psnd_stft_w.hip).
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(ln));
        return ln;
    };
    (void)lane;
    const int hop = p.hop;
    const TileWalk tw = tile_walk(p.total_groups);

    // ---- the wave's quad of a group: clip, first frame, frames that exist -------------------------------------------------------
    auto quad_of = [&](int group, int &clip, int &f0, int &nval) __attribute__((always_inline)) {
        const unsigned q = (unsigned)group * kWaves + w;            // (< 2^31: checked at launch)
        if (q >= (unsigned)p.total_quads) {
            nval = 0, clip = 0, f0 = 0;
            return;
        }
        clip = (int)(q / (unsigned)p.nq);
        f0 = (int)(q - (unsigned)clip * (unsigned)p.nq) * 4;
        const long long left = p.F - f0;
        nval = left < 4 ? (int)left : 4;
    };
    constexpr int kSPV = HOP256 ? 7 : kSPVMax;
    constexpr int kPiece = 256 + (HOP256 ? 32 : 0);              // LDS floats between two 256-sample pieces of the span
    typedef __attribute__((address_space(3))) char *lds_ptr;
    const unsigned xw_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lds_ptr)xw));        // LDS byte address of the buffer (SGPR)
    // The quad's span of samples goes STRAIGHT into the wave's LDS buffer (buffer_load_dwordx4 ... lds: 1 KiB = 256 samples per
    // instruction, lane-linear - exactly the span's layout, the skew of HOP256 falls between two instructions): no staging registers,
    // no commit pass.  Range-checked by the descriptor (lanes past the span deliver zeros).  The instructions are written out: as a
    // builtin hipcc answers the next LDS read with vmcnt(0), i.e. it waits for the STORES issued behind the transfer as well (one
    // in-order counter for loads and stores); here the wait at the top of the next quad is counted - vmcnt(kStoresPerQuad).
    // Clip edges (reflect indexing; the first and the last two quads of a clip): one 4-byte load per sample, all in flight at once.
    auto request_span = [&](int clip, int f0, int nval) __attribute__((always_inline)) {
        const float *x_ = p.wav + (size_t)clip * p.T;
        const int span_len = (nval - 1) * hop + kNFFT;
        const long long g0 = (long long)f0 * hop - p.pad;
#ifdef PSND_LAB
        if (p.ablate & 4) return;                                // A/B: no sample loads
#endif
        if (g0 >= 0 && g0 + span_len <= p.T) {                   // wave-uniform
            const unsigned long long a = reinterpret_cast<unsigned long long>(x_ + g0);
            u32x4 rs;
            rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
            rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
            rs.z = __builtin_amdgcn_readfirstlane((unsigned)(span_len * 4));
            rs.w = 0x00020000u;
            const int voff = fresh_lane() * 16;
#pragma unroll
            for (int j = 0; j < kSPV; ++j) {
                const unsigned dst = xw_lds + 4u * kPiece * j;
                const int soff = 1024 * j;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" PSND_Q_LOAD_MOD " lds"
                             :: "s"(dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
            }
        } else {
            // A span that crosses a clip edge (the first and the last two quads of a clip), piece by piece (256 samples = one transfer instruction
            // of the main path; every test below is wave-uniform): a piece inside the clip goes the way of the main path, a piece the reflection
            // reaches is fetched as four 4-byte loads per lane with reflected indices (all in flight before the first lands) and written into its
            // place, a piece past the span (quads of fewer than four frames) is left alone - its frames are never stored.  Round 6, first form:
            // all 28 loads per lane reflected whatever the piece - ~600 VALU instructions of index arithmetic per edge quad, next to ~700 for the
            // quad's transforms, in 3 of a clip's 44 quads.
            const int Ti = (int)p.T, gb = (int)g0;
            const int ln = fresh_lane();
            static_for<0, kSPV>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const int a0 = gb + 256 * j;                     // first sample of the piece in the clip (wave-uniform)
                if (256 * j < span_len) {
                    if (a0 >= 0 && a0 + 256 <= Ti) {
                        const unsigned long long a = reinterpret_cast<unsigned long long>(x_ + a0);
                        u32x4 rs;
                        rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
                        rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
                        rs.z = 1024u;
                        rs.w = 0x00020000u;
                        const unsigned dst = xw_lds + 4u * kPiece * j;
                        const int voff = ln * 16;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen" PSND_Q_LOAD_MOD " lds"
                                     :: "s"(dst), "v"(voff), "s"(rs) : "memory");
                    } else {
                        float ev[4];
                        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
                            constexpr int i = decltype(ic)::value;
                            const int sj = 256 * j + ln + 64 * i;
                            ev[i] = x_[reflect_idx32(gb + (sj < span_len ? sj : span_len - 1), Ti)];
                        });
                        // (written unconditionally: a slot past the span holds a copy of its last sample)
                        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
                            constexpr int i = decltype(ic)::value;
                            xw[kPiece * j + ln + 64 * i] = ev[i];
                        });
                    }
                }
            });
            static_assert(kPiece * (kSPV - 1) + 255 < kXW, "edge fill stays inside the wave's buffer");
        }
    };

    int clip = 0, f0 = 0, nval = 0;
    if (tw.first < tw.end) {
        quad_of(tw.first, clip, f0, nval);
        if (nval > 0) request_span(clip, f0, nval);             // HBM first, then the tables (L2)
    }
    for (int i = t; i < kTab / 4; i += 1024) reinterpret_cast<f32x4 *>(smem)[i] = reinterpret_cast<const f32x4 *>(p.plan)[i];
    __syncthreads();                                            // the only workgroup barrier: tables visible

    for (int group = tw.first; group < tw.end; group += tw.step) {
        int nclip = 0, nf0 = 0, nnval = 0;
        if (group + tw.step < tw.end) quad_of(group + tw.step, nclip, nf0, nnval);
        if (nval > 0) {                                         // wave-uniform
            // the span transfer was issued in front of the previous quad's stores: wait for IT, not for them
#ifdef PSND_LAB
            if (group != tw.first && !(p.ablate & 2)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kStoresPerQuad) : "memory");
#else
            if (group != tw.first) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kStoresPerQuad) : "memory");
#endif
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- pass 1: taps, window, radix-32, inter-pass twiddle ---------------------------------------------------------------
            v2f z[kR1];
            f32x4 wq[3][4];                                     // window pieces of three butterfly groups (a group = J = 4 g .. 4 g + 3)
            const int ln = fresh_lane(), fi = ln >> 4, l = ln & 15;
            const unsigned wa_lds = lds_addr(s_wt + l * kRow);
            // piece (g, 0 | 1): w[J], w[J + 16] for J = 4 g, 4 g + 1; (g, 2 | 3): for J = 4 g + 2, 4 g + 3
            auto window_group = [&](auto gc, f32x4 (&dst)[4]) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                lds_rd128<32 * g>(dst[0], wa_lds), lds_rd128<32 * g + 4 * kR1>(dst[1], wa_lds);
                lds_rd128<32 * g + 16>(dst[2], wa_lds), lds_rd128<32 * g + 16 + 4 * kR1>(dst[3], wa_lds);
            };
            auto butterflies = [&](auto gc, f32x4 (&wv)[4]) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                // the window rides in the first butterfly stage: z[a] w[a] +- z[a + 16] w[a + 16] = one multiply + two fused multiply-adds
                pk::bfly_windowed<kR1, 4 * g>(z, pk::lo(wv[0]), pk::lo(wv[1]));
                pk::bfly_windowed<kR1, 4 * g + 1>(z, pk::hi(wv[0]), pk::hi(wv[1]));
                pk::bfly_windowed<kR1, 4 * g + 2>(z, pk::lo(wv[2]), pk::lo(wv[3]));
                pk::bfly_windowed<kR1, 4 * g + 3>(z, pk::hi(wv[2]), pk::hi(wv[3]));
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            {
                const unsigned ta = lds_addr(xw + fi * hop + 2 * l + (HOP256 ? 32 * fi : 0));
                // explicit ds_read_b64: hipcc pairs neighbouring taps into ds_read2_b64, which the LDS serves at half the rate of two
                // ds_read_b64 (MI355X_MICROARCH.md, LDS table; round 5: psnd_stft_r.hip)
                static_for<0, kR1>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = decltype(ac)::value;
                    // sample 2 (l + 16 a) of the frame; HOP256: 32 a + 2 l < 256 (a % 8 + 1), so the block of the skew is fi + a / 8
                    lds_rd64<4 * (32 * a + (HOP256 ? 32 * (a / 8) : 0))>(z[a], ta);
                });
                window_group(I0{}, wq[0]);
                window_group(I1{}, wq[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]),
                             "+v"(z[8]), "+v"(z[9]), "+v"(z[10]), "+v"(z[11]), "+v"(z[12]), "+v"(z[13]), "+v"(z[14]), "+v"(z[15]) :: "memory");
                asm volatile("" : "+v"(z[16]), "+v"(z[17]), "+v"(z[18]), "+v"(z[19]), "+v"(z[20]), "+v"(z[21]), "+v"(z[22]), "+v"(z[23]),
                             "+v"(z[24]), "+v"(z[25]), "+v"(z[26]), "+v"(z[27]), "+v"(z[28]), "+v"(z[29]), "+v"(z[30]), "+v"(z[31]));
                asm volatile("" : "+v"(wq[0][0]), "+v"(wq[0][1]), "+v"(wq[0][2]), "+v"(wq[0][3]), "+v"(wq[1][0]), "+v"(wq[1][1]), "+v"(wq[1][2]), "+v"(wq[1][3]));
            }
            window_group(I2{}, wq[2]);
            butterflies(I0{}, wq[0]);
            Q_SB();
            window_group(I3{}, wq[0]);
            butterflies(I1{}, wq[1]);
            Q_SB();
            lds_wait<4>(wq[2][0]);                              // (the four pieces of group 3 are behind it)
            lds_touch(wq[2][1]), lds_touch(wq[2][2]), lds_touch(wq[2][3]);
            butterflies(I2{}, wq[2]);
            Q_SB();
            // ---- exchange inside the wave, two half rounds (rows 0..15, then 16..31) ------------------------------------------------
            Q_PRIO(PSND_Q_PRIO_X);
            const unsigned tw_lds = lds_addr(s_tw + l * kRow);
            float *oz = xw + fi * kXF + 2 * l;
            const int rowB = l == 0 ? 0 : 16 - l;               // row inside the second half (rows 16 .. 31)
            f32x4 twa[8], twb[8];                               // twiddle pieces (q0, q0 + 1) of the two halves
            auto twiddles = [&](auto hc, f32x4 (&dst)[8]) __attribute__((always_inline)) {
                constexpr int Q0 = decltype(hc)::value;
                static_for<0, 8>([&](auto ic) __attribute__((always_inline)) { lds_rd128<8 * (Q0 + 2 * decltype(ic)::value)>(dst[decltype(ic)::value], tw_lds); });
            };
            auto write_pair = [&](auto hc, auto ic, const f32x4 &wv) __attribute__((always_inline)) {
                constexpr int Q0 = decltype(hc)::value, q0 = Q0 + 2 * decltype(ic)::value, q1 = q0 + 1;
                constexpr int s0_ = ct::bitrev(q0, 5), s1_ = ct::bitrev(q1, 5);
                if constexpr (q0 == 0) *reinterpret_cast<v2f *>(oz) = z[s0_];
                else *reinterpret_cast<v2f *>(oz + (q0 - Q0) * kRP) = pk::cmul(z[s0_], pk::lo(wv));
                *reinterpret_cast<v2f *>(oz + (q1 - Q0) * kRP) = pk::cmul(z[s1_], pk::hi(wv));
            };
            auto read_row = [&](int row, v2f (&r)[kL]) __attribute__((always_inline)) {
                const unsigned ra = lds_addr(xw + fi * kXF + row * kRP);
                static_for<0, kL / 2>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    lds_rd128<16 * i>(*reinterpret_cast<f32x4 *>(&r[2 * i]), ra);
                });
            };
            using H0 = std::integral_constant<int, 0>;
            using H1 = std::integral_constant<int, 16>;
            lds_wait<0>(wq[0][0]);                              // window group 3
            lds_touch(wq[0][1]), lds_touch(wq[0][2]), lds_touch(wq[0][3]);
            butterflies(I3{}, wq[0]);
            Q_SB();
            twiddles(H0{}, twa);                                // in flight under the three radix-32 stages that follow
            pk::stage<kR1, kR1 / 4, -1>(z);
            Q_SB();
            // (the taps above were read out of the same buffer: one wave's LDS operations execute in order)
            // first half: behind piece i stand pieces i + 1 .. 7 and the stores of the i pairs in front of it - AT LEAST i instructions
            // (hipcc fuses the two stores of a pair into one ds_write2_b64 when it can): every count below is such a lower bound
            static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                lds_wait<7>(twa[i]);
                write_pair(H0{}, ic, twa[i]);
            });
            twiddles(H1{}, twb);                                // the registers of the four pairs written hold the second half's pieces
            static_for<4, 8>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                lds_wait<15>(twa[i]);                           // (behind it: 7 - i pieces, >= i stores, 8 pieces of the second half)
                write_pair(H0{}, ic, twa[i]);
            });
            v2f za[kL], zb[kL];
            read_row(l, za);
            // second half: behind piece i stand >= 4 stores of the first half, the 8 reads of rows A and 7 more: the widest wait serves
            static_for<0, 8>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                lds_wait<15>(twb[i]);
                write_pair(H1{}, ic, twb[i]);
            });
            Q_SB();
            Q_PRIO(0);
            // rows B and the split twiddles v[qA + 32 pp], v[qB + 32 pp], v[256]: requested now, used behind the two radix-16 transforms
            read_row(rowB, zb);
            VkRegs vk;
            vk.va = lds_addr(s_vk + 2 * l), vk.vb = lds_addr(s_vk + 2 * (l == 0 ? kR1 / 2 : kR1 - l));
            vk.first(lds_addr(s_vk));
            lds_wait<15>(za[0]);                                // rows A: >= 8 stores + 8 + 9 reads behind them
            static_for<1, kL>([&](auto ic) __attribute__((always_inline)) { lds_touch(za[decltype(ic)::value]); });
            pk::fft<kL>(za);
            Q_SB();
            lds_wait<9>(zb[0]);                                 // rows B: the 9 split twiddles behind them
            static_for<1, kL>([&](auto ic) __attribute__((always_inline)) { lds_touch(zb[decltype(ic)::value]); });
            pk::fft<kL>(zb);
            Q_SB();
            lds_wait<0>(vk.mid_);                               // the buffer has been read out, the first split twiddles are here
            vk.first_arrived();
            // ---- real-FFT split + magnitude into the wave's buffer: the quad's region of the output, byte for byte (4 x 2052 B) -----------
            {
                const int ln3 = fresh_lane(), qq = ln3 & 15;
                const bool special = qq == 0;
                EmitStage emit{xw, p.mag_eps};
                post_emit_pk_vk<kR1, kL>(za, zb, special, qq, special ? kR1 / 2 : kR1 - qq, vk, emit, 1, (ln3 >> 4) * kFPitch);
            }
            Q_SB();
            // ---- out again as 16 bytes per lane: a store instruction writes 1 KiB of contiguous memory ---------------------------------
            Q_PRIO(PSND_Q_PRIO_MEM);
            // Frame f's 2052 bytes lie at f * kFPitch in the buffer: 128 whole 16-byte pieces + one word.  Store instruction j takes the 64 pieces
            // [64 (j % 2), 64 (j % 2) + 64) of frame j / 2 - 1 KiB of contiguous memory, LDS reads 16-byte aligned - and a ninth one the four last words.
            constexpr int kPieces = 8;
            f32x4 o[kPieces];
            const int lnf = fresh_lane();
            const int bytes = nval * (kK * 4);
            {
                const char *src = reinterpret_cast<const char *>(xw) + lnf * 16;
                static_for<0, kPieces>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    o[j] = *reinterpret_cast<const f32x4 *>(src + (j / 2) * kFPitch + (j % 2) * 1024);
                });
            }
            const float tv = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xw) + (lnf & 3) * kFPitch + 2048);   // lane f < 4: bin 512 of frame f
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the buffer has been read out
            Q_SB();
            // the next quad's span: into the (free) buffer, AHEAD of this quad's stores in the in-order vector-memory queue
            if (nnval > 0) request_span(nclip, nf0, nnval);
            Q_SB();
#ifdef PSND_LAB
            if (!(p.ablate & 2))
#endif
            {
                // every store is ISSUED whatever the quad (the wait above counts them): what lies past the quad's bytes is dropped by
                // the descriptor's range check (frames past F of a clip's last quad)
                const __amdgpu_buffer_rsrc_t ro = make_uniform_rsrc(p.mag + ((size_t)clip * (size_t)p.F + (size_t)f0) * kK, bytes);
                static_for<0, kPieces>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    // (frames past nval: offsets at or beyond `bytes`, dropped by the range check)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[j]), ro, (j / 2) * (kK * 4) + (j % 2) * 1024 + lnf * 16, 0, PSND_Q_STORE_AUX);
                });
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tv), ro, lnf < 4 ? lnf * (kK * 4) + 2048 : (1 << 30), 0, 0);
            }
        }
        Q_PRIO(0);
        clip = nclip, f0 = nf0, nval = nnval;
    }
}

}  // namespace

bool psnd_stft1024q_ok(long long T, long long F, int hop, int pad) {
    (void)T;
    (void)pad;
    // even hop (8-byte tap reads), the four-frame span inside the wave's buffer, 32-bit offsets
    return hop >= 2 && hop % 2 == 0 && 3 * hop + kNFFT + (hop == 256 ? 32 * 7 : 0) <= kXW && 3 * hop + kNFFT <= 64 * 4 * kSPVMax && F > 0 &&
           (long long)kK * F < (1ll << 31);
}

int psnd_stft1024q_launch(const float *wav, const float *plan, float *mag_nfk, long long N, long long T, long long F, int hop, int pad,
                          float mag_eps, int ablate, hipStream_t stream) {
    QParams p;
    p.wav = wav, p.plan = plan, p.mag = mag_nfk, p.T = T, p.F = F, p.hop = hop, p.pad = pad, p.mag_eps = mag_eps, p.ablate = ablate;
    const long long nq = (F + 3) / 4;
    p.nq = (int)nq;
    p.total_quads = nq * N;
    const long long groups = (p.total_quads + kWaves - 1) / kWaves;
    if (p.total_quads + kWaves >= (1ll << 31)) PSND_FAIL(PSND_E_SHAPE, "stft_mag_nfk(n1024q): too many quads");
    if (groups >= (1ll << 31)) PSND_FAIL(PSND_E_SHAPE, "stft_mag_nfk(n1024q): too many quads");
    p.total_groups = (int)groups;
    int grid = p.total_groups < 256 ? p.total_groups : 256;            // one persistent workgroup per CU
    grid = (grid + 7) & ~7;
    constexpr size_t lds = sizeof(float) * kLdsFloats;
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_mag_nfk(n1024q): set LDS size: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, stream, p);
        return PSND_OK;
    };
    const int rc = hop == 256 ? launch(stft_fwd_n1024q_kernel<true>) : launch(stft_fwd_n1024q_kernel<false>);
    if (rc != PSND_OK) return rc;
    PSND_CHECK_LAUNCH("stft_mag_nfk(n1024q)");
    return PSND_OK;
}
