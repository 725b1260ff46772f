// psnd_stft_r.hip - n_fft = 4096, hop = 1024 forward STFT, magnitude in the BIN-FASTEST layout (N, F, K) of psnd_stft_mag_nfk:
// the wave-per-frame transform of psnd_stft_w.hip fed from a SAMPLE RING in LDS that the workgroup fills once per sample.
//
// Replaces STFT.transform (pytorch_sound/models/transforms.py:53-69) for BASELINE config 5 (44.1 kHz, 4096 / 1024, 30 s clips) where the
// consumer takes (N, F, K).
//
// Why a fourth 4096 kernel (round 5).  stft_fwd_n4096w_kernel<NFK> lets every wave load its frame's 4096 samples itself, 8 bytes per
// lane: 32 load instructions per frame, each sample fetched by the four waves whose frames overlap it.  Counters
// (profiles/r04_stft4096w_nfk_pmc.txt): FETCH 263.6 MB for 169.3 MB of samples (1.56 x), 54 % of the wave cycles parked.  The vector-
// memory front end of a CU takes a wave instruction every ~16 cycles whatever its width, so 16 waves x 32 loads are 8 k of a tile's
// ~38 k cycles spent ISSUING loads, and a wave's load latency is covered only by the three other waves of its SIMD.  Here:
//
//   * a workgroup owns a contiguous run of frames of ONE clip (a "segment"; global frame range split evenly over the grid, cut at clip
//     boundaries): consecutive frames share 3 of their 4 hops, so the run's samples are one contiguous stream;
//   * the stream is cut in chunks of one hop (1024 samples = 4 KB).  A chunk enters the LDS ring ONCE, by four
//     `buffer_load_dwordx4 ... lds` (1 KB per instruction, no registers, no commit pass): 4 vector-memory instructions per frame
//     instead of 32, every sample fetched once per segment ((nfr + 3) / nfr of the algorithmic reads);
//   * ring of 16 slots (64 KB).  The wave that STARTS frame u issues the transfer of chunk u + kLead - needed first by frame
//     u + kLead - 3, i.e. nine frame starts later - and publishes it (ready[slot] = chunk + 1, left[slot] = its number of readers) in the
//     middle of its own transform, behind one `s_waitcnt vmcnt(0)` that by then costs nothing.  A frame waits for its four chunks'
//     tags, copies its samples to registers (32 ds_read_b64) and takes itself off the chunks' reader counts; a slot is refilled when
//     its count is back to zero.  Every dependency points to an EARLIER frame of the segment: no cycle, no workgroup barrier in the
//     frame loop, the 16 waves drift as they like;
//   * clip edges (reflect padding, transforms.py:55-60; the first two and the last three chunks of a clip) are gathered element by
//     element by the issuing wave - the frames themselves never see an edge;
//   * to make room for the ring the per-wave transpose buffer holds one COMPONENT of one half-wave at a time (4.1 KB instead of
//     8.3 KB: re then im, ds_*_b32) - 16 x 4.1 + 64 + 25 KB of tables = 155 KB of LDS, one 1024-thread workgroup per CU.
//
// The transform itself (radix-2 in lane, v_permlane32_swap, radix-32, 32 x 32 transpose per half-wave, radix-32, real-FFT split through
// ds_bpermute, magnitudes stored straight from registers) is the one of psnd_stft_w.hip; see there for the index algebra.
// Bound: HBM (4 hop + 4 K = 12 292 B per frame); DESIGN.md 4.1e for the measured fraction.
#include "psnd_pk.h"
#include "psnd_stft_pass.h"
#include "psnd_stft_w.h"
#include <stdlib.h>

#ifndef PSND_R_STORE_AUX
#define PSND_R_STORE_AUX 2     // cache-policy bits of the output stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef PSND_R_LEAD
#define PSND_R_LEAD 12         // chunks between the frame a wave starts and the chunk it requests (<= kSlots - 1)
#endif

#ifndef PSND_R_ABL
#define PSND_R_ABL 0           // timing ablations (tools/r05): 1 no transpose, 2 no sample reads, 4 no window, 8 no radix-32s, 16 no split, 32 no polls, 64 no transfers
#endif

namespace {
using namespace psnd_stft;

constexpr int kC = 2048, kNFFT = 4096, kK = 2049, kHop = 1024;
constexpr int kWaves = 16;                        // waves per workgroup = frames in flight
constexpr int kSlots = 16;                        // ring slots of one hop each
constexpr int kLead = PSND_R_LEAD;
constexpr int kXP = 33;                           // transpose row pitch (complex values)
constexpr int kXbFloats = 2 * 32 * kXP * 2;       // one transpose buffer: [half-wave][q1][lam] (re, im), 16.9 KB
constexpr int kXbN = 4;                           // four of them: one per SIMD, shared by the four waves of that SIMD under a lock
constexpr int kOffWin = 0;                        // [32 loads][64 lanes] of (0.5 w[2n], 0.5 w[2n+1]) in the order a lane takes its samples
constexpr int kOffTw = 4096;                      // W_1024^(lam q1) as [q1][lam] (re, im)
constexpr int kOffCl = kOffTw + 2048;             // per lane: cL = W_2048^lam (-i)^g, then v_c = -i W_4096^c (c = 2 lam + g), v_(C/2)
constexpr int kOffRing = kOffCl + 260;
constexpr int kOffXa = kOffRing + kSlots * kHop;
constexpr int kOffFlags = kOffXa + kXbN * kXbFloats;        // int ready[16] | int left[16] | int lock[4]
constexpr int kLdsFloats = kOffFlags + 2 * kSlots + kXbN;
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");
static_assert(kLead >= 4 && kLead < kSlots, "a frame's own four chunks come from earlier requests; the ring holds kSlots chunks");
static_assert(kOffRing % 4 == 0, "16-byte aligned ring slots (LDS-DMA writes 16 bytes per lane)");
constexpr int kStoresPerFrame = 17;

struct RParams {
    const float *wav;
    const float *plan;
    float *mag;
    int T, F, total_frames;                       // (< 2^31: checked at launch)
    int pad;
    float mag_eps;
    int ablate;                                   // debug (PSND_ABLATE): 2 = no global stores
};

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

__device__ __forceinline__ float bperm(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// explicit LDS accesses (a function, not a macro around the asm: a variable named only inside an asm is not captured by a lambda)
template <int OFF>
__device__ __forceinline__ void lds_rd64(v2f &dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_wr64(unsigned addr, v2f v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ int lds_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__global__ __launch_bounds__(1024, 1) void stft_fwd_n4096r_kernel(RParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_win = smem + kOffWin, *s_tw = smem + kOffTw, *s_cl = smem + kOffCl, *s_ring = smem + kOffRing;
    int *s_ready = reinterpret_cast<int *>(smem + kOffFlags), *s_left = s_ready + kSlots;
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *plan = p.plan;

    // ---- tables (as psnd_stft_w.hip) -----------------------------------------------------------------------------------------
    for (int e = t; e < 2048; e += 1024) {
        const int i = e >> 6, ln = e & 63;
        const int n = (ln & 31) + 32 * ((i & 15) + 16 * (ln >> 5)) + 1024 * (i >> 4);
        const f32x2 wv = *reinterpret_cast<const f32x2 *>(plan + 2 * n);
        *reinterpret_cast<f32x2 *>(s_win + 2 * e) = f32x2{0.5f * wv.x, 0.5f * wv.y};
    }
    if (t < 512) reinterpret_cast<f32x4 *>(s_tw)[t] = reinterpret_cast<const f32x4 *>(plan + kW4096TwOff)[t];
    if (t < 64) {
        *reinterpret_cast<v2f *>(s_cl + 2 * t) = *reinterpret_cast<const v2f *>(plan + kW4096ClOff + 2 * t);
        const int c = 2 * (t & 31) + (t >> 5);
        *reinterpret_cast<v2f *>(s_cl + 128 + 2 * t) = *reinterpret_cast<const v2f *>(plan + kW4096VkOff + 2 * c);
        if (t == 0) *reinterpret_cast<v2f *>(s_cl + 256) = *reinterpret_cast<const v2f *>(plan + kW4096VkOff + 2 * 1024);
    }
    int *s_lock = s_left + kSlots;
    if (t < kXbN) s_lock[t] = 0;
    typedef __attribute__((address_space(3))) char *lds_ptr;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lds_ptr)s_ring));     // LDS byte address of the ring (SGPR)

    // per-lane values are re-derived in every phase from a laundered lane id (psnd_stft_w.hip: hoisted addresses cost ~25 VGPRs)
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(ln));
        return ln;
    };

    // this workgroup's run of global frames (clip-major), cut into segments at clip boundaries
    const int ga = (int)((long long)blockIdx.x * p.total_frames / gridDim.x);
    const int gb = (int)((long long)(blockIdx.x + 1) * p.total_frames / gridDim.x);
    const int F = p.F;
    v2f z[32];
    int clip = ga / F, fa = ga - clip * F;                       // (the only division: later segments start at frame 0 of the next clip)

    for (int g = ga; g < gb; ++clip, fa = 0) {
        const int rest = gb - g, room = F - fa;
        const int nfr = rest < room ? rest : room;               // frames of this segment: fa .. fa + nfr - 1 of `clip`
        const int nch = nfr + 3;                                 // chunks fa .. fa + nfr + 2 (chunk c = padded samples [c hop, (c + 1) hop))
        g += nfr;
        const float *xclip = p.wav + (size_t)clip * (size_t)p.T;

        // chunk r of the segment -> ring slot r % kSlots.  Interior: four 1-KB LDS-DMA instructions; clip edge: reflect gather.
        auto request_chunk = [&](int r) __attribute__((always_inline)) {
            const int g0 = (fa + r) * kHop - p.pad;                             // first sample of the chunk (may be < 0 or reach past T)
            const int slot = r & (kSlots - 1);
            if (g0 >= 0 && g0 + kHop <= p.T) {                                   // wave-uniform
                const unsigned long long a = reinterpret_cast<unsigned long long>(xclip + g0);
                u32x4 rs;
                rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
                rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
                rs.z = 4 * kHop;
                rs.w = 0x00020000u;
                const int voff = fresh_lane() * 16;
                const unsigned dst0 = ring_lds + (unsigned)slot * (4u * kHop);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned dst = dst0 + 1024u * j;
                    const int soff = 1024 * j;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                 :: "s"(dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
                }
            } else {
                float *dst = s_ring + slot * kHop;
#pragma unroll 1
                for (int e = fresh_lane(); e < kHop; e += 64) dst[e] = xclip[reflect_idx32(g0 + e, p.T)];
            }
        };
        // the transfer has landed (this wave's vmcnt / lgkmcnt), tell the readers: number of frames that take samples from chunk r
        // (`behind` = this wave issued a frame's stores AFTER the transfer: one in-order counter - wait for the transfer, not for them)
        auto publish_chunk = [&](int r, bool behind) __attribute__((always_inline)) {
            if (behind) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(kStoresPerFrame) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const int lo = r - 3 > 0 ? r - 3 : 0, hi = r < nfr - 1 ? r : nfr - 1;
            const int slot = r & (kSlots - 1);
            lds_store(s_left + slot, hi - lo + 1);
            asm volatile("" ::: "memory");
            lds_store(s_ready + slot, r + 1);
        };

        __syncthreads();                                        // the previous segment (or the tables) is done with: ring and flags are free
        if (t < 2 * kSlots) s_ready[t] = 0;
        __syncthreads();                                        // flags reset before anyone publishes

        const bool nostore = p.ablate & 2;
        // The chunk "of" frame u is chunk u + kLead (the "frames" u in [-kLead, 0) stand for the segment's first kLead chunks).  It is
        // requested at the END of the wave's previous frame, in FRONT of that frame's stores, and published in the middle of frame u behind
        // `s_waitcnt vmcnt(17)`: the counter runs in order, so the wait is for the transfer, not for the acknowledgement of 8 KB of stores
        // on a write path that is busy most of the time.  A slot is free once every reader of the chunk 16 back is through.
        auto wait_slot = [&](int r) __attribute__((always_inline)) {
            if (r >= kSlots && !(PSND_R_ABL & 32)) {
                const int slot = r & (kSlots - 1), tag = r - kSlots + 1;
                while (!(lds_load(s_ready + slot) == tag && lds_load(s_left + slot) == 0)) __builtin_amdgcn_s_sleep(2);
            }
        };
        int pend = -1;                                          // chunk requested by this wave and not yet published
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
            const int r = w - kWaves + kLead + kWaves * k;
            if (r >= 0 && r < nch && (k == 0 || w < nfr)) {
                wait_slot(r);
                if (!(PSND_R_ABL & 64)) request_chunk(r);
                if (k == 0) publish_chunk(r, false);
                else pend = r;
            }
        }
        for (int u = w; u < nfr; u += kWaves) {
            // ---- this frame's four chunks ----------------------------------------------------------------------------------------
            if (!(PSND_R_ABL & 32)) {
                const int ln = fresh_lane();
                const int idx = ln & 3;
                for (;;) {
                    const int v = lds_load(s_ready + ((u + idx) & (kSlots - 1)));
                    if (__builtin_amdgcn_ballot_w64(v == u + idx + 1) == ~0ull) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            asm volatile("" ::: "memory");
            // LDS reads of this phase are explicit ds_read_b64 (hipcc fuses neighbouring ones into ds_read2_b64, which the LDS serves at half
            // the rate: MI355X_MICROARCH.md, LDS table); the values become usable behind a wait that names them (PSND_R_WAIT*)
#define PSND_R_LD(dst_, addr_, off_) lds_rd64<(off_)>(dst_, addr_)
#define PSND_R_WAIT8(a_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(a_[4]), "+v"(a_[5]), "+v"(a_[6]), "+v"(a_[7]))
#define PSND_R_TIEZ(o_)                                                                                                                 \
    asm volatile("" : "+v"(z[o_]), "+v"(z[o_ + 1]), "+v"(z[o_ + 2]), "+v"(z[o_ + 3]), "+v"(z[o_ + 4]), "+v"(z[o_ + 5]), "+v"(z[o_ + 6]),   \
                 "+v"(z[o_ + 7]), "+v"(z[o_ + 8]), "+v"(z[o_ + 9]), "+v"(z[o_ + 10]), "+v"(z[o_ + 11]), "+v"(z[o_ + 12]), "+v"(z[o_ + 13]),  \
                 "+v"(z[o_ + 14]), "+v"(z[o_ + 15]))
            {
                // lane (lam, g): complex point n = lam + 32 (a + 16 g) = samples 2 n, 2 n + 1 -> chunk u + g, offset 2 lam + 64 a;
                // point n + 1024 -> chunk u + g + 2
                const int ln = fresh_lane(), lam_ = ln & 31, g_ = ln >> 5;
                const unsigned lo = static_cast<unsigned>(reinterpret_cast<uintptr_t>(s_ring + ((u + g_) & (kSlots - 1)) * kHop + 2 * lam_));
                const unsigned hi = static_cast<unsigned>(reinterpret_cast<uintptr_t>(s_ring + ((u + g_ + 2) & (kSlots - 1)) * kHop + 2 * lam_));
                static_for<0, 16>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = decltype(ac)::value;
                    if constexpr (PSND_R_ABL & 2) {
                        z[a] = v2f{(float)a, (float)ln}, z[16 + a] = v2f{(float)ln, (float)a};
                    } else {
                        const unsigned lo_ = lo, hi_ = hi;     // (named here: a variable used only inside an asm is not captured)
                        PSND_R_LD(z[a], lo_, 256 * a);
                        PSND_R_LD(z[16 + a], hi_, 256 * a);
                    }
                });
                // off the reader counts (LDS operations of one wave execute in order: behind the reads above)
                asm volatile("" ::: "memory");
                if (ln < 4) __hip_atomic_fetch_add(s_left + ((u + ln) & (kSlots - 1)), -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            float mlo[16], mhi[16], mext;
            // ---- window, radix-2 (decimation in frequency) in lane: the window values arrive one group of 4 + 4 ahead ---------------
            {
                const int ln1 = fresh_lane();
                const unsigned winl = static_cast<unsigned>(reinterpret_cast<uintptr_t>(s_win + 2 * ln1));
                const v2f cL = *reinterpret_cast<const v2f *>(s_cl + 2 * ln1);
                v2f wb[2][8];
                auto ld_win = [&](auto cc, v2f (&d)[8]) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    static_for<0, 4>([&](auto ac) __attribute__((always_inline)) {
                        constexpr int a = c * 4 + decltype(ac)::value;
                        const unsigned wl_ = winl;
                        PSND_R_LD(d[decltype(ac)::value], wl_, a * 512);
                        PSND_R_LD(d[4 + decltype(ac)::value], wl_, (16 + a) * 512);
                    });
                };
                ld_win(std::integral_constant<int, 0>{}, wb[0]);
                PSND_R_WAIT8(wb[0]);
                PSND_R_TIEZ(0);
                PSND_R_TIEZ(16);
                static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c < 3) ld_win(std::integral_constant<int, c + 1>{}, wb[(c + 1) & 1]);
                    static_for<0, 4>([&](auto ac) __attribute__((always_inline)) {
                        constexpr int i = decltype(ac)::value, a = c * 4 + i;
                        const v2f lo = (PSND_R_ABL & 4) ? z[a] : z[a] * wb[c & 1][i];
                        const v2f hi = (PSND_R_ABL & 4) ? z[16 + a] : z[16 + a] * wb[c & 1][4 + i];
                        z[a] = lo + hi;
                        z[16 + a] = pk::cmul(cmul_ct<a, 64>(lo - hi), cL);
                    });
                    if constexpr (c < 3) PSND_R_WAIT8(wb[(c + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // ---- lanes < 32 take every u, lanes >= 32 every v -----------------------------------------------------------------------
            static_for<0, 16>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                float ux = z[a].x, uy = z[a].y, vx = z[16 + a].x, vy = z[16 + a].y;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(ux), "+v"(vx), "+v"(uy), "+v"(vy));
                z[a] = v2f{ux, uy};
                z[16 + a] = v2f{vx, vy};
            });
            // ---- first radix-32 + inter-pass twiddle (table values one group of 8 ahead) -------------------------------------------------
            const int ln2 = fresh_lane();
            {
                const unsigned twl = static_cast<unsigned>(reinterpret_cast<uintptr_t>(s_tw + 2 * (ln2 & 31)));
                v2f tb[2][8];
                auto ld_tw = [&](auto cc, v2f (&d)[8]) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q1 = c * 8 + decltype(qc)::value;
                        const unsigned tl_ = twl;
                        PSND_R_LD(d[decltype(qc)::value], tl_, q1 * 256);
                    });
                };
                ld_tw(std::integral_constant<int, 0>{}, tb[0]);          // (travels under the butterflies)
                if constexpr (!(PSND_R_ABL & 8)) pk::fft<32>(z);
                PSND_R_WAIT8(tb[0]);
                static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c < 3) ld_tw(std::integral_constant<int, c + 1>{}, tb[(c + 1) & 1]);
                    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int i = decltype(qc)::value, q1 = c * 8 + i, sl = ct::bitrev(q1, 5);
                        if constexpr (q1 != 0) z[sl] = pk::cmul(z[sl], tb[c & 1][i]);
                    });
                    if constexpr (c < 3) PSND_R_WAIT8(tb[(c + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // ---- the chunk requested at the top has had a third of a transform to arrive: publish it ----------------------------------
            if (pend >= 0) publish_chunk(pend, u != w && !nostore);
            // ---- 32 x 32 transpose per half-wave, both at once (ds_write_b64 / ds_read_b64 over all 64 lanes: a third of the LDS cycles of
            //      half-masked or component-wise rounds), through one of four 16.9 KB buffers - the one of this wave's SIMD, taken under a lock:
            //      a wave holds it for one burst of 64 LDS instructions per frame
            if constexpr (!(PSND_R_ABL & 1)) {
                const int lam3 = ln2 & 31, g3 = ln2 >> 5;
                float *xb = smem + kOffXa + (w & (kXbN - 1)) * kXbFloats;
                int *lock = s_lock + (w & (kXbN - 1));
                for (;;) {
                    int seen = 1;
                    if (ln2 == 0) {
                        int expect = 0;
                        __hip_atomic_compare_exchange_strong(lock, &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        seen = expect;
                    }
                    if (__builtin_amdgcn_readfirstlane(seen) == 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                const unsigned wr_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(xb + 2 * (g3 * 32 * kXP + lam3)));
                const unsigned rd_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(xb + 2 * (g3 * 32 * kXP + lam3 * kXP)));
                static_for<0, 32>([&](auto sc) __attribute__((always_inline)) {
                    constexpr int sl = decltype(sc)::value, q1 = ct::bitrev(sl, 5);
                    lds_wr64<q1 * kXP * 8>(wr_addr, z[sl]);
                });
                static_for<0, 32>([&](auto lc) __attribute__((always_inline)) {
                    constexpr int l2 = decltype(lc)::value;
                    const unsigned ra_ = rd_addr;
                    PSND_R_LD(z[l2], ra_, l2 * 8);
                });
                asm volatile("" ::: "memory");
                if (ln2 == 0) lds_store(lock, 0);                 // (in-order LDS: the buffer has been read out when this lands)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PSND_R_TIEZ(0);
                PSND_R_TIEZ(16);
            }
#undef PSND_R_TIEZ
#undef PSND_R_WAIT8
#undef PSND_R_LD
            // ---- second radix-32: slot bitrev(q2) holds Zh[lam + 32 q2] = Z[64 q2 + c] ----------------------------------------------------
            if constexpr (!(PSND_R_ABL & 8)) pk::fft<32>(z);
            // ---- real-FFT split: own lower 16 (q2 = j) against the partner lane's upper 16 (q2 = 31 - j) ---------------------------------
            {
                const int ln4 = fresh_lane();
                const bool special = ln4 == 0;
                const v2f vL = *reinterpret_cast<const v2f *>(s_cl + 128 + 2 * ln4);
                const int paddr = ((ln4 >> 5) == 0 ? ((32 - ln4) & 31) : (95 - ln4)) * 4;
                if constexpr (PSND_R_ABL & 16) {
                    static_for<0, 16>([&](auto jc) __attribute__((always_inline)) {
                        constexpr int j = decltype(jc)::value;
                        mlo[j] = z[j].x + z[j].y, mhi[j] = z[16 + j].x + z[16 + j].y;
                    });
                } else
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
                        const v2f s = pk::fma(zp, v2f{1.f, -1.f}, za);
                        const v2f d = pk::fma(zp, v2f{-1.f, 1.f}, za);
                        const v2f e = pk::cmul(cmul_ct<j, 64>(d), vL);
                        const v2f xk = s + e, xc = s - e;
                        mlo[j] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.x, xk.x, __builtin_fmaf(xk.y, xk.y, p.mag_eps)));
                        mhi[j] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xc.x, xc.x, __builtin_fmaf(xc.y, xc.y, p.mag_eps)));
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                const v2f vMid = *reinterpret_cast<const v2f *>(s_cl + 256);
                const v2f mid = z[ct::bitrev(16, 5)];
                v2f xk, xc;
                rfft_pair_pk(mid, mid, vMid, xk, xc);
                mext = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.x, xk.x, __builtin_fmaf(xk.y, xk.y, p.mag_eps)));
            }
            // ---- the next frame's chunk, then this frame's spectrum: K contiguous floats, stored straight from registers -----------------
            {
                const int rn = u + kWaves + kLead;
                pend = -1;
                if (u + kWaves < nfr && rn < nch) {
                    wait_slot(rn);
                    if (!(PSND_R_ABL & 64)) request_chunk(rn);
                    pend = rn;
                }
            }
            if (!nostore) {
                const int ln = fresh_lane();
                const int lam_ = ln & 31, g_ = ln >> 5;
                const __amdgpu_buffer_rsrc_t ro = make_uniform_rsrc(p.mag + ((size_t)clip * (size_t)F + (size_t)(fa + u)) * kK, kK * 4);
                const int vlo = (64 * g_ + 2 * lam_) * 4;
                const int vhi = (2047 - 64 * 14 - 64 * g_ - 2 * lam_) * 4;
                static_for<0, 8>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = 2 * decltype(jc)::value;
                    // lanes < 32 hold c = 2 lam, lanes >= 32 c = 2 lam + 1: after the swap a lane of the lower half owns the bin PAIR of row
                    // j, a lane of the upper half the pair of row j + 1
                    float a0 = mlo[j], a1 = mlo[j + 1], b0 = mhi[j], b1 = mhi[j + 1];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2f{a0, a1}), ro, vlo, 256 * j, PSND_R_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2f{b1, b0}), ro, vhi, 256 * (14 - j), PSND_R_STORE_AUX);
                });
                if (ln == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mext), ro, 1024 * 4, 0, 0);
            }
        }
    }
}

}  // namespace

bool psnd_stft4096r_ok(long long T, long long F, int hop, int pad) {
    // hop = one ring chunk; 16-byte aligned chunk starts inside every clip (LDS-DMA moves 16 bytes per lane); 32-bit reflect indices
    return hop == kHop && pad % 4 == 0 && T % 4 == 0 && F > 0 && T + 2ll * kNFFT < (1ll << 31) && F < (1ll << 31);
}

int psnd_stft4096r_launch(const float *wav, const float *plan, float *mag_nfk, long long N, long long T, long long F, int pad, float mag_eps,
                          int ablate, hipStream_t stream) {
    if ((reinterpret_cast<uintptr_t>(wav) & 15) != 0) PSND_FAIL(PSND_E_ARG, "stft_mag_nfk(n4096r): waveform not 16-byte aligned");
    RParams p;
    p.wav = wav, p.plan = plan, p.mag = mag_nfk, p.T = (int)T, p.F = (int)F, p.pad = pad, p.mag_eps = mag_eps, p.ablate = ablate;
    if (N * F >= (1ll << 31)) PSND_FAIL(PSND_E_SHAPE, "stft_mag_nfk(n4096r): too many frames");
    p.total_frames = (int)(N * F);
    long long want = (p.total_frames + kWaves - 1) / kWaves;                  // at least one frame per wave
    int grid = want < 256 ? (int)want : 256;                                  // one persistent workgroup per CU
    if (const char *e = PSND_ENV("PSND_STFT4096_GRID")) grid = psnd_env_int(e, grid, 1, 65535);
    constexpr size_t lds = sizeof(float) * kLdsFloats;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_fwd_n4096r_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_mag_nfk(n4096r): set LDS size: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(stft_fwd_n4096r_kernel, dim3(grid), dim3(1024), lds, stream, p);
    PSND_CHECK_LAUNCH("stft_mag_nfk(n4096r)");
    return PSND_OK;
}
