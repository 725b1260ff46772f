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
//   * ring of 16 slots (64 KB), one LOADER wave (wave 15) and 15 frame waves.  The loader walks the chunks in order, four at a time:
//     one poll tells it that the four slots are free (a slot is free once every reader of the chunk 16 back has taken its samples:
//     left[slot] == 0), 16 LDS-DMA instructions go out, and behind `s_waitcnt vmcnt(16)` - its own transfers, counted in order - the
//     previous batch is published with two LDS stores (left[slot] = number of frames that read the chunk, ready[slot] = chunk + 1).
//     First version (profiles/r05_stft4096r_trace.txt): every frame wave requested and published "its" chunk from inside its transform;
//     a third of a frame's 24 k cycles went into waiting for tags and slots, because a request or a publish happened only when its wave
//     came by;
//   * the frame waves take the segment's frames IN ORDER from an LDS counter (frames start in the order their samples arrive; the SIMD
//     that carries the loader simply takes fewer).  A frame waits for its four chunks' tags, copies its samples to registers (32
//     ds_read_b64) and takes itself off the chunks' reader counts.  Every dependency points to an EARLIER frame of the segment: no
//     cycle, no workgroup barrier in the frame loop, the waves drift as they like;
//   * clip edges (reflect padding, transforms.py:55-60; the first two and the last three chunks of a clip) are gathered element by
//     element by the loader wave - the frames themselves never see an edge;
//   * to make room for the ring the sixteen 8.3 KB per-wave transpose buffers are gone: the 32 x 32 transposes of both half-waves go
//     through one of FOUR 16.9 KB buffers (one per SIMD) in a single burst of 32 ds_write_b64 + 32 ds_read_b64 over all 64 lanes, taken
//     under a lock (an LDS compare-and-swap; a wave holds it for ~1 k cycles of a ~20 k-cycle frame).  Full-wave 8-byte accesses cost
//     256 LDS cycles per frame against 512 for the half-masked rounds of psnd_stft_w.hip; every other LDS read of the transform (samples,
//     window, twiddles) is an explicit ds_read_b64 - hipcc fuses neighbours into ds_read2_b64, which the LDS serves at half the rate;
//   * LDS: 25.6 KB of tables + 64 KB ring + 67.6 KB transpose buffers + flags = 157.4 KB, one 1024-thread workgroup per CU.
//
// The transform itself (radix-2 in lane, v_permlane32_swap, radix-32, 32 x 32 transpose per half-wave, radix-32, real-FFT split through
// ds_bpermute, magnitudes stored straight from registers) is the one of psnd_stft_w.hip; see there for the index algebra.
// Bound: HBM (4 hop + 4 K = 12 292 B per frame); DESIGN.md 4.1 for the measured fraction.
#include "psnd_pk.h"
#include "psnd_stft_pass.h"
#include "psnd_stft_w.h"
#include <stdlib.h>

#ifndef PSND_R_STORE_AUX
#define PSND_R_STORE_AUX 2     // cache-policy bits of the output stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef PSND_R_LOADER_PRIO
#define PSND_R_LOADER_PRIO 3   // wave priority of the loader wave - the wave every frame wave waits for (round 6: 123.5 -> 122.0 us, same-box medians)
#endif
#ifndef PSND_R_BATCH
#define PSND_R_BATCH 4         // ring chunks the loader wave requests per poll (4 LDS-DMA instructions each); two batches in flight
#endif

#ifndef PSND_R_EARLY
#define PSND_R_EARLY 0         // 1: the next frame's index and chunk tags are asked for between the two halves of the stores (measured: 135 against 130 us)
#endif
#ifndef PSND_R_BPIPE
#define PSND_R_BPIPE 1         // the split's ds_bpermute one pair of bins ahead of the arithmetic
#endif
#ifdef PSND_R_NOSB             // A/B builds: no scheduling fences between the groups of a phase
#define PSND_R_SB()
#else
#define PSND_R_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef PSND_R_ABL
#define PSND_R_ABL 0           // timing ablations (tools/r05): 1 no transpose, 2 no sample reads, 4 no window, 8 no radix-32s, 16 no split, 32 no polls, 64 no transfers
#endif

namespace {
using namespace psnd_stft;

constexpr int kC = 2048, kNFFT = 4096, kK = 2049, kHop = 1024;
constexpr int kWaves = 16;                        // waves per workgroup = frames in flight
constexpr int kSlots = 16;                        // ring slots of one hop each
constexpr int kBatch = PSND_R_BATCH;              // chunks the loader wave requests per poll; two batches in flight
constexpr int kXP = 33;                           // transpose row pitch (complex values)
constexpr int kXbFloats = 2 * 32 * kXP * 2;       // one transpose buffer: [half-wave][q1][lam] (re, im), 16.9 KB
constexpr int kXbN = 4;                           // four of them: one per SIMD, shared by the four waves of that SIMD under a lock
constexpr int kOffWin = 0;                        // [32 loads][64 lanes] of (0.5 w[2n], 0.5 w[2n+1]) in the order a lane takes its samples
constexpr int kOffTw = 4096;                      // W_1024^(lam q1) as [q1][lam] (re, im)
constexpr int kOffCl = kOffTw + 2048;             // per lane: cL = W_2048^lam (-i)^g, then v_c = -i W_4096^c (c = 2 lam + g), v_(C/2)
constexpr int kOffRing = kOffCl + 260;
constexpr int kOffXa = kOffRing + kSlots * kHop;
constexpr int kOffFlags = kOffXa + kXbN * kXbFloats;        // int ready[16] | int left[16] | int lock[4] | int next
constexpr int kLdsFloats = kOffFlags + 2 * kSlots + kXbN + 1;
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");
static_assert(kBatch >= 1 && 4 * kBatch <= 60 && 2 * kBatch <= kSlots - 4, "vmcnt is a 6-bit counter; two batches next to a frame's four chunks in the ring");
static_assert(kOffRing % 4 == 0, "16-byte aligned ring slots (LDS-DMA writes 16 bytes per lane)");

struct RParams {
    const float *wav;
    const float *plan;
    float *mag;
    int T, F, total_frames;                       // (< 2^31: checked at launch)
    int pad;
    float mag_eps;
    int ablate;                                   // debug (PSND_ABLATE): 2 = no global stores
#ifdef PSND_R_TRACE
    long long *trace;                             // tools/r05/trace_r.py: s_memtime stamps [block][wave][16] of the wave's frame number trace_iter
    int trace_iter;
#endif
};
#ifdef PSND_R_TRACE
#define PSND_R_STAMP(i_)                                                                                       \
    do {                                                                                                       \
        if (p.trace && titer == p.trace_iter) {                                                                \
            const long long tm_ = __builtin_amdgcn_s_memtime();                                                \
            if ((threadIdx.x & 63) == 0) p.trace[((size_t)blockIdx.x * 16 + w) * 16 + (i_)] = tm_;              \
        }                                                                                                      \
    } while (0)
#else
#define PSND_R_STAMP(i_)
#endif

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
#ifdef PSND_R_TRACE
    if (p.trace && (t & 63) == 0) p.trace[((size_t)blockIdx.x * 16 + w) * 16 + 14] = __builtin_amdgcn_s_memtime();
#endif

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
    int *s_lock = s_left + kSlots, *s_next = s_lock + kXbN;
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

        // chunk r of the segment -> ring slot r % kSlots.  Interior: four 1-KB LDS-DMA instructions (returns true); clip edge: reflect gather.
        auto request_chunk = [&](int r) __attribute__((always_inline)) -> bool {
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
                return true;
            }
            float *dst = s_ring + slot * kHop;
#pragma unroll 1
            for (int e = fresh_lane(); e < kHop; e += 64) dst[e] = xclip[reflect_idx32(g0 + e, p.T)];
            return false;
        };
        // tell the readers that chunks [c0, c1) are in their slots (c1 - c0 <= 64): per chunk the number of frames that take samples from it,
        // then the tag - two LDS instructions for the lot
        auto publish_chunks = [&](int c0, int c1) __attribute__((always_inline)) {
            const int r = c0 + fresh_lane();
            const bool mine = r < c1;
            const int lo = r - 3 > 0 ? r - 3 : 0, hi = r < nfr - 1 ? r : nfr - 1;
            const int slot = r & (kSlots - 1);
            if (mine) lds_store(s_left + slot, hi - lo + 1);
            asm volatile("" ::: "memory");
            if (mine) lds_store(s_ready + slot, r + 1);
        };

        __syncthreads();                                        // the previous segment (or the tables) is done with: ring and flags are free
        if (t < 2 * kSlots) s_ready[t] = 0;
        if (t == 0) *s_next = 0;
        __syncthreads();                                        // flags reset before anyone publishes

        const bool nostore = PSND_ABL(p, 2);
        if (w == kWaves - 1) {
#if PSND_R_LOADER_PRIO
            __builtin_amdgcn_s_setprio(PSND_R_LOADER_PRIO);     // the wave every frame wave waits for
#endif
            // ---- THE LOADER WAVE: chunks in order, kBatch at a time, two batches in flight; a slot is free once every reader of the chunk 16
            //      back is through.  vmcnt counts this wave's transfers in order (4 instructions per chunk): everything in front of the newest
            //      batch has landed when at most 4 kBatch instructions are outstanding.  Clip-edge chunks (gathered with ordinary loads) and the
            //      last, short batch drain the queue.
            int head = 0;                                       // chunks [head, r0) are requested and not yet published
            for (int r0 = 0; r0 < nch; r0 += kBatch) {
                const int nb = nch - r0 < kBatch ? nch - r0 : kBatch;
#ifdef PSND_R_TRACE
                const int titer = r0 / kBatch - 10;             // (loader trace: batch 10 + trace_iter of the segment)
#endif
                PSND_R_STAMP(0);
                if (r0 + nb > kSlots && !(PSND_R_ABL & 32)) {   // one poll for the batch: lane i looks at the slot of chunk r0 + i
                    const int ln = fresh_lane(), rr = r0 + ln;
                    const int slot = rr & (kSlots - 1), tag = rr - kSlots + 1;
                    for (;;) {
                        bool ok = true;
                        if (ln < nb && rr >= kSlots) ok = lds_load(s_ready + slot) == tag && lds_load(s_left + slot) == 0;
                        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                PSND_R_STAMP(1);
                bool all_dma = true;
                for (int j = 0; j < nb; ++j) all_dma &= (PSND_R_ABL & 64) ? true : request_chunk(r0 + j);
                PSND_R_STAMP(2);
                if (!all_dma || nb < kBatch) {                  // a gathered chunk (ordinary loads + LDS stores) or a short batch: wait for everything
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    publish_chunks(head, r0 + nb);
                    head = r0 + nb;
                } else if (head < r0) {                         // everything in front of this batch's 4 kBatch instructions has landed
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * kBatch) : "memory");
                    PSND_R_STAMP(3);
                    publish_chunks(head, r0);
                    head = r0;
                }
                PSND_R_STAMP(4);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (head < nch) publish_chunks(head, nch);
#if PSND_R_LOADER_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            continue;                                           // next segment (its first barrier waits for the frame waves)
        }
        int titer = -1;
        (void)titer;
        // ---- THE FRAME WAVES take the segment's frames in order from a counter: frames start in the order their samples arrive, a SIMD that
        //      carries the loader (three frame waves) simply takes fewer.  The NEXT frame's index, chunk tags and samples are asked for
        //      between the stores of the current one (the data registers are dead by then): their LDS round trips - a few hundred cycles
        //      each under load - run while the vector-memory unit takes the stores.
        // LDS reads of these phases are explicit ds_read_b64 (hipcc fuses neighbouring ones into ds_read2_b64, which the LDS serves at half
        // the rate: MI355X_MICROARCH.md, LDS table); the values become usable behind a wait that names them (PSND_R_WAIT*)
#define PSND_R_LD(dst_, addr_, off_) lds_rd64<(off_)>(dst_, addr_)
#define PSND_R_WAIT8(a_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(a_[4]), "+v"(a_[5]), "+v"(a_[6]), "+v"(a_[7]))
#define PSND_R_TIEZ(o_)                                                                                                                 \
    asm volatile("" : "+v"(z[o_]), "+v"(z[o_ + 1]), "+v"(z[o_ + 2]), "+v"(z[o_ + 3]), "+v"(z[o_ + 4]), "+v"(z[o_ + 5]), "+v"(z[o_ + 6]),   \
                 "+v"(z[o_ + 7]), "+v"(z[o_ + 8]), "+v"(z[o_ + 9]), "+v"(z[o_ + 10]), "+v"(z[o_ + 11]), "+v"(z[o_ + 12]), "+v"(z[o_ + 13]),  \
                 "+v"(z[o_ + 14]), "+v"(z[o_ + 15]))
        auto grab_issue = [&]() __attribute__((always_inline)) {
            int got = 0;
            if (fresh_lane() == 0) got = __hip_atomic_fetch_add(s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return got;
        };
        auto tags_issue = [&](int u_) __attribute__((always_inline)) {          // lane i looks at the tag of chunk u + (i & 3)
            const int idx = fresh_lane() & 3;
            return lds_load(s_ready + ((u_ + idx) & (kSlots - 1))) - (u_ + idx + 1);      // 0 = there
        };
        auto wait_chunks = [&](int u_, int first) __attribute__((always_inline)) {
            if (PSND_R_ABL & 32) return;
            int d = first;
            while (__builtin_amdgcn_ballot_w64(d == 0) != ~0ull) {
                __builtin_amdgcn_s_sleep(2);
                d = tags_issue(u_);
            }
        };
        // (always executed, whether frame u_ exists or not: a conditional definition of the 64 data registers costs a second set of them)
        auto issue_samples = [&](int u_, bool exists) __attribute__((always_inline)) {
            // lane (lam, g): complex point n = lam + 32 (a + 16 g) = samples 2 n, 2 n + 1 -> chunk u + g, offset 2 lam + 64 a;
            // point n + 1024 -> chunk u + g + 2
            asm volatile("" ::: "memory");
            const int ln = fresh_lane(), lam_ = ln & 31, g_ = ln >> 5;
            const unsigned lo = static_cast<unsigned>(reinterpret_cast<uintptr_t>(s_ring + ((u_ + g_) & (kSlots - 1)) * kHop + 2 * lam_));
            const unsigned hi = static_cast<unsigned>(reinterpret_cast<uintptr_t>(s_ring + ((u_ + g_ + 2) & (kSlots - 1)) * kHop + 2 * lam_));
            static_for<0, 16>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                if constexpr (PSND_R_ABL & 2) {
                    z[a] = v2f{(float)a, (float)ln}, z[16 + a] = v2f{(float)ln, (float)a};
                } else {
                    PSND_R_LD(z[a], lo, 256 * a);
                    PSND_R_LD(z[16 + a], hi, 256 * a);
                }
            });
            // off the reader counts (LDS operations of one wave execute in order: behind the reads above)
            asm volatile("" ::: "memory");
            if (exists && ln < 4) __hip_atomic_fetch_add(s_left + ((u_ + ln) & (kSlots - 1)), -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        int u = __builtin_amdgcn_readfirstlane(grab_issue());
        if (u < nfr) wait_chunks(u, tags_issue(u));
        issue_samples(u, u < nfr);
        while (u < nfr) {
            ++titer;
            PSND_R_STAMP(0);
            PSND_R_STAMP(1);
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
                PSND_R_STAMP(2);
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
                    PSND_R_SB();
                });
            }
            PSND_R_STAMP(3);
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
                PSND_R_STAMP(4);
                if constexpr (!(PSND_R_ABL & 8)) pk::fft<32>(z);
                PSND_R_STAMP(5);
                PSND_R_WAIT8(tb[0]);
                static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (c < 3) ld_tw(std::integral_constant<int, c + 1>{}, tb[(c + 1) & 1]);
                    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int i = decltype(qc)::value, q1 = c * 8 + i, sl = ct::bitrev(q1, 5);
                        if constexpr (q1 != 0) z[sl] = pk::cmul(z[sl], tb[c & 1][i]);
                    });
                    if constexpr (c < 3) PSND_R_WAIT8(tb[(c + 1) & 1]);
                    PSND_R_SB();
                });
            }
            PSND_R_STAMP(6);
            // ---- 32 x 32 transpose per half-wave, both at once (ds_write_b64 / ds_read_b64 over all 64 lanes: a third of the LDS cycles of
            //      half-masked or component-wise rounds), through one of four 16.9 KB buffers - the one of this wave's SIMD, taken under a lock:
            //      a wave holds it for one burst of 64 LDS instructions per frame
            PSND_R_STAMP(7);
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
                PSND_R_STAMP(8);
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
            PSND_R_STAMP(9);
            // ---- second radix-32: slot bitrev(q2) holds Zh[lam + 32 q2] = Z[64 q2 + c] ----------------------------------------------------
            if constexpr (!(PSND_R_ABL & 8)) pk::fft<32>(z);
            PSND_R_STAMP(10);
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
                } else {
                    // the partners' values of pair j + 2, j + 3 travel (ds_bpermute) while pair j, j + 1 is evaluated
                    v2f zb[2][2];
                    auto fetch = [&](auto cc, v2f (&d)[2]) __attribute__((always_inline)) {
                        static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(cc)::value * 2 + decltype(jc)::value;
                            const v2f snd = z[ct::bitrev(31 - j, 5)];
                            const v2f own = z[ct::bitrev(j == 0 ? 0 : 32 - j, 5)];
                            const v2f got = v2f{bperm(paddr, snd.x), bperm(paddr, snd.y)};
                            d[decltype(jc)::value] = special ? own : got;
                        });
                    };
                    if constexpr (PSND_R_BPIPE) fetch(std::integral_constant<int, 0>{}, zb[0]);
                    static_for<0, 8>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr (!PSND_R_BPIPE) fetch(cc, zb[c & 1]);
                        else if constexpr (c < 7) fetch(std::integral_constant<int, c + 1>{}, zb[(c + 1) & 1]);
                        static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = c * 2 + decltype(jc)::value;
                            const v2f za = z[ct::bitrev(j, 5)], zp = zb[c & 1][decltype(jc)::value];
                            const v2f s = pk::fma(zp, v2f{1.f, -1.f}, za);
                            const v2f d = pk::fma(zp, v2f{-1.f, 1.f}, za);
                            const v2f e = pk::cmul(cmul_ct<j, 64>(d), vL);
                            const v2f xk = s + e, xc = s - e;
                            mlo[j] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.x, xk.x, __builtin_fmaf(xk.y, xk.y, p.mag_eps)));
                            mhi[j] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xc.x, xc.x, __builtin_fmaf(xc.y, xc.y, p.mag_eps)));
                        });
                        PSND_R_SB();
                    });
                }
                const v2f vMid = *reinterpret_cast<const v2f *>(s_cl + 256);
                const v2f mid = z[ct::bitrev(16, 5)];
                v2f xk, xc;
                rfft_pair_pk(mid, mid, vMid, xk, xc);
                mext = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.x, xk.x, __builtin_fmaf(xk.y, xk.y, p.mag_eps)));
            }
            PSND_R_STAMP(11);
            // ---- this frame's spectrum: K contiguous floats, stored straight from registers; the next frame's requests in between ------------
            PSND_R_STAMP(12);
            int got = 0;
            if constexpr (PSND_R_EARLY) got = grab_issue();
            const int ln = fresh_lane();
            const int lam_ = ln & 31, g_ = ln >> 5;
            const __amdgpu_buffer_rsrc_t ro = make_uniform_rsrc(p.mag + ((size_t)clip * (size_t)F + (size_t)(fa + u)) * kK, kK * 4);
            const int vlo = (64 * g_ + 2 * lam_) * 4;
            const int vhi = (2047 - 64 * 14 - 64 * g_ - 2 * lam_) * 4;
            auto store_pairs = [&](auto j0c) __attribute__((always_inline)) {
                static_for<0, 4>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(j0c)::value + 2 * decltype(jc)::value;
                    // lanes < 32 hold c = 2 lam, lanes >= 32 c = 2 lam + 1: after the swap a lane of the lower half owns the bin PAIR of row
                    // j, a lane of the upper half the pair of row j + 1
                    float a0 = mlo[j], a1 = mlo[j + 1], b0 = mhi[j], b1 = mhi[j + 1];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2f{a0, a1}), ro, vlo, 256 * j, PSND_R_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2f{b1, b0}), ro, vhi, 256 * (14 - j), PSND_R_STORE_AUX);
                });
            };
            if (!nostore) store_pairs(std::integral_constant<int, 0>{});
            int un = 0, d0 = 0;
            if constexpr (PSND_R_EARLY) {
                un = __builtin_amdgcn_readfirstlane(got);
                if (un < nfr) d0 = tags_issue(un);
            }
            if (!nostore) {
                store_pairs(std::integral_constant<int, 8>{});
                if (ln == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mext), ro, 1024 * 4, 0, 0);
            }
            if constexpr (!PSND_R_EARLY) {
                un = __builtin_amdgcn_readfirstlane(grab_issue());
                if (un < nfr) d0 = tags_issue(un);
            }
            PSND_R_STAMP(13);
            if (un < nfr) wait_chunks(un, d0);
            issue_samples(un, un < nfr);
            u = un;
        }
#undef PSND_R_TIEZ
#undef PSND_R_WAIT8
#undef PSND_R_LD
#ifdef PSND_R_TRACE
        if (p.trace && (t & 63) == 0) p.trace[((size_t)blockIdx.x * 16 + w) * 16 + 15] = __builtin_amdgcn_s_memtime();
#endif
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
#ifdef PSND_R_TRACE
    {
        const char *tp = PSND_ENV("PSND_R_TRACE_PTR"), *ti = PSND_ENV("PSND_R_TRACE_ITER");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
        p.trace_iter = ti ? atoi(ti) : 4;
    }
#endif
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
