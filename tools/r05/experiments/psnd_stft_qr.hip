// psnd_stft_qr.hip - n_fft = 1024, hop = 256 forward STFT, magnitude in the BIN-FASTEST layout (N, F, K) of psnd_stft_mag_nfk: the
// wave-owns-four-frames transform of psnd_stft_q.hip fed from a SAMPLE RING in LDS that a loader wave fills (round 5; the protocol of
// psnd_stft_r.hip at this size).
//
// Replaces STFT.transform (pytorch_sound/models/transforms.py:53-69) for the settings.py defaults (1024 / 256) where the consumer takes
// (N, F, K).
//
// Why.  stft_fwd_n1024q_kernel requests a quad's span of samples (LDS-DMA into the wave's own buffer) when the buffer is free - after the
// previous quad's magnitudes have been read out of it - and needs it at once: the transfer's latency is exposed in every quad, covered
// only by the three other waves of the SIMD.  With the loads ablated the kernel runs 121 -> 107 us (1024 clips x 2 s, tools/r05).  Here
//   * wave 15 is a LOADER: the samples of the workgroup's run of quads (a "segment": consecutive quads of one clip) enter a ring of 16
//     chunks of 1024 samples (one quad = 4 hops of new samples) by LDS-DMA, four chunks per poll, two batches in flight, published with
//     tags and reader counts exactly as in psnd_stft_r.hip; every sample is fetched once ((nq + 1) / nq of the algorithmic reads
//     instead of 1.75 x through the L2);
//   * the 15 quad waves take quads in order from an LDS counter; a quad waits for the tags of its two chunks and reads its taps straight
//     from the ring (the 32-float skew per 256 samples that keeps the two frames of a 32-lane group on disjoint banks is applied by the
//     transfer: each 1-KB DMA instruction lands 1152 bytes after the previous one);
//   * to make room for the ring (72 KB) the per-wave buffer shrinks from 9.2 to 4.6 KB: the exchange between the passes goes in FOUR
//     rounds of 8 rows (the lanes whose row pair lies in the round read, explicit ds_read_b128 under the exec mask into the same
//     registers), and the four spectra are staged in two halves - post_emit_pk's first sweep writes bins 0 .. 256 only, its second
//     bins 257 .. 512 only - each read out as 16 bytes per lane and stored as 1 KiB runs per frame.
// LDS: 9.7 KB tables + 72 KB ring + 15 x 4.5 KB + flags = 151 KB, one 1024-thread workgroup per CU.
// Bound: HBM, 4 hop + 4 K = 3076 B per frame (DESIGN.md 4.1f for the measured fraction).
#include "psnd_pk.h"
#include "psnd_stft_pass.h"
#include "psnd_stft_emit.h"
#include "psnd_stft_q.h"
#include <stdlib.h>

#ifndef PSND_QR_STORE_AUX
#define PSND_QR_STORE_AUX 2    // cache-policy bits of the output stores (gfx950: 2 = nt)
#endif
#ifndef PSND_QR_BATCH
#define PSND_QR_BATCH 4
#endif

namespace {
using namespace psnd_stft;

constexpr int kR1 = 32, kL = 16, kC = 512, kNFFT = 1024, kK = 513, kHop = 256;
constexpr int kWaves = 16, kSlots = 16, kBatch = PSND_QR_BATCH;
constexpr int kChunk = 1024;                       // samples per ring chunk = one quad of hops
constexpr int kBlk = 256 + 32;                     // LDS floats per 256 samples (the skew)
constexpr int kSlotF = 4 * kBlk;                   // 1152 floats per slot
constexpr int kRow = 2 * kR1 + 4;                  // pitch of the window / twiddle tables (plan layout, psnd_stft_plan.h)
constexpr int kVkp = (2 * (kC / 2 + 1) + 3) & ~3;
constexpr int kTab = 2 * kL * kRow + kVkp;         // wt[16][68] | tw[16][68] | vk[257](re, im)
constexpr int kRP = 2 * kL + 4;                    // exchange row pitch (floats): 16 complex + 4
constexpr int kXF = 8 * kRP;                       // one frame's quarter exchange (8 rows): 288 floats
constexpr int kXW = 4 * kXF;                       // a wave's buffer: 1152 floats = 4608 B (quarter exchange | half of the staged spectra)
constexpr int kStgP = 260;                         // staging pitch per frame (floats): 257 bins of a half, rows 16-byte aligned
static_assert(4 * kStgP <= kXW, "half of the four spectra fits the wave's buffer");
constexpr int kOffRing = kTab;
constexpr int kOffWork = kOffRing + kSlots * kSlotF;
constexpr int kOffFlags = kOffWork + (kWaves - 1) * kXW;    // int ready[16] | left[16] | next
constexpr int kLdsFloats = kOffFlags + 2 * kSlots + 1;
static_assert(kLdsFloats * 4 <= 160 * 1024 && kOffRing % 4 == 0 && kOffWork % 4 == 0, "LDS budget / 16-byte alignment");
static_assert(4 * kBatch <= 60 && 2 * kBatch <= kSlots - 2, "vmcnt is a 6-bit counter; two batches next to a quad's two chunks");

struct QRParams {
    const float *wav;
    const float *plan;
    float *mag;
    int T, F, nq, total_quads;                     // nq = quads per clip
    int pad;
    float mag_eps;
    int ablate;
};

template <int OFF>
__device__ __forceinline__ void lds_rd64(v2f &dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_rd128_tied(f32x4 &dst, unsigned addr) {          // (under an exec mask: the other lanes keep their value)
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ int lds_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// magnitude writer into the wave's half-staging buffer: post_emit_pk's byte offsets (frame * 4 K + 4 bin) minus `adj` (per lane: the frame's
// 1012-byte pitch difference, plus 1028 for the upper half)
struct EmitHalf {
    char *buf;
    const int &adj;
    v2f eps2;
    template <bool CONJ>
    __device__ __forceinline__ OutVal make(v2f x) const {
        OutVal o;
        const v2f sq = pk::fma(x, x, eps2);
        o.m = __builtin_amdgcn_sqrtf(sq.x + sq.y);
        return o;
    }
    __device__ __forceinline__ void store(int voff, int soff, const OutVal &o) const {
        *reinterpret_cast<float *>(buf + voff + soff - adj) = o.m;
    }
};

__global__ __launch_bounds__(1024, 1) void stft_fwd_n1024r_kernel(QRParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_wt = smem, *s_tw = smem + kL * kRow, *s_vk = smem + 2 * kL * kRow, *s_ring = smem + kOffRing;
    int *s_ready = reinterpret_cast<int *>(smem + kOffFlags), *s_left = s_ready + kSlots, *s_next = s_left + kSlots;
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    auto fresh_lane = [&]() __attribute__((always_inline)) {
        int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(ln));
        return ln;
    };
    for (int i = t; i < kTab / 4; i += 1024) reinterpret_cast<f32x4 *>(smem)[i] = reinterpret_cast<const f32x4 *>(p.plan)[i];
    typedef __attribute__((address_space(3))) char *lds_ptr;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((lds_ptr)s_ring));
    float *xw = smem + kOffWork + (w < kWaves - 1 ? w : 0) * kXW;

    const int ga = (int)((long long)blockIdx.x * p.total_quads / gridDim.x);
    const int gb = (int)((long long)(blockIdx.x + 1) * p.total_quads / gridDim.x);
    int clip = ga / p.nq, q0 = ga - clip * p.nq;                // (the only division: later segments start at quad 0 of the next clip)

    for (int g = ga; g < gb; ++clip, q0 = 0) {
        const int rest = gb - g, room = p.nq - q0;
        const int nqs = rest < room ? rest : room;               // quads of this segment: q0 .. q0 + nqs - 1 of `clip`
        const int nch = nqs + 1;                                 // chunks q0 .. q0 + nqs (chunk c = padded samples [1024 c, 1024 c + 1024))
        g += nqs;
        const float *xclip = p.wav + (size_t)clip * (size_t)p.T;

        __syncthreads();                                        // the previous segment (or the tables) is done with
        if (t < 2 * kSlots) s_ready[t] = 0;
        if (t == 0) *s_next = 0;
        __syncthreads();

        if (w == kWaves - 1) {
            // ---- THE LOADER WAVE (psnd_stft_r.hip): chunks in order, kBatch per poll, two batches in flight ---------------------------
            auto request_chunk = [&](int r) __attribute__((always_inline)) -> bool {
                const int g0 = (q0 + r) * kChunk - p.pad;
                const int slot = r & (kSlots - 1);
                if (g0 >= 0 && g0 + kChunk <= p.T) {
                    const unsigned long long a = reinterpret_cast<unsigned long long>(xclip + g0);
                    u32x4 rs;
                    rs.x = __builtin_amdgcn_readfirstlane((unsigned)a);
                    rs.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
                    rs.z = 4 * kChunk;
                    rs.w = 0x00020000u;
                    const int voff = fresh_lane() * 16;
                    const unsigned dst0 = ring_lds + (unsigned)slot * (4u * kSlotF);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                                   // 256 samples per instruction, 288 floats apart in LDS
                        const unsigned dst = dst0 + 4u * kBlk * j;
                        const int soff = 1024 * j;
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                     :: "s"(dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
                    }
                    return true;
                }
                float *dst = s_ring + slot * kSlotF;
#pragma unroll 1
                for (int e = fresh_lane(); e < kChunk; e += 64) dst[e + 32 * (e >> 8)] = xclip[reflect_idx32(g0 + e, p.T)];
                return false;
            };
            auto publish_chunks = [&](int c0, int c1) __attribute__((always_inline)) {
                const int r = c0 + fresh_lane();
                const bool mine = r < c1;
                const int readers = (r >= 1 ? 1 : 0) + (r < nqs ? 1 : 0);             // quads r - 1 and r
                const int slot = r & (kSlots - 1);
                if (mine) lds_store(s_left + slot, readers);
                asm volatile("" ::: "memory");
                if (mine) lds_store(s_ready + slot, r + 1);
            };
            int head = 0;
            for (int r0 = 0; r0 < nch; r0 += kBatch) {
                const int nb = nch - r0 < kBatch ? nch - r0 : kBatch;
                if (r0 + nb > kSlots) {
                    const int ln = fresh_lane(), rr = r0 + ln;
                    const int slot = rr & (kSlots - 1), tag = rr - kSlots + 1;
                    for (;;) {
                        bool ok = true;
                        if (ln < nb && rr >= kSlots) ok = lds_load(s_ready + slot) == tag && lds_load(s_left + slot) == 0;
                        if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                bool all_dma = true;
                for (int j = 0; j < nb; ++j) all_dma &= request_chunk(r0 + j);
                if (!all_dma || nb < kBatch) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    publish_chunks(head, r0 + nb);
                    head = r0 + nb;
                } else if (head < r0) {
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * kBatch) : "memory");
                    publish_chunks(head, r0);
                    head = r0;
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (head < nch) publish_chunks(head, nch);
            continue;
        }

        // ---- THE QUAD WAVES ---------------------------------------------------------------------------------------------------------
        for (;;) {
            int u;
            {
                int got = 0;
                if (fresh_lane() == 0) got = __hip_atomic_fetch_add(s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                u = __builtin_amdgcn_readfirstlane(got);
            }
            if (u >= nqs) break;
            const int f0 = (q0 + u) * 4;
            const int left_f = p.F - f0;
            const int nval = left_f < 4 ? left_f : 4;
            {   // the quad's two chunks
                const int idx = fresh_lane() & 1;
                for (;;) {
                    const int v = lds_load(s_ready + ((u + idx) & (kSlots - 1)));
                    if (__builtin_amdgcn_ballot_w64(v == u + idx + 1) == ~0ull) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            asm volatile("" ::: "memory");
            // ---- pass 1: taps (sample 2 (l + 16 a) of frame fi: 256-sample block fi + a / 8 of the quad's span), window, radix-32 ---------
            v2f z[kR1];
            {
                const int ln = fresh_lane(), fi = ln >> 4, l = ln & 15;
                static_for<0, 4>([&](auto gc) __attribute__((always_inline)) {
                    constexpr int gi = decltype(gc)::value;
                    const int b = fi + gi;
                    const unsigned base = ring_lds + (unsigned)(((u + (b >> 2)) & (kSlots - 1)) * kSlotF + (b & 3) * kBlk + 2 * l) * 4u;
                    static_for<0, 8>([&](auto ac) __attribute__((always_inline)) {
                        constexpr int a = gi * 8 + decltype(ac)::value;
                        lds_rd64<128 * (a % 8)>(z[a], base);
                    });
                });
                asm volatile("" ::: "memory");
                if (ln < 2) __hip_atomic_fetch_add(s_left + ((u + ln) & (kSlots - 1)), -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]),
                             "+v"(z[8]), "+v"(z[9]), "+v"(z[10]), "+v"(z[11]), "+v"(z[12]), "+v"(z[13]), "+v"(z[14]), "+v"(z[15]));
                asm volatile("" : "+v"(z[16]), "+v"(z[17]), "+v"(z[18]), "+v"(z[19]), "+v"(z[20]), "+v"(z[21]), "+v"(z[22]), "+v"(z[23]),
                             "+v"(z[24]), "+v"(z[25]), "+v"(z[26]), "+v"(z[27]), "+v"(z[28]), "+v"(z[29]), "+v"(z[30]), "+v"(z[31]));
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float *wrow = s_wt + (fresh_lane() & 15) * kRow;
                static_for<0, kR1 / 2>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
                    z[2 * i] *= pk::lo(wv);
                    z[2 * i + 1] *= pk::hi(wv);
                    if constexpr (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                });
            }
            pk::fft<kR1>(z);
            __builtin_amdgcn_sched_barrier(0);

            // ---- exchange inside the wave in four rounds of 8 rows; a lane reads its row in the round that holds it --------------------------
            const int ln2 = fresh_lane(), fi = ln2 >> 4, l = ln2 & 15;
            const float *trow = s_tw + l * kRow;
            float *oz = xw + fi * kXF + 2 * l;
            const unsigned xw_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>(xw + fi * kXF));
            auto write_quarter = [&](auto qc) __attribute__((always_inline)) {
                constexpr int Q0 = decltype(qc)::value;
                static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int qa = Q0 + 2 * decltype(ic)::value, qb = qa + 1;
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(trow + 2 * qa);
                    constexpr int sa = ct::bitrev(qa, 5), sb = ct::bitrev(qb, 5);
                    if constexpr (qa == 0) *reinterpret_cast<v2f *>(oz) = z[sa];
                    else *reinterpret_cast<v2f *>(oz + (qa - Q0) * kRP) = pk::cmul(z[sa], pk::lo(wv));
                    *reinterpret_cast<v2f *>(oz + (qb - Q0) * kRP) = pk::cmul(z[sb], pk::hi(wv));
                });
                asm volatile("" ::: "memory");
            };
            f32x4 A[kL / 2], B[kL / 2];
#pragma unroll
            for (int i = 0; i < kL / 2; ++i) A[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto read_row8 = [&](f32x4 (&D)[kL / 2], int r) __attribute__((always_inline)) {     // row r (0 .. 7) of the round in the buffer
                const unsigned addr = xw_lds + (unsigned)(r * kRP) * 4u;
                static_for<0, kL / 2>([&](auto ic) __attribute__((always_inline)) { lds_rd128_tied<16 * decltype(ic)::value>(D[decltype(ic)::value], addr); });
            };
            const int rB = l == 0 ? 16 : 32 - l;                 // the second row of the lane's pair (row l is the first)
            write_quarter(std::integral_constant<int, 0>{});
            if (l < 8) read_row8(A, l);
            write_quarter(std::integral_constant<int, 8>{});
            if (l >= 8) read_row8(A, l - 8);
            write_quarter(std::integral_constant<int, 16>{});
#pragma unroll
            for (int i = 0; i < kL / 2; ++i) B[i] = f32x4{0.f, 0.f, 0.f, 0.f};      // (defined only now: 32 registers less while z is still whole)
            if (rB < 24) read_row8(B, rB - 16);
            write_quarter(std::integral_constant<int, 24>{});
            if (rB >= 24) read_row8(B, rB - 24);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]),
                         "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5]), "+v"(B[6]), "+v"(B[7]));
            v2f za[kL], zb[kL];
#pragma unroll
            for (int i = 0; i < kL / 2; ++i) za[2 * i] = pk::lo(A[i]), za[2 * i + 1] = pk::hi(A[i]), zb[2 * i] = pk::lo(B[i]), zb[2 * i + 1] = pk::hi(B[i]);
            __builtin_amdgcn_sched_barrier(0);
            pk::fft<kL>(za);
            pk::fft<kL>(zb);
            __builtin_amdgcn_sched_barrier(0);

            // ---- real-FFT split + magnitudes, the quad's four spectra in two halves through the wave's buffer ------------------------------
            const int ln3 = fresh_lane(), qq = ln3 & 15, fq = ln3 >> 4;
            const bool special = qq == 0;
            const bool nostore = p.ablate & 2;
            const __amdgpu_buffer_rsrc_t ro = make_uniform_rsrc(p.mag + ((size_t)clip * (size_t)p.F + (size_t)f0) * kK, nval * (kK * 4));
            // a store instruction = one frame's 1 KiB run (64 lanes x 16 bytes); frames past nval are dropped
            auto flush_half = [&](int byte0, bool with_mid) __attribute__((always_inline)) {
                f32x4 o[4];
                const char *src = reinterpret_cast<const char *>(xw) + ln3 * 16;
                static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { o[decltype(ic)::value] = *reinterpret_cast<const f32x4 *>(src + decltype(ic)::value * (kStgP * 4)); });
                float mid = 0.f;
                if (with_mid) mid = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xw) + (ln3 & 3) * (kStgP * 4) + 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // the buffer has been read out
                if (nostore) return;
                static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    const int off = i < nval ? i * (kK * 4) + byte0 + 16 * ln3 : (1 << 30);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[i]), ro, off, 0, PSND_QR_STORE_AUX);
                });
                if (with_mid) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mid), ro,
                                                                    (ln3 < 4 && ln3 < nval) ? ln3 * (kK * 4) + 1024 : (1 << 30), 0, 0);
            };
            int adj = fq * (kK * 4 - kStgP * 4);                                          // lower half: frame pitch 2052 -> 1040 bytes
            EmitHalf emit{reinterpret_cast<char *>(xw), adj, v2f{p.mag_eps, 0.f}};
            auto between = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                flush_half(0, true);                                                      // bins 0 .. 255 of every frame, and bin 256
                adj = fq * (kK * 4 - kStgP * 4) + 1028;                                    // upper half: bins 257 .. 512 at 0 .. 255
                __builtin_amdgcn_sched_barrier(0);
            };
            post_emit_pk<kR1, kL>(za, zb, special, qq, special ? kR1 / 2 : kR1 - qq, s_vk, emit, 1, fq * (kK * 4), between);
            __builtin_amdgcn_sched_barrier(0);
            flush_half(1028, false);
        }
    }
}

}  // namespace

bool psnd_stft1024r_ok(long long T, long long F, int hop, int pad) {
    return hop == kHop && pad % 4 == 0 && T % 4 == 0 && F > 0 && T + 2ll * kNFFT < (1ll << 31) && F < (1ll << 31);
}

int psnd_stft1024r_launch(const float *wav, const float *plan, float *mag_nfk, long long N, long long T, long long F, int pad, float mag_eps,
                          int ablate, hipStream_t stream) {
    QRParams p;
    p.wav = wav, p.plan = plan, p.mag = mag_nfk, p.T = (int)T, p.F = (int)F, p.pad = pad, p.mag_eps = mag_eps, p.ablate = ablate;
    const long long nq = (F + 3) / 4;
    if (nq * N >= (1ll << 31)) PSND_FAIL(PSND_E_SHAPE, "stft_mag_nfk(n1024r): too many quads");
    p.nq = (int)nq, p.total_quads = (int)(nq * N);
    long long want = (p.total_quads + kWaves - 2) / (kWaves - 1);
    int grid = want < 256 ? (int)want : 256;                                   // one persistent workgroup per CU
    constexpr size_t lds = sizeof(float) * kLdsFloats;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_fwd_n1024r_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_mag_nfk(n1024r): set LDS size: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(stft_fwd_n1024r_kernel, dim3(grid), dim3(1024), lds, stream, p);
    PSND_CHECK_LAUNCH("stft_mag_nfk(n1024r)");
    return PSND_OK;
}
