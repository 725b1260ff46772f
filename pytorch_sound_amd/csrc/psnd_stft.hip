// psnd_stft.hip - STFT forward for gfx950 (MI355X).
//
// Replaces STFT.transform (pytorch_sound/models/transforms.py:53-69: reflect pad + strided
// conv1d with a dense (2K x n) windowed-DFT basis = 2.1 MFLOP/frame at n=1024) by a real FFT
// (~25 kFLOP/frame) so the stage becomes HBM-bound: 4*hop bytes read + 4*K bytes written per
// frame (3076 B at 1024/256).
//
// Two-pass register FFT, one workgroup (256 threads) per tile of FT consecutive frames of one
// clip.  The n-point real frame is packed as C = n/2 complex points z[m] = x[2m] + i x[2m+1],
// C = R1 * L:
//   pass 1  L lanes per frame; lane l holds z[l + L*a], a < R1 (loaded straight from HBM/L2 as
//           8-byte pieces: 16 lanes x 8 B = one 128-B line per instruction per frame), applies
//           the window, runs a radix-R1 FFT entirely in VGPRs, multiplies by W_C^(l*q) and
//           writes Y[q][l] to LDS (conflict-free layout [frame][q][l], frame stride C+4).
//   pass 2  thread = (frame f, pair qq): reads the L values of butterflies q = qq and R1 - qq
//           with ds_read_b128, runs two radix-L FFTs in VGPRs -> Z[q + R1*p]; the real-FFT
//           split X[k] = S + v_k D, X[C-k] = conj(S - v_k D) pairs (q,p) with (R1-q, L-1-p), i.e.
//           exactly the two butterflies the thread already holds: no further exchange.
//           Lanes of a wave are 16 (or 32) consecutive frames x 4 (2) bins, so every store
//           instruction writes 64-B (128-B) runs of the (N,K,F) frame-fastest output.
// The only LDS round trip is the transposing exchange between the passes (4 KB/frame each way).
#include "psnd_stft_pass.h"
#include "psnd_stft_w.h"
#include "psnd_stft_q.h"
#include "psnd_pk.h"
#include "psnd_stft_emit.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

#ifndef PSND_STORE_AUX
#define PSND_STORE_AUX 0   // cache-policy bits of the output stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#endif

namespace {
using namespace psnd_stft;

struct StftFwdParams {
    const float *wav;
    const float *plan;
    float *mag, *phase, *re, *im;
    long long T, F;
    int hop, pad, ntile, total_tiles;
    float mag_eps;
    int ablate;   // debug only (PSND_ABLATE): bit1 skip global stores
    int nfk = 0;  // psnd_stft_mag_nfk: the output is (N, F, K), bin axis fastest (generic kernel; the tuned NFK kernels live in psnd_stft_w.hip / psnd_stft_q.hip)
    // fused wav -> log-mel (psnd_logmel_fwd): the magnitude tile stays in LDS and is projected there
    const int *mel_plan;
    float *mel_out;
    int mel_M, log_kind;
    float log_offset, pre_clamp_min, clamp_lo, clamp_hi;
    // fused multi_stft_loss partial sums (psnd_stft_fwd_msl): |X| is compared with the target magnitudes in registers
    const float *msl_t;
    double *msl_part;
    float msl_eps;
#ifdef PSND_TRACE
    long long *trace;   // tools/trace_stft.py: 8 s_memtime stamps per wave
    int trace_iter;     // which tile iteration of a persistent workgroup is stamped
#endif
};
#ifdef PSND_TRACE
#define PSND_STAMP(i_)                                                                               \
    do {                                                                                             \
        if ((threadIdx.x & 63) == 0 && p.trace && psnd_it == p.trace_iter)                           \
            p.trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PSND_STAMP(i_)
#endif

// Output writer.  Buffer stores: the resource descriptor (clip base + first frame of the tile) and the
// row offset (soff = bin * F * 4, wave-uniform) live in SGPRs, the per-thread part (voff) is ONE
// 32-bit VGPR - no 64-bit address arithmetic or per-row address registers.
template <bool MAG, bool PHASE, bool REIM>
struct Emit {
    __amdgpu_buffer_rsrc_t rmag, rphase, rre, rim;
    float eps;
    bool valid;
    __device__ __forceinline__ Emit(float *mag, float *phase, float *re, float *im, size_t base, int bytes, float eps_,
                                    bool valid_)
        : eps(eps_), valid(valid_) {
        if constexpr (MAG) rmag = make_uniform_rsrc(mag + base, bytes);
        if constexpr (PHASE) rphase = make_uniform_rsrc(phase + base, bytes);
        if constexpr (REIM) {
            rre = make_uniform_rsrc(re + base, bytes);
            rim = make_uniform_rsrc(im + base, bytes);
        }
    }
    // callers run the whole post-processing under ONE `if (emit.valid)` (frames past F in a last tile)
    __device__ __forceinline__ void operator()(int voff, int soff, float xr, float xi) const {
        if constexpr (MAG) {
            const float m = __builtin_amdgcn_sqrtf(__builtin_fmaf(xr, xr, __builtin_fmaf(xi, xi, eps)));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), rmag, voff, soff, PSND_STORE_AUX);
        }
        if constexpr (PHASE)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, atan2f(xi, xr)), rphase, voff, soff, PSND_STORE_AUX);
        if constexpr (REIM) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xr), rre, voff, soff, PSND_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xi), rim, voff, soff, PSND_STORE_AUX);
        }
    }
};

// real-FFT split of the two butterflies a thread holds + output: Z'[q + R1 p] sits in slot bitrev(p).
// Every pair yields one bin in the lower half of the spectrum and its mirror in the upper half.  The
// lower bins are stored at once, the upper ones are parked and stored afterwards in ASCENDING order, so
// a workgroup writes the (N,K,F) output as one upward sweep over the rows (measured on MI355X: the
// two-converging-sweeps order streams ~25 % slower into HBM).
template <int R1, int L, class EmitT>
__device__ __forceinline__ void post_emit(float (&ar)[L], float (&ai)[L], float (&br)[L], float (&bi)[L], bool special,
                                          int qA, int qB, const float *s_vk, const EmitT &emit, int iF, int col) {
    constexpr int LB = ct::ilog2(L);
    const int stepF = R1 * iF * 4;                 // bytes between bins q and q + R1 (wave-uniform)
    const int offA = qA * iF * 4 + col, offB = qB * iF * 4 + col;   // per-thread byte offsets
    float xkr, xki;
    float h1r[L / 2], h1i[L / 2], h2r[L / 2], h2i[L / 2];
    if (!special) {
        // qA < qB: ascending rows are  qA + R1 p,  qB + R1 p  for p = 0 .. L-1
        static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int pp = decltype(pc)::value;
            constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
            {   // primary A[pp]: k = qA + R1*pp (< C/2), partner B[L-1-pp] -> bin qB + R1 (L-1-pp)
                const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (qA + R1 * pp));
                rfft_pair(ar[sa], ai[sa], br[sb], bi[sb], v.x, v.y, xkr, xki, h1r[pp], h1i[pp]);
                emit(offA, pp * stepF, xkr, xki);
            }
            {   // primary B[pp]: k = qB + R1*pp (< C/2), partner A[L-1-pp] -> bin qA + R1 (L-1-pp)
                const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (qB + R1 * pp));
                rfft_pair(br[sa], bi[sa], ar[sb], ai[sb], v.x, v.y, xkr, xki, h2r[pp], h2i[pp]);
                emit(offB, pp * stepF, xkr, xki);
            }
        });
        static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int pp = L / 2 - 1 - decltype(pc)::value;          // upper half, ascending
            emit(offA, (L - 1 - pp) * stepF, h2r[pp], h2i[pp]);
            emit(offB, (L - 1 - pp) * stepF, h1r[pp], h1i[pp]);
        });
    } else {
        // butterfly q = 0: bins R1*p pair with C - R1*p = R1*(L-p); p = 0 gives X[0] and X[C]
        // butterfly q = R1/2: bins R1/2 + R1*p pair with R1/2 + R1*(L-1-p)
        float hcr, hci;
        static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int pp = decltype(pc)::value;
            {
                constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev((L - pp) % L, LB);
                const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (R1 * pp));
                if constexpr (pp == 0) rfft_pair(ar[sa], ai[sa], ar[sb], ai[sb], v.x, v.y, xkr, xki, hcr, hci);
                else rfft_pair(ar[sa], ai[sa], ar[sb], ai[sb], v.x, v.y, xkr, xki, h1r[pp], h1i[pp]);
                emit(col, pp * stepF, xkr, xki);
            }
            {
                constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
                const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (R1 / 2 + R1 * pp));
                rfft_pair(br[sa], bi[sa], br[sb], bi[sb], v.x, v.y, xkr, xki, h2r[pp], h2i[pp]);
                emit(offB, pp * stepF, xkr, xki);
            }
        });
        {   // middle bin C/2 (self-paired)
            constexpr int sm = ct::bitrev(L / 2, LB);
            const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (R1 * (L / 2)));
            float ur, ui;
            rfft_pair(ar[sm], ai[sm], ar[sm], ai[sm], v.x, v.y, xkr, xki, ur, ui);
            emit(col, (L / 2) * stepF, xkr, xki);
        }
        static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int pp = L / 2 - 1 - decltype(pc)::value;
            emit(offB, (L - 1 - pp) * stepF, h2r[pp], h2i[pp]);                 // bin R1/2 + R1 (L-1-pp)
            if constexpr (pp != 0) emit(col, (L - pp) * stepF, h1r[pp], h1i[pp]);   // bin R1 (L-pp)
        });
        emit(col, L * stepF, hcr, hci);                                         // bin C (Nyquist)
    }
}

template <int R1, int L, bool MAG, bool PHASE, bool REIM>
__global__ __launch_bounds__(256) void stft_fwd_kernel(StftFwdParams p) {
    using G = Cfg<R1, L>;
    constexpr int C = G::C, FT = G::FT, P1R = G::P1R, LB = G::LB;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Smem s = carve<R1, L>(smem);
    const int t = threadIdx.x;
    load_tables<R1, L>(p.plan, smem, t);
    __syncthreads();

    int f2, qq;
    pass2_identity<R1, L>(t, f2, qq);
    const bool special = (qq == 0);
    const int qA = qq;
    const int qB = special ? R1 / 2 : R1 - qq;

    const TileWalk tw = tile_walk(p.total_tiles);
    for (int tile = tw.first; tile < tw.end; tile += tw.step) {
        const int clip = tile / p.ntile;
        const long long f0 = (long long)(tile - clip * p.ntile) * FT;
        const float *x = p.wav + (size_t)clip * p.T;

        // ------------------------------- pass 1 -------------------------------------------
#pragma unroll 1
        for (int r = 0; r < P1R; ++r) {
            const int task = r * 256 + t;
            fwd_pass1<R1, L>(s, x, p.T, p.F, f0, p.hop, p.pad, task / L, task % L);
        }
        __syncthreads();

        // ------------------------------- pass 2 -------------------------------------------
        {
            float ar[L], ai[L], br[L], bi[L];
            fwd_pass2_fft<R1, L>(s, f2, qA, qB, ar, ai, br, bi);

            const long long F = p.F;
            // descriptor base = (clip, bin 0, frame f0): wave-uniform; per-thread part = frame column f2
            const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)f0;
            const int cbytes = (int)(((long long)(C + 1) * F - f0) * 4);
            Emit<MAG, PHASE, REIM> emit(p.mag, p.phase, p.re, p.im, cbase, cbytes, p.mag_eps, (f0 + f2) < F);
            if (emit.valid) post_emit<R1, L>(ar, ai, br, bi, special, qA, qB, s.vk, emit, (int)F, f2 * 4);
        }
        __syncthreads();  // exchange buffer is reused by the next tile
    }
}

// ---------------------------------------------------------------------------------------------
// n_fft = 1024 kernel (C = 32 x 16).  Two findings shaped it:
//  (1) the VECTOR-MEMORY pipeline bounded the first version (rocprof: ~300 L1 accesses per frame - every
//      sample fetched by 4 overlapping frames with 8-byte loads):
//   * the tile's contiguous waveform span ((FT-1)*hop + n samples, 19 KB) is fetched ONCE with
//     16-byte loads and parked in LDS (reflect indexing is resolved in this fill, so there is no
//     per-frame edge path); pass-1 lanes take their taps with ds_read_b64.  The span carries 4 pad
//     floats per 256 so that the 2 frames of a ds_read_b64 half-wave (frames fl and fl+8) hit
//     disjoint banks;
//   * the 32 q-rows of the exchange travel in two halves (rows 0..15, then 16..31): every pass-2
//     thread (pair qq) needs one row of each half (qq and 32-qq); the span buffer aliases the
//     exchange area -> 44 KB LDS per workgroup, 3 workgroups per CU.
//  (2) all complex arithmetic is PACKED (psnd_pk.h: one VGPR pair per complex value, v_pk_add/mul/fma_f32):
//      1426 scalar fp32 instructions per thread became 725 packed ones, VALU busy time per wave fell from
//      6.2 k to 3.6 k cycles (SQ_ACTIVE_INST_VALU).  Measured issue cost on gfx950 (tools/mb/mb_valu.hip, 4
//      waves/SIMD): v_add_f32 2.8, v_fma_f32 4.2, v_pk_* 5.1, v_sqrt_f32 8.2 cycles per wave instruction - a
//      packed op is worth 1.2 ... 1.9 scalar ones, not 2.  The exchange holds (re, im) pairs:
//      ds_write_b64 per value, ds_read_b128 per two values, zero bank conflicts (SQ_LDS_BANK_CONFLICT = 0).
//  (3) what is left (tools/trace_stft.py, s_memtime stamps per phase; PSND_ABLATE A/B runs, 1024 clips x 2 s):
//      everything but the real split + magnitude + stores runs in 78 us, adding that arithmetic WITHOUT its
//      stores 143 us, the full kernel 180 us (3.0 TB/s).  A workgroup lives ~20 k cycles of which ~7 k are the
//      prologue (its loads queue behind the stores of the workgroups finishing on the same CU), and the three
//      waves a SIMD holds are rarely all runnable, so a VALU instruction costs ~8 cycles instead of 4-5.
//      Tried and measured WORSE: persistent workgroups with the next span prefetched (the prefetch loads stall
//      ~9 k cycles behind the previous tile's stores in the CU's in-order vector-memory queue: 200 us), four
//      exchange rounds of 8 rows for 30 KB LDS and 4 workgroups/CU (8 barriers per tile: 200 us), an L2
//      prefetch of the span of a later workgroup (185 us).
// ---------------------------------------------------------------------------------------------
template <bool MAG, bool PHASE, bool REIM>
struct EmitPk {
    __amdgpu_buffer_rsrc_t rmag, rphase, rre, rim;
    v2f eps2;
    bool valid;
    bool nostore = false;   // PSND_ABLATE bit 2 (value 4): all the arithmetic, no store reaches memory
    __device__ __forceinline__ EmitPk(float *mag, float *phase, float *re, float *im, size_t base, int bytes, float eps_,
                                      bool valid_)
        : eps2(v2f{eps_, 0.f}), valid(valid_) {
        if constexpr (MAG) rmag = make_uniform_rsrc(mag + base, bytes);
        if constexpr (PHASE) rphase = make_uniform_rsrc(phase + base, bytes);
        if constexpr (REIM) {
            rre = make_uniform_rsrc(re + base, bytes);
            rim = make_uniform_rsrc(im + base, bytes);
        }
    }
    // x (or its conjugate) -> the values to store
    template <bool CONJ>
    __device__ __forceinline__ OutVal make(v2f x) const {
        OutVal o;
        if constexpr (MAG) {
            const v2f sq = pk::fma(x, x, eps2);
            o.m = __builtin_amdgcn_sqrtf(sq.x + sq.y);
        }
        if constexpr (PHASE) o.ph = atan2f(CONJ ? -x.y : x.y, x.x);
        if constexpr (REIM) {
            o.re = x.x;
            o.im = CONJ ? -x.y : x.y;
        }
        return o;
    }
    __device__ __forceinline__ void store(int voff, int soff, const OutVal &o) const {
#ifdef PSND_ABLATE_BUILD
        // A/B builds only (tools/build_variant.sh ablate -DPSND_ABLATE_BUILD, PSND_ABLATE=4: all the arithmetic, no store reaches
        // memory).  As a RUN-TIME test in the production kernel it put an exec-mask branch in front of every one of the ~34 stores of a
        // thread: the real split became 34 basic blocks, each waiting for its own twiddle read and square root.
        if constexpr (MAG) if (nostore && o.m > -1.f) return;
#endif
        if constexpr (MAG) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o.m), rmag, voff, soff, PSND_STORE_AUX);
        if constexpr (PHASE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o.ph), rphase, voff, soff, PSND_STORE_AUX);
        if constexpr (REIM) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o.re), rre, voff, soff, PSND_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o.im), rim, voff, soff, PSND_STORE_AUX);
        }
    }
};

// magnitude-only writer into an LDS tile [bin][16 frames]: with iF = 16 and col = 4 * frame the byte offsets that
// post_emit_pk computes for the (N,K,F) tensor ARE the offsets into that tile
struct EmitLds {
    float *tile;
    v2f eps2;
    bool valid;
    template <bool CONJ>
    __device__ __forceinline__ OutVal make(v2f x) const {
        OutVal o;
        const v2f sq = pk::fma(x, x, eps2);
        o.m = __builtin_amdgcn_sqrtf(sq.x + sq.y);
        return o;
    }
    __device__ __forceinline__ void store(int voff, int soff, const OutVal &o) const {
        *reinterpret_cast<float *>(reinterpret_cast<char *>(tile) + voff + soff) = o.m;
    }
};


// multi_stft_loss partial sums instead of stores (psnd_stft_fwd_msl): every bin the thread would store is compared with the target
// magnitude at the same place: sum (t - |X|)^2, sum t^2, sum |log(t + eps) - log(|X| + eps)|  (loss_partial_kernel of psnd_loss.hip)
struct EmitLoss {
    __amdgpu_buffer_rsrc_t rt;
    v2f eps2;
    float leps;
    bool valid;
    mutable float s_d2 = 0.f, s_t2 = 0.f, s_l1 = 0.f;
    __device__ __forceinline__ EmitLoss(const float *t_mag, size_t base, int bytes, float mag_eps, float log_eps, bool valid_)
        : rt(make_uniform_rsrc(t_mag + base, bytes)), eps2(v2f{mag_eps, 0.f}), leps(log_eps), valid(valid_) {}
    template <bool CONJ>
    __device__ __forceinline__ OutVal make(v2f x) const {
        OutVal o;
        const v2f sq = pk::fma(x, x, eps2);
        o.m = __builtin_amdgcn_sqrtf(sq.x + sq.y);
        return o;
    }
    __device__ __forceinline__ void store(int voff, int soff, const OutVal &o) const {
        const float tv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, voff, soff, 0));
        const float d = tv - o.m;
        s_d2 = __builtin_fmaf(d, d, s_d2);
        s_t2 = __builtin_fmaf(tv, tv, s_t2);
        // log2 on the hardware unit (v_log_f32; both arguments >= eps, no denormal scaling), the sum is scaled by ln 2 at the end:
        // __logf expands to ~14 instructions per call here and made this variant slower than the one that stores
        s_l1 += __builtin_fabsf(__builtin_amdgcn_logf(tv + leps) - __builtin_amdgcn_logf(o.m + leps));
    }
};
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Geometry of the span-staged kernel for C = R1 x L complex points ((32, 16): n_fft = 1024, (16, 16): 512, (32, 32): 2048): a
// workgroup of 256 threads owns FT = 512 / R1 frames; pass 1 runs in rounds of FPR = 256 / L frames x L lanes, pass 2 is one thread
// per (frame, row pair).
template <int R1, int L_ = 16>
struct SpanGeom {
    static constexpr int L = L_, C = R1 * L, NFFT = 2 * C, FT = 512 / R1, FPR = 256 / L, NR = FT / FPR, HR = R1 / 2, ROW = 2 * R1 + 4;
    static constexpr int TPB = 256 / (2 * L);    // taps of a lane inside one 256-sample block of the span
    static constexpr int VKP = (2 * (C / 2 + 1) + 3) & ~3;
    static constexpr int SFH = HR * L * 2 + 4;   // exchange frame stride (floats): SFH/4 odd -> b128 reads conflict-free,
                                                 // 8*SFH = 32 (mod 64) -> the b64 writes of frames fl, fl+8 disjoint
    static constexpr int TAB = 2 * L * ROW + VKP;
    static constexpr int WT_OFF = FT * SFH - L * ROW;   // window table at the tail of the exchange area (one-tile mode)
    static_assert((SFH / 4) % 2 == 1 && (8 * SFH) % 64 == 32, "exchange pitch");
    // exchange area (floats): half of the q-rows for FT frames, or the tile's waveform span (+ 4 pad floats per 256)
    static int area_floats(int hop) {
        const int span = (FT - 1) * hop + NFFT;
        const int spanp = span + 4 * (span / 256 + 1);
        return spanp > FT * SFH ? spanp : FT * SFH;
    }
};
constexpr int kN1024Sfh = SpanGeom<32>::SFH;
constexpr int kN1024TabFloats = SpanGeom<32>::TAB;
constexpr int kN1024WtOff = SpanGeom<32>::WT_OFF;
inline int n1024_area_floats(int hop) { return SpanGeom<32>::area_floats(hop); }

// FUSE: 0 = outputs to memory, 1 = log-mel projection of the magnitude tile (psnd_logmel_fwd), 2 = multi_stft_loss partial sums
template <bool MAG, bool PHASE, bool REIM, int SPV, bool PERSIST, int FUSE = 0, int R1 = 32, int L_ = 16>
__global__ __launch_bounds__(256, L_ == 32 ? 2 : (PERSIST ? 3 : 4)) void stft_fwd_n1024_kernel(StftFwdParams p) {
    constexpr bool FUSE_MEL = FUSE == 1, FUSE_LOSS = FUSE == 2;
    static_assert(!FUSE_MEL || (MAG && !PHASE && !REIM && !PERSIST && R1 == 32 && L_ == 16), "fused log-mel: magnitude only, one tile per workgroup");
    static_assert(!FUSE_LOSS || (MAG && !PHASE && !REIM && !PERSIST), "fused loss sums: magnitude only, one tile per workgroup");
    using G = SpanGeom<R1, L_>;
    constexpr int L = G::L, C = G::C, NFFT = G::NFFT, FT = G::FT, NR = G::NR, FPR = G::FPR, TPB = G::TPB, ROW = G::ROW, VKP = G::VKP, RB = ct::ilog2(R1);
    constexpr int HR = R1 / 2;          // rows per exchange half
    constexpr int SFH = G::SFH;
    constexpr int TAB = G::TAB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // One tile per workgroup (!PERSIST): the window table is only needed before the first exchange write, so
    // it sits in the tail of the exchange area that the span leaves free: 39.4 KB of LDS, 4 workgroups per CU.
    constexpr int WT_IN_AREA = G::WT_OFF;
    float *s_tw = PERSIST ? smem + L * ROW : smem;
    float *s_vk = s_tw + L * ROW;
    float *s_x = s_vk + VKP;            // exchange [frame][row][l] of (re, im), also the span buffer
    float *s_wt = PERSIST ? smem : s_x + WT_IN_AREA;
    float *s_span = s_x;
    const int t = threadIdx.x;
#ifdef PSND_TRACE
    int psnd_it = p.trace_iter;      // prologue stamps always taken
#endif
    PSND_STAMP(0);

    // pass-1 identity.  L = 16: half-waves hold frames fl, fl+8; L = 32: a half-wave is the 32 lanes of one frame
    const int l = t & (L - 1), fl = L == 16 ? ((t >> 4) & 1) * 8 + (t >> 5) : (t >> 5);
    const int f2 = t & (FT - 1);                                // pass-2 identity: frame, row pair qq < R1 / 2
    // pairs of a wave: 4 apart (stores: one upward sweep per wave) or, for the LDS magnitude tile of the fused
    // log-mel kernel, consecutive (4 consecutive bins x 16 frames = 64 distinct banks per ds_write)
    const int qq = FUSE_MEL ? 4 * (t >> 6) + ((t >> 4) & 3) : (t >> 6) + 4 * ((t & 63) / FT);
    const bool special = (qq == 0);
    const int qA = qq, qB = special ? R1 / 2 : R1 - qq;
    const int rowA = qq, rowB = special ? 0 : HR - qq; // row inside its half
    const int hop = p.hop;
    const int span_len = (FT - 1) * hop + NFFT;        // samples a tile touches
    // span position of sample s: s + skew * (s / 256)
    const int skew = (L == 16 && hop % 256 == 0) ? 4 : 0;

    // The span of tile t+1 is REQUESTED (into SPV registers) before the stores of tile t are issued:
    // gfx9 has one in-order vmcnt for loads and stores, so a wait for loads issued AFTER the stores
    // would also wait for the stores; this way the stores drain under the next tile's pass 1.
    f32x4 spv[SPV];
    auto request_span = [&](int tile_) __attribute__((always_inline)) {
        const int clip_ = tile_ / p.ntile;
        const float *x_ = p.wav + (size_t)clip_ * p.T;
        const long long g0 = (long long)(tile_ - clip_ * p.ntile) * FT * hop - p.pad;   // sample index of span[0]
        const int Ti = (int)p.T;
        static_for<0, SPV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int s4 = (t + 256 * j) * 4;
            if (s4 < span_len) {
                const long long g = g0 + s4;
                if (g >= 0 && g + 3 < p.T) {
                    spv[j] = *reinterpret_cast<const f32x4_u *>(x_ + g);
                } else {
                    const int gi = (int)g;
                    spv[j].x = x_[reflect_idx32(gi, Ti)];
                    spv[j].y = x_[reflect_idx32(gi + 1, Ti)];
                    spv[j].z = x_[reflect_idx32(gi + 2, Ti)];
                    spv[j].w = x_[reflect_idx32(gi + 3, Ti)];
                }
            }
        });
    };

    auto commit_span = [&]() __attribute__((always_inline)) {
        static_for<0, SPV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int s4 = (t + 256 * j) * 4;
            if (s4 < span_len) *reinterpret_cast<f32x4 *>(s_span + s4 + skew * (s4 >> 8)) = spv[j];
        });
    };

    // Schedule per tile (every wait on the span loads happens BEFORE this tile's stores are issued, so
    // the stores - single in-order vmcnt on gfx9 - are never on a wave's critical path):
    //   [span(t) is in LDS] barrier | taps, window, radix-32 | barrier | rows 0..15 -> LDS | barrier | read row A
    //   request span(t+1) -> registers | barrier | rows 16..31 -> LDS, radix-16 A | barrier | read row B
    //   barrier | commit span(t+1) -> LDS (aliases the exchange area) | radix-16 B, real split, STORES
    // strided XCD-contiguous walk: neighbouring tiles run at the SAME time on neighbouring CUs of one XCD.
    // (a contiguous run per workgroup - neighbours written ~10 us apart by one CU - measured 45 % slower:
    // L2 keeps partially written lines only briefly; PSND_ABLATE=8 selects it for A/B runs)
    const TileWalk tw = PSND_ABL(p, 8) ? tile_run(p.total_tiles) : tile_walk(p.total_tiles);
    if (tw.first < tw.end) {
        // every global load of the prologue is in flight before the first wait: span first (HBM), then the
        // tables (L2); the first barrier of the tile loop publishes both
        request_span(tw.first);
        constexpr int TV = (TAB / 4 + 255) / 256;
        constexpr int WT4 = L * ROW / 4;         // 16-byte pieces of the window table (first in the plan)
        const f32x4 *src = reinterpret_cast<const f32x4 *>(p.plan);
        f32x4 *dst_wt = reinterpret_cast<f32x4 *>(s_wt), *dst_tw = reinterpret_cast<f32x4 *>(s_tw);
        f32x4 tv[TV];
        static_for<0, TV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if (t + 256 * j < TAB / 4) tv[j] = src[t + 256 * j];
        });
        static_for<0, TV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int i = t + 256 * j;
            if (i < WT4) dst_wt[i] = tv[j];
            else if (i < TAB / 4) dst_tw[i - WT4] = tv[j];
        });
        commit_span();
        PSND_STAMP(1);
    }
    int tile = tw.first;
    if (tile < tw.end) do {              // one trip when !PERSIST (one tile per workgroup): no loop, nothing hoisted
        const int clip = tile / p.ntile;
        const long long f0 = (long long)(tile - clip * p.ntile) * FT;
        const bool more = PERSIST && (tile + tw.step < tw.end);
        v2f z[NR][R1];
#ifdef PSND_TRACE
        psnd_it = (tile - tw.first) / tw.step;
#endif

        __syncthreads();                 // span(t) visible to every wave
        PSND_STAMP(2);
#ifdef PSND_TRACE
        if (PERSIST && (threadIdx.x & 63) == 0 && p.trace && psnd_it == p.trace_iter + 1)   // slot 7 = next tile may start
            p.trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            // taps of lane l of frame fr = fl + FPR r: samples fr*hop + 2*(l + L a) + {0,1}.  Affine addressing: one base
            // per group of TPB taps (the bank skew steps once per 256 samples), immediates inside the group.
            constexpr int r = decltype(rc)::value;
            const int sb = (fl + FPR * r) * hop + 2 * l;
            const float *tb0 = s_span + sb + skew * (sb >> 8);
            static_for<0, R1 / TPB>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                const float *tb = tb0 + g * (256 + skew);
                static_for<0, TPB>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = TPB * g + decltype(ac)::value;
                    z[r][a] = *reinterpret_cast<const v2f *>(tb + 2 * L * (a - TPB * g));
                });
            });
        });
        if (more && (PSND_ABL(p, 16))) request_span(tile + tw.step);   // A/B: prefetch a whole tile ahead
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            const float *wrow = s_wt + l * ROW;
            static_for<0, R1 / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
                z[r][2 * i] *= pk::lo(w);
                z[r][2 * i + 1] *= pk::hi(w);
                // at most 4 window pieces (16 VGPRs) in flight: left alone the scheduler hoists all 16 reads
                if constexpr (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_sched_barrier(0);
            pk::fft<R1>(z[r]);
            __builtin_amdgcn_sched_barrier(0);
        });
        PSND_STAMP(3);

        const float *trow = s_tw + l * ROW;
        float *oz = s_x + fl * SFH + 2 * l;
        auto write_half = [&](auto hc) __attribute__((always_inline)) {
            constexpr int Q0 = decltype(hc)::value;
            static_for<0, HR / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int q0 = Q0 + 2 * i, q1 = q0 + 1;
                const f32x4 w = *reinterpret_cast<const f32x4 *>(trow + 2 * q0);
                constexpr int s0_ = ct::bitrev(q0, RB), s1_ = ct::bitrev(q1, RB);
                static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = decltype(rc)::value;
                    float *ozr = oz + FPR * r * SFH;
                    if constexpr (q0 == 0) *reinterpret_cast<v2f *>(ozr) = z[r][s0_];
                    else *reinterpret_cast<v2f *>(ozr + (q0 - Q0) * 2 * L) = pk::cmul(z[r][s0_], pk::lo(w));
                    *reinterpret_cast<v2f *>(ozr + (q1 - Q0) * 2 * L) = pk::cmul(z[r][s1_], pk::hi(w));
                });
            });
        };
        auto read_row = [&](int row, v2f (&r)[L]) __attribute__((always_inline)) {
            const float *pr = s_x + f2 * SFH + row * 2 * L;
            static_for<0, L / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(pr + 4 * i);
                r[2 * i] = pk::lo(v);
                r[2 * i + 1] = pk::hi(v);
            });
        };

        v2f za[L], zb[L];
        __syncthreads();                 // every lane holds its taps: the span area becomes the exchange
        write_half(std::integral_constant<int, 0>{});
        __syncthreads();
        read_row(rowA, za);
        if (more && !(PSND_ABL(p, 16))) request_span(tile + tw.step);
        __syncthreads();                 // first-half rows consumed: the buffer may take rows 16..31
        PSND_STAMP(4);
        write_half(std::integral_constant<int, HR>{});
        __builtin_amdgcn_sched_barrier(0);
        pk::fft<L>(za);
        __syncthreads();
        read_row(rowB, zb);
        if constexpr (PERSIST) {
            __syncthreads();             // exchange fully consumed: it becomes the span buffer of tile t+1
            if (more) commit_span();
        }
        __builtin_amdgcn_sched_barrier(0);
        PSND_STAMP(5);
        pk::fft<L>(zb);
        __builtin_amdgcn_sched_barrier(0);

        const long long F = p.F;
        if constexpr (FUSE_MEL) {
            // ---- magnitude tile [513][16] in LDS (the exchange area, 32.8 KB), then the band-sparse mel product on the
            //      fp32 matrix cores straight out of LDS: the (N,K,F) magnitude never exists in HBM (transforms.py:232-243
            //      computes and then discards it) - 1344 B per frame end to end instead of 3076 + 2372.
            float *s_mag = s_x;
            __syncthreads();                 // every wave has taken its second row out of the exchange
            EmitLds emit{s_mag, v2f{p.mag_eps, 0.f}, true};
            post_emit_pk<R1, L>(za, zb, special, qA, qB, s_vk, emit, FT, f2 * 4);
            __syncthreads();
            const int lane = t & 63, wave = t >> 6;
            const int *plan = p.mel_plan;
            const int MT = plan[2], KS = plan[3];
            const float *Wf = reinterpret_cast<const float *>(plan) + plan[6];
            const int fn = lane & 15, kq = lane >> 4;
            const bool fvalid = (f0 + fn) < F;
            const __amdgpu_buffer_rsrc_t rout = make_uniform_rsrc(p.mel_out + ((size_t)clip * p.mel_M) * (size_t)F + (size_t)f0,
                                                                  (int)(((long long)p.mel_M * F - f0) * 4));
            // mel row tiles from the top (longest band) down, round-robin over the 4 waves
            for (int mt = MT - 1 - wave; mt >= 0; mt -= 4) {
                const int lo = plan[8 + 2 * mt], hi = plan[8 + 2 * mt + 1];
                const float *W = Wf + (size_t)mt * KS * 64 + lane;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // 8 weight loads (L2) and 8 magnitude reads (LDS) in flight per batch: the chain of dependent MFMAs
                // would otherwise wait for one global load per step
                for (int s0 = lo; s0 < hi; s0 += 8) {
                    float wv[8], bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int s = s0 + u, k = 4 * s + kq;
                        wv[u] = s < hi ? W[(size_t)s * 64] : 0.f;
                        bv[u] = (s < hi && k <= C) ? s_mag[k * FT + fn] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u], bv[u], acc, 0, 0, 0);
                }
                // D: row = 4 * (lane >> 4) + reg, column = lane & 15 = frame
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mt + 4 * kq + r;
                    if (row < p.mel_M && fvalid) {
                        const float y = fminf(fmaxf(log_apply(acc[r], p.log_kind, p.log_offset, p.pre_clamp_min), p.clamp_lo), p.clamp_hi);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rout, fn * 4, row * (int)F * 4, 0);
                    }
                }
            }
        } else if constexpr (FUSE_LOSS) {
            const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)f0;
            const int cbytes = (int)(((long long)(C + 1) * F - f0) * 4);
            EmitLoss emit(p.msl_t, cbase, cbytes, p.mag_eps, p.msl_eps, (f0 + f2) < F);
            if (emit.valid) post_emit_pk<R1, L>(za, zb, special, qA, qB, s_vk, emit, (int)F, f2 * 4);
            const float w1 = wave_sum_f(emit.s_d2), w2 = wave_sum_f(emit.s_t2), w3 = wave_sum_f(emit.s_l1) * 0.69314718056f;
            __syncthreads();                 // every wave has taken its second row out of the exchange
            if ((t & 63) == 0) s_x[3 * (t >> 6)] = w1, s_x[3 * (t >> 6) + 1] = w2, s_x[3 * (t >> 6) + 2] = w3;
            __syncthreads();
            if (t < 3) p.msl_part[(size_t)tile * 3 + t] = (double)s_x[t] + (double)s_x[3 + t] + (double)s_x[6 + t] + (double)s_x[9 + t];
        } else {
        const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)f0;
        const int cbytes = (int)(((long long)(C + 1) * F - f0) * 4);
        EmitPk<MAG, PHASE, REIM> emit(p.mag, p.phase, p.re, p.im, cbase, cbytes, p.mag_eps, ((f0 + f2) < F) && !(PSND_ABL(p, 2)));
        emit.nostore = PSND_ABL(p, 4);
        if (emit.valid && !((PSND_ABL(p, 32)) && special)) post_emit_pk<R1, L>(za, zb, special, qA, qB, s_vk, emit, (int)F, f2 * 4);
        }
#ifdef PSND_TRACE
        __builtin_amdgcn_sched_barrier(0);
        PSND_STAMP(6);
        if constexpr (!PERSIST) {
            __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): every store acknowledged
            PSND_STAMP(7);
        }
#endif
    } while (PERSIST && (tile += tw.step) < tw.end);
}

// ---------------------------------------------------------------------------------------------
// n_fft = 4096 kernel (config 5: 44.1 kHz music, 30 s clips).  C = 2048 = 8 x 16 x 16 complex points, three passes,
// one 512-thread workgroup per tile of 8 consecutive frames; the frames live in LDS between the passes
// (X[frame][q_a][...], 142 KB: one workgroup of 16 waves per CU), the waveform span of the tile (7*hop + 4096 samples)
// is staged once into the same area.
//   pass A  thread = (column j < 256, half of the frames): taps z[j + 256 a] from the span (ds_read_b64), window and W_2048^(j q_a)
//           straight from the plan (the same for the 4 frames), radix-8 in VGPRs   -> X[f][q_a][j]
//   pass B  thread = (f, q_a, j2): radix-16 over b of X[f][q_a][j2 + 16 b], twiddle W_256^(j2 q_b), IN PLACE
//                                                                                  -> X[f][q_a][16 q_b + j2]
//   pass C  thread = (f, pair qq): rows r = q_a + 8 q_b = qq and 128 - qq (16 contiguous values each, ds_read_b128),
//           two radix-16 FFTs -> Z'[r + 128 q_c]; the real-FFT split pairs (r, q_c) with (128 - r, 15 - q_c): exactly
//           post_emit_pk<128, 16>, the packed split + magnitude + ascending-sweep stores of the n = 1024 kernel.
// Packed fp32 arithmetic throughout (psnd_pk.h).  Stores are 32-B runs (8 frames) per bin - LDS capacity, not the
// algorithm, limits the frames per workgroup (16 frames of 2048 complex points would need 256 KB).
// plan(4096) = [win[4096] | wA[256][16] | twA[256][8](re,im) | twB[16][16](re,im) | vk[1025](re,im), padded]
// ---------------------------------------------------------------------------------------------
// LDS pitches (complex values): q_a blocks 274*8 B = 144 (mod 256), frames 2212*8 B = 32 (mod 256):
//   pass C, ds_read_b128 by 16 lanes = 8 frames x 2 consecutive q_a -> 32 f + 144 q_a (mod 256) = 16 distinct 16-B slots
//   pass B, ds_read/write_b64 by 32 lanes = 16 j2 x frames (f, f+4)  -> the two 128-B runs are 128 B apart (mod 256)
constexpr int k4096FT = 8;
constexpr int k4096QP = 256 + 18;
constexpr int k4096FP = 8 * k4096QP + 20;
constexpr int k4096VK = 2052;                      // floats of the vk table (1025 entries, padded)
constexpr int k4096LdsFloats = k4096FT * k4096FP * 2 + k4096VK + 512;
constexpr int k4096PlanFloats = kW4096PlanFloats;     // the tables of the three 4096 kernels (psnd_stft_w.h)
static_assert(kW4096TwOff == 4096 * 3 + 512 + k4096VK, "plan layout");

template <bool MAG, bool PHASE, bool REIM>
__global__ __launch_bounds__(512, 1) void stft_fwd_n4096_kernel(StftFwdParams p) {
    constexpr int C = 2048, NFFT = 4096, FT = k4096FT, QP = k4096QP, FP = k4096FP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_x = smem;                               // X[f][q_a][QP] of (re, im); first the waveform span
    float *s_vk = s_x + FT * FP * 2;
    float *s_twb = s_vk + k4096VK;
    const int t = threadIdx.x;
    const float *plan = p.plan;
    const float *g_wa = plan + NFFT, *g_twa = plan + 2 * NFFT, *g_twb = plan + 3 * NFFT, *g_vk = plan + 3 * NFFT + 512;

    // One workgroup per CU (142 KB of LDS) walks its tiles (XCD-contiguous strided walk, as tile_walk).  With a single
    // resident workgroup nothing else hides the HBM latency of the next span, so it is prefetched into registers
    // (SPV 16-byte pieces per thread) while pass B / C of the current tile run, and committed to LDS once the last row
    // has been taken out of X - every wait on those loads precedes this tile's stores (one in-order vmcnt).
    constexpr int SPV = 8;                               // 512 threads x 8 x 16 B = 64 KB >= span of hop <= 1792
    const TileWalk tw = tile_walk(p.total_tiles);
    if (tw.first >= tw.end) return;
    const int hop = p.hop;
    const int span_len = (FT - 1) * hop + NFFT;
    f32x4 spv[SPV];
    auto request_span = [&](int tile_) __attribute__((always_inline)) {
        const int clip_ = tile_ / p.ntile;
        const float *x_ = p.wav + (size_t)clip_ * p.T;
        const long long g0 = (long long)(tile_ - clip_ * p.ntile) * FT * hop - p.pad;
        const int Ti = (int)p.T;
        static_for<0, SPV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int s4 = (t + 512 * j) * 4;
            if (s4 < span_len) {
                const long long g = g0 + s4;
                if (g >= 0 && g + 3 < p.T) {
                    spv[j] = *reinterpret_cast<const f32x4_u *>(x_ + g);
                } else {
                    const int gi = (int)g;
                    spv[j].x = x_[reflect_idx32(gi, Ti)], spv[j].y = x_[reflect_idx32(gi + 1, Ti)];
                    spv[j].z = x_[reflect_idx32(gi + 2, Ti)], spv[j].w = x_[reflect_idx32(gi + 3, Ti)];
                }
            }
        });
    };
    auto commit_span = [&]() __attribute__((always_inline)) {
        static_for<0, SPV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int s4 = (t + 512 * j) * 4;
            if (s4 < span_len) *reinterpret_cast<f32x4 *>(s_x + s4) = spv[j];
        });
    };
    request_span(tw.first);
    for (int i = t; i < (k4096VK + 512) / 4; i += 512) {
        const f32x4 v = i < k4096VK / 4 ? reinterpret_cast<const f32x4 *>(g_vk)[i]
                                        : reinterpret_cast<const f32x4 *>(g_twb)[i - k4096VK / 4];
        reinterpret_cast<f32x4 *>(s_vk)[i] = v;          // s_twb follows s_vk
    }
    // window and pass-A twiddles of column j: the same for all frames
    const int ja = t & 255, fa0 = 4 * (t >> 8);      // pass-A identity: column, first of its 4 frames
    f32x4 wv[4], tv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wv[i] = reinterpret_cast<const f32x4 *>(g_wa + 16 * ja)[i];
        tv[i] = reinterpret_cast<const f32x4 *>(g_twa + 16 * ja)[i];
    }
    commit_span();

    for (int tile = tw.first; tile < tw.end; tile += tw.step) {
    const int clip = tile / p.ntile;
    const long long f0 = (long long)(tile - clip * p.ntile) * FT;
    const bool more = tile + tw.step < tw.end;
    __syncthreads();                                 // span(tile) (and, the first time, the tables) visible

    // ---- pass A ----------------------------------------------------------------------------------------------
    v2f z[4][8];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const v2f w = (a & 1) ? pk::hi(wv[a >> 1]) : pk::lo(wv[a >> 1]);
            z[f][a] = *reinterpret_cast<const v2f *>(s_x + (fa0 + f) * hop + 2 * (ja + 256 * a)) * w;
        }
    __syncthreads();                                 // span consumed: the area becomes X
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        pk::fft<8>(z[f]);
        v2f *o = reinterpret_cast<v2f *>(s_x) + (fa0 + f) * FP + ja;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const v2f w = (q & 1) ? pk::hi(tv[q >> 1]) : pk::lo(tv[q >> 1]);
            o[q * QP] = q == 0 ? z[f][0] : pk::cmul(z[f][ct::bitrev(q, 3)], w);
        }
    }
    if (more) request_span(tile + tw.step);          // in flight during pass B and the row reads of pass C
    __syncthreads();

    // ---- pass B (in place) -------------------------------------------------------------------------------------
    {
        const int j2 = t & 15;
        // W_256^(j2 q) is symmetric in (j2, q): reading the table as [q][j2] puts the 16 j2 lanes on consecutive 8-B
        // words (as [j2][q] the rows are 128 B apart: an 8-way bank conflict on every twiddle read)
        const v2f *twr = reinterpret_cast<const v2f *>(s_twb) + j2;
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int f = 4 * ((t >> 4) & 1) + 2 * (t >> 8) + h, qa = (t >> 5) & 7;   // a half-wave: frames f, f + 4 of one q_a
            v2f *base = reinterpret_cast<v2f *>(s_x) + f * FP + qa * QP + j2;
            v2f y[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) y[b] = base[16 * b];
            pk::fft<16>(y);
#pragma unroll
            for (int q = 0; q < 16; ++q) base[16 * q] = q == 0 ? y[0] : pk::cmul(y[ct::bitrev(q, 4)], twr[16 * q]);
        }
    }
    __syncthreads();

    // ---- pass C + real split + output ---------------------------------------------------------------------------
    const int f = t & 7, qq = t >> 3;
    const bool special = (qq == 0);
    const int rA = qq, rB = special ? 64 : 128 - qq;
    v2f za[16], zb[16];
    auto read_row = [&](int r, v2f (&dst)[16]) __attribute__((always_inline)) {
        const float *pr = s_x + 2 * (f * FP + (r & 7) * QP + (r >> 3) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(pr + 4 * i);
            dst[2 * i] = pk::lo(v);
            dst[2 * i + 1] = pk::hi(v);
        }
    };
    read_row(rA, za);
    read_row(rB, zb);
    if (more) {
        __syncthreads();                             // X fully consumed: it takes the next span
        commit_span();
    }
    pk::fft<16>(za);
    pk::fft<16>(zb);
    const long long F = p.F;
    const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)f0;
    const int cbytes = (int)(((long long)(C + 1) * F - f0) * 4);
    EmitPk<MAG, PHASE, REIM> emit(p.mag, p.phase, p.re, p.im, cbase, cbytes, p.mag_eps, ((f0 + f) < F) && !(PSND_ABL(p, 2)));
    emit.nostore = PSND_ABL(p, 4);
    if (emit.valid) post_emit_pk<128, 16>(za, zb, special, rA, rB, s_vk, emit, (int)F, f * 4);
    }   // tile loop
}

// ---------------------------------------------------------------------------------------------
// n_fft = 4096, second generation: TWO independent 512-thread workgroups per CU, 4 frames per tile.
// The kernel above holds 8 frames in 142 KB of LDS: one workgroup = 2 waves per SIMD that meet at 5 barriers per tile, so the
// VALU (1050 packed instructions per thread and tile, ~5 cycles each: the roof of this size) idles whenever the whole
// workgroup sits in an LDS or memory phase - 38 % VALU utilisation, 26 % of the HBM roof.  Here a tile is 4 frames
// (X = 66 KB + 10 KB of tables -> 2 workgroups = 16 waves per CU, 4 per SIMD, phases of the two workgroups overlap):
//   pass A  thread = (column j < 256, 2 of the 4 frames)            radix-8, as above
//   pass B  thread = (f, q_a, j2): ONE radix-16 in place             (above: two per thread)
//   pass C  thread = (f, row r < 128): ONE row (above: the pair r, 128 - r).  The real-FFT split needs Z[C - k] of the partner
//           row 128 - r: of the 16 bin pairs (r + 128 q, 128 - r + 128 (15 - q)) a thread evaluates q = 0..7 - its own lower
//           half against the partner's UPPER half, which the partner parks in place in its row (4 ds_write_b128, one barrier);
//           the partner does the complementary 8.  No arithmetic is duplicated.  Row 64 is its own partner (falls out of the
//           indexing); row 0 pairs q with 16 - q inside itself and owns the two self-paired bins 0 | C and C/2.
// LDS pitches: q_a blocks 258 complex, frames 8 * 258 + 8: pass B's two 128-B runs per 32 lanes are 128 B apart (mod 256),
// pass C's ds_read_b128 groups (own row and partner half) hit 16 distinct 16-B slots (tools: bank model in DESIGN.md 4.1).
// Stores are 16-B runs (4 frames) per bin row; neighbouring tiles run on neighbouring workgroups of one XCD at the same time and
// their runs merge in that L2.  Measured (32 clips x 30 s, 508 MB): 224 us with the stores, 141 us without (the 8-frame kernel
// above: 226 us / ~215 us) - the FFT phases now overlap, the write path is what is left.  Tried on top of this kernel and
// measured WORSE (profiles/r02_stft4096_*): holding the magnitudes of two consecutive tiles in registers and storing 32-B runs
// (252 us: 25 % of the write requests leave the L2 as 32-B partials, 1.41x the algorithmic write bytes), and walking four
// consecutive tiles per workgroup so that one CU writes the halves of a 64-B segment a few microseconds apart (299 us, 1.74x):
// partial lines survive in the L2 only when their neighbours arrive at the same time from other workgroups, not later.
// ---------------------------------------------------------------------------------------------
constexpr int k4096bFT = 4;
constexpr int k4096bQP = 258;
constexpr int k4096bFP = 8 * k4096bQP + 8;
constexpr int k4096bLdsFloats = k4096bFT * k4096bFP * 2 + k4096VK + 512;

template <bool MAG, bool PHASE, bool REIM>
__global__ __launch_bounds__(512, 4) void stft_fwd_n4096b_kernel(StftFwdParams p) {
    constexpr int C = 2048, NFFT = 4096, FT = k4096bFT, QP = k4096bQP, FP = k4096bFP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_x = smem;                               // X[f][q_a][QP] of (re, im); first the waveform span
    float *s_vk = s_x + FT * FP * 2;
    float *s_twb = s_vk + k4096VK;
    const int t = threadIdx.x;
    const float *plan = p.plan;
    const float *g_wa = plan + NFFT, *g_twa = plan + 2 * NFFT, *g_twb = plan + 3 * NFFT, *g_vk = plan + 3 * NFFT + 512;

    constexpr int SPV = 4;                               // 512 threads x 4 x 16 B = 32 KB >= span of hop <= 1364
    const TileWalk tw = tile_walk(p.total_tiles);
    if (tw.first >= tw.end) return;
    const int hop = p.hop;
    const int span_len = (FT - 1) * hop + NFFT;
    f32x4 spv[SPV];
    auto request_span = [&](int tile_) __attribute__((always_inline)) {
        const int clip_ = tile_ / p.ntile;
        const float *x_ = p.wav + (size_t)clip_ * p.T;
        const long long g0 = (long long)(tile_ - clip_ * p.ntile) * FT * hop - p.pad;
        const int Ti = (int)p.T;
        static_for<0, SPV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int s4 = (t + 512 * j) * 4;
            if (s4 < span_len) {
                const long long g = g0 + s4;
                if (g >= 0 && g + 3 < p.T) {
                    spv[j] = *reinterpret_cast<const f32x4_u *>(x_ + g);
                } else {
                    const int gi = (int)g;
                    spv[j].x = x_[reflect_idx32(gi, Ti)], spv[j].y = x_[reflect_idx32(gi + 1, Ti)];
                    spv[j].z = x_[reflect_idx32(gi + 2, Ti)], spv[j].w = x_[reflect_idx32(gi + 3, Ti)];
                }
            }
        });
    };
    auto commit_span = [&]() __attribute__((always_inline)) {
        static_for<0, SPV>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int s4 = (t + 512 * j) * 4;
            if (s4 < span_len) *reinterpret_cast<f32x4 *>(s_x + s4) = spv[j];
        });
    };
    request_span(tw.first);
    for (int i = t; i < (k4096VK + 512) / 4; i += 512) {
        const f32x4 v = i < k4096VK / 4 ? reinterpret_cast<const f32x4 *>(g_vk)[i]
                                        : reinterpret_cast<const f32x4 *>(g_twb)[i - k4096VK / 4];
        reinterpret_cast<f32x4 *>(s_vk)[i] = v;          // s_twb follows s_vk
    }
    const int ja = t & 255, fa0 = 2 * (t >> 8);      // pass-A identity: column, first of its 2 frames
    // window and pass-A twiddles of column ja are re-read from the plan (L2) every tile: held in registers across the tile
    // loop they cost 32 VGPRs of the 128 that 4 waves per SIMD leave
    // pass-C identity: frame, row; the partner row's upper half
    const int fc = t & 3, r = t >> 2, rp = (128 - r) & 127;
    const bool special = (r == 0);
    float *row_own = s_x + 2 * (fc * FP + (r & 7) * QP + (r >> 3) * 16);
    const float *row_par = s_x + 2 * (fc * FP + (rp & 7) * QP + (rp >> 3) * 16) + 16;

    for (int tile = tw.first; tile < tw.end; tile += tw.step) {
    const int clip = tile / p.ntile;
    const long long f0 = (long long)(tile - clip * p.ntile) * FT;
    const bool more = tile + tw.step < tw.end;
    f32x4 wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = reinterpret_cast<const f32x4 *>(g_wa + 16 * ja)[i];
    if (tile != tw.first) __syncthreads();           // every partner half of the previous tile has been read: X is free
    commit_span();
    __syncthreads();                                 // span(tile) (and, the first time, the tables) visible

    // ---- pass A ----------------------------------------------------------------------------------------------
    v2f z[2][8];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const v2f w = (a & 1) ? pk::hi(wv[a >> 1]) : pk::lo(wv[a >> 1]);
            z[f][a] = *reinterpret_cast<const v2f *>(s_x + (fa0 + f) * hop + 2 * (ja + 256 * a)) * w;
        }
    f32x4 tv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tv[i] = reinterpret_cast<const f32x4 *>(g_twa + 16 * ja)[i];
    __syncthreads();                                 // span consumed: the area becomes X
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        pk::fft<8>(z[f]);
        v2f *o = reinterpret_cast<v2f *>(s_x) + (fa0 + f) * FP + ja;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const v2f w = (q & 1) ? pk::hi(tv[q >> 1]) : pk::lo(tv[q >> 1]);
            o[q * QP] = q == 0 ? z[f][0] : pk::cmul(z[f][ct::bitrev(q, 3)], w);
        }
    }
    __syncthreads();

    // ---- pass B (in place) -------------------------------------------------------------------------------------
    {
        const int j2 = t & 15;
        const v2f *twr = reinterpret_cast<const v2f *>(s_twb) + j2;      // W_256^(j2 q) read as [q][j2] (symmetric table)
        const int f = 2 * ((t >> 4) & 1) + (t >> 8), qa = (t >> 5) & 7;     // a half-wave: frames f, f + 2 of one q_a
        v2f *base = reinterpret_cast<v2f *>(s_x) + f * FP + qa * QP + j2;
        v2f y[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) y[b] = base[16 * b];
        pk::fft<16>(y);
#pragma unroll
        for (int q = 0; q < 16; ++q) base[16 * q] = q == 0 ? y[0] : pk::cmul(y[ct::bitrev(q, 4)], twr[16 * q]);
    }
    __syncthreads();

    // ---- pass C: one row per thread, upper half parked for the partner -------------------------------------------
    v2f zr[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row_own + 4 * i);
        zr[2 * i] = pk::lo(v);
        zr[2 * i + 1] = pk::hi(v);
    }
    pk::fft<16>(zr);                                 // Z[r + 128 q] in slot bitrev(q)
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // Z[8 + 2i], Z[9 + 2i] -> row positions 8 + 2i, 9 + 2i (natural order)
        const v2f a = zr[ct::bitrev(8 + 2 * i, 4)], b = zr[ct::bitrev(9 + 2 * i, 4)];
        *reinterpret_cast<f32x4 *>(row_own + 16 + 4 * i) = f32x4{a.x, a.y, b.x, b.y};
    }
    __syncthreads();
    v2f pz[8];                                       // pz[q] = Z_partner[15 - q]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row_par + 4 * i);     // partner Z[8 + 2i], Z[9 + 2i]
        pz[7 - 2 * i] = pk::lo(v);
        pz[6 - 2 * i] = pk::hi(v);
    }
    if (__builtin_amdgcn_ballot_w64(special) != 0) {                 // the wave holding row 0: pairs q with 16 - q inside the row
        static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            const v2f own = zr[ct::bitrev(q == 0 ? 0 : 16 - q, 4)];
            pz[q] = special ? own : pz[q];
        });
    }
    if (more) request_span(tile + tw.step);          // in flight during the split and the stores, committed at the next tile's top
    const long long F = p.F;
    const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)f0;
    const int cbytes = (int)(((long long)(C + 1) * F - f0) * 4);
    EmitPk<MAG, PHASE, REIM> emit(p.mag, p.phase, p.re, p.im, cbase, cbytes, p.mag_eps, ((f0 + fc) < F) && !(PSND_ABL(p, 2)));
    emit.nostore = PSND_ABL(p, 4);
    if (emit.valid) {
        const int iF = (int)F;
        const int off_lo = r * iF * 4 + fc * 4, off_hi = (128 - r) * iF * 4 + fc * 4;
        const int step = 128 * iF * 4;
        OutVal hold[8];
        v2f xk, xc;
        static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            rfft_pair_pk(zr[ct::bitrev(q, 4)], pz[q], *reinterpret_cast<const v2f *>(s_vk + 2 * (r + 128 * q)), xk, xc);
            emit.store(off_lo, q * step, emit.template make<false>(xk));      // bin r + 128 q
            hold[q] = emit.template make<true>(xc);                           // bin 128 - r + 128 (15 - q), stored below in ascending order
        });
        if (special) {                                                        // the middle bin C/2 = 128 * 8 of row 0 (self-paired)
            const v2f mid = zr[ct::bitrev(8, 4)];
            rfft_pair_pk(mid, mid, *reinterpret_cast<const v2f *>(s_vk + 2 * 1024), xk, xc);
            emit.store(fc * 4, 8 * step, emit.template make<false>(xk));
        }
        static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = 7 - decltype(qc)::value;
            emit.store(off_hi, (15 - q) * step, hold[q]);
        });
    }
    }   // tile loop
}

int launch_n4096(const StftFwdParams &p, bool mag, bool phase, bool reim, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * k4096LdsFloats;
    int grid = p.total_tiles < 256 ? p.total_tiles : 256;      // one persistent workgroup per CU
    if (const char *e = PSND_ENV("PSND_STFT4096_GRID")) grid = psnd_env_int(e, grid, 1, 65535);
    grid = (grid + 7) & ~7;
#define PSND_LAUNCH(M_, P_, R_)                                                                                     \
    do {                                                                                                            \
        auto kern = stft_fwd_n4096_kernel<M_, P_, R_>;                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_fwd(n4096): set LDS size: %s", hipGetErrorString(e));       \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);                                            \
    } while (0)
    if (mag && !phase && !reim) PSND_LAUNCH(true, false, false);
    else if (mag && phase && !reim) PSND_LAUNCH(true, true, false);
    else if (!mag && !phase && reim) PSND_LAUNCH(false, false, true);
    else if (mag && !phase && reim) PSND_LAUNCH(true, false, true);
    else PSND_LAUNCH(true, true, true);
#undef PSND_LAUNCH
    PSND_CHECK_LAUNCH("stft_fwd(n4096)");
    return PSND_OK;
}

int launch_n4096b(const StftFwdParams &p, bool mag, bool phase, bool reim, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * k4096bLdsFloats;
    int grid = p.total_tiles < 512 ? p.total_tiles : 512;      // two persistent workgroups per CU
    if (const char *e = PSND_ENV("PSND_STFT4096_GRID")) grid = psnd_env_int(e, grid, 1, 65535);
    grid = (grid + 7) & ~7;
#define PSND_LAUNCH(M_, P_, R_)                                                                                     \
    do {                                                                                                            \
        auto kern = stft_fwd_n4096b_kernel<M_, P_, R_>;                                                             \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_fwd(n4096b): set LDS size: %s", hipGetErrorString(e));      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);                                            \
    } while (0)
    if (mag && !phase && !reim) PSND_LAUNCH(true, false, false);
    else if (mag && phase && !reim) PSND_LAUNCH(true, true, false);
    else if (!mag && !phase && reim) PSND_LAUNCH(false, false, true);
    else if (mag && !phase && reim) PSND_LAUNCH(true, false, true);
    else PSND_LAUNCH(true, true, true);
#undef PSND_LAUNCH
    PSND_CHECK_LAUNCH("stft_fwd(n4096b)");
    return PSND_OK;
}

// ---------------------------------------------------------------------------------------------
// generic fallback: any power-of-two n_fft in [16, 8192] without a tuned decomposition.
// One workgroup per (clip, frame); radix-2 FFT of the packed C-point signal in LDS.
// Correct, not fast (strided 4-byte stores) - tuned sizes never come here.
// plan = [win[n]] only.
// ---------------------------------------------------------------------------------------------
template <bool MAG, bool PHASE, bool REIM>
__global__ __launch_bounds__(256) void stft_fwd_generic_kernel(StftFwdParams p, int n_fft) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = n_fft / 2;
    float *sr = smem, *si = smem + C;
    const int t = threadIdx.x;
    const long long f = blockIdx.x;
    const int clip = blockIdx.y;
    const float *x = p.wav + (size_t)clip * p.T;
    const float *win = p.plan;
    const long long s0 = f * p.hop - p.pad;
    int bits = 0;
    while ((1 << bits) < C) ++bits;
    for (int m = t; m < C; m += 256) {
        const float a = x[reflect_idx(s0 + 2 * m, p.T)] * win[2 * m];
        const float b = x[reflect_idx(s0 + 2 * m + 1, p.T)] * win[2 * m + 1];
        const int r = __brev((unsigned)m) >> (32 - bits);
        sr[r] = a;
        si[r] = b;
    }
    __syncthreads();
    for (int h = 1; h < C; h <<= 1) {   // DIT, natural-order output
        for (int j = t; j < C / 2; j += 256) {
            const int blk = (j / h) * 2 * h, jj = j % h;
            const int i0 = blk + jj, i1 = i0 + h;
            float sn, cs;
            sincospif(-(float)jj / (float)h, &sn, &cs);
            const float br = sr[i1] * cs - si[i1] * sn, bi = sr[i1] * sn + si[i1] * cs;
            const float ar = sr[i0], ai = si[i0];
            sr[i0] = ar + br, si[i0] = ai + bi;
            sr[i1] = ar - br, si[i1] = ai - bi;
        }
        __syncthreads();
    }
    // (N, K, F): element (k, f) of the clip at k F + f; (N, F, K): at f K + k
    const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)p.F + (p.nfk ? (size_t)f * (size_t)(C + 1) : (size_t)f);
    const long long kstep = p.nfk ? 1 : p.F;
    Emit<MAG, PHASE, REIM> emit(p.mag, p.phase, p.re, p.im, cbase, p.nfk ? (C + 1) * 4 : (int)(((long long)(C + 1) * p.F - f) * 4), p.mag_eps, true);
    for (int k = t; k <= C / 2; k += 256) {
        const int kc = (C - k) % C;
        float sn, cs;
        sincospif(-(float)k / (float)C, &sn, &cs);   // W_n^k = cs + i sn
        // v_k = -i W_n^k = sn - i cs ; inputs here are unscaled -> halve
        float xkr, xki, xcr, xci;
        rfft_pair(0.5f * sr[k], 0.5f * si[k], 0.5f * sr[kc], 0.5f * si[kc], sn, -cs, xkr, xki, xcr, xci);
        emit((int)((long long)k * kstep * 4), 0, xkr, xki);
        if (k != C - k) emit((int)((long long)(C - k) * kstep * 4), 0, xcr, xci);
    }
}

template <int R1, int L>
int launch_tuned(const StftFwdParams &p, bool mag, bool phase, bool reim, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * Cfg<R1, L>::LDS_FLOATS;
    int grid = p.total_tiles;
    const int cap = 256 * 8;  // ~8 tiles in flight per CU slot, grid-stride beyond
    if (grid > cap) grid = cap;
    grid = (grid + 7) & ~7;
#define PSND_LAUNCH(M_, P_, R_)                                                                     \
    do {                                                                                            \
        auto kern = stft_fwd_kernel<R1, L, M_, P_, R_>;                                             \
        if (lds > 64 * 1024) {                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_fwd: set LDS size: %s", hipGetErrorString(e)); \
        }                                                                                           \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);                            \
    } while (0)
    if (mag && !phase && !reim) PSND_LAUNCH(true, false, false);
    else if (mag && phase && !reim) PSND_LAUNCH(true, true, false);
    else if (!mag && !phase && reim) PSND_LAUNCH(false, false, true);
    else if (mag && !phase && reim) PSND_LAUNCH(true, false, true);
    else PSND_LAUNCH(true, true, true);
#undef PSND_LAUNCH
    PSND_CHECK_LAUNCH("stft_fwd");
    return PSND_OK;
}

// span-staged kernel: (R1, L) = (32, 16) n_fft 1024, (16, 16) n_fft 512, (32, 32) n_fft 2048
template <int R1, int L = 16>
bool span_kernel_ok(int hop) {
    using G = SpanGeom<R1, L>;
    const long long span = (long long)(G::FT - 1) * hop + G::NFFT;
    // taps are 8-byte LDS reads at frame * hop + 2 l; the tile's span (+ bank padding) must fit the span pieces of a thread and the
    // exchange area (L = 32: one tile per workgroup, the window table rides in the tail of the area)
    if (L == 32) return hop % 2 == 0 && span <= 10 * 1024 && span + 4 * (span / 256 + 1) <= G::WT_OFF;
    return hop % 2 == 0 && span <= 8 * 1024 && sizeof(float) * (size_t)(G::TAB + G::area_floats(hop)) <= 64 * 1024;
}

template <class K>
int span_launch_one(K kern, int grid, size_t lds, hipStream_t stream, const StftFwdParams &p) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_fwd(span): set LDS size: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    return PSND_OK;
}

template <int R1, int L = 16>
int launch_span(const StftFwdParams &p, bool mag, bool phase, bool reim, hipStream_t stream) {
    using G = SpanGeom<R1, L>;
    size_t lds = sizeof(float) * (size_t)(G::TAB + G::area_floats(p.hop));
    int grid = p.total_tiles;
    int cap = 1 << 20;       // one tile per workgroup up to 1M tiles: in-order dispatch keeps neighbouring tiles
                             // concurrent (204 us vs 221 us persistent at 1024 x 2 s clips)
    if (const char *e = PSND_ENV("PSND_STFT_GRIDCAP")) cap = psnd_env_int(e, cap, 8, 1 << 20);
    if (L == 32) cap = 1 << 30;
    if (grid > cap) grid = cap;
    grid = (grid + 7) & ~7;
    // one tile per workgroup needs 5 span pieces per thread and span + window table inside the exchange area (n = 1024: hop <= 256)
    const int span = (G::FT - 1) * p.hop + G::NFFT;
    const bool persist = L == 16 && (grid < p.total_tiles || span > 5 * 1024 || span + 4 * (span / 256 + 1) > G::WT_OFF);
    if (!persist) lds -= sizeof(float) * G::L * G::ROW;
    int rc = PSND_OK;
#define PSND_LAUNCH(M_, P_, R_)                                                                                          \
    do {                                                                                                                 \
        if constexpr (L == 32) {                                                                                         \
            if (span <= 6 * 1024) rc = span_launch_one(stft_fwd_n1024_kernel<M_, P_, R_, 6, false, false, R1, L>, grid, lds, stream, p);   \
            else rc = span_launch_one(stft_fwd_n1024_kernel<M_, P_, R_, 10, false, false, R1, L>, grid, lds, stream, p); \
        } else if (span <= 5 * 1024) {                                                                                   \
            if (persist) rc = span_launch_one(stft_fwd_n1024_kernel<M_, P_, R_, 5, true, false, R1, L>, grid, lds, stream, p);  \
            else rc = span_launch_one(stft_fwd_n1024_kernel<M_, P_, R_, 5, false, false, R1, L>, grid, lds, stream, p);  \
        } else {                                                                                                         \
            rc = span_launch_one(stft_fwd_n1024_kernel<M_, P_, R_, 8, true, false, R1, L>, grid, lds, stream, p);        \
        }                                                                                                                \
    } while (0)
    if (mag && !phase && !reim) PSND_LAUNCH(true, false, false);
    else if (mag && phase && !reim) PSND_LAUNCH(true, true, false);
    else if (!mag && !phase && reim) PSND_LAUNCH(false, false, true);
    else if (mag && !phase && reim) PSND_LAUNCH(true, false, true);
    else PSND_LAUNCH(true, true, true);
#undef PSND_LAUNCH
    if (rc != PSND_OK) return rc;
    PSND_CHECK_LAUNCH("stft_fwd(span)");
    return PSND_OK;
}

}  // namespace

using namespace psnd_stft;

extern "C" size_t psnd_stft_plan_bytes(int n_fft) {
    if (const Decomp *d = find_decomp(n_fft)) return sizeof(float) * (size_t)plan_layout(n_fft, d->R1, d->L).total;
    if (n_fft == 4096) return sizeof(float) * (size_t)k4096PlanFloats;
    if (generic_ok(n_fft)) return sizeof(float) * (size_t)n_fft;
    return 0;
}

extern "C" int psnd_stft_plan_build(int n_fft, const float *window_host, void *plan_host) {
    if (!window_host || !plan_host) PSND_FAIL(PSND_E_ARG, "stft_plan_build: null pointer");
    if (psnd_stft_plan_bytes(n_fft) == 0) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_plan_build: n_fft=%d unsupported (power of two in [16,8192])", n_fft);
    float *pl = static_cast<float *>(plan_host);
    const Decomp *d = find_decomp(n_fft);
    if (!d) {
        memcpy(pl, window_host, sizeof(float) * n_fft);      // every kernel without a tuned decomposition reads this
        if (n_fft == 4096) {                                  // tables of stft_fwd_n4096_kernel behind the raw window
            const double two_pi = 6.283185307179586476925286766559;
            float *wa = pl + 4096, *twa = pl + 2 * 4096, *twb = pl + 3 * 4096, *vk = pl + 3 * 4096 + 512;
            memset(wa, 0, sizeof(float) * (k4096PlanFloats - 4096));
            for (int j = 0; j < 256; ++j)
                for (int a = 0; a < 8; ++a) {
                    wa[16 * j + 2 * a] = 0.5f * window_host[2 * (j + 256 * a)];
                    wa[16 * j + 2 * a + 1] = 0.5f * window_host[2 * (j + 256 * a) + 1];
                    const double th = two_pi * (double)(j * a) / 2048.0;          // a doubles as q_a here
                    twa[16 * j + 2 * a] = (float)cos(th);
                    twa[16 * j + 2 * a + 1] = (float)(-sin(th));
                }
            for (int j2 = 0; j2 < 16; ++j2)
                for (int q = 0; q < 16; ++q) {
                    const double th = two_pi * (double)(j2 * q) / 256.0;
                    twb[2 * (16 * j2 + q)] = (float)cos(th);
                    twb[2 * (16 * j2 + q) + 1] = (float)(-sin(th));
                }
            for (int k = 0; k <= 1024; ++k) {
                const double th = two_pi * (double)k / 4096.0;
                vk[2 * k] = (float)(-sin(th));
                vk[2 * k + 1] = (float)(-cos(th));
            }
            psnd_stft4096w_plan_fill(pl);
        }
        return PSND_OK;
    }
    const int R1 = d->R1, L = d->L, C = n_fft / 2;
    const PlanLayout lay = plan_layout(n_fft, R1, L);
    memset(pl, 0, sizeof(float) * lay.total);
    const double two_pi = 6.283185307179586476925286766559;
    for (int l = 0; l < L; ++l) {
        for (int a = 0; a < R1; ++a) {
            const int m = l + L * a;
            pl[l * lay.row + 2 * a] = 0.5f * window_host[2 * m];
            pl[l * lay.row + 2 * a + 1] = 0.5f * window_host[2 * m + 1];
        }
        for (int q = 0; q < R1; ++q) {
            const double th = two_pi * (double)((long long)l * q % C) / (double)C;
            pl[lay.tab + l * lay.row + 2 * q] = (float)cos(th);
            pl[lay.tab + l * lay.row + 2 * q + 1] = (float)(-sin(th));
        }
    }
    for (int k = 0; k <= C / 2; ++k) {
        const double th = two_pi * (double)k / (double)n_fft;
        pl[lay.vk + 2 * k] = (float)(-sin(th));
        pl[lay.vk + 2 * k + 1] = (float)(-cos(th));
    }
    memcpy(pl + lay.win, window_host, sizeof(float) * n_fft);
    for (int i = 0; i < R1 / 2; ++i)
        for (int l = 0; l < L; ++l)
            for (int c = 0; c < 4; ++c) {
                const int a = 2 * i + (c >> 1);
                pl[lay.wtg + (i * L + l) * 4 + c] = 0.5f * window_host[2 * (l + L * a) + (c & 1)];
            }
    return PSND_OK;
}

static int stft_fwd_impl(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                         const void *plan, float mag_eps, float *mag, float *phase, float *re, float *im,
                         int nfk, void *stream) {
    if (!wav || !plan) PSND_FAIL(PSND_E_ARG, "stft_fwd: null wav/plan");
    if ((re == nullptr) != (im == nullptr)) PSND_FAIL(PSND_E_ARG, "stft_fwd: re and im must be given together");
    if (!mag && !phase && !re) PSND_FAIL(PSND_E_ARG, "stft_fwd: no output requested");
    if (framing < PSND_FRAMING_CENTER || framing > PSND_FRAMING_NONE) PSND_FAIL(PSND_E_ARG, "stft_fwd: framing=%d", framing);
    if (hop <= 0 || N < 0) PSND_FAIL(PSND_E_ARG, "stft_fwd: hop=%d N=%lld", hop, (long long)N);
    if (psnd_stft_plan_bytes(n_fft) == 0) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_fwd: n_fft=%d unsupported", n_fft);
    const int pad = framing == PSND_FRAMING_NONE ? 0 : (framing == PSND_FRAMING_CENTER ? n_fft / 2 : (n_fft - hop) / 2);
    if (pad < 0 || T <= pad) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: reflect padding %d needs T > pad (T=%lld)", pad, (long long)T);
    if (T >= ((int64_t)1 << 31) - 4 * (int64_t)n_fft) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: T=%lld exceeds 2^31 samples per clip", (long long)T);
    const int64_t F = psnd_frame_count(T, n_fft, hop, framing);
    if (N == 0 || F <= 0) return PSND_OK;
    const int64_t K = n_fft / 2 + 1;
    if (K * F >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: K*F=%lld exceeds 2^31 per clip", (long long)(K * F));
    if (N > 65535 * 1024) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: N too large");
    StftFwdParams p;
    p.wav = wav, p.plan = static_cast<const float *>(plan);
    p.mag = mag, p.phase = phase, p.re = re, p.im = im;
    p.T = T, p.F = F, p.hop = hop, p.pad = pad, p.mag_eps = mag_eps, p.nfk = nfk;
    {
        const char *ab = PSND_ENV("PSND_ABLATE");
        p.ablate = ab ? atoi(ab) : 0;
#ifdef PSND_TRACE
        const char *tp = PSND_ENV("PSND_TRACE_PTR");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
        const char *ti = PSND_ENV("PSND_TRACE_ITER");
        p.trace_iter = ti ? atoi(ti) : 0;
#endif
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nfk) {
        // bin-fastest magnitudes: a frame's spectrum is one contiguous run.  n_fft = 1024: a wave owns four frames (psnd_stft_q.hip);
        // n_fft = 4096: one wave per frame (psnd_stft_w.hip); any other size: the one-frame-per-workgroup kernel below.
        if (n_fft == 1024 && psnd_stft1024q_ok(T, F, hop, pad) && !PSND_ENV("PSND_STFT_GENERIC"))
            return psnd_stft1024q_launch(wav, static_cast<const float *>(plan), mag, N, T, F, hop, pad, mag_eps, p.ablate, s);
        // n_fft = 4096 at hop 1024 (config 5): the same transform fed from a workgroup-shared LDS sample ring (psnd_stft_r.hip)
        if (n_fft == 4096 && psnd_stft4096r_ok(T, F, hop, pad) && (reinterpret_cast<uintptr_t>(wav) & 15) == 0 && !PSND_ENV("PSND_STFT_GENERIC") &&
            !PSND_ENV("PSND_STFT4096_NORING"))
            return psnd_stft4096r_launch(wav, static_cast<const float *>(plan), mag, N, T, F, pad, mag_eps, p.ablate, s);
        if (n_fft == 4096 && psnd_stft4096w_ok(T, F, hop, pad) && !PSND_ENV("PSND_STFT_GENERIC"))
            return psnd_stft4096w_launch(wav, static_cast<const float *>(plan), mag, N, T, F, hop, pad, mag_eps, p.ablate, 1, s);
        if (F > 0x7fffffff || N > 65535) PSND_FAIL(PSND_E_SHAPE, "stft_mag_nfk(generic): grid too large");
        p.ntile = 0, p.total_tiles = 0;
        if (const Decomp *dd = find_decomp(n_fft)) p.plan += plan_layout(n_fft, dd->R1, dd->L).win;      // the raw window inside a tuned plan
        hipLaunchKernelGGL((stft_fwd_generic_kernel<true, false, false>), dim3((unsigned)F, (unsigned)N), dim3(256), sizeof(float) * (size_t)n_fft, s, p, n_fft);
        PSND_CHECK_LAUNCH("stft_mag_nfk(generic)");
        return PSND_OK;
    }
    const Decomp *d = find_decomp(n_fft);
    if (d) {
        const int FT = 512 / d->R1;
        const int64_t ntile = (F + FT - 1) / FT;
        if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: too many tiles");
        p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
        switch (n_fft) {
            case 256: return launch_tuned<16, 8>(p, mag, phase, re, s);
            case 512:
                // span-staged packed kernel (32 frames per workgroup, two pass-1 rounds); odd hops take the two-pass kernel
                if (!span_kernel_ok<16>(hop) || PSND_ENV("PSND_STFT_V1")) return launch_tuned<16, 16>(p, mag, phase, re, s);
                return launch_span<16>(p, mag, phase, re, s);
            case 1024: {
                // span-staged kernel: hop multiple of 4 and the tile's span (+ bank padding) must fit the
                // exchange area; anything else takes the generic two-pass kernel
                if (hop % 4 != 0 || !span_kernel_ok<32>(hop) || PSND_ENV("PSND_STFT_V1")) return launch_tuned<32, 16>(p, mag, phase, re, s);
                return launch_span<32>(p, mag, phase, re, s);
            }
            case 2048:
                // span-staged packed kernel, 32-point second pass: 2 workgroups per CU (78.6 KB of LDS, ~200 VGPRs)
                if (!span_kernel_ok<32, 32>(hop) || PSND_ENV("PSND_STFT_V1")) return launch_tuned<32, 32>(p, mag, phase, re, s);
                return launch_span<32, 32>(p, mag, phase, re, s);
        }
    }
    if (n_fft == 4096 && mag && !phase && !re && psnd_stft4096w_ok(T, F, hop, pad) && !PSND_ENV("PSND_STFT_GENERIC") && !PSND_ENV("PSND_STFT4096_V1") &&
        !PSND_ENV("PSND_STFT4096_V2")) {
        // magnitude only (LogMelSpectrogram, the losses): one wave per frame, 16 frames per workgroup, 64-byte store runs
        // (psnd_stft_w.hip).  One workgroup per CU: it pays from ~8 tiles per CU on (32 clips x 30 s: 229 us against 245 us for the
        // 4-frame kernel below, write traffic 1.27 x instead of 1.94 x the magnitudes; 16 clips: 119 us against 94 us) - smaller
        // launches keep the 4-frame kernel with its two workgroups per CU.
        const int64_t tiles16 = N * ((F + 15) / 16);
        if (tiles16 >= 2048 || PSND_ENV("PSND_STFT4096_W"))
            return psnd_stft4096w_launch(wav, static_cast<const float *>(plan), mag, N, T, F, hop, pad, mag_eps, p.ablate, 0, s);
    }
    if (n_fft == 4096 && hop % 2 == 0 && hop <= 1364 && !PSND_ENV("PSND_STFT_GENERIC") && !PSND_ENV("PSND_STFT4096_V1")) {
        // 4-frame tiles, two workgroups per CU (span <= 4 pieces per thread)
        const int64_t ntile = (F + k4096bFT - 1) / k4096bFT;
        if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: too many tiles");
        p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
        return launch_n4096b(p, mag, phase, re, s);
    }
    if (n_fft == 4096 && hop % 2 == 0 && hop <= 1792 && !PSND_ENV("PSND_STFT_GENERIC")) {   // span <= 8 pieces per thread
        const int64_t ntile = (F + k4096FT - 1) / k4096FT;
        if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: too many tiles");
        p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
        return launch_n4096(p, mag, phase, re, s);
    }
    // generic path
    if (F > 0x7fffffff || N > 65535) PSND_FAIL(PSND_E_SHAPE, "stft_fwd(generic): grid too large");
    p.ntile = 0, p.total_tiles = 0;
    const size_t lds = sizeof(float) * (size_t)n_fft;
    dim3 grid((unsigned)F, (unsigned)N);
    const bool m = mag, ph = phase, ri = re;
    if (m && !ph && !ri) hipLaunchKernelGGL((stft_fwd_generic_kernel<true, false, false>), grid, dim3(256), lds, s, p, n_fft);
    else if (m && ph && !ri) hipLaunchKernelGGL((stft_fwd_generic_kernel<true, true, false>), grid, dim3(256), lds, s, p, n_fft);
    else if (!m && !ph && ri) hipLaunchKernelGGL((stft_fwd_generic_kernel<false, false, true>), grid, dim3(256), lds, s, p, n_fft);
    else if (m && !ph && ri) hipLaunchKernelGGL((stft_fwd_generic_kernel<true, false, true>), grid, dim3(256), lds, s, p, n_fft);
    else hipLaunchKernelGGL((stft_fwd_generic_kernel<true, true, true>), grid, dim3(256), lds, s, p, n_fft);
    PSND_CHECK_LAUNCH("stft_fwd(generic)");
    return PSND_OK;
}

extern "C" int psnd_stft_fwd(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                             const void *plan, float mag_eps, float *mag, float *phase, float *re, float *im,
                             void *stream) {
    return stft_fwd_impl(wav, N, T, n_fft, hop, framing, plan, mag_eps, mag, phase, re, im, 0, stream);
}

// psnd_stft_fwd (magnitude only) with the BIN axis fastest: mag_nfk (N, F, K).  Same arithmetic, same algorithmic bytes; a frame's
// spectrum is one contiguous run, so a wave writes whole lines (no 64-byte runs at a 4 F-byte pitch shared with the neighbouring
// workgroup).  For callers that own their consumer (the mel / channels-last kernels of this library take either layout).
extern "C" int psnd_stft_mag_nfk(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                                 const void *plan, float mag_eps, float *mag_nfk, void *stream) {
    if (!mag_nfk) PSND_FAIL(PSND_E_ARG, "stft_mag_nfk: null output");
    return stft_fwd_impl(wav, N, T, n_fft, hop, framing, plan, mag_eps, mag_nfk, nullptr, nullptr, nullptr, 1, stream);
}

// multi_stft_loss (models/sound.py:106-133), one resolution: STFT magnitude of the PREDICTION compared with the target magnitudes in
// registers - part[(n * B + b) * 3 + {0,1,2}] = the three sums over tile b of clip n (B = psnd_stft_fwd_msl_blocks), as
// psnd_stft_loss_partial writes them for its own chunking; the prediction's magnitude never exists in HBM.
template <int R1, int L>
static bool span_one_tile_ok(int hop) {
    using G = SpanGeom<R1, L>;
    const int span = (G::FT - 1) * hop + G::NFFT;
    if (!span_kernel_ok<R1, L>(hop)) return false;
    return L == 32 || (span <= 5 * 1024 && span + 4 * (span / 256 + 1) <= G::WT_OFF);
}
static bool fwd_msl_ok(int n_fft, int hop) {
    if (hop <= 0 || PSND_ENV("PSND_STFT_V1")) return false;
    switch (n_fft) {
        case 512: return span_one_tile_ok<16, 16>(hop);
        case 1024: return hop % 4 == 0 && span_one_tile_ok<32, 16>(hop);
        case 2048: return span_one_tile_ok<32, 32>(hop);
    }
    return false;
}
extern "C" int64_t psnd_stft_fwd_msl_blocks(int64_t T, int n_fft, int hop) {
    if (!fwd_msl_ok(n_fft, hop)) return 0;
    const int64_t F = psnd_frame_count(T, n_fft, hop, PSND_FRAMING_CENTER);
    const int FT = 512 / find_decomp(n_fft)->R1;
    return F <= 0 ? 0 : (F + FT - 1) / FT;
}

template <int R1, int L>
static int launch_span_loss(const StftFwdParams &p, hipStream_t stream) {
    using G = SpanGeom<R1, L>;
    const size_t lds = sizeof(float) * (size_t)(G::TAB + G::area_floats(p.hop) - G::L * G::ROW);
    const int grid = (p.total_tiles + 7) & ~7;
    const int span = (G::FT - 1) * p.hop + G::NFFT;
    int rc;
    if constexpr (L == 32) {
        if (span <= 6 * 1024) rc = span_launch_one(stft_fwd_n1024_kernel<true, false, false, 6, false, 2, R1, L>, grid, lds, stream, p);
        else rc = span_launch_one(stft_fwd_n1024_kernel<true, false, false, 10, false, 2, R1, L>, grid, lds, stream, p);
    } else {
        rc = span_launch_one(stft_fwd_n1024_kernel<true, false, false, 5, false, 2, R1, L>, grid, lds, stream, p);
    }
    if (rc != PSND_OK) return rc;
    PSND_CHECK_LAUNCH("stft_fwd_msl");
    return PSND_OK;
}

extern "C" int psnd_stft_fwd_msl(const float *wav, int64_t N, int64_t T, int n_fft, int hop, const void *plan, float mag_eps,
                                 const float *t_mag, float eps, double *part, void *stream) {
    if (!wav || !plan || !t_mag || !part) PSND_FAIL(PSND_E_ARG, "stft_fwd_msl: null pointer");
    if (N < 0) PSND_FAIL(PSND_E_ARG, "stft_fwd_msl: N=%lld", (long long)N);
    if (!fwd_msl_ok(n_fft, hop)) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_fwd_msl: n_fft=%d hop=%d", n_fft, hop);
    const int pad = n_fft / 2;
    if (T <= pad) PSND_FAIL(PSND_E_SHAPE, "stft_fwd_msl: reflect padding %d needs T > pad (T=%lld)", pad, (long long)T);
    if (T >= ((int64_t)1 << 31) - 4 * (int64_t)n_fft) PSND_FAIL(PSND_E_SHAPE, "stft_fwd_msl: T=%lld exceeds 2^31 samples per clip", (long long)T);
    const int64_t F = psnd_frame_count(T, n_fft, hop, PSND_FRAMING_CENTER);
    if (N == 0 || F <= 0) return PSND_OK;
    if ((int64_t)(n_fft / 2 + 1) * F >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_fwd_msl: K*F too large");
    StftFwdParams p;
    memset(&p, 0, sizeof(p));
    p.wav = wav, p.plan = static_cast<const float *>(plan);
    p.T = T, p.F = F, p.hop = hop, p.pad = pad, p.mag_eps = mag_eps;
    p.msl_t = t_mag, p.msl_part = part, p.msl_eps = eps;
    const int FT = 512 / find_decomp(n_fft)->R1;
    const int64_t ntile = (F + FT - 1) / FT;
    if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_fwd_msl: too many tiles");
    p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (n_fft) {
        case 512: return launch_span_loss<16, 16>(p, s);
        case 1024: return launch_span_loss<32, 16>(p, s);
        default: return launch_span_loss<32, 32>(p, s);
    }
}

// Fused wav -> log-mel for LogMelSpectrogram.forward (transforms.py:229-244), Audio2Mel.forward (:351-366) and the
// interface's MelSpectrogram (interface/hifi_gan.py:46-63): n_fft = 1024, hop <= 256 (the span-staged tile kernel).
extern "C" int psnd_logmel_fwd(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing, const void *stft_plan,
                               float mag_eps, int M, const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min,
                               float clamp_lo, float clamp_hi, float *out, void *stream) {
    if (!wav || !stft_plan || !mel_plan || !out) PSND_FAIL(PSND_E_ARG, "logmel_fwd: null pointer");
    if (n_fft != 1024 || hop <= 0 || hop > 256 || hop % 4 != 0)
        PSND_FAIL(PSND_E_UNSUPPORTED, "logmel_fwd: fused kernel covers n_fft=1024, hop<=256 (multiple of 4); got %d/%d", n_fft, hop);
    if (framing < PSND_FRAMING_CENTER || framing > PSND_FRAMING_NONE) PSND_FAIL(PSND_E_ARG, "logmel_fwd: framing=%d", framing);
    if (M <= 0 || N < 0) PSND_FAIL(PSND_E_ARG, "logmel_fwd: M=%d N=%lld", M, (long long)N);
    if (log_kind < PSND_LOG_NONE || log_kind > PSND_LOG_10) PSND_FAIL(PSND_E_ARG, "logmel_fwd: log_kind=%d", log_kind);
    const int pad = framing == PSND_FRAMING_NONE ? 0 : (framing == PSND_FRAMING_CENTER ? n_fft / 2 : (n_fft - hop) / 2);
    if (T <= pad) PSND_FAIL(PSND_E_SHAPE, "logmel_fwd: reflect padding %d needs T > pad (T=%lld)", pad, (long long)T);
    if (T >= ((int64_t)1 << 31) - 4 * (int64_t)n_fft) PSND_FAIL(PSND_E_SHAPE, "logmel_fwd: T=%lld exceeds 2^31 samples per clip", (long long)T);
    const int64_t F = psnd_frame_count(T, n_fft, hop, framing);
    if (N == 0 || F <= 0) return PSND_OK;
    if ((int64_t)M * F >= (int64_t)1 << 29) PSND_FAIL(PSND_E_SHAPE, "logmel_fwd: M*F too large");
    StftFwdParams p;
    memset(&p, 0, sizeof(p));
    p.wav = wav, p.plan = static_cast<const float *>(stft_plan);
    p.T = T, p.F = F, p.hop = hop, p.pad = pad, p.mag_eps = mag_eps;
    p.mel_plan = static_cast<const int *>(mel_plan), p.mel_out = out, p.mel_M = M, p.log_kind = log_kind;
    p.log_offset = log_offset, p.pre_clamp_min = pre_clamp_min, p.clamp_lo = clamp_lo, p.clamp_hi = clamp_hi;
    const int64_t ntile = (F + 15) / 16;
    if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "logmel_fwd: too many tiles");
    p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
    const size_t lds = sizeof(float) * (size_t)(kN1024TabFloats + n1024_area_floats(hop) - 16 * 68);
    const int grid = (p.total_tiles + 7) & ~7;
    hipLaunchKernelGGL((stft_fwd_n1024_kernel<true, false, false, 5, false, 1>), dim3(grid), dim3(256), lds,
                       static_cast<hipStream_t>(stream), p);
    PSND_CHECK_LAUNCH("logmel_fwd");
    return PSND_OK;
}
