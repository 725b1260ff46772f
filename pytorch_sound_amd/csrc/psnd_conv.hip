// psnd_conv.hip - Conv1d stacks of pytorch_sound/models/vocoders/hifi_gan.py:32-147 (ResBlock1/2, conv_pre /
// conv_post) as ONE implicit-GEMM bf16 MFMA kernel with fused epilogue, for gfx950.
//
// Layout ("CL"): activations are channels-last bf16 matrices  X[r][c],  r = clip * Lp + l,  Lp >= L + 2*HP,
// rows l in [HP, HP+L) hold the clip, all other rows are ZERO.  A stride-1 dilated convolution is then
//      Y[r][co] = sum_j sum_ci  X[r + off_j][ci] * W[j][co][ci],        off_j = off0 + j * dstep
// i.e. every tap is a row shift of the same matrix: clip boundaries need no special case (the halo rows
// supply the zero padding), both MFMA operands are K(=ci)-contiguous, and the SAME kernel computes
//      forward        A = X,  W = weight[co][ci][j] packed [j][co][ci],  off_j =  j*dil - pad
//      backward-data  A = gY, W = weight packed [j][ci][co],             off_j = pad - j*dil
// Epilogue (fused, replaces 3-4 elementwise passes per conv of the reference):
//      v = acc (+ bias[co]) ; v *= leaky'(mask_src) ; v += res ; rows outside the clip -> 0
//      out_raw = bf16(v) (optional) ; out_act = bf16(leaky_relu(v, slope)) (optional)
//
// Tiling: 256 threads = 2 x 2 waves, block tile 64 (or 128) rows x 64 output channels, one (two) v_mfma_f32_32x32x16_bf16
// accumulator(s) (32 x 32) per wave; K walks input channels in stages of 32.  The A tile of a stage goes through LDS (row
// stride 80 B = conflict-free ds_read_b128 fragments) and carries its +-HM halo rows, so the k taps re-use one staged tile
// (that is the implicit-GEMM saving); the weights are packed in MFMA fragment order by the prep kernels and go straight
// from L2 into the B operand registers (pack_index, conv_cl_body).  The backward of a conv is ONE launch: input-gradient
// and weight-gradient roles share the grid (conv_bwd_pair_kernel), the weight-norm backward of a whole chain another one.
#include "psnd_common.h"
#include "psnd_conv_pair.h"
#include <stdlib.h>
#include <string.h>
#include <atomic>

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;     // 8 bf16 = 16 B MFMA operand
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // round to nearest even (NaN stays NaN)
    unsigned u = __builtin_bit_cast(unsigned, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair, round to nearest even, ONE instruction (v_cvt_pk_bf16_f32; the integer f2bf above costs ~6)
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hwbf16x2));
}

struct ConvParams {
    const bf16_t *A;         // (R, Ca)
    const bf16_t *A2;        // (R, Ca) or null: A_eff = A + A2 * (AM > 0 ? 1 : a2_slope)  (A may be null -> 0)
    const bf16_t *AM;        // (R, Ca) sign source for A2
    const bf16_t *W;         // [k][Cb][Ca] in fragment order (pack_index)
    const float *bias;       // Cb or null
    const bf16_t *res;       // (R, Cb) or null
    const bf16_t *mask_src;  // (R, Cb) or null: v *= (mask_src > 0 ? 1 : mask_slope)
    bf16_t *out_raw;         // (R, Cb) or null
    bf16_t *out_act;         // (R, Cb) or null
    bf16_t *a_eff_out;       // (R, Ca) or null: the combined operand A + A2 * leaky'(AM), materialised (COMBINE only)
    long long R;
    int Lp, L, HP, Ca, Cb, k, off0, dstep, hm;
    float act_slope, mask_slope, a2_slope;
    // transposed-conv ("up") mode, ConvTranspose1d(k = 2u, stride u, padding p) of hifi_gan.py:109 in polyphase form.  The GEMM
    // rows are the LOW-resolution rows (geometry Lp / L / HP above); the HIGH-resolution CL buffer (N, LpO, Cr), clip rows
    // [HPO, HPO + LO), is seen as a matrix of u * Cr columns: low row l <-> the u consecutive high rows starting at
    // HPO - p + (l - HP) * u, i.e. column block phi of row l is high row t = (l - HP) * u + phi - p of the clip.
    //   up_role 1: the OUTPUT is that view (forward: Y[l][phi, co] = x[l] W[phi] + x[l-1] W[phi + u], Cb = u * Cr)
    //   up_role 2: the A operand is that view (input gradient: gx[l] = G[l] Wt[phi] + G[l+1] Wt[phi + u], Ca = u * Cr)
    int up_role, up_u, up_p, up_LpO, up_HPO, up_LO, up_Cr;
    // res_is_a_eff: the residual of the epilogue is the combined operand A + A2 * leaky'(AM) of THIS launch (Ca == Cb).  It is
    // formed again from its three sources - reading a_eff_out back would race with the workgroups that are still writing it
    // (round 1 did exactly that for ResBlock2 chains wider than one column tile: hifi_gan_v3's 128-channel stage)
    int res_is_a_eff;
#ifdef PSND_TRACE
    long long *trace;
#endif
};
// first high-resolution row (global row index of the (N, LpO, Cr) buffer) of low-resolution row r, or -1 when the u rows do
// not lie inside the clip's buffer (such low rows carry no valid output)
__device__ __forceinline__ long long up_row_base(int Lp, int HP, int u, int pp, int LpO, int HPO, long long r) {
    const long long n = r / Lp;
    const int l = (int)(r - n * Lp);
    const int b = HPO - pp + (l - HP) * u;
    return (b >= 0 && b + u <= LpO) ? n * (long long)LpO + b : -1;
}
#ifdef PSND_TRACE
#define PSND_CSTAMP(i_)                                                                                         \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && p.trace)                                                                 \
            p.trace[(tblk * 4 + (threadIdx.x >> 6)) * 8 + (i_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PSND_CSTAMP(i_)
#endif

constexpr int BM = 64, BN = 64, RS = 40;   // RS: LDS row stride in bf16 (80 B) of the 32-channel tiles (wgrad)

// 8 consecutive bf16 of  g = G1 + G2 * leaky'(M)  (either term optional): the gradient wrt a conv output that
// was handed out both raw and through leaky_relu (sign(M) == sign of the pre-activation).
__device__ __forceinline__ uint4 load_combined(const bf16_t *G1, const bf16_t *G2, const bf16_t *M, float slope, size_t o) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (G1) v = *reinterpret_cast<const uint4 *>(G1 + o);
    if (G2) {
        const uint4 g2 = *reinterpret_cast<const uint4 *>(G2 + o);
        const uint4 m = *reinterpret_cast<const uint4 *>(M + o);
        const unsigned *pv = reinterpret_cast<const unsigned *>(&v), *pg = reinterpret_cast<const unsigned *>(&g2),
                       *pm = reinterpret_cast<const unsigned *>(&m);
        unsigned out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = bf2f((bf16_t)(pv[i] & 0xffff)), a1 = bf2f((bf16_t)(pv[i] >> 16));
            const float b0 = bf2f((bf16_t)(pg[i] & 0xffff)), b1 = bf2f((bf16_t)(pg[i] >> 16));
            const float m0 = bf2f((bf16_t)(pm[i] & 0xffff)), m1 = bf2f((bf16_t)(pm[i] >> 16));
            const float r0 = a0 + b0 * (m0 > 0.f ? 1.f : slope), r1 = a1 + b1 * (m1 > 0.f ? 1.f : slope);
            out[i] = pack_bf16(r0, r1);
        }
        v = make_uint4(out[0], out[1], out[2], out[3]);
    }
    return v;
}

// Weight packs are stored in MFMA B-fragment order: for tap j, 32-column tile nt, 16-channel step
// ks, lane l = (c % 16 / 8) * 32 + n % 32 holds the 8 consecutive reduction channels 8 * (c % 16 / 8) .. + 7 of column n -
// one wave reads its whole B operand of a k-step as ONE contiguous 1 KB load, straight from L2 into registers (no LDS).
// Same bytes as the row-major [j][n][c] pack (n, c multiples of 32).
constexpr int KFRAG = 16;     // every k: fragment order (taps are walked in chunks of TC = 3 through the register ring)
#ifndef PSND_DB_PLAIN
#define PSND_DB_PLAIN 4
#endif
#ifndef PSND_DB_COMBINE
#define PSND_DB_COMBINE 2
#endif
__device__ __host__ __forceinline__ size_t pack_index(int /*k*/, int j, int n, int c, int Nn, int Nc) {
    return ((((size_t)j * (Nn >> 5) + (n >> 5)) * (Nc >> 4) + (c >> 4)) * 64 + (((c & 15) >> 3) << 5) + (n & 31)) * 8 + (c & 7);
}

constexpr int MAXK = 16;
constexpr int KC = 32;                     // input channels per pipeline stage
constexpr int PCS = KC / 8;                // 16-byte pieces per staged row
constexpr int NA = (BM + 2 * 25) * PCS / 256 + 1;   // A pieces per thread per stage (tap reach hm <= 25)

// One conv = 384 workgroups of ~6 MFLOP at the config-2 shape: every workgroup is a LATENCY chain, not a
// throughput problem (first version: one stage prefetched ahead, 8 stages x ~2.5 k cycles of load latency =
// 24 us per launch with the MFMA pipe 4 % busy).  So the loads run D stages ahead in a register ring (the
// kernel only ever has 1-2 workgroups per CU: registers are free), the raw pieces of the on-load gradient
// combine are kept apart until the commit (the arithmetic would otherwise wait for its loads inside the
// fetch), LDS is double buffered (one barrier per stage).
//   KT: taps the register ring is sized for (k <= KT);  D: stages in flight;  COMBINE: A = A + A2 * leaky'(AM)
//   MT: 32-row MFMA tiles per wave (1: 64-row workgroup tile; 2: 128 rows - every B fragment feeds two MFMAs, for launches with
//   enough rows to fill the chip anyway: the stage loop is bound by the B fragment loads, whose bytes per FLOP go with 1 / rows)
//   HMX: largest tap reach the staged A tile (and its register ring) is sized for: 25 covers every conv of hifi_gan_v1 / v2
//   (k = 11, dilation 5); the 7-tap instances also exist with 40 (hifi_gan_v3: k = 7, dilation 12 -> 36)
//   UPM: the transposed-conv ("up") address maps are compiled in (ConvParams::up_role); the plain instances carry none of it
//   KCT: input channels per pipeline stage.  Only 32 is instantiated: 64-channel stages (half the barriers, same bytes in flight)
//   were built for the 3-tap instances in round 2 and measured no faster - config-2 step 1.242 ms against 1.236 ms, forward
//   launch 13.8 us against 13.5 us - so the stage loop is not bound by the barrier / LDS hand-over latency per stage
// WNC = waves across the output channels: 2 (tile 64 MT rows x 64 channels) or 1 (narrow layers, Cb <= 32: the four waves stack
// along the rows, tile 128 MT rows x 32 channels - with the 2 x 2 arrangement half of the waves multiplied zero weight fragments)
template <int KT, int D, bool COMBINE, int NBUF, int MT, int HMX = 25, bool UPM = false, int KCT = 32, int WNC = 2>
__device__ __forceinline__ void conv_cl_body(const ConvParams &p, const int bx, const int by, bf16_t *smem_c, const size_t tblk) {
    static_assert(KCT == 32, "only 32-channel stages are validated");
    static_assert(WNC == 1 || WNC == 2, "waves across the channels");
    constexpr int KC = KCT, PCS = KCT / 8, KS = KCT / 16;
    constexpr int RS = KC + 8;             // LDS row stride (bf16): 80 B / 144 B, odd multiples of 16 B
    constexpr int BNt = 32 * WNC, BMt = 32 * (4 / WNC) * MT, NAt = (BMt + 2 * HMX) * PCS / 256 + 1;
    const int rowsA = BMt + 2 * p.hm;
    const int buf_elems = rowsA * RS;      // two A stage buffers; the weights never enter LDS
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = WNC == 2 ? wave >> 1 : wave, wn = WNC == 2 ? wave & 1 : 0;
    const long long r0 = (long long)bx * BMt;
    const int n0 = by * BNt;
    const int li = lane & 31, kg = lane >> 5;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    PSND_CSTAMP(0);

    // B operand: the pack is in fragment order (pack_index) and every wave loads its own B fragments straight from L2 into the
    // registers the MFMAs read - 16 B per lane, 1 KB contiguous per wave instruction - in units of TC = 3 taps x 2 k-steps, DB
    // units ahead.  The weights never enter LDS: with them staged there the stage loop was LDS-bandwidth bound at k = 3 (48 KB of
    // fragment reads + 17 KB of stage writes per stage and workgroup, half of it weights), and the 7- and 11-tap convs of the
    // HiFi-GAN blocks (43 / 65 KB of weights per stage buffer) ran ONE workgroup per CU.
    // (With a ROW-MAJOR pack the same idea lost, 11 -> 18 us: 64 separate 32-byte pieces per wave instruction.)
    constexpr int TC = 3, NCH = (KT + TC - 1) / TC, UNITS = D * NCH;
    // D = 1: the single-stage instances for 32 input channels (no A ring to turn; a shallow B ring keeps them under 168 VGPRs)
    constexpr int DB = D == 1 ? (UNITS % 2 == 0 ? 2 : (UNITS % 3 == 0 ? 3 : 1))
                     : KCT == 64 ? (UNITS % 2 == 0 ? 2 : 1)          // 64-channel stages: a unit is twice the registers
                     : KT <= 3 ? (COMBINE ? PSND_DB_COMBINE : PSND_DB_PLAIN) : (UNITS % 4 == 0 ? 4 : (UNITS % 3 == 0 ? 3 : 2));
    static_assert(UNITS % DB == 0, "B ring depth must divide the units of a ring turn (slots are static inside a turn)");
    uint4 ra[D][NAt], ra2[COMBINE ? D : 1][NAt], ram[COMBINE ? D : 1][NAt], rbf[DB][KS * TC];
    const int nA = rowsA * PCS;
    // Every load is a buffer load with a 32-bit byte offset; an offset of OOB (or any offset past the tensor)
    // returns zeros.  That supplies the rows before / after the tensor, the channels past Ca of a padding stage
    // and the taps past k WITHOUT a branch around the load: hipcc's s_waitcnt insertion counts outstanding loads
    // exactly only in straight-line code - one conditional fetch and every later wait degrades to vmcnt(0),
    // which serialises the whole ring (measured: 1.7 k cycles per stage instead of 0.5 k).
    constexpr unsigned OOB = 0xffffffffu;
    const bool upA = UPM && p.up_role == 2;    // the A operand is the high-resolution view
    const unsigned a_bytes = upA ? (unsigned)((size_t)(p.R / p.Lp) * p.up_LpO * p.up_Cr * sizeof(bf16_t))
                                 : (unsigned)((size_t)p.R * p.Ca * sizeof(bf16_t));
    const unsigned w_bytes = (unsigned)((size_t)p.k * p.Cb * p.Ca * sizeof(bf16_t));
    const __amdgpu_buffer_rsrc_t rA = make_uniform_rsrc(p.A ? p.A : p.A2, (int)a_bytes);
    __amdgpu_buffer_rsrc_t rA2 = rA, rAM = rA;
    if constexpr (COMBINE) rA2 = make_uniform_rsrc(p.A2, (int)a_bytes), rAM = make_uniform_rsrc(p.AM, (int)a_bytes);
    const __amdgpu_buffer_rsrc_t rW = make_uniform_rsrc(p.W, (int)w_bytes);
    // the combined operand is written back by the first column block only; a null pointer gives a zero-sized buffer,
    // lanes that must not store use the OOB offset: the store instruction itself is unconditional (see above)
    const __amdgpu_buffer_rsrc_t rG = make_uniform_rsrc(p.a_eff_out ? p.a_eff_out : p.W,
                                                        (COMBINE && p.a_eff_out && by == 0) ? (int)a_bytes : 0);
    const bool haveA = p.A != nullptr;
    unsigned aoff[NAt];
#pragma unroll
    for (int u = 0; u < NAt; ++u) {
        const int idx = tid + 256 * u;
        const int rr = idx / PCS, pc = idx % PCS;
        const long long r = r0 - p.hm + rr;
        if (!upA) {
            aoff[u] = (idx < nA && r >= 0 && r < p.R) ? (unsigned)(((size_t)r * p.Ca + 8 * pc) * sizeof(bf16_t)) : OOB;
        } else {
            const long long hb = (idx < nA && r >= 0 && r < p.R) ? up_row_base(p.Lp, p.HP, p.up_u, p.up_p, p.up_LpO, p.up_HPO, r) : -1;
            aoff[u] = hb >= 0 ? (unsigned)(((size_t)hb * p.up_Cr + 8 * pc) * sizeof(bf16_t)) : OOB;
        }
    }
    auto ld16 = [&](__amdgpu_buffer_rsrc_t r, unsigned off) __attribute__((always_inline)) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return __builtin_bit_cast(uint4, v);
    };

    auto fetch = [&](auto sc, int c0) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const bool live = c0 < p.Ca;                  // false for the padding stages of the last ring turn
        const unsigned cb = (unsigned)c0 * (unsigned)sizeof(bf16_t);
#pragma unroll
        for (int u = 0; u < NAt; ++u) {
            const bool chan = KCT == 32 || c0 + 8 * ((tid + 256 * u) % PCS) < p.Ca;     // Ca % 32 == 0: a 64-channel stage may be half empty
            const unsigned o = (live && chan && aoff[u] != OOB) ? aoff[u] + cb : OOB;
            ra[s][u] = ld16(rA, haveA ? o : OOB);
            if constexpr (COMBINE) {
                ra2[s][u] = ld16(rA2, o);
                ram[s][u] = ld16(rAM, o);
            }
        }
    };
    // fragment-ordered pack: tile (n0 / 32 + wn) of tap j, k-step c0 / 16 + kk  ->  1 KB per wave
    const unsigned fbase = (n0 + wn * 32 < p.Cb) ? (unsigned)((size_t)((n0 >> 5) + wn) * (size_t)(p.Ca >> 4) * 1024u) + (unsigned)lane * 16u : OOB;
    const unsigned ftap = (unsigned)((size_t)(p.Cb >> 5) * (size_t)(p.Ca >> 4) * 1024u);
    // unit u of a ring turn = (stage u / NCH, tap chunk u % NCH); slot = u % DB
    auto fetch_b = [&](auto slotc, auto chunkc, int c0) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value, q = decltype(chunkc)::value;
        const bool live = c0 < p.Ca && fbase != OOB;
        const unsigned ks = (unsigned)(c0 >> 4) * 1024u;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                rbf[slot][KS * t + kk] = ld16(rW, (live && q * TC + t < p.k && c0 + 16 * kk < p.Ca) ? fbase + (unsigned)(q * TC + t) * ftap + ks + (unsigned)kk * 1024u : OOB);
    };
    auto combine = [&](uint4 v, uint4 g2, uint4 m) __attribute__((always_inline)) {
        const unsigned *pv = reinterpret_cast<const unsigned *>(&v), *pg = reinterpret_cast<const unsigned *>(&g2),
                       *pm = reinterpret_cast<const unsigned *>(&m);
        unsigned out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a0 = bf2f((bf16_t)(pv[i] & 0xffff)), a1 = bf2f((bf16_t)(pv[i] >> 16));
            const float b0 = bf2f((bf16_t)(pg[i] & 0xffff)), b1 = bf2f((bf16_t)(pg[i] >> 16));
            const float m0 = bf2f((bf16_t)(pm[i] & 0xffff)), m1 = bf2f((bf16_t)(pm[i] >> 16));
            const float r0_ = a0 + b0 * (m0 > 0.f ? 1.f : p.a2_slope), r1_ = a1 + b1 * (m1 > 0.f ? 1.f : p.a2_slope);
            out[i] = pack_bf16(r0_, r1_);
        }
        return make_uint4(out[0], out[1], out[2], out[3]);
    };
    auto commit = [&](auto sc, bf16_t *sA, int c0) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int u = 0; u < NAt; ++u) {
            const int idx = tid + 256 * u;
            uint4 v = ra[s][u];
            if constexpr (COMBINE) {
                v = combine(v, ra2[s][u], ram[s][u]);
                const int rr = idx / PCS;
                const bool own = rr >= p.hm && rr < p.hm + BMt && aoff[u] != OOB && c0 + 8 * (idx % PCS) < p.Ca;     // rows of this tile
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rG,
                                                       (int)(own ? aoff[u] + (unsigned)c0 * 2u : OOB), 0, 0);
            }
            if (idx < nA) *reinterpret_cast<uint4 *>(sA + (idx / PCS) * RS + 8 * (idx % PCS)) = v;
        }
    };

    // the ring turns whole: stages past Ca / KC load and multiply zeros (at most D - 1 of them)
    const int nchunk = (p.Ca / KC + D - 1) / D * D;
    static_for<0, D>([&](auto sc) __attribute__((always_inline)) { fetch(sc, decltype(sc)::value * KC); });
    static_for<0, DB>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        fetch_b(uc, std::integral_constant<int, u % NCH>{}, (u / NCH) * KC);
    });
    for (int c = 0; c < nchunk; c += D) {
        static_for<0, D>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int ch = c + s;
            bf16_t *sA = smem_c + (ch & 1) * buf_elems;
            commit(sc, sA, ch * KC);
            if (ch == 0) PSND_CSTAMP(1);
            __syncthreads();
            if (ch == 0) PSND_CSTAMP(2);
            fetch(sc, (ch + D) * KC);
            static_for<0, NCH>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, u = s * NCH + q, slot = u % DB;
                static_for<0, TC>([&](auto tc) __attribute__((always_inline)) {
                    constexpr int tap = q * TC + decltype(tc)::value;
                    // no run-time test of tap < p.k: a uniform branch per tap cuts the stage into basic blocks and every pair of MFMAs
                    // then waits for its own two ds_reads (~130 cycles each time).  Taps past k multiply ZERO weight fragments
                    // (fetch_b: offset OOB) with the last real tap's rows - nothing is added.
                    if constexpr (tap < KT) {
                        const int off = p.off0 + (tap < p.k ? tap : p.k - 1) * p.dstep + p.hm;
                        const bf16_t *pa = sA + (wm * 32 * MT + li + off) * RS + 8 * kg;
#pragma unroll
                        for (int kk = 0; kk < KC / 16; ++kk) {
                            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, rbf[slot][KS * decltype(tc)::value + kk]);
#pragma unroll
                            for (int m = 0; m < MT; ++m) {
                                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(pa + m * 32 * RS + 16 * kk);
                                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag, acc[m], 0, 0, 0);
                            }
                        }
                    }
                });
                // the unit DB ahead: same slot; its stage is ch + (q + DB) / NCH (static part), its chunk (q + DB) % NCH
                fetch_b(std::integral_constant<int, slot>{}, std::integral_constant<int, (q + DB) % NCH>{}, (ch + (q + DB) / NCH) * KC);
            });
        });
    }

    PSND_CSTAMP(3);
    // ---- epilogue.  D[i][j]: j = lane & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  The fp32 tile goes
    // through LDS so that every thread finishes 8 consecutive channels of one row: 16-byte loads of the mask /
    // residual operands and 16-byte stores (the first version stored 2 bytes per lane and paid a 64-bit
    // modulo per element: 8.9 k of the workgroup's 22 k cycles).
    constexpr int OS = BNt + 8;                                  // fp32 row stride: 4 rows apart = 32 banks apart
    constexpr int CGN = BNt / 8;                                 // 8-channel groups of a row
    float *sO = reinterpret_cast<float *>(smem_c);
    __syncthreads();                                             // every fragment read of the last stage is done
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int i = (rg & 3) + 8 * (rg >> 2) + 4 * kg;
            sO[(wm * 32 * MT + m * 32 + i) * OS + wn * 32 + li] = acc[m][rg];
        }
    __syncthreads();
    const int l0 = (int)(r0 % p.Lp);                             // uniform
#pragma unroll
    for (int u = 0; u < BMt * CGN / 256; ++u) {
        const int idx = tid + 256 * u, row = idx / CGN, cg = idx % CGN;
        const long long r = r0 + row;
        const int col = n0 + 8 * cg;
        if (r >= p.R || col >= p.Cb) continue;
        const int l = (l0 + row) % p.Lp;
        size_t o = (size_t)r * p.Cb + col;
        float v[8];
        bool inside = l >= p.HP && l < p.HP + p.L;
        if (UPM && p.up_role == 1) {           // forward of the transposed conv: column block phi of low row l = high row t
            // every high row that SOME low row maps to is written (zeros outside the clip), also those of a group of u rows that lies
            // only partly inside the buffer: the caller need not zero the output first as long as the low rows reach both ends
            const int phi = col / p.up_Cr;
            const int hr = p.up_HPO - p.up_p + (l - p.HP) * p.up_u + phi;      // row inside the clip's high-resolution buffer
            if (hr < 0 || hr >= p.up_LpO) continue;
            const int t = hr - p.up_HPO;
            inside = t >= 0 && t < p.up_LO;
            o = ((size_t)(r / p.Lp) * p.up_LpO + hr) * p.up_Cr + (col - phi * p.up_Cr);
        }
        if (inside) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(sO + row * OS + 8 * cg);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(sO + row * OS + 8 * cg + 4);
            v[0] = a0.x, v[1] = a0.y, v[2] = a0.z, v[3] = a0.w, v[4] = a1.x, v[5] = a1.y, v[6] = a1.z, v[7] = a1.w;
            if (p.bias) {
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(p.bias + col), b1 = *reinterpret_cast<const f32x4 *>(p.bias + col + 4);
                v[0] += b0.x, v[1] += b0.y, v[2] += b0.z, v[3] += b0.w, v[4] += b1.x, v[5] += b1.y, v[6] += b1.z, v[7] += b1.w;
            }
            if (p.mask_src) {
                const uint4 m = *reinterpret_cast<const uint4 *>(p.mask_src + o);
                const unsigned *pm = reinterpret_cast<const unsigned *>(&m);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] *= bf2f((bf16_t)(pm[e] & 0xffff)) > 0.f ? 1.f : p.mask_slope;
                    v[2 * e + 1] *= bf2f((bf16_t)(pm[e] >> 16)) > 0.f ? 1.f : p.mask_slope;
                }
            }
            if (p.res_is_a_eff) {
                const uint4 q = load_combined(p.A, p.A2, p.AM, p.a2_slope, o);
                const unsigned *pq = reinterpret_cast<const unsigned *>(&q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += bf2f((bf16_t)(pq[e] & 0xffff));
                    v[2 * e + 1] += bf2f((bf16_t)(pq[e] >> 16));
                }
            }
            if (p.res) {
                const uint4 q = *reinterpret_cast<const uint4 *>(p.res + o);
                const unsigned *pq = reinterpret_cast<const unsigned *>(&q);
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
                w[e] = pack_bf16(x0 > 0.f ? x0 : x0 * p.act_slope, x1 > 0.f ? x1 : x1 * p.act_slope);
            }
            *reinterpret_cast<uint4 *>(p.out_act + o) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
#ifdef PSND_TRACE
    PSND_CSTAMP(4);
    __builtin_amdgcn_s_waitcnt(0);
    PSND_CSTAMP(5);
#endif
}

#ifndef PSND_CONV_OCC
#define PSND_CONV_OCC 2
#endif
template <int KT, int D, bool COMBINE, int NBUF, int MT, int HMX = 25, bool UPM = false, int KCT = 32, int WNC = 2>
__global__ __launch_bounds__(256, (D == 1 ? 3 : PSND_CONV_OCC)) void conv_cl_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_dyn[];
    conv_cl_body<KT, D, COMBINE, NBUF, MT, HMX, UPM, KCT, WNC>(p, blockIdx.x, blockIdx.y, smem_dyn, (size_t)blockIdx.y * gridDim.x + blockIdx.x);
}

// ---- layout conversion: (N, C, T) fp32  <->  CL bf16 (N, Lp, Cp) with zero halo rows / padded channels ----
// to_cl: optional pre-op  0: none, 1: log1p(x)
// MASKBWD: the backward of the mask head below: out = x * mul * s * (1 - s), s = sigmoid(ycl) read at the output position
// MASKBWD with l1_est: x (may be NULL) + coef * g[0] * sign(est - ref) is the gradient on the mask head's output (fused F.l1_loss(est, ref))
// All loads are branch-free buffer loads (an element outside the tensor reads zero through the descriptor's range check) and a thread
// issues its four elements' loads together: with `if (in range) v = x[o]` every load sat in its own basic block and a thread walked
// its four elements as four dependent round trips (15-19 us per launch at 32 x 513 x 173 for 17 MB of traffic).
template <bool MASKBWD>
__global__ __launch_bounds__(256) void to_cl_kernel(const float *x, bf16_t *out, int N, int C, int T, int Lp, int HP, int Cp,
                                                    int preop, const float *mul, const bf16_t *ycl, const float *l1_est = nullptr,
                                                    const float *l1_ref = nullptr, const float *l1_g = nullptr, float l1_coef = 0.f) {
    // tile 32 (t) x 32 (c) through LDS so both sides are coalesced
    __shared__ float tile[32][33];
    constexpr unsigned OOBW = 0xffffffffu;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // 32 x 8
    const int n = blockIdx.z;
    const int t0 = blockIdx.x * 32 - HP, c0 = blockIdx.y * 32;  // rows of the CL buffer: l = t + HP
    const size_t clip = (size_t)n * C * T;
    const int cb = (int)((size_t)C * T * sizeof(float));
    const __amdgpu_buffer_rsrc_t rx = make_uniform_rsrc(x ? x + clip : reinterpret_cast<const float *>(out), x ? cb : 0);
    const bool l1 = MASKBWD && l1_est != nullptr;               // uniform
    const bool tanhb = !MASKBWD && preop == 2;                  // uniform: x * (1 - mul^2), the backward of out = tanh(from_cl(.)) with mul = out
    const __amdgpu_buffer_rsrc_t rm = make_uniform_rsrc((MASKBWD || tanhb) ? mul + clip : reinterpret_cast<const float *>(out), (MASKBWD || tanhb) ? cb : 0);
    const __amdgpu_buffer_rsrc_t re = make_uniform_rsrc(l1 ? l1_est + clip : reinterpret_cast<const float *>(out), l1 ? cb : 0);
    const __amdgpu_buffer_rsrc_t rr = make_uniform_rsrc(l1 ? l1_ref + clip : reinterpret_cast<const float *>(out), l1 ? cb : 0);
    const float l1c = l1 ? l1_coef * l1_g[0] : 0.f;
    float v[4], m[4], e[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, t = t0 + tx;
        const unsigned o = (c < C && t >= 0 && t < T) ? (unsigned)(((size_t)c * T + t) * sizeof(float)) : OOBW;
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)o, 0, 0));
        m[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, (int)o, 0, 0));      // (zero-sized descriptor when unused)
        if constexpr (MASKBWD) {
            e[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(re, (int)o, 0, 0));
            r[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)o, 0, 0));
        }
    }
    // the sigmoid's argument at the OUTPUT positions of this thread: requested now as well
    unsigned short yq[MASKBWD ? 4 : 1];
    if constexpr (MASKBWD) {
        const __amdgpu_buffer_rsrc_t ry = make_uniform_rsrc(ycl + (size_t)n * Lp * Cp, (int)((size_t)Lp * Cp * sizeof(bf16_t)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = blockIdx.x * 32 + ty + 8 * i, c = c0 + tx;
            yq[i] = __builtin_amdgcn_raw_buffer_load_b16(ry, (l < Lp && c < Cp) ? (unsigned)(((size_t)l * Cp + c) * sizeof(bf16_t)) : OOBW, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float w = v[i];
        if (preop == 1) {
            // log1p for a bf16 result: u = 1 + w rounds w away, log(u) * w / (u - 1) puts it back (the classic correction); hardware log2
            // and reciprocal.  ~12 instructions against ~140 for log1pf - the launch was VALU-bound on it (568 VALU instructions per wave
            // for four elements per thread, profiles/r03_step_pmc_per_kernel.txt); the error stays below 1e-6 relative, 2^-13 of a bf16 step.
            const float u = 1.f + w;
            w = u == 1.f ? w : __logf(u) * __fdividef(w, u - 1.f);
        }
        if constexpr (MASKBWD) {
            const float d = e[i] - r[i];
            w += l1c * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            w *= m[i];
        } else if (tanhb) {
            w *= 1.f - m[i] * m[i];
        }
        tile[ty + 8 * i][tx] = w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = ty + 8 * i, l = blockIdx.x * 32 + j, c = c0 + tx;
        if (l < Lp && c < Cp) {
            float w = tile[tx][j];
            if constexpr (MASKBWD) {
                const float sg = 1.f / (1.f + __expf(-bf2f(yq[i])));
                w *= sg * (1.f - sg);
            }
            out[((size_t)n * Lp + l) * Cp + c] = f2bf(w);
        }
    }
}

// from_cl: (N, Lp, Cp) bf16 -> (N, C, T) fp32.   MASK: the separator's mask head, out = sigmoid(x) * mul
// MASK with l1_ref: also part[block] = sum |out - ref| of the block (double; fused F.l1_loss(est, ref))
template <bool MASK>
__global__ __launch_bounds__(256) void from_cl_kernel(const bf16_t *x, float *out, int N, int C, int T, int Lp, int HP, int Cp,
                                                      const float *mul, const float *l1_ref = nullptr, double *l1_part = nullptr, int post = 0) {
    // post == 1 (plain instance): out = tanh(x) - the generator's output non-linearity (hifi_gan.py:134) in the layout change
    __shared__ float tile[32][33];
    constexpr unsigned OOBW = 0xffffffffu;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const size_t clip = (size_t)n * C * T;
    const int cb = (int)((size_t)C * T * sizeof(float));
    const __amdgpu_buffer_rsrc_t rx = make_uniform_rsrc(x + (size_t)n * Lp * Cp, (int)((size_t)Lp * Cp * sizeof(bf16_t)));
    const bool l1 = MASK && l1_ref != nullptr;
    const __amdgpu_buffer_rsrc_t rm = make_uniform_rsrc(MASK ? mul + clip : out, MASK ? cb : 0);
    const __amdgpu_buffer_rsrc_t rr = make_uniform_rsrc(l1 ? l1_ref + clip : out, l1 ? cb : 0);
    unsigned short xq[4];
    float m[4], r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, c = c0 + tx;
        xq[i] = __builtin_amdgcn_raw_buffer_load_b16(rx, (t < T && c < Cp) ? (unsigned)(((size_t)(t + HP) * Cp + c) * sizeof(bf16_t)) : OOBW, 0, 0);
    }
    if constexpr (MASK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, t = t0 + tx;
            const unsigned o = (c < C && t < T) ? (unsigned)(((size_t)c * T + t) * sizeof(float)) : OOBW;
            m[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, (int)o, 0, 0));
            r[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, (int)o, 0, 0));
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = bf2f(xq[i]);
        if constexpr (MASK) v = 1.f / (1.f + __expf(-v));
        else if (post == 1) v = tanhf(v);
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
    float l1acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = ty + 8 * i, c = c0 + j, t = t0 + tx;
        if (c < C && t < T) {
            const size_t o = clip + (size_t)c * T + t;
            const float v = MASK ? tile[tx][j] * m[i] : tile[tx][j];
            out[o] = v;
            if (l1) l1acc += fabsf(v - r[i]);
        }
    }
    if (MASK && l1_part) {
        double d = (double)l1acc;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) d += __shfl_xor(d, m, 64);
        __shared__ double red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0)
            l1_part[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient:  gw[j][co][ci] = sum_r g[r][co] * xa[r + off_j][ci]   (fp32, one partial slab per row range)
//                   gbias[co]     = sum_r g[r][co] ;  g_out[r][co] = g (optional materialisation, for the residual)
// The reduction runs over ROWS, the non-contiguous dimension of both CL operands, so tiles are transposed on
// their way into LDS (T[c][row], row-contiguous) and MFMA fragments are read with ds_read_b128.  All taps
// of a group share the staged g tile; each tap stages its own row-shifted copy of xa.
// ---------------------------------------------------------------------------------------------------------
constexpr int WKT = 3;     // taps per pass (accumulators: WKT x 16 VGPRs); the register ring must leave 2 workgroups per CU
struct WgradParams {
    const bf16_t *G1, *G2, *GM;   // g = G1 + G2 * leaky'(GM)
    const bf16_t *xa;
    float *gw, *gbias;
    bf16_t *g_out;
    long long R;
    int Ca, Cb, k, off0, dstep, rows_per_split;
    float g2_slope;
    // up_u > 0: the `xa` operand is the high-resolution view of a transposed conv (ConvParams, up_role 2): Ca = up_u * up_Cr
    int up_u, up_p, up_Lp, up_HP, up_LpO, up_HPO, up_Cr;
#ifdef PSND_TRACE
    long long *trace;
#endif
};
#ifdef PSND_TRACE
#define PSND_WSTAMP(i_)                                                                                         \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && p.trace)                                                                 \
            p.trace[(tblk * 4 + (threadIdx.x >> 6)) * 8 + (i_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PSND_WSTAMP(i_)
#endif

// Each workgroup owns one 64 (co) x 64 (ci) tile of all k taps over ONE range of rows and writes its partial sums
// as a plain slab gw_part[split][j][co][ci]; the weight-norm backward kernel adds the slabs up.  (The first
// version accumulated with fp32 atomics: 6.3 M of them per launch at the config-2 shape = 65 % of a
// workgroup's 52 k cycles, the L2 retiring about one fp32 atomic per channel per clock.)
// Row chunks (32 rows) are fetched WD chunks ahead as raw pieces (the gradient combine is done at staging).
// Transposed staging: lanes rr and rr^1 swap halves (DPP) so every lane writes 4-byte (2 rows x 1 channel) words.
#ifndef PSND_WD
#define PSND_WD 2
#endif
constexpr int WD = PSND_WD;
#ifndef PSND_WGRAD_WAVES
#define PSND_WGRAD_WAVES 2
#endif
constexpr int kWgradLdsBytes = 2 * (1 + WKT) * 64 * RS * (int)sizeof(bf16_t);
__device__ __forceinline__ void conv_wgrad_body_v1(const WgradParams &p, const int bx, const int by, const int bz, bf16_t *sT, const size_t tblk) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kg = lane >> 5;
    const int co0 = bx * 64, ci0 = by * 64;
    const int ntg = (p.k + WKT - 1) / WKT;               // tap groups: one per workgroup (bz = split * ntg + group)
    const int split = bz / ntg, tgrp = bz - split * ntg;
    const long long rs = (long long)split * p.rows_per_split;
    const long long re = min(rs + p.rows_per_split, p.R);
    const int rr = tid & 31, cg = tid >> 5;          // staging identity: row rr of the chunk, channels 8 cg .. 8 cg + 7
    const bool do_bias = (by == 0) && tgrp == 0;
    const bool comb = p.G2 != nullptr;
    const bool gok = co0 + 8 * cg < p.Cb, xok = ci0 + 8 * cg < p.Ca;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    PSND_WSTAMP(0);
    constexpr unsigned OOB = 0xffffffffu;
    const int g_bytes = (int)((size_t)p.R * p.Cb * sizeof(bf16_t));
    const int x_bytes = p.up_u > 0 ? (int)((size_t)(p.R / p.up_Lp) * p.up_LpO * p.up_Cr * sizeof(bf16_t)) : (int)((size_t)p.R * p.Ca * sizeof(bf16_t));
    const bool haveG1 = p.G1 != nullptr;
    const __amdgpu_buffer_rsrc_t rG1 = make_uniform_rsrc(haveG1 ? p.G1 : p.G2, g_bytes);
    const __amdgpu_buffer_rsrc_t rG2 = make_uniform_rsrc(comb ? p.G2 : p.G1, g_bytes);
    const __amdgpu_buffer_rsrc_t rGM = make_uniform_rsrc(comb ? p.GM : p.G1, g_bytes);
    const __amdgpu_buffer_rsrc_t rX = make_uniform_rsrc(p.xa, x_bytes);
    auto ld16 = [&](__amdgpu_buffer_rsrc_t r, unsigned off) __attribute__((always_inline)) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return __builtin_bit_cast(uint4, v);
    };

    // (own, neighbour) dwords -> one dword holding rows (rr & ~1, rr | 1) of channel 2*i + (rr & 1)
    auto pair_rows = [&](unsigned own) __attribute__((always_inline)) {
        const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        // v_perm_b32(hi, lo, sel): byte i of the result = byte sel[i] of {hi, lo} (0-3 = lo, 4-7 = hi)
        // even lane: (own.lo16 | nb.lo16 << 16) = channel 2i, rows (rr, rr+1); odd: (nb.hi16 | own.hi16 << 16) = channel 2i+1
        return (rr & 1) ? __builtin_amdgcn_perm(own, nb, 0x07060302u) : __builtin_amdgcn_perm(nb, own, 0x05040100u);
    };

    {
        const int t0 = tgrp * WKT;
        const int nt = min(WKT, p.k - t0);
        f32x16 acc[WKT];
#pragma unroll
        for (int j = 0; j < WKT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        uint4 vg[WD], vg2[WD], vgm[WD], vx[WD][WKT];
        // branch-free buffer loads (offset OOB -> zeros), see conv_cl_kernel
        auto fetch = [&](auto sc, long long r0) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const long long r = r0 + rr;
            const bool rok = r < re;
            const unsigned og = (rok && gok) ? (unsigned)(((size_t)r * p.Cb + co0 + 8 * cg) * sizeof(bf16_t)) : OOB;
            vg[s] = ld16(rG1, haveG1 ? og : OOB);
            vg2[s] = ld16(rG2, comb ? og : OOB);
            vgm[s] = ld16(rGM, comb ? og : OOB);
#pragma unroll
            for (int j = 0; j < WKT; ++j) {
                const long long rx = r + p.off0 + (t0 + j) * p.dstep;
                bool ok = j < nt && rok && rx >= 0 && rx < p.R && xok;
                size_t xo = (size_t)rx * p.Ca;
                if (p.up_u > 0) {                       // uniform
                    const long long hb = ok ? up_row_base(p.up_Lp, p.up_HP, p.up_u, p.up_p, p.up_LpO, p.up_HPO, rx) : -1;
                    ok = hb >= 0;
                    xo = (size_t)hb * p.up_Cr;
                }
                vx[s][j] = ld16(rX, ok ? (unsigned)((xo + ci0 + 8 * cg) * sizeof(bf16_t)) : OOB);
            }
        };
        auto stage = [&](auto sc, long long r0, bf16_t *sA, bf16_t *sB) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            uint4 g = vg[s];
            if (comb) {
                const unsigned *pv = reinterpret_cast<const unsigned *>(&vg[s]), *pg = reinterpret_cast<const unsigned *>(&vg2[s]),
                               *pm = reinterpret_cast<const unsigned *>(&vgm[s]);
                unsigned out[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a0 = bf2f((bf16_t)(pv[i] & 0xffff)), a1 = bf2f((bf16_t)(pv[i] >> 16));
                    const float b0 = bf2f((bf16_t)(pg[i] & 0xffff)), b1 = bf2f((bf16_t)(pg[i] >> 16));
                    const float m0 = bf2f((bf16_t)(pm[i] & 0xffff)), m1 = bf2f((bf16_t)(pm[i] >> 16));
                    out[i] = pack_bf16(a0 + b0 * (m0 > 0.f ? 1.f : p.g2_slope), a1 + b1 * (m1 > 0.f ? 1.f : p.g2_slope));
                }
                g = make_uint4(out[0], out[1], out[2], out[3]);
            }
            const unsigned *pg4 = reinterpret_cast<const unsigned *>(&g);
            if (do_bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += bf2f((bf16_t)((pg4[e >> 1] >> (16 * (e & 1))) & 0xffff));
                const long long r = r0 + rr;
                if (p.g_out && r < re && gok) *reinterpret_cast<uint4 *>(p.g_out + (size_t)r * p.Cb + co0 + 8 * cg) = g;
            }
            const int r2 = rr & ~1, par = rr & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)   // channel 8cg + 2i + par, rows r2, r2+1
                *reinterpret_cast<unsigned *>(sA + (8 * cg + 2 * i + par) * RS + r2) = pair_rows(pg4[i]);
#pragma unroll
            for (int j = 0; j < WKT; ++j)
                if (j < nt) {
                    const unsigned *px = reinterpret_cast<const unsigned *>(&vx[s][j]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<unsigned *>(sB + (j * 64 + 8 * cg + 2 * i + par) * RS + r2) = pair_rows(px[i]);
                }
        };
        // whole ring turns: chunks past the row range load and multiply zeros (at most WD - 1 of them)
        const int nchunk = ((int)((re - rs + 31) / 32) + WD - 1) / WD * WD;
        static_for<0, WD>([&](auto sc) __attribute__((always_inline)) { fetch(sc, rs + 32ll * decltype(sc)::value); });
        for (int c = 0; c < nchunk; c += WD) {
            static_for<0, WD>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                const int ch = c + s;
                {
                    bf16_t *sA = sT + (ch & 1) * (1 + WKT) * 64 * RS, *sB = sA + 64 * RS;
                    stage(sc, rs + 32ll * ch, sA, sB);
                    if (ch == 0) PSND_WSTAMP(1);
                    __syncthreads();
                    fetch(sc, rs + 32ll * (ch + WD));
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(sA + (wm * 32 + li) * RS + 16 * kk + 8 * kg);
#pragma unroll
                        for (int j = 0; j < WKT; ++j)
                            if (j < nt) {
                                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(sB + (j * 64 + wn * 32 + li) * RS + 16 * kk + 8 * kg);
                                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                            }
                    }
                }
            });
        }
        __syncthreads();                      // a second tap group restarts in LDS buffer 0
        PSND_WSTAMP(2);
        // D[i = co][j = ci]: col = lane & 31 -> ci, rows -> co.  128-B runs per (co, tap): plain stores.
        const int ci = ci0 + wn * 32 + li;
        if (ci < p.Ca) {
#pragma unroll
            for (int j = 0; j < WKT; ++j)
                if (j < nt) {
                    float *dst = p.gw + (((size_t)split * p.k + t0 + j) * p.Cb) * p.Ca + ci;
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int co = co0 + wm * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kg;
                        if (co < p.Cb) dst[(size_t)co * p.Ca] = acc[j][rg];
                    }
                }
        }
    }
    PSND_WSTAMP(3);
#ifdef PSND_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    PSND_WSTAMP(4);
#endif
    if (do_bias && p.gbias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (rr == 0 && co0 + 8 * cg + e < p.Cb) p.gbias[(size_t)split * p.Cb + co0 + 8 * cg + e] = v;
        }
    }
}
// ---- round 2: the same reduction on the gfx950 transposing LDS read ---------------------------------------------------------
// Both MFMA operands of the weight gradient need the REDUCTION dimension (rows) along the fragment, while the tensors are row-major
// (channels contiguous).  Version 1 above transposes while staging: lane pairs swap halves with DPP and write 4-byte words - 16
// ds_write_b32 + 32 VALU per thread and chunk - and since a tap is a row shift, i.e. a 2-byte shift along the transposed axis, it stages
// the activation chunk once PER TAP (3 of its 4 loads, 12 of its 16 writes); its lanes run along the rows, so every load instruction
// touches 64 cache lines (tools/trace_wgrad.py: 2.5 k cycles per 32-row chunk, 1.0 k of it staging, 1.3 k the fetch issue + 6 MFMAs).
// Here the chunk goes into LDS AS IT IS ([row][channel], one 16-byte ds_write per 16-byte load, eight consecutive lanes per row = one
// full line) and `ds_read_b64_tr_b16` hands each lane 4 consecutive ROWS of its channel: two of them make the 8-deep MFMA fragment, and a
// tap is just another row offset of the same tile (staged once, with the halo rows of its tap group).
typedef short v4s_t __attribute__((ext_vector_type(4)));
#ifndef PSND_WGRAD_TR
#define PSND_WGRAD_TR 1
#endif
constexpr int WTP = 96;                      // LDS row pitch of the chunk tiles (bf16): 192 B = 48 banks - the four rows of a transposing
                                             // read start 16 banks apart and the two channel halves 8: conflict-free (144 B: 37 % conflict cycles)
constexpr int WXR = 64;                      // rows of the activation tile buffer: 32 + the row span of a tap group (<= 32)
static_assert(2 * (32 + WXR) * WTP * (int)sizeof(bf16_t) <= kWgradLdsBytes, "the transposing-read tiles fit version 1's LDS");
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t *p0) {   // rows r .. r+3 and r+4 .. r+7 of this lane's channel
    const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3))) *)(p0));
    const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3))) *)(p0 + 4 * WTP));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// NARROW (Ca, Cb <= 32: one 32 x 32 block per tap, three quarters of the 64 x 64 tile would be zeros): the four waves take four TAP
// groups on the same staged tiles - a workgroup covers 12 taps, the rows are fetched once instead of once per tap group, and the
// activation tile holds 32 + the span of all 12 taps (<= WXRN = 96 rows, three pieces per thread).
constexpr int WXRN = 96;
constexpr int kWgradNarrowLdsBytes = 2 * (32 + WXRN) * WTP * (int)sizeof(bf16_t);
// WK: taps per (wave of a) workgroup - 3, or 4 / 6 for the 7- / 11-tap layers of the paired backward (half the tap groups: the rows
// are staged half as often and a split count of twice the size fits the same number of workgroups)
template <bool COMB, bool NARROW = false, int WK = WKT>
__device__ __forceinline__ void conv_wgrad_body(const WgradParams &p, const int bx, const int by, const int bz, bf16_t *sT, const size_t tblk) {
#if !PSND_WGRAD_TR
    conv_wgrad_body_v1(p, bx, by, bz, sT, tblk);
#else
    constexpr int XU = NARROW ? 3 : 2, XR = NARROW ? WXRN : WXR, TPW = NARROW ? 4 * WK : WK;   // x pieces per thread, x tile rows, taps per workgroup
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = NARROW ? 0 : wave >> 1, wn = NARROW ? 0 : wave & 1, li = lane & 31, kg = lane >> 5;
    const int co0 = bx * 64, ci0 = by * 64;
    const int ntg = (p.k + TPW - 1) / TPW;               // tap groups: one per workgroup (bz = split * ntg + group)
    const int split = bz / ntg, tgrp = bz - split * ntg;
    const long long rs = (long long)split * p.rows_per_split;
    const long long re = min(rs + p.rows_per_split, p.R);
    const int rr = tid >> 3, cg = tid & 7;               // staging identity: row rr of the chunk, channels 8 cg .. 8 cg + 7
    const bool do_bias = (by == 0) && tgrp == 0;
    constexpr bool comb = COMB;                          // the two extra loads per chunk of the combine are compiled out otherwise
    const bool gok = co0 + 8 * cg < p.Cb, xok = ci0 + 8 * cg < p.Ca;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    PSND_WSTAMP(0);
    constexpr unsigned OOB = 0xffffffffu;
    const int g_bytes = (int)((size_t)p.R * p.Cb * sizeof(bf16_t));
    const int x_bytes = p.up_u > 0 ? (int)((size_t)(p.R / p.up_Lp) * p.up_LpO * p.up_Cr * sizeof(bf16_t)) : (int)((size_t)p.R * p.Ca * sizeof(bf16_t));
    const bool haveG1 = p.G1 != nullptr;
    const __amdgpu_buffer_rsrc_t rG1 = make_uniform_rsrc(haveG1 ? p.G1 : p.G2, g_bytes);
    const __amdgpu_buffer_rsrc_t rG2 = make_uniform_rsrc(comb ? p.G2 : p.G1, g_bytes);
    const __amdgpu_buffer_rsrc_t rGM = make_uniform_rsrc(comb ? p.GM : p.G1, g_bytes);
    const __amdgpu_buffer_rsrc_t rX = make_uniform_rsrc(p.xa, x_bytes);
    auto ld16 = [&](__amdgpu_buffer_rsrc_t r, unsigned off) __attribute__((always_inline)) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return __builtin_bit_cast(uint4, v);
    };
    const int tw0 = tgrp * TPW, ntw = min(TPW, p.k - tw0);      // the workgroup's taps
    const int t0 = NARROW ? tw0 + wave * WK : tw0;             // this wave's taps (NARROW: a wave past k has none)
    const int nt = max(0, min(WK, p.k - t0));
    int lo = p.off0 + tw0 * p.dstep, hi = lo;            // row offsets of the workgroup's taps: the activation tile covers [lo, 32 + hi)
    for (int j = 1; j < ntw; ++j) {
        const int o = p.off0 + (tw0 + j) * p.dstep;
        lo = min(lo, o), hi = max(hi, o);
    }
    const int xrows = 32 + hi - lo;                      // <= XR (the launchers check the dilation)
    int toff[WK];                                       // tile row offset of tap j (taps past nt: the group's first tap's)
#pragma unroll
    for (int j = 0; j < WK; ++j) toff[j] = p.off0 + (j < nt ? t0 + j : (nt > 0 ? t0 : tw0)) * p.dstep - lo;
    f32x16 acc[WK];
#pragma unroll
    for (int j = 0; j < WK; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    uint4 vg[WD], vg2[COMB ? WD : 1], vgm[COMB ? WD : 1], vx[WD][XU];
    // branch-free buffer loads (offset OOB -> zeros), see conv_cl_kernel
    auto fetch = [&](auto sc, long long r0) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const long long r = r0 + rr;
        const unsigned og = (r < re && gok) ? (unsigned)(((size_t)r * p.Cb + co0 + 8 * cg) * sizeof(bf16_t)) : OOB;
        vg[s] = ld16(rG1, haveG1 ? og : OOB);
        if constexpr (COMB) {
            vg2[s] = ld16(rG2, og);
            vgm[s] = ld16(rGM, og);
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int xi = rr + 32 * u;
            const long long rx = r0 + lo + xi;
            bool ok = xi < xrows && rx >= 0 && rx < p.R && xok;
            size_t xo = (size_t)rx * p.Ca;
            if (p.up_u > 0) {                       // uniform
                const long long hb = ok ? up_row_base(p.up_Lp, p.up_HP, p.up_u, p.up_p, p.up_LpO, p.up_HPO, rx) : -1;
                ok = hb >= 0;
                xo = (size_t)hb * p.up_Cr;
            }
            vx[s][u] = ld16(rX, ok ? (unsigned)((xo + ci0 + 8 * cg) * sizeof(bf16_t)) : OOB);
        }
    };
    auto stage = [&](auto sc, long long r0, bf16_t *sG, bf16_t *sX) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        uint4 g = vg[s];
        if constexpr (COMB) {
            const unsigned *pv = reinterpret_cast<const unsigned *>(&vg[s]), *pg = reinterpret_cast<const unsigned *>(&vg2[s]),
                           *pm = reinterpret_cast<const unsigned *>(&vgm[s]);
            unsigned out[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a0 = bf2f((bf16_t)(pv[i] & 0xffff)), a1 = bf2f((bf16_t)(pv[i] >> 16));
                const float b0 = bf2f((bf16_t)(pg[i] & 0xffff)), b1 = bf2f((bf16_t)(pg[i] >> 16));
                const float m0 = bf2f((bf16_t)(pm[i] & 0xffff)), m1 = bf2f((bf16_t)(pm[i] >> 16));
                out[i] = pack_bf16(a0 + b0 * (m0 > 0.f ? 1.f : p.g2_slope), a1 + b1 * (m1 > 0.f ? 1.f : p.g2_slope));
            }
            g = make_uint4(out[0], out[1], out[2], out[3]);
        }
        if (do_bias) {
            const unsigned *pg4 = reinterpret_cast<const unsigned *>(&g);
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += bf2f((bf16_t)((pg4[e >> 1] >> (16 * (e & 1))) & 0xffff));
            const long long r = r0 + rr;
            if (p.g_out && r < re && gok) *reinterpret_cast<uint4 *>(p.g_out + (size_t)r * p.Cb + co0 + 8 * cg) = g;
        }
        *reinterpret_cast<uint4 *>(sG + rr * WTP + 8 * cg) = g;
        *reinterpret_cast<uint4 *>(sX + rr * WTP + 8 * cg) = vx[s][0];
        if (rr + 32 < xrows) *reinterpret_cast<uint4 *>(sX + (rr + 32) * WTP + 8 * cg) = vx[s][1];
        if constexpr (NARROW)
            if (rr + 64 < xrows) *reinterpret_cast<uint4 *>(sX + (rr + 64) * WTP + 8 * cg) = vx[s][2];
    };
    // this lane's piece of a transposing read: row (lane & 15) >> 2 of the 4-row block, channels 16 ((lane >> 4) & 1) + 4 (lane & 3) ..
    const int trow = 8 * kg + ((lane & 15) >> 2), tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    // whole ring turns: chunks past the row range load and multiply zeros (at most WD - 1 of them)
    const int nchunk = ((int)((re - rs + 31) / 32) + WD - 1) / WD * WD;
    static_for<0, WD>([&](auto sc) __attribute__((always_inline)) { fetch(sc, rs + 32ll * decltype(sc)::value); });
    for (int c = 0; c < nchunk; c += WD) {
        static_for<0, WD>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int ch = c + s;
            bf16_t *sG = sT + (ch & 1) * (32 + XR) * WTP, *sX = sG + 32 * WTP;
            if (ch == 4) PSND_WSTAMP(5);
            stage(sc, rs + 32ll * ch, sG, sX);
            if (ch == 0) PSND_WSTAMP(1);
            if (ch == 4) PSND_WSTAMP(6);
            __syncthreads();
            if (ch == 4) PSND_WSTAMP(7);
            fetch(sc, rs + 32ll * (ch + WD));
            // all sixteen transposing reads first, then the six MFMAs, in ONE basic block: taps past nt read tap 0's rows again and
            // accumulate into registers nobody stores (a branch per tap made every MFMA wait for its own two reads: 6 x ~160 cycles)
            bf16x8 fa[2], fb[2][WK];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                fa[kk] = tr_frag(sG + (16 * kk + trow) * WTP + wm * 32 + tcol);
#pragma unroll
                for (int j = 0; j < WK; ++j) fb[kk][j] = tr_frag(sX + (16 * kk + trow + toff[j]) * WTP + wn * 32 + tcol);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < WK; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk], fb[kk][j], acc[j], 0, 0, 0);
        });
    }
    __syncthreads();
    PSND_WSTAMP(2);
    // D[i = co][j = ci]: lane & 31 -> ci, registers -> co.  Through LDS (the chunk tiles are dead), so that a thread stores four consecutive
    // ci: 12 16-byte stores per thread instead of 48 4-byte ones (the store tail is issue-bound: 3.7 k cycles of 26 k).
    if constexpr (NARROW) {
        // every wave holds the 32 x 32 blocks of ITS taps: four blocks per pass j go through LDS, a thread stores four consecutive ci
        constexpr int EPN = 36;
        float *sE = reinterpret_cast<float *>(sT);
        static_assert(4 * 32 * EPN * 4 <= kWgradNarrowLdsBytes, "four 32 x 32 blocks fit the chunk buffers");
        const bool vec = (p.Ca % 4 == 0);
#pragma unroll
        for (int j = 0; j < WK; ++j) {
            if (j > 0) __syncthreads();
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) sE[(wave * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kg) * EPN + li] = acc[j][rg];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + 256 * u, w = idx >> 8, row = (idx >> 3) & 31, c4 = 4 * (idx & 7);
                const int tap = tw0 + w * WK + j;
                const int co = co0 + row, ci = ci0 + c4;
                if (tap >= p.k || co >= p.Cb || ci >= p.Ca) continue;
                float *dst = p.gw + (((size_t)split * p.k + tap) * p.Cb) * p.Ca;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sE + (w * 32 + row) * EPN + c4);
                if (vec && ci + 3 < p.Ca) {
                    *reinterpret_cast<f32x4 *>(dst + (size_t)co * p.Ca + ci) = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
                    for (int q = 0; q < 4 && ci + q < p.Ca; ++q) dst[(size_t)co * p.Ca + ci + q] = e[q];
                }
            }
        }
    } else {
        constexpr int EP = 68;                                     // fp32 row pitch of the 64 x 64 tile of one tap
        float *sE = reinterpret_cast<float *>(sT);
        static_assert(64 * EP * 4 <= kWgradLdsBytes, "one tap's slab tile fits the chunk buffers");
        const bool vec = (p.Ca % 4 == 0);
#pragma unroll
        for (int j = 0; j < WK; ++j)
            if (j < nt) {
                if (j > 0) __syncthreads();
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) sE[(wm * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kg) * EP + wn * 32 + li] = acc[j][rg];
                __syncthreads();
                float *dst = p.gw + (((size_t)split * p.k + t0 + j) * p.Cb) * p.Ca;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = tid + 256 * u, row = idx >> 4, c4 = 4 * (idx & 15);
                    const int co = co0 + row, ci = ci0 + c4;
                    if (co >= p.Cb || ci >= p.Ca) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(sE + row * EP + c4);
                    if (vec && ci + 3 < p.Ca) {
                        *reinterpret_cast<f32x4 *>(dst + (size_t)co * p.Ca + ci) = v;
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
                        for (int q = 0; q < 4 && ci + q < p.Ca; ++q) dst[(size_t)co * p.Ca + ci + q] = e[q];
                    }
                }
            }
    }
    __syncthreads();                                               // the bias partial sums below re-use the same LDS
    PSND_WSTAMP(3);
#ifdef PSND_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    PSND_WSTAMP(4);
#endif
    if (do_bias && p.gbias) {                // uniform per workgroup.  A channel group's rows sit in all four waves: through LDS
        float *sb = reinterpret_cast<float *>(sT);       // [4 waves][64 channels]; the tiles are no longer read (barrier above)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
#pragma unroll
            for (int m = 32; m >= 8; m >>= 1) v += __shfl_xor(v, m, 64);
            if ((lane >> 3) == 0) sb[wave * 64 + 8 * cg + e] = v;
        }
        __syncthreads();
        if (tid < 64 && co0 + tid < p.Cb) p.gbias[(size_t)split * p.Cb + co0 + tid] = (sb[tid] + sb[64 + tid]) + (sb[128 + tid] + sb[192 + tid]);
    }
#endif
}
__global__ __launch_bounds__(256, PSND_WGRAD_WAVES) void conv_wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_dyn[];
    if (p.G2) conv_wgrad_body<true>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem_dyn, ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    else conv_wgrad_body<false>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem_dyn, ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// The two independent kernels of a conv's backward - weight gradient (partial slabs) and input gradient (with the on-load
// gradient combine) - as ONE launch: each alone fills part of the chip with 10-20 us latency chains (256 + 384 workgroups
// at the config-2 shape, 2 per CU), both read the same incoming gradient.  Workgroups [0, nw) take the weight-gradient
// role (the longer chain goes first), the rest the input-gradient role; registers and LDS are the maximum of the two.
// taps per weight-gradient workgroup of the paired backward instance <KT, HMX>: the 7- / 11- / 16-tap instances (reach <= 25) take
// 4 / 6 / 6 taps per pass - two tap groups instead of three / four / six
constexpr int pair_wk(int KT, int HMX) { return HMX != 25 ? WKT : (KT >= 9 ? 6 : (KT >= 5 ? 4 : WKT)); }
// WN: the weight-gradient role of a 32 -> 32 channel layer (conv_wgrad_body<.., NARROW>: 12 taps per workgroup, one per 3 per wave)
template <int KT, int D, int NBUF, bool COMBINE, int MT, int HMX = 25, int KCT = 32, int WNC = 2, bool WN = false>
__global__ __launch_bounds__(256, (D == 1 ? 3 : PSND_CONV_OCC)) void conv_bwd_pair_kernel(ConvParams pc, WgradParams pw, int nw, int wgx, int wgy, int cgx) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_dyn[];
    const int b = blockIdx.x;
    if (b < nw) {
        const int bx = b % wgx, r = b / wgx;
        conv_wgrad_body<COMBINE, WN, ((WN || D == 1) ? WKT : pair_wk(KT, HMX))>(pw, bx, r % wgy, r / wgy, smem_dyn, 0);   // D == 1: the 168-VGPR instances
    } else {
        const int c = b - nw;
        conv_cl_body<KT, D, COMBINE, NBUF, MT, HMX, false, KCT, WNC>(pc, c % cgx, c / cgx, smem_dyn, 0);
    }
}


// Backward of a whole residual pair as ONE launch (round 2): the input-gradient chain of both convs is the two-chained-convs body of
// psnd_conv_pair.h (256 threads here: the weight-gradient workgroups supply the other waves of a SIMD), next to TWO weight-gradient
// roles - the pair's second conv (its gradient is this launch's input) and a conv whose gradient an EARLIER launch produced (the first
// conv of the pair handled before: its gradient is that launch's `mid`).
#ifndef PSND_PAIRBWD_MR
#define PSND_PAIRBWD_MR 1
#endif
#ifndef PSND_PAIRBWD_RU
#define PSND_PAIRBWD_RU 6
#endif
__global__ __launch_bounds__(256, 2) void conv_pair_bwd_kernel(pairk::PairParams pp, WgradParams wa, WgradParams wb_, int nwa, int nwb, int wgx,
                                                               int wgy) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_dyn[];
    const int b = blockIdx.x;
    if (b < nwa) {
        const int bx = b % wgx, r = b / wgx;
        conv_wgrad_body<false>(wa, bx, r % wgy, r / wgy, smem_dyn, 0);
    } else if (b < nwa + nwb) {
        const int c = b - nwa, bx = c % wgx, r = c / wgx;
        conv_wgrad_body<false>(wb_, bx, r % wgy, r / wgy, smem_dyn, 0);
    } else {
        pairk::conv_pair_body<256, PSND_PAIRBWD_MR, true, 4, PSND_PAIRBWD_RU>(pp, b - nwa - nwb, smem_dyn);
    }
}

// The weight gradients of MANY convs of one shape in one launch (psnd_conv1d_cl_wgrad_multi): workgroup b takes tile / row range
// b % per_conv of conv b / per_conv.  For the batched backward of a conv chain (cl.py): the input-gradient chain runs first, alone,
// and every weight gradient it left behind is computed here at full occupancy instead of riding along in 12 latency-bound launches.
constexpr int WGRAD_MULTI_MAX = 32;
struct WgradMultiArgs {
    WgradParams base;
    int n;
    int blk0[WGRAD_MULTI_MAX + 1];      // first workgroup of conv i
    struct {
        const bf16_t *g, *xa;
        float *gw, *gb;
        int off0, dstep, Ca, Cb, k, rps, wgx, wgy;
    } c[WGRAD_MULTI_MAX];
};
__global__ __launch_bounds__(256, 2) void conv_wgrad_multi_kernel(WgradMultiArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem_dyn[];
    int c = 0;
    while (c + 1 < a.n && (int)blockIdx.x >= a.blk0[c + 1]) ++c;          // uniform
    const int b = (int)blockIdx.x - a.blk0[c];
    WgradParams w = a.base;
    w.G1 = a.c[c].g, w.xa = a.c[c].xa, w.gw = a.c[c].gw, w.gbias = a.c[c].gb, w.off0 = a.c[c].off0, w.dstep = a.c[c].dstep;
    w.Ca = a.c[c].Ca, w.Cb = a.c[c].Cb, w.k = a.c[c].k, w.rows_per_split = a.c[c].rps;
    const int wgx = a.c[c].wgx, wgy = a.c[c].wgy;
    const int bx = b % wgx, r = b / wgx;
    conv_wgrad_body<false>(w, bx, r % wgy, r / wgy, smem_dyn, 0);
}

// ---- weight prep: weight norm (dim 0) + both bf16 packs + padded bias, one block per output channel ------
//   w = g * v / ||v|| ; wf[j][co][ci] (forward), wb[j][ci][co] (input gradient) ; pads are zero-filled by the caller
__global__ __launch_bounds__(256) void conv_prep_kernel(const float *v, const float *g, const float *bias, int Cout, int Cin, int k,
                                                        int Cb, int Ca, bf16_t *wf, bf16_t *wb, float *bp) {
    __shared__ float red[4];
    const int co = blockIdx.x, n = Cin * k;
    const float *vr = v + (size_t)co * n;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) ss = __builtin_fmaf(vr[i], vr[i], ss);
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float scale = g[co] / __builtin_sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ci = i / k, j = i - ci * k;
        const bf16_t w = f2bf(vr[i] * scale);
        wf[pack_index(k, j, co, ci, Cb, Ca)] = w;
        wb[pack_index(k, j, ci, co, Ca, Cb)] = w;
    }
    if (threadIdx.x == 0) bp[co] = bias ? bias[co] : 0.f;
}

// the same for MANY convs in one launch (one descriptor per conv): a model's 26 prep launches of ~5 us each were launch latency, not
// work.  Round 4: one workgroup per EIGHT output channels instead of one per channel.  The per-channel kernel wrote every bf16 of both
// packs on its own (2-byte stores, 64 different lines per store instruction: 11 M line requests for the 5.5 M weights of the separator,
// 35 us, 74 % of the wave time waiting); with eight rows in LDS every store is a whole 16-byte piece of the fragment order - the forward
// pack [j][co][ci] takes 8 consecutive ci of one row (8 rows x 16 B = 128 contiguous bytes per lane octet), the backward pack [j][ci][co]
// the 8 rows of one (ci, j) (consecutive ci = consecutive pieces: 512-byte runs).  The row norms keep the per-channel kernel's summation
// order (256 strided partial sums, xor tree, four wave sums added in order), so the packs are bit-identical to psnd_conv1d_prep's.
// Pads of wf / wb / bp beyond the 8-channel groups are never written: zero them once.
struct PrepDesc {                 // 72 bytes, mirrored by pytorch_sound_amd/cl.py (struct format '<6Q6i')
    const float *v, *g, *bias;
    bf16_t *wf, *wb;
    float *bp;
    int Cout, Cin, k, Cb, Ca, blk0;          // blk0: sum of ceil(Cout / 8) of the records before this one
};
constexpr int PREP_ROWS = 8;
__global__ __launch_bounds__(256) void conv_prep_multi_kernel(const PrepDesc *descs, int n, int which) {
    extern __shared__ __attribute__((aligned(16))) bf16_t s_prep[];          // [8 rows][nn] scaled weights, natural (ci, j) order
    int lo = 0, hi = n - 1;                                  // last descriptor with blk0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const PrepDesc d = descs[lo];
    const int co0 = (blockIdx.x - d.blk0) * PREP_ROWS, nn = d.Cin * d.k;
    if (co0 >= d.Cout) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // ---- norms: wave w owns rows 2 w and 2 w + 1; lane l carries the partial sums of the per-channel kernel's threads l + 64 q ------------
    constexpr int NR = 32;                                   // row values a lane keeps in registers: rows up to 64 * NR = 2048 weights
    if (nn <= 64 * NR) {
        // both rows' values are requested at once and stay in registers: one load latency per workgroup instead of four
        float x[2][NR];
        const float *vr0 = d.v + (size_t)(co0 + 2 * w) * nn;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const bool live = co0 + 2 * w + rr < d.Cout;     // (wave-uniform)
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int i = 64 * q + lane;
                x[rr][q] = (live && i < nn) ? vr0[(size_t)rr * nn + i] : 0.f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * w + rr, co = co0 + r;
            bf16_t *srow = s_prep + (size_t)r * nn;
            float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NR; ++q) ss[q & 3] = __builtin_fmaf(x[rr][q], x[rr][q], ss[q & 3]);     // fused, as the per-channel kernel's loop compiles
#pragma unroll
            for (int q = 0; q < 4; ++q)
                for (int m = 32; m >= 1; m >>= 1) ss[q] += __shfl_xor(ss[q], m, 64);
            const float scale = co < d.Cout ? d.g[co] / __builtin_sqrtf(ss[0] + ss[1] + ss[2] + ss[3]) : 0.f;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int i = 64 * q + lane;
                if (i < nn) srow[i] = f2bf(x[rr][q] * scale);
            }
            if (lane == 0 && (which & 1) && co < d.Cout) d.bp[co] = d.bias ? d.bias[co] : 0.f;
        }
    } else {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * w + rr, co = co0 + r;
        bf16_t *srow = s_prep + (size_t)r * nn;
        if (co >= d.Cout) {                                  // (wave-uniform) rows past the conv: zeros
            for (int i = lane; i < nn; i += 64) srow[i] = 0;
            continue;
        }
        const float *vr = d.v + (size_t)co * nn;
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = 0; i0 < nn; i0 += 256) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 64 * q + lane;
                const float x = i < nn ? vr[i] : 0.f;
                ss[q] = __builtin_fmaf(x, x, ss[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            for (int m = 32; m >= 1; m >>= 1) ss[q] += __shfl_xor(ss[q], m, 64);
        const float scale = d.g[co] / __builtin_sqrtf(ss[0] + ss[1] + ss[2] + ss[3]);
        for (int i = lane; i < nn; i += 64) srow[i] = f2bf(vr[i] * scale);        // (second read: L1 / L2)
        if (lane == 0 && (which & 1)) d.bp[co] = d.bias ? d.bias[co] : 0.f;
    }
    }
    __syncthreads();
    const int k = d.k, Ca8 = d.Ca >> 3;
    // ---- forward pack: piece (j, row r, ci octet o) = wf[pack_index(j, co0 + r, 8 o)] .. + 8 ------------------------------------------------
    if (which & 1)
    for (int q = tid; q < k * Ca8 * PREP_ROWS; q += 256) {
        const int r = q & (PREP_ROWS - 1), rest = q >> 3;
        const int o = rest % Ca8, j = rest / Ca8;
        const bf16_t *srow = s_prep + (size_t)r * nn + j;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 8 * o + e;
            v[e] = ci < d.Cin ? (short)srow[ci * k] : (short)0;
        }
        *reinterpret_cast<bf16x8 *>(d.wf + pack_index(k, j, co0 + r, 8 * o, d.Cb, d.Ca)) = v;
    }
    // ---- backward pack: piece (j, ci) = wb[pack_index(j, ci, co0)] .. + 8: the eight rows of one (ci, j) ------------------------------------
    if (which & 2)
    for (int q = tid; q < k * d.Cin; q += 256) {
        const int ci = q % d.Cin, j = q / d.Cin;
        const bf16_t *sp = s_prep + ci * k + j;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (short)sp[(size_t)e * nn];
        *reinterpret_cast<bf16x8 *>(d.wb + pack_index(k, j, ci, co0, d.Ca, d.Cb)) = v;
    }
}

// ---- weight-norm backward: from gw[j][co][ci] (fp32) to g_v (Cout,Cin,k) and g_g (Cout) ------------------
//   vhat = v/||v||, d = sum(gw * vhat), g_g = d, g_v = (g/||v||) * (gw - vhat * d)
template <bool MANY>   // MANY: more than 16 slabs (narrow layers over long clips) - batches of 16; else one straight batch
__device__ __forceinline__ void conv_finish_body(const float *gw_part, const float *gb_part, int splits, const float *v,
                                                 const float *g, int Cin, int k, int Cb, int Ca, float *gv,
                                                 float *gg, float *gbias, unsigned kmagic, unsigned cmagic, const int co, float *s_gw,
                                                 float *red) {
    // one workgroup (256 or 1024 threads) per output channel: sums the wgrad slabs (coalesced over ci), then the weight-norm
    // backward.  s_gw holds the summed gradients as [j][Cin + 1] (the +1 keeps the k-strided reads of the second loop on
    // different banks); i / k by multiplication with kmagic = ceil(2^32 / k) (exact for i < 2^32 / k).
    const int n = Cin * k, BD = blockDim.x, tid = threadIdx.x;
    const int pitch = Cin + 1;
    if constexpr (!MANY) {
        // Lean form for 4-aligned layers (Cin, Ca multiples of 4: every conv of the separator's body): 16-byte loads of the slabs, of v and
        // 16-byte stores of g_v, one index division per four elements.  The element-wise form below spends ~480 VALU and ~600 scalar
        // instructions per wave (magic divisions and a predicated load per slab slot and element) on 12 KB per workgroup: 48 us per launch
        // for the 26 convs of config 2, instruction-bound (profiles/r03_wnorm_pmc.txt).
        if ((Cin & 3) == 0 && (Ca & 3) == 0 && splits <= 16) {
            const int cin4 = Cin >> 2, n4 = n >> 2;                      // float4s per tap row / per output channel
            const size_t slab4 = (size_t)k * Cb * Ca;
            const float *vrow = v + (size_t)co * n;
            float bsum = 0.f;
            if (tid < 64 && gbias && gb_part)
                for (int sp = tid; sp < splits; sp += 64) bsum += gb_part[(size_t)sp * Cb + co];
            for (int q = tid; q < n4; q += BD) {                        // (j, ci) order: coalesced slab rows
                const int j = q / cin4, c4 = q - j * cin4;
                const float *src = gw_part + ((size_t)j * Cb + co) * Ca + 4 * c4;
                // the slabs are added in order, but their loads are all in flight together (one load per trip of a `for (sp < splits)`
                // loop waited for the one before: 6-8 dependent L2 round trips per thread, 35 us per launch at config 2, 91-139 us for the
                // 256-channel 11-tap chains of config 3).  Slots past `splits` re-read the last slab and add zero.
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                auto add_slabs = [&](auto nc) __attribute__((always_inline)) {
                    constexpr int NS = decltype(nc)::value;
                    f32x4 t[NS];
#pragma unroll
                    for (int u = 0; u < NS; ++u) t[u] = *reinterpret_cast<const f32x4 *>(src + (size_t)(u < splits ? u : splits - 1) * slab4);
#pragma unroll
                    for (int u = 0; u < NS; ++u) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        a += u < splits ? t[u] : z;
                    }
                };
                if (splits <= 2) add_slabs(std::integral_constant<int, 2>{});             // (uniform)
                else if (splits <= 4) add_slabs(std::integral_constant<int, 4>{});
                else if (splits <= 8) add_slabs(std::integral_constant<int, 8>{});
                else add_slabs(std::integral_constant<int, 16>{});
                float *dst = s_gw + j * pitch + 4 * c4;
                dst[0] = a.x, dst[1] = a.y, dst[2] = a.z, dst[3] = a.w;
            }
            __syncthreads();
            float ss4 = 0.f, dot4 = 0.f;
            for (int q = tid; q < n4; q += BD) {                        // natural (ci, j) order
                const f32x4 vv = *reinterpret_cast<const f32x4 *>(vrow + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e, ci = (int)__umulhi((unsigned)i, kmagic), j = i - ci * k;
                    ss4 = __builtin_fmaf(vv[e], vv[e], ss4);
                    dot4 = __builtin_fmaf(vv[e], s_gw[j * pitch + ci], dot4);
                }
            }
            for (int m = 32; m >= 1; m >>= 1) {
                ss4 += __shfl_xor(ss4, m, 64);
                dot4 += __shfl_xor(dot4, m, 64);
            }
            const int nw4 = BD >> 6;
            if ((tid & 63) == 0) red[tid >> 6] = ss4, red[16 + (tid >> 6)] = dot4;
            __syncthreads();
            float sst = 0.f, dott = 0.f;
            for (int w = 0; w < nw4; ++w) sst += red[w], dott += red[16 + w];
            const float inv = 1.f / __builtin_sqrtf(sst);
            const float dd = dott * inv;
            const float gs = g[co] * inv;
            for (int q = tid; q < n4; q += BD) {
                const f32x4 vv = *reinterpret_cast<const f32x4 *>(vrow + 4 * q);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e, ci = (int)__umulhi((unsigned)i, kmagic), j = i - ci * k;
                    o[e] = gs * (s_gw[j * pitch + ci] - vv[e] * inv * dd);
                }
                *reinterpret_cast<f32x4 *>(gv + (size_t)co * n + 4 * q) = o;
            }
            if (tid == 0) gg[co] = dd;
            if (tid < 64 && gbias && gb_part) {
                float b = bsum;
                for (int m = 32; m >= 1; m >>= 1) b += __shfl_xor(b, m, 64);
                if (tid == 0) gbias[co] = b;
            }
            return;
        }
    }
    const size_t slab = (size_t)k * Cb * Ca;
    const float *vr = v + (size_t)co * n;
    float ss = 0.f, dot = 0.f;
    if constexpr (MANY) {
        // Narrow layers over long clips (HiFi-GAN's 32- and 64-channel stages: n = Cin k of 96 ... 704, 48 ... 192 slabs): with one
        // thread per element the slab sum was a serial chain of `splits` / 16 load batches on n of the workgroup's threads (33-83 us
        // per launch at config 3).  The 1024 threads form NSG = 1024 / EP groups (EP = n rounded up to a power of two); group sg
        // adds slabs sg, sg + NSG, ... (8 loads in flight), the groups' sums meet in LDS in a fixed order.
        int EP = 64;
        while (EP < n && EP < BD) EP <<= 1;
        const int NSG = BD / EP, e = tid & (EP - 1), sg = tid / EP;
        float *s_part = s_gw + pitch * k;                // [NSG][EP], behind the summed gradients
        for (int e0 = 0; e0 < n; e0 += EP) {             // one trip unless n > 1024
            const int ee = e0 + e;
            const bool ok = ee < n;
            const int j = (int)__umulhi((unsigned)ee, cmagic), c = ee - j * Cin;
            const float *src = gw_part + ((size_t)j * Cb + co) * Ca + c;
            float acc = 0.f;
            for (int sp0 = sg; sp0 < splits; sp0 += 8 * NSG) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = (ok && sp0 + u * NSG < splits) ? src[(size_t)(sp0 + u * NSG) * slab] : 0.f;
                acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
            if (NSG > 1) {
                s_part[sg * EP + e] = acc;
                __syncthreads();
                if (sg == 0 && ok) {
                    float a2 = 0.f;
                    for (int q = 0; q < NSG; ++q) a2 += s_part[q * EP + e];
                    s_gw[j * pitch + c] = a2;
                }
                __syncthreads();
            } else if (ok) {
                s_gw[j * pitch + c] = acc;
            }
        }
    } else
    for (int e0 = tid; e0 < n; e0 += 4 * BD) {           // (j, ci) order: coalesced slab reads, 4 x 16 loads in flight
        int jj[4], cc[4];
        float gsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + BD * q;
            jj[q] = (int)__umulhi((unsigned)e, cmagic);
            cc[q] = e - jj[q] * Cin;
        }
        {
            float t[4][16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float *src = gw_part + ((size_t)jj[q] * Cb + co) * Ca + cc[q];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[q][u] = (e0 + BD * q < n && u < splits) ? src[(size_t)u * slab] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                gsum[q] += ((t[q][0] + t[q][1]) + (t[q][2] + t[q][3])) + ((t[q][4] + t[q][5]) + (t[q][6] + t[q][7])) +
                           (((t[q][8] + t[q][9]) + (t[q][10] + t[q][11])) + ((t[q][12] + t[q][13]) + (t[q][14] + t[q][15])));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (e0 + BD * q < n) s_gw[jj[q] * pitch + cc[q]] = gsum[q];
    }
    __syncthreads();
    for (int i = tid; i < n; i += BD) {                  // natural (ci, j) order: coalesced reads of v
        const int ci = (int)__umulhi((unsigned)i, kmagic), j = i - ci * k;
        const float vv = vr[i];
        ss += vv * vv;
        dot += vv * s_gw[j * pitch + ci];
    }
    for (int m = 32; m >= 1; m >>= 1) {
        ss += __shfl_xor(ss, m, 64);
        dot += __shfl_xor(dot, m, 64);
    }
    const int nw = BD >> 6;
    if ((tid & 63) == 0) red[tid >> 6] = ss, red[16 + (tid >> 6)] = dot;
    __syncthreads();
    float sst = 0.f, dott = 0.f;
    for (int w = 0; w < nw; ++w) sst += red[w], dott += red[16 + w];
    const float inv = 1.f / __builtin_sqrtf(sst);
    const float d = dott * inv;                                     // sum(gw * vhat)
    const float gs = g[co] * inv;
    for (int i = tid; i < n; i += BD) {
        const int ci = (int)__umulhi((unsigned)i, kmagic), j = i - ci * k;
        gv[(size_t)co * n + i] = gs * (s_gw[j * pitch + ci] - vr[i] * inv * d);
    }
    if (tid == 0) gg[co] = d;
    if (tid < 64 && gbias && gb_part) {              // the first wave adds up the bias slabs (one serial chain of `splits` loads before)
        float b = 0.f;
        for (int sp = tid; sp < splits; sp += 64) b += gb_part[(size_t)sp * Cb + co];
        for (int m = 32; m >= 1; m >>= 1) b += __shfl_xor(b, m, 64);
        if (tid == 0) gbias[co] = b;
    }
}

template <bool MANY>
__global__ __launch_bounds__(1024) void conv_finish_kernel(const float *gw_part, const float *gb_part, int splits, const float *v,
                                                           const float *g, int Cout, int Cin, int k, int Cb, int Ca, float *gv,
                                                           float *gg, float *gbias, unsigned kmagic, unsigned cmagic) {
    extern __shared__ float s_gw[];
    __shared__ float red[32];
    conv_finish_body<MANY>(gw_part, gb_part, splits, v, g, Cin, k, Cb, Ca, gv, gg, gbias, kmagic, cmagic, blockIdx.x, s_gw, red);
}

// the weight-norm backward of up to PSND_WNORM_MAX convs (a residual block's) in one launch; descriptors travel as kernel
// arguments (no table upload: the launch is capturable in a hipGraph as it is)
struct FinishArgs {
    psnd_wnorm_desc d[PSND_WNORM_MAX];
    unsigned kmagic[PSND_WNORM_MAX], cmagic[PSND_WNORM_MAX];
    int blk0[PSND_WNORM_MAX + 1];
    int n;
};
template <bool MANY>
__global__ __launch_bounds__(1024) void conv_finish_multi_kernel(FinishArgs a) {
    extern __shared__ float s_gw[];
    __shared__ float red[32];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.blk0[i + 1]) ++i;          // uniform
    const psnd_wnorm_desc &d = a.d[i];
    conv_finish_body<MANY>(d.gw_part, d.gbias_part, d.splits, d.v, d.g, d.Cin, d.k, d.Cb, d.Ca, d.gv, d.gg, d.gbias, a.kmagic[i],
                           a.cmagic[i], (int)blockIdx.x - a.blk0[i], s_gw, red);
}

}  // namespace

// launch statistics (which tile instances ran): [0] forward/input-gradient launches with 64-row tiles, [1] with 128-row tiles,
// [2] paired backward launches with 64-row tiles, [3] with 128-row tiles.  Host-side counters, read by the parity tests to
// prove that a test shape reached the instance it is meant to cover.
static std::atomic<long long> g_conv_stats[4];
// the residual-pair launches: [0] pair forward launches with 32-row tiles, [1] with 64-row tiles, [2] pair backward launches (chained
// input gradients + weight-gradient roles), [3] weight-gradient row ranges (splits) of the last pair backward launch
std::atomic<long long> g_conv_pair_stats[4];
extern "C" int psnd_conv_pair_stats(int64_t *out4, int reset) {
    if (out4)
        for (int i = 0; i < 4; ++i) out4[i] = g_conv_pair_stats[i].load();
    if (reset)
        for (int i = 0; i < 4; ++i) g_conv_pair_stats[i].store(0);
    return PSND_OK;
}
extern "C" int psnd_conv_stats(int64_t *out4, int reset) {
    if (out4)
        for (int i = 0; i < 4; ++i) out4[i] = g_conv_stats[i].load();
    if (reset)
        for (int i = 0; i < 4; ++i) g_conv_stats[i].store(0);
    return PSND_OK;
}

// Row tile of a forward / input-gradient launch: 64, 128 or (with3) 192 rows.  At ~200 VGPRs a CU holds 2 workgroups, the chip 512: the
// launch takes ceil(tiles / 512) rounds, and a workgroup's time is a fixed part (first stage in, epilogue out: ~3.9 us) plus ~0.042 us
// per row (fitted on HiFi-GAN's 64-channel stage: 9.3 us at 128 rows, 12 us at 192) - about 100 rows' worth.  Pick the cheapest.
// (Before: 128 rows from 1024 tiles of 64 on - config 2's 256 -> 513 conv ran 828 workgroups of 64 rows in two rounds, HiFi-GAN's
// 64- / 128-channel stages 519 / 526 of 128 rows.)
static int conv_row_tiles(int64_t R, int Cb, bool with3 = false) {
    const char *fe = PSND_ENV("PSND_CONV_MT");              // read per call: the parity tests flip it inside one process
    const int force = fe ? atoi(fe) : 0;
    if (force == 1 || force == 2) return force;
    if (PSND_ENV("PSND_CONV_MT_V1")) return ((R + 63) / 64) * ((Cb + BN - 1) / BN) >= 1024 ? 2 : 1;   // A/B: the rule of before
    const int64_t coltiles = (Cb + BN - 1) / BN;
    int best = 1;
    int64_t best_cost = 0;
    for (int mt = 1; mt <= (with3 ? 3 : 2); ++mt) {
        const int64_t tiles = (R + 64 * mt - 1) / (64 * mt) * coltiles;
        const int64_t cost = (tiles + 511) / 512 * (100 + 64 * mt);
        if (mt == 1 || cost < best_cost) best = mt, best_cost = cost;
    }
    return best;
}

// enqueue one conv_cl_kernel launch for a filled ConvParams (tile instance by size, tap count, operand combine, tap reach)
static int conv_launch(ConvParams &p, hipStream_t st, const char *what) {
    const bool combine = p.A2 != nullptr;
    const int k = p.k, Cb = p.Cb, hm = p.hm;
    if (hm > 40 || (hm > 25 && k > 7)) PSND_FAIL(PSND_E_UNSUPPORTED, "%s: tap reach %d beyond the staged A tile (25; 40 for k <= 7)", what, hm);
    // 192-row tiles exist for the plain instances (no combined operand, reach <= 25, not the transposed conv); PSND_CONV_MT3=0: off
    const char *e3 = PSND_ENV("PSND_CONV_MT3");
    const bool can3 = !combine && p.up_role == 0 && hm <= 25 && Cb > 32 && !(e3 && atoi(e3) == 0);
    int mt = conv_row_tiles(p.R, Cb, can3);
    const bool mt3 = mt == 3;
    if (mt3) mt = 2;
    // narrow layers over long clips (HiFi-GAN's last stage: 32 channels x 131 k rows): 256-row tiles, the four waves along the rows
    // (not with the combined operand: three A rings of 5 pieces per stage spill ~100 VGPRs at 256-row tiles)
    const bool narrow = Cb <= 32 && mt == 2 && p.up_role == 0 && hm <= 25 && !combine && !PSND_ENV("PSND_CONV_NO_NARROW");
    const int bm = narrow ? 256 : (mt3 ? 192 : 64 * mt), bn = narrow ? 32 : BN;
    const int kct = 32;
    size_t lds = 2 * sizeof(bf16_t) * (kct + 8) * (size_t)(bm + 2 * hm);             // two A stage buffers (the weights never enter LDS)
    if (lds < sizeof(float) * bm * (bn + 8)) lds = sizeof(float) * bm * (bn + 8);   // the epilogue's fp32 tile
    if (lds > 160 * 1024) PSND_FAIL(PSND_E_SHAPE, "%s: LDS %zu too large", what, lds);
    dim3 grid((unsigned)((p.R + bm - 1) / bm), (unsigned)((Cb + bn - 1) / bn));
#define PSND_CONV_LAUNCH(KT_, D_, C_, H_, U_)                                                                         \
    do {                                                                                                              \
        auto kern = mt == 2 ? conv_cl_kernel<KT_, (C_ ? 2 : (D_ > 4 ? 4 : D_)), C_, 2, 2, H_, U_> : conv_cl_kernel<KT_, D_, C_, 2, 1, H_, U_>;   \
        if constexpr (!U_ && H_ == 25 && !C_) {                                                                       \
            if (narrow && p.Ca <= 32) kern = conv_cl_kernel<KT_, 1, false, 2, 2, H_, false, 32, 1>;  /* one k-stage: 3 workgroups per CU */ \
            else if (narrow) kern = conv_cl_kernel<KT_, (D_ > 4 ? 4 : D_), false, 2, 2, H_, false, 32, 1>;            \
            else if (mt3) kern = conv_cl_kernel<KT_, (D_ > 4 ? 4 : D_), false, 2, 3, H_, false, 32, 2>;               \
        }                                                                                                             \
        if (lds > 64 * 1024) {                                                                                        \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "%s: set LDS size: %s", what, hipGetErrorString(e));           \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);                                                        \
    } while (0)
    if (p.up_role != 0) {
        if (k > 3 || hm > 25) PSND_FAIL(PSND_E_UNSUPPORTED, "%s: transposed-conv mode is a 2-tap convolution", what);
        if (!combine) PSND_CONV_LAUNCH(3, 8, false, 25, true);
        else PSND_CONV_LAUNCH(3, 4, true, 25, true);
    } else if (hm > 25 && !combine) PSND_CONV_LAUNCH(7, 3, false, 40, false);
    else if (hm > 25) PSND_CONV_LAUNCH(7, 3, true, 40, false);
    else if (k <= 3 && !combine) PSND_CONV_LAUNCH(3, 8, false, 25, false);
    else if (k <= 3) PSND_CONV_LAUNCH(3, 4, true, 25, false);
    else if (k <= 7 && !combine) PSND_CONV_LAUNCH(7, 3, false, 25, false);
    else if (k <= 7) PSND_CONV_LAUNCH(7, 3, true, 25, false);
    else if (k <= 11 && !combine) PSND_CONV_LAUNCH(11, 2, false, 25, false);
    else if (k <= 11) PSND_CONV_LAUNCH(11, 2, true, 25, false);
    else if (!combine) PSND_CONV_LAUNCH(16, 2, false, 25, false);
    else PSND_CONV_LAUNCH(16, 2, true, 25, false);
#undef PSND_CONV_LAUNCH
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) PSND_FAIL(PSND_E_HIP, "%s: %s", what, hipGetErrorString(e_));
    g_conv_stats[mt == 2 ? 1 : 0]++;
    return PSND_OK;
}

static void conv_params_plain(ConvParams &p) {
    p.up_role = 0, p.up_u = 0, p.up_p = 0, p.up_LpO = 0, p.up_HPO = 0, p.up_LO = 0, p.up_Cr = 0;
    p.res_is_a_eff = 0;
#ifdef PSND_TRACE
    p.trace = nullptr;
#endif
}
static void wgrad_params_plain(WgradParams &p) {
    p.up_u = 0, p.up_p = 0, p.up_Lp = 0, p.up_HP = 0, p.up_LpO = 0, p.up_HPO = 0, p.up_Cr = 0;
#ifdef PSND_TRACE
    p.trace = nullptr;
#endif
}

extern "C" int psnd_conv1d_cl(const void *A, const void *A2, const void *AM, float a2_slope, const void *W, const float *bias,
                              const void *res, const void *mask_src, int64_t N, int Lp, int L, int HP, int Ca, int Cb, int k,
                              int off0, int dstep, float act_slope, float mask_slope, void *out_raw, void *out_act,
                              void *a_eff_out, void *stream) {
    if ((!A && !A2) || !W || (!out_raw && !out_act) || (A2 && !AM)) PSND_FAIL(PSND_E_ARG, "conv1d_cl: null pointer");
    if (a_eff_out && !A2) PSND_FAIL(PSND_E_ARG, "conv1d_cl: a_eff_out needs the combined operand (A2, AM)");
    if (Ca % 32 != 0 || Cb % 8 != 0) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl: Ca=%d must be a multiple of 32, Cb=%d of 8", Ca, Cb);
    if (Cb % 32 != 0) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl: Cb=%d must be a multiple of 32 (fragment-ordered weight pack)", Cb);
    if (k < 1 || k > 16 || N < 0 || Lp < L + 2 * HP || L <= 0) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl: k=%d N=%lld Lp=%d L=%d HP=%d", k, (long long)N, Lp, L, HP);
    int hm = 0;
    for (int j = 0; j < k; ++j) {
        const int o = off0 + j * dstep;
        hm = (o < 0 ? -o : o) > hm ? (o < 0 ? -o : o) : hm;
    }
    if (hm > HP) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl: tap reach %d exceeds the halo HP=%d", hm, HP);
    if (N == 0) return PSND_OK;
    if ((size_t)N * Lp * (Ca > Cb ? Ca : Cb) * 2 >= ((size_t)1 << 32) || (size_t)k * Cb * Ca * 2 >= ((size_t)1 << 32))
        PSND_FAIL(PSND_E_SHAPE, "conv1d_cl: operand larger than 4 GB (32-bit buffer offsets)");
    ConvParams p;
    conv_params_plain(p);
    p.A = static_cast<const bf16_t *>(A), p.A2 = static_cast<const bf16_t *>(A2), p.AM = static_cast<const bf16_t *>(AM);
    p.a2_slope = a2_slope;
    p.W = static_cast<const bf16_t *>(W), p.bias = bias;
    p.res = static_cast<const bf16_t *>(res), p.mask_src = static_cast<const bf16_t *>(mask_src);
    p.out_raw = static_cast<bf16_t *>(out_raw), p.out_act = static_cast<bf16_t *>(out_act);
    p.a_eff_out = static_cast<bf16_t *>(a_eff_out);
    p.R = N * (int64_t)Lp, p.Lp = Lp, p.L = L, p.HP = HP, p.Ca = Ca, p.Cb = Cb, p.k = k, p.off0 = off0, p.dstep = dstep, p.hm = hm;
    p.act_slope = act_slope, p.mask_slope = mask_slope;
#ifdef PSND_TRACE
    {
        const char *tp = PSND_ENV("PSND_TRACE_PTR");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
    }
#endif
    return conv_launch(p, static_cast<hipStream_t>(stream), "conv1d_cl");
}

extern "C" int psnd_to_cl(const float *x, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, int preop, void *out,
                          void *stream) {
    if (!x || !out) PSND_FAIL(PSND_E_ARG, "to_cl: null pointer");
    if (preop != 0 && preop != 1) PSND_FAIL(PSND_E_ARG, "to_cl: preop=%d (0: none, 1: log1p)", preop);
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "to_cl: Lp=%d T=%lld HP=%d Cp=%d C=%d", Lp, (long long)T, HP, Cp, C);
    if (N == 0) return PSND_OK;
    dim3 grid((Lp + 31) / 32, (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(to_cl_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, static_cast<bf16_t *>(out),
                       (int)N, C, (int)T, Lp, HP, Cp, preop, nullptr, nullptr);
    PSND_CHECK_LAUNCH("to_cl");
    return PSND_OK;
}

extern "C" int psnd_from_cl(const void *x, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, float *out, void *stream) {
    if (!x || !out) PSND_FAIL(PSND_E_ARG, "from_cl: null pointer");
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "from_cl: bad shape");
    if (N == 0) return PSND_OK;
    dim3 grid((unsigned)((T + 31) / 32), (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(from_cl_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(x), out,
                       (int)N, C, (int)T, Lp, HP, Cp, nullptr);
    PSND_CHECK_LAUNCH("from_cl");
    return PSND_OK;
}

// out = tanh(from_cl(x)) in one pass (the generator's last two operations, hifi_gan.py:134-135) and its backward gx = to_cl(g * (1 - out^2))
extern "C" int psnd_from_cl_tanh(const void *x, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, float *out, void *stream) {
    if (!x || !out) PSND_FAIL(PSND_E_ARG, "from_cl_tanh: null pointer");
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "from_cl_tanh: bad shape");
    if (N == 0) return PSND_OK;
    dim3 grid((unsigned)((T + 31) / 32), (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(from_cl_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(x), out,
                       (int)N, C, (int)T, Lp, HP, Cp, nullptr, nullptr, nullptr, 1);
    PSND_CHECK_LAUNCH("from_cl_tanh");
    return PSND_OK;
}

extern "C" int psnd_to_cl_tanh_bwd(const float *g, const float *out_fwd, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, void *gx, void *stream) {
    if (!g || !out_fwd || !gx) PSND_FAIL(PSND_E_ARG, "to_cl_tanh_bwd: null pointer");
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "to_cl_tanh_bwd: bad shape");
    if (N == 0) return PSND_OK;
    dim3 grid((Lp + 31) / 32, (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(to_cl_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), g, static_cast<bf16_t *>(gx),
                       (int)N, C, (int)T, Lp, HP, Cp, 2, out_fwd, nullptr);
    PSND_CHECK_LAUNCH("to_cl_tanh_bwd");
    return PSND_OK;
}

// mask head of a spectrogram-masking model: est = sigmoid(from_cl(y)) * mag in one pass, and its backward
// gy = to_cl(gest * mag * s * (1 - s)) with s recomputed from y (halo rows / padded channels of gy are written as zeros)
extern "C" int psnd_mask_head_fwd(const void *y, const float *mag, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, float *est,
                                  void *stream) {
    if (!y || !mag || !est) PSND_FAIL(PSND_E_ARG, "mask_head_fwd: null pointer");
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "mask_head_fwd: bad shape");
    if (N == 0) return PSND_OK;
    dim3 grid((unsigned)((T + 31) / 32), (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(from_cl_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(y), est,
                       (int)N, C, (int)T, Lp, HP, Cp, mag);
    PSND_CHECK_LAUNCH("mask_head_fwd");
    return PSND_OK;
}

// the same head with F.l1_loss(est, ref) folded in: forward also leaves sum |est - ref| per workgroup in part[psnd_mask_head_l1_blocks]
// (double; fold with psnd_l1_loss_combine); backward takes the gradient on est from elsewhere (gest, may be NULL) PLUS
// coef * g[0] * sign(est - ref) (g: device scalar = gradient of the loss value, coef = weight / numel)
extern "C" int64_t psnd_mask_head_l1_blocks(int64_t N, int64_t T, int Cp) {
    if (N <= 0 || T <= 0 || Cp <= 0) return 0;
    return N * ((T + 31) / 32) * ((Cp + 31) / 32);
}
extern "C" int psnd_mask_head_l1_fwd(const void *y, const float *mag, const float *ref, int64_t N, int C, int64_t T, int Lp, int HP, int Cp,
                                     float *est, double *part, void *stream) {
    if (!y || !mag || !ref || !est || !part) PSND_FAIL(PSND_E_ARG, "mask_head_l1_fwd: null pointer");
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "mask_head_l1_fwd: bad shape");
    if (N == 0) return PSND_OK;
    dim3 grid((unsigned)((T + 31) / 32), (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(from_cl_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t *>(y), est,
                       (int)N, C, (int)T, Lp, HP, Cp, mag, ref, part);
    PSND_CHECK_LAUNCH("mask_head_l1_fwd");
    return PSND_OK;
}
extern "C" int psnd_mask_head_l1_bwd(const float *gest, const float *mag, const void *y, const float *est, const float *ref, const float *g,
                                     float coef, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, void *gy, void *stream) {
    if (!mag || !y || !est || !ref || !g || !gy) PSND_FAIL(PSND_E_ARG, "mask_head_l1_bwd: null pointer");
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "mask_head_l1_bwd: bad shape");
    if (N == 0) return PSND_OK;
    dim3 grid((Lp + 31) / 32, (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(to_cl_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), gest, static_cast<bf16_t *>(gy), (int)N, C,
                       (int)T, Lp, HP, Cp, 0, mag, static_cast<const bf16_t *>(y), est, ref, g, coef);
    PSND_CHECK_LAUNCH("mask_head_l1_bwd");
    return PSND_OK;
}

extern "C" int psnd_mask_head_bwd(const float *gest, const float *mag, const void *y, int64_t N, int C, int64_t T, int Lp, int HP, int Cp,
                                  void *gy, void *stream) {
    if (!gest || !mag || !y || !gy) PSND_FAIL(PSND_E_ARG, "mask_head_bwd: null pointer");
    if (Lp < T + 2 * HP || Cp < C || N > 65535 || (size_t)C * (size_t)T * 4 >= ((size_t)1 << 31) || (size_t)Lp * Cp * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "mask_head_bwd: bad shape");
    if (N == 0) return PSND_OK;
    dim3 grid((Lp + 31) / 32, (Cp + 31) / 32, (unsigned)N);
    hipLaunchKernelGGL(to_cl_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), gest, static_cast<bf16_t *>(gy), (int)N, C,
                       (int)T, Lp, HP, Cp, 0, mag, static_cast<const bf16_t *>(y));
    PSND_CHECK_LAUNCH("mask_head_bwd");
    return PSND_OK;
}

static int wgrad_splits(int64_t R, int Ca, int Cb, int k, int64_t *rps_out) {
    const bool narrow_w = Ca <= 32 && Cb <= 32 && (R + 63) / 64 >= 1024;          // conv_wgrad_body<.., NARROW>: 12 taps per workgroup
    const int wk = narrow_w ? 4 * WKT : pair_wk(k <= 3 ? 3 : (k <= 7 ? 7 : (k <= 11 ? 11 : 16)), 25);   // taps per workgroup (paired backward)
    const int tx = (Cb + 63) / 64, ty = (Ca + 63) / 64 * ((k + wk - 1) / wk);   // tap groups are workgroups too
    // workgroups of the weight-gradient role.  Measured on the config-2 step with the paired backward launch (256 CUs, 2
    // workgroups each, 368 input-gradient workgroups alongside): 128 -> 1.87 ms, 160 -> 1.74, 192 -> 1.73, 208 -> 1.75,
    // 256 -> 1.90, 320 -> 2.28 (more splits = shorter chains but more slabs for the weight-norm backward to add up, and a
    // second partial wave of workgroups)
    // round 2, transposing-read weight gradient (a 22 k-cycle chain at 16 chunks) next to 128-row input-gradient tiles in the paired
    // launch, config-2 step: 160 -> 0.952 ms, 192 -> 0.948, 208 -> 0.953, 224 -> 0.962, 256 -> 0.965, 288 -> 1.29 (a second wave of
    // workgroups), 128 -> 0.979, 96 -> 1.056; config-3 step 5.31 ms with 192 everywhere against 5.46 with 128 for its short stages
    int64_t target = 192;
    // 32 -> 32 channel layers over long clips run next to the single-stage input-gradient instance at THREE workgroups per CU (768
    // slots, 516 short input-gradient tiles at HiFi-GAN's last stage) and the weight-gradient role is the long pole of that launch
    // (86 chunks of 32 rows per workgroup at 192): config-3 step 4.33 (192) -> 4.24 (252), and on another box 4.13 (252), 4.14 (320),
    // 4.10 (384), 4.08 (512), 4.10 (640)
    if (narrow_w) {
        target = 192;
        if (const char *e = PSND_ENV("PSND_WGRAD_BLOCKS_NARROW")) target = psnd_env_int(e, target, 1, 65535);
    }
    if (const char *e = PSND_ENV("PSND_WGRAD_BLOCKS")) target = psnd_env_int(e, target, 1, 65535);
    int64_t splits = target / ((int64_t)tx * ty);
    if (splits < 1) splits = 1;
    int64_t rps = (R + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    if (rps < 64) rps = 64;
    splits = (R + rps - 1) / rps;
    if (rps_out) *rps_out = rps;
    return (int)splits;
}

extern "C" int psnd_conv1d_cl_wgrad_splits(int64_t N, int Lp, int Ca, int Cb, int k) {
    if (N <= 0 || Lp <= 0 || Ca <= 0 || Cb <= 0 || k <= 0) return 0;
    return wgrad_splits(N * (int64_t)Lp, Ca, Cb, k, nullptr);
}

extern "C" int psnd_conv1d_cl_wgrad(const void *G1, const void *G2, const void *GM, float g2_slope, const void *xa, int64_t N,
                                    int Lp, int Ca, int Cb, int k, int off0, int dstep, float *gw_part, float *gbias_part,
                                    void *g_out, void *stream) {
    if ((!G1 && !G2) || (G2 && !GM) || !xa || !gw_part) PSND_FAIL(PSND_E_ARG, "conv1d_cl_wgrad: null pointer");
    if (Ca % 8 != 0 || Cb % 8 != 0 || k < 1 || k > 16 || N < 0) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_wgrad: Ca=%d Cb=%d k=%d", Ca, Cb, k);
    if ((k < WKT ? k - 1 : WKT - 1) * (dstep < 0 ? -dstep : dstep) > WXR - 32)
        PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_wgrad: dilation %d: a tap group spans more than %d rows", dstep, WXR - 32);
    if (N == 0) return PSND_OK;
    if ((size_t)N * Lp * (Ca > Cb ? Ca : Cb) * 2 >= ((size_t)1 << 32)) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_wgrad: operand larger than 4 GB");
    WgradParams p;
    wgrad_params_plain(p);
    p.G1 = static_cast<const bf16_t *>(G1), p.G2 = static_cast<const bf16_t *>(G2), p.GM = static_cast<const bf16_t *>(GM);
    p.xa = static_cast<const bf16_t *>(xa), p.gw = gw_part, p.gbias = gbias_part, p.g_out = static_cast<bf16_t *>(g_out);
    p.R = N * (int64_t)Lp, p.Ca = Ca, p.Cb = Cb, p.k = k, p.off0 = off0, p.dstep = dstep, p.g2_slope = g2_slope;
    const int tx = (Cb + 63) / 64, ty = (Ca + 63) / 64;
    int64_t rps;
    const int splits = wgrad_splits(p.R, Ca, Cb, k, &rps);
    p.rows_per_split = (int)rps;
#ifdef PSND_TRACE
    {
        const char *tp = PSND_ENV("PSND_TRACE_PTR");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
    }
#endif
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tx, ty, (unsigned)(splits * ((k + WKT - 1) / WKT))), dim3(256), kWgradLdsBytes,
                       static_cast<hipStream_t>(stream), p);
    PSND_CHECK_LAUNCH("conv1d_cl_wgrad");
    return PSND_OK;
}

// Backward of one conv as a single launch (conv_bwd_pair_kernel): input gradient gx = conv(g; transposed pack wb, mirrored
// taps) and the partial weight-gradient slabs, both from g = G1 + G2 * leaky'(GM).  Outside the paired instances (operands beyond the
// 32-bit offsets) the two kernels are enqueued one after the other, same results.
// ---- weight gradients of many convs in one launch (conv_wgrad_multi_kernel) -----------------------------------------------------------
static int wgrad_multi_splits(int64_t R, int Ca, int Cb, int k, int n, int64_t *rps_out) {
    const int tiles = ((Ca + 63) / 64) * ((Cb + 63) / 64) * ((k + WKT - 1) / WKT);
    // 24 convs x 16 tiles at the config-2 size, launch alone (tools/perf_wgrad_multi.py): 1 row range per conv (384 workgroups of 184
    // chunks) 106 us, 2 -> 81 us, 3 -> 100, 4 -> 90, 6 -> 100, 8 -> 108 (and the slabs the weight-norm backward has to add up grow with it)
    int64_t target = 768;
    if (const char *e = PSND_ENV("PSND_WGRAD_MULTI_BLOCKS")) target = psnd_env_int(e, target, 1, 65535);
    int64_t splits = target / ((int64_t)tiles * (n > 0 ? n : 1));
    if (splits < 1) splits = 1;
    int64_t rps = (R + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    if (rps < 64) rps = 64;
    splits = (R + rps - 1) / rps;
    if (rps_out) *rps_out = rps;
    return (int)splits;
}

// row ranges for each of n_convs convs of the launch's MAIN shape (the 256 -> 256 body convs); convs of another shape in the same launch
// (a model's head / tail) take the same number, so that every workgroup walks the same number of row chunks
extern "C" int psnd_conv1d_cl_wgrad_multi_splits(int64_t N, int Lp, int Ca, int Cb, int k, int n_convs) {
    if (N <= 0 || Lp <= 0 || Ca <= 0 || Cb <= 0 || k <= 0 || n_convs <= 0) return 0;
    return wgrad_multi_splits(N * (int64_t)Lp, Ca, Cb, k, n_convs, nullptr);
}

extern "C" int psnd_conv1d_cl_wgrad_multi(const psnd_wgrad_desc *d, int n, int64_t N, int Lp, void *stream) {
    if (!d || n < 0) PSND_FAIL(PSND_E_ARG, "conv1d_cl_wgrad_multi: null pointer");
    if (n == 0 || N == 0) return PSND_OK;
    if (n > WGRAD_MULTI_MAX) PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_wgrad_multi: %d convs (at most %d per launch)", n, WGRAD_MULTI_MAX);
    if (N < 0 || Lp <= 0) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_wgrad_multi: N=%lld Lp=%d", (long long)N, Lp);
    const int64_t R = N * (int64_t)Lp;
    WgradMultiArgs a;
    memset(&a, 0, sizeof(a));
    wgrad_params_plain(a.base);
    a.base.G2 = nullptr, a.base.GM = nullptr, a.base.g_out = nullptr, a.base.R = R, a.base.g2_slope = 1.f;
    a.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const int Ca = d[i].Ca, Cb = d[i].Cb, k = d[i].k, splits = d[i].splits;
        if (!d[i].g || !d[i].xa || !d[i].gw_part) PSND_FAIL(PSND_E_ARG, "conv1d_cl_wgrad_multi: conv %d: null pointer", i);
        if (Ca <= 0 || Cb <= 0 || k <= 0 || k > 16 || Ca % 8 || Cb % 8 || splits < 1)
            PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_wgrad_multi: conv %d: Ca=%d Cb=%d k=%d splits=%d", i, Ca, Cb, k, splits);
        if ((size_t)R * (size_t)(Ca > Cb ? Ca : Cb) * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_wgrad_multi: operand larger than 2 GB");
        if ((k < WKT ? k - 1 : WKT - 1) * (d[i].dstep < 0 ? -d[i].dstep : d[i].dstep) > WXR - 32)
            PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_wgrad_multi: conv %d: dilation %d: a tap group spans more than %d rows", i, d[i].dstep, WXR - 32);
        int64_t rps = (R + splits - 1) / splits;
        rps = (rps + 31) / 32 * 32;
        if (rps < 64) rps = 64;
        if ((R + rps - 1) / rps != splits)
            PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_wgrad_multi: conv %d: %d row ranges do not cut %lld rows (ask psnd_conv1d_cl_wgrad_multi_splits)", i, splits, (long long)R);
        a.c[i].g = static_cast<const bf16_t *>(d[i].g), a.c[i].xa = static_cast<const bf16_t *>(d[i].xa);
        a.c[i].gw = d[i].gw_part, a.c[i].gb = d[i].gbias_part, a.c[i].off0 = d[i].off0, a.c[i].dstep = d[i].dstep;
        a.c[i].Ca = Ca, a.c[i].Cb = Cb, a.c[i].k = k, a.c[i].rps = (int)rps;
        a.c[i].wgx = (Cb + 63) / 64, a.c[i].wgy = (Ca + 63) / 64;
        a.blk0[i] = total;
        total += a.c[i].wgx * a.c[i].wgy * splits * ((k + WKT - 1) / WKT);
    }
    a.blk0[n] = total;
    hipLaunchKernelGGL(conv_wgrad_multi_kernel, dim3((unsigned)total), dim3(256), kWgradLdsBytes, static_cast<hipStream_t>(stream), a);
    PSND_CHECK_LAUNCH("conv1d_cl_wgrad_multi");
    return PSND_OK;
}

// ---- backward of a residual pair in one launch (conv_pair_bwd_kernel) -----------------------------------------------------------------
static int pair_bwd_splits(int64_t R, int C, int k, int64_t *rps_out) {
    const int tiles = ((C + 63) / 64) * ((C + 63) / 64) * ((k + WKT - 1) / WKT);
    // per weight-gradient role (two of them run next to ~200-270 pair workgroups); config-2 step: 112 -> 0.816 ms, 96 -> 0.819, 128 -> 0.838,
    // 80 -> 0.841, 144 -> 0.846, 160 -> 0.855, 64 -> 0.866, 192 -> 0.868, 48 -> 0.940
    int64_t target = 112;
    if (const char *e = PSND_ENV("PSND_PAIRBWD_BLOCKS")) target = psnd_env_int(e, target, 1, 65535);
    int64_t splits = target / tiles;
    if (splits < 1) splits = 1;
    int64_t rps = (R + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    if (rps < 64) rps = 64;
    splits = (R + rps - 1) / rps;
    if (rps_out) *rps_out = rps;
    return (int)splits;
}

extern "C" int psnd_conv1d_cl_pair_bwd_supported(int C, int k, int pad2, int dil2, int pad1, int dil1) {
    return C == 256 && k == 3 && psnd_conv1d_cl_pair_supported(C, k, pad2, -dil2, pad1, -dil1) && 32 - 2 * pairk::reach3(pad1, -dil1) >= 16 ? 1 : 0;
}

extern "C" int psnd_conv1d_cl_pair_bwd_splits(int64_t N, int Lp, int C, int k) {
    if (N <= 0 || Lp <= 0 || C <= 0 || k <= 0) return 0;
    return pair_bwd_splits(N * (int64_t)Lp, C, k, nullptr);
}

extern "C" int psnd_conv1d_cl_pair_bwd(const void *G, const void *wb2, const void *M1, float m1_slope, void *g_mid, const void *wb1, const void *M2,
                                       float m2_slope, const void *res, int64_t N, int Lp, int L, int HP, int C, int k, int pad2, int dil2,
                                       int pad1, int dil1, void *gx, const void *xa_a, float *gw_a, float *gb_a, const void *G_b,
                                       const void *xa_b, int off_b, int dstep_b, float *gw_b, float *gb_b, void *stream) {
    const bool have_pair = G != nullptr;
    const bool have_b = G_b != nullptr;
    if (!have_pair && !have_b) return PSND_OK;
    if (N <= 0 || Lp <= 0 || L <= 0 || HP < 0 || Lp < L + HP) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_pair_bwd: N=%lld Lp=%d L=%d HP=%d", (long long)N, Lp, L, HP);
    if (C != 256 || k != 3) PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_pair_bwd: C=%d k=%d (256 channels, 3 taps)", C, k);
    if ((size_t)N * Lp * C * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_pair_bwd: operand larger than 2 GB");
    if (have_pair && (!wb2 || !wb1 || !M1 || !g_mid || !gx || !xa_a || !gw_a)) PSND_FAIL(PSND_E_ARG, "conv1d_cl_pair_bwd: null pointer");
    if (have_pair && !psnd_conv1d_cl_pair_bwd_supported(C, k, pad2, dil2, pad1, dil1))
        PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_pair_bwd: taps (%d,%d) (%d,%d)", pad2, dil2, pad1, dil1);
    if (have_b && (!xa_b || !gw_b)) PSND_FAIL(PSND_E_ARG, "conv1d_cl_pair_bwd: null pointer (second weight gradient)");
    const int64_t R = N * (int64_t)Lp;
    int64_t rps;
    const int splits = pair_bwd_splits(R, C, k, &rps);
    const int wgx = (C + 63) / 64, wgy = (C + 63) / 64, wgz = splits * ((k + WKT - 1) / WKT);
    auto fill = [&](WgradParams &w, const void *g, const void *xa, int off0, int dstep, float *gw, float *gb) {
        wgrad_params_plain(w);
        w.G1 = static_cast<const bf16_t *>(g), w.G2 = nullptr, w.GM = nullptr;
        w.xa = static_cast<const bf16_t *>(xa), w.gw = gw, w.gbias = gb, w.g_out = nullptr;
        w.R = R, w.Ca = C, w.Cb = C, w.k = k, w.off0 = off0, w.dstep = dstep, w.g2_slope = 1.f, w.rows_per_split = (int)rps;
    };
    WgradParams wa, wb_;
    fill(wa, G, xa_a, -pad2, dil2, gw_a, gb_a);
    fill(wb_, G_b, xa_b, off_b, dstep_b, gw_b, gb_b);
    pairk::PairParams pp;
    memset(&pp, 0, sizeof(pp));
    int tiles = 0;
    size_t lds = kWgradLdsBytes;
    if (have_pair) {
        pp.A = static_cast<const pairk::bf16_t *>(G), pp.W1 = static_cast<const pairk::bf16_t *>(wb2), pp.W2 = static_cast<const pairk::bf16_t *>(wb1);
        pp.bias1 = nullptr, pp.bias2 = nullptr;
        pp.M1 = static_cast<const pairk::bf16_t *>(M1), pp.M2 = static_cast<const pairk::bf16_t *>(M2), pp.res = static_cast<const pairk::bf16_t *>(res);
        pp.mid_out = static_cast<pairk::bf16_t *>(g_mid), pp.out_raw = static_cast<pairk::bf16_t *>(gx), pp.out_act = nullptr;
        pp.R = R, pp.Lp = Lp, pp.L = L, pp.HP = HP;
        pp.off1 = pad2, pp.dstep1 = -dil2, pp.h1 = pairk::reach3(pad2, -dil2);
        pp.off2 = pad1, pp.dstep2 = -dil1, pp.h2 = pairk::reach3(pad1, -dil1);
        pp.m1_slope = m1_slope, pp.m2_slope = m2_slope, pp.act1_slope = 1.f, pp.act2_slope = 1.f, pp.trace = nullptr;
        const int MRW = 32 * PSND_PAIRBWD_MR, TS = MRW - 2 * pp.h2;
        tiles = (int)((R + TS - 1) / TS);
        const size_t lp = (size_t)(MRW + 2 * pp.h1 + MRW + 2 * pp.h2) * (C + 8) * 2, lo = (size_t)MRW * (C + 8) * 4;
        if (lds < lp) lds = lp;
        if (lds < lo) lds = lo;
    }
    const int nwa = have_pair ? wgx * wgy * wgz : 0, nwb = have_b ? wgx * wgy * wgz : 0;
    hipLaunchKernelGGL(conv_pair_bwd_kernel, dim3((unsigned)(nwa + nwb + tiles)), dim3(256), lds, static_cast<hipStream_t>(stream), pp, wa, wb_, nwa,
                       nwb, wgx, wgy);
    PSND_CHECK_LAUNCH("conv1d_cl_pair_bwd");
    if (have_pair) g_conv_pair_stats[2]++;
    g_conv_pair_stats[3].store(splits);
    return PSND_OK;
}

extern "C" int psnd_conv1d_cl_bwd(const void *G1, const void *G2, const void *GM, float g2_slope, const void *wb, const void *xa,
                                  int64_t N, int Lp, int L, int HP, int Ca, int Cb, int k, int pad, int dil, void *gx, void *g_out,
                                  const void *gx_mask, float gx_mask_slope, const void *gx_res, float *gw_part, float *gbias_part,
                                  void *stream) {
    static const bool no_pair = PSND_ENV("PSND_NO_BWD_PAIR") != nullptr;
    if (gw_part && (k < WKT ? k - 1 : WKT - 1) * (dil < 0 ? -dil : dil) > WXR - 32)
        PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_bwd: dilation %d: a tap group spans more than %d rows", dil, WXR - 32);
    int hm = 0;
    for (int j = 0; j < k; ++j) {
        const int o = pad - j * dil;
        hm = (o < 0 ? -o : o) > hm ? (o < 0 ? -o : o) : hm;
    }
    const size_t buf = sizeof(bf16_t) * 40 * (size_t)(BM + 2 * hm);
    const bool pairable = !no_pair && (G1 || G2) && (!G2 || GM) && gx && wb && xa && gw_part && k >= 1 && k <= 16 && (hm <= 25 || (hm <= 40 && k <= 7)) && hm <= HP && N > 0 &&
                          Ca % 32 == 0 && Cb % 32 == 0 && L > 0 && Lp >= L + 2 * HP &&
                          (size_t)N * Lp * (Ca > Cb ? Ca : Cb) * 2 < ((size_t)1 << 32) && (size_t)k * Cb * Ca * 2 < ((size_t)1 << 32);
    if (!pairable) {
        if (gw_part) {                           // NULL: the input gradient alone (the weight gradient is computed elsewhere: psnd_conv1d_cl_wgrad_multi)
            int rc = psnd_conv1d_cl_wgrad(G1, G2, GM, g2_slope, xa, N, Lp, Ca, Cb, k, -pad, dil, gw_part, gbias_part, nullptr, stream);
            if (rc != PSND_OK) return rc;
        }
        if (gx_res && gx_res == g_out) PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_bwd: residual = combined gradient outside the paired launch");
        return psnd_conv1d_cl(G1, G2, GM, g2_slope, wb, nullptr, gx_res, gx_mask, N, Lp, L, HP, Cb, Ca, k, pad, -dil, 1.f, gx_mask_slope, gx,
                              nullptr, g_out, stream);
    }
    // input-gradient role: a conv from Cb to Ca channels
    ConvParams pc;
    conv_params_plain(pc);
    pc.A = static_cast<const bf16_t *>(G1), pc.A2 = static_cast<const bf16_t *>(G2), pc.AM = static_cast<const bf16_t *>(GM);
    pc.a2_slope = g2_slope;
    pc.W = static_cast<const bf16_t *>(wb), pc.bias = nullptr;
    pc.res = static_cast<const bf16_t *>(gx_res), pc.mask_src = static_cast<const bf16_t *>(gx_mask);
    pc.out_raw = static_cast<bf16_t *>(gx), pc.out_act = nullptr, pc.a_eff_out = static_cast<bf16_t *>(g_out);
    if (gx_res && gx_res == g_out) {             // the residual IS the combined gradient this launch materialises
        if (Ca != Cb) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_bwd: gx_res == g_out needs Ca == Cb");
        pc.res = nullptr, pc.res_is_a_eff = 1;
    }
    pc.R = N * (int64_t)Lp, pc.Lp = Lp, pc.L = L, pc.HP = HP, pc.Ca = Cb, pc.Cb = Ca, pc.k = k, pc.off0 = pad, pc.dstep = -dil, pc.hm = hm;
    pc.act_slope = 1.f, pc.mask_slope = gx_mask_slope;
    // weight-gradient role
    WgradParams pw;
    wgrad_params_plain(pw);
    pw.G1 = pc.A, pw.G2 = pc.A2, pw.GM = pc.AM;
    pw.xa = static_cast<const bf16_t *>(xa), pw.gw = gw_part, pw.gbias = gbias_part, pw.g_out = nullptr;
    pw.R = pc.R, pw.Ca = Ca, pw.Cb = Cb, pw.k = k, pw.off0 = -pad, pw.dstep = dil, pw.g2_slope = g2_slope;
    int64_t rps;
    const int splits = wgrad_splits(pw.R, Ca, Cb, k, &rps);
    pw.rows_per_split = (int)rps;
    const char *pe = PSND_ENV("PSND_PAIR_MT");
    const int pair_mt = pe ? atoi(pe) : 0;
    // 128-row input-gradient tiles at every size: half as many workgroups of that role share the CUs with the weight-gradient role
    // (config-2 launch 17.4 -> 15.5 us with 192 weight-gradient workgroups; no change for the long HiFi-GAN stages, which took them anyway)
    // (round 5: 192-row input-gradient tiles - three MFMAs per weight fragment - where the forward takes them too (conv_row_tiles: stages
    //  with enough rows), for the plain instances: config-3 step 2.988 -> 2.955 ms, six interleaved runs each)
    const bool can3 = !G2 && hm <= 25 && Ca > 32;
    const int mt = (pair_mt == 1 || pair_mt == 2 || (pair_mt == 3 && can3)) ? pair_mt : ((can3 && conv_row_tiles(pc.R, Ca, true) == 3) ? 3 : 2);
    // narrow input gradient over long clips (Ca <= 32): 256-row tiles, the four waves along the rows (conv_cl_body, WNC = 1)
    const bool narrow = Ca <= 32 && mt == 2 && hm <= 25 && !G2 && (pc.R + 63) / 64 >= 1024 && !PSND_ENV("PSND_CONV_NO_NARROW");
    const int bm = narrow ? 256 : 64 * mt, bn = narrow ? 32 : BN;
    const int cgx = (int)((pc.R + bm - 1) / bm), cgy = (Ca + bn - 1) / bn;
    // 32 -> 32 channels over long clips: the weight-gradient role spreads 12 taps over the four waves (their span must fit its tile);
    // without the combined operand the input-gradient role is the single-stage instance at three workgroups per CU
    const bool wn = Ca <= 32 && Cb <= 32 && (pc.R + 63) / 64 >= 1024 && hm <= 25 && gw_part &&
                    ((k < 4 * WKT ? k : 4 * WKT) - 1) * (dil < 0 ? -dil : dil) <= WXRN - 32 && !PSND_ENV("PSND_CONV_NO_NARROW");
    const bool narrow1 = narrow && Cb <= 32;
    // taps per weight-gradient workgroup: as the instance picked below (KT by k, HMX by the reach)
    const int wk = wn ? 4 * WKT : (narrow1 ? WKT : pair_wk(k <= 3 ? 3 : (k <= 7 ? 7 : (k <= 11 ? 11 : 16)), hm > 25 ? 40 : 25));
    if (gw_part && !wn && ((k < wk ? k : wk) - 1) * (dil < 0 ? -dil : dil) > WXR - 32)
        PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_bwd: dilation %d: a group of %d taps spans more than %d rows", dil, wk, WXR - 32);
    const int wgx = (Cb + 63) / 64, wgy = (Ca + 63) / 64, wgz = splits * ((k + wk - 1) / wk);
    const int nw = wgx * wgy * wgz;
    const int kct = 32;
    size_t lds = 2 * sizeof(bf16_t) * (kct + 8) * (size_t)(bm + 2 * hm);
    if (lds < sizeof(float) * bm * (bn + 8)) lds = sizeof(float) * bm * (bn + 8);
    if (lds < (size_t)kWgradLdsBytes) lds = kWgradLdsBytes;
    if (wn && lds < (size_t)kWgradNarrowLdsBytes) lds = kWgradNarrowLdsBytes;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PSND_PAIR_LAUNCH(KT_, D_, C_, H_)                                                                              \
    do {                                                                                                              \
        auto kern = mt == 2 ? conv_bwd_pair_kernel<KT_, (C_ ? 2 : (D_ > 4 ? 4 : D_)), 2, C_, 2, H_> : conv_bwd_pair_kernel<KT_, D_, 2, C_, 1, H_>;   \
        if constexpr (H_ == 25 && !C_) {                                                                              \
            if (mt == 3) kern = conv_bwd_pair_kernel<KT_, (D_ > 4 ? 4 : D_), 2, false, 3, H_>;                        \
        }                                                                                                             \
        if constexpr (H_ == 25) {                                                                                     \
            if (wn && mt == 2) kern = conv_bwd_pair_kernel<KT_, (C_ ? 2 : (D_ > 4 ? 4 : D_)), 2, C_, 2, H_, 32, 2, true>;   \
        }                                                                                                             \
        if constexpr (H_ == 25 && !C_) {                                                                              \
            if (narrow1 && wn) kern = conv_bwd_pair_kernel<KT_, 1, 2, false, 2, H_, 32, 1, true>;   /* one k-stage: 3 workgroups per CU */ \
            else if (narrow1) kern = conv_bwd_pair_kernel<KT_, 1, 2, false, 2, H_, 32, 1>;                            \
            else if (narrow) kern = conv_bwd_pair_kernel<KT_, (D_ > 4 ? 4 : D_), 2, false, 2, H_, 32, 1>;             \
        }                                                                                                             \
        if (lds > 64 * 1024) {                                                                                        \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "conv1d_cl_bwd: set LDS size: %s", hipGetErrorString(e));      \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)(nw + cgx * cgy)), dim3(256), lds, st, pc, pw, nw, wgx, wgy, cgx);    \
    } while (0)
    if (hm > 25 && !G2) PSND_PAIR_LAUNCH(7, 3, false, 40);
    else if (hm > 25) PSND_PAIR_LAUNCH(7, 3, true, 40);
    else if (k <= 3 && !G2) PSND_PAIR_LAUNCH(3, 8, false, 25);
    else if (k <= 3) PSND_PAIR_LAUNCH(3, 4, true, 25);
    else if (k <= 7 && !G2) PSND_PAIR_LAUNCH(7, 3, false, 25);
    else if (k <= 7) PSND_PAIR_LAUNCH(7, 3, true, 25);
    else if (k <= 11 && !G2) PSND_PAIR_LAUNCH(11, 2, false, 25);
    else if (k <= 11) PSND_PAIR_LAUNCH(11, 2, true, 25);
    else if (!G2) PSND_PAIR_LAUNCH(16, 2, false, 25);
    else PSND_PAIR_LAUNCH(16, 2, true, 25);
#undef PSND_PAIR_LAUNCH
    PSND_CHECK_LAUNCH("conv1d_cl_bwd");
    g_conv_stats[mt == 2 ? 3 : 2]++;
    return PSND_OK;
}

extern "C" int psnd_conv1d_prep(const float *v, const float *g, const float *bias, int Cout, int Cin, int k, int Cb, int Ca,
                                void *wf, void *wb, float *bias_padded, void *stream) {
    if (!v || !g || !wf || !wb || !bias_padded) PSND_FAIL(PSND_E_ARG, "conv1d_prep: null pointer");
    if (Cb < Cout || Ca < Cin) PSND_FAIL(PSND_E_SHAPE, "conv1d_prep: padded sizes too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nw = sizeof(bf16_t) * (size_t)k * Cb * Ca;
    hipError_t e = hipSuccess;
    if (Cb != Cout || Ca != Cin) {
        e = hipMemsetAsync(wf, 0, nw, s);
        if (e == hipSuccess) e = hipMemsetAsync(wb, 0, nw, s);
        if (e == hipSuccess) e = hipMemsetAsync(bias_padded, 0, sizeof(float) * Cb, s);
    }
    if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "conv1d_prep: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(conv_prep_kernel, dim3(Cout), dim3(256), 0, s, v, g, bias, Cout, Cin, k, Cb, Ca,
                       static_cast<bf16_t *>(wf), static_cast<bf16_t *>(wb), bias_padded);
    PSND_CHECK_LAUNCH("conv1d_prep");
    return PSND_OK;
}

extern "C" int psnd_conv1d_prep_multi(const void *descs_dev, int n, int total_blocks, int max_row, int which, void *stream) {
    if (!descs_dev || n <= 0 || total_blocks <= 0 || max_row <= 0 || which < 1 || which > 3) PSND_FAIL(PSND_E_ARG, "conv1d_prep_multi: bad arguments");
    const size_t lds = (size_t)PREP_ROWS * (size_t)max_row * sizeof(bf16_t);
    if (lds > 160 * 1024) PSND_FAIL(PSND_E_SHAPE, "conv1d_prep_multi: Cin * k = %d too large for the 8-row tile (use psnd_conv1d_prep)", max_row);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_prep_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "conv1d_prep_multi: set LDS size: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(conv_prep_multi_kernel, dim3(total_blocks), dim3(256), lds, static_cast<hipStream_t>(stream),
                       static_cast<const PrepDesc *>(descs_dev), n, which);
    PSND_CHECK_LAUNCH("conv1d_prep_multi");
    return PSND_OK;
}

extern "C" int psnd_conv1d_wnorm_bwd(const float *gw_part, const float *gbias_part, int splits, const float *v, const float *g,
                                     int Cout, int Cin, int k, int Cb, int Ca, float *gv, float *gg, float *gbias, void *stream) {
    if (!gw_part || !v || !g || !gv || !gg || splits < 1) PSND_FAIL(PSND_E_ARG, "conv1d_wnorm_bwd: null pointer / splits");
    const size_t lds = sizeof(float) * (size_t)(Cin + 1) * k;
    if (lds > 60 * 1024) PSND_FAIL(PSND_E_SHAPE, "conv1d_wnorm_bwd: Cin*k=%d too large", Cin * k);   // + 4 KB of slab-group sums
    const unsigned kmagic = (unsigned)((0x100000000ull + (unsigned)k - 1) / (unsigned)k);
    const unsigned cmagic = (unsigned)((0x100000000ull + (unsigned)Cin - 1) / (unsigned)Cin);
    const int threads = Cin * k >= 2048 ? 1024 : 256;         // big rows: 4x the loads in flight per output channel
    if (splits > 16)                                          // slab groups: 1024 threads + their partial sums (conv_finish_body)
        hipLaunchKernelGGL(conv_finish_kernel<true>, dim3(Cout), dim3(1024), lds + 4096, static_cast<hipStream_t>(stream), gw_part,
                           gbias_part, splits, v, g, Cout, Cin, k, Cb, Ca, gv, gg, gbias, kmagic, cmagic);
    else
        hipLaunchKernelGGL(conv_finish_kernel<false>, dim3(Cout), dim3(threads), lds, static_cast<hipStream_t>(stream), gw_part,
                           gbias_part, splits, v, g, Cout, Cin, k, Cb, Ca, gv, gg, gbias, kmagic, cmagic);
    PSND_CHECK_LAUNCH("conv1d_wnorm_bwd");
    return PSND_OK;
}

extern "C" int psnd_conv1d_wnorm_bwd_multi(const psnd_wnorm_desc *descs, int n, void *stream) {
    if (!descs || n < 1 || n > PSND_WNORM_MAX) PSND_FAIL(PSND_E_ARG, "conv1d_wnorm_bwd_multi: %d descriptors (1..%d)", n, PSND_WNORM_MAX);
    FinishArgs a;
    a.n = n;
    size_t lds = 0;
    int total = 0, maxn = 0, maxsplits = 0;
    for (int i = 0; i < n; ++i) {
        const psnd_wnorm_desc &d = descs[i];
        if (!d.gw_part || !d.v || !d.g || !d.gv || !d.gg || d.splits < 1 || d.Cout < 1 || d.Cin < 1 || d.k < 1)
            PSND_FAIL(PSND_E_ARG, "conv1d_wnorm_bwd_multi: descriptor %d: null pointer / bad sizes", i);
        const size_t l = sizeof(float) * (size_t)(d.Cin + 1) * d.k;
        if (l > 60 * 1024) PSND_FAIL(PSND_E_SHAPE, "conv1d_wnorm_bwd_multi: Cin*k=%d too large", d.Cin * d.k);
        lds = l > lds ? l : lds;
        a.d[i] = d;
        a.kmagic[i] = (unsigned)((0x100000000ull + (unsigned)d.k - 1) / (unsigned)d.k);
        a.cmagic[i] = (unsigned)((0x100000000ull + (unsigned)d.Cin - 1) / (unsigned)d.Cin);
        a.blk0[i] = total;
        total += d.Cout;
        maxn = d.Cin * d.k > maxn ? d.Cin * d.k : maxn;
        maxsplits = d.splits > maxsplits ? d.splits : maxsplits;
    }
    a.blk0[n] = total;
    const int threads = maxn >= 2048 ? 1024 : 256;
    if (maxsplits > 16)
        hipLaunchKernelGGL(conv_finish_multi_kernel<true>, dim3(total), dim3(1024), lds + 4096, static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(conv_finish_multi_kernel<false>, dim3(total), dim3(threads), lds, static_cast<hipStream_t>(stream), a);
    PSND_CHECK_LAUNCH("conv1d_wnorm_bwd_multi");
    return PSND_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ConvTranspose1d(Cin, Cout, K = 2u, stride u, padding p = u/2) of the HiFi-GAN upsamplers (hifi_gan.py:109, 118-121) in
// POLYPHASE form on the same implicit-GEMM kernels:  y[t] = x[i] w[:, :, phi] + x[i-1] w[:, :, phi + u],  (t + p) = i u + phi.
// Seen from the low-resolution rows it is ONE 2-tap convolution from Cin to u * Cout channels whose output row l, column
// block phi IS high-resolution row (l - HP) u + phi - p (ConvParams::up_role); no zero is multiplied, the activations never
// leave the CL layout.  Weight norm of nn.ConvTranspose1d runs over dim 0 = per INPUT channel.
//   forward        psnd_convtr1d_cl_fwd : A = xa (low),  W = wf [2][u*Cr][Cip],  taps (0, -1),  output = high view (role 1)
//   input gradient psnd_convtr1d_cl_bwd : A = g (high view, role 2), W = wb [2][Cip][u*Cr], taps (0, +1) -> gx (low); the combined
//                  gradient g = g_raw + g_act * leaky'(act) is written back (g_eff) for the weight gradient
//   weight gradient: the wgrad kernel with the roles of its operands swapped ("g" := xa, "xa" := g_eff as the high view):
//                  slabs gw[S][2][Cip][u*Cr] - channel-contiguous per input channel, which is the weight-norm group
//   psnd_convtr1d_wnorm_bwd: slabs -> (g_v, g_g);  the bias gradient is the column sum of g_eff (caller)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// one block per input channel ci: w = g[ci] * v[ci] / ||v[ci]||;  v: (Cin, Cout, K), K = 2u
__global__ __launch_bounds__(256) void convtr_prep_kernel(const float *v, const float *g, const float *bias, int Cin, int Cout, int K, int u,
                                                          int Cr, int Cip, bf16_t *wf, bf16_t *wb, float *bp) {
    __shared__ float red[4];
    const int ci = blockIdx.x, n = Cout * K;
    const float *vr = v + (size_t)ci * n;
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) ss += vr[i] * vr[i];
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float scale = g[ci] / __builtin_sqrtf(red[0] + red[1] + red[2] + red[3]);
    const int Nn = u * Cr;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int co = i / K, jk = i - co * K;
        const int tap = jk / u, phi = jk - tap * u;
        const bf16_t w = f2bf(vr[i] * scale);
        wf[pack_index(2, tap, phi * Cr + co, ci, Nn, Cip)] = w;
        wb[pack_index(2, tap, ci, phi * Cr + co, Cip, Nn)] = w;
    }
    if (ci == 0)
        for (int i = threadIdx.x; i < Nn; i += 256) {
            const int co = i % Cr;
            bp[i] = (bias && co < Cout) ? bias[co] : 0.f;
        }
}

// one block per input channel ci: s[co][jk] = sum over slabs of gw[tap][ci][phi * Cr + co] (jk = phi + tap * u), then the weight-norm
// backward of the group v[ci] (Cout * K elements):  g_g = <s, vhat>,  g_v = (g / ||v||) (s - vhat g_g)
// Many slabs (the narrow upsamplers over long clips): launched with 1024 threads that form BD / EP slab groups (EP = the group's
// elements rounded up to a power of two), as conv_finish_body<true> - one thread per element walked `splits` dependent loads.
__global__ __launch_bounds__(1024) void convtr_finish_kernel(const float *gw_part, int splits, const float *v, const float *g, int Cin,
                                                             int Cout, int K, int u, int Cr, int Cip, float *gv, float *gg) {
    extern __shared__ float s_gw[];          // Cout * K, then the slab groups' partial sums [NSG][EP]
    __shared__ float red[32];
    const int ci = blockIdx.x, n = Cout * K, Nn = u * Cr, BD = blockDim.x, tid = threadIdx.x;
    const size_t slab = (size_t)2 * Cip * Nn;
    const float *vr = v + (size_t)ci * n;
    float ss = 0.f, dot = 0.f;
    const int E = 2 * u * Cout;
    int EP = 64;
    while (EP < E && EP < BD) EP <<= 1;
    const int NSG = BD / EP, el = tid & (EP - 1), sg = tid / EP;
    float *s_part = s_gw + n;
    for (int e0 = 0; e0 < E; e0 += EP) {                             // (tap, phi, co) with co fastest: coalesced slab reads
        const int e = e0 + el;
        const bool ok = e < E;
        const int co = e % Cout, tp = e / Cout;                      // tp = tap * u + phi = jk
        const int tap = tp / u, phi = tp - tap * u;
        const float *src = gw_part + ((size_t)tap * Cip + ci) * Nn + phi * Cr + co;
        float a = 0.f;
        for (int sp0 = sg; sp0 < splits; sp0 += 8 * NSG) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = (ok && sp0 + q * NSG < splits) ? src[(size_t)(sp0 + q * NSG) * slab] : 0.f;
            a += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        if (NSG > 1) {
            s_part[sg * EP + el] = a;
            __syncthreads();
            if (sg == 0 && ok) {
                float a2 = 0.f;
                for (int q = 0; q < NSG; ++q) a2 += s_part[q * EP + el];
                s_gw[co * K + tp] = a2;
            }
            __syncthreads();
        } else if (ok) {
            s_gw[co * K + tp] = a;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += BD) {
        const float vv = vr[i];
        ss += vv * vv;
        dot += vv * s_gw[i];
    }
    for (int m = 32; m >= 1; m >>= 1) {
        ss += __shfl_xor(ss, m, 64);
        dot += __shfl_xor(dot, m, 64);
    }
    const int nw = BD >> 6;
    if ((tid & 63) == 0) red[tid >> 6] = ss, red[16 + (tid >> 6)] = dot;
    __syncthreads();
    float sst = 0.f, dott = 0.f;
    for (int w = 0; w < nw; ++w) sst += red[w], dott += red[16 + w];
    const float inv = 1.f / __builtin_sqrtf(sst);
    const float d = dott * inv;
    const float gs = g[ci] * inv;
    for (int i = tid; i < n; i += BD) gv[(size_t)ci * n + i] = gs * (s_gw[i] - vr[i] * inv * d);
    if (tid == 0) gg[ci] = d;
}

// y = leaky_relu((a + b + c + d) / count, slope) over bf16 buffers (the mean of a stage's resblocks + the next activation,
// hifi_gan.py:122-131), and its backward g_in = g * leaky'(y) / count (the same tensor for every branch)
__global__ __launch_bounds__(256) void cl_mean_act_kernel(const bf16_t *a, const bf16_t *b, const bf16_t *c, const bf16_t *d, int count,
                                                          float slope, bf16_t *out, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const bf16_t *src[4] = {a, b, c, d};
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < count; ++s) {
        const uint4 q = reinterpret_cast<const uint4 *>(src[s])[i];
        const unsigned *pq = reinterpret_cast<const unsigned *>(&q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[2 * e] += bf2f((bf16_t)(pq[e] & 0xffff)), acc[2 * e + 1] += bf2f((bf16_t)(pq[e] >> 16));
    }
    unsigned w[4];
    const float cnt = (float)count;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = acc[2 * e] / cnt, x1 = acc[2 * e + 1] / cnt;
        w[e] = pack_bf16(x0 > 0.f ? x0 : x0 * slope, x1 > 0.f ? x1 : x1 * slope);
    }
    reinterpret_cast<uint4 *>(out)[i] = make_uint4(w[0], w[1], w[2], w[3]);
}
// out_g = sum of up to four bf16 buffers (fp32 accumulation, one rounding), for TWO groups in one launch (blockIdx.y): the gradients that
// arrive at an upsampler's two outputs from the resblocks of its stage (hifi_gan.py:122-131: every resblock reads x and adds to xs) -
// autograd's accumulation was 2 x (count - 1) library add launches per stage
struct SumGroup {
    const bf16_t *src[4];
    bf16_t *out;
    int count;
};
__global__ __launch_bounds__(256) void cl_sum2_kernel(SumGroup ga, SumGroup gb, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const SumGroup &g = blockIdx.y == 0 ? ga : gb;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < g.count; ++s) {
        const uint4 q = reinterpret_cast<const uint4 *>(g.src[s])[i];
        const unsigned *pq = reinterpret_cast<const unsigned *>(&q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[2 * e] += bf2f((bf16_t)(pq[e] & 0xffff)), acc[2 * e + 1] += bf2f((bf16_t)(pq[e] >> 16));
    }
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack_bf16(acc[2 * e], acc[2 * e + 1]);
    reinterpret_cast<uint4 *>(g.out)[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void cl_mean_act_bwd_kernel(const bf16_t *g, const bf16_t *y, int count, float slope, bf16_t *gin, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 qg = reinterpret_cast<const uint4 *>(g)[i], qy = reinterpret_cast<const uint4 *>(y)[i];
    const unsigned *pg = reinterpret_cast<const unsigned *>(&qg), *py = reinterpret_cast<const unsigned *>(&qy);
    unsigned w[4];
    const float cnt = (float)count;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g0 = bf2f((bf16_t)(pg[e] & 0xffff)), g1 = bf2f((bf16_t)(pg[e] >> 16));
        const float y0 = bf2f((bf16_t)(py[e] & 0xffff)), y1 = bf2f((bf16_t)(py[e] >> 16));
        w[e] = pack_bf16(g0 * (y0 > 0.f ? 1.f : slope) / cnt, g1 * (y1 > 0.f ? 1.f : slope) / cnt);
    }
    reinterpret_cast<uint4 *>(gin)[i] = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace

static int convtr_check(const char *what, int64_t N, int Lp, int L, int HP, int Cip, int Cr, int u, int p, int LpO, int HPO) {
    if (u < 1 || u > 16 || p < 0 || p > u) PSND_FAIL(PSND_E_SHAPE, "%s: stride %d, padding %d (kernel size must be 2 * stride, padding <= stride)", what, u, p);
    if (Cip % 32 != 0 || Cr % 32 != 0) PSND_FAIL(PSND_E_SHAPE, "%s: channel counts %d, %d must be multiples of 32", what, Cip, Cr);
    if (N < 0 || L <= 0 || HP < 1 || Lp < L + 2 * HP) PSND_FAIL(PSND_E_SHAPE, "%s: low-resolution geometry N=%lld Lp=%d L=%d HP=%d (HP >= 1)", what, (long long)N, Lp, L, HP);
    if (HPO < p || LpO < HPO + L * u + (u - p)) PSND_FAIL(PSND_E_SHAPE, "%s: high-resolution geometry LpO=%d HPO=%d for %d rows (HPO >= p, u - p rows behind the clip)", what, LpO, HPO, L * u);
    if ((size_t)N * LpO * Cr * 2 >= ((size_t)1 << 32) || (size_t)N * Lp * Cip * 2 >= ((size_t)1 << 32) || (size_t)4 * u * Cr * Cip >= ((size_t)1 << 32))
        PSND_FAIL(PSND_E_SHAPE, "%s: operand larger than 4 GB (32-bit buffer offsets)", what);
    return PSND_OK;
}

// psnd_convtr1d_prep for MANY transposed convs in one launch (the four upsamplers of a HiFi-GAN generator: four ~10 us launches were launch
// latency).  One workgroup per input channel of every layer; the pads of wf / wb are never written (zero them once).
namespace {
struct ConvtrPrepDesc {           // 80 bytes, mirrored by pytorch_sound_amd/cl.py ('<6Q8i')
    const float *v, *g, *bias;
    bf16_t *wf, *wb;
    float *bp;
    int Cin, Cout, K, u, Cr, Cip, blk0, pad_;
};
__global__ __launch_bounds__(256) void convtr_prep_multi_kernel(const ConvtrPrepDesc *descs, int n) {
    __shared__ float red[4];
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const ConvtrPrepDesc d = descs[lo];
    const int ci = blockIdx.x - d.blk0, nn = d.Cout * d.K;
    if (ci >= d.Cin) return;
    const float *vr = d.v + (size_t)ci * nn;
    float ss = 0.f;
    for (int i = threadIdx.x; i < nn; i += 256) ss = __builtin_fmaf(vr[i], vr[i], ss);
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float scale = d.g[ci] / __builtin_sqrtf(red[0] + red[1] + red[2] + red[3]);
    const int Nn = d.u * d.Cr;
    for (int i = threadIdx.x; i < nn; i += 256) {
        const int co = i / d.K, jk = i - co * d.K;
        const int tap = jk / d.u, phi = jk - tap * d.u;
        const bf16_t w = f2bf(vr[i] * scale);
        d.wf[pack_index(2, tap, phi * d.Cr + co, ci, Nn, d.Cip)] = w;
        d.wb[pack_index(2, tap, ci, phi * d.Cr + co, d.Cip, Nn)] = w;
    }
    if (ci == 0)
        for (int i = threadIdx.x; i < Nn; i += 256) {
            const int co = i % d.Cr;
            d.bp[i] = (d.bias && co < d.Cout) ? d.bias[co] : 0.f;
        }
}
}  // namespace

extern "C" int psnd_convtr1d_prep_multi(const void *descs_dev, int n, int total_blocks, void *stream) {
    if (!descs_dev || n <= 0 || total_blocks <= 0) PSND_FAIL(PSND_E_ARG, "convtr1d_prep_multi: bad arguments");
    hipLaunchKernelGGL(convtr_prep_multi_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const ConvtrPrepDesc *>(descs_dev), n);
    PSND_CHECK_LAUNCH("convtr1d_prep_multi");
    return PSND_OK;
}

extern "C" int psnd_convtr1d_prep(const float *v, const float *g, const float *bias, int Cin, int Cout, int K, int stride, int Cr, int Cip,
                                  void *wf, void *wb, float *bias_rep, void *stream) {
    if (!v || !g || !wf || !wb || !bias_rep) PSND_FAIL(PSND_E_ARG, "convtr1d_prep: null pointer");
    if (K != 2 * stride || Cr < Cout || Cip < Cin || Cr % 32 != 0 || Cip % 32 != 0) PSND_FAIL(PSND_E_SHAPE, "convtr1d_prep: K=%d stride=%d Cr=%d Cip=%d", K, stride, Cr, Cip);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cr != Cout || Cip != Cin) {
        const size_t nb = sizeof(bf16_t) * (size_t)2 * stride * Cr * Cip;
        hipError_t e = hipMemsetAsync(wf, 0, nb, s);
        if (e == hipSuccess) e = hipMemsetAsync(wb, 0, nb, s);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "convtr1d_prep: memset: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(convtr_prep_kernel, dim3(Cin), dim3(256), 0, s, v, g, bias, Cin, Cout, K, stride, Cr, Cip, static_cast<bf16_t *>(wf),
                       static_cast<bf16_t *>(wb), bias_rep);
    PSND_CHECK_LAUNCH("convtr1d_prep");
    return PSND_OK;
}

extern "C" int psnd_convtr1d_cl_fwd(const void *xa, const void *wf, const float *bias_rep, int64_t N, int Lp, int L, int HP, int Cip, int Cr,
                                    int stride, int padding, int LpO, int HPO, float act_slope, void *out_raw, void *out_act, void *stream) {
    if (!xa || !wf || (!out_raw && !out_act)) PSND_FAIL(PSND_E_ARG, "convtr1d_cl_fwd: null pointer");
    int rc = convtr_check("convtr1d_cl_fwd", N, Lp, L, HP, Cip, Cr, stride, padding, LpO, HPO);
    if (rc != PSND_OK) return rc;
    if (N == 0) return PSND_OK;
    ConvParams p;
    conv_params_plain(p);
    p.A = static_cast<const bf16_t *>(xa), p.A2 = nullptr, p.AM = nullptr, p.a2_slope = 1.f;
    p.W = static_cast<const bf16_t *>(wf), p.bias = bias_rep, p.res = nullptr, p.mask_src = nullptr;
    p.out_raw = static_cast<bf16_t *>(out_raw), p.out_act = static_cast<bf16_t *>(out_act), p.a_eff_out = nullptr;
    p.R = N * (int64_t)Lp, p.Lp = Lp, p.L = L, p.HP = HP, p.Ca = Cip, p.Cb = stride * Cr, p.k = 2, p.off0 = 0, p.dstep = -1, p.hm = 1;
    p.act_slope = act_slope, p.mask_slope = 1.f;
    p.up_role = 1, p.up_u = stride, p.up_p = padding, p.up_LpO = LpO, p.up_HPO = HPO, p.up_LO = L * stride, p.up_Cr = Cr;
    return conv_launch(p, static_cast<hipStream_t>(stream), "convtr1d_cl_fwd");
}

extern "C" int psnd_convtr1d_cl_wgrad_splits(int64_t N, int Lp, int Cip, int Cr, int stride) {
    if (N <= 0 || Lp <= 0 || Cip <= 0 || Cr <= 0 || stride <= 0) return 0;
    return wgrad_splits(N * (int64_t)Lp, stride * Cr, Cip, 2, nullptr);
}

extern "C" int psnd_convtr1d_cl_bwd(const void *g_raw, const void *g_act, const void *act, float act_slope, const void *wb, const void *xa,
                                    int64_t N, int Lp, int L, int HP, int Cip, int Cr, int stride, int padding, int LpO, int HPO, void *gx,
                                    void *g_eff, float *gw_part, void *stream) {
    // gx == NULL: the weight-gradient launch alone (g_eff already holds the combined gradient); gw_part == NULL: the input-gradient launch alone
    // - the two roles of one backward enqueued on two streams by two calls
    if ((!g_raw && !g_act) || (g_act && !act) || !wb || !xa || (!gx && !gw_part)) PSND_FAIL(PSND_E_ARG, "convtr1d_cl_bwd: null pointer");
    if (g_act && !g_eff) PSND_FAIL(PSND_E_ARG, "convtr1d_cl_bwd: g_eff (the combined gradient, N x LpO x Cr) is needed when g_act is given");
    int rc = convtr_check("convtr1d_cl_bwd", N, Lp, L, HP, Cip, Cr, stride, padding, LpO, HPO);
    if (rc != PSND_OK) return rc;
    if (N == 0) return PSND_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ConvParams p;
    conv_params_plain(p);
    p.A = static_cast<const bf16_t *>(g_raw), p.A2 = static_cast<const bf16_t *>(g_act), p.AM = static_cast<const bf16_t *>(act);
    p.a2_slope = act_slope;
    p.W = static_cast<const bf16_t *>(wb), p.bias = nullptr, p.res = nullptr, p.mask_src = nullptr;
    p.out_raw = static_cast<bf16_t *>(gx), p.out_act = nullptr, p.a_eff_out = g_act ? static_cast<bf16_t *>(g_eff) : nullptr;
    p.R = N * (int64_t)Lp, p.Lp = Lp, p.L = L, p.HP = HP, p.Ca = stride * Cr, p.Cb = Cip, p.k = 2, p.off0 = 0, p.dstep = 1, p.hm = 1;
    p.act_slope = 1.f, p.mask_slope = 1.f;
    p.up_role = 2, p.up_u = stride, p.up_p = padding, p.up_LpO = LpO, p.up_HPO = HPO, p.up_LO = L * stride, p.up_Cr = Cr;
    if (gx) {
        rc = conv_launch(p, st, "convtr1d_cl_bwd(data)");
        if (rc != PSND_OK) return rc;
    }
    if (!gw_part) return PSND_OK;
    // weight gradient, operands swapped: gw[tap][ci][phi, co] = sum_l xa[l][ci] * G[l + tap][phi, co]
    WgradParams w;
    wgrad_params_plain(w);
    w.G1 = static_cast<const bf16_t *>(xa), w.G2 = nullptr, w.GM = nullptr;
    w.xa = g_act ? static_cast<const bf16_t *>(g_eff) : static_cast<const bf16_t *>(g_raw);
    w.gw = gw_part, w.gbias = nullptr, w.g_out = nullptr;
    w.R = p.R, w.Ca = stride * Cr, w.Cb = Cip, w.k = 2, w.off0 = 0, w.dstep = 1, w.g2_slope = 1.f;
    w.up_u = stride, w.up_p = padding, w.up_Lp = Lp, w.up_HP = HP, w.up_LpO = LpO, w.up_HPO = HPO, w.up_Cr = Cr;
    int64_t rps;
    const int splits = wgrad_splits(w.R, w.Ca, w.Cb, 2, &rps);
    w.rows_per_split = (int)rps;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((w.Cb + 63) / 64, (w.Ca + 63) / 64, (unsigned)splits), dim3(256), kWgradLdsBytes, st, w);
    PSND_CHECK_LAUNCH("convtr1d_cl_bwd(wgrad)");
    return PSND_OK;
}

extern "C" int psnd_convtr1d_wnorm_bwd(const float *gw_part, int splits, const float *v, const float *g, int Cin, int Cout, int K, int stride,
                                       int Cr, int Cip, float *gv, float *gg, void *stream) {
    if (!gw_part || !v || !g || !gv || !gg || splits < 1) PSND_FAIL(PSND_E_ARG, "convtr1d_wnorm_bwd: null pointer / splits");
    if (K != 2 * stride) PSND_FAIL(PSND_E_SHAPE, "convtr1d_wnorm_bwd: K=%d stride=%d", K, stride);
    const size_t lds = sizeof(float) * ((size_t)Cout * K + 1024);     // + the slab groups' partial sums
    if (lds > 64 * 1024) PSND_FAIL(PSND_E_SHAPE, "convtr1d_wnorm_bwd: Cout*K=%d too large", Cout * K);
    hipLaunchKernelGGL(convtr_finish_kernel, dim3(Cin), dim3(splits > 16 ? 1024 : 256), lds, static_cast<hipStream_t>(stream), gw_part, splits,
                       v, g, Cin, Cout, K, stride, Cr, Cip, gv, gg);
    PSND_CHECK_LAUNCH("convtr1d_wnorm_bwd");
    return PSND_OK;
}


// ---- column sums of a channels-last bf16 matrix: out[c] = sum over rows of g[r][c] in fp32 - the bias gradient of a transposed conv
//      (hifi_gan.py:107-110: d/d bias = the sum of the output gradient over clips and positions; halo rows are zero).  Two stages, fixed
//      summation order: a workgroup sums the rows of its split (a thread owns 8 consecutive columns, 16-byte loads), the partial rows
//      [splits][C] are added by the second launch.  Replaces torch.sum(g.view(-1, C), 0) - 21 us per upsampler at config 3 for ~3 us of
//      traffic (a library reduction over 32 .. 256 narrow columns).
static __global__ __launch_bounds__(256) void cl_colsum_partial_kernel(const bf16_t *g, long long rows, int C, int splits, float *part, int Lp, int lo,
                                                                        int hi) {
    __shared__ float red[256 * 8];
    const int cols8 = C >> 3;                           // column groups of 8
    const int rpp = 256 / cols8;                        // rows per pass (cols8 <= 256: checked at launch)
    const int tid = threadIdx.x, cg = tid % cols8, rr = tid / cols8;
    const long long r0 = rows * blockIdx.x / splits, r1 = rows * (blockIdx.x + 1) / splits;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rr < rpp) {
        int l = Lp > 0 ? (int)((r0 + rr) % Lp) : 0;    // Lp > 0: only the rows [lo, hi) of every Lp-row clip buffer count (the rest may be unwritten)
        for (long long r = r0 + rr; r < r1; r += rpp) {
            const bool in = Lp <= 0 || (l >= lo && l < hi);
            if (Lp > 0) {
                l += rpp;
                while (l >= Lp) l -= Lp;
            }
            if (!in) continue;
            const u32x4 q = *reinterpret_cast<const u32x4 *>(g + (size_t)r * C + 8 * cg);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] += __builtin_bit_cast(float, q[j] << 16);
                acc[2 * j + 1] += __builtin_bit_cast(float, q[j] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
    __syncthreads();
    if (tid < C) {                                      // column tid: the rpp partial sums of its group, in order
        const int cgc = tid >> 3, j = tid & 7;
        float sum = 0.f;
        for (int k = 0; k < rpp; ++k) sum += red[(k * cols8 + cgc) * 8 + j];
        part[(size_t)blockIdx.x * C + tid] = sum;
    }
}
static __global__ __launch_bounds__(1024) void cl_colsum_final_kernel(const float *part, int C, int splits, float *out) {
    // one workgroup: thread (k, c) adds the partial rows k, k + K, ... of column c (K = 1024 / C row groups, eight loads in flight per thread),
    // then the K sums per column are added in order.  (A serial loop over the splits by C threads alone took 73 us per launch at config 3,
    // 256 threads with one load in flight 24 us.)
    __shared__ float red[1024];
    const int tid = threadIdx.x, K = 1024 / C, k = tid / C, c = tid % C;
    float sum = 0.f;
    if (k < K) {
        int s = k;
        for (; s + 7 * K < splits; s += 8 * K) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(s + j * K) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[j];
        }
        for (; s < splits; s += K) sum += part[(size_t)s * C + c];
    }
    red[tid] = sum;
    __syncthreads();
    if (tid < C) {
        float tot = 0.f;
        for (int j = 0; j < K; ++j) tot += red[j * C + tid];
        out[tid] = tot;
    }
}

extern "C" int psnd_cl_colsum_splits(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 1;
    const long long want = (rows * (long long)C * 2 + (1 << 15) - 1) >> 15;      // ~32 KB of the matrix per workgroup, at most 128 of them
    return (int)(want < 1 ? 1 : (want > 128 ? 128 : want));
}

extern "C" int psnd_cl_colsum(const void *g, int64_t rows, int C, int Lp, int lo, int hi, float *part, float *out, void *stream) {
    if (!g || !part || !out) PSND_FAIL(PSND_E_ARG, "cl_colsum: null pointer");
    if (C % 8 != 0 || C < 8 || C > 256 || rows <= 0) PSND_FAIL(PSND_E_SHAPE, "cl_colsum: rows=%lld, C=%d (C % 8 == 0, 8 <= C <= 256)", (long long)rows, C);
    if (Lp < 0 || (Lp > 0 && (lo < 0 || hi > Lp || lo > hi || rows % Lp != 0)))
        PSND_FAIL(PSND_E_SHAPE, "cl_colsum: rows=%lld in buffers of Lp=%d rows, window [%d, %d)", (long long)rows, Lp, lo, hi);
    const int splits = psnd_cl_colsum_splits(rows, C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(cl_colsum_partial_kernel, dim3(splits), dim3(256), 0, st, static_cast<const bf16_t *>(g), (long long)rows, C, splits, part, Lp, lo, hi);
    PSND_CHECK_LAUNCH("cl_colsum(partial)");
    hipLaunchKernelGGL(cl_colsum_final_kernel, dim3(1), dim3(1024), 0, st, part, C, splits, out);
    PSND_CHECK_LAUNCH("cl_colsum(final)");
    return PSND_OK;
}

extern "C" int psnd_cl_mean_act_fwd(const void *a, const void *b, const void *c, const void *d, int count, float slope, void *out, int64_t n,
                                    void *stream) {
    if (!a || !out || count < 1 || count > 4 || (count > 1 && !b) || (count > 2 && !c) || (count > 3 && !d)) PSND_FAIL(PSND_E_ARG, "cl_mean_act_fwd: arguments");
    if (n % 8 != 0) PSND_FAIL(PSND_E_SHAPE, "cl_mean_act_fwd: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return PSND_OK;
    const size_t n8 = (size_t)n / 8;
    hipLaunchKernelGGL(cl_mean_act_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t *>(a), static_cast<const bf16_t *>(b), static_cast<const bf16_t *>(c), static_cast<const bf16_t *>(d),
                       count, slope, static_cast<bf16_t *>(out), n8);
    PSND_CHECK_LAUNCH("cl_mean_act_fwd");
    return PSND_OK;
}

extern "C" int psnd_cl_sum2(const void *const *a, int na, void *out_a, const void *const *b, int nb, void *out_b, int64_t n, void *stream) {
    if (na < 0 || na > 4 || nb < 0 || nb > 4 || (na == 0 && nb == 0)) PSND_FAIL(PSND_E_ARG, "cl_sum2: %d / %d sources (0..4 each, not both 0)", na, nb);
    if ((na > 0 && (!a || !out_a)) || (nb > 0 && (!b || !out_b))) PSND_FAIL(PSND_E_ARG, "cl_sum2: null pointer");
    if (n % 8 != 0 || n < 0) PSND_FAIL(PSND_E_SHAPE, "cl_sum2: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return PSND_OK;
    SumGroup ga = {}, gb = {};
    for (int i = 0; i < na; ++i) {
        if (!a[i]) PSND_FAIL(PSND_E_ARG, "cl_sum2: null source");
        ga.src[i] = static_cast<const bf16_t *>(a[i]);
    }
    for (int i = 0; i < nb; ++i) {
        if (!b[i]) PSND_FAIL(PSND_E_ARG, "cl_sum2: null source");
        gb.src[i] = static_cast<const bf16_t *>(b[i]);
    }
    ga.out = static_cast<bf16_t *>(out_a), ga.count = na, gb.out = static_cast<bf16_t *>(out_b), gb.count = nb;
    if (na == 0) ga = gb, gb.count = 0;               // one group only: it is group 0
    const size_t n8 = (size_t)n / 8;
    hipLaunchKernelGGL(cl_sum2_kernel, dim3((unsigned)((n8 + 255) / 256), gb.count > 0 ? 2 : 1), dim3(256), 0, static_cast<hipStream_t>(stream), ga, gb, n8);
    PSND_CHECK_LAUNCH("cl_sum2");
    return PSND_OK;
}

extern "C" int psnd_cl_mean_act_bwd(const void *g, const void *y, int count, float slope, void *gin, int64_t n, void *stream) {
    if (!g || !y || !gin || count < 1) PSND_FAIL(PSND_E_ARG, "cl_mean_act_bwd: arguments");
    if (n % 8 != 0) PSND_FAIL(PSND_E_SHAPE, "cl_mean_act_bwd: n=%lld must be a multiple of 8", (long long)n);
    if (n == 0) return PSND_OK;
    const size_t n8 = (size_t)n / 8;
    hipLaunchKernelGGL(cl_mean_act_bwd_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t *>(g), static_cast<const bf16_t *>(y), count, slope, static_cast<bf16_t *>(gin), n8);
    PSND_CHECK_LAUNCH("cl_mean_act_bwd");
    return PSND_OK;
}
