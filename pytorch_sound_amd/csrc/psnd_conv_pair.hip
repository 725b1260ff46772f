// psnd_conv_pair.hip - TWO chained Conv1d (3, 7 or 11 taps) of a residual pair in ONE launch, for gfx950:
//      mid = leaky( mask1( conv(A; W1) + bias1 ) )            (kept in LDS, also written out: the backward needs it)
//      out = mask2( conv(mid; W2) + bias2 ) + res              -> out_raw, out_act = leaky(out)
// which is `leaky -> conv(d) -> leaky -> conv(1) -> + x` of hifi_gan.py:56-62 in the CL layout of psnd_conv.hip (forward: A = the
// activated input, no masks), and - with the transposed packs, mirrored taps and the leaky' masks - the input-gradient chain of the
// same pair (A = g_out, mask1 = the activated mid, mask2 = the activated input, res = g_out).
//
// Why: at the config-2 size one conv launch is a latency chain (psnd_conv.hip): of a workgroup's 16.5 k cycles, 5.4 k
// pass before its first stage is in LDS and 3.3 k in the epilogue.  Here a workgroup owns 32 (or 64) rows of `mid` over ALL channels:
// the input tile (rows + 2 h1, all of K) is fetched in one burst and stays in LDS, `mid` never leaves the chip between the two convs,
// and there is one prologue and one store epilogue per PAIR.  What bounds it (tools/trace_pair.py, s_memtime stamps): every workgroup
// streams BOTH weight packs (768 KB at C = 256) through its CU's 64 B/clk vector-memory path - 12 k cycles - so the row tile is chosen
// to put a workgroup on as many CUs as there are (32-row tiles at the config-2 size: 197 workgroups, 20.7 k cycles each, 13.1 us per
// launch back to back against 2 x 11.7 us for two launches).
//
// Tiling: 512 threads = 8 waves; at C = 256 wave w owns output channels [32 w, 32 w + 32) of both convs for all MR row blocks of 32, at
// C = 128 the waves are 4 column groups x 2 row blocks.  The workgroup's output rows are [r0, r0 + 32 MR - 2 h2), h the tap reach of a
// conv; mid rows start h2 earlier, input rows h1 earlier still.  B fragments come straight from L2 in the fragment-ordered packs of
// psnd_conv.hip (pack_index: 1 KB contiguous per wave and k-step), a ring of RU = 12 units (4 k-steps x 3 taps) ahead; the A
// fragments of a unit are read from LDS two units ahead.  The ring loop is laid out by hand (sched_group_barrier): see run_conv.
#include "psnd_conv_pair.h"

#include <atomic>
extern std::atomic<long long> g_conv_pair_stats[4];     // psnd_conv.hip (psnd_conv_pair_stats)

namespace {
using namespace pairk;
template <int C, int MR, bool HASM1, int KT = 3>
__global__ __launch_bounds__(512, 1) void conv_pair_kernel(PairParams p) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    // ring depth: 4 k-steps of 3 taps, 2 of 7, 1 of 11 - 11 ... 14 KB of weight fragments in flight per wave
    constexpr int RU = KT == 3 ? (C == 32 ? 6 : RU8) : (KT == 7 ? 14 : 11);      // (32 channels: two k-steps in all)
    conv_pair_body<C, MR, HASM1, 8, RU, KT, (KT == 3 ? HMAXP : HMAXW)>(p, blockIdx.x, smem);
}
// rows of `mid` per workgroup: 64, or 128 at 64 channels (the eight waves are 2 column groups x 4 row blocks there)
inline int pair_rows(int C) { return C == 32 ? 256 : (C == 64 ? 128 : 64); }
}  // namespace

// Round 5: beside the 3-tap pairs (C = 128 / 256, both convs reach <= 8) the 7- and 11-tap pairs of a HiFi-GAN stage's other two resblocks
// (hifi_gan.py:32-63: kernel sizes 3 / 7 / 11, dilations 1 / 3 / 5 then 1) at 32 / 64 / 128 / 256 channels and the 3-tap pair at 32 / 64, in the
// FORWARD order: the first conv may reach 25 rows, the second (dilation 1) at most 8 - a second conv with a long reach would leave few rows
// of a 64-row tile (the input-gradient order of those pairs stays on the per-conv kernels).
extern "C" int psnd_conv1d_cl_pair_supported(int C, int k, int off1, int dstep1, int off2, int dstep2) {
    if (k == 3 && (C == 128 || C == 256)) return reach3(off1, dstep1) <= HMAXP && reach3(off2, dstep2) <= HMAXP ? 1 : 0;
    if ((k == 3 && (C == 64 || C == 32)) || ((k == 7 || k == 11) && (C == 32 || C == 64 || C == 128 || C == 256)))
        return reachk(off1, dstep1, k) <= (k == 3 ? HMAXP : HMAXW) && reachk(off2, dstep2, k) <= HMAXP ? 1 : 0;
    return 0;
}

extern "C" int psnd_conv1d_cl_pair(const void *A, const void *W1, const float *bias1, const void *M1, float m1_slope, float act1_slope,
                                   void *mid_out, const void *W2, const float *bias2, const void *M2, float m2_slope, const void *res,
                                   int64_t N, int Lp, int L, int HP, int C, int k, int off1, int dstep1, int off2, int dstep2,
                                   float act2_slope, void *out_raw, void *out_act, void *stream) {
    if (!A || !W1 || !W2 || (!out_raw && !out_act)) PSND_FAIL(PSND_E_ARG, "conv1d_cl_pair: null pointer");
    if (N < 0 || Lp <= 0 || L <= 0 || HP < 0 || Lp < L + HP) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_pair: N=%lld Lp=%d L=%d HP=%d", (long long)N, Lp, L, HP);
    if (!psnd_conv1d_cl_pair_supported(C, k, off1, dstep1, off2, dstep2))
        PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_pair: C=%d k=%d taps (%d,%d) (%d,%d): k = 3 / 7 / 11 at C = 32 / 64 / 128 / 256, reach <= %d (first conv of a "
                  "7- / 11-tap pair: %d)", C, k, off1, dstep1, off2, dstep2, HMAXP, HMAXW);
    const bool wide = !(k == 3 && C >= 128);           // the instances of round 5: no masked (input-gradient) form
    if (wide && (M1 || M2)) PSND_FAIL(PSND_E_UNSUPPORTED, "conv1d_cl_pair: C=%d k=%d: the masked (input-gradient) form exists for k = 3, C = 128 / 256 only", C, k);
    if (N == 0) return PSND_OK;
    if ((size_t)N * Lp * C * 2 >= ((size_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "conv1d_cl_pair: operand larger than 2 GB");
    PairParams p;
    p.A = static_cast<const bf16_t *>(A), p.W1 = static_cast<const bf16_t *>(W1), p.W2 = static_cast<const bf16_t *>(W2);
    p.bias1 = bias1, p.bias2 = bias2, p.M1 = static_cast<const bf16_t *>(M1), p.M2 = static_cast<const bf16_t *>(M2);
    p.res = static_cast<const bf16_t *>(res), p.mid_out = static_cast<bf16_t *>(mid_out);
    p.out_raw = static_cast<bf16_t *>(out_raw), p.out_act = static_cast<bf16_t *>(out_act);
    p.R = N * (int64_t)Lp, p.Lp = Lp, p.L = L, p.HP = HP;
    p.off1 = off1, p.dstep1 = dstep1, p.h1 = reachk(off1, dstep1, k), p.off2 = off2, p.dstep2 = dstep2, p.h2 = reachk(off2, dstep2, k);
    p.m1_slope = m1_slope, p.m2_slope = m2_slope, p.act1_slope = act1_slope, p.act2_slope = act2_slope;
    p.trace = nullptr;
#ifdef PSND_TRACE      // tools/trace_pair.py builds only
    {
        const char *tp = PSND_ENV("PSND_PAIR_TRACE_PTR");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
    }
#endif
    // 64-row tiles when they fill the chip; at the config-2 size (95 of them) 32-row tiles: twice the CUs share the stores
    int MR = pair_rows(C) / 32;
    if (wide && (C == 64 || C == 128)) {
        // twice the rows per workgroup (every weight fragment feeds two MFMAs of its wave) once that still leaves a workgroup for every CU
        const int big = 2 * MR;
        const char *e = PSND_ENV("PSND_PAIR_BIG");
        const int64_t tiles_big = (p.R + (32 * big - 2 * p.h2) - 1) / (32 * big - 2 * p.h2);
        if (e ? atoi(e) != 0 : tiles_big >= 256) MR = big;
    }
    if (!wide) {
        if (C == 256 && (p.R + (64 - 2 * p.h2) - 1) / (64 - 2 * p.h2) < 256 && 32 - 2 * p.h2 >= 16) MR = 1;
        if (const char *e = PSND_ENV("PSND_PAIR_MR")) MR = (atoi(e) == 1 && C == 256 && 32 - 2 * p.h2 >= 8) ? 1 : 2;
    }
    const int MROWS = 32 * MR, TS = MROWS - 2 * p.h2;
    const int64_t tiles = (p.R + TS - 1) / TS;
    const int RS = C + 8;
    size_t lds = (size_t)(MROWS + 2 * p.h1 + MROWS + 2 * p.h2) * RS * 2;
    const size_t lds_o = (size_t)MROWS * (C + 8) * 4;
    if (lds < lds_o) lds = lds_o;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kern) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, st, p);
    };
    if (wide) {
        if (C == 256 && k == 7) launch(conv_pair_kernel<256, 2, false, 7>);
        else if (C == 256) launch(conv_pair_kernel<256, 2, false, 11>);
        else if (C == 128 && k == 7 && MR == 4) launch(conv_pair_kernel<128, 4, false, 7>);
        else if (C == 128 && MR == 4) launch(conv_pair_kernel<128, 4, false, 11>);
        else if (C == 128 && k == 7) launch(conv_pair_kernel<128, 2, false, 7>);
        else if (C == 128) launch(conv_pair_kernel<128, 2, false, 11>);
        else if (C == 32 && k == 3) launch(conv_pair_kernel<32, 8, false, 3>);
        else if (C == 32 && k == 7) launch(conv_pair_kernel<32, 8, false, 7>);
        else if (C == 32) launch(conv_pair_kernel<32, 8, false, 11>);
        else if (k == 3 && MR == 8) launch(conv_pair_kernel<64, 8, false, 3>);
        else if (k == 7 && MR == 8) launch(conv_pair_kernel<64, 8, false, 7>);
        else if (MR == 8) launch(conv_pair_kernel<64, 8, false, 11>);
        else if (k == 3) launch(conv_pair_kernel<64, 4, false, 3>);
        else if (k == 7) launch(conv_pair_kernel<64, 4, false, 7>);
        else launch(conv_pair_kernel<64, 4, false, 11>);
    } else if (C == 256 && MR == 1) {
        if (M1) launch(conv_pair_kernel<256, 1, true>);
        else launch(conv_pair_kernel<256, 1, false>);
    } else if (C == 256) {
        if (M1) launch(conv_pair_kernel<256, 2, true>);
        else launch(conv_pair_kernel<256, 2, false>);
    } else {
        if (M1) launch(conv_pair_kernel<128, 2, true>);
        else launch(conv_pair_kernel<128, 2, false>);
    }
    PSND_CHECK_LAUNCH("conv1d_cl_pair");
    g_conv_pair_stats[MR == 1 ? 0 : 1]++;
    return PSND_OK;
}
