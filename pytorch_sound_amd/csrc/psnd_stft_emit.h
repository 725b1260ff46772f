// psnd_stft_emit.h - the real-FFT split + output sweep shared by the span-staged kernels (psnd_stft.hip) and the wave-per-four-frames
// kernel (psnd_stft_q.hip).
#pragma once
#include "psnd_pk.h"
#include "psnd_stft_pass.h"

namespace psnd_stft {

// Output of one bin in registers (only the requested members are ever touched).
struct OutVal {
    float m, ph, re, im;
};

// real-FFT split of the two butterflies a thread holds + output (packed twin of post_emit): lower bins are
// stored at once, their mirrors are parked (already reduced to the values to store) and written
// afterwards in ASCENDING row order.
template <int R1, int L, class EmitT>
__device__ __forceinline__ void post_emit_pk(v2f (&za)[L], v2f (&zb)[L], bool special, int qA, int qB, const float *s_vk,
                                             const EmitT &emit, int iF, int col) {
    constexpr int LB = ct::ilog2(L);
    const int stepF = R1 * iF * 4;
    const int offA = qA * iF * 4 + col, offB = qB * iF * 4 + col;
    auto vk = [&](int k) __attribute__((always_inline)) { return *reinterpret_cast<const v2f *>(s_vk + 2 * k); };
    // The pair of a `special` lane (qA = 0, qB = R1/2) is two SELF-paired rows: row 0 pairs p with L - p (p = 0 gives X[0] and
    // X[C], p = L/2 the middle bin), row R1/2 pairs p with L-1-p.  A second code path for them made the wave holding those
    // 16 lanes run the whole split twice (10 % of the kernel: one SIMD per CU carried 1.45x the work).  Instead the
    // registers of those lanes are re-arranged so that the general code below computes exactly their outputs:
    //   A' = [ row0[0 .. L/2) | rowH[L/2 .. L) ],   B' = [ rowH[0 .. L/2) | row0[(p + 1) % L] for p in [L/2, L) ]
    //   first evaluation  f(A'[pp], B'[L-1-pp]) = f(row0[pp], row0[(L - pp) % L])  -> bins R1 pp        | R1 (L - pp)
    //   second evaluation f(B'[pp], A'[L-1-pp]) = f(rowH[pp], rowH[L-1-pp])        -> bins R1/2 + R1 pp | R1/2 + R1 (L-1-pp)
    // only the mirror rows differ (offsets below) and the middle bin of row 0 is one extra evaluation.
    v2f mid = za[ct::bitrev(L / 2, LB)];
    if (__builtin_amdgcn_ballot_w64(special) != 0) {                 // wave-uniform: only the wave that holds special lanes
        v2f ta[L / 2];
        static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {    // row0[(p + 1) % L], p = L/2 + i
            constexpr int p = L / 2 + decltype(pc)::value;
            ta[decltype(pc)::value] = za[ct::bitrev((p + 1) % L, LB)];
        });
        static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int i = decltype(pc)::value, sl = ct::bitrev(L / 2 + i, LB);
            const v2f oa = za[sl], ob = zb[sl];
            za[sl] = special ? ob : oa;
            zb[sl] = special ? ta[i] : ob;
        });
    }
    const int off1m = special ? R1 * iF * 4 + col : offB;           // mirror rows of the first / second evaluation
    const int off2m = special ? offB : offA;
    OutVal h1[L / 2], h2[L / 2];
    v2f xk, xc;
    static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pp = decltype(pc)::value;
        constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
        rfft_pair_pk(za[sa], zb[sb], vk(qA + R1 * pp), xk, xc);      // bins qA + R1 pp | qB + R1 (L-1-pp)
        emit.store(offA, pp * stepF, emit.template make<false>(xk));
        h1[pp] = emit.template make<true>(xc);
        rfft_pair_pk(zb[sa], za[sb], vk(qB + R1 * pp), xk, xc);      // bins qB + R1 pp | qA + R1 (L-1-pp)
        emit.store(offB, pp * stepF, emit.template make<false>(xk));
        h2[pp] = emit.template make<true>(xc);
        if constexpr (pp == L / 2 - 1) {
            if (special) {                                           // middle bin C/2 of row 0 (self-paired), in sweep order
                rfft_pair_pk(mid, mid, vk(R1 * (L / 2)), xk, xc);
                emit.store(col, (L / 2) * stepF, emit.template make<false>(xk));
            }
        }
    });
    static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pp = L / 2 - 1 - decltype(pc)::value;
        emit.store(off2m, (L - 1 - pp) * stepF, h2[pp]);
        emit.store(off1m, (L - 1 - pp) * stepF, h1[pp]);
    });
}

}  // namespace psnd_stft
