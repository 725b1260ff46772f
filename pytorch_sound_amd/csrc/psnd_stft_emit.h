// psnd_stft_emit.h - the real-FFT split + output sweep shared by the span-staged kernels (psnd_stft.hip) and the wave-per-four-frames
// kernel (psnd_stft_q.hip).
#pragma once
#include "psnd_pk.h"
#include "psnd_stft_pass.h"
#include <type_traits>

namespace psnd_stft {

// Output of one bin in registers (only the requested members are ever touched).
struct OutVal {
    float m, ph, re, im;
};

// An emitter that only wants magnitudes may define `pair_mag(za, zb, v) -> (|X[k]|, |X[C-k]|)`: the last butterfly of the split then
// produces the real parts of both bins in one register pair and the imaginary parts in another, so that the two squared magnitudes are
// ONE packed multiply + ONE packed fused multiply-add (no cross-half add per bin).
template <class T, class = void>
struct emits_pair_mag : std::false_type {};
template <class T>
struct emits_pair_mag<T, std::void_t<decltype(T::kPairMag)>> : std::true_type {};

// za = Z'[k], zb = Z'[C-k], v = v_k -> (|X[k]|^2 + eps, |X[C-k]|^2 + eps)
__device__ __forceinline__ v2f rfft_pair_sq(v2f za, v2f zb, v2f v, v2f eps) {
    const v2f s = pk::fma(zb, v2f{1.f, -1.f}, za);     // za + conj(zb)
    const v2f d = pk::fma(zb, v2f{-1.f, 1.f}, za);     // za - conj(zb)
    const v2f e = pk::cmul(d, v);
    v2f re, im;                                         // (Re X[k], Re X[C-k]) = (s.x + e.x, s.x - e.x), (Im X[k], -Im X[C-k]) = (s.y + e.y, s.y - e.y)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(re) : "v"(s), "v"(e));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(im) : "v"(s), "v"(e));
    return pk::fma(re, re, pk::fma(im, im, eps));
}

// the split twiddles v_k read from the LDS table where they are used (post_emit_pk); a caller that has them in registers already passes
// its own source to post_emit_pk_vk: a(pp) = v[qA + R1 pp], b(pp) = v[qB + R1 pp] (pp an integral_constant), mid() = v[R1 L / 2]
template <int R1, int L>
struct VkFromLds {
    const float *s_vk;
    int qA, qB;
    template <class PC>
    __device__ __forceinline__ v2f a(PC) const { return *reinterpret_cast<const v2f *>(s_vk + 2 * (qA + R1 * PC::value)); }
    template <class PC>
    __device__ __forceinline__ v2f b(PC) const { return *reinterpret_cast<const v2f *>(s_vk + 2 * (qB + R1 * PC::value)); }
    __device__ __forceinline__ v2f mid() const { return *reinterpret_cast<const v2f *>(s_vk + 2 * (R1 * (L / 2))); }
};

// real-FFT split of the two butterflies a thread holds + output (packed twin of post_emit): lower bins are
// stored at once, their mirrors are parked (already reduced to the values to store) and written
// afterwards in ASCENDING row order.
template <class T, class = void>
struct vk_has_hooks : std::false_type {};
template <class T>
struct vk_has_hooks<T, std::void_t<decltype(T::kHooks)>> : std::true_type {};

// A twiddle source with `kHooks` is told where the sweep stands: begin() in front of the first evaluation (behind the register shuffle of
// the special lanes), before(pp) in front of evaluation pair pp - a source that holds the values in registers issues and awaits its own
// LDS reads there.
template <int R1, int L, class EmitT, class VkT>
__device__ __forceinline__ void post_emit_pk_vk(v2f (&za)[L], v2f (&zb)[L], bool special, int qA, int qB, VkT &vk,
                                                const EmitT &emit, int iF, int col) {
    constexpr int LB = ct::ilog2(L);
    const int stepF = R1 * iF * 4;
    const int offA = qA * iF * 4 + col, offB = qB * iF * 4 + col;
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
    if constexpr (vk_has_hooks<VkT>::value) vk.begin();
    static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pp = decltype(pc)::value;
        constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
        if constexpr (vk_has_hooks<VkT>::value) vk.before(pc);
        if constexpr (emits_pair_mag<EmitT>::value) {
            OutVal o;
            v2f m = emit.pair_mag(za[sa], zb[sb], vk.a(pc));         // bins qA + R1 pp | qB + R1 (L-1-pp)
            o.m = m.x, h1[pp].m = m.y;
            emit.store(offA, pp * stepF, o);
            m = emit.pair_mag(zb[sa], za[sb], vk.b(pc));             // bins qB + R1 pp | qA + R1 (L-1-pp)
            o.m = m.x, h2[pp].m = m.y;
            emit.store(offB, pp * stepF, o);
        } else {
            rfft_pair_pk(za[sa], zb[sb], vk.a(pc), xk, xc);          // bins qA + R1 pp | qB + R1 (L-1-pp)
            emit.store(offA, pp * stepF, emit.template make<false>(xk));
            h1[pp] = emit.template make<true>(xc);
            rfft_pair_pk(zb[sa], za[sb], vk.b(pc), xk, xc);          // bins qB + R1 pp | qA + R1 (L-1-pp)
            emit.store(offB, pp * stepF, emit.template make<false>(xk));
            h2[pp] = emit.template make<true>(xc);
        }
        if constexpr (pp == L / 2 - 1) {
            if (special) {                                           // middle bin C/2 of row 0 (self-paired), in sweep order
                rfft_pair_pk(mid, mid, vk.mid(), xk, xc);
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

template <int R1, int L, class EmitT>
__device__ __forceinline__ void post_emit_pk(v2f (&za)[L], v2f (&zb)[L], bool special, int qA, int qB, const float *s_vk,
                                             const EmitT &emit, int iF, int col) {
    VkFromLds<R1, L> vk{s_vk, qA, qB};
    post_emit_pk_vk<R1, L>(za, zb, special, qA, qB, vk, emit, iF, col);
}

}  // namespace psnd_stft
