// psnd_pk.h - packed-fp32 complex arithmetic for gfx950.
//
// The fp32 vector peak of CDNA3/4 (157 TFLOP/s on MI355X) is only reachable with the packed
// instructions v_pk_{add,mul,fma}_f32, which process two floats per lane per issue slot; the scalar
// forms run at half that.  A complex number lives in one 64-bit VGPR pair (re, im): a complex add is
// ONE instruction, a multiply by a compile-time twiddle TWO (v_pk_mul + v_pk_fma, the re/im swap
// rides in the op_sel modifier), a multiply by -i is folded into the following butterfly.
#pragma once
#include "psnd_common.h"

typedef float v2f __attribute__((ext_vector_type(2)));

namespace pk {

__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ v2f fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f lo(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ v2f hi(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// a * w for RUNTIME w = (wx, wy):  (ax wx - ay wy, ax wy + ay wx).  The second instruction needs a swapped
// first source AND a one-sided negation of the broadcast w.y - hipcc folds the swap but not the half
// negation (it emits v_xor + v_mov instead), hence the explicit modifiers.
__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
    v2f t = a * __builtin_shufflevector(w, w, 0, 0);
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w)
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f w) {
    v2f t = a * __builtin_shufflevector(w, w, 0, 0);
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}

// ---------------------------------------------------------------------------------------
// in-register complex FFT of compile-time size R, forward sign, decimation in frequency, fully
// unrolled; natural-order input, X[q] lands in slot bitrev(q) (same contract as fft_inreg).
// A twiddle of -i is not applied where it arises (stage 2H, J = H): that slot is the `b` operand of a
// twiddle-free butterfly of the next stage, which absorbs the rotation in its two v_pk_fma.
// SIGN = +1 gives the inverse-sign transform (e^{+i..}).
// ---------------------------------------------------------------------------------------
// the twiddle of butterfly (H, J) on the difference t (1, or -i left pending: nothing)
template <int R, int H, int BLK, int J, int SIGN>
__device__ __forceinline__ void bfly_tail(v2f (&z)[R], v2f t) {
    constexpr int i1 = BLK + J + H;
    constexpr float sg = SIGN > 0 ? -1.f : 1.f;
    if constexpr (J == 0 || 2 * J == H) {
        z[i1] = t;   // 1, or -i left pending
    } else if constexpr (4 * J == H) {           // forward (1 - i)/sqrt2: r * (tx + ty, ty - tx)
        constexpr float r = (float)ct::cos2pi(1, 8);
        z[i1] = fma(swp(t), v2f{sg, -sg}, t) * v2f{r, r};
    } else if constexpr (4 * J == 3 * H) {       // forward (-1 - i)/sqrt2: -r * (tx - ty, ty + tx)
        constexpr float r = (float)ct::cos2pi(1, 8);
        z[i1] = fma(swp(t), v2f{-sg, sg}, t) * v2f{-r, -r};
    } else {                                      // forward t * (c - i s) = (tx c + ty s, ty c - tx s)
        constexpr float c = (float)ct::cos2pi(J, 2 * H);
        constexpr float s = sg * (float)ct::sin2pi(J, 2 * H);
        z[i1] = fma(swp(t), v2f{s, -s}, t * v2f{c, c});
    }
}

template <int R, int H, int BLK, int J, int SIGN>
__device__ __forceinline__ void bfly(v2f (&z)[R]) {
    constexpr int i0 = BLK + J, i1 = BLK + J + H;
    constexpr bool pending = (2 * H < R) && J == 0 && (BLK % (4 * H)) == 2 * H;   // b carries an unapplied -i (or +i)
    constexpr float sg = SIGN > 0 ? -1.f : 1.f;
    const v2f a = z[i0], b = z[i1];
    v2f t;
    if constexpr (pending) {
        // forward: b' = -i b = (b.y, -b.x)
        const v2f bs = swp(b);
        z[i0] = fma(bs, v2f{sg, -sg}, a);
        t = fma(bs, v2f{-sg, sg}, a);
    } else {
        z[i0] = a + b;
        t = a - b;
    }
    bfly_tail<R, H, BLK, J, SIGN>(z, t);
}

// first stage (H = R/2) of fft<R> on WINDOWED data, the per-element real window (wa for z[J], wb for z[J + R/2]; one weight per
// component) folded into the butterfly: a wa +- b wb as one multiply and two fused multiply-adds instead of two multiplies and two adds
template <int R, int J, int SIGN = -1>
__device__ __forceinline__ void bfly_windowed(v2f (&z)[R], v2f wa, v2f wb) {
    constexpr int H = R / 2;
    const v2f aw = z[J] * wa, b = z[J + H];
    z[J] = fma(b, wb, aw);
    bfly_tail<R, H, 0, J, SIGN>(z, fma(b, -wb, aw));
}

template <int R, int H, int SIGN>
__device__ __forceinline__ void stage(v2f (&z)[R]) {
    if constexpr (H >= 1) {
        static_for<0, R / (2 * H)>([&](auto bc) __attribute__((always_inline)) {
            constexpr int blk = decltype(bc)::value * 2 * H;
            static_for<0, H>([&](auto jc) __attribute__((always_inline)) { bfly<R, H, blk, decltype(jc)::value, SIGN>(z); });
        });
        stage<R, H / 2, SIGN>(z);
    }
}

template <int R, int SIGN = -1>
__device__ __forceinline__ void fft(v2f (&z)[R]) {
    stage<R, R / 2, SIGN>(z);
}

// ---------------------------------------------------------------------------------------
// decimation in TIME: input element p sits in slot bitrev(p), output X[q] lands in slot q (natural order) - the
// counterpart of fft<> for data that an earlier DIF transform (or an in-place pairing) left in bit-reversed slots.
// SIGN = -1 forward (e^{-i..}), +1 inverse.  Twiddle on the second operand before the butterfly; +-i is folded
// into the two v_pk_fma of the butterfly.
// ---------------------------------------------------------------------------------------
template <int R, int H, int BLK, int J, int SIGN>
__device__ __forceinline__ void bfly_dit(v2f (&z)[R]) {
    constexpr int i0 = BLK + J, i1 = BLK + J + H;
    constexpr float sg = SIGN > 0 ? 1.f : -1.f;       // w = c + i sg s,  (c, s) = (cos, sin)(2 pi J / (2H))
    const v2f a = z[i0], b = z[i1];
    if constexpr (J == 0) {
        z[i0] = a + b;
        z[i1] = a - b;
    } else if constexpr (2 * J == H) {                // w = sg * i: t = sg * (-b.y, b.x)
        const v2f bs = swp(b);
        z[i0] = fma(bs, v2f{-sg, sg}, a);
        z[i1] = fma(bs, v2f{sg, -sg}, a);
    } else {                                          // t = b * w = (bx c - sg by s, by c + sg bx s)
        constexpr float c = (float)ct::cos2pi(J, 2 * H);
        constexpr float s = sg * (float)ct::sin2pi(J, 2 * H);
        const v2f t = fma(swp(b), v2f{-s, s}, b * v2f{c, c});
        z[i0] = a + t;
        z[i1] = a - t;
    }
}

template <int R, int H, int SIGN>
__device__ __forceinline__ void stage_dit(v2f (&z)[R]) {
    if constexpr (H < R) {
        static_for<0, R / (2 * H)>([&](auto bc) __attribute__((always_inline)) {
            constexpr int blk = decltype(bc)::value * 2 * H;
            static_for<0, H>([&](auto jc) __attribute__((always_inline)) { bfly_dit<R, H, blk, decltype(jc)::value, SIGN>(z); });
        });
        stage_dit<R, 2 * H, SIGN>(z);
    }
}

template <int R, int SIGN = -1>
__device__ __forceinline__ void fft_dit(v2f (&z)[R]) {
    stage_dit<R, 1, SIGN>(z);
}

}  // namespace pk

// za = Z'[k], zb = Z'[C-k], v = v_k  ->  xk = X[k] = S + E,  xc = S - E with X[C-k] = conj(xc)
__device__ __forceinline__ void rfft_pair_pk(v2f za, v2f zb, v2f v, v2f &xk, v2f &xc) {
    const v2f s = pk::fma(zb, v2f{1.f, -1.f}, za);     // za + conj(zb)
    const v2f d = pk::fma(zb, v2f{-1.f, 1.f}, za);     // za - conj(zb)
    const v2f e = pk::cmul(d, v);
    xk = s + e;
    xc = s - e;
}
