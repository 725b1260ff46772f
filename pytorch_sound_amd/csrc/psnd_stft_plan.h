// psnd_stft_plan.h - STFT plan layout shared by the forward / backward / inverse kernels.
#pragma once
#include "psnd_common.h"

namespace psnd_stft {

// ---------------------------------------------------------------------------------------------
// plan layout (floats), for a (R1, L) decomposition of C = n/2 complex points, C = R1 * L:
//   [0, L*ROW)            wt[l][2a+c] = 0.5 * win[2(l + L a) + c]          ROW = 2*R1 + 4
//   [L*ROW, 2 L*ROW)      tw[l][2q+{0,1}] = (cos, -sin)(2 pi l q / C)
//   [2 L*ROW, +VKP)       vk[k] = (-sin, -cos)(2 pi k / n), k = 0..C/2      VKP = round4(2(C/2+1))
//   [.., +n)              win[n] raw analysis window
//   [.., +n)              wtg[i][l][4] = 0.5*win at taps 2(l+L*2i), +1, 2(l+L*(2i+1)), +1  (window for
//                         kernels that read it from global/L1 with one coalesced 16-B piece per lane)
// Sizes without a tuned decomposition use plan = win[n] only.
// ---------------------------------------------------------------------------------------------
struct Decomp {
    int n_fft, R1, L;
};
constexpr Decomp kDecomp[] = {{256, 16, 8}, {512, 16, 16}, {1024, 32, 16}, {2048, 32, 32}};

inline const Decomp *find_decomp(int n_fft) {
    for (const Decomp &d : kDecomp)
        if (d.n_fft == n_fft) return &d;
    return nullptr;
}
inline int round4(int x) { return (x + 3) & ~3; }
inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline bool generic_ok(int n_fft) { return is_pow2(n_fft) && n_fft >= 16 && n_fft <= 8192; }

struct PlanLayout {
    int row, tab, vk, vkp, win, wtg, total;  // offsets in floats
};
inline PlanLayout plan_layout(int n_fft, int R1, int L) {
    PlanLayout p;
    int C = n_fft / 2;
    p.row = 2 * R1 + 4;
    p.tab = L * p.row;
    p.vk = 2 * p.tab;
    p.vkp = round4(2 * (C / 2 + 1));
    p.win = p.vk + p.vkp;
    p.wtg = p.win + n_fft;
    p.total = p.wtg + n_fft;
    return p;
}

// za = Z'[k], zb = Z'[C-k], v = v_k  ->  X[k] = S + E,  X[C-k] = conj(S - E)      (forward split)
__device__ __forceinline__ void rfft_pair(float zar, float zai, float zbr, float zbi, float vr, float vi,
                                          float &xkr, float &xki, float &xcr, float &xci) {
    const float sr = zar + zbr, si = zai - zbi;
    const float dr = zar - zbr, di = zai + zbi;
    const float er = __builtin_fmaf(vr, dr, -vi * di);
    const float ei = __builtin_fmaf(vr, di, vi * dr);
    xkr = sr + er;
    xki = si + ei;
    xcr = sr - er;
    xci = ei - si;
}

// ha = H[k], hb = H[C-k], v = v_k -> Zs[k] = S + conj(v) D, Zs[C-k] = conj(S - conj(v) D)  (adjoint split)
__device__ __forceinline__ void irfft_pair(float har, float hai, float hbr, float hbi, float vr, float vi,
                                           float &zkr, float &zki, float &zcr, float &zci) {
    const float sr = har + hbr, si = hai - hbi;
    const float dr = har - hbr, di = hai + hbi;
    const float er = __builtin_fmaf(vr, dr, vi * di);
    const float ei = __builtin_fmaf(vr, di, -vi * dr);
    zkr = sr + er;
    zki = si + ei;
    zcr = sr - er;
    zci = ei - si;
}

}  // namespace psnd_stft
