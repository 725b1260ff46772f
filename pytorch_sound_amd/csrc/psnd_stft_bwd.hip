// psnd_stft_bwd.hip - adjoint of the STFT stage for gfx950: what autograd computes through
// F.pad(reflect) + F.conv1d(forward_basis) + sqrt(re^2+im^2) (pytorch_sound/models/transforms.py:55-69)
// and through torch.stft (transforms.py:297-311), without the dense basis.
//
//   gwav[t] += win[m] * Re sum_{k=0..C} G[k] e^{+2 pi i k m / n}      for every (frame f, tap m) that
//              reads sample t (reflect map), G = gre + i gim (+ gmag * X/|X| with X recomputed).
//
// It is the forward kernel run backwards on the same tiling (psnd_stft.hip):
//   phase 0  (gmag only) forward pass 1 -> exchange, to recompute X
//   phase 1  thread (frame, bin pair): forward radix-L + real split -> X[k], X[C-k]; G; adjoint
//            split -> Zs[k], Zs[C-k] (same two butterflies); inverse radix-L in VGPRs; conj twiddle;
//            written back over the thread's own two exchange rows
//   phase 2  lane (frame, l): inverse radix-R1 over q in VGPRs -> the 2*R1 windowed time samples of
//            the lane; overlap-add of the tile's FT frames in LDS (ds_add_f32 into a span buffer that
//            aliases the exchange area), then ONE pass over the span: samples completed inside the
//            tile are stored plainly (coalesced), the n-hop head/tail and reflected samples are
//            atomically added to the zero-initialised gwav.
#include "psnd_stft_pass.h"
#include "psnd_pk.h"
#include <math.h>

namespace {
using namespace psnd_stft;

struct StftBwdParams {
    const float *wav, *plan, *gmag, *gre, *gim;
    float *gwav;
    long long T, F;
    int hop, pad, ntile, total_tiles;
    float mag_eps;
    // inverse-STFT mode (psnd_istft): gmag = magnitude, gre = phase, gwav = output (N, T = (F-1)*hop)
    int win_off;        // float offset of the raw window inside the plan
    float inv_n, env_eps;
    // multi_stft_loss mode (psnd_stft_bwd_msl): gmag = TARGET magnitudes; the gradient of the loss w.r.t. the magnitude is formed
    // from the recomputed |X| inside the kernel (loss_bwd_kernel of psnd_loss.hip): norms = (||t - p||, ||t||) per clip, g = upstream
    const float *msl_norms, *msl_g;
    float msl_invLN, msl_invLNKF, msl_eps;
    int msl_accumulate;   // add to gwav (the other resolutions' gradient) instead of overwriting it
#ifdef PSND_TRACE
    long long *trace;
#endif
};
#ifdef PSND_TRACE
#define PSND_BSTAMP(i_)                                                                              \
    do {                                                                                             \
        if ((threadIdx.x & 63) == 0 && p.trace)                                                      \
            p.trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PSND_BSTAMP(i_)
#endif

// The gradient (or, for the inverse transform, the spectrum) values of a thread's bins are loaded into registers BEFORE
// anything is computed from them: read one at a time where they are used, the 33-66 dependent global loads of a thread
// were a serial latency chain (~55 us per tile at one wave per SIMD).
struct GVal {
    float a, b;      // FROM_MAG: a = gmag ; FROM_REIM: a = gre (+ gmag in c), b = gim ; ISTFT: a = magnitude, b = phase
    float c;
};
template <bool FROM_MAG, bool FROM_REIM, bool ISTFT>
struct GLoad {
    const float *gmag, *gre, *gim;
    float eps, inv_n;
    bool valid;
    __device__ __forceinline__ GVal fetch(int off) const {
        GVal v{0.f, 0.f, 0.f};
        if (!valid) return v;
        if constexpr (ISTFT) {
            v.a = gmag[off], v.b = gre[off];
        } else {
            if constexpr (FROM_MAG) v.c = gmag[off];
            if constexpr (FROM_REIM) v.a = gre[off], v.b = gim[off];
        }
        return v;
    }
    // gradient wrt (re, im) of a bin from its preloaded values, given the recomputed X there.
    // ISTFT: the "gradient" is the spectrum itself, X = mag * e^{i phase}, scaled so that the adjoint
    // transform below IS the inverse real DFT: 1/n for the DC / Nyquist bins (`edge`), 2/n otherwise.
    __device__ __forceinline__ void operator()(const GVal &v, float xr, float xi, float &gr, float &gi, bool edge = false) const {
        gr = 0.f, gi = 0.f;
        if (!valid) return;
        if constexpr (ISTFT) {
            float sn, cs;
            sincosf(v.b, &sn, &cs);
            const float m = v.a * (edge ? inv_n : 2.f * inv_n);
            gr = m * cs;
            gi = m * sn;
            return;
        }
        if constexpr (FROM_MAG) {
            const float m = __builtin_amdgcn_sqrtf(__builtin_fmaf(xr, xr, __builtin_fmaf(xi, xi, eps)));
            const float g = v.c / m;   // 0/0 -> NaN exactly like autograd of sqrt at 0
            gr = g * xr;
            gi = g * xi;
        }
        if constexpr (FROM_REIM) {
            gr += v.a;
            gi += v.b;
        }
    }
};

// squared-window overlap-add envelope at padded sample tp (STFT.inverse, transforms.py:85-98)
__device__ __forceinline__ float ola_envelope(const float *win, long long tp, int n, int hop, long long F) {
    long long f_hi = tp / hop;
    if (f_hi > F - 1) f_hi = F - 1;
    long long f_lo = (tp - n + hop) / hop;      // ceil((tp - n + 1) / hop) for tp - n + 1 > 0
    if (tp - n + 1 <= 0) f_lo = 0;
    float e = 0.f;
    for (long long f = f_lo; f <= f_hi; ++f) {
        const float w = win[tp - f * hop];
        e = __builtin_fmaf(w, w, e);
    }
    return e;
}

__device__ __forceinline__ long long reflect64(long long i, long long T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}

template <int R1, int L, bool FROM_MAG, bool FROM_REIM, bool ISTFT = false>
__global__ __launch_bounds__(256) void stft_bwd_kernel(StftBwdParams p) {
    using G = Cfg<R1, L>;
    constexpr int C = G::C, NFFT = G::NFFT, FT = G::FT, P1R = G::P1R, LB = G::LB, RB = G::RB, SF = G::SF, ROW = G::ROW;

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

    const int span_len = (FT - 1) * p.hop + NFFT;
    // overlap-add inside the tile by gather (phase 2); false = every sample straight to global atomics (round 1's path for the
    // instances with two pass-1 rounds, n = 512 / 2048, 1.76 / 1.10 ms at the multi_stft_loss shapes)
    constexpr bool use_span = true;

    const TileWalk tw = tile_walk(p.total_tiles);
    for (int tile = tw.first; tile < tw.end; tile += tw.step) {
        const int clip = tile / p.ntile;
        const long long f0 = (long long)(tile - clip * p.ntile) * FT;
        const float *x = p.wav + (size_t)clip * p.T;
        float *gw = p.gwav + (size_t)clip * p.T;

        if constexpr (FROM_MAG) {
#pragma unroll 1
            for (int r = 0; r < P1R; ++r) {
                const int task = r * 256 + t;
                fwd_pass1<R1, L>(s, x, p.T, p.F, f0, p.hop, p.pad, task / L, task % L);
            }
            __syncthreads();
        }

        // ------------------------------- phase 1 ------------------------------------------
        {
            float ar[L], ai[L], br[L], bi[L];      // forward butterflies (slot = bitrev(p))
            float uAr[L], uAi[L], uBr[L], uBi[L];  // adjoint inputs Zs[q + R1 p] in natural p order
            const long long F = p.F;
            const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)(f0 + f2);
            GLoad<FROM_MAG, FROM_REIM, ISTFT> gload{(FROM_MAG || ISTFT) ? p.gmag + cbase : nullptr,
                                                    (FROM_REIM || ISTFT) ? p.gre + cbase : nullptr,
                                                    FROM_REIM ? p.gim + cbase : nullptr, p.mag_eps, p.inv_n, (f0 + f2) < F};
            const int iF = (int)F;
            const int stepF = R1 * iF;
            const int offA = qA * iF, offB = qB * iF;
            // rows of this thread, all in flight now: A row bins qA + R1 pp (lo: pp < L/2, hi: the mirrored half), same
            // for the B row; the special pair (qA = 0, qB = R1/2) needs exactly the same bins plus the Nyquist one
            GVal gAl[L / 2], gAh[L / 2], gBl[L / 2], gBh[L / 2], gNy;
            static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
                constexpr int pp = decltype(pc)::value;
                gAl[pp] = gload.fetch(offA + pp * stepF);
                gBh[pp] = gload.fetch(offB + (L - 1 - pp) * stepF);
                gBl[pp] = gload.fetch(offB + pp * stepF);
                gAh[pp] = gload.fetch(offA + (L - 1 - pp) * stepF);
            });
            gNy = gload.fetch(special ? L * stepF : offA);
            if constexpr (FROM_MAG) {
                fwd_pass2_fft<R1, L>(s, f2, qA, qB, ar, ai, br, bi);
            } else {
                static_for<0, L>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    ar[i] = ai[i] = br[i] = bi[i] = 0.f;
                });
            }
            const float *s_vk = s.vk;
            float xkr = 0.f, xki = 0.f, xcr = 0.f, xci = 0.f, gkr, gki, gcr, gci;
            if (!special) {
                static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int pp = decltype(pc)::value;
                    constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
                    {   // bins k = qA + R1*pp  and  C-k = qB + R1*(L-1-pp)
                        const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (qA + R1 * pp));
                        if constexpr (FROM_MAG) rfft_pair(ar[sa], ai[sa], br[sb], bi[sb], v.x, v.y, xkr, xki, xcr, xci);
                        gload(gAl[pp], xkr, xki, gkr, gki);
                        gload(gBh[pp], xcr, xci, gcr, gci);
                        irfft_pair(gkr, gki, gcr, gci, v.x, v.y, uAr[pp], uAi[pp], uBr[L - 1 - pp], uBi[L - 1 - pp]);
                    }
                    {   // bins k = qB + R1*pp  and  C-k = qA + R1*(L-1-pp)
                        const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (qB + R1 * pp));
                        if constexpr (FROM_MAG) rfft_pair(br[sa], bi[sa], ar[sb], ai[sb], v.x, v.y, xkr, xki, xcr, xci);
                        gload(gBl[pp], xkr, xki, gkr, gki);
                        gload(gAh[pp], xcr, xci, gcr, gci);
                        irfft_pair(gkr, gki, gcr, gci, v.x, v.y, uBr[pp], uBi[pp], uAr[L - 1 - pp], uAi[L - 1 - pp]);
                    }
                });
            } else {
                // butterfly q = 0: bins R1*p and C - R1*p = R1*(L-p)
                static_for<0, L / 2 + 1>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int pp = decltype(pc)::value;
                    constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev((L - pp) % L, LB);
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (R1 * pp));
                    if constexpr (FROM_MAG) rfft_pair(ar[sa], ai[sa], ar[sb], ai[sb], v.x, v.y, xkr, xki, xcr, xci);
                    gload(pp < L / 2 ? gAl[pp < L / 2 ? pp : 0] : gAh[L / 2 - 1], xkr, xki, gkr, gki, pp == 0);
                    float z0r, z0i, z1r, z1i;
                    if constexpr (pp == 0) {
                        // H[0] = 2 Re G[0], H[C] = 2 Re G[C] (the imaginary parts of the DC / Nyquist
                        // bins do not reach the real signal)
                        gload(gNy, xcr, xci, gcr, gci, true);
                        irfft_pair(2.f * gkr, 0.f, 2.f * gcr, 0.f, v.x, v.y, z0r, z0i, z1r, z1i);
                        uAr[0] = z0r, uAi[0] = z0i;
                    } else if constexpr (2 * pp == L) {
                        irfft_pair(gkr, gki, gkr, gki, v.x, v.y, z0r, z0i, z1r, z1i);
                        uAr[pp] = z0r, uAi[pp] = z0i;
                    } else {
                        gload(gAh[(pp >= 1 && pp <= L / 2) ? pp - 1 : 0], xcr, xci, gcr, gci);      // bin R1 (L - pp) = row 0, index L-1-(pp-1)
                        irfft_pair(gkr, gki, gcr, gci, v.x, v.y, z0r, z0i, z1r, z1i);
                        uAr[pp] = z0r, uAi[pp] = z0i;
                        uAr[L - pp] = z1r, uAi[L - pp] = z1i;
                    }
                });
                // butterfly q = R1/2: bins R1/2 + R1*p and R1/2 + R1*(L-1-p)
                static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int pp = decltype(pc)::value;
                    constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (R1 / 2 + R1 * pp));
                    if constexpr (FROM_MAG) rfft_pair(br[sa], bi[sa], br[sb], bi[sb], v.x, v.y, xkr, xki, xcr, xci);
                    gload(gBl[pp], xkr, xki, gkr, gki);
                    gload(gBh[pp], xcr, xci, gcr, gci);
                    irfft_pair(gkr, gki, gcr, gci, v.x, v.y, uBr[pp], uBi[pp], uBr[L - 1 - pp], uBi[L - 1 - pp]);
                });
            }
            // inverse radix-L over p (swap trick: FFT of (im, re) = swapped inverse FFT); U[q][l] in slot bitrev(l)
            fft_inreg<L>(uAi, uAr);
            fft_inreg<L>(uBi, uBr);
            // conj twiddle W_C^{-lq} and write back over this thread's own two rows
            float *oa_r = s.xr + f2 * SF + qA * L, *oa_i = s.xi + f2 * SF + qA * L;
            float *ob_r = s.xr + f2 * SF + qB * L, *ob_i = s.xi + f2 * SF + qB * L;
            static_for<0, L / 4>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                f32x4 var, vai, vbr, vbi;
                static_for<0, 4>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int l = 4 * i + j, sl = ct::bitrev(l, LB);
                    const f32x2 ta = *reinterpret_cast<const f32x2 *>(s.tw + l * ROW + 2 * qA);
                    const f32x2 tb = *reinterpret_cast<const f32x2 *>(s.tw + l * ROW + 2 * qB);
                    // U * conj(t), t = (cos, -sin) stored -> conj = (t.x, -t.y)
                    var[j] = __builtin_fmaf(uAr[sl], ta.x, uAi[sl] * ta.y);
                    vai[j] = __builtin_fmaf(uAi[sl], ta.x, -uAr[sl] * ta.y);
                    vbr[j] = __builtin_fmaf(uBr[sl], tb.x, uBi[sl] * tb.y);
                    vbi[j] = __builtin_fmaf(uBi[sl], tb.x, -uBr[sl] * tb.y);
                });
                *reinterpret_cast<f32x4 *>(oa_r + 4 * i) = var;
                *reinterpret_cast<f32x4 *>(oa_i + 4 * i) = vai;
                *reinterpret_cast<f32x4 *>(ob_r + 4 * i) = vbr;
                *reinterpret_cast<f32x4 *>(ob_i + 4 * i) = vbi;
            });
        }
        __syncthreads();

        // ------------------------------- phase 2 ------------------------------------------
        const long long t_start = f0 * p.hop - p.pad;
#pragma unroll 1
        for (int r = 0; r < P1R; ++r) {
            const int task = r * 256 + t;
            const int fl = task / L, l = task % L;
            float zr[R1], zi[R1];
            float *ixr = s.xr + fl * SF + l, *ixi = s.xi + fl * SF + l;
            static_for<0, R1>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                zr[q] = ixr[q * L];
                zi[q] = ixi[q * L];
            });
            fft_inreg<R1>(zi, zr);   // inverse radix-R1 over q: z[l + L a] in slot bitrev(a)
            const float *wrow = s.wt + l * ROW;
            static_for<0, R1 / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
                constexpr int s0_ = ct::bitrev(2 * i, RB), s1_ = ct::bitrev(2 * i + 1, RB);
                zr[s0_] *= w.x;
                zi[s0_] *= w.y;
                zr[s1_] *= w.z;
                zi[s1_] *= w.w;
            });
            if (use_span) {
                // The lane's 2*R1 windowed samples go back IN PLACE over the 2*R1 exchange words it has just read (no other lane
                // touches them): sample m = 2 (l + L a) + c of frame fl lands in (c ? xi : xr)[fl * SF + (m >> 1)], i.e. the
                // exchange area turns into Y[frame][m] without a second buffer or a barrier per pass-1 round.
                static_for<0, R1>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = decltype(ac)::value, sl = ct::bitrev(a, RB);
                    ixr[a * L] = zr[sl];
                    ixi[a * L] = zi[sl];
                });
            } else if ((f0 + fl) < p.F) {
                const long long tb = (f0 + fl) * p.hop - p.pad + 2 * l;
                static_for<0, R1>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = decltype(ac)::value, sl = ct::bitrev(a, RB);
                    if constexpr (ISTFT) {
                        const long long t0 = tb + 2 * L * a, t1 = t0 + 1;
                        if (t0 >= 0 && t0 < p.T)
                            unsafeAtomicAdd(gw + t0, zr[sl] / (ola_envelope(p.plan + p.win_off, t0 + p.pad, NFFT, p.hop, p.F) + p.env_eps));
                        if (t1 >= 0 && t1 < p.T)
                            unsafeAtomicAdd(gw + t1, zi[sl] / (ola_envelope(p.plan + p.win_off, t1 + p.pad, NFFT, p.hop, p.F) + p.env_eps));
                    } else {
                        const long long t0 = reflect64(tb + 2 * L * a, p.T), t1 = reflect64(tb + 2 * L * a + 1, p.T);
                        if (t0 >= 0 && t0 < p.T) unsafeAtomicAdd(gw + t0, zr[sl]);
                        if (t1 >= 0 && t1 < p.T) unsafeAtomicAdd(gw + t1, zi[sl]);
                    }
                });
            }
        }
        if (use_span) {
            // overlap-add by GATHER (no LDS atomics: ds_add_f32 retires ~one lane per 12 cycles on gfx950): each span sample sums
            // its <= n/hop frames out of Y
            __syncthreads();
            const int int_lo = NFFT - p.hop, int_hi = FT * p.hop;
            const int nfr = (int)((p.F - f0) < FT ? (p.F - f0) : FT);
            const unsigned hmagic = 0xffffffffu / (unsigned)p.hop + 1u;
            for (int i = t; i < span_len; i += 256) {
                int f_hi = (int)__umulhi((unsigned)i, hmagic);
                if (f_hi > nfr - 1) f_hi = nfr - 1;
                const int above = i - NFFT + p.hop;
                const int f_lo = above > 0 ? (int)__umulhi((unsigned)above, hmagic) : 0;
                float v = 0.f;
                for (int f = f_lo; f <= f_hi; ++f) {
                    const int m = i - f * p.hop;
                    v += ((m & 1) ? s.xi : s.xr)[f * SF + (m >> 1)];
                }
                const long long tg = t_start + i;
                if constexpr (ISTFT) {
                    // no reflection: samples outside [0, T) are the trimmed n/2 margins; divide by the envelope
                    if (tg < 0 || tg >= p.T) continue;
                    // squared-window envelope from the LDS copy of the window (wt[l][2a+c] = win[2(l + L a) + c] / 2)
                    const long long tp = tg + p.pad;
                    long long e_hi = tp < (1ll << 24) ? (long long)__umulhi((unsigned)tp, hmagic) : tp / p.hop;
                    if (e_hi > p.F - 1) e_hi = p.F - 1;
                    const long long ab = tp - NFFT + p.hop;
                    const long long e_lo = ab > 0 ? (ab < (1ll << 24) ? (long long)__umulhi((unsigned)ab, hmagic) : ab / p.hop) : 0;
                    float env = 0.f;
                    for (long long f = e_lo; f <= e_hi; ++f) {
                        const int m = (int)(tp - f * p.hop), h2 = m >> 1;
                        const float w = 2.f * s.wt[(h2 % L) * ROW + 2 * (h2 / L) + (m & 1)];
                        env = __builtin_fmaf(w, w, env);
                    }
                    v /= env + p.env_eps;
                    if (i >= int_lo && i < int_hi) gw[tg] = v;
                    else if (v != 0.f) unsafeAtomicAdd(gw + tg, v);
                } else if (i >= int_lo && i < int_hi && tg > p.pad && tg < p.T - 1 - p.pad) {
                    gw[tg] = v;
                } else if (v != 0.f) {
                    const long long tr = reflect64(tg, p.T);
                    if (tr >= 0 && tr < p.T) unsafeAtomicAdd(gw + tr, v);
                }
            }
        }
        __syncthreads();   // exchange / span area is rewritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------
// n_fft = 1024, gradient of the magnitude (the path LogMelSpectrogram / magnitude losses take): the adjoint on the
// structure of stft_fwd_n1024_kernel - span staged once in LDS, packed fp32 (psnd_pk.h), every gmag value of a thread
// loaded before the first FLOP, the adjoint split written IN PLACE over the forward butterflies (element p of a row
// stays in slot bitrev(p)), inverse transforms as decimation-in-time FFTs (bit-reversed in, natural out), so the
// kernel needs ~130 VGPRs instead of 294 and two workgroups share a CU.  (The general kernel above - one wave per SIMD,
// scalar, frame taps from global memory - runs this case at 4-5 % of the HBM roof.)
// LDS: tables + one FULL exchange [16 frames][32 rows][16 l] of (re, im) (65.8 KB); the forward span, the forward
// exchange, the adjoint exchange and the overlap-add span all live in it, one after the other.
// ---------------------------------------------------------------------------------------------
// Geometry for C = R1 x 16 complex points (R1 = 32: n_fft = 1024, 16 frames per workgroup; R1 = 16: n_fft = 512, 32 frames in two
// lane-pass rounds), as SpanGeom of psnd_stft.hip but with the FULL exchange.
// (32, 32): n_fft = 2048 - 512 threads (one lane-pass round of 16 frames x 32 lanes; the 256 pair threads are waves 0..3), the full
// exchange of 16 frames is 131 KB: one workgroup of 8 waves per CU.
template <int R1, int L_ = 16>
struct BwdGeom {
    static constexpr int L = L_, NT = L == 32 ? 512 : 256;
    static constexpr int C = R1 * L, NFFT = 2 * C, FT = 512 / R1, FPR = NT / L, NR = FT / FPR, ROW = 2 * R1 + 4, TPB = 256 / (2 * L);
    static constexpr int VKP = (2 * (C / 2 + 1) + 3) & ~3;
    static constexpr int SF = R1 * L * 2 + 4;     // exchange frame stride (floats): SF/4 odd, 8*SF = 32 (mod 64)
    static constexpr int TAB = 2 * L * ROW + VKP;
    static constexpr int LDS_FLOATS = TAB + FT * SF;
    static_assert((SF / 4) % 2 == 1 && (8 * SF) % 64 == 32, "exchange pitch");
};

// ISTFT = true: the same kernel as the inverse transform of psnd_istft - no forward recompute, the "gradient" is the spectrum
// mag * e^{i phase} scaled to make the adjoint the inverse real DFT, and the overlap-added signal is divided by the squared-
// window envelope (STFT.inverse, transforms.py:71-101); no reflection (the n/2 margins are trimmed).

// sin / cos on the hardware units (v_sin_f32 / v_cos_f32 work in revolutions): |error| < 2e-6 after the fract() range
// reduction, inside the 1e-5 round-trip tolerance of the inverse transform; the libm sincosf costs ~10x the instructions.
__device__ __forceinline__ void fast_sincos(float x, float &sn, float &cs) {
    const float r = x * 0.15915494309189535f;
    const float f = r - __builtin_floorf(r);
    sn = __builtin_amdgcn_sinf(f);
    cs = __builtin_amdgcn_cosf(f);
}

template <bool ISTFT, int R1 = 32, int L_ = 16, bool MSL = false>
__global__ __launch_bounds__((L_ == 32 ? 512 : 256), (L_ == 32 ? 1 : 2)) void stft_bwd_n1024_mag_kernel(StftBwdParams p) {
    static_assert(!(ISTFT && MSL), "the fused loss gradient is a mode of the STFT adjoint");
    using G = BwdGeom<R1, L_>;
    constexpr int L = G::L, NT = G::NT, C = G::C, NFFT = G::NFFT, FT = G::FT, FPR = G::FPR, NR = G::NR, TPB = G::TPB, ROW = G::ROW, VKP = G::VKP;
    constexpr int RB = ct::ilog2(R1), LB = ct::ilog2(L), SF = G::SF;
    constexpr int TAB = G::TAB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_wt = smem, *s_tw = s_wt + L * ROW, *s_vk = s_tw + L * ROW, *s_x = s_vk + VKP;
    const int t = threadIdx.x;
    // lane identity (passes over a).  L = 16: half-waves hold frames fl, fl+8; L = 32: a half-wave is the 32 lanes of one frame
    const int l = t & (L - 1), fl = L == 16 ? ((t >> 4) & 1) * 8 + (t >> 5) : (t >> 5);
    // pair identity (passes over l): frame, row pair < R1 / 2 - the first 256 threads
    const bool pair_thread = NT == 256 || t < 256;
    const int f2 = t & (FT - 1), qq = ((t >> 6) & 3) + 4 * ((t & 63) / FT);
    const bool special = (qq == 0);
    const int qA = qq, qB = special ? R1 / 2 : R1 - qq;
    const int hop = p.hop;
    const int span_len = (FT - 1) * hop + NFFT;
    const int skew = (L == 16 && hop % 256 == 0) ? 4 : 0;

    const int chunk = (p.total_tiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || tile >= p.total_tiles) return;
    const int clip = tile / p.ntile;
    const long long f0 = (long long)(tile - clip * p.ntile) * FT;
    const float *x = p.wav + (size_t)clip * p.T;
    float *gw = p.gwav + (size_t)clip * p.T;
    const long long F = p.F;
    const int iF = (int)F;
    PSND_BSTAMP(0);

    // ---- prologue: span, tables, and every gmag value of this thread (bins qA + 32 pp, qB + 32 pp and their mirrors) ----
    {
        if constexpr (!ISTFT) {
            const long long g0 = f0 * hop - p.pad;
            const int Ti = (int)p.T;
            for (int s4 = t * 4; s4 < span_len; s4 += 4 * NT) {
                const long long g = g0 + s4;
                f32x4 v;
                if (g >= 0 && g + 3 < p.T) {
                    v = *reinterpret_cast<const f32x4_u *>(x + g);
                } else {
                    const int gi = (int)g;
                    v.x = x[reflect_idx32(gi, Ti)], v.y = x[reflect_idx32(gi + 1, Ti)];
                    v.z = x[reflect_idx32(gi + 2, Ti)], v.w = x[reflect_idx32(gi + 3, Ti)];
                }
                *reinterpret_cast<f32x4 *>(s_x + s4 + skew * (s4 >> 8)) = v;
            }
        }
        for (int i = t; i < TAB / 4; i += NT) reinterpret_cast<f32x4 *>(smem)[i] = reinterpret_cast<const f32x4 *>(p.plan)[i];
    }
    const bool fvalid2 = pair_thread && (f0 + f2) < F;
    // multi_stft_loss: d loss / d |X| = k1 (|X| - t) + c_mag sign(|X| - t) / (|X| + eps)   (sign of the log difference: log is monotonic)
    float msl_k1 = 0.f, msl_cm = 0.f;
    if constexpr (MSL) {
        const float nd = p.msl_norms[2 * clip], nt = p.msl_norms[2 * clip + 1];
        const float g0 = p.msl_g[0];
        const float c_sc = (g0 + p.msl_g[1]) * p.msl_invLN;
        msl_cm = fvalid2 ? (g0 + p.msl_g[2]) * p.msl_invLNKF : 0.f;
        msl_k1 = fvalid2 ? c_sc / (nd * nt) : 0.f;
    }
    auto msl_grad = [&](float m, float tv) __attribute__((always_inline)) {
        const float d = m - tv;
        // sign(d) without compares (selects in the divergent self-paired rows cost 64 VGPRs): +-1 for |d| >= 2^-120, 0 at 0
        const float sg = msl_cm * __builtin_amdgcn_fmed3f(d * 0x1p+120f, -1.f, 1.f);
        return __builtin_fmaf(msl_k1, d, sg * __builtin_amdgcn_rcpf(m + p.msl_eps));
    };
    const size_t goff = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)(f0 + f2);
    const float *gbase = p.gmag + goff;
    const int stepF = R1 * iF, offA = qA * iF, offB = qB * iF;
    float gAl[L / 2], gAh[L / 2], gBl[L / 2], gBh[L / 2], gNy = 0.f;
    static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pp = decltype(pc)::value;
        gAl[pp] = fvalid2 ? gbase[offA + pp * stepF] : 0.f;
        gBh[pp] = fvalid2 ? gbase[offB + (L - 1 - pp) * stepF] : 0.f;
        gBl[pp] = fvalid2 ? gbase[offB + pp * stepF] : 0.f;
        gAh[pp] = fvalid2 ? gbase[offA + (L - 1 - pp) * stepF] : 0.f;
    });
    if (special && fvalid2) gNy = gbase[L * stepF];
    // inverse transform: the phases of the same bins
    float pAl[ISTFT ? L / 2 : 1], pAh[ISTFT ? L / 2 : 1], pBl[ISTFT ? L / 2 : 1], pBh[ISTFT ? L / 2 : 1], pNy = 0.f;
    if constexpr (ISTFT) {
        const float *pbase = p.gre + goff;
        static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int pp = decltype(pc)::value;
            pAl[pp] = fvalid2 ? pbase[offA + pp * stepF] : 0.f;
            pBh[pp] = fvalid2 ? pbase[offB + (L - 1 - pp) * stepF] : 0.f;
            pBl[pp] = fvalid2 ? pbase[offB + pp * stepF] : 0.f;
            pAh[pp] = fvalid2 ? pbase[offA + (L - 1 - pp) * stepF] : 0.f;
        });
        if (special && fvalid2) pNy = pbase[L * stepF];
    }
    __syncthreads();
    PSND_BSTAMP(1);

    // ---- forward pass 1 (recompute X): taps, window, radix-32, twiddle, all 32 rows to the exchange -------------------
    if constexpr (!ISTFT) {
        v2f z[NR][R1];
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            const int sb = (fl + FPR * r) * hop + 2 * l;
            const float *tb0 = s_x + sb + skew * (sb >> 8);
            static_for<0, R1 / TPB>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                const float *tb = tb0 + g * (256 + skew);
                static_for<0, TPB>([&](auto ac) __attribute__((always_inline)) {
                    constexpr int a = TPB * g + decltype(ac)::value;
                    z[r][a] = *reinterpret_cast<const v2f *>(tb + 2 * L * (a - TPB * g));
                });
            });
            const float *wrow = s_wt + l * ROW;
            static_for<0, R1 / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
                z[r][2 * i] *= pk::lo(w);
                z[r][2 * i + 1] *= pk::hi(w);
                if constexpr (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            });
            pk::fft<R1>(z[r]);
        });
        __syncthreads();                             // every lane holds its taps: the span area becomes the exchange
        const float *trow = s_tw + l * ROW;
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            float *oz = s_x + (fl + FPR * r) * SF + 2 * l;
            static_for<0, R1 / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int q0 = 2 * decltype(ic)::value, q1 = q0 + 1;
                const f32x4 w = *reinterpret_cast<const f32x4 *>(trow + 2 * q0);
                if constexpr (q0 == 0) *reinterpret_cast<v2f *>(oz) = z[r][0];
                else *reinterpret_cast<v2f *>(oz + q0 * 2 * L) = pk::cmul(z[r][ct::bitrev(q0, RB)], pk::lo(w));
                *reinterpret_cast<v2f *>(oz + q1 * 2 * L) = pk::cmul(z[r][ct::bitrev(q1, RB)], pk::hi(w));
            });
        });
        __syncthreads();
    }
    PSND_BSTAMP(2);

    // ---- pair threads: forward radix-L -> X, G = gmag X / |X|, adjoint split in place, inverse radix-L, conj twiddle ----
    if (pair_thread) {
        v2f za[L], zb[L];
        float *rowA = s_x + f2 * SF + qA * 2 * L, *rowB = s_x + f2 * SF + qB * 2 * L;
        if constexpr (!ISTFT) {
            static_for<0, L / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const f32x4 va = *reinterpret_cast<const f32x4 *>(rowA + 4 * i), vb = *reinterpret_cast<const f32x4 *>(rowB + 4 * i);
                za[2 * i] = pk::lo(va), za[2 * i + 1] = pk::hi(va);
                zb[2 * i] = pk::lo(vb), zb[2 * i + 1] = pk::hi(vb);
            });
            pk::fft<L>(za);
            pk::fft<L>(zb);
        }
        auto vk = [&](int k) __attribute__((always_inline)) { return *reinterpret_cast<const v2f *>(s_vk + 2 * k); };
        const v2f eps2 = v2f{p.mag_eps, 0.f};
        // (Z'[k], Z'[C-k]) -> adjoint inputs (Zs[k], Zs[C-k]) for the gradient magnitudes gk, gc of the two bins
        auto pair = [&](v2f &a, v2f &b, v2f v, float gk, float gc) __attribute__((always_inline)) {
            v2f xk, xc;
            rfft_pair_pk(a, b, v, xk, xc);                           // X[k] = xk, X[C-k] = conj(xc)
            const v2f sk = pk::fma(xk, xk, eps2), sc = pk::fma(xc, xc, eps2);
            if constexpr (MSL) {                                     // gk, gc are the TARGET magnitudes of the two bins
                gk = msl_grad(__builtin_amdgcn_sqrtf(sk.x + sk.y), gk);
                gc = msl_grad(__builtin_amdgcn_sqrtf(sc.x + sc.y), gc);
            }
            const float rk = gk * __builtin_amdgcn_rsqf(sk.x + sk.y);   // 0 * inf = NaN for a zero bin, as autograd of sqrt
            const float rc = gc * __builtin_amdgcn_rsqf(sc.x + sc.y);
            const v2f ha = xk * v2f{rk, rk}, hbc = xc * v2f{rc, rc};  // G[k] and conj(G[C-k])
            const v2f s = ha + hbc, d = ha - hbc;
            const v2f e = pk::cmul_conj(d, v);
            a = s + e;                                               // Zs[k]
            b = (s - e) * v2f{1.f, -1.f};                            // Zs[C-k] = conj(S - conj(v) D)
        };
        // inverse transform: spectrum values instead of gradients.  H[k] = m_k e^{i phi_k} * 2/n, conj(H[C-k]) likewise
        auto pair_inv = [&](v2f &a, v2f &b, v2f v, float mk, float phk, float mc, float phc) __attribute__((always_inline)) {
            float sk, ck, sc, cc;
            fast_sincos(phk, sk, ck);
            fast_sincos(phc, sc, cc);
            const float ak = mk * 2.f * p.inv_n, ac = mc * 2.f * p.inv_n;
            const v2f ha = v2f{ak * ck, ak * sk}, hbc = v2f{ac * cc, -ac * sc};
            const v2f s = ha + hbc, d = ha - hbc;
            const v2f e = pk::cmul_conj(d, v);
            a = s + e;
            b = (s - e) * v2f{1.f, -1.f};
        };
        if (!special) {
            static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
                constexpr int pp = decltype(pc)::value;
                constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
                if constexpr (ISTFT) {
                    pair_inv(za[sa], zb[sb], vk(qA + R1 * pp), gAl[pp], pAl[pp], gBh[pp], pBh[pp]);
                    pair_inv(zb[sa], za[sb], vk(qB + R1 * pp), gBl[pp], pBl[pp], gAh[pp], pAh[pp]);
                } else {
                    pair(za[sa], zb[sb], vk(qA + R1 * pp), gAl[pp], gBh[pp]);      // bins qA + R1 pp | qB + R1 (L-1-pp)
                    pair(zb[sa], za[sb], vk(qB + R1 * pp), gBl[pp], gAh[pp]);      // bins qB + R1 pp | qA + R1 (L-1-pp)
                    // keep the pairs apart: interleaved by the scheduler, the temporaries of all L / 2 evaluations are live at once
                    // (fused loss gradient: 238 VGPRs instead of 172; L = 32: spills)
                    if constexpr (MSL || L == 32) __builtin_amdgcn_sched_barrier(0);
                }
            });
        } else {
            // rows 0 and R1/2 are self-paired (divergent for these FT lanes only).  Row 0 pairs p with (L - p) % L, row R1/2 pairs p
            // with L - 1 - p: the adjoint of a pair lands in the two slots it was read from, so both rows are rewritten IN PLACE
            // (element p lives in slot bitrev(p)).
            // gradient (or spectrum value) of a bin; `ph` and `edge` only matter for the inverse transform
            auto gof = [&](float gm, float xr, float xi, float &gr, float &gi, float ph = 0.f, bool edge = false) __attribute__((always_inline)) {
                if constexpr (ISTFT) {
                    float sn, cs;
                    fast_sincos(ph, sn, cs);
                    const float m = gm * (edge ? p.inv_n : 2.f * p.inv_n);
                    gr = m * cs, gi = m * sn;
                } else {
                    const float m = __builtin_amdgcn_sqrtf(__builtin_fmaf(xr, xr, __builtin_fmaf(xi, xi, p.mag_eps)));
                    if constexpr (MSL) gm = msl_grad(m, gm);
                    const float g = gm / m;
                    gr = g * xr, gi = g * xi;
                }
            };
            float xkr = 0.f, xki = 0.f, xcr = 0.f, xci = 0.f, gkr, gki, gcr, gci;
            static_for<0, L / 2 + 1>([&](auto pc) __attribute__((always_inline)) {
                constexpr int pp = decltype(pc)::value;
                constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev((L - pp) % L, LB);
                const v2f v = vk(R1 * pp);
                if constexpr (!ISTFT) rfft_pair(za[sa].x, za[sa].y, za[sb].x, za[sb].y, v.x, v.y, xkr, xki, xcr, xci);
                gof(pp < L / 2 ? gAl[pp < L / 2 ? pp : 0] : gAh[L / 2 - 1], xkr, xki, gkr, gki,
                    ISTFT ? (pp < L / 2 ? pAl[(ISTFT && pp < L / 2) ? pp : 0] : pAh[ISTFT ? L / 2 - 1 : 0]) : 0.f, pp == 0);
                float z0r, z0i, z1r, z1i;
                if constexpr (pp == 0) {
                    gof(gNy, xcr, xci, gcr, gci, pNy, true);
                    irfft_pair(2.f * gkr, 0.f, 2.f * gcr, 0.f, v.x, v.y, z0r, z0i, z1r, z1i);
                    za[sa] = v2f{z0r, z0i};
                } else if constexpr (2 * pp == L) {
                    irfft_pair(gkr, gki, gkr, gki, v.x, v.y, z0r, z0i, z1r, z1i);
                    za[sa] = v2f{z0r, z0i};
                } else {
                    gof(gAh[(pp >= 1 && pp <= L / 2) ? pp - 1 : 0], xcr, xci, gcr, gci,
                        ISTFT ? pAh[(ISTFT && pp >= 1 && pp <= L / 2) ? pp - 1 : 0] : 0.f);
                    irfft_pair(gkr, gki, gcr, gci, v.x, v.y, z0r, z0i, z1r, z1i);
                    za[sa] = v2f{z0r, z0i};
                    za[sb] = v2f{z1r, z1i};
                }
                __builtin_amdgcn_sched_barrier(0);   // one evaluation at a time (interleaved, the fused loss gradient needed 238 VGPRs)
            });
            static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
                constexpr int pp = decltype(pc)::value;
                constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
                const v2f v = vk(R1 / 2 + R1 * pp);
                if constexpr (!ISTFT) rfft_pair(zb[sa].x, zb[sa].y, zb[sb].x, zb[sb].y, v.x, v.y, xkr, xki, xcr, xci);
                gof(gBl[pp], xkr, xki, gkr, gki, ISTFT ? pBl[ISTFT ? pp : 0] : 0.f);
                gof(gBh[pp], xcr, xci, gcr, gci, ISTFT ? pBh[ISTFT ? pp : 0] : 0.f);
                float z0r, z0i, z1r, z1i;
                irfft_pair(gkr, gki, gcr, gci, v.x, v.y, z0r, z0i, z1r, z1i);
                zb[sa] = v2f{z0r, z0i};
                zb[sb] = v2f{z1r, z1i};
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        pk::fft_dit<L, 1>(za);                                        // inverse over p: U[q][l] in slot l
        pk::fft_dit<L, 1>(zb);
        static_for<0, L / 2>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const v2f ta0 = *reinterpret_cast<const v2f *>(s_tw + (2 * i) * ROW + 2 * qA), ta1 = *reinterpret_cast<const v2f *>(s_tw + (2 * i + 1) * ROW + 2 * qA);
            const v2f tb0 = *reinterpret_cast<const v2f *>(s_tw + (2 * i) * ROW + 2 * qB), tb1 = *reinterpret_cast<const v2f *>(s_tw + (2 * i + 1) * ROW + 2 * qB);
            const v2f a0 = pk::cmul_conj(za[2 * i], ta0), a1 = pk::cmul_conj(za[2 * i + 1], ta1);
            const v2f b0 = pk::cmul_conj(zb[2 * i], tb0), b1 = pk::cmul_conj(zb[2 * i + 1], tb1);
            *reinterpret_cast<f32x4 *>(rowA + 4 * i) = f32x4{a0.x, a0.y, a1.x, a1.y};
            *reinterpret_cast<f32x4 *>(rowB + 4 * i) = f32x4{b0.x, b0.y, b1.x, b1.y};
        });
    }
    __syncthreads();
    PSND_BSTAMP(3);

    // ---- lane threads: inverse radix-32 over q, window, overlap-add of the tile in LDS, one pass over the span ---------
    {
        v2f z[NR][R1];
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            const float *iz = s_x + (fl + FPR * r) * SF + 2 * l;
            static_for<0, R1>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                z[r][ct::bitrev(q, RB)] = *reinterpret_cast<const v2f *>(iz + q * 2 * L);
            });
            pk::fft_dit<R1, 1>(z[r]);                                 // z[l + 16 a] in slot a
            const float *wrow = s_wt + l * ROW;
            static_for<0, R1 / 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
                z[r][2 * i] *= pk::lo(w);
                z[r][2 * i + 1] *= pk::hi(w);
            });
        });
        __syncthreads();                             // exchange consumed: it takes the frames' time samples Y[frame][m]
        PSND_BSTAMP(4);
        // Overlap-add WITHOUT LDS atomics: ds_add_f32 retires about one lane per 12 cycles on gfx950 (measured: the 64
        // atomics of a thread were 51 k of the workgroup's 126 k cycles and stalled the co-resident workgroup's LDS
        // traffic as well).  Every lane parks its 64 windowed samples, then each span sample gathers its <= n/hop frames.
        static_for<0, NR>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            float *yo = s_x + (fl + FPR * r) * SF + 2 * l;
            static_for<0, R1>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                *reinterpret_cast<v2f *>(yo + 2 * L * a) = z[r][a];
            });
        });
        __syncthreads();
        PSND_BSTAMP(5);
        const int t_start = (int)(f0 * hop - p.pad);               // clips are shorter than 2^31 samples (checked on the host)
        const int int_lo = NFFT - hop, int_hi = FT * hop;
        const int nfr = (int)((F - f0) < FT ? (F - f0) : FT);      // valid frames of this tile
        const unsigned hmagic = 0xffffffffu / (unsigned)hop + 1u;  // i / hop = umulhi(i, hmagic) for i < 2^32 / hop
        const int Ti = (int)p.T;
        for (int i = t; i < span_len; i += NT) {
            // frames f with 0 <= i - f*hop < n
            int f_hi = (int)__umulhi((unsigned)i, hmagic);
            if (f_hi > nfr - 1) f_hi = nfr - 1;
            const int above = i - NFFT + hop;                      // f_lo = ceil((i - n + 1) / hop) = floor(above / hop) when > 0
            const int f_lo = above > 0 ? (int)__umulhi((unsigned)above, hmagic) : 0;
            float v = 0.f;
            for (int f = f_lo; f <= f_hi; ++f) v += s_x[f * SF + (i - f * hop)];
            const int tg = t_start + i;
            if constexpr (ISTFT) {
                if (tg < 0 || tg >= Ti) continue;                  // the trimmed n/2 margins
                const int tp = tg + p.pad;                         // squared-window envelope over ALL frames of the clip
                int e_hi = tp < (1 << 24) ? (int)__umulhi((unsigned)tp, hmagic) : tp / hop;
                if (e_hi > iF - 1) e_hi = iF - 1;
                const int ab = tp - NFFT + hop;
                const int e_lo = ab > 0 ? (ab < (1 << 24) ? (int)__umulhi((unsigned)ab, hmagic) : ab / hop) : 0;
                float env = 0.f;
                for (int f = e_lo; f <= e_hi; ++f) {
                    const int m = tp - f * hop, h2 = m >> 1;
                    const float w = 2.f * s_wt[(h2 & (L - 1)) * ROW + 2 * (h2 >> LB) + (m & 1)];
                    env = __builtin_fmaf(w, w, env);
                }
                v /= env + p.env_eps;
                if (i >= int_lo && i < int_hi) gw[tg] = v;
                else if (v != 0.f) unsafeAtomicAdd(gw + tg, v);
            } else if (i >= int_lo && i < int_hi && tg > p.pad && tg < Ti - 1 - p.pad) {
                if (MSL && p.msl_accumulate) gw[tg] += v;          // this tile is the only writer of the sample in this launch
                else gw[tg] = v;
            } else if (v != 0.f) {
                int tr = tg < 0 ? -tg : tg;                        // reflect
                tr = tr >= Ti ? 2 * (Ti - 1) - tr : tr;
                if (tr >= 0 && tr < Ti) unsafeAtomicAdd(gw + tr, v);
            }
        }
    }
#ifdef PSND_TRACE
    PSND_BSTAMP(6);
    __builtin_amdgcn_s_waitcnt(0);
    PSND_BSTAMP(7);
#endif
}

// ---------------------------------------------------------------------------------------------
// n_fft = 4096 (config 5), gradient of the magnitude: the adjoint on the structure of stft_fwd_n4096b_kernel (psnd_stft.hip) -
// 4-frame tiles, 512 threads, X[f][q_a][258] in 66 KB of LDS, two workgroups per CU.  Round 1 ran this size on the generic
// one-frame-per-workgroup LDS FFT (~4 % of the HBM roof).
//   forward recompute: span -> pass A (radix 8) -> pass B (radix 16, in place) -> pass C (radix 16, one row per thread) ->
//                      partner exchange -> X[k], X[C-k] for the thread's 8 bin pairs
//   adjoint split:     G = gmag X / |X| -> Zs[k] (own row, q < 8) and Zs[C-k] (the PARTNER row's upper half, handed over in place)
//   inverse passes:    C (decimation in time, natural out) -> B (conj twiddle, in place) -> A (conj twiddle, window)
//   overlap-add:       the 4 frames' 4096 windowed samples parked in LDS, every span sample gathers its <= 4 frames; the hop
//                      that is complete inside the tile is stored plainly, the rest is added atomically (as the n = 1024 kernel)
// plan(4096) = [win[4096] | wA[256][16] | twA[256][8](re,im) | twB[16][16](re,im) | vk[1025](re,im), padded]
// ---------------------------------------------------------------------------------------------
constexpr int kB4096VK = 2052, kB4096QP = 258, kB4096FP = 8 * kB4096QP + 8, kB4096FT = 4;
constexpr int kB4096LdsFloats = kB4096FT * kB4096FP * 2 + kB4096VK + 512;

__global__ __launch_bounds__(512, 4) void stft_bwd_n4096_mag_kernel(StftBwdParams p) {
    constexpr int C = 2048, NFFT = 4096, FT = kB4096FT, QP = kB4096QP, FP = kB4096FP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_x = smem;
    float *s_vk = s_x + FT * FP * 2;
    float *s_twb = s_vk + kB4096VK;
    const int t = threadIdx.x;
    const float *plan = p.plan;
    const float *g_wa = plan + NFFT, *g_twa = plan + 2 * NFFT, *g_twb = plan + 3 * NFFT, *g_vk = plan + 3 * NFFT + 512;
    constexpr int SPV = 4;
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
    for (int i = t; i < (kB4096VK + 512) / 4; i += 512) {
        const f32x4 v = i < kB4096VK / 4 ? reinterpret_cast<const f32x4 *>(g_vk)[i] : reinterpret_cast<const f32x4 *>(g_twb)[i - kB4096VK / 4];
        reinterpret_cast<f32x4 *>(s_vk)[i] = v;
    }
    const int ja = t & 255, fa0 = 2 * (t >> 8);
    const int fc = t & 3, r = t >> 2, rp = (128 - r) & 127;
    const bool special = (r == 0);
    float *row_own = s_x + 2 * (fc * FP + (r & 7) * QP + (r >> 3) * 16);
    float *row_par = s_x + 2 * (fc * FP + (rp & 7) * QP + (rp >> 3) * 16) + 16;
    const v2f eps2 = v2f{p.mag_eps, 0.f};
    const long long F = p.F;
    const int iF = (int)F, Ti = (int)p.T;

    for (int tile = tw.first; tile < tw.end; tile += tw.step) {
        const int clip = tile / p.ntile;
        const long long f0 = (long long)(tile - clip * p.ntile) * FT;
        const bool more = tile + tw.step < tw.end;
        float *gw = p.gwav + (size_t)clip * p.T;
        f32x4 wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = reinterpret_cast<const f32x4 *>(g_wa + 16 * ja)[i];
        if (tile != tw.first) __syncthreads();           // the previous tile's gather has read every parked sample
        commit_span();
        __syncthreads();
        // ---- forward pass A
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
        __syncthreads();
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
        // the gradient magnitudes of this thread's 16 bins (+ the middle bin for row 0): in flight during pass B
        const bool fvalid = (f0 + fc) < F;
        const float *gbase = p.gmag + (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)(f0 + fc);
        float gk[8], gc[8], gmid = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            gk[q] = fvalid ? gbase[(size_t)(r + 128 * q) * F] : 0.f;
            gc[q] = fvalid ? gbase[(size_t)(128 - r + 128 * (15 - q)) * F] : 0.f;
        }
        if (special && fvalid) gmid = gbase[(size_t)1024 * F];
        __syncthreads();
        // ---- forward pass B (in place)
        {
            const int j2 = t & 15;
            const v2f *twr = reinterpret_cast<const v2f *>(s_twb) + j2;
            const int f = 2 * ((t >> 4) & 1) + (t >> 8), qa = (t >> 5) & 7;
            v2f *base = reinterpret_cast<v2f *>(s_x) + f * FP + qa * QP + j2;
            v2f y[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) y[b] = base[16 * b];
            pk::fft<16>(y);
#pragma unroll
            for (int q = 0; q < 16; ++q) base[16 * q] = q == 0 ? y[0] : pk::cmul(y[ct::bitrev(q, 4)], twr[16 * q]);
        }
        __syncthreads();
        // ---- forward pass C, partner exchange
        v2f zr[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row_own + 4 * i);
            zr[2 * i] = pk::lo(v), zr[2 * i + 1] = pk::hi(v);
        }
        pk::fft<16>(zr);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v2f a = zr[ct::bitrev(8 + 2 * i, 4)], b = zr[ct::bitrev(9 + 2 * i, 4)];
            *reinterpret_cast<f32x4 *>(row_own + 16 + 4 * i) = f32x4{a.x, a.y, b.x, b.y};
        }
        __syncthreads();
        v2f pz[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row_par + 4 * i);
            pz[7 - 2 * i] = pk::lo(v);
            pz[6 - 2 * i] = pk::hi(v);
        }
        if (__builtin_amdgcn_ballot_w64(special) != 0) {
            static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value;
                const v2f own = zr[ct::bitrev(q == 0 ? 0 : 16 - q, 4)];
                pz[q] = special ? own : pz[q];
            });
        }
        // ---- X, G = gmag X / |X|, adjoint split: za <- adjoint of Z[k], zb <- adjoint of Z[C - k]
        auto pair = [&](v2f &a, v2f &b, v2f v, float gkk, float gcc) __attribute__((always_inline)) {
            v2f xk, xc;
            rfft_pair_pk(a, b, v, xk, xc);                           // X[k] = xk, X[C-k] = conj(xc)
            const v2f sk = pk::fma(xk, xk, eps2), sc = pk::fma(xc, xc, eps2);
            const float rk = fvalid ? gkk * __builtin_amdgcn_rsqf(sk.x + sk.y) : 0.f;   // 0 * inf = NaN for a zero bin, as autograd of sqrt
            const float rc = fvalid ? gcc * __builtin_amdgcn_rsqf(sc.x + sc.y) : 0.f;
            const v2f ha = xk * v2f{rk, rk}, hbc = xc * v2f{rc, rc};
            const v2f s = ha + hbc, d = ha - hbc;
            const v2f e = pk::cmul_conj(d, v);
            a = s + e;
            b = (s - e) * v2f{1.f, -1.f};
        };
        v2f midadj = v2f{0.f, 0.f};
        {
            v2f ma = zr[ct::bitrev(8, 4)], mb = ma;                   // middle bin C/2 of row 0: Z[8] paired with itself, only X[k] exists
            v2f xk, xc;
            const v2f v = *reinterpret_cast<const v2f *>(s_vk + 2 * 1024);
            rfft_pair_pk(ma, mb, v, xk, xc);
            const v2f sk = pk::fma(xk, xk, eps2);
            const float rk = (fvalid && special) ? gmid * __builtin_amdgcn_rsqf(sk.x + sk.y) : 0.f;
            const v2f ha = xk * v2f{rk, rk};
            const v2f e = pk::cmul_conj(ha, v);
            midadj = (ha + e) + (ha - e) * v2f{1.f, -1.f};           // the same value fed both inputs: the adjoints add
        }
        static_for<0, 8>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            pair(zr[ct::bitrev(q, 4)], pz[q], *reinterpret_cast<const v2f *>(s_vk + 2 * (r + 128 * q)), gk[q], gc[q]);
        });
        // hand Zs[C - k] over: the partner row's upper half, natural order (index 8 + j <- pz[7 - j]); row 0 keeps its own:
        // index 16 - q <- pz[q] (q >= 1), index 8 <- the middle bin's adjoint, and Z[0] fed both inputs of its pair
        {
            v2f up[8];
            static_for<0, 8>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const v2f gen = pz[7 - j];
                const v2f spc = j == 0 ? midadj : pz[8 - (j == 0 ? 1 : j)];
                up[j] = special ? spc : gen;
            });
            if (special) zr[0] = zr[0] + pz[0];
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(row_par + 4 * i) = f32x4{up[2 * i].x, up[2 * i].y, up[2 * i + 1].x, up[2 * i + 1].y};
        }
        __syncthreads();
        // ---- inverse pass C: adjoint row, q in slot bitrev(q) -> natural order, back into the row
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(row_own + 16 + 4 * i);
                zr[ct::bitrev(8 + 2 * i, 4)] = pk::lo(v);
                zr[ct::bitrev(9 + 2 * i, 4)] = pk::hi(v);
            }
            pk::fft_dit<16, 1>(zr);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4 *>(row_own + 4 * i) = f32x4{zr[2 * i].x, zr[2 * i].y, zr[2 * i + 1].x, zr[2 * i + 1].y};
        }
        __syncthreads();
        // ---- inverse pass B (in place): conj twiddle, inverse radix-16 over q
        {
            const int j2 = t & 15;
            const v2f *twr = reinterpret_cast<const v2f *>(s_twb) + j2;
            const int f = 2 * ((t >> 4) & 1) + (t >> 8), qa = (t >> 5) & 7;
            v2f *base = reinterpret_cast<v2f *>(s_x) + f * FP + qa * QP + j2;
            v2f y[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) y[ct::bitrev(q, 4)] = q == 0 ? base[0] : pk::cmul_conj(base[16 * q], twr[16 * q]);
            pk::fft_dit<16, 1>(y);
#pragma unroll
            for (int b = 0; b < 16; ++b) base[16 * b] = y[b];
        }
        __syncthreads();
        // ---- inverse pass A: conj twiddle, inverse radix-8 over q, window -> the frames' time samples
        // (window and twiddles of the column re-read from the plan: 32 VGPRs that must not live through the whole tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wv[i] = reinterpret_cast<const f32x4 *>(g_wa + 16 * ja)[i];
            tv[i] = reinterpret_cast<const f32x4 *>(g_twa + 16 * ja)[i];
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const v2f *o = reinterpret_cast<const v2f *>(s_x) + (fa0 + f) * FP + ja;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const v2f w = (q & 1) ? pk::hi(tv[q >> 1]) : pk::lo(tv[q >> 1]);
                z[f][ct::bitrev(q, 3)] = q == 0 ? o[0] : pk::cmul_conj(o[q * QP], w);
            }
            pk::fft_dit<8, 1>(z[f]);
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const v2f w = (a & 1) ? pk::hi(wv[a >> 1]) : pk::lo(wv[a >> 1]);
                z[f][a] *= w;
            }
        }
        __syncthreads();                                 // X consumed: the area takes Y[frame][4096]
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int a = 0; a < 8; ++a) *reinterpret_cast<v2f *>(s_x + (fa0 + f) * NFFT + 2 * (ja + 256 * a)) = z[f][a];
        __syncthreads();
        if (more) request_span(tile + tw.step);          // in flight during the gather, committed at the next tile's top
        // ---- overlap-add by gathering: span sample i sums the frames f with 0 <= i - f hop < n
        {
            const int t_start = (int)(f0 * hop - p.pad);
            const int int_lo = NFFT - hop, int_hi = FT * hop;
            const int nfr = (int)((F - f0) < FT ? (F - f0) : FT);
            for (int i = t; i < span_len; i += 512) {
                float v = 0.f;
#pragma unroll
                for (int f = 0; f < FT; ++f) {
                    const int m = i - f * hop;
                    if (f < nfr && m >= 0 && m < NFFT) v += s_x[f * NFFT + m];
                }
                const int tg = t_start + i;
                if (i >= int_lo && i < int_hi && tg > p.pad && tg < Ti - 1 - p.pad) {
                    gw[tg] = v;
                } else if (v != 0.f) {
                    int tr = tg < 0 ? -tg : tg;
                    tr = tr >= Ti ? 2 * (Ti - 1) - tr : tr;
                    if (tr >= 0 && tr < Ti) unsafeAtomicAdd(gw + tr, v);
                }
            }
        }
    }   // tile loop
}

// ---------------------------------------------------------------------------------------------
// generic fallback (power-of-two n_fft without a tuned decomposition): one workgroup per
// (clip, frame), radix-2 FFTs in LDS, global atomics.  plan = win[n].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_fft_radix2(float *sr, float *si, int C, int t, float sign) {
    for (int h = 1; h < C; h <<= 1) {   // DIT on bit-reversed input, natural-order output
        for (int j = t; j < C / 2; j += 256) {
            const int blk = (j / h) * 2 * h, jj = j % h;
            const int i0 = blk + jj, i1 = i0 + h;
            float sn, cs;
            sincospif(sign * (float)jj / (float)h, &sn, &cs);
            const float br = sr[i1] * cs - si[i1] * sn, bi = sr[i1] * sn + si[i1] * cs;
            const float ar = sr[i0], ai = si[i0];
            sr[i0] = ar + br, si[i0] = ai + bi;
            sr[i1] = ar - br, si[i1] = ai - bi;
        }
        __syncthreads();
    }
}

template <bool FROM_MAG, bool FROM_REIM, bool ISTFT = false>
__global__ __launch_bounds__(256) void stft_bwd_generic_kernel(StftBwdParams p, int n_fft) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = n_fft / 2;
    float *sr = smem, *si = smem + C, *hr = smem + 2 * C, *hi = hr + (C + 1);
    const int t = threadIdx.x;
    const long long f = blockIdx.x;
    const int clip = blockIdx.y;
    const float *x = p.wav + (size_t)clip * p.T;
    float *gw = p.gwav + (size_t)clip * p.T;
    const float *win = p.plan;
    const long long s0 = f * p.hop - p.pad;
    int bits = 0;
    while ((1 << bits) < C) ++bits;
    const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)p.F + (size_t)f;
    if (FROM_MAG) {
        for (int m = t; m < C; m += 256) {
            const int r = __brev((unsigned)m) >> (32 - bits);
            sr[r] = x[reflect_idx(s0 + 2 * m, p.T)] * win[2 * m];
            si[r] = x[reflect_idx(s0 + 2 * m + 1, p.T)] * win[2 * m + 1];
        }
        __syncthreads();
        lds_fft_radix2(sr, si, C, t, -1.f);
    }
    // H[k], k = 0..C
    for (int k = t; k <= C; k += 256) {
        float gr = 0.f, gi = 0.f;
        const size_t off = cbase + (size_t)k * p.F;
        if (FROM_MAG) {
            const int kk = k % C, kc = (C - k) % C;
            float sn, cs;
            sincospif(-(float)k / (float)C, &sn, &cs);
            float xkr, xki, xcr, xci;
            rfft_pair(0.5f * sr[kk], 0.5f * si[kk], 0.5f * sr[kc], 0.5f * si[kc], sn, -cs, xkr, xki, xcr, xci);
            const float m = __builtin_amdgcn_sqrtf(xkr * xkr + xki * xki + p.mag_eps);
            const float g = p.gmag[off] / m;
            gr = g * xkr, gi = g * xki;
        }
        if (FROM_REIM) gr += p.gre[off], gi += p.gim[off];
        if (ISTFT) {
            float sn, cs;
            sincosf(p.gre[off], &sn, &cs);
            const float m = p.gmag[off] * ((k == 0 || k == C) ? p.inv_n : 2.f * p.inv_n);
            gr = m * cs, gi = m * sn;
        }
        if (k == 0 || k == C) gr *= 2.f, gi = 0.f;
        hr[k] = gr, hi[k] = gi;
    }
    __syncthreads();
    for (int k = t; k < C; k += 256) {
        float sn, cs;
        sincospif(-(float)k / (float)C, &sn, &cs);
        float zkr, zki, zcr, zci;
        irfft_pair(hr[k], hi[k], hr[C - k], hi[C - k], sn, -cs, zkr, zki, zcr, zci);
        const int r = __brev((unsigned)k) >> (32 - bits);
        sr[r] = zkr, si[r] = zki;
    }
    __syncthreads();
    lds_fft_radix2(sr, si, C, t, 1.f);
    for (int m = t; m < C; m += 256) {
        if (ISTFT) {
            const long long t0 = s0 + 2 * m, t1 = t0 + 1;
            if (t0 >= 0 && t0 < p.T)
                unsafeAtomicAdd(gw + t0, 0.5f * sr[m] * win[2 * m] / (ola_envelope(win, t0 + p.pad, n_fft, p.hop, p.F) + p.env_eps));
            if (t1 >= 0 && t1 < p.T)
                unsafeAtomicAdd(gw + t1, 0.5f * si[m] * win[2 * m + 1] / (ola_envelope(win, t1 + p.pad, n_fft, p.hop, p.F) + p.env_eps));
            continue;
        }
        const long long t0 = reflect_idx(s0 + 2 * m, p.T), t1 = reflect_idx(s0 + 2 * m + 1, p.T);
        unsafeAtomicAdd(gw + t0, 0.5f * sr[m] * win[2 * m]);
        unsafeAtomicAdd(gw + t1, 0.5f * si[m] * win[2 * m + 1]);
    }
}

template <int R1, int L>
int launch_bwd(const StftBwdParams &p, bool from_mag, bool from_reim, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * Cfg<R1, L>::LDS_FLOATS;
    int grid = p.total_tiles;
    if (grid > 2048) grid = 2048;
    grid = (grid + 7) & ~7;
#define PSND_LAUNCH(M_, R_)                                                                          \
    do {                                                                                             \
        auto kern = stft_bwd_kernel<R1, L, M_, R_>;                                                  \
        if (lds > 64 * 1024) {                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_bwd: set LDS size: %s", hipGetErrorString(e)); \
        }                                                                                            \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);                             \
    } while (0)
    if (from_mag && !from_reim) PSND_LAUNCH(true, false);
    else if (!from_mag && from_reim) PSND_LAUNCH(false, true);
    else PSND_LAUNCH(true, true);
#undef PSND_LAUNCH
    PSND_CHECK_LAUNCH("stft_bwd");
    return PSND_OK;
}

template <int R1, int L>
int launch_istft(const StftBwdParams &p, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * Cfg<R1, L>::LDS_FLOATS;
    int grid = p.total_tiles;
    if (grid > 2048) grid = 2048;
    grid = (grid + 7) & ~7;
    auto kern = stft_bwd_kernel<R1, L, false, false, true>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "istft: set LDS size: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    PSND_CHECK_LAUNCH("istft");
    return PSND_OK;
}

}  // namespace

using namespace psnd_stft;

// span-staged adjoint (stft_bwd_n1024_mag_kernel): which (n_fft, hop) it takes, and its launch
static bool span_bwd_ok(int n_fft, int hop) {
    if (PSND_ENV("PSND_STFT_BWD_V1")) return false;
    switch (n_fft) {
        case 512: return hop % 2 == 0 && hop <= 256;
        case 1024: return hop % 4 == 0 && hop <= 256;
        case 2048: return hop % 2 == 0 && hop <= 1024;   // span of 16 frames must fit the exchange: always for hop <= n / 2
    }
    return false;
}

template <bool ISTFT, int R1, int L, bool MSL>
static int launch_span_bwd_one(const StftBwdParams &p, hipStream_t s) {
    using G = BwdGeom<R1, L>;
    constexpr size_t lds = sizeof(float) * G::LDS_FLOATS;
    auto kern = stft_bwd_n1024_mag_kernel<ISTFT, R1, L, MSL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_bwd(span): set LDS size: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3((p.total_tiles + 7) & ~7), dim3(G::NT), lds, s, p);
    PSND_CHECK_LAUNCH(ISTFT ? "istft(span)" : (MSL ? "stft_bwd_msl(span)" : "stft_bwd(span, mag)"));
    return PSND_OK;
}

template <bool ISTFT, bool MSL>
static int launch_span_bwd(int n_fft, const StftBwdParams &p, hipStream_t s) {
    switch (n_fft) {
        case 512: return launch_span_bwd_one<ISTFT, 16, 16, MSL>(p, s);
        case 1024: return launch_span_bwd_one<ISTFT, 32, 16, MSL>(p, s);
        case 2048:
            if constexpr (!ISTFT) return launch_span_bwd_one<false, 32, 32, MSL>(p, s);
    }
    PSND_FAIL(PSND_E_UNSUPPORTED, "stft_bwd(span): n_fft=%d", n_fft);
}

struct MslArgs {
    const float *norms, *g3;
    int L;
    float eps;
    int accumulate;
};

static int stft_bwd_impl(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing, const void *plan, float mag_eps,
                         const float *gmag, const float *gre, const float *gim, float *gwav, void *stream, const MslArgs *msl) {
    if (!plan || !gwav) PSND_FAIL(PSND_E_ARG, "stft_bwd: null plan/gwav");
    if ((gre == nullptr) != (gim == nullptr)) PSND_FAIL(PSND_E_ARG, "stft_bwd: gre and gim must be given together");
    if (!gmag && !gre) PSND_FAIL(PSND_E_ARG, "stft_bwd: no gradient source");
    if (gmag && !wav) PSND_FAIL(PSND_E_ARG, "stft_bwd: gmag needs wav (recompute)");
    if (framing < PSND_FRAMING_CENTER || framing > PSND_FRAMING_NONE) PSND_FAIL(PSND_E_ARG, "stft_bwd: framing=%d", framing);
    if (hop <= 0 || N < 0) PSND_FAIL(PSND_E_ARG, "stft_bwd: hop=%d N=%lld", hop, (long long)N);
    if (psnd_stft_plan_bytes(n_fft) == 0) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_bwd: n_fft=%d unsupported", n_fft);
    const int pad = framing == PSND_FRAMING_NONE ? 0 : (framing == PSND_FRAMING_CENTER ? n_fft / 2 : (n_fft - hop) / 2);
    if (pad < 0 || T <= pad) PSND_FAIL(PSND_E_SHAPE, "stft_bwd: reflect padding %d needs T > pad (T=%lld)", pad, (long long)T);
    if (T >= ((int64_t)1 << 31) - 4 * (int64_t)n_fft) PSND_FAIL(PSND_E_SHAPE, "stft_bwd: T too large");
    if (N == 0) return PSND_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!(msl && msl->accumulate)) {
        hipError_t me = hipMemsetAsync(gwav, 0, sizeof(float) * (size_t)N * (size_t)T, s);
        if (me != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_bwd: memset: %s", hipGetErrorString(me));
    }
    const int64_t F = psnd_frame_count(T, n_fft, hop, framing);
    if (F <= 0) return PSND_OK;
    const int64_t K = n_fft / 2 + 1;
    if (K * F >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_bwd: K*F too large");
    StftBwdParams p;
    p.wav = wav, p.plan = static_cast<const float *>(plan), p.gmag = gmag, p.gre = gre, p.gim = gim, p.gwav = gwav;
    p.T = T, p.F = F, p.hop = hop, p.pad = pad, p.mag_eps = mag_eps;
    p.win_off = 0, p.inv_n = 0.f, p.env_eps = 0.f;
    p.msl_norms = p.msl_g = nullptr, p.msl_invLN = p.msl_invLNKF = p.msl_eps = 0.f, p.msl_accumulate = 0;
    if (msl) {
        const double ln = (double)msl->L * (double)N;
        p.msl_norms = msl->norms, p.msl_g = msl->g3, p.msl_eps = msl->eps, p.msl_accumulate = msl->accumulate;
        p.msl_invLN = (float)(1.0 / ln), p.msl_invLNKF = (float)(1.0 / (ln * (double)(K * F)));
    }
#ifdef PSND_TRACE
    {
        const char *tp = PSND_ENV("PSND_TRACE_PTR");
        p.trace = tp ? reinterpret_cast<long long *>(strtoull(tp, nullptr, 0)) : nullptr;
    }
#endif
    const Decomp *d = find_decomp(n_fft);
    if (d) {
        const int FT = 512 / d->R1;
        const int64_t ntile = (F + FT - 1) / FT;
        if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_bwd: too many tiles");
        p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
        switch (n_fft) {
            case 256:
                if (msl) break;
                return launch_bwd<16, 8>(p, gmag, gre, s);
            case 512:
            case 1024:
            case 2048:
                if (gmag && !gre && span_bwd_ok(n_fft, hop))
                    return msl ? launch_span_bwd<false, true>(n_fft, p, s) : launch_span_bwd<false, false>(n_fft, p, s);
                if (msl) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_bwd_msl: n_fft=%d hop=%d", n_fft, hop);
                if (n_fft == 512) return launch_bwd<16, 16>(p, gmag, gre, s);
                if (n_fft == 1024) return launch_bwd<32, 16>(p, gmag, gre, s);
                return launch_bwd<32, 32>(p, gmag, gre, s);
        }
    }
    if (msl) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_bwd_msl: n_fft=%d hop=%d", n_fft, hop);
    if (n_fft == 4096 && gmag && !gre && hop % 2 == 0 && hop <= 1364 && 4096 % hop == 0 && !PSND_ENV("PSND_STFT_GENERIC")) {
        // magnitude gradient at the config-5 size: the adjoint of stft_fwd_n4096b_kernel (4-frame tiles, two workgroups per CU)
        const int64_t ntile = (F + kB4096FT - 1) / kB4096FT;
        if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_bwd: too many tiles");
        p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
        constexpr size_t lds = sizeof(float) * kB4096LdsFloats;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_bwd_n4096_mag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stft_bwd(n4096): set LDS size: %s", hipGetErrorString(e));
        int grid = p.total_tiles < 512 ? p.total_tiles : 512;
        grid = (grid + 7) & ~7;
        hipLaunchKernelGGL(stft_bwd_n4096_mag_kernel, dim3(grid), dim3(512), lds, s, p);
        PSND_CHECK_LAUNCH("stft_bwd(n4096, mag)");
        return PSND_OK;
    }
    if (F > 0x7fffffff || N > 65535) PSND_FAIL(PSND_E_SHAPE, "stft_bwd(generic): grid too large");
    p.ntile = 0, p.total_tiles = 0;
    const size_t lds = sizeof(float) * (size_t)(2 * n_fft + 2);
    dim3 grid((unsigned)F, (unsigned)N);
    if (gmag && !gre) hipLaunchKernelGGL((stft_bwd_generic_kernel<true, false>), grid, dim3(256), lds, s, p, n_fft);
    else if (!gmag && gre) hipLaunchKernelGGL((stft_bwd_generic_kernel<false, true>), grid, dim3(256), lds, s, p, n_fft);
    else hipLaunchKernelGGL((stft_bwd_generic_kernel<true, true>), grid, dim3(256), lds, s, p, n_fft);
    PSND_CHECK_LAUNCH("stft_bwd(generic)");
    return PSND_OK;
}

extern "C" int psnd_stft_bwd(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                             const void *plan, float mag_eps, const float *gmag, const float *gre,
                             const float *gim, float *gwav, void *stream) {
    return stft_bwd_impl(wav, N, T, n_fft, hop, framing, plan, mag_eps, gmag, gre, gim, gwav, stream, nullptr);
}

// multi_stft_loss (models/sound.py:106-133), gradient w.r.t. the PREDICTED waveform of one resolution in one launch: the adjoint STFT
// recomputes |X| of `wav`, forms d loss / d |X| from it and the target magnitudes (what psnd_stft_loss_bwd writes to HBM as `gp`) and
// carries on as psnd_stft_bwd.  norms / g3 / L / eps as psnd_stft_loss_bwd.
extern "C" int psnd_stft_bwd_msl_supported(int n_fft, int hop) { return hop > 0 && span_bwd_ok(n_fft, hop) ? 1 : 0; }

extern "C" int psnd_stft_bwd_msl(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing, const void *plan, float mag_eps,
                                 const float *t_mag, const float *norms, const float *g3, int L, float eps, int accumulate, float *gwav,
                                 void *stream) {
    if (!wav || !t_mag || !norms || !g3) PSND_FAIL(PSND_E_ARG, "stft_bwd_msl: null pointer");
    if (L <= 0) PSND_FAIL(PSND_E_SHAPE, "stft_bwd_msl: L=%d", L);
    if (!psnd_stft_bwd_msl_supported(n_fft, hop)) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_bwd_msl: n_fft=%d hop=%d", n_fft, hop);
    const MslArgs m{norms, g3, L, eps, accumulate ? 1 : 0};
    return stft_bwd_impl(wav, N, T, n_fft, hop, framing, plan, mag_eps, t_mag, nullptr, nullptr, gwav, stream, &m);
}

// STFT.inverse (pytorch_sound/models/transforms.py:71-101): conv_transpose1d with pinv(n/h * basis)^T * window
// == (h/n) * window * irDFT per frame, overlap-added, divided by the squared-window envelope (+eps), scaled
// by n/h and trimmed by n/2 on both sides: out[t] = OLA(w * irfft(X_f))[t + n/2] / (env[t + n/2] + eps).
extern "C" int psnd_istft(const float *mag, const float *phase, int64_t N, int64_t F, int n_fft, int hop,
                          const void *plan, float eps, float *out, void *stream) {
    if (!mag || !phase || !plan || !out) PSND_FAIL(PSND_E_ARG, "istft: null pointer");
    if (hop <= 0 || N < 0 || F < 0) PSND_FAIL(PSND_E_ARG, "istft: hop=%d N=%lld F=%lld", hop, (long long)N, (long long)F);
    if (psnd_stft_plan_bytes(n_fft) == 0) PSND_FAIL(PSND_E_UNSUPPORTED, "istft: n_fft=%d unsupported", n_fft);
    const int64_t T = (F - 1) * hop;
    if (N == 0 || F <= 1) return PSND_OK;
    const int64_t K = n_fft / 2 + 1;
    if (K * F >= (int64_t)1 << 31 || T >= ((int64_t)1 << 31) - 4 * (int64_t)n_fft) PSND_FAIL(PSND_E_SHAPE, "istft: clip too long");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t me = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N * (size_t)T, s);
    if (me != hipSuccess) PSND_FAIL(PSND_E_HIP, "istft: memset: %s", hipGetErrorString(me));
    StftBwdParams p;
    p.wav = nullptr, p.plan = static_cast<const float *>(plan), p.gmag = mag, p.gre = phase, p.gim = nullptr, p.gwav = out;
    p.T = T, p.F = F, p.hop = hop, p.pad = n_fft / 2, p.mag_eps = 0.f;
    p.inv_n = 1.0f / (float)n_fft, p.env_eps = eps;
    const Decomp *d = find_decomp(n_fft);
    if (d) {
        const PlanLayout lay = plan_layout(n_fft, d->R1, d->L);
        p.win_off = lay.win;
        const int FT = 512 / d->R1;
        const int64_t ntile = (F + FT - 1) / FT;
        if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "istft: too many tiles");
        p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
        switch (n_fft) {
            case 256: return launch_istft<16, 8>(p, s);
            case 512:
                if (span_bwd_ok(512, hop)) return launch_span_bwd<true, false>(512, p, s);
                return launch_istft<16, 16>(p, s);
            case 1024:
                if (span_bwd_ok(1024, hop)) return launch_span_bwd<true, false>(1024, p, s);
                return launch_istft<32, 16>(p, s);
            case 2048: return launch_istft<32, 32>(p, s);
        }
    }
    if (F > 0x7fffffff || N > 65535) PSND_FAIL(PSND_E_SHAPE, "istft(generic): grid too large");
    p.win_off = 0, p.ntile = 0, p.total_tiles = 0;
    const size_t lds = sizeof(float) * (size_t)(2 * n_fft + 2);
    hipLaunchKernelGGL((stft_bwd_generic_kernel<false, false, true>), dim3((unsigned)F, (unsigned)N), dim3(256), lds, s, p, n_fft);
    PSND_CHECK_LAUNCH("istft(generic)");
    return PSND_OK;
}
