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
#include <math.h>
#include <string.h>
#include <vector>

namespace {
using namespace psnd_stft;

struct StftFwdParams {
    const float *wav;
    const float *plan;
    float *mag, *phase, *re, *im;
    long long T, F;
    int hop, pad, ntile, total_tiles;
    float mag_eps;
};

template <bool MAG, bool PHASE, bool REIM>
struct Emit {
    float *mag, *phase, *re, *im;
    float eps;
    bool valid;
    __device__ __forceinline__ void operator()(int off, float xr, float xi) const {
        if (!valid) return;
        if constexpr (MAG) mag[off] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xr, xr, __builtin_fmaf(xi, xi, eps)));
        if constexpr (PHASE) phase[off] = atan2f(xi, xr);
        if constexpr (REIM) {
            re[off] = xr;
            im[off] = xi;
        }
    }
};

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
            const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)F + (size_t)(f0 + f2);
            Emit<MAG, PHASE, REIM> emit{MAG ? p.mag + cbase : nullptr, PHASE ? p.phase + cbase : nullptr,
                                        REIM ? p.re + cbase : nullptr, REIM ? p.im + cbase : nullptr,
                                        p.mag_eps, (f0 + f2) < F};
            const int iF = (int)F;
            const int stepF = R1 * iF;
            const int offA = qA * iF, offB = qB * iF;
            const float *s_vk = s.vk;
            float xkr, xki, xcr, xci;
            if (!special) {
                static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int pp = decltype(pc)::value;
                    constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
                    {   // primary A[pp]: k = qA + R1*pp (< C/2), partner B[L-1-pp]
                        const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (qA + R1 * pp));
                        rfft_pair(ar[sa], ai[sa], br[sb], bi[sb], v.x, v.y, xkr, xki, xcr, xci);
                        emit(offA + pp * stepF, xkr, xki);
                        emit(offB + (L - 1 - pp) * stepF, xcr, xci);
                    }
                    {   // primary B[pp]: k = qB + R1*pp (< C/2), partner A[L-1-pp]
                        const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (qB + R1 * pp));
                        rfft_pair(br[sa], bi[sa], ar[sb], ai[sb], v.x, v.y, xkr, xki, xcr, xci);
                        emit(offB + pp * stepF, xkr, xki);
                        emit(offA + (L - 1 - pp) * stepF, xcr, xci);
                    }
                });
            } else {
                // butterfly q = 0: bins R1*p pair with C - R1*p = R1*(L-p); p = 0 gives X[0] and X[C]
                static_for<0, L / 2 + 1>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int pp = decltype(pc)::value;
                    constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev((L - pp) % L, LB);
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (R1 * pp));
                    rfft_pair(ar[sa], ai[sa], ar[sb], ai[sb], v.x, v.y, xkr, xki, xcr, xci);
                    emit(pp * stepF, xkr, xki);
                    if constexpr (2 * pp != L) emit((L - pp) * stepF, xcr, xci);
                });
                // butterfly q = R1/2: bins R1/2 + R1*p pair with R1/2 + R1*(L-1-p)
                static_for<0, L / 2>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int pp = decltype(pc)::value;
                    constexpr int sa = ct::bitrev(pp, LB), sb = ct::bitrev(L - 1 - pp, LB);
                    const f32x2 v = *reinterpret_cast<const f32x2 *>(s_vk + 2 * (R1 / 2 + R1 * pp));
                    rfft_pair(br[sa], bi[sa], br[sb], bi[sb], v.x, v.y, xkr, xki, xcr, xci);
                    emit(offB + pp * stepF, xkr, xki);
                    emit(offB + (L - 1 - pp) * stepF, xcr, xci);
                });
            }
        }
        __syncthreads();  // exchange buffer is reused by the next tile
    }
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
    const size_t cbase = (size_t)clip * (size_t)(C + 1) * (size_t)p.F + (size_t)f;
    Emit<MAG, PHASE, REIM> emit{MAG ? p.mag + cbase : nullptr, PHASE ? p.phase + cbase : nullptr,
                                REIM ? p.re + cbase : nullptr, REIM ? p.im + cbase : nullptr, p.mag_eps, true};
    for (int k = t; k <= C / 2; k += 256) {
        const int kc = (C - k) % C;
        float sn, cs;
        sincospif(-(float)k / (float)C, &sn, &cs);   // W_n^k = cs + i sn
        // v_k = -i W_n^k = sn - i cs ; inputs here are unscaled -> halve
        float xkr, xki, xcr, xci;
        rfft_pair(0.5f * sr[k], 0.5f * si[k], 0.5f * sr[kc], 0.5f * si[kc], sn, -cs, xkr, xki, xcr, xci);
        emit((int)((long long)k * p.F), xkr, xki);
        if (k != C - k) emit((int)((long long)(C - k) * p.F), xcr, xci);
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

}  // namespace

using namespace psnd_stft;

extern "C" size_t psnd_stft_plan_bytes(int n_fft) {
    if (const Decomp *d = find_decomp(n_fft)) return sizeof(float) * (size_t)plan_layout(n_fft, d->R1, d->L).total;
    if (generic_ok(n_fft)) return sizeof(float) * (size_t)n_fft;
    return 0;
}

extern "C" int psnd_stft_plan_build(int n_fft, const float *window_host, void *plan_host) {
    if (!window_host || !plan_host) PSND_FAIL(PSND_E_ARG, "stft_plan_build: null pointer");
    if (psnd_stft_plan_bytes(n_fft) == 0) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_plan_build: n_fft=%d unsupported (power of two in [16,8192])", n_fft);
    float *pl = static_cast<float *>(plan_host);
    const Decomp *d = find_decomp(n_fft);
    if (!d) {
        memcpy(pl, window_host, sizeof(float) * n_fft);
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
    return PSND_OK;
}

extern "C" int psnd_stft_fwd(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                             const void *plan, float mag_eps, float *mag, float *phase, float *re, float *im,
                             void *stream) {
    if (!wav || !plan) PSND_FAIL(PSND_E_ARG, "stft_fwd: null wav/plan");
    if ((re == nullptr) != (im == nullptr)) PSND_FAIL(PSND_E_ARG, "stft_fwd: re and im must be given together");
    if (!mag && !phase && !re) PSND_FAIL(PSND_E_ARG, "stft_fwd: no output requested");
    if (framing != PSND_FRAMING_CENTER && framing != PSND_FRAMING_HIFIGAN) PSND_FAIL(PSND_E_ARG, "stft_fwd: framing=%d", framing);
    if (hop <= 0 || N < 0) PSND_FAIL(PSND_E_ARG, "stft_fwd: hop=%d N=%lld", hop, (long long)N);
    if (psnd_stft_plan_bytes(n_fft) == 0) PSND_FAIL(PSND_E_UNSUPPORTED, "stft_fwd: n_fft=%d unsupported", n_fft);
    const int pad = framing == PSND_FRAMING_CENTER ? n_fft / 2 : (n_fft - hop) / 2;
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
    p.T = T, p.F = F, p.hop = hop, p.pad = pad, p.mag_eps = mag_eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Decomp *d = find_decomp(n_fft);
    if (d) {
        const int FT = 512 / d->R1;
        const int64_t ntile = (F + FT - 1) / FT;
        if (ntile * N >= (int64_t)1 << 31) PSND_FAIL(PSND_E_SHAPE, "stft_fwd: too many tiles");
        p.ntile = (int)ntile, p.total_tiles = (int)(ntile * N);
        switch (n_fft) {
            case 256: return launch_tuned<16, 8>(p, mag, phase, re, s);
            case 512: return launch_tuned<16, 16>(p, mag, phase, re, s);
            case 1024: return launch_tuned<32, 16>(p, mag, phase, re, s);
            case 2048: return launch_tuned<32, 32>(p, mag, phase, re, s);
        }
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
