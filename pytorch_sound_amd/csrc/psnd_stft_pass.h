// psnd_stft_pass.h - device building blocks of the two-pass register FFT (see psnd_stft.hip).
#pragma once
#include "psnd_stft_plan.h"

namespace psnd_stft {

template <int R1, int L>
struct Cfg {
    static constexpr int C = R1 * L, NFFT = 2 * C;
    static constexpr int FT = 512 / R1;          // frames per tile: FT * R1/2 pass-2 tasks == 256
    static constexpr int ROW = 2 * R1 + 4;
    static constexpr int SF = C + 4;             // exchange frame stride; SF/4 odd -> b128 conflict-free
    static constexpr int P1R = (FT * L) / 256;   // pass-1 rounds
    static constexpr int VKP = ((2 * (C / 2 + 1)) + 3) & ~3;
    static constexpr int LB = ct::ilog2(L), RB = ct::ilog2(R1);
    static constexpr int TAB = 2 * L * ROW + VKP;            // floats of tables staged in LDS
    static constexpr int LDS_FLOATS = TAB + 2 * FT * SF;
    static_assert(FT * L % 256 == 0 && P1R >= 1, "pass-1 tiling");
    static_assert((SF / 4) % 2 == 1 && (ROW / 4) % 2 == 1, "LDS strides");
};

struct Smem {
    float *wt, *tw, *vk, *xr, *xi;
};

template <int R1, int L>
__device__ __forceinline__ Smem carve(float *smem) {
    using G = Cfg<R1, L>;
    Smem s;
    s.wt = smem;
    s.tw = s.wt + L * G::ROW;
    s.vk = s.tw + L * G::ROW;
    s.xr = s.vk + G::VKP;
    s.xi = s.xr + G::FT * G::SF;
    return s;
}

template <int R1, int L>
__device__ __forceinline__ void load_tables(const float *plan, float *smem, int t) {
    using G = Cfg<R1, L>;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(plan);
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    for (int i = t; i < G::TAB / 4; i += 256) dst[i] = src[i];
}

// XCD-aware tile walk: block b runs on XCD b % 8 (observed, speed only).  Every XCD gets one
// contiguous range of tiles so neighbouring frame tiles of a clip - which share the 128-B output
// lines at their common edge - go through the same L2.
struct TileWalk {
    int first, end, step;
};
// contiguous run of tiles per workgroup: consecutive frame tiles of a clip share the 128-B output lines
// at their common edge; written a few microseconds apart by the SAME CU they are merged in L2 before
// eviction (non-temporal / write-through stores, which defeat that merging, measured 35-60 % slower).
__device__ __forceinline__ TileWalk tile_run(int total_tiles) {
    TileWalk w;
    w.first = (int)((long long)blockIdx.x * total_tiles / gridDim.x);
    w.end = (int)((long long)(blockIdx.x + 1) * total_tiles / gridDim.x);
    w.step = 1;
    return w;
}
__device__ __forceinline__ TileWalk tile_walk(int total_tiles) {
    const int nb = gridDim.x, xcd = blockIdx.x & 7, bx = blockIdx.x >> 3;
    const int nbx = (nb - xcd + 7) >> 3;
    const int chunk = (total_tiles + 7) >> 3;
    const int lo = xcd * chunk;
    TileWalk w;
    w.first = lo + bx;
    w.end = min(lo + chunk, total_tiles);
    w.step = nbx;
    return w;
}

// pass-2 identity of thread t: frame f2 of the tile and bin pair qq (butterflies qq and R1-qq)
template <int R1, int L>
__device__ __forceinline__ void pass2_identity(int t, int &f2, int &qq) {
    constexpr int FT = Cfg<R1, L>::FT;
    if constexpr (FT == 16) {
        f2 = t & 15;
        qq = (t >> 6) + 4 * ((t >> 4) & 3);   // the 4 pairs of a wave differ by 4: same LDS slot phase
    } else {
        f2 = t & (FT - 1);
        qq = t / FT;
    }
}

// ---- forward pass 1, split in two: load_frame (global -> registers) and pass1_compute.
//
// load_frame: raw (un-windowed) samples of lane l of frame f0+fl into registers.  Interior frames:
// R1 8-byte loads at compile-time offsets.  Clip-edge frames (rare): per-sample reflect indexing.
template <int R1, int L>
__device__ __forceinline__ void load_frame(const float *x, long long T, long long F, long long f0, int hop, int pad,
                                           int fl, int l, float (&zr)[R1], float (&zi)[R1]) {
    constexpr int NFFT = Cfg<R1, L>::NFFT;
    const long long f = f0 + fl;
    if (f < F) {
        const long long s0 = f * hop - pad;
        if (s0 >= 0 && s0 + NFFT <= T) {
            const float *px = x + s0 + 2 * l;
            static_for<0, R1>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                const f32x2_u v = *reinterpret_cast<const f32x2_u *>(px + 2 * L * a);
                zr[a] = v.x;
                zi[a] = v.y;
            });
        } else {
            const int Ti = (int)T, b0 = (int)s0 + 2 * l;
            static_for<0, R1>([&](auto ac) __attribute__((always_inline)) {
                constexpr int a = decltype(ac)::value;
                zr[a] = x[reflect_idx32(b0 + 2 * L * a, Ti)];
                zi[a] = x[reflect_idx32(b0 + 2 * L * a + 1, Ti)];
            });
        }
    } else {
        static_for<0, R1>([&](auto ac) __attribute__((always_inline)) {
            zr[decltype(ac)::value] = 0.f;
            zi[decltype(ac)::value] = 0.f;
        });
    }
}

// pass1_compute: window (LDS table), radix-R1 FFT in VGPRs, inter-pass twiddle, write Y[q][l].
template <int R1, int L>
__device__ __forceinline__ void pass1_compute(const Smem &s, int fl, int l, float (&zr)[R1], float (&zi)[R1]) {
    using G = Cfg<R1, L>;
    constexpr int ROW = G::ROW, SF = G::SF, RB = G::RB;
    const float *wrow = s.wt + l * ROW;
    static_for<0, R1 / 2>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const f32x4 w = *reinterpret_cast<const f32x4 *>(wrow + 4 * i);
        zr[2 * i] *= w.x;
        zi[2 * i] *= w.y;
        zr[2 * i + 1] *= w.z;
        zi[2 * i + 1] *= w.w;
    });
    fft_inreg<R1>(zr, zi);
    const float *trow = s.tw + l * ROW;
    float *oxr = s.xr + fl * SF + l;
    float *oxi = s.xi + fl * SF + l;
    static_for<0, R1 / 2>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const f32x4 w = *reinterpret_cast<const f32x4 *>(trow + 4 * i);
        constexpr int s0_ = ct::bitrev(2 * i, RB), s1_ = ct::bitrev(2 * i + 1, RB);
        if constexpr (i == 0) {
            oxr[0] = zr[s0_];
            oxi[0] = zi[s0_];
        } else {
            oxr[(2 * i) * L] = __builtin_fmaf(zr[s0_], w.x, -zi[s0_] * w.y);
            oxi[(2 * i) * L] = __builtin_fmaf(zr[s0_], w.y, zi[s0_] * w.x);
        }
        oxr[(2 * i + 1) * L] = __builtin_fmaf(zr[s1_], w.z, -zi[s1_] * w.w);
        oxi[(2 * i + 1) * L] = __builtin_fmaf(zr[s1_], w.w, zi[s1_] * w.z);
    });
}

template <int R1, int L>
__device__ __forceinline__ void fwd_pass1(const Smem &s, const float *x, long long T, long long F, long long f0,
                                          int hop, int pad, int fl, int l) {
    float zr[R1], zi[R1];
    load_frame<R1, L>(x, T, F, f0, hop, pad, fl, l, zr, zi);
    pass1_compute<R1, L>(s, fl, l, zr, zi);
}

// ---- forward pass 2, first half: read rows qA / qB of frame f2 and run the two radix-L FFTs.
//      Z'[q + R1 p] ends in slot bitrev(p).
template <int R1, int L>
__device__ __forceinline__ void fwd_pass2_fft(const Smem &s, int f2, int qA, int qB, float (&ar)[L], float (&ai)[L],
                                              float (&br)[L], float (&bi)[L]) {
    constexpr int SF = Cfg<R1, L>::SF;
    const float *pa_r = s.xr + f2 * SF + qA * L, *pa_i = s.xi + f2 * SF + qA * L;
    const float *pb_r = s.xr + f2 * SF + qB * L, *pb_i = s.xi + f2 * SF + qB * L;
    static_for<0, L / 4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(pa_r + 4 * i);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(pa_i + 4 * i);
        const f32x4 v2 = *reinterpret_cast<const f32x4 *>(pb_r + 4 * i);
        const f32x4 v3 = *reinterpret_cast<const f32x4 *>(pb_i + 4 * i);
        ar[4 * i] = v0.x, ar[4 * i + 1] = v0.y, ar[4 * i + 2] = v0.z, ar[4 * i + 3] = v0.w;
        ai[4 * i] = v1.x, ai[4 * i + 1] = v1.y, ai[4 * i + 2] = v1.z, ai[4 * i + 3] = v1.w;
        br[4 * i] = v2.x, br[4 * i + 1] = v2.y, br[4 * i + 2] = v2.z, br[4 * i + 3] = v2.w;
        bi[4 * i] = v3.x, bi[4 * i + 1] = v3.y, bi[4 * i + 2] = v3.z, bi[4 * i + 3] = v3.w;
    });
    fft_inreg<L>(ar, ai);
    fft_inreg<L>(br, bi);
}

}  // namespace psnd_stft
