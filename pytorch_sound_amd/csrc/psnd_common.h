// psnd_common.h - shared host/device helpers for libpsnd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>
#include "../../include/psnd.h"

// ---------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------
void psnd_set_error(const char *fmt, ...);

#define PSND_FAIL(code, ...)          \
    do {                              \
        psnd_set_error(__VA_ARGS__);  \
        return (code);                \
    } while (0)

#define PSND_CHECK_LAUNCH(what)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) PSND_FAIL(PSND_E_HIP, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------
// A/B switches (host) - LAB builds only (-DPSND_LAB: `python -m pytorch_sound_amd._build --lab` -> libpsnd_hip_lab.so, what tools/ and the
// parity tests of kernel instances load).  In the product build PSND_ENV(...) is a null constant: no getenv, no switch - every dispatcher
// takes its measured default and the compiler drops the branches.  In a lab build the PSND_* environment switches of the dispatchers
// (kernel-instance choices for parity tests and same-box A/B timings, ablations, trace pointers) are looked up ONCE per call site and kept; a
// process that changes its environment afterwards calls psnd_env_refresh() (tests/conftest.py does, around monkeypatch.setenv).
// ---------------------------------------------------------------------------------------
#include <stdlib.h>
#ifdef PSND_LAB
#include <atomic>
extern std::atomic<int> g_psnd_env_gen;
struct PsndEnvSlot {
    std::atomic<int> gen{-1};
    const char *val = nullptr;
};
inline const char *psnd_env_lookup(PsndEnvSlot &c, const char *name) {
    const int g = g_psnd_env_gen.load(std::memory_order_acquire);
    if (c.gen.load(std::memory_order_acquire) != g) {
        c.val = getenv(name);
        c.gen.store(g, std::memory_order_release);
    }
    return c.val;
}
#define PSND_ENV(name_) ([]() -> const char * { static PsndEnvSlot slot_; return psnd_env_lookup(slot_, name_); }())
#define PSND_ABL(p_, bits_) ((p_).ablate & (bits_))      // ablation bits of a kernel's parameter block (PSND_ABLATE)
#else
#define PSND_ENV(name_) (static_cast<const char *>(nullptr))
#define PSND_ABL(p_, bits_) 0
#endif
// integer switch clamped to [lo, hi]; `dflt` when unset or unparsable
inline int psnd_env_int(const char *v, int dflt, int lo, int hi) {
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    if (end == v) return dflt;
    return x < lo ? lo : (x > hi ? hi : (int)x);
}

// ---------------------------------------------------------------------------------------
// compile-time loops
// ---------------------------------------------------------------------------------------
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---------------------------------------------------------------------------------------
// compile-time trigonometry: cos/sin(2*pi*j/n), exact octant reduction + Taylor on [0, pi/4]
// ---------------------------------------------------------------------------------------
namespace ct {
constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double taylor_sin(double x) {
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 12; ++i) {
        term *= -x2 / double((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr double taylor_cos(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i < 12; ++i) {
        term *= -x2 / double((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
// cos(2*pi*j/n) and sin(2*pi*j/n) for integer j, n (n multiple of 8 or small power of two)
constexpr double cos2pi(long j, long n) {
    j %= n;
    if (j < 0) j += n;
    // use symmetries to land in [0, n/8]
    if (2 * j > n) return cos2pi(n - j, n);              // cos(2pi - x) = cos x
    if (4 * j > n) return -cos2pi(n / 2 - j, n);                   // cos(pi - x) = -cos x
    if (8 * j > n) {                                     // x in (pi/4, pi/2]: cos x = sin(pi/2 - x)
        // pi/2 - x = 2pi*(n/4 - j)/n ; n divisible by 4 for every size we use
        return taylor_sin(2.0 * kPi * double(n - 4 * j) / double(4 * n));
    }
    return taylor_cos(2.0 * kPi * double(j) / double(n));
}
constexpr double sin2pi(long j, long n) {
    j %= n;
    if (j < 0) j += n;
    if (2 * j > n) return -sin2pi(n - j, n);             // sin(2pi - x) = -sin x
    if (4 * j > n) return sin2pi(n / 2 - j, n);          // sin(pi - x) = sin x
    if (8 * j > n) return taylor_cos(2.0 * kPi * double(n - 4 * j) / double(4 * n));
    return taylor_sin(2.0 * kPi * double(j) / double(n));
}
constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }
constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
}  // namespace ct

// ---------------------------------------------------------------------------------------
// in-register complex FFT of compile-time size R (power of two), forward sign (e^{-i..}),
// decimation in frequency, fully unrolled.  Input natural order; output X[q] lands in
// slot bitrev(q).  All twiddles are immediates.
// ---------------------------------------------------------------------------------------
template <int R, int H, int BLK, int J>
__device__ __forceinline__ void fft_bfly(float (&re)[R], float (&im)[R]) {
    constexpr int i0 = BLK + J, i1 = BLK + J + H;
    const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
    re[i0] = ar + br;
    im[i0] = ai + bi;
    const float tr = ar - br, ti = ai - bi;
    // (tr + i ti) * (c - i s),  theta = 2*pi*J/(2H)
    if constexpr (J == 0) {
        re[i1] = tr;
        im[i1] = ti;
    } else if constexpr (2 * J == H) {  // -i
        re[i1] = ti;
        im[i1] = -tr;
    } else if constexpr (4 * J == H) {  // (1 - i)/sqrt2
        constexpr float r = (float)ct::cos2pi(1, 8);
        re[i1] = (tr + ti) * r;
        im[i1] = (ti - tr) * r;
    } else if constexpr (4 * J == 3 * H) {  // (-1 - i)/sqrt2
        constexpr float r = (float)ct::cos2pi(1, 8);
        re[i1] = (ti - tr) * r;
        im[i1] = -(tr + ti) * r;
    } else {
        constexpr float c = (float)ct::cos2pi(J, 2 * H);
        constexpr float s = (float)ct::sin2pi(J, 2 * H);
        re[i1] = __builtin_fmaf(ti, s, tr * c);
        im[i1] = __builtin_fmaf(-tr, s, ti * c);
    }
}

template <int R, int H>
__device__ __forceinline__ void fft_stage(float (&re)[R], float (&im)[R]) {
    if constexpr (H >= 1) {
        static_for<0, R / (2 * H)>([&](auto bc) {
            constexpr int blk = decltype(bc)::value * 2 * H;
            static_for<0, H>([&](auto jc) { fft_bfly<R, H, blk, decltype(jc)::value>(re, im); });
        });
        fft_stage<R, H / 2>(re, im);
    }
}

template <int R>
__device__ __forceinline__ void fft_inreg(float (&re)[R], float (&im)[R]) {
    fft_stage<R, R / 2>(re, im);
}

// unaligned-safe vector types (global multi-dword accesses only need dword alignment on gfx9)
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int64_t reflect_idx(int64_t i, int64_t T) {
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    // out-of-contract inputs (pad >= T) are rejected on the host; clamp keeps loads in bounds
    if (i < 0) i = 0;
    if (i >= T) i = T - 1;
    return i;
}

__device__ __forceinline__ int reflect_idx32(int i, int T) {
    i = i < 0 ? -i : i;
    i = i >= T ? 2 * (T - 1) - i : i;
    i = i < 0 ? 0 : i;
    return i >= T ? T - 1 : i;
}

// wave-uniform buffer resource descriptor.  The inputs ARE uniform (kernel arguments, tile index);
// readfirstlane makes that provable, otherwise hipcc wraps every buffer op in a waterfall loop.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_uniform_rsrc(const void *p, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// log epilogue of the mel kernels: log(max(mel, pre) + off), pre < 0 disables the inner max
__device__ __forceinline__ float log_apply(float mel, int kind, float off, float pre) {
    float v = mel;
    if (pre >= 0.f) v = fmaxf(v, pre);
    v += off;
    if (kind == PSND_LOG_E) return logf(v);
    if (kind == PSND_LOG_10) return log10f(v);
    return v;
}
