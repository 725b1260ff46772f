// The experiment DESIGN.md (round 1, section 9) left untried: the second (radix-16) pass of the n = 1024 STFT on the f32 matrix
// cores instead of the VALU, alone and co-issued with VALU butterflies on the same SIMD.
//   pass 2 of a frame = 32 DFTs of 16 complex points = one real (32 x 32) x (32 x 32) product  [Re; Im] = W [Re; Im]
//   VALU form   : pk::fft<16> on two rows per thread (what stft_fwd_n1024_kernel does), rows from / to LDS
//   MFMA form   : 16 x v_mfma_f32_32x32x2_f32 per frame, W in registers (A operand), the rows from LDS (B operand), result to LDS
//   co-issue    : 8 waves per workgroup - waves 0-3 the VALU form, waves 4-7 the MFMA form, one of each per SIMD
// Prints shader cycles per frame and SIMD (s_memtime), the numbers quoted in DESIGN.md 4.1.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../pytorch_sound_amd/csrc mb_mfma_dft.hip -o mb_mfma_dft
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "psnd_pk.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int ROWP = 34;     // LDS row pitch in floats (16 complex + pad)

// frames per iteration and wave: VALU form 64 threads x 2 rows = 128 rows = 4 frames; MFMA form: 4 frames as well
__device__ __forceinline__ void valu_form(float *lds, int lane) {
    v2f za[16], zb[16];
    float *ra = lds + (2 * lane) * ROWP, *rb = ra + ROWP;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 va = *reinterpret_cast<const f32x4 *>(ra + 4 * i), vb = *reinterpret_cast<const f32x4 *>(rb + 4 * i);
        za[2 * i] = pk::lo(va), za[2 * i + 1] = pk::hi(va);
        zb[2 * i] = pk::lo(vb), zb[2 * i + 1] = pk::hi(vb);
    }
    pk::fft<16>(za);
    pk::fft<16>(zb);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        *reinterpret_cast<f32x4 *>(ra + 4 * i) = f32x4{za[2 * i].x, za[2 * i].y, za[2 * i + 1].x, za[2 * i + 1].y};
        *reinterpret_cast<f32x4 *>(rb + 4 * i) = f32x4{zb[2 * i].x, zb[2 * i].y, zb[2 * i + 1].x, zb[2 * i + 1].y};
    }
}
// one frame: X (32 real rows = re/im of 16 points) x 32 columns (the frame's 32 DFTs), tile [k][col] in LDS with pitch 33
__device__ __forceinline__ void mfma_form(float *tile, const float (&w)[16], int lane) {
    const int li = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        float *t = tile + f * 32 * 33;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s], t[(2 * s + kk) * 33 + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * kk) * 33 + li] = acc[r];
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void k(long long *cyc, float *sink, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8][4 * 32 * 34];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *mine = lds[wave];
    for (int i = lane; i < 4 * 32 * 34; i += 64) mine[i] = 1e-3f * (float)((i * 7 + wave) % 97);
    float w[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) w[s] = __cosf(0.19634954f * (float)(((lane & 31) * (2 * s + (lane >> 5))) & 31));   // some DFT-like matrix
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || (MODE == 2 && wave < 4)) valu_form(mine, lane);
        else mfma_form(mine, w, lane);
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = mine[lane];
}

template <int MODE>
void run(const char *name, int threads, long long *dcyc, float *sink) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, dcyc, sink, 10);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, dcyc, sink, iters);
    CK(hipDeviceSynchronize());
    static long long h[256 * 8];
    CK(hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost));
    const int nw = threads / 64;
    double sv = 0, sm = 0; int nv = 0, nm = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < nw; ++w) {
            const bool is_valu = MODE == 0 || (MODE == 2 && w < 4);
            (is_valu ? sv : sm) += (double)h[b * 8 + w];
            (is_valu ? nv : nm)++;
        }
    // every wave processes 4 frames per iteration
    if (nv) printf("%-44s VALU waves: %8.1f cycles per frame per SIMD\n", name, sv / nv / iters / 4.0 / (MODE == 2 ? 1.0 : nw / 4.0));
    if (nm) printf("%-44s MFMA waves: %8.1f cycles per frame per SIMD\n", name, sm / nm / iters / 4.0 / (MODE == 2 ? 1.0 : nw / 4.0));
}

int main() {
    long long *dcyc; float *sink;
    CK(hipMalloc(&dcyc, 256 * 8 * sizeof(long long)));
    CK(hipMalloc(&sink, 256 * 512 * sizeof(float)));
    printf("cycles a SIMD spends per frame = wave wall cycles / frames per wave / waves per SIMD; co-issue rows: per wave, the two run side by side on one SIMD\n");
    run<0>("VALU radix-16 pass, 1 wave per SIMD", 256, dcyc, sink);
    run<0>("VALU radix-16 pass, 2 waves per SIMD", 512, dcyc, sink);
    run<1>("MFMA 32x32x2 f32 pass, 1 wave per SIMD", 256, dcyc, sink);
    run<1>("MFMA 32x32x2 f32 pass, 2 waves per SIMD", 512, dcyc, sink);
    run<2>("co-issue: 1 VALU + 1 MFMA wave per SIMD", 512, dcyc, sink);
    return 0;
}
