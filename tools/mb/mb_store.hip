// store-pattern microbenchmark: write an (N,K,F) fp32 tensor in (clip, frame-tile) tiles the way the STFT
// kernel does, with different chunk widths / store widths / tile->block mappings.  No compute, no loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));

// mode 0: dword stores, lanes = 16 frames x 4 rows (current kernel)        chunk = 16 frames
// mode 1: dwordx4 stores, lane = (row, 4-frame group): 16 rows x 64 B per instr   chunk = 16 frames
// mode 2: dword stores, lanes = 32 frames x 2 rows                           chunk = 32 frames
// mode 3: dwordx4 stores, lane = (row, 4-frame group): 8 rows x 128 B         chunk = 32 frames
// mode 4: dwordx4, 64-frame chunk: 4 rows x 256 B
// mode 5: dwordx4, full row (F frames) per wave-loop: rows contiguous
template <int MODE>
__global__ __launch_bounds__(256) void k_store(float *out, int N, int K, int F, int ntile, int total, int xcd_map) {
    constexpr int FT = (MODE <= 1) ? 16 : (MODE <= 3 ? 32 : 64);
    const int t = threadIdx.x;
    int nb = gridDim.x;
    for (int b = blockIdx.x; b < total; b += nb) {
        int tile = b;
        if (xcd_map) { int xcd = b & 7, bx = b >> 3, chunk = (total + 7) >> 3; tile = xcd * chunk + bx; if (tile >= total || bx >= chunk) continue; }
        const int clip = tile / ntile, f0 = (tile - clip * ntile) * FT;
        float *o = out + (size_t)clip * K * F + f0;
        if (MODE == 0 || MODE == 2) {
            const int f = t % FT, g = t / FT;
            if (f0 + f < F)
                for (int k = g; k < K; k += 256 / FT) o[(size_t)k * F + f] = (float)k;
        } else {
            constexpr int FG = FT / 4;               // 4-frame groups per row
            const int fg = t % FG, g = t / FG;
            for (int k = g; k < K; k += 256 / FG) {
                float *pp = o + (size_t)k * F + 4 * fg;
                if (f0 + 4 * fg + 3 < F) *reinterpret_cast<f4 *>(pp) = f4{(float)k, 1.f, 2.f, 3.f};
                else for (int j = 0; j < 4; ++j) if (f0 + 4 * fg + j < F) pp[j] = (float)k;
            }
        }
    }
}
// mode K: exactly the row order of stft_fwd_n1024_kernel::post_emit.  variant 0: qq = w + 4 j (rows of a wave 4 apart,
// two converging sweeps); variant 1: qq = 4 w + j (consecutive rows); variant 2: consecutive + single ascending sweep
__global__ __launch_bounds__(256) void k_kernelpat(float *out, int N, int K, int F, int ntile, int total, int variant) {
    const int t = threadIdx.x, f = t & 15, w = t >> 6, j = (t >> 4) & 3;
    const int qq = variant == 0 ? (w + 4 * j) : (4 * w + j);
    int nb = gridDim.x;
    int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, chunk = (total + 7) >> 3, nbx = (nb - xcd + 7) >> 3;
    for (int tile = xcd * chunk + bx; tile < min((xcd + 1) * chunk, total); tile += nbx) {
        const int clip = tile / ntile, f0 = (tile - clip * ntile) * 16;
        float *o = out + (size_t)clip * K * F + f0 + f;
        if (f0 + f >= F) continue;
        if (variant == 2) {
            for (int pp = 0; pp < 16; ++pp) { o[(size_t)(qq + 32 * pp) * F] = 1.f; o[(size_t)(32 - qq + 32 * pp - (qq == 0 ? 16 : 0)) * F] = 2.f; }
            if (qq == 0) o[(size_t)512 * F] = 3.f;
        } else if (qq != 0) {
            for (int pp = 0; pp < 8; ++pp) {
                o[(size_t)(qq + 32 * pp) * F] = 1.f; o[(size_t)(32 - qq + 32 * (15 - pp)) * F] = 2.f;
                o[(size_t)(32 - qq + 32 * pp) * F] = 3.f; o[(size_t)(qq + 32 * (15 - pp)) * F] = 4.f;
            }
        } else {
            for (int pp = 0; pp <= 8; ++pp) { o[(size_t)(32 * pp) * F] = 1.f; if (pp != 8) o[(size_t)(32 * (16 - pp)) * F] = 2.f; }
            for (int pp = 0; pp < 8; ++pp) { o[(size_t)(16 + 32 * pp) * F] = 1.f; o[(size_t)(16 + 32 * (15 - pp)) * F] = 2.f; }
        }
    }
}
__global__ __launch_bounds__(256) void k_rows(float *out, long long total4) {   // plain contiguous fill, 16 B per lane
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total4; i += (long long)gridDim.x * 256)
        reinterpret_cast<f4 *>(out)[i] = f4{1.f, 2.f, 3.f, 4.f};
}
template <int MODE> float run(float *d, int N, int K, int F, int grid, int xcd) {
    constexpr int FT = (MODE <= 1) ? 16 : (MODE <= 3 ? 32 : 64);
    int ntile = (F + FT - 1) / FT, total = ntile * N;
    if (grid <= 0 || grid > total) grid = total;
    grid = (grid + 7) & ~7;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, d, N, K, F, ntile, total, xcd);
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, d, N, K, F, ntile, total, xcd);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}
int main(int argc, char **argv) {
    int N = argc > 1 ? atoi(argv[1]) : 1024, K = 513, F = argc > 2 ? atoi(argv[2]) : 173;
    size_t bytes = (size_t)N * K * F * 4;
    float *d; CK(hipMalloc(&d, bytes + 4096));
    printf("N=%d K=%d F=%d  %.1f MB\n", N, K, F, bytes / 1e6);
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_rows, dim3(4096), dim3(256), 0, 0, d, (long long)(bytes / 16));
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_rows, dim3(4096), dim3(256), 0, 0, d, (long long)(bytes / 16));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
        printf("contiguous fill 16B/lane           : %7.1f us  %6.0f GB/s\n", ms * 1e3, bytes / ms / 1e6);
    }
    for (int variant = 0; variant < 3; ++variant)
        for (int grid : {3072, 11264}) {
            int ntile = (F + 15) / 16, total = ntile * N;
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_kernelpat, dim3(grid), dim3(256), 0, 0, d, N, K, F, ntile, total, variant);
            CK(hipEventRecord(a));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_kernelpat, dim3(grid), dim3(256), 0, 0, d, N, K, F, ntile, total, variant);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
            printf("kernel pattern variant %d grid %5d        : %7.1f us  %6.0f GB/s\n", variant, grid, ms * 1e3, bytes / ms / 1e6);
        }
    const char *names[] = {"dword  x 16-frame chunk (64 B)", "dwordx4 x 16-frame chunk (64 B)", "dword  x 32-frame chunk (128 B)",
                           "dwordx4 x 32-frame chunk (128 B)", "dwordx4 x 64-frame chunk (256 B)"};
    for (int xcd = 0; xcd < 2; ++xcd)
        for (int grid : {0, 2048, 768}) {
            float ms[5] = {run<0>(d, N, K, F, grid, xcd), run<1>(d, N, K, F, grid, xcd), run<2>(d, N, K, F, grid, xcd),
                           run<3>(d, N, K, F, grid, xcd), run<4>(d, N, K, F, grid, xcd)};
            for (int m = 0; m < 5; ++m)
                printf("xcd_map=%d grid=%5d %-34s: %7.1f us  %6.0f GB/s\n", xcd, grid, names[m], ms[m] * 1e3, bytes / ms[m] / 1e6);
        }
    return 0;
}
