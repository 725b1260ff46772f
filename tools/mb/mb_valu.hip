// VALU issue-rate microbenchmark (gfx950): wave64 instructions per cycle per SIMD for scalar vs packed fp32
// and v_sqrt_f32, at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mb_valu.hip -o mb_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    v2f pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (MODE == 1) {   // 8 independent v_pk_fma_f32
#define X(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##n) : "v"(pa), "v"(pb));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (MODE == 2) {   // v_pk_add_f32
#define X(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##n) : "v"(pa));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (MODE == 3) {   // v_add_f32
#define X(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x##n) : "v"(a));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (MODE == 4) {   // v_sqrt_f32
#define X(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x##n));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (MODE == 5) {   // v_pk_mul_f32
#define X(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##n) : "v"(pa));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (MODE == 6) {   // v_pk_fma_f32 with op_sel swap
#define X(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(p##n) : "v"(pa), "v"(pb));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
}

template <int MODE>
void run(const char *name, float *out, int wps) {
    const int iters = 4096, blocks = 256 * 4, threads = 64 * wps * 4 / 4;   // blocks = 4 per CU -> one per SIMD if threads = 64
    // launch 256 CUs x 4 SIMDs x wps waves: blocks of 64*wps threads, 1024 blocks
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * wps), 0, 0, out, 16, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * wps), 0, 0, out, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(ev1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
    const double insts_per_simd = (double)iters * 32 * wps;   // each SIMD runs wps waves (1024 blocks / 1024 SIMDs)
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles(@2.4GHz) per wave-instruction per SIMD\n", name, wps, ms, cyc / insts_per_simd);
    (void)threads;
}

int main() {
    float *out; CK(hipMalloc(&out, 1024 * 1024 * 4));
    for (int wps : {1, 2, 4}) {
        run<0>("v_fma_f32", out, wps);
        run<3>("v_add_f32", out, wps);
        run<1>("v_pk_fma_f32", out, wps);
        run<6>("v_pk_fma_f32 op_sel", out, wps);
        run<2>("v_pk_add_f32", out, wps);
        run<5>("v_pk_mul_f32", out, wps);
        run<4>("v_sqrt_f32", out, wps);
    }
    return 0;
}
