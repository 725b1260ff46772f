// Do VALU and LDS work of DIFFERENT waves overlap on a gfx950 CU?  One 1024-thread workgroup per CU (16 waves, 4 per SIMD).  Per loop
// iteration a wave runs NV independent-chain v_pk_fma_f32 and NL ds_read_b64 (conflict-free, 512 B per instruction):
//   mode 0  VALU only                 mode 1  LDS only (wait at the end of the iteration)
//   mode 2  reads issued, then the VALU block, then the wait (overlap inside the wave)
//   mode 3  reads, wait, VALU block (phases of one wave serial; only OTHER waves can overlap them)
//   mode 4  as 3 with a transpose-like burst: NL/2 ds_write_b64 + NL/2 ds_read_b64
//   mode 5  as 3, but waves start staggered (s_sleep by wave id) so that the phases of the 4 waves of a SIMD do not line up
// Build: hipcc --offload-arch=gfx950 -O3 mb_valu_lds.hip -o mb_valu_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define VBLK32 REP8(XV) REP8(XV) REP8(XV) REP8(XV)
#define XV(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##n) : "v"(pa), "v"(pb));
#define XL(n) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q##n) : "v"(addr), "n"(n * 512));
#define XW(n) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(q##n), "n"(n * 512) : "memory");

template <int MODE, int NV32, int NL8>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a, float b) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float x0 = threadIdx.x;
    v2f p0 = {x0, x0 + 1}, p1 = {x0 + 2, x0}, p2 = {x0 + 4, x0}, p3 = {x0 + 6, x0}, p4 = {x0, x0 + 3}, p5 = {x0 + 3, x0}, p6 = {x0, x0 - 1}, p7 = {x0 + 7, 1.f};
    v2f q0 = p0, q1 = p1, q2 = p2, q3 = p3, q4 = p4, q5 = p5, q6 = p6, q7 = p7;
    v2f pa = {a, a}, pb = {b, b};
    for (int i = threadIdx.x; i < 16 * 1024 * 2; i += 1024) smem[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(smem) + w * 8192 + lane * 8;      // 4 KB + per wave: 8 reads x 512 B
    if (MODE == 5) for (int i = 0; i < (w >> 2) * 40; ++i) __builtin_amdgcn_s_sleep(8);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < NV32; ++r) { VBLK32 }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < NL8; ++r) { REP8(XL) }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < NL8; ++r) { REP8(XL) }
#pragma unroll
            for (int r = 0; r < NV32; ++r) { VBLK32 }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
        } else if (MODE == 3 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < NL8; ++r) { REP8(XL) }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
#pragma unroll
            for (int r = 0; r < NV32; ++r) { VBLK32 }
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < NL8 / 2; ++r) { REP8(XW) }
#pragma unroll
            for (int r = 0; r < NL8 / 2; ++r) { REP8(XL) }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
#pragma unroll
            for (int r = 0; r < NV32; ++r) { VBLK32 }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x;
}

template <int MODE, int NV32, int NL8>
float run(float *out) {
    const int iters = 400;
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    auto kern = k<MODE, NV32, NL8>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 128 * 1024, 0, out, 8, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(ev0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 128 * 1024, 0, out, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(ev1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
        if (ms < best) best = ms;
    }
    return best * 1e3 / iters;       // us per iteration
}

template <int NV32, int NL8>
void suite(float *out) {
    const float v = run<0, NV32, NL8>(out), l = run<1, NV32, NL8>(out), b2 = run<2, NV32, NL8>(out), b3 = run<3, NV32, NL8>(out),
                b4 = run<4, NV32, NL8>(out), b5 = run<5, NV32, NL8>(out);
    printf("per iteration and wave: %4d v_pk_fma, %3d ds_read_b64 | VALU only %.3f us  LDS only %.3f us | in-wave overlap %.3f  phased %.3f  phased w+r %.3f  phased staggered %.3f | sum %.3f max %.3f\n",
           32 * NV32, 8 * NL8, v, l, b2, b3, b4, b5, v + l, v > l ? v : l);
}

int main() {
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4));
    suite<8, 8>(out);      // 256 VALU, 64 LDS reads (a radix-32 + a transpose)
    suite<8, 4>(out);
    suite<8, 16>(out);
    suite<2, 1>(out);      // fine-grained alternation: 64 VALU, 8 reads
    suite<1, 1>(out);
    suite<16, 2>(out);
    return 0;
}
