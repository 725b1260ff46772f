// mb_groupsync.hip - cost of an 8-workgroup hand-over through global memory (release fence + counter + acquire fence + reload of the
// exchanged tile), the building block of a per-clip conv chain: 256 workgroups = 32 groups of 8 (consecutive block ids), every round each
// workgroup writes its 11 KB slice of the group's 88 KB tile, arrives, waits for the other seven, reads the whole tile into LDS.
// hipcc --offload-arch=gfx950 -O3 tools/mb/mb_groupsync.hip -o tools/mb/mb_groupsync && tools/mb/mb_groupsync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int ROWS = 173, C = 256, SL = 32, GROUP = 8, ROUNDS = 6;

template <int MODE>   // 0: agent-scope fences (correct); 1: no fences (timing only); 2: no exchange at all (floor)
__global__ __launch_bounds__(384) void k(unsigned short *buf, unsigned *cnt, unsigned base, int *errs) {
    extern __shared__ uint4 lds[];
    const int g = blockIdx.x / GROUP, s = blockIdx.x % GROUP, tid = threadIdx.x;
    unsigned short *tile = buf + (size_t)g * 2 * ROWS * C;     // two tiles per group (ping-pong between rounds)
    unsigned acc = 0;
    for (int r = 0; r < ROUNDS; ++r) {
        unsigned short *t = tile + (size_t)(r & 1) * ROWS * C;
        // write my slice: ROWS x 32 channels, 64 B per row -> 4 x 16 B pieces per row
        for (int i = tid; i < ROWS * 4; i += blockDim.x) {
            const int row = i >> 2, pc = i & 3;
            const unsigned v = (base + r) * 16 + s;
            *reinterpret_cast<uint4 *>(t + (size_t)row * C + s * SL + pc * 8) = make_uint4(v, v, v, v);
        }
        if (MODE != 2) {
            if (MODE == 0) __threadfence();
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(&cnt[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned need = (base * ROUNDS + r + 1) * GROUP;
                int spins = 0;
                while (__hip_atomic_load(&cnt[g], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { atomicAdd(errs + 1, 1); break; }
                }
            }
            __syncthreads();
            if (MODE == 0) __threadfence();
        }
        // read the whole tile into LDS
        for (int i = tid; i < ROWS * C / 8; i += blockDim.x) lds[i] = reinterpret_cast<const uint4 *>(t)[i];
        __syncthreads();
        // check: piece p of row holds (base + r) * 16 + slice
        for (int i = tid; i < ROWS * C / 8; i += blockDim.x) {
            const int sl = (i % (C / 8)) / 4;
            const unsigned want = (base + r) * 16 + sl;
            if (MODE != 2 && lds[i].x != want) acc++;
        }
        __syncthreads();
    }
    if (acc) atomicAdd(errs, (int)acc);
}

int main() {
    const int NB = 256;
    unsigned short *buf; unsigned *cnt; int *errs;
    hipMalloc(&buf, (size_t)(NB / GROUP) * 2 * ROWS * C * 2);
    hipMalloc(&cnt, 64 * 4); hipMalloc(&errs, 8);
    hipMemset(buf, 0, (size_t)(NB / GROUP) * 2 * ROWS * C * 2);
    const size_t lds = (size_t)ROWS * C * 2;
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(cnt, 0, 64 * 4); hipMemset(errs, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 200;
        unsigned base = 0;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(384), lds, 0, buf, cnt, base, errs);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(NB), dim3(384), lds, 0, buf, cnt, base, errs);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(NB), dim3(384), lds, 0, buf, cnt, base, errs);
            ++base;
        };
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int h[2]; hipMemcpy(h, errs, 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.2f us per launch of %d rounds = %.2f us per round; mismatches %d, spin timeouts %d\n", mode,
               mode == 0 ? "agent fences" : mode == 1 ? "no fences" : "no exchange", ms * 1e3 / iters, ROUNDS, ms * 1e3 / iters / ROUNDS, h[0], h[1]);
    }
    return 0;
}
