// probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds e[i] = i (u16).  Every lane passes a byte address; prints what each lane gets.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int mode, unsigned *out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = 0;                                             // uniform address
    else if (mode == 1) addr = (unsigned)(l * 8);                        // lane-linear, 8 bytes apart
    else addr = (unsigned)((((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16) * 2);   // [row = i>>2][64 cols], cols 4 (i&3) + 16 group
    unsigned base = (unsigned)(size_t)lds;   // LDS addresses are 32-bit offsets
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + base));
    out[2 * l] = (unsigned)v;
    out[2 * l + 1] = (unsigned)(v >> 32);
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %4u %4u %4u %4u", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
