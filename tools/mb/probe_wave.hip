// probes for psnd_stft_w.hip: v_permlane32_swap semantics, in-wave LDS write -> read ordering, ds_bpermute pull semantics
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    __shared__ unsigned buf[64 * 33];
    const int lane = threadIdx.x & 63;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
    out[128 + lane] = __builtin_amdgcn_ds_bpermute((63 - lane) * 4, (int)(300 + lane));
    // in-wave transpose of a 32 x 32 matrix per half-wave, half by half, no barrier
    const int lam = lane & 31, g = lane >> 5;
    unsigned v[32], rd[32];
    for (int i = 0; i < 32; ++i) v[i] = (g << 16) | (lam << 8) | i;
    for (int hh = 0; hh < 2; ++hh) {
        if (g == hh) {
            for (int q = 0; q < 32; ++q) buf[q * 33 + lam] = v[q];
            for (int l2 = 0; l2 < 32; ++l2) rd[l2] = buf[lam * 33 + l2];
        }
    }
    unsigned bad = 0;
    for (int l2 = 0; l2 < 32; ++l2) bad += rd[l2] != (unsigned)((g << 16) | (l2 << 8) | lam);
    out[192 + lane] = bad;
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("swap r0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
    printf("swap r1: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
    printf("bperm: lane0 %u lane5 %u lane63 %u\n", h[128], h[133], h[191]);
    unsigned bad = 0;
    for (int i = 0; i < 64; ++i) bad += h[192 + i];
    printf("transpose mismatches: %u\n", bad);
    return 0;
}
