// HBM ceiling for the STFT stage's traffic MIX (1 byte read : 2 bytes written, SURVEY 8(d): 4 hop read + 4 K written per frame):
// each thread reads one float4 and writes two (plain / nontemporal), grid-stride, 544 MB per launch as the judged launch.  Also 1:1 (copy) and 0:1 (fill).
// usage: mb_mix [MB_read=181] [launches=600]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int W, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = W == 0 ? f4{1.f, 2.f, 3.f, 4.f} : (NT ? __builtin_nontemporal_load(in + i) : in[i]);
        if (W == 0 || W == 1) {
            if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
        } else {
            // two 16-byte stores per load, each wave writing 2 KiB contiguous: out[2 * (i - lane) + lane] and + 64
            const size_t base = 2 * (i - (threadIdx.x & 63)) + (threadIdx.x & 63);
            if (NT) { __builtin_nontemporal_store(v, out + base); __builtin_nontemporal_store(v, out + base + 64); }
            else { out[base] = v; out[base + 64] = v; }
        }
    }
}
template <int W, bool NT>
void run(const char *name, const f4 *in, f4 *out, size_t n, double bytes, int L, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 600; ++i) mix_kernel<W, NT><<<grid, 256>>>(in, out, n);
    std::vector<float> ts;
    for (int i = 0; i < L; ++i) {
        hipEventRecord(e0); mix_kernel<W, NT><<<grid, 256>>>(in, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1e3f);
    }
    // back-to-back block too
    hipEventRecord(e0); for (int i = 0; i < L; ++i) mix_kernel<W, NT><<<grid, 256>>>(in, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::sort(ts.begin(), ts.end());
    printf("%-34s grid %5d  median %.1f us  b2b %.1f us/launch -> %.3f of 8 TB/s (%.2f TB/s)\n", name, grid, ts[ts.size() / 2], ms * 1e3 / L, bytes / (ms * 1e-3 / L) / 8e12, bytes / (ms * 1e-3 / L) / 1e12);
}
int main(int argc, char **argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 181.4;
    const int L = argc > 2 ? atoi(argv[2]) : 200;
    const size_t n = (size_t)(mb * 1e6 / 16) / 64 * 64;
    f4 *in, *out; hipMalloc(&in, n * 16 * 2); hipMalloc(&out, n * 32 + 4096);
    hipMemset(in, 0, n * 32);
    for (int grid : {2048, 8192, 65536}) {
        run<2, false>("read 1 : write 2", in, out, n, n * 48.0, L, grid);
        run<2, true>("read 1 : write 2 (nontemporal)", in, out, n, n * 48.0, L, grid);
    }
    run<1, false>("copy 1 : 1 (same total bytes)", in, out, n * 3 / 2, n * 48.0, L, 8192);
    run<1, true>("copy 1 : 1 (nontemporal)", in, out, n * 3 / 2, n * 48.0, L, 8192);
    run<0, false>("fill 0 : 1 (same total bytes)", in, out, n * 2, n * 32.0, L, 8192);
    run<0, true>("fill 0 : 1 (nontemporal)", in, out, n * 2, n * 32.0, L, 8192);
    return 0;
}
