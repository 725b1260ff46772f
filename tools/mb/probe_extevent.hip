// probe: hipEventRecordWithFlags(hipEventRecordExternal) inside a captured graph, waited on by a side stream after the
// graph launch.  Prints OVERLAP / SERIAL / BROKEN (see tools/probe_graph_events.py).   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__global__ void spin(float *p, long long cycles, float v) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}
__global__ void snap(const float *a, const float *b, float *out) { out[0] = *a; out[1] = *b; }
int main() {
    float *m; CK(hipMalloc(&m, 16)); CK(hipMemset(m, 0, 16));
    float *hout; CK(hipHostMalloc(&hout, 8));
    hipStream_t s, side; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&side));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    spin<<<1, 64, 0, s>>>(m, 2000000, 1.f);          // segment A ~1 ms
    hipError_t er = hipEventRecordWithFlags(ev, s, hipEventRecordExternal);
    printf("record external during capture: %s\n", hipGetErrorString(er));
    spin<<<1, 64, 0, s>>>(m + 1, 20000000, 2.f);     // segment B ~10 ms
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int trial = 0; trial < 3; ++trial) {
        CK(hipMemsetAsync(m, 0, 16, s));
        CK(hipStreamSynchronize(s));
        CK(hipGraphLaunch(ge, s));
        hipError_t ew = hipStreamWaitEvent(side, ev, 0);
        snap<<<1, 1, 0, side>>>(m, m + 1, m + 2);
        CK(hipMemcpyAsync(hout, m + 2, 8, hipMemcpyDeviceToHost, side));
        CK(hipDeviceSynchronize());
        printf("trial %d: wait=%s markers (A, B) = (%g, %g)\n", trial, hipGetErrorString(ew), hout[0], hout[1]);
    }
    if (hout[0] == 1.f && hout[1] == 0.f) printf("EXTERNAL_EVENT_HIP: OVERLAP\n");
    else if (hout[0] == 1.f && hout[1] == 2.f) printf("EXTERNAL_EVENT_HIP: SERIAL\n");
    else printf("EXTERNAL_EVENT_HIP: BROKEN\n");
    return 0;
}
