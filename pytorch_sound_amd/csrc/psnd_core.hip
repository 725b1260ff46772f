// psnd_core.hip - version, error string, integer framing contract (host side).
#include "psnd_common.h"

static thread_local char g_err[512] = "";

void psnd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int psnd_version(void) { return 137; }  // 0.1.37: psnd_groupnorm1_* take 2 C doubles of scratch per sample (one pair of sums per row: no atomics; round 6); 0.1.36: + psnd_linear1x1_fwd_ex, psnd_linear1x1_bwd_ex takes io_h (round 6: the hidden tensor of Conv1d -> ReLU -> Conv1d stored as bf16 under autocast); 0.1.35: + psnd_linear1x1_bwd_ex (round 6: the ReLU between two projections masks in the producing GEMM); 0.1.34: + psnd_im2col_f32 / psnd_col2im_f32 (round 6: the fp32 instance of the conv stack); 0.1.33: + psnd_cl_colsum, psnd_grad_pack_bf16 / _unpack_bf16 (round 5); 0.1.32: + psnd_mha_bwd_parts, psnd_convtr1d_cl_bwd takes either role alone (round 4); 0.1.31: psnd_mha_fwd / _bwd take `bf16` (round 3); 0.1.30: + psnd_polar_bwd, psnd_grad_sumsq

extern "C" const char *psnd_last_error(void) { return g_err; }

#ifdef PSND_LAB
std::atomic<int> g_psnd_env_gen{0};
extern "C" void psnd_env_refresh(void) { g_psnd_env_gen.fetch_add(1, std::memory_order_acq_rel); }
#endif

static inline int64_t pad_of(int n_fft, int hop, int framing) {
    if (framing == PSND_FRAMING_NONE) return 0;
    return framing == PSND_FRAMING_CENTER ? n_fft / 2 : (n_fft - hop) / 2;
}

// pytorch_sound/models/transforms.py:55-66 (pad n/2 + conv1d stride hop, no padding) and
// :352-360 (pad (n-h)/2 + torch.stft(center=False)).
extern "C" int64_t psnd_frame_count(int64_t T, int n_fft, int hop, int framing) {
    if (T <= 0 || n_fft <= 0 || hop <= 0) return 0;
    const int64_t L = T + 2 * pad_of(n_fft, hop, framing);
    if (L < n_fft) return 0;
    return (L - n_fft) / hop + 1;
}

// F.pad(mode='reflect'): x[-i] = x[i], x[T-1+i] = x[T-1-i]
extern "C" int64_t psnd_frame_sample_index(int64_t f, int m, int64_t T, int n_fft, int hop, int framing) {
    int64_t i = f * hop - pad_of(n_fft, hop, framing) + m;
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    return i;
}

// ---- stream events for the data-parallel step (pytorch_sound_amd/distributed.py) ---------------------------------------
// The replayed hipGraph of a training step must release the RCCL all-reduce of gradient bucket i as soon as the captured
// backward has produced it - while the rest of the backward still runs.  A record node of an EXTERNAL event
// (hipEventRecordWithFlags(..., hipEventRecordExternal)) inside the captured stream does that: after hipGraphLaunch, a
// hipStreamWaitEvent on another stream waits for exactly that node of THIS launch.  torch's Event wrapper refuses external
// events on ROCm builds, hence these four entry points (measured on MI355X / ROCm 7: the waiting stream is released at the node,
// tools/mb/probe_extevent.hip).
extern "C" void *psnd_event_create(void) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        psnd_set_error("event_create: %s", hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    return ev;
}
extern "C" int psnd_event_destroy(void *ev) {
    if (!ev) return PSND_OK;
    if (hipEventDestroy(static_cast<hipEvent_t>(ev)) != hipSuccess) PSND_FAIL(PSND_E_HIP, "event_destroy: %s", hipGetErrorString(hipGetLastError()));
    return PSND_OK;
}
extern "C" int psnd_event_record_external(void *ev, void *stream) {
    if (!ev) PSND_FAIL(PSND_E_ARG, "event_record_external: null event");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) st = hipStreamCaptureStatusNone;
    const hipError_t e = st == hipStreamCaptureStatusActive ? hipEventRecordWithFlags(static_cast<hipEvent_t>(ev), s, hipEventRecordExternal)
                                                            : hipEventRecord(static_cast<hipEvent_t>(ev), s);
    if (e != hipSuccess) {
        (void)hipGetLastError();            // do not leave the failure behind as the runtime's sticky "last error"
        PSND_FAIL(PSND_E_HIP, "event_record_external: %s", hipGetErrorString(e));
    }
    return PSND_OK;
}
// 1 when this HIP runtime accepts an external event-record node inside a stream capture (a throw-away capture on a private stream;
// nothing is launched), 0 otherwise.  Runtimes differ: ROCm 7.2's does, the 7.0 one bundled with some torch wheels may not.
extern "C" int psnd_event_external_supported(void) {
    static int cached = -1;
    if (cached >= 0) return cached;
    int ok = 0;
    hipStream_t s = nullptr;
    hipEvent_t ev = nullptr;
    hipGraph_t g = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const hipError_t e = hipEventRecordWithFlags(ev, s, hipEventRecordExternal);
            const hipError_t e2 = hipStreamEndCapture(s, &g);
            ok = (e == hipSuccess && e2 == hipSuccess) ? 1 : 0;
            if (g) (void)hipGraphDestroy(g);
        }
    }
    if (ev) (void)hipEventDestroy(ev);
    if (s) (void)hipStreamDestroy(s);
    (void)hipGetLastError();
    cached = ok;
    return ok;
}
extern "C" int psnd_stream_wait_event(void *stream, void *ev) {
    if (!ev) PSND_FAIL(PSND_E_ARG, "stream_wait_event: null event");
    const hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(ev), 0);
    if (e != hipSuccess) PSND_FAIL(PSND_E_HIP, "stream_wait_event: %s", hipGetErrorString(e));
    return PSND_OK;
}

// ---- gradient of (magnitude, phase) -> gradient of (re, im):  STFTTorchAudio.transform (transforms.py:305-311) keeps the
// phase differentiable: with X = m e^{i phi}, d m = cos phi d re + sin phi d im, d phi = (re d im - im d re) / m^2, so
//   g_re = g_m cos phi - g_phi sin phi / m,   g_im = g_m sin phi + g_phi cos phi / m     (m = 0: inf / NaN, as atan2's own gradient).
// One pass; the result feeds psnd_stft_bwd(gre, gim).  Bound: HBM, 24 bytes per element.
__global__ __launch_bounds__(256) void polar_bwd_kernel(const float *__restrict__ gmag, const float *__restrict__ gphase,
                                                        const float *__restrict__ mag, const float *__restrict__ phase, long long n,
                                                        float *__restrict__ gre, float *__restrict__ gim) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        float gm[4] = {0.f, 0.f, 0.f, 0.f}, gp[4] = {0.f, 0.f, 0.f, 0.f}, m[4], ph[4], o_re[4], o_im[4];
        const int cnt = n - i >= 4 ? 4 : (int)(n - i);
        if (cnt == 4 && (n & 3) == 0) {
            *reinterpret_cast<float4 *>(m) = *reinterpret_cast<const float4 *>(mag + i);
            *reinterpret_cast<float4 *>(ph) = *reinterpret_cast<const float4 *>(phase + i);
            if (gmag) *reinterpret_cast<float4 *>(gm) = *reinterpret_cast<const float4 *>(gmag + i);
            if (gphase) *reinterpret_cast<float4 *>(gp) = *reinterpret_cast<const float4 *>(gphase + i);
        } else {
            for (int j = 0; j < cnt; ++j) {
                m[j] = mag[i + j], ph[j] = phase[i + j];
                if (gmag) gm[j] = gmag[i + j];
                if (gphase) gp[j] = gphase[i + j];
            }
        }
        for (int j = 0; j < 4; ++j) {
            float sn, cs;
            sn = sinf(ph[j]), cs = cosf(ph[j]);
            const float q = gphase ? gp[j] / m[j] : 0.f;
            o_re[j] = gm[j] * cs - q * sn;
            o_im[j] = gm[j] * sn + q * cs;
        }
        if (cnt == 4 && (n & 3) == 0) {
            *reinterpret_cast<float4 *>(gre + i) = *reinterpret_cast<float4 *>(o_re);
            *reinterpret_cast<float4 *>(gim + i) = *reinterpret_cast<float4 *>(o_im);
        } else {
            for (int j = 0; j < cnt; ++j) gre[i + j] = o_re[j], gim[i + j] = o_im[j];
        }
    }
}
extern "C" int psnd_polar_bwd(const float *gmag, const float *gphase, const float *mag, const float *phase, int64_t n, float *gre,
                              float *gim, void *stream) {
    if (!mag || !phase || !gre || !gim) PSND_FAIL(PSND_E_ARG, "polar_bwd: null mag/phase/gre/gim");
    if (!gmag && !gphase) PSND_FAIL(PSND_E_ARG, "polar_bwd: neither gmag nor gphase given");
    if (n < 0) PSND_FAIL(PSND_E_ARG, "polar_bwd: n=%lld", (long long)n);
    if (n == 0) return PSND_OK;
    long long blocks = (n + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(polar_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), gmag, gphase, mag, phase,
                       (long long)n, gre, gim);
    PSND_CHECK_LAUNCH("polar_bwd");
    return PSND_OK;
}
