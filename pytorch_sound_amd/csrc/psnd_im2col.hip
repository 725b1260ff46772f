// psnd_im2col.hip - the fp32 instance of the Conv1d / ConvTranspose1d stack (pytorch_sound/models/vocoders/hifi_gan.py:32-147 computes its
// convolutions in fp32): a convolution is the exact-fp32 matrix-core GEMM of psnd_linear1x1_* (v_mfma_f32_32x32x2_f32, psnd_attn.hip)
// over the unfolded input
//     col[n][ci * k + j][t] = act( x[n][ci][src(t, j)] ),      y = W.reshape(Cout, Cin * k) @ col + bias,
// with the leaky-relu that precedes every conv of a ResBlock (hifi_gan.py:58-61, 86-87) applied while gathering.  The two kernels here
// are the gather and its adjoint; both are HBM-bound element passes (k x the activation bytes written / read).  This is the PRECISION
// path (fp32 tensors outside autocast, held to the reference's goldens at 1e-4); the throughput path is the channels-last bf16
// implicit-GEMM stack of psnd_conv*.hip, selected under torch.autocast.
//
// Index map (one formula for both layer types):  s' = t * stride + j * dil - pad;  valid when s' >= 0, s' % up == 0, s' / up < T;
// src = s' / up.
//   Conv1d(k, dilation d, padding p):            stride 1, up 1, pad p            (F.conv1d)
//   ConvTranspose1d(k, stride u, padding p):     stride 1, up u, pad k - 1 - p, taps flipped by the caller (a convolution over the
//                                                zero-spread input, To = (T - 1) u - 2 p + k as hifi_gan.py:107-110 gives)
#include "psnd_common.h"

namespace {

struct ColParams {
    const float *x;      // (N, C, T)
    float *col;          // (N, C * k, To)
    const float *gcol;   // adjoint: (N, C * k, To)
    float *gx;           // adjoint: (N, C, T)
    long long T, To;
    int C, k, dil, pad, stride, up;
    float slope;         // act(v) = v > 0 ? v : slope * v   (1: identity)
    long long rows;      // rows of the launch (grid.y x grid.z may overshoot)
};

__device__ __forceinline__ long long src_of(const ColParams &p, long long t, int j) {
    const long long s = t * p.stride + (long long)j * p.dil - p.pad;
    if (s < 0) return -1;
    if (p.up == 1) return s < p.T ? s : -1;
    const long long q = s / p.up;
    return (q * p.up == s && q < p.T) ? q : -1;
}

// one thread per 4 consecutive t of one (n, ci, j) row: 16-byte stores when To % 4 == 0
__global__ __launch_bounds__(256) void im2col_kernel(ColParams p) {
    const long long row = blockIdx.y + (long long)blockIdx.z * gridDim.y;       // (n * C + ci) * k + j
    if (row >= p.rows) return;
    const long long rows_per_n = (long long)p.C * p.k;
    const long long n = row / rows_per_n;
    const int r = (int)(row - n * rows_per_n), ci = r / p.k, j = r - ci * p.k;
    const float *xr = p.x + ((size_t)n * p.C + ci) * p.T;
    float *cr = p.col + (size_t)row * p.To;
    const long long t0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (t0 >= p.To) return;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long s = t0 + i < p.To ? src_of(p, t0 + i, j) : -1;
        const float a = s >= 0 ? xr[s] : 0.f;
        v[i] = a > 0.f ? a : a * p.slope;
    }
    if ((p.To & 3) == 0) *reinterpret_cast<f32x4 *>(cr + t0) = f32x4{v[0], v[1], v[2], v[3]};
    else
        for (int i = 0; i < 4 && t0 + i < p.To; ++i) cr[t0 + i] = v[i];
}

// adjoint: gx[n][ci][s] = act'(x[n][ci][s]) * sum_j gcol[n][ci k + j][t(s, j)],  t(s, j) = (s up + pad - j dil) / stride when that divides
// and lies in [0, To) - a gather per input sample, no atomics; one thread per sample
__global__ __launch_bounds__(256) void col2im_kernel(ColParams p) {
    const long long row = blockIdx.y + (long long)blockIdx.z * gridDim.y;       // n * C + ci
    if (row >= p.rows) return;
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s >= p.T) return;
    const float *gr = p.gcol + (size_t)row * p.k * p.To;
    float acc = 0.f;
    for (int j = 0; j < p.k; ++j) {
        const long long num = s * p.up + p.pad - (long long)j * p.dil;
        if (num < 0) continue;
        long long t = num;
        if (p.stride != 1) {
            t = num / p.stride;
            if (t * p.stride != num) continue;
        }
        if (t < p.To) acc += gr[(size_t)j * p.To + t];
    }
    const float xv = p.x ? p.x[(size_t)row * p.T + s] : 1.f;
    p.gx[(size_t)row * p.T + s] = xv > 0.f ? acc : acc * p.slope;
}

int check_geom(const char *what, int64_t N, int C, int64_t T, int k, int dil, int pad, int stride, int up, int64_t To) {
    if (N < 0 || C <= 0 || T <= 0 || k <= 0 || dil <= 0 || stride <= 0 || up <= 0 || To <= 0 || pad < 0)
        PSND_FAIL(PSND_E_SHAPE, "%s: N=%lld C=%d T=%lld k=%d dil=%d pad=%d stride=%d up=%d To=%lld", what, (long long)N, C, (long long)T, k, dil, pad, stride,
                  up, (long long)To);
    if ((long long)C * k >= (1ll << 31) || T >= (1ll << 40) || To >= (1ll << 40)) PSND_FAIL(PSND_E_SHAPE, "%s: size out of range", what);
    return PSND_OK;
}

// rows over grid.y x grid.z (65535 each)
bool split_rows(long long rows, dim3 &grid) {
    const long long gy = rows < 65535 ? rows : 65535;
    const long long gz = (rows + gy - 1) / gy;
    if (gz > 65535) return false;
    grid.y = (unsigned)gy, grid.z = (unsigned)gz;
    return true;
}

}  // namespace

extern "C" int psnd_im2col_f32(const float *x, int64_t N, int C, int64_t T, int k, int dil, int pad, int stride, int up, int64_t To, float slope,
                               float *col, void *stream) {
    if (!x || !col) PSND_FAIL(PSND_E_ARG, "im2col_f32: null pointer");
    const int rc = check_geom("im2col_f32", N, C, T, k, dil, pad, stride, up, To);
    if (rc != PSND_OK) return rc;
    if (N == 0) return PSND_OK;
    ColParams p = {};
    p.x = x, p.col = col, p.T = T, p.To = To, p.C = C, p.k = k, p.dil = dil, p.pad = pad, p.stride = stride, p.up = up, p.slope = slope;
    dim3 grid((unsigned)((To + 1023) / 1024), 1, 1);
    p.rows = (long long)N * C * k;
    if (!split_rows(p.rows, grid)) PSND_FAIL(PSND_E_SHAPE, "im2col_f32: %lld rows do not fit one launch", (long long)N * C * k);
    hipLaunchKernelGGL(im2col_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    PSND_CHECK_LAUNCH("im2col_f32");
    return PSND_OK;
}

extern "C" int psnd_col2im_f32(const float *gcol, const float *x, int64_t N, int C, int64_t T, int k, int dil, int pad, int stride, int up,
                               int64_t To, float slope, float *gx, void *stream) {
    if (!gcol || !gx) PSND_FAIL(PSND_E_ARG, "col2im_f32: null pointer");
    if (!x && slope != 1.f) PSND_FAIL(PSND_E_ARG, "col2im_f32: the input is needed for the activation's derivative (slope %g)", (double)slope);
    const int rc = check_geom("col2im_f32", N, C, T, k, dil, pad, stride, up, To);
    if (rc != PSND_OK) return rc;
    if (N == 0) return PSND_OK;
    ColParams p = {};
    p.x = slope == 1.f ? nullptr : x, p.gcol = gcol, p.gx = gx, p.T = T, p.To = To, p.C = C, p.k = k, p.dil = dil, p.pad = pad, p.stride = stride, p.up = up,
    p.slope = slope;
    dim3 grid((unsigned)((T + 255) / 256), 1, 1);
    p.rows = (long long)N * C;
    if (!split_rows(p.rows, grid)) PSND_FAIL(PSND_E_SHAPE, "col2im_f32: %lld rows do not fit one launch", (long long)N * C);
    hipLaunchKernelGGL(col2im_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    PSND_CHECK_LAUNCH("col2im_f32");
    return PSND_OK;
}
