/*
 * psnd.h - C ABI of libpsnd_hip.so, the MI355X (gfx950) native hot path of pytorch_sound.
 *
 * The reference (AppleHolic/pytorch_sound, /root/reference) is pure Python: it has no FFI
 * for this path; every "kernel" there is an ATen call.  This header is therefore the binding
 * a maintainer adds (ctypes stub in INTEGRATION.md); each entry point cites the reference
 * lines whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - device pointers unless the name says _host; caller owns every buffer, guarantees
 *     contiguity and 4-byte alignment (16-byte where noted) and lifetime until the stream
 *     has completed the work.
 *   - work is ENQUEUED on `stream` (a hipStream_t passed as void*); no entry point
 *     synchronises the device, allocates device memory or keeps mutable global state.
 *   - return 0 on success, a negative PSND_E_* otherwise; psnd_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - "plans" are small read-only tables (window, twiddles, filterbank bands) built on the
 *     HOST by psnd_*_plan_build into caller memory; the caller uploads them once to the
 *     device and passes the device copy to the compute entry points.
 */
#ifndef PSND_H
#define PSND_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSND_OK 0
#define PSND_E_ARG (-1)         /* null pointer / bad enum / unsupported size            */
#define PSND_E_SHAPE (-2)       /* shapes inconsistent (e.g. T <= pad, K != n_fft/2+1)    */
#define PSND_E_HIP (-3)         /* HIP runtime reported an error at enqueue time         */
#define PSND_E_UNSUPPORTED (-4) /* valid request the library has no kernel for (yet)     */

/* framing conventions */
#define PSND_FRAMING_CENTER 0  /* reflect-pad n_fft/2       : transforms.py:55-60, :298-301   */
#define PSND_FRAMING_HIFIGAN 1 /* reflect-pad (n_fft-hop)/2 : transforms.py:352-353,
                                  interface/hifi_gan.py:37,49 */
#define PSND_FRAMING_NONE 2    /* no padding: frame f covers samples [f*hop, f*hop + n_fft)  (used by the
                                  backward of psnd_istft)                                          */

int psnd_version(void);
const char *psnd_last_error(void);
#ifdef PSND_LAB
/* LAB builds only (libpsnd_hip_lab.so, `python -m pytorch_sound_amd._build --lab`): the dispatchers read PSND_* A/B switches (kernel-instance
 * choices, ablations, trace pointers) from the environment once per call site; a process that changes its environment afterwards (parity
 * tests flipping kernel instances) calls this to have them looked up again.  The product library has no environment switch. */
void psnd_env_refresh(void);
#endif

/* ---- stream events: release points inside a captured step graph (data-parallel training) ------------------------
 * psnd_event_record_external on a CAPTURING stream adds an external event-record node (hipEventRecordExternal): after the
 * graph launch, psnd_stream_wait_event(other_stream, ev) makes `other_stream` wait for that node of that launch only - the
 * gradient all-reduce of a bucket starts while the rest of the replayed backward still runs.  Outside a capture it is a plain
 * event record.  The reference has no distributed code; north_star: "all-reduce overlapped with backward". */
void *psnd_event_create(void);                 /* NULL on failure (psnd_last_error) */
int psnd_event_destroy(void *ev);
int psnd_event_record_external(void *ev, void *stream);
int psnd_stream_wait_event(void *stream, void *ev);
int psnd_event_external_supported(void);      /* 1: this HIP runtime captures external event-record nodes (probed once) */

/* ---- integer contract: frame indexing (bit-exact) -------------------------------------- */
/* number of frames of the strided conv over the reflect-padded signal
 * (transforms.py:55-66; F = T/hop + 1 for CENTER, T/hop for HIFIGAN when hop | T). */
int64_t psnd_frame_count(int64_t T, int n_fft, int hop, int framing);
/* original-sample index read by tap m of frame f (reflect map x[-i]=x[i], x[T-1+i]=x[T-1-i]). */
int64_t psnd_frame_sample_index(int64_t f, int m, int64_t T, int n_fft, int hop, int framing);

/* ---- STFT plan ------------------------------------------------------------------------- */
/* bytes of the plan for an n_fft-point transform (0 if n_fft unsupported). */
size_t psnd_stft_plan_bytes(int n_fft);
/* window_host: n_fft taps, already zero-centre-padded from win_length (transforms.py:30-32).
 * Fills plan_host (psnd_stft_plan_bytes(n_fft) bytes). */
int psnd_stft_plan_build(int n_fft, const float *window_host, void *plan_host);

/* ---- STFT forward: replaces STFT.transform (transforms.py:53-69),
 *      STFTTorchAudio.forward/transform (transforms.py:297-311), the torch.stft + sqrt of
 *      Audio2Mel.forward (transforms.py:351-363) and interface MelSpectrogram (hifi_gan.py:46-55).
 *   wav   : (N,T) fp32
 *   mag   : (N,K,F) fp32 or NULL     = sqrt(re^2 + im^2 + mag_eps)
 *   phase : (N,K,F) fp32 or NULL     = atan2(im, re)
 *   re,im : (N,K,F) fp32 or NULL (both or neither)
 *   K = n_fft/2+1, F = psnd_frame_count(T, n_fft, hop, framing); frame axis fastest. */
int psnd_stft_fwd(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                  const void *plan, float mag_eps,
                  float *mag, float *phase, float *re, float *im, void *stream);

/* psnd_stft_fwd (magnitude only) with the BIN axis fastest: mag_nfk : (N,F,K) fp32 = the transpose of psnd_stft_fwd's mag over its last
 * two axes, same values.  transforms.py:53-69 fixes (N,K,F) at the module boundary (STFT.transform returns that); every consumer INSIDE
 * this library (psnd_mel_*, the channels-last conv stack, the spectral losses) can take either layout, and in this one a frame's spectrum
 * is one contiguous 4K-byte run: a wave stores whole cache lines that it owns alone.  Same algorithmic bytes (4NT + 4NKF).  Tuned
 * kernels for n_fft 1024 and 4096, the one-frame-per-workgroup kernel for every other supported size. */
int psnd_stft_mag_nfk(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                      const void *plan, float mag_eps, float *mag_nfk, void *stream);

/* ---- STFT backward (autograd through transforms.py:55-69 / torch.stft):
 *   gwav[n,t] = sum over (f,m) with frame_sample_index(f,m)==t of
 *               win[m] * sum_k ( gre[n,k,f] cos(2 pi k m/n) - gim[n,k,f] sin(2 pi k m/n) )
 *   where, when gmag != NULL, gre += gmag*re/mag, gim += gmag*im/mag with re/im/mag recomputed
 *   from wav (mag = sqrt(re^2+im^2+mag_eps); 0/0 -> NaN exactly as autograd of sqrt does).
 *   gmag / gre / gim: (N,K,F) or NULL (gre and gim both or neither; at least one source).
 *   gwav : (N,T) fp32, fully overwritten. */
int psnd_stft_bwd(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing,
                  const void *plan, float mag_eps,
                  const float *gmag, const float *gre, const float *gim,
                  float *gwav, void *stream);

/* ---- gradient of STFTTorchAudio.transform (transforms.py:305-311: magnitude AND a differentiable atan2 phase):
 *      (g_mag, g_phase | mag, phase) -> (g_re, g_im), all (N,K,F) fp32 = n elements; either of g_mag / g_phase may be NULL.
 *      The result is what psnd_stft_bwd takes as gre / gim. */
int psnd_polar_bwd(const float *gmag, const float *gphase, const float *mag, const float *phase, int64_t n,
                   float *gre, float *gim, void *stream);

/* ---- inverse STFT: replaces STFT.inverse (transforms.py:71-101): per frame (hop/n) * window * irDFT(mag e^{i phase})
 *      (what the pinv synthesis basis computes), overlap-add, divide by the squared-window envelope + eps,
 *      scale n/hop, trim n/2 on both sides.   mag, phase : (N,K,F);  out : (N, (F-1)*hop) fp32, fully overwritten. */
int psnd_istft(const float *mag, const float *phase, int64_t N, int64_t F, int n_fft, int hop,
               const void *plan, float eps, float *out, void *stream);

/* ---- mel projection + log + clamp: replaces transforms.py:235-243, :364-365,
 *      interface/hifi_gan.py:58-61 -------------------------------------------------------- */
#define PSND_LOG_NONE 0  /* linear mel                                   */
#define PSND_LOG_E 1     /* ln                                           */
#define PSND_LOG_10 2    /* log10                                        */
size_t psnd_mel_plan_bytes(int M, int K);
/* mel_filter_host: (M,K) fp32 row-major (the module's `mel_filter` buffer). */
int psnd_mel_plan_build(int M, int K, const float *mel_filter_host, void *plan_host);
/*   mag     : (N,K,F) fp32
 *   out     : (N,M,F) fp32 = clamp( log( max(mel, pre_clamp_min) + log_offset ), lo, hi )
 *             pre_clamp_min < 0 disables the inner max; lo/hi: pass -INF/+INF to disable.
 *   mel_lin : (N,M,F) fp32 or NULL - the un-logged product, saved for backward. */
int psnd_mel_fwd(const float *mag, int64_t N, int64_t F, int M, int K, const void *mel_plan,
                 int log_kind, float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi,
                 float *out, float *mel_lin, void *stream);
/*   gmag[n,k,f] = sum_m W[m,k] * gout[n,m,f] * dlog(mel_lin[n,m,f]) (0 where a clamp is active) */
int psnd_mel_bwd(const float *gout, const float *mel_lin, int64_t N, int64_t F, int M, int K,
                 const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min,
                 float clamp_lo, float clamp_hi, float *gmag, void *stream);

/* Fused wav -> log-mel: psnd_stft_fwd (magnitude) + psnd_mel_fwd in ONE kernel - the (N,K,F) magnitude lives only in
 * LDS.  Forward only (LogMelSpectrogram.forward transforms.py:229-244 discards the magnitude and the phase; Audio2Mel
 * :351-366; interface/hifi_gan.py:46-63).  Covers n_fft = 1024, hop <= 256, hop % 4 == 0 (PSND_E_UNSUPPORTED otherwise:
 * call the two functions above).  Arguments as in psnd_stft_fwd / psnd_mel_fwd; out : (N,M,F) fp32. */
int psnd_logmel_fwd(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing, const void *stft_plan,
                    float mag_eps, int M, const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min,
                    float clamp_lo, float clamp_hi, float *out, void *stream);

/* ---- Conv1d stacks of models/vocoders/hifi_gan.py:32-147 on channels-last bf16 ("CL") matrices -------------
 *  CL buffer: (N, Lp, Cp) bf16, row r = clip*Lp + l; rows l in [HP, HP+L) hold the clip, every other row is
 *  zero (they are the conv zero padding), channels [C, Cp) are zero padding up to a multiple of 32.
 *  psnd_conv1d_cl:  Y[r][co] = sum_j sum_ci A_eff[r + off0 + j*dstep][ci] * W[j][co][ci]   (stride 1)
 *      forward       : W = pack [j][co][ci] of the weight, off0 = -pad, dstep = +dil
 *      backward-data : A = gY,  W = pack [j][ci][co],       off0 = +pad, dstep = -dil
 *    A_eff = A + A2 * (AM > 0 ? 1 : a2_slope): the gradient of a conv output that was handed out both raw (A)
 *            and through leaky_relu (A2, with AM the activated output) is combined while it is staged.
 *            a_eff_out (N, Lp, Ca) or NULL: A_eff written back (the gradient that continues along a residual branch).
 *    epilogue: v = acc (+ bias[co]); v *= (mask_src > 0 ? 1 : mask_slope) when mask_src; v += res when res;
 *              rows outside the clip -> 0; out_raw = v, out_act = leaky_relu(v, act_slope) (either may be NULL).
 *    replaces F.leaky_relu + Conv1d + bias (+ residual add)  (hifi_gan.py:56-62, 84-88) and their backward.
 *  psnd_conv1d_cl_wgrad: gw[j][co][ci] = sum_r g[r][co] * xa[r + off0 + j*dstep][ci] as S partial slabs over row
 *              ranges, gw_part fp32 [S][k][Cb][Ca] and gbias_part [S][Cb] (may be NULL), every element written (no
 *              zero fill, no atomics); S = psnd_conv1d_cl_wgrad_splits(N, Lp, Ca, Cb, k) (host helper).
 *              g_out = g materialised (may be NULL); g = G1 + G2*leaky'(GM).
 *  psnd_conv1d_prep: weight norm w = g*v/||v|| (norm over dim 0, as torch weight_norm) -> bf16 packs
 *              wf [k][Cb][Ca] and wb [k][Ca][Cb] (for k <= 3 both in MFMA B-fragment order: tap j, 32-column tile, 16-channel
 *              step, lane (c%16/8)*32 + n%32, 8 channels - an opaque layout shared by prep and conv), zero-padded bias (Cb).  psnd_conv1d_wnorm_bwd: adds the S
 *              slabs up and runs the weight-norm backward: (g_v, g_g) and gbias (Cb, may be NULL).
 *  psnd_to_cl / psnd_from_cl: (N,C,T) fp32 <-> CL bf16 (preop 1 = log1p on the way in). */
int psnd_conv1d_cl(const void *A, const void *A2, const void *AM, float a2_slope, const void *W, const float *bias,
                   const void *res, const void *mask_src, int64_t N, int Lp, int L, int HP, int Ca, int Cb, int k,
                   int off0, int dstep, float act_slope, float mask_slope, void *out_raw, void *out_act,
                   void *a_eff_out, void *stream);
int psnd_conv1d_cl_wgrad_splits(int64_t N, int Lp, int Ca, int Cb, int k);
/* Two chained convs (3, 7 or 11 taps each) of a residual pair in one launch (csrc/psnd_conv_pair.hip; hifi_gan.py:56-62 `leaky -> conv(d) -> leaky ->
 * conv(1) -> + x` and the input-gradient chain of the same pair):
 *      mid = leaky( (conv(A; W1, taps off1 + t*dstep1) + bias1) * (M1 > 0 ? 1 : m1_slope), act1_slope )   -> mid_out (may be NULL)
 *      out = (conv(mid; W2, taps off2 + t*dstep2) + bias2) * (M2 > 0 ? 1 : m2_slope) + res               -> out_raw, out_act
 * A, M1, M2, res, mid_out, out_*: CL bf16 (N, Lp, C); W1, W2: the [k][C][C] packs of psnd_conv1d_prep (forward: the [j][co][ci] pack,
 * input gradient: the [j][ci][co] pack with mirrored taps); bias / masks / res may be NULL; rows outside the clip are written as zero.
 * Same values as two psnd_conv1d_cl launches (mid is rounded to bf16 in both).  Covered: k = 3, C = 128 or 256, both tap reaches <= 8,
 * with or without masks (forward and input-gradient form); since round 5 also, WITHOUT masks (the forward order of a pair): k = 7 / 11 at
 * C = 32 / 64 / 128 / 256 and k = 3 at C = 32 / 64, first conv reach <= 25, second <= 8 (hifi_gan.py:32-63: kernel sizes 3 / 7 / 11,
 * dilations 1 / 3 / 5, then 1).  Ask psnd_conv1d_cl_pair_supported (1 / 0) first; PSND_E_UNSUPPORTED otherwise. */
int psnd_conv1d_cl_pair_supported(int C, int k, int off1, int dstep1, int off2, int dstep2);
int psnd_conv1d_cl_pair(const void *A, const void *W1, const float *bias1, const void *M1, float m1_slope, float act1_slope,
                        void *mid_out, const void *W2, const float *bias2, const void *M2, float m2_slope, const void *res,
                        int64_t N, int Lp, int L, int HP, int C, int k, int off1, int dstep1, int off2, int dstep2,
                        float act2_slope, void *out_raw, void *out_act, void *stream);
/* The weight gradients of n (<= 32) convs over the same N x Lp rows in one launch: per conv the operands and slabs of psnd_conv1d_cl_wgrad
 * with a plain gradient (g = the conv's combined output gradient (N, Lp, Cb), xa = its activated input (N, Lp, Ca), taps off0 + j * dstep;
 * gbias_part may be NULL) and `splits` row ranges = slabs.  psnd_conv1d_cl_wgrad_multi_splits(N, Lp, Ca, Cb, k, n): the number to use for n
 * convs of that shape; convs of another shape in the same launch (a model's head / tail) take the same number. */
typedef struct psnd_wgrad_desc {
    const void *g, *xa;
    float *gw_part, *gbias_part;
    int off0, dstep;
    int Ca, Cb, k, splits;
} psnd_wgrad_desc;
int psnd_conv1d_cl_wgrad_multi_splits(int64_t N, int Lp, int Ca, int Cb, int k, int n_convs);
int psnd_conv1d_cl_wgrad_multi(const psnd_wgrad_desc *d, int n, int64_t N, int Lp, void *stream);
/* A CHAIN of such pairs in one launch (csrc/psnd_conv_chain.hip; forward only: no masks): the pairs of a ResBlock1 (hifi_gan.py:56-62),
 * pair i + 1 reading the activated output and the residual stream of pair i on the chip.  Values bit-identical to n_pairs
 * psnd_conv1d_cl_pair launches; every pair's mid_out / out_raw / out_act (each may be NULL) is written as those launches would.  A, res: the
 * activated input and the residual stream of the first pair.  psnd_conv1d_cl_chain_rows(C, k, n_pairs, taps): rows a workgroup OWNS of the 64
 * it computes (the rest is recomputed by its neighbours), 0 = unsupported (taps: off1, dstep1, off2, dstep2 per pair; k = 3, C = 256,
 * reach <= 8, 1 ... 4 pairs, >= 16 rows left).  psnd_conv_chain_stats: out2 = { chain launches, pairs they carried }.
 * With M1 / M2 on every pair the launch is the INPUT-GRADIENT chain of those pairs walked backwards (the values of psnd_conv1d_cl_pair
 * with masks, transposed packs, mirrored taps): mid = conv(A; W1) * (M1 > 0 ? 1 : m1_slope), out = conv(mid; W2) * (M2 > 0 ? 1 : m2_slope)
 * + A; res == A, activation slopes 1, out_act NULL, out_raw of pair i is the input of pair i + 1. */
typedef struct psnd_chain_pair {
    const void *W1;
    const float *bias1;
    float act1_slope;
    void *mid_out;
    const void *W2;
    const float *bias2;
    int off1, dstep1, off2, dstep2;
    float act2_slope;
    void *out_raw, *out_act;
    const void *M1, *M2;        /* the input-gradient form (below): leaky' masks of the two convs' outputs, or both NULL */
    float m1_slope, m2_slope;
} psnd_chain_pair;
int psnd_conv1d_cl_chain_rows(int C, int k, int n_pairs, const int *taps);
/* the rows a workgroup owns for a launch over R = N * Lp rows and its row tile (*mr_out: 1 = 32 computed rows, 2 = 64; may be NULL):
 * 64-row tiles (32-row tiles exist for A/B runs, PSND_CHAIN_MR=1); 0 = unsupported */
int psnd_conv1d_cl_chain_plan(int C, int k, int n_pairs, const int *taps, int64_t R, int *mr_out);
int psnd_conv1d_cl_chain(const void *A, const void *res, const psnd_chain_pair *pairs, int n_pairs, int64_t N, int Lp, int L, int HP,
                         int C, int k, void *stream);
void psnd_conv_chain_stats(long long *out2);
/* host-side launch counters of the conv kernels' tile instances: out4 = { forward / input-gradient launches with 64-row
 * workgroup tiles, with 128-row tiles, paired backward launches with 64-row tiles, with 128-row tiles } (out4 may be NULL);
 * reset != 0 clears them.  Test instrumentation: lets a parity test assert that its shape ran the instance it covers. */
/* ---- the same passes with the magnitude side BIN-FASTEST, (N, F, K) (psnd_stft_mag_nfk): (N, F, K) is the channels-last order (frame t
 *      of clip n = row n*Lp + HP + t, bin c = channel c), so the layout changes around the conv stack become plain streams, no transposes.
 *  psnd_to_cl_nfk            : x_nfk (N,F,K) fp32 -> CL bf16 (N,Lp,Cp), preop 1 = log1p; halo rows / padded channels written as zeros
 *  psnd_mask_head_l1_fwd_nfk : est_nfk = sigmoid(y) * mag_nfk; ref_nfk + part (both or neither): part[b] = sum |est - ref| of workgroup b
 *                              (psnd_mask_head_l1_blocks_nfk doubles)
 *  psnd_mask_head_l1_bwd_nfk : gy (CL) = (gest_nfk (may be NULL) + coef * g[0] * sign(est - ref)) * mag * s (1 - s)
 *  psnd_mel_fwd_nfk / psnd_mel_l1_fwd_nfk : psnd_mel_fwd / psnd_mel_l1_fwd reading mag_nfk (N,F,K); outputs stay (N,M,F)
 *  psnd_mel_bwd_nfk / psnd_mel_l1_bwd_nfk : psnd_mel_bwd / psnd_mel_l1_bwd writing gmag_nfk (N,F,K)
 *  Cp % 8 == 0, Cp >= round_up(K, 8); N*F*K*4 < 2^32 bytes. */
int psnd_to_cl_nfk(const float *x_nfk, int64_t N, int K, int64_t F, int Lp, int HP, int Cp, int preop, void *out, void *stream);
int64_t psnd_mask_head_l1_blocks_nfk(int64_t N, int64_t F, int K);
int psnd_mask_head_l1_fwd_nfk(const void *y, const float *mag_nfk, const float *ref_nfk, int64_t N, int K, int64_t F, int Lp, int HP, int Cp,
                              float *est_nfk, double *part, void *stream);
int psnd_mask_head_l1_bwd_nfk(const float *gest_nfk, const float *mag_nfk, const void *y, const float *est_nfk, const float *ref_nfk,
                              const float *g, float coef, int64_t N, int K, int64_t F, int Lp, int HP, int Cp, void *gy, void *stream);
int psnd_mel_fwd_nfk(const float *mag_nfk, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind, float log_offset,
                     float pre_clamp_min, float clamp_lo, float clamp_hi, float *out, float *mel_lin, void *stream);
int psnd_mel_bwd_nfk(const float *gout, const float *mel_lin, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind,
                     float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi, float *gmag_nfk, void *stream);
int psnd_mel_l1_fwd_nfk(const float *mag_nfk, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind, float log_offset,
                        float pre_clamp_min, float clamp_lo, float clamp_hi, const float *ref, float *mel_lin, double *part, void *stream);
int psnd_mel_l1_bwd_nfk(const float *ref, const float *mel_lin, const float *g, float coef, int64_t N, int64_t F, int M, int K,
                        const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi,
                        float *gmag_nfk, void *stream);
/* ---- a spectral-masking recipe's loss without its intermediate tensors (round 3):
 *      w1 * F.l1_loss(est, mag_ref) + w2 * F.l1_loss(log_mel(est), mel_ref),  est = sigmoid(from_cl(y)) * mag
 *  psnd_mask_head_l1_fwd : psnd_mask_head_fwd + part[b] = sum |est - ref| of workgroup b (psnd_mask_head_l1_blocks doubles)
 *  psnd_mel_l1_fwd       : psnd_mel_fwd that writes only the linear mel + part[w] = sum |log_mel - ref| of wave w (psnd_mel_l1_blocks)
 *  psnd_l1_loss_combine  : out[0] = sum_i scale[i] * sum(parts[i]), scale = weight / numel (host arrays of <= 4 device pointers);
 *                          nan_flag (may be NULL): nan_flag[0] = 1 if that loss is NaN, else 0 (the trainer's `loss != loss`, trainer.py:205)
 *  psnd_mel_l1_bwd       : gmag = W^T (coef * g[0] * sign(log_mel - ref) * dlog) with the sign formed on operand load (g: device scalar)
 *  psnd_mask_head_l1_bwd : psnd_mask_head_bwd on gest (may be NULL) + coef * g[0] * sign(est - ref)                              */
int64_t psnd_mask_head_l1_blocks(int64_t N, int64_t T, int Cp);
int psnd_mask_head_l1_fwd(const void *y, const float *mag, const float *ref, int64_t N, int C, int64_t T, int Lp, int HP, int Cp,
                          float *est, double *part, void *stream);
int psnd_mask_head_l1_bwd(const float *gest, const float *mag, const void *y, const float *est, const float *ref, const float *g,
                          float coef, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, void *gy, void *stream);
int64_t psnd_mel_l1_blocks(int64_t N, int64_t F, int M);
int psnd_mel_l1_fwd(const float *mag, int64_t N, int64_t F, int M, int K, const void *mel_plan, int log_kind, float log_offset,
                    float pre_clamp_min, float clamp_lo, float clamp_hi, const float *ref, float *mel_lin, double *part, void *stream);
int psnd_mel_l1_bwd(const float *ref, const float *mel_lin, const float *g, float coef, int64_t N, int64_t F, int M, int K,
                    const void *mel_plan, int log_kind, float log_offset, float pre_clamp_min, float clamp_lo, float clamp_hi,
                    float *gmag, void *stream);
/* flag[0] = 1.0 if any of x[0 .. n) is NaN, else 0.0 - the device-side form of the trainer's `loss != loss` (trainer.py:205) */
int psnd_nan_flag(const float *x, int64_t n, float *flag, void *stream);
int psnd_l1_loss_combine(const double *const *parts, const int64_t *nb, const double *scale, int terms, float *out, float *nan_flag, void *stream);
int psnd_conv_stats(int64_t *out4, int reset);
/* the same for the residual-pair launches: out4 = { psnd_conv1d_cl_pair launches with 32-row tiles, with 64-row tiles,
 * psnd_conv1d_cl_pair_bwd launches that carried a pair, weight-gradient row ranges of the last psnd_conv1d_cl_pair_bwd launch }. */
int psnd_conv_pair_stats(int64_t *out4, int reset);
int psnd_conv1d_cl_wgrad(const void *G1, const void *G2, const void *GM, float g2_slope, const void *xa, int64_t N,
                         int Lp, int Ca, int Cb, int k, int off0, int dstep, float *gw_part, float *gbias_part,
                         void *g_out, void *stream);
int psnd_conv1d_prep(const float *v, const float *g, const float *bias, int Cout, int Cin, int k, int Cb, int Ca,
                     void *wf, void *wb, float *bias_padded, void *stream);
/* ---- ConvTranspose1d(Cin, Cout, K = 2 * stride, stride, padding <= stride) of the HiFi-GAN upsamplers (hifi_gan.py:109,
 *  118-121) in polyphase form on the CL kernels:  y[t] = x[i] w[:, :, phi] + x[i-1] w[:, :, phi + stride],  t + padding =
 *  i * stride + phi.  Low-resolution CL buffer (N, Lp, Cip) with clip rows [HP, HP + L), HP >= 1; high-resolution CL buffer
 *  (N, LpO, Cr) with clip rows [HPO, HPO + L * stride), HPO >= padding and stride - padding rows behind the clip; rows outside
 *  the clip of every output are written as zeros or left untouched (allocate the outputs zeroed).
 *  psnd_convtr1d_prep : weight norm over dim 0 of v (Cin, Cout, K) (per INPUT channel, as torch weight_norm does for
 *                       nn.ConvTranspose1d) -> bf16 packs wf, wb (2 * stride * Cr * Cip elements each, opaque fragment order) and
 *                       bias_rep (stride * Cr floats: the bias once per phase).
 *  psnd_convtr1d_cl_fwd: out_raw = y, out_act = leaky_relu(y, act_slope) (either may be NULL).
 *  psnd_convtr1d_cl_bwd: g = g_raw + g_act * leaky'(act) (either part may be NULL) -> gx (N, Lp, Cip), g_eff (N, LpO, Cr; the
 *                       combined gradient, required when g_act is given: its column sums are the bias gradient) and the
 *                       weight-gradient slabs gw_part fp32 [S][2][Cip][stride * Cr], S = psnd_convtr1d_cl_wgrad_splits(...).
 *                       gw_part == NULL: the input-gradient launch alone; gx == NULL: the weight-gradient launch alone (g_eff as
 *                       the first call wrote it) - the two roles of one backward on two streams.
 *  psnd_convtr1d_wnorm_bwd: slabs -> g_v (Cin, Cout, K), g_g (Cin). */
int psnd_convtr1d_prep(const float *v, const float *g, const float *bias, int Cin, int Cout, int K, int stride, int Cr, int Cip,
                       void *wf, void *wb, float *bias_rep, void *stream);
/* psnd_convtr1d_prep for n transposed convs in ONE launch.  descs_dev: device array of n records
 *   { const float *v, *g, *bias; void *wf, *wb; float *bp; int Cin, Cout, K, u, Cr, Cip, blk0, pad; }   (80 bytes; blk0 = sum of the Cin of the
 *   records before it), total_blocks = the sum of all Cin.  Same packs as psnd_convtr1d_prep; the pads of wf / wb are not written (zero once). */
int psnd_convtr1d_prep_multi(const void *descs_dev, int n, int total_blocks, void *stream);
int psnd_convtr1d_cl_fwd(const void *xa, const void *wf, const float *bias_rep, int64_t N, int Lp, int L, int HP, int Cip, int Cr,
                         int stride, int padding, int LpO, int HPO, float act_slope, void *out_raw, void *out_act, void *stream);
int psnd_convtr1d_cl_wgrad_splits(int64_t N, int Lp, int Cip, int Cr, int stride);
int psnd_convtr1d_cl_bwd(const void *g_raw, const void *g_act, const void *act, float act_slope, const void *wb, const void *xa,
                         int64_t N, int Lp, int L, int HP, int Cip, int Cr, int stride, int padding, int LpO, int HPO, void *gx,
                         void *g_eff, float *gw_part, void *stream);
int psnd_convtr1d_wnorm_bwd(const float *gw_part, int splits, const float *v, const float *g, int Cin, int Cout, int K, int stride,
                            int Cr, int Cip, float *gv, float *gg, void *stream);
/* out = leaky_relu((a + b + c + d) / count, slope) over `count` (1..4) bf16 buffers of n elements (n % 8 == 0): the mean of a
 * stage's resblocks and the activation in front of the next upsampler (hifi_gan.py:122-131); backward gin = g * leaky'(y) / count. */
int psnd_cl_mean_act_fwd(const void *a, const void *b, const void *c, const void *d, int count, float slope, void *out, int64_t n,
                         void *stream);
int psnd_cl_mean_act_bwd(const void *g, const void *y, int count, float slope, void *gin, int64_t n, void *stream);
/* out_a = a[0] + .. + a[na-1], out_b = b[0] + .. + b[nb-1] over bf16 buffers of n elements (n % 8 == 0; fp32 accumulation, one rounding), both in ONE
 * launch: the gradients arriving at an upsampler's raw / activated outputs from the resblocks of its stage (hifi_gan.py:122-131).  a, b: HOST
 * arrays of na, nb (0..4) device pointers. */
int psnd_cl_sum2(const void *const *a, int na, void *out_a, const void *const *b, int nb, void *out_b, int64_t n, void *stream);
/* out[c] = sum over the rows of a channels-last bf16 matrix g (rows, C) in fp32, C % 8 == 0, 8 <= C <= 256: the bias gradient of a transposed
 * conv (reference hifi_gan.py:107-110; replaces a library column reduction).  part: fp32 scratch of psnd_cl_colsum_splits(rows, C) * C
 * elements; two launches, fixed summation order.  Lp > 0: g is rows / Lp clip buffers of Lp rows and only the rows [lo, hi) of each are
 * summed (the others are never read: they may be unwritten); Lp = 0: every row. */
int psnd_cl_colsum_splits(int64_t rows, int C);
int psnd_cl_colsum(const void *g, int64_t rows, int C, int Lp, int lo, int hi, float *part, float *out, void *stream);
/* psnd_conv1d_prep for n convs in ONE launch (one workgroup per 8 output channels: every store a 16-byte piece of the packs).
 * descs_dev: device array of n records
 *   { const float *v, *g, *bias; void *wf, *wb; float *bp; int Cout, Cin, k, Cb, Ca, blk0; }   (72 bytes, blk0 = sum of
 *   ceil(Cout / 8) of the records before it), total_blocks = that sum over all records, max_row = the largest Cin * k
 *   (<= 10240: the 8 rows of a workgroup live in LDS; larger layers take psnd_conv1d_prep).  Same packs, bit for bit, as
 *   psnd_conv1d_prep.  which: 1 = forward pack + bias, 2 = backward pack, 3 = both (the backward pack can be written when the backward
 *   starts: lines written by 16-byte stores are read fastest while they are fresh).  The pad regions of wf / wb / bp beyond the 8-channel
 *   groups are not written (zero them once). */
int psnd_conv1d_prep_multi(const void *descs_dev, int n, int total_blocks, int max_row, int which, void *stream);
/* psnd_conv1d_wnorm_bwd for up to PSND_WNORM_MAX convs (the six of a ResBlock1, the 26 of the separator body) in one launch; descs is a HOST array,
 * passed to the kernel by value (nothing is uploaded, the launch can be captured in a hipGraph). */
#define PSND_WNORM_MAX 32
typedef struct psnd_wnorm_desc {
    const float *gw_part, *gbias_part;      /* slabs of psnd_conv1d_cl_wgrad / psnd_conv1d_cl_bwd (gbias_part may be NULL) */
    const float *v, *g;                     /* weight_v (Cout,Cin,k), weight_g (Cout)                                      */
    float *gv, *gg, *gbias;                 /* outputs (gbias may be NULL)                                                 */
    int splits, Cout, Cin, k, Cb, Ca;
} psnd_wnorm_desc;
int psnd_conv1d_wnorm_bwd_multi(const psnd_wnorm_desc *descs, int n, void *stream);
/* backward of one conv in ONE launch: gx = input gradient (Ca channels, via the transposed pack wb, taps mirrored) and the
 * partial weight-gradient slabs gw_part / gbias_part (as psnd_conv1d_cl_wgrad), both from g = G1 + G2 * leaky'(GM); g_out (may
 * be NULL) receives the combined g for the residual branch (needs G2).  gx_mask / gx_res (CL bf16, Ca channels, may be NULL):
 * epilogue of the input gradient, gx = gx * leaky'(gx_mask; gx_mask_slope) + gx_res - i.e. the NEXT conv's combined incoming gradient
 * (its own leaky-relu derivative and the residual branch) is formed here, so that conv's backward needs no combine on load.
 * gw_part == NULL: the input gradient alone (the weight gradient is left to psnd_conv1d_cl_wgrad_multi). */
int psnd_conv1d_cl_bwd(const void *G1, const void *G2, const void *GM, float g2_slope, const void *wb, const void *xa,
                       int64_t N, int Lp, int L, int HP, int Ca, int Cb, int k, int pad, int dil, void *gx, void *g_out,
                       const void *gx_mask, float gx_mask_slope, const void *gx_res, float *gw_part, float *gbias_part,
                       void *stream);
/* Backward of a whole residual pair (`leaky -> conv1(d) -> leaky -> conv2(1) -> + x`, hifi_gan.py:56-62) as ONE launch: the input gradients
 * of both convs chained on chip (psnd_conv1d_cl_pair's body with the transposed packs wb2 / wb1, mirrored taps, the leaky' masks M1 = the
 * activated conv1 output and M2 = the activated pair input, res = G the gradient on the residual stream: g_mid = gradient wrt conv1's
 * output, gx = gradient wrt the pair's input), the weight-gradient slabs of conv2 (gradient G, input xa_a = M1; gw_a / gb_a) and the slabs
 * of ANOTHER conv whose gradient an earlier launch produced (G_b, xa_b with taps off_b + t * dstep_b; gw_b / gb_b; all NULL: none) -
 * in a chain walked backwards that is conv1 of the pair handled before.  G == NULL: only the `_b` weight gradient (the chain's last
 * conv1).  Slabs: psnd_conv1d_cl_pair_bwd_splits(N, Lp, C, k) per conv, layout as psnd_conv1d_cl_wgrad.  C = 256, k = 3 and
 * psnd_conv1d_cl_pair_bwd_supported(...) == 1, PSND_E_UNSUPPORTED otherwise. */
int psnd_conv1d_cl_pair_bwd_supported(int C, int k, int pad2, int dil2, int pad1, int dil1);
int psnd_conv1d_cl_pair_bwd_splits(int64_t N, int Lp, int C, int k);
int psnd_conv1d_cl_pair_bwd(const void *G, const void *wb2, const void *M1, float m1_slope, void *g_mid, const void *wb1, const void *M2,
                            float m2_slope, const void *res, int64_t N, int Lp, int L, int HP, int C, int k, int pad2, int dil2,
                            int pad1, int dil1, void *gx, const void *xa_a, float *gw_a, float *gb_a, const void *G_b,
                            const void *xa_b, int off_b, int dstep_b, float *gw_b, float *gb_b, void *stream);
int psnd_conv1d_wnorm_bwd(const float *gw_part, const float *gbias_part, int splits, const float *v, const float *g,
                          int Cout, int Cin, int k, int Cb, int Ca, float *gv, float *gg, float *gbias, void *stream);
int psnd_to_cl(const float *x, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, int preop, void *out, void *stream);
/* out = tanh(from_cl(x)) - the generator's output non-linearity (vocoders/hifi_gan.py:134-135) inside the layout change - and its backward
 * gx = to_cl(g * (1 - out_fwd^2)) (halo rows / padded channels written as zeros). */
int psnd_from_cl_tanh(const void *x, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, float *out, void *stream);
int psnd_to_cl_tanh_bwd(const float *g, const float *out_fwd, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, void *gx, void *stream);
int psnd_from_cl(const void *x, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, float *out, void *stream);
/* mask head of a spectrogram-masking model: est (N,C,T) fp32 = sigmoid(from_cl(y)) * mag in one pass; backward
 * gy (CL bf16) = to_cl(gest * mag * s (1 - s)), s = sigmoid(y) recomputed (halo rows / padded channels written as zeros). */
int psnd_mask_head_fwd(const void *y, const float *mag, int64_t N, int C, int64_t T, int Lp, int HP, int Cp, float *est,
                       void *stream);
int psnd_mask_head_bwd(const float *gest, const float *mag, const void *y, int64_t N, int C, int64_t T, int Lp, int HP, int Cp,
                       void *gy, void *stream);

/* ---- dense contractions of models/modules.py on the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32) --------------------
 * psnd_linear1x1_fwd: y[n][co][t] = sum_ci w[co][ci] x[n][ci][t] + bias[co] (, relu) - the 1x1 Conv1d projections
 *   (modules.py:21-22 linear_kvq / linear, :93-95 the feed-forward pair).  x (N,Cin,T), w (Cout,Cin), y (N,Cout,T) fp32.
 *   bf16 != 0: both operands are rounded to bf16 while they are staged (v_mfma_f32_32x32x16_bf16, fp32 accumulate, 16x the matrix
 *   rate) - what the modules select under torch.autocast(bfloat16); 0: exact fp32 products.
 * psnd_linear1x1_bwd: gy' = gy where ymask > 0 (ymask = the relu output y, or NULL); gx = w^T gy' (NULL: skipped),
 *   gw = sum_{n,t} gy' x^T via S = psnd_linear1x1_wgrad_slabs(...) partial slabs gw_part (S*Cout*Cin floats), gbias = sum gy'.
 * psnd_mha_fwd: MultiHeadAttention.scale_dot_att over all heads (modules.py:38-48, 61-79).  kvq (N, 3C, T): rows [0,C) keys,
 *   [C,2C) values, [2C,3C) queries (the reference's chunk order), head h = channels [h*d, (h+1)*d), d = C/H <= 64 (PSND_E_UNSUPPORTED
 *   otherwise), batch index b = h*N + n.  mask (N,T) bytes, 1 = padding, or NULL: padded keys get no
 *   weight, padded queries are zeroed.  out (N, C, T) (heads unfolded); att (H*N, T_key, T_query) or NULL - the scores never
 *   leave the chip otherwise; stats (H*N, T, 2): per query column (max, 1/sum), kept for the backward.
 *   bf16 != 0: the operands of the four score / accumulate products (K, Q, V and the probabilities) are rounded to bf16 on their
 *   way to the matrix cores; scores, softmax statistics and accumulation stay fp32 (selected under torch.autocast(bfloat16)).
 *   bf16 == 2 (round 6; att == NULL only): as 1, and kvq is read and out WRITTEN as bf16 tensors (2-byte elements, same shapes - what
 *   psnd_linear1x1_fwd_ex writes with io_h = 2 and takes with io_h = 1): the values multiplied are the same, the kernels convert
 *   nothing in their loops.
 * psnd_mha_bwd: gkvq (N, 3C, T) from gout (N, C, T) and, optionally, gatt (needs att).  delta (H*N, T) scratch.  bf16 as above (the
 *   probabilities are recomputed from the saved statistics: pass the forward's choice); bf16 == 2 (att, gatt NULL): kvq, out and gout
 *   are read and gkvq WRITTEN as bf16 (psnd_linear1x1_bwd_ex writes gout with io_h = 2 and takes gkvq with io_h = 1); stats, delta fp32
 *   (delta is then formed by the query-gradient kernel from the fragments it holds - no separate pass over out and gout; psnd_mha_bwd_parts
 *   takes bf16 = 2 with parts = 7 only). */
int psnd_linear1x1_fwd(const float *x, const float *w, const float *bias, int64_t N, int Cin, int Cout, int64_t T, int relu, int bf16,
                       float *y, void *stream);
int64_t psnd_linear1x1_wgrad_slabs(int64_t N, int Cin, int Cout, int64_t T);
int psnd_linear1x1_bwd(const float *gy, const float *ymask, const float *x, const float *w, int64_t N, int Cin, int Cout, int64_t T,
                       int bf16, float *gx, float *gw, float *gw_part, float *gbias, void *stream);
/* psnd_linear1x1_bwd with gx = w^T gy' + gx_addend (N, Cin, T; NULL: none): the gradient that reaches x along a residual connection
 * (`x + self.drop_out(...)` before the GroupNorm, modules.py:56-58, 114-116) is added in the GEMM's epilogue - autograd's separate
 * accumulation pass over both tensors disappears. */
int psnd_linear1x1_bwd_acc(const float *gy, const float *ymask, const float *x, const float *w, int64_t N, int Cin, int Cout, int64_t T,
                           int bf16, const float *gx_addend, float *gx, float *gw, float *gw_part, float *gbias, void *stream);
/* ... or with gx = (gx_mask > 0) ? w^T gy' : 0, gx_mask (N, Cin, T; NULL: none; not together with gx_addend): x is the output of a ReLU
 * (Conv1d -> ReLU -> Conv1d, modules.py:93-95; gx_mask = x) and gx is wanted for the ReLU's input.  The layer before the ReLU then calls
 * with ymask = NULL: its two GEMMs and its bias sum read the gradient alone instead of gradient + mask.
 * io_h (with bf16 != 0; 0: every tensor is fp32): 1 = gy is STORED as bf16 (2-byte elements, same shape; ymask NULL), 2 = x, gx_mask and gx
 * are (gx_addend NULL) - the hidden tensor of that pair under torch.autocast(bfloat16): the products round their operands to bf16 when
 * they load them, so the values multiplied are the same and the tensor and its gradient move at half the bytes.  The weight, bias and
 * parameter gradients are always fp32. */
int psnd_linear1x1_bwd_ex(const void *gy, const float *ymask, const void *x, const float *w, int64_t N, int Cin, int Cout, int64_t T, int bf16,
                          int io_h, int64_t ld_h, const float *gx_addend, const void *gx_mask, void *gx, float *gw, float *gw_part, float *gbias,
                          void *stream);
/* psnd_linear1x1_fwd with io_h (bf16 != 0): 1 = x is stored as bf16, 2 = y is (see psnd_linear1x1_bwd_ex); 0 = psnd_linear1x1_fwd.
 * ld_h (both entry points): row pitch in elements of the bf16-stored tensors - (N, C, ld_h) in memory, the first T frames of a row used;
 * 0 = T.  They are the caller's own tensors between two calls: rows that start on 128-byte lines (ld_h a multiple of 64) spare the
 * GEMMs the straddling loads and stores of odd row lengths. */
int psnd_linear1x1_fwd_ex(const void *x, const float *w, const float *bias, int64_t N, int Cin, int Cout, int64_t T, int relu, int bf16, int io_h,
                          int64_t ld_h, void *y, void *stream);
int psnd_mha_fwd(const float *kvq, const unsigned char *mask, int64_t N, int H, int C, int64_t T, float *out, float *att, float *stats,
                 int bf16, void *stream);
int psnd_mha_bwd(const float *kvq, const unsigned char *mask, const float *out, const float *att, const float *stats, const float *gout,
                 const float *gatt, int64_t N, int H, int C, int64_t T, float *delta, float *gkvq, int bf16, void *stream);
/* psnd_mha_bwd in parts (bit mask): 1 = delta, 2 = key / value gradients, 4 = query gradients (7 = psnd_mha_bwd).  The two gradient
 * kernels need `delta` only and write disjoint rows of gkvq: a caller may enqueue them on two streams behind the delta launch. */
int psnd_mha_bwd_parts(const float *kvq, const unsigned char *mask, const float *out, const float *att, const float *stats, const float *gout,
                       const float *gatt, int64_t N, int H, int C, int64_t T, float *delta, float *gkvq, int bf16, int parts, void *stream);

/* ---- transformer blocks of models/modules.py: the parts that are not plain GEMMs --------------------------
 *  psnd_groupnorm1_fwd: y = GroupNorm(1, C)(x + res) [relu]  (modules.py:30,58 / :98,114-116): mean / variance over
 *      (C x T) per sample (accumulated in double), per-channel affine.  x, res (may be NULL), y : (N,C,T) fp32;
 *      stats : (N,2) {mean, rstd} saved for backward; ws : (N,C,2) double scratch (caller owned, overwritten;
 *      PSND_GN_WS_DOUBLES(C) doubles per sample): one pair of sums per row, written and then added up in a fixed order - no
 *      atomics, the statistics are the same bits from run to run (round 6; rounds 3-5 took 32 doubles per sample).
 *  psnd_groupnorm1_bwd: gx (gradient wrt x and wrt res), ggamma, gbeta (C) - all fully overwritten; ws as above.
 *  psnd_softmax_keys_fwd: in place on scores (B,Tk,Tq): a = softmax over Tk of scale*s with key-padded rows at -inf,
 *      query-padded columns set to 0 (modules.py:66-76); mask (B,T) uint8, 1 = padded, or NULL.
 *  psnd_softmax_keys_bwd: gscores = scale * a * (gatt - sum_tk gatt*a). */
#define PSND_GN_WS_DOUBLES(C) (2 * (size_t)(C))
/* PositionalEncoding.forward (modules.py:119-145): y = x * scale + pe[:, :T] in one pass; x, y (N,C,T) fp32, pe (C, pe_len) rows (the module's
 * buffer (1, C, max_len)), pe_len >= T.  pe == NULL: y = x * scale (the backward: gx = g * scale). */
int psnd_posenc(const float *x, const float *pe, float scale, int64_t N, int C, int64_t T, int64_t pe_len, float *y, void *stream);
int psnd_groupnorm1_fwd(const float *x, const float *res, const float *gamma, const float *beta, int64_t N, int C,
                        int64_t T, float eps, int relu, float *y, float *stats, double *ws, void *stream);
int psnd_groupnorm1_bwd(const float *gy, const float *x, const float *res, const float *gamma, const float *y,
                        const float *stats, int64_t N, int C, int64_t T, int relu, float *gx, float *ggamma,
                        float *gbeta, double *ws, void *stream);
int psnd_softmax_keys_fwd(float *scores, const uint8_t *mask, int64_t B, int64_t T, float scale, void *stream);
int psnd_softmax_keys_bwd(const float *att, const float *gatt, int64_t B, int64_t T, float scale, float *gscores,
                          void *stream);

/* ---- models/sound.py: PreEmphasis (:66-81) and the reductions of multi_stft_loss (:106-133) -----------------
 *  psnd_preemphasis_fwd: y[n][t] = x[n][t] - coef * x[n][t-1] with x[n][-1] := x[n][1] (one reflect-padded sample,
 *      conv1d with the flipped filter [-coef, 1]);  x, y : (N, T) fp32 (the module's (N,1,T) is the same memory).
 *  psnd_preemphasis_bwd: the adjoint, gx fully overwritten.
 *  multi_stft_loss on magnitudes p, t : (N, K, F) fp32 per resolution (KF = K*F elements per clip):
 *  psnd_stft_loss_blocks(KF): workgroups per clip B of the partial pass (host helper).
 *  psnd_stft_loss_partial: part[(n*B + b)*3 + {0,1,2}] = sum (t-p)^2, sum t^2, sum |log(t+eps) - log(p+eps)| over
 *      chunk b of clip n (double; every entry written, no atomics).
 *  psnd_stft_loss_final: combines L <= 8 resolutions (host arrays parts[L] of device pointers, KF[L]) into
 *      out3 = {loss, sc_loss, mag_loss} (sound.py:119-133) and norms[(i*N + n)*2 + {0,1}] = ||t-p||_F, ||t||_F.
 *  psnd_stft_loss_bwd: one resolution; norms = that resolution's N pairs; g3 = DEVICE pointer to the upstream gradient
 *      of (loss, sc_loss, mag_loss); gp / gt (either may be NULL) = gradient wrt p / t, fully overwritten. */
int psnd_preemphasis_fwd(const float *x, int64_t N, int64_t T, float coef, float *y, void *stream);
int psnd_preemphasis_bwd(const float *gy, int64_t N, int64_t T, float coef, float *gx, void *stream);
int64_t psnd_stft_loss_blocks(int64_t KF);
int psnd_stft_loss_partial(const float *p_mag, const float *t_mag, int64_t N, int64_t KF, float eps, double *part,
                           void *stream);
int psnd_stft_loss_final(const double *const *parts, const int64_t *KF, int L, int64_t N, float *norms, float *out3,
                         void *stream);
int psnd_stft_loss_bwd(const float *p_mag, const float *t_mag, int64_t N, int64_t KF, float eps, const float *norms,
                       const float *g3, int L, float *gp, float *gt, void *stream);
/*  psnd_stft_bwd_msl: psnd_stft_loss_bwd (gp only) FUSED into psnd_stft_bwd for one resolution - the gradient of multi_stft_loss
 *      w.r.t. the predicted waveform in one launch: the adjoint recomputes |X| of `wav` (the prediction), forms
 *      d loss / d |X| = k1 (|X| - t) + c_mag sign(|X| - t) / (|X| + eps) from it and the target magnitudes t_mag (N, K, F) in
 *      registers (the magnitude gradient never exists in HBM) and continues as psnd_stft_bwd.  norms / g3 / L / eps as
 *      psnd_stft_loss_bwd; gwav (N, T) fully overwritten, or - accumulate != 0 - added to (the sum over the resolutions
 *      without separate add launches).  psnd_stft_bwd_msl_supported(n_fft, hop): 1 where the span-staged
 *      adjoint runs (n_fft 512 / 1024 / 2048, even hop up to n/2 resp. 256), else callers use the two-launch path. */
/*  psnd_stft_fwd_msl: psnd_stft_fwd (magnitude, centre framing) + psnd_stft_loss_partial FUSED for the prediction of one resolution:
 *      |X| is compared with t_mag (N, K, F) in registers and only the three sums leave the kernel - part[(n*B + b)*3 + {0,1,2}],
 *      B = psnd_stft_fwd_msl_blocks(T, n_fft, hop) entries per clip (0: this (n_fft, hop) is not covered - span-staged kernel,
 *      n_fft 512 / 1024 / 2048, one tile per workgroup).  psnd_stft_loss_final_blocks: psnd_stft_loss_final with the entries per
 *      clip given per resolution (blocks[L], host array; NULL = psnd_stft_loss_blocks(KF[i])). */
int64_t psnd_stft_fwd_msl_blocks(int64_t T, int n_fft, int hop);
int psnd_stft_fwd_msl(const float *wav, int64_t N, int64_t T, int n_fft, int hop, const void *plan, float mag_eps,
                      const float *t_mag, float eps, double *part, void *stream);
int psnd_stft_loss_final_blocks(const double *const *parts, const int64_t *KF, const int64_t *blocks, int L, int64_t N,
                                float *norms, float *out3, void *stream);
int psnd_stft_bwd_msl_supported(int n_fft, int hop);
int psnd_stft_bwd_msl(const float *wav, int64_t N, int64_t T, int n_fft, int hop, int framing, const void *plan,
                      float mag_eps, const float *t_mag, const float *norms, const float *g3, int L, float eps,
                      int accumulate, float *gwav, void *stream);

/* ---- fp32 instance of the Conv1d / ConvTranspose1d stack (models/vocoders/hifi_gan.py:32-147 computes its convolutions in fp32) -------
 *  A convolution = psnd_linear1x1_fwd (exact fp32 matrix-core GEMM) over the unfolded input
 *      col[n][ci * k + j][t] = act(x[n][ci][src]),  s' = t * stride + j * dil - pad,  src = s' / up  (0 where s' < 0, s' % up != 0 or src >= T),
 *  act(v) = v > 0 ? v : slope * v (slope 1: none) - the leaky-relu in front of every ResBlock conv (hifi_gan.py:58-61, 86-87).
 *  Conv1d(k, dilation d, padding p): stride 1, up 1, pad p, To = T + 2 p - d (k - 1);  ConvTranspose1d(k, stride u, padding p): up u,
 *  pad k - 1 - p, taps flipped by the caller, To = (T - 1) u - 2 p + k (hifi_gan.py:107-110).  x (N, C, T), col (N, C * k, To) fp32.
 *  psnd_col2im_f32 is the adjoint: gx (N, C, T) from gcol, times act'(x) (x may be NULL when slope == 1). */
int psnd_im2col_f32(const float *x, int64_t N, int C, int64_t T, int k, int dil, int pad, int stride, int up, int64_t To, float slope,
                    float *col, void *stream);
int psnd_col2im_f32(const float *gcol, const float *x, int64_t N, int C, int64_t T, int k, int dil, int pad, int stride, int up, int64_t To,
                    float slope, float *gx, void *stream);

/* ---- data/dataset.py:196-250, SpeechDataLoader.pad_collate_fn on the audio column, device side --------------------
 *  flat : the batch's clips back to back (device, fp32); offs[n], lens[n] : start / length of clip n in flat (device
 *  int64).  out : (N, Tmax) fp32 = clip n zero-padded (or cut) to Tmax;  mask (may be NULL) : (N, Tmax) fp32, 1 on valid
 *  samples and 0 on padding - the reference's np.ones_like(item) after zero padding (dataset.py:70-71, 88-89). */
int psnd_pad_collate(const float *flat, const int64_t *offs, const int64_t *lens, int64_t N, int64_t Tmax, float *out,
                     float *mask, void *stream);

/* ---- SpectrogramMasker.forward (models/transforms.py:409-416): wave-level mask (N, T) fp32 -> frame-level mask (N, F) fp32,
 *  F = psnd_frame_mask_frames(T, win, hop) = (T + 2 (win / 2) - win) / hop + 1: out[n][f] = ceil(mean over frame f of the mask padded with
 *  win / 2 ones in front and win / 2 zeros behind) - the reference's constant-weight Conv1d + ceil without a library convolution. */
int64_t psnd_frame_mask_frames(int64_t T, int win, int hop);
int psnd_frame_mask(const float *mask, int64_t N, int64_t T, int win, int hop, float *out, void *stream);

/* bf16 communication image of a flat fp32 gradient bucket (pytorch_sound_amd/distributed.py, FlatGradReducer(comm_dtype=bfloat16); the reference
 * has no distributed code): dst[i] = bf16(src[i] * scale), round to nearest even / dst[i] = float(src[i]) * scale.  n % 8 == 0. */
int psnd_grad_pack_bf16(const float *src, void *dst, int64_t n, float scale, void *stream);
int psnd_grad_unpack_bf16(const void *src, float *dst, int64_t n, float scale, void *stream);
/* ---- the optimizer step of Trainer.train (trainer.py:215-216) for Adam / AdamW: one launch over all tensors ---------
 *  table : n_tensors records {float *p; const float *g; float *m; float *v; float *step; int64 numel} (device,
 *      psnd_adam_table_bytes() each); the work list: workgroup b updates elements [chunk_off[b], chunk_off[b] +
 *      psnd_adam_chunk()) of tensor chunk_tensor[b] (device arrays, built once per parameter set by the caller).
 *  torch.optim.Adam semantics (amsgrad / maximize off): step += 1; m, v moments; p -= lr / (1 - b1^step) * m /
 *      (sqrt(v) / sqrt(1 - b2^step) + eps); weight_decay as L2 (decoupled = 0) or AdamW (decoupled = 1).
 *      Hyper-parameters are doubles: 1 - beta and the bias corrections are formed in double, as torch does.
 *  found_inf (device, may be NULL): non-zero skips the whole update including the step count (AMP protocol);
 *  grad_scale (device, may be NULL): gradients are divided by it;  corr : 2 * n_tensors floats of device scratch (the
 *      bias corrections, formed in double by a first tiny launch that also advances the step counts).
 *  Trainer.clip_grad (trainer.py:184-191) folded into the same pass: clip_value > 0 clamps every (scaled) gradient element to
 *      [-clip_value, clip_value] (`p.grad.clamp`), clip_coef (device, may be NULL) then multiplies it (`clip_grad_norm_`'s factor,
 *      from psnd_grad_sumsq).  The gradients in memory are left as they are.
 *  psnd_grad_sumsq: the global norm over the SAME table / work list: *sumsq (+)= sum clamp(g / grad_scale, +-clip_value)^2 in double
 *      (partial: n_chunks doubles of device scratch, summed in index order: deterministic), coef[0] = min(1, max_norm / (norm + 1e-6))
 *      (1 when max_norm = 0), coef[1] = norm; accumulate != 0 adds to *sumsq (second and later parameter groups). */
int64_t psnd_adam_chunk(void);
int64_t psnd_adam_table_bytes(void);
int psnd_adam_step(const void *table, int n_tensors, const int *chunk_tensor, const int64_t *chunk_off, int64_t n_chunks,
                   double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled,
                   const float *found_inf, const float *grad_scale, float *corr, float clip_value, const float *clip_coef,
                   void *stream);
/* psnd_adam_step + a host-visible record of the step's skip flag: flag_log = a pinned, device-mapped int ring of {flag, seq} pairs (or NULL);
 * the launch writes flag_log[2 slot] = (found_inf != 0) and then, behind a system-scope fence, flag_log[2 slot + 1] = log_seq.  The host
 * polls the sequence number: no device-to-host copy, no event (Trainer's NaN log line, trainer.py:205-207). */
int psnd_adam_step_logged(const void *table, int n_tensors, const int *chunk_tensor, const int64_t *chunk_off, int64_t n_chunks,
                          double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled,
                          const float *found_inf, const float *grad_scale, float *corr, float clip_value, const float *clip_coef,
                          int *flag_log, int log_slot, int log_seq, void *stream);
int psnd_grad_sumsq(const void *table, int n_tensors, const int *chunk_tensor, const int64_t *chunk_off, int64_t n_chunks,
                    float clip_value, const float *grad_scale, int accumulate, float max_norm, double *partial, double *sumsq,
                    float *coef, void *stream);

/* ---- PQMF (models/transforms.py:492-560), polyphase --------------------------------------------------------------
 *  filt : (subbands, taps + 1) fp32 (analysis_filter / synthesis_filter of the module), taps even, P = taps / 2.
 *  psnd_pqmf_analysis : x (B, T) -> out (B, subbands, T / subbands):  out[b][k][m] = scale * sum_j F[k][j] x[b][m S + j - P]
 *      = F.conv1d(pad(x), analysis_filter) followed by the stride-S pick (transforms.py:541-542), zeros outside the signal.
 *  psnd_pqmf_synthesis: x (B, subbands, M) -> y (B, T_out), T_out = M * subbands for the module's synthesis (the adjoint of an
 *      analysis over T samples asks for T_out = T: the tail samples past M * subbands still reach the last frames):
 *      y[b][t] = scale * sum_k sum_{j: (t + j - P) % S == 0} F[k][j] x[b][k][(t + j - P) / S]
 *      = zero-stuffing conv_transpose1d (x S) + F.conv1d(pad(.), synthesis_filter) (transforms.py:552-553) with scale = S.
 *  flip = 1 reverses the taps: each op is then the adjoint of the other (the backward passes). */
int psnd_pqmf_analysis(const float *x, const float *filt, int64_t B, int64_t T, int subbands, int taps, int flip, float scale,
                       float *out, void *stream);
int psnd_pqmf_synthesis(const float *x, const float *filt, int64_t B, int64_t M, int64_t T_out, int subbands, int taps, int flip,
                        float scale, float *y, void *stream);

/* ---- F.l1_loss (reduction 'mean') as used by the training recipes' spectral losses --------------------------------
 *  psnd_l1_loss_fwd: out[0] = mean |a - b| over n fp32 elements; part: psnd_l1_loss_blocks(n) doubles of scratch (one partial
 *      per 16384-element chunk, summed in a fixed order by a second tiny launch: bit-reproducible).
 *  psnd_l1_loss_bwd: ga = g[0] * sign(a - b) / n, gb = -ga (either NULL); g = device pointer to the upstream gradient. */
int64_t psnd_l1_loss_blocks(int64_t n);
/* masked L1 of padded batches (data/dataset.py:196-250 pads the clips of a batch to its longest): a, b (N,C,T) fp32, frame_weight (N,T)
 * fp32 (1 = the frame holds signal): out[0] = sum |a - b| w / (C sum w); part: 2 * psnd_masked_l1_blocks(N*C*T) doubles of scratch;
 * inv_den[0] = 1 / (C sum w) is kept for the backward: ga = g[0] * sign(a - b) * w * inv_den, gb = -ga (either NULL). */
int64_t psnd_masked_l1_blocks(int64_t n);
int psnd_masked_l1_fwd(const float *a, const float *b, const float *frame_weight, int64_t N, int C, int64_t T, double *part, float *out,
                       float *inv_den, void *stream);
int psnd_masked_l1_bwd(const float *a, const float *b, const float *frame_weight, int64_t N, int C, int64_t T, const float *g,
                       const float *inv_den, float *ga, float *gb, void *stream);
int psnd_l1_loss_fwd(const float *a, const float *b, int64_t n, double *part, float *out, void *stream);
int psnd_l1_loss_bwd(const float *a, const float *b, int64_t n, const float *g, float *ga, float *gb, void *stream);
/* a weighted sum of up to 4 mean-L1 terms as ONE scalar (a recipe's `l1(a, b) + 0.5 * l1(c, d)`): out[0] = sum_i w[i] * mean|a[i] - b[i]|;
 * a, b, n, w: HOST arrays of `terms` entries (read during the call); part: sum_i psnd_l1_loss_blocks(n[i]) doubles of scratch.
 * psnd_l1_loss_bwd_w: the backward of one term, ga = weight * g[0] * sign(a - b) / n. */
int psnd_l1_loss_sum_fwd(const float *const *a, const float *const *b, const int64_t *n, const double *w, int terms, double *part,
                         float *out, void *stream);
int psnd_l1_loss_bwd_w(const float *a, const float *b, int64_t n, const float *g, double weight, float *ga, float *gb, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PSND_H */
