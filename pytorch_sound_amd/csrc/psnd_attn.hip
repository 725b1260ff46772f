// psnd_attn.hip - the dense contractions of pytorch_sound/models/modules.py on the gfx950 matrix cores, exact fp32:
//   * the 1x1 Conv1d projections (modules.py:21-22 linear_kvq / linear, :93-95 the feed-forward pair) as ONE strided GEMM kernel
//     (forward, input gradient, weight gradient are the same kernel with other strides) - psnd_linear1x1_*
//   * MultiHeadAttention.scale_dot_att (modules.py:61-79): k^T q / sqrt(d) -> key mask -> softmax over KEYS -> query mask ->
//     v att, forward and backward, without the (H*N, T, T) score tensor ever leaving the chip unless the caller wants `att` -
//     psnd_mha_fwd / psnd_mha_bwd
// All products run on v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate - bitwise an fmaf chain (MI355X guide: "exact f32 at the
// vector rate"), so the parity bounds of the fp32 torch formulation (tests/golden/modules.npz, 1e-4) hold unchanged.
//
// MFMA operand convention (32x32x2, wave64): lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// D[i][j]: j = l & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
// The attention kernels lean on one property of that map: the accumulator register s of lane half h holds row
// rho(s, h) = (s & 3) + 8 * (s >> 2) + 4 * h - so a probability tile P[tk][tq] in accumulator registers IS the B operand of
// the next product over tk (k-step s <-> row rho(s, h)) without any data movement; only the A operand (V, K, Q or gO from
// LDS) has to be read in the same permuted order.
#include "psnd_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ int rho(int s, int half) { return (s & 3) + 8 * (s >> 2) + 4 * half; }

// =====================================================================================================================
// strided batched GEMM   C_z[m][n] = sum_k A_z[m][k] * B_z[k][n]  (+ bias[m]) (relu),  n contiguous in C
// =====================================================================================================================
struct GemmParams {
    const float *A, *B;
    float *C;
    const float *bias;           // [M] or null
    const float *amask, *bmask;  // null, or same indexing as A / B: the operand element counts only where mask > 0 (relu backward)
    int M, N, K, Z;
    long long sAm, sAk, sAz, sBk, sBn, sBz, sCm, sCz;
    int relu;
    int zchunk;                  // > 0: weight-gradient mode - a workgroup sums over zchunk batches into its slab (tile z = chunk * ksplit + part)
    long long sCslab;
    int ksplit, kpart;           // weight-gradient mode: the K axis (frames of a clip) in ksplit parts of kpart (a multiple of 32) each - small weight
                                 // matrices over few clips otherwise give fewer workgroups than CUs (256 x 256 over 32 clips: 128)
    int flatT;                   // > 0 (n-contiguous B only): the columns of all Z batches form ONE axis of Z * flatT columns,
                                 // column j = (z = j / flatT, t = j % flatT) - no partly filled column tile per batch (T = 173: 68 % -> 98 %)
    // Tile order (round 4).  The launch is ONE-dimensional; workgroup L is tile gemm_tile(L).  Tiles that read the same operand bytes - the
    // M tiles over one column tile of B (projections: B is the 42-169 MB activation tensor, read once PER M TILE before: 8 x at 256 -> 1024),
    // all the tiles of one batch chunk in weight-gradient mode - are numbered consecutively AND land on one XCD (workgroup L runs on XCD
    // L % 8): the first of them brings the bytes into that XCD's L2, the others hit.  tiles_x / tiles_y / tiles_z: the logical grid.
    int tiles_x, tiles_y, tiles_z;
    const float *addend;         // null, or a tensor indexed like C: C = A B + addend (the residual branch's gradient folded into the
                                 // input-gradient GEMM of the branch's first projection: no separate accumulation pass)
    int ablate;                  // lab builds only (PSND_GEMM_ABLATE): 1 = no operand fetches behind the first two k-tiles, 2 = no MFMAs, 4 = no epilogue
    int a_h, b_h, c_h;           // (bf16 kernel only) the operand / the output is STORED as bf16 (2-byte elements, same indexing): the hidden
                                 // tensor of the feed-forward pair and its gradient under autocast - the GEMMs round their operands to bf16 when
                                 // they load them, so the stored values are the ones multiplied either way, at half the bytes (round 6).
                                 // c_h: C and omask are bf16 (bias / addend stay fp32)
    const float *omask;          // null, or a tensor indexed like C: C = omask > 0 ? A B : 0 - the input of this projection is the output of a
                                 // ReLU (the feed-forward pair, modules.py:93-95): the ReLU's backward rides in the epilogue of the GEMM that
                                 // PRODUCES its gradient, and the two GEMMs behind the ReLU read that gradient without a mask (round 6)
};

// logical tile (x, y, z) of workgroup L; false = padding workgroup (the numbering is padded to whole groups of 8 outer tiles)
__device__ __forceinline__ bool gemm_tile(const GemmParams &p, int &bx, int &by, int &bz) {
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    if (p.zchunk > 0) {                          // weight gradient: inner = the tiles of one batch chunk, outer = the chunks
        const int inner = p.tiles_x * p.tiles_y, o = (j / inner) * 8 + xcd, i = j % inner;
        if (o >= p.tiles_z) return false;
        bx = i % p.tiles_x, by = i / p.tiles_x, bz = o;
    } else {                                     // inner = the M tiles over one (column tile, batch), outer = those
        const int inner = p.tiles_y, o = (j / inner) * 8 + xcd, i = j % inner;
        if (o >= p.tiles_x * p.tiles_z) return false;
        bx = o % p.tiles_x, bz = o / p.tiles_x, by = i;
    }
    return true;
}

constexpr int GBM = 128, GBN = 128, GBK = 16, GP = 132;   // LDS pitch of both tiles ([k][m] and [k][n], m / n fastest)

// where the (up to) four consecutive columns / rows of a thread live when the 128-direction is contiguous in memory
struct MinorSpan {
    long long off;       // element offset of the first one (batch offset included in flat mode)
    int nval;            // how many of the four exist
    int wrap_at;         // flat mode: elements e >= wrap_at belong to the next batch ...
    long long wrap_add;  // ... wrap_add elements further on
};
__device__ __forceinline__ MinorSpan minor_span(int x, int lim, int flatT, long long sz) {
    MinorSpan m;
    m.nval = min(max(lim - x, 0), 4);
    if (flatT > 0) {
        const int z = x / flatT, t = x - z * flatT;
        m.off = z * sz + t, m.wrap_at = flatT - t, m.wrap_add = sz - flatT;
    } else {
        m.off = x, m.wrap_at = 4, m.wrap_add = 0;
    }
    return m;
}

// global -> registers: the thread's two float4 of a (128 x 16) tile.  MINOR_CONTIG: the 128-direction (m or n) is contiguous in memory.
// 16-byte loads need only 4-byte alignment on gfx9 global memory (f32x4_u), so odd row pitches (T = 173) keep the vector path.
template <bool MINOR_CONTIG>
__device__ __forceinline__ void gemm_fetch(const float *base, long long s_major128, long long s_k, int lim128, int limk, int o128, int ok, int tid,
                                           const MinorSpan &ms, f32x4_t (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        f32x4_t r = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MINOR_CONTIG) {            // float4 along the 128-direction: thread -> (k = tid / 32 + 8 u, x4 = tid % 32)
            const int k = ok + (tid >> 5) + 8 * u;
            if (k < limk) {
                const float *p = base + (long long)k * s_k + ms.off;
                if (ms.nval == 4 && ms.wrap_at >= 4) {
                    r = *reinterpret_cast<const f32x4_u *>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < ms.nval) r[e] = p[e + (e >= ms.wrap_at ? ms.wrap_add : 0)];
                }
            }
        } else {                                 // float4 along k: thread -> (x = tid / 4 + 64 u, k4 = tid % 4)
            const int x = o128 + (tid >> 2) + 64 * u, k = ok + 4 * (tid & 3);
            if (x < lim128) {
                const float *p = base + (long long)x * s_major128 + k;
                if (k + 3 < limk) {
                    r = *reinterpret_cast<const f32x4_u *>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < limk) r[e] = p[e];
                }
            }
        }
        v[u] = r;
    }
}
__device__ __forceinline__ void gemm_apply_mask(f32x4_t (&v)[2], const f32x4_t (&m)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = m[u][e] > 0.f ? v[u][e] : 0.f;
}
template <bool MINOR_CONTIG>
__device__ __forceinline__ void gemm_commit(float *tile, int tid, const f32x4_t (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if constexpr (MINOR_CONTIG) {
            *reinterpret_cast<f32x4_t *>(tile + ((tid >> 5) + 8 * u) * GP + 4 * (tid & 31)) = v[u];
        } else {
            const int x = (tid >> 2) + 64 * u, k = 4 * (tid & 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(k + e) * GP + x] = v[u][e];
        }
    }
}

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __bf16 hwbf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
    const f32x2_v v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hwbf16x2_t));
}
// epilogue shared by both GEMM kernels: bias, relu, store (flat mode: column -> (batch, t))
__device__ __forceinline__ void gemm_store(const GemmParams &p, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int li, int half, int z0, int bz) {
    float *C = p.C + (p.zchunk > 0 ? bz * p.sCslab : (p.flatT > 0 ? 0 : z0 * p.sCz));
    const int ncols = p.flatT > 0 ? p.Z * p.flatT : p.N;
    // the thread's 32 bias values BEFORE its first store (round 6): loaded next to the stores each one was waited for behind the stores in
    // front of it - C and bias may alias as far as hipcc knows, and gfx9 counts loads and stores in one in-order counter - a store latency
    // per block of sixteen (the wide-output projections were bound by this epilogue, not by their k-loop: PSND_GEMM_ABLATE, notebook)
    float bv[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + t * 32 + rho(r, half);
            bv[t][r] = p.bias ? p.bias[m < p.M ? m : p.M - 1] : 0.f;
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int n = n0 + wn * 64 + u * 32 + li;
        if (n >= ncols) continue;
        long long coff = n;
        if (p.flatT > 0) {
            const int z = n / p.flatT;
            coff = z * p.sCz + (n - z * p.flatT);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float ad[16];
            const float *side = p.addend ? p.addend : p.omask;      // never both (psnd_linear1x1_bwd_ex refuses)
            if (side) {                          // all sixteen loads of the block in flight before its first store (C and addend may alias
                                                 // as far as hipcc knows: next to the stores every load would be waited for in turn)
                const float *ap = side + (C - p.C) + coff;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + t * 32 + rho(r, half);
                    ad[r] = ap[(long long)(m < p.M ? m : p.M - 1) * p.sCm];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + t * 32 + rho(r, half);
                if (m < p.M) {
                    float v = acc[t][u][r] + bv[t][r];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    if (p.addend) v += ad[r];
                    else if (p.omask) v = ad[r] > 0.f ? v : 0.f;
                    C[(long long)m * p.sCm + coff] = v;
                }
            }
        }
    }
}

// Pipeline of both kernels: the operands of k-tiles it + 1 and it + 2 are in flight (raw fp32 in registers, two stages) while
// k-tile it is multiplied out of LDS (two buffers); masks travel with their operand and are applied when it is committed.
template <bool A_MCONTIG, bool B_NCONTIG>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float sA[2][GBK * GP], sB[2][GBK * GP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kk = lane >> 5;
    int tbx, tby, tbz;
    if (!gemm_tile(p, tbx, tby, tbz)) return;
    const int m0 = tby * GBM, n0 = tbx * GBN;
    const int zc = p.zchunk > 0 ? tbz / p.ksplit : tbz, kbeg = p.zchunk > 0 ? (tbz - zc * p.ksplit) * p.kpart : 0;
    const int kend = p.zchunk > 0 ? min(p.K, kbeg + p.kpart) : p.K;
    const int z0 = p.zchunk > 0 ? zc * p.zchunk : tbz;
    const int z1 = p.zchunk > 0 ? min(z0 + p.zchunk, p.Z) : z0 + 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int nk = kend > kbeg ? (kend - kbeg + GBK - 1) / GBK : 0;
    const int steps = (z1 - z0) * nk;
    const MinorSpan msA = minor_span(m0 + 4 * (tid & 31), p.M, 0, 0);
    const MinorSpan msB = minor_span(n0 + 4 * (tid & 31), p.flatT > 0 ? p.Z * p.flatT : p.N, p.flatT, p.sBz);
    const bool ma = p.amask != nullptr, mb = p.bmask != nullptr;
    f32x4_t va[2][2], vb[2][2], vam[2][2], vbm[2][2];
    auto fetch = [&](int it, auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        const int zi = it / nk, k0 = kbeg + (it - zi * nk) * GBK;
        const long long za = (long long)(z0 + zi) * p.sAz, zb = p.flatT > 0 ? 0 : (long long)(z0 + zi) * p.sBz;
        gemm_fetch<A_MCONTIG>(p.A + za, p.sAm, p.sAk, p.M, kend, m0, k0, tid, msA, va[S]);
        gemm_fetch<B_NCONTIG>(p.B + zb, p.sBn, p.sBk, p.N, kend, n0, k0, tid, msB, vb[S]);
        if (ma) gemm_fetch<A_MCONTIG>(p.amask + za, p.sAm, p.sAk, p.M, kend, m0, k0, tid, msA, vam[S]);
        if (mb) gemm_fetch<B_NCONTIG>(p.bmask + zb, p.sBn, p.sBk, p.N, kend, n0, k0, tid, msB, vbm[S]);
    };
    auto body = [&](int it, auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        float *tA = sA[S], *tB = sB[S];
        if (ma) gemm_apply_mask(va[S], vam[S]);
        if (mb) gemm_apply_mask(vb[S], vbm[S]);
        gemm_commit<A_MCONTIG>(tA, tid, va[S]);
        gemm_commit<B_NCONTIG>(tB, tid, vb[S]);
        __syncthreads();
        if (it + 2 < steps) fetch(it + 2, sc);
        // fragments of k-step s + 1 are read while the four MFMAs of step s run (the compiler left every step waiting for its own reads)
        float fa[2][2], fb[2][2];
        auto frag = [&](int s, float (&a)[2], float (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = tA[(2 * s + kk) * GP + wm * 64 + t * 32 + li];
                b[t] = tB[(2 * s + kk) * GP + wn * 64 + t * 32 + li];
            }
        };
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < GBK / 2; ++s) {
            if (s + 1 < GBK / 2) frag(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][t], fb[s & 1][u], acc[t][u], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // next step's two ds_read2 first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // ... under this step's four MFMAs (left alone hipcc sinks the reads
            __builtin_amdgcn_sched_barrier(0);                      // back next to their use to save two registers)
        }
        // buffer S is written again two k-tiles from now: the barrier of the next k-tile orders these reads before that
    };
    if (steps > 0) fetch(0, std::integral_constant<int, 0>{});
    if (steps > 1) fetch(1, std::integral_constant<int, 1>{});
    for (int it = 0; it < steps; it += 2) {
        body(it, std::integral_constant<int, 0>{});
        if (it + 1 < steps) body(it + 1, std::integral_constant<int, 1>{});
    }
    gemm_store(p, acc, m0, n0, wm, wn, li, kk, z0, tbz);
}

// ---------------------------------------------------------------------------------------------------------------------
// the same strided GEMM with bf16 OPERANDS (fp32 in HBM, rounded to bf16 when they are committed to LDS; fp32 accumulate, fp32 out)
// on v_mfma_f32_32x32x16_bf16 - 16x the matrix rate of the exact-fp32 form.  Chosen by the caller (torch.autocast(bfloat16) around
// the modules); tolerance = bf16 rounding of both operands (tests: exact against float64 arithmetic on the rounded operands).
// LDS holds both tiles as [row (m or n)][k] bf16 with an 80-byte pitch (odd multiple of 16 B: conflict-free ds_read_b128 fragments):
//   k-contiguous source  : a thread converts 8 consecutive k of one row (2 x 16-byte loads) -> one ds_write_b128
//   row-contiguous source: a thread gathers 8 consecutive k of ONE row with 8 dword loads (lanes along the rows: coalesced) -> one
//                          ds_write_b128 - no transposing scatter into LDS
// With 8 MFMAs of 32 cycles per 32-deep k-tile the kernel is bound by the 32 KB of fp32 operands it stages per tile, not by MFMA time.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HBK = 32, HP = 40;      // k-tile depth, LDS row pitch (bf16)

// the thread's share of a (128 rows x 32 k) tile: two runs of 8 consecutive k of one row, raw fp32.  ROW_CONTIG: the 128-direction is
// contiguous in memory; `rowoff` is then the element offset of the thread's row (batch offset included in flat mode), < 0 = no such row.
template <bool ROW_CONTIG>
__device__ __forceinline__ void hgemm_fetch(const float *base, long long s_row, long long s_k, int limrow, int limk, int orow, int ok, int tid,
                                            long long rowoff, float (&f)[2][8]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[u][e] = 0.f;
        if constexpr (!ROW_CONTIG) {             // k contiguous: row = tid / 4 + 64 u, k8 = 8 (tid % 4)
            const int row = orow + (tid >> 2) + 64 * u, k = ok + 8 * (tid & 3);
            if (row < limrow) {
                const float *p = base + (long long)row * s_row + k;
                if (k + 7 < limk) {
                    const f32x4_t a = reinterpret_cast<const f32x4_u *>(p)[0], b = reinterpret_cast<const f32x4_u *>(p)[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[u][e] = a[e], f[u][4 + e] = b[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (k + e < limk) f[u][e] = p[e];
                }
            }
        } else {                                 // rows contiguous: row = tid % 128, k8 = 8 (tid / 128 + 2 u)
            const int k = ok + 8 * ((tid >> 7) + 2 * u);
            if (rowoff >= 0) {
                const float *p = base + (long long)k * s_k + rowoff;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k + e < limk) f[u][e] = p[(long long)e * s_k];
            }
        }
    }
}
template <bool ROW_CONTIG>
__device__ __forceinline__ void hgemm_commit(unsigned short *tile, int tid, const float (&f)[2][8], const float (&m)[2][8], bool masked) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (masked && !(m[u][e] > 0.f)) ? 0.f : f[u][e];
        const int row = ROW_CONTIG ? (tid & 127) : (tid >> 2) + 64 * u;
        const int k = ROW_CONTIG ? 8 * ((tid >> 7) + 2 * u) : 8 * (tid & 3);
        *reinterpret_cast<uint4 *>(tile + row * HP + k) =
            make_uint4(pack2_bf16(g[0], g[1]), pack2_bf16(g[2], g[3]), pack2_bf16(g[4], g[5]), pack2_bf16(g[6], g[7]));
    }
}

// Epilogue for a bf16 OUTPUT (GemmParams c_h; no slabs, no addend): the workgroup's (128 x 128) tile goes through LDS - the accumulator
// layout has a lane's neighbours along n in OTHER lanes, so direct stores are 2-byte scatters (64-byte runs) - and leaves as 16-byte
// stores, 256 bytes of one output row per 16 lanes; the ReLU mask of psnd_linear1x1_bwd_ex (bf16, indexed like C) is read the same way.
// `sT`: the kernel's operand tiles (free behind a barrier), 128 rows of CPH bf16.
constexpr int CPH = 136;             // 272-byte rows: 16-byte aligned, the rows of one store on different banks
__device__ __forceinline__ void gemm_store_h(const GemmParams &p, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int li, int half, int z0,
                                             int tid, unsigned short *sT) {
    __syncthreads();                 // every wave is done with the operand tiles
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 64 + t * 32 + rho(r, half), m = m0 + ml;
            const float bv = (p.bias && m < p.M) ? p.bias[m] : 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v = acc[t][u][r] + bv;
                if (p.relu) v = v > 0.f ? v : 0.f;
                sT[ml * CPH + wn * 64 + u * 32 + li] = (unsigned short)(pack2_bf16(v, 0.f) & 0xffffu);
            }
        }
    __syncthreads();
    unsigned short *C = reinterpret_cast<unsigned short *>(p.C) + (p.flatT > 0 ? 0 : z0 * p.sCz);
    const unsigned short *M = p.omask ? reinterpret_cast<const unsigned short *>(p.omask) + (p.flatT > 0 ? 0 : z0 * p.sCz) : nullptr;
    const int ncols = p.flatT > 0 ? p.Z * p.flatT : p.N;
    typedef unsigned u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 4, col = (c & 15) * 8;
        const int m = m0 + row, n = n0 + col;
        if (m >= p.M || n >= ncols) continue;
        const uint4 q = *reinterpret_cast<const uint4 *>(sT + row * CPH + col);
        unsigned w[4] = {q.x, q.y, q.z, q.w};
        long long coff = n;
        bool whole = n + 7 < ncols;
        if (p.flatT > 0) {
            const int z = n / p.flatT, tt = n - z * p.flatT;
            coff = z * p.sCz + tt;
            whole = whole && tt + 7 < p.flatT;
        }
        const long long off = (long long)m * p.sCm + coff;
        if (whole) {
            if (M) {
                const u32x4_a2 mk = *reinterpret_cast<const u32x4_a2 *>(M + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) {         // a bf16 is > 0 when its sign bit is clear and it is not zero (the mask is a ReLU's output)
                    const unsigned lo = mk[j] & 0xffffu, hi = mk[j] >> 16;
                    const unsigned keep = ((lo - 1u) < 0x7fffu ? 0xffffu : 0u) | ((hi - 1u) < 0x7fffu ? 0xffff0000u : 0u);
                    w[j] &= keep;
                }
            }
            u32x4_a2 o = {w[0], w[1], w[2], w[3]};
            *reinterpret_cast<u32x4_a2 *>(C + off) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ne = n + e;
                if (ne >= ncols) break;
                long long ce = ne;
                if (p.flatT > 0) {
                    const int z = ne / p.flatT;
                    ce = z * p.sCz + (ne - z * p.flatT);
                }
                const long long oe = (long long)m * p.sCm + ce;
                unsigned short v = (unsigned short)((w[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                if (M && !((unsigned)(M[oe] - 1u) < 0x7fffu)) v = 0;
                C[oe] = v;
            }
        }
    }
}

// ---- an operand STORED as bf16 (GemmParams a_h / b_h): half the bytes, no conversion.  The thread's share of the (128 rows x 32 k) tile is
// eight dwords:
//   k-contiguous source  : 8 consecutive k of rows tid / 4 and tid / 4 + 64 - one 16-byte load each, written to LDS as they are
//   row-contiguous source: rows 2 (tid % 64) and + 1 at the 8 k of group tid / 64 - one dword load per k (the two rows are neighbours in
//                          memory; a pair that straddles two batches of a flat column axis, or ends the matrix, takes two 2-byte loads),
//                          the halves sorted into the two rows' 16-byte LDS writes by v_perm_b32
// `rowoff` (row-contiguous): element offsets of the thread's two rows, < 0 = no such row.
__device__ __forceinline__ unsigned ldg_u16(const unsigned short *p) { return *p; }
template <bool ROW_CONTIG>
__device__ __forceinline__ void hgemm_fetch_h(const unsigned short *base, long long s_row, long long s_k, int limrow, int limk, int orow, int ok, int tid,
                                              const long long (&rowoff)[2], unsigned (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0u;
    if constexpr (!ROW_CONTIG) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = orow + (tid >> 2) + 64 * u, k = ok + 8 * (tid & 3);
            if (row < limrow) {
                const unsigned short *p = base + (long long)row * s_row + k;
                if (k + 7 < limk) {
                    typedef unsigned u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
                    const u32x4_u a = *reinterpret_cast<const u32x4_u *>(p);
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[4 * u + e] = a[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (k + e < limk) f[4 * u + (e >> 1)] |= ldg_u16(p + e) << (16 * (e & 1));
                }
            }
        }
    } else {
        const int k = ok + 8 * (tid >> 6);
        const bool pair = rowoff[0] >= 0 && rowoff[1] == rowoff[0] + 1;
        typedef unsigned u32_u2 __attribute__((aligned(2)));
        if (k + 7 < limk && __builtin_amdgcn_ballot_w64(!pair) == 0) {        // (wave-uniform) the whole wave on whole pairs: eight plain loads
            const unsigned short *p = base + (long long)k * s_k + rowoff[0];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = *reinterpret_cast<const u32_u2 *>(p + (long long)e * s_k);
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (k + e >= limk) continue;
            const unsigned short *p = base + (long long)(k + e) * s_k;
            if (pair) {
                typedef unsigned u32_u __attribute__((aligned(2)));
                f[e] = *reinterpret_cast<const u32_u *>(p + rowoff[0]);
            } else {
                if (rowoff[0] >= 0) f[e] = ldg_u16(p + rowoff[0]);
                if (rowoff[1] >= 0) f[e] |= ldg_u16(p + rowoff[1]) << 16;
            }
        }
    }
}
template <bool ROW_CONTIG>
__device__ __forceinline__ void hgemm_commit_h(unsigned short *tile, int tid, const unsigned (&f)[8]) {
    if constexpr (!ROW_CONTIG) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            *reinterpret_cast<uint4 *>(tile + ((tid >> 2) + 64 * u) * HP + 8 * (tid & 3)) = make_uint4(f[4 * u], f[4 * u + 1], f[4 * u + 2], f[4 * u + 3]);
    } else {
        const int row = 2 * (tid & 63), k = 8 * (tid >> 6);
        unsigned lo[4], hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = __builtin_amdgcn_perm(f[2 * j + 1], f[2 * j], 0x05040100u);      // the low halves of dwords 2 j, 2 j + 1: row `row`
            hi[j] = __builtin_amdgcn_perm(f[2 * j + 1], f[2 * j], 0x07060302u);      // the high halves: row + 1
        }
        *reinterpret_cast<uint4 *>(tile + row * HP + k) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4 *>(tile + (row + 1) * HP + k) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    }
}

// MASKED: a relu mask travels with an operand (input / weight gradient behind a ReLU).  Without one the two mask stages (64 registers) do not
// exist: 3 workgroups per CU instead of 2 - the kernel is a latency chain per workgroup (8 k-steps behind barriers, operands two steps
// ahead), so residency is what hides it, and 646 workgroups (256 -> 256 at 32 x 1292) fit the chip in ONE round instead of 1.26.
// DT: 1 = A is stored as bf16, 2 = B is, 4 = C (and omask) are (never together with MASKED).
template <bool A_MCONTIG, bool B_NCONTIG, bool MASKED, int DT = 0>
__global__ __launch_bounds__(256, MASKED ? 2 : 3) void gemm_bf16_kernel(GemmParams p) {
    static_assert(!(MASKED && DT != 0), "operand masks come with fp32 storage only");
    constexpr bool A_H = (DT & 1) != 0, B_H = (DT & 2) != 0, C_H = (DT & 4) != 0;
    __shared__ __attribute__((aligned(16))) unsigned short sAB[2 * GBM * HP + 2 * GBN * HP];      // the operand tiles; a bf16 output tile afterwards
    static_assert(128 * CPH <= 2 * GBM * HP + 2 * GBN * HP, "the bf16 output tile fits the operand tiles");
    unsigned short(*sA)[GBM * HP] = reinterpret_cast<unsigned short(*)[GBM * HP]>(sAB);
    unsigned short(*sB)[GBN * HP] = reinterpret_cast<unsigned short(*)[GBN * HP]>(sAB + 2 * GBM * HP);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kg = lane >> 5;
    int tbx, tby, tbz;
    if (!gemm_tile(p, tbx, tby, tbz)) return;
    const int m0 = tby * GBM, n0 = tbx * GBN;
    const int zc = p.zchunk > 0 ? tbz / p.ksplit : tbz, kbeg = p.zchunk > 0 ? (tbz - zc * p.ksplit) * p.kpart : 0;
    const int kend = p.zchunk > 0 ? min(p.K, kbeg + p.kpart) : p.K;
    const int z0 = p.zchunk > 0 ? zc * p.zchunk : tbz;
    const int z1 = p.zchunk > 0 ? min(z0 + p.zchunk, p.Z) : z0 + 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int nk = kend > kbeg ? (kend - kbeg + HBK - 1) / HBK : 0;
    const int steps = (z1 - z0) * nk;
    long long rowA = -1, rowB = -1;
    {
        const int ra = m0 + (tid & 127), rb = n0 + (tid & 127);
        if (ra < p.M) rowA = ra;
        if (p.flatT > 0) {
            if (rb < p.Z * p.flatT) {
                const int z = rb / p.flatT;
                rowB = z * p.sBz + (rb - z * p.flatT);
            }
        } else if (rb < p.N) {
            rowB = rb;
        }
    }
    long long rowAh[2] = {-1, -1}, rowBh[2] = {-1, -1};        // bf16-stored, row-contiguous operands: the thread's two neighbouring rows
    if constexpr (A_H && A_MCONTIG) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (m0 + 2 * (tid & 63) + q < p.M) rowAh[q] = m0 + 2 * (tid & 63) + q;
    }
    if constexpr (B_H && B_NCONTIG) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rb = n0 + 2 * (tid & 63) + q;
            if (p.flatT > 0) {
                if (rb < p.Z * p.flatT) {
                    const int z = rb / p.flatT;
                    rowBh[q] = z * p.sBz + (rb - z * p.flatT);
                }
            } else if (rb < p.N) {
                rowBh[q] = rb;
            }
        }
    }
    const bool ma = MASKED && p.amask != nullptr, mb = MASKED && p.bmask != nullptr;
    float fa[A_H ? 1 : 2][2][8], fb[B_H ? 1 : 2][2][8], fam[MASKED ? 2 : 1][2][8], fbm[MASKED ? 2 : 1][2][8];
    unsigned ha[A_H ? 2 : 1][8], hb[B_H ? 2 : 1][8];
    auto fetch = [&](int it, auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        const int zi = it / nk, k0 = kbeg + (it - zi * nk) * HBK;
        const long long za = (long long)(z0 + zi) * p.sAz, zb = p.flatT > 0 ? 0 : (long long)(z0 + zi) * p.sBz;
        if constexpr (A_H) hgemm_fetch_h<A_MCONTIG>(reinterpret_cast<const unsigned short *>(p.A) + za, p.sAm, p.sAk, p.M, kend, m0, k0, tid, rowAh, ha[S]);
        else hgemm_fetch<A_MCONTIG>(p.A + za, p.sAm, p.sAk, p.M, kend, m0, k0, tid, rowA, fa[S]);
        if constexpr (B_H) hgemm_fetch_h<B_NCONTIG>(reinterpret_cast<const unsigned short *>(p.B) + zb, p.sBn, p.sBk, p.N, kend, n0, k0, tid, rowBh, hb[S]);
        else hgemm_fetch<B_NCONTIG>(p.B + zb, p.sBn, p.sBk, p.N, kend, n0, k0, tid, rowB, fb[S]);
        if constexpr (MASKED) {
            if (ma) hgemm_fetch<A_MCONTIG>(p.amask + za, p.sAm, p.sAk, p.M, kend, m0, k0, tid, rowA, fam[S]);
            if (mb) hgemm_fetch<B_NCONTIG>(p.bmask + zb, p.sBn, p.sBk, p.N, kend, n0, k0, tid, rowB, fbm[S]);
        }
    };
    auto body = [&](int it, auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        unsigned short *tA = sA[S], *tB = sB[S];
        if constexpr (A_H) hgemm_commit_h<A_MCONTIG>(tA, tid, ha[S]);
        else hgemm_commit<A_MCONTIG>(tA, tid, fa[S], fam[MASKED ? S : 0], ma);
        if constexpr (B_H) hgemm_commit_h<B_NCONTIG>(tB, tid, hb[S]);
        else hgemm_commit<B_NCONTIG>(tB, tid, fb[S], fbm[MASKED ? S : 0], mb);
        __syncthreads();
        if (it + 2 < steps && !PSND_ABL(p, 1)) fetch(it + 2, sc);
        if (PSND_ABL(p, 2)) return;
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
            bf16x8_t a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const bf16x8_t *>(tA + (wm * 64 + t * 32 + li) * HP + 16 * s + 8 * kg);
                b[t] = *reinterpret_cast<const bf16x8_t *>(tB + (wn * 64 + t * 32 + li) * HP + 16 * s + 8 * kg);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
        }
    };
    if (steps > 0) fetch(0, std::integral_constant<int, 0>{});
    if (steps > 1) fetch(1, std::integral_constant<int, 1>{});
    for (int it = 0; it < steps; it += 2) {
        body(it, std::integral_constant<int, 0>{});
        if (it + 1 < steps) body(it + 1, std::integral_constant<int, 1>{});
    }
    if (PSND_ABL(p, 4)) return;
    if constexpr (C_H) gemm_store_h(p, acc, m0, n0, wm, wn, li, kg, z0, tid, sAB);
    else gemm_store(p, acc, m0, n0, wm, wn, li, kg, z0, tbz);
}

// out[i] = sum over slabs of part[s][i].  64 elements x 4 slab groups per workgroup, four loads in flight per thread: one thread walking
// 128-160 slabs one dependent load at a time took 29 us per launch for a 256 x 256 matrix.  Fixed summation order (deterministic).
__global__ __launch_bounds__(256) void slab_sum_kernel(const float *part, int slabs, long long n, float *out) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int s = g;
        for (; s + 12 < slabs; s += 16) {
            a0 += part[(long long)s * n + i];
            a1 += part[(long long)(s + 4) * n + i];
            a2 += part[(long long)(s + 8) * n + i];
            a3 += part[(long long)(s + 12) * n + i];
        }
        for (; s < slabs; s += 4) a0 += part[(long long)s * n + i];
    }
    red[g][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < n) out[i] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// gbias[c] = sum over (z, t) of g[z][c][t] (where mask[z][c][t] > 0 when given): one workgroup per channel.  1024 threads = 4 clip
// groups x 256 frame lanes, four clips per thread in flight and no branch around a load: the first version (256 threads, one load per
// thread and iteration behind the mask test, clips walked one after the other) ran 76 us per launch at 32 x 256 x 1292 (0.55 TB/s).
template <bool MASK>
__global__ __launch_bounds__(1024) void rowsum_kernel(const float *g, const float *mask, int Z, int C, long long T, float *out) {
    const int c = blockIdx.x, tid = threadIdx.x, zg = tid >> 8, tt = tid & 255;
    double acc = 0.0;
    for (int z0 = zg; z0 < Z; z0 += 16)
        for (long long t = tt; t < T; t += 256) {
            float v[4], m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = min(z0 + 4 * u, Z - 1);
                const long long o = ((long long)z * C + c) * T + t;
                v[u] = g[o];
                m[u] = MASK ? mask[o] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (z0 + 4 * u < Z && m[u] > 0.f) ? (double)v[u] : 0.0;
        }
    __shared__ double red[1024];
    red[tid] = acc;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[c] = (float)red[0];
}

// The same sum for rows of whole 16-byte pieces (T % 4 == 0; round 6): a workgroup per channel walks the (clip, 16-byte piece) items of its
// Z rows flat - no partly filled pass over a row (T = 1292: the sixth 256-frame pass of the kernel above had 12 live lanes of 256) -
// with four 16-byte loads per operand in flight per thread; fp32 partial sums of at most a few hundred terms per thread, the tree in
// double.  338 MB (32 x 1024 x 1292, masked) in 228 -> ~90 us.
template <bool MASK>
__global__ __launch_bounds__(512) void rowsum4_kernel(const float *g, const float *mask, int Z, int C, int Q /* T / 4 */, float *out) {
    const int c = blockIdx.x, tid = threadIdx.x;
    const long long items = (long long)Z * Q;
    const long long rowq = (long long)C * Q;                   // 16-byte pieces between the same channel of two clips
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g) + (long long)c * Q;
    const f32x4 *m4 = reinterpret_cast<const f32x4 *>(mask) + (long long)c * Q;
    float acc = 0.f;
    for (long long i0 = tid; i0 < items; i0 += 4 * 512) {
        f32x4 v[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            long long i = i0 + 512 * u;
            i = i < items ? i : items - 1;
            const long long z = i / Q, o = z * rowq + (i - z * Q);
            v[u] = g4[o];
            if (MASK) m[u] = m4[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + 512 * u >= items) continue;
            if (MASK) acc += ((m[u].x > 0.f ? v[u].x : 0.f) + (m[u].y > 0.f ? v[u].y : 0.f)) + ((m[u].z > 0.f ? v[u].z : 0.f) + (m[u].w > 0.f ? v[u].w : 0.f));   // same grouping as below
            else acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        }
    }
    __shared__ double red[512];
    red[tid] = (double)acc;
    __syncthreads();
    for (int s = 256; s >= 1; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[c] = (float)red[0];
}

// ... over a tensor STORED as bf16 (the gradient of the feed-forward pair's hidden tensor): four elements per 8-byte load when T % 4 == 0,
// fp32 partial sums per thread, the tree in double (as rowsum4_kernel)
__global__ __launch_bounds__(512) void rowsum_h_kernel(const unsigned short *g, int Z, int C, long long T, long long LD, float *out) {
    const int c = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    if ((T & 3) == 0 && (LD & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 7) == 0) {
        const long long Q = T / 4, items = (long long)Z * Q, rowq = (long long)C * (LD / 4);
        const uint2 *g4 = reinterpret_cast<const uint2 *>(g) + (long long)c * (LD / 4);
        for (long long i0 = tid; i0 < items; i0 += 4 * 512) {
            uint2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                long long i = i0 + 512 * u;
                i = i < items ? i : items - 1;
                const long long z = i / Q;
                v[u] = g4[z * rowq + (i - z * Q)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + 512 * u >= items) continue;
                acc += (__builtin_bit_cast(float, v[u].x << 16) + __builtin_bit_cast(float, v[u].x & 0xffff0000u)) +
                       (__builtin_bit_cast(float, v[u].y << 16) + __builtin_bit_cast(float, v[u].y & 0xffff0000u));
            }
        }
    } else {
        for (int z = 0; z < Z; ++z)
            for (long long t = tid; t < T; t += 512) acc += __builtin_bit_cast(float, (unsigned)g[((long long)z * C + c) * LD + t] << 16);
    }
    __shared__ double red[512];
    red[tid] = (double)acc;
    __syncthreads();
    for (int s = 256; s >= 1; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[c] = (float)red[0];
}

// =====================================================================================================================
// attention.  kvq: (N, 3C, T) - rows [0, C) keys, [C, 2C) values, [2C, 3C) queries, head h = channels [h d, (h+1) d), d = 64.
// batch index b = h * N + n (heads folded head-major, modules.py:38).  mask: (N, T) bytes, 1 = padding, or null.
// =====================================================================================================================
// HDP: head dimension padded to 32 or 64 (template parameter); rows dd >= d of every tile / fragment read as zero
constexpr int TP = 33;          // LDS pitch of the (HDP x 32) tiles: rows on distinct banks for the permuted-order A reads
struct AttnParams {
    const float *kvq;           // (N, 3C, T)
    const unsigned char *mask;  // (N, T) or null
    float *out;                 // fwd: (N, C, T)
    float *att;                 // fwd: (H*N, T, T) or null
    float *stats;               // (H*N, T, 2): (max, 1 / sum) of every query column
    const float *gout;          // bwd: (N, C, T)
    const float *gatt;          // bwd: (H*N, T, T) or null
    const float *delta;         // bwd: (H*N, T)
    float *gkvq;                // bwd: (N, 3C, T)
    int N, H, C, T, d;          // d = C / H, the head dimension (<= HDP)
    float scale;
    int tiles;                  // 128-column tiles per (head, clip)
};

// Workgroup -> (128-column tile, head x clip).  The tiles of one (head, clip) read the same rows in their loops - K / V (forward, query
// gradient) or Q / dOut (key / value gradient) - so they are numbered consecutively on ONE XCD (workgroup L runs on XCD L % 8, each XCD
// has its own L2): numbered tile-fastest across the grid they sat on all eight, and every L2 fetched every (head, clip)'s rows (round 6).
// The launch is one-dimensional, (head x clip) padded to a multiple of 8; false = padding workgroup.
__device__ __forceinline__ bool attn_tile(const AttnParams &p, int &tx, int &b) {
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    b = (j / p.tiles) * 8 + xcd;
    tx = j - (j / p.tiles) * p.tiles;
    return b < p.H * p.N;
}

// cooperative load of a (64 x 32) tile X[dd][t0 + c] of a (d x T) matrix into LDS [dd][TP]; elements outside the matrix read as zero
// in two halves - fetch (global -> registers, issued BEFORE the tile in flight is multiplied) and commit (registers -> LDS, after) -
// so the load latency hides behind the MFMAs of the current tile.  Buffer loads: an out-of-range offset returns 0 without a branch
// (a conditional load, or a select after the load that the compiler turns into one, costs an exec-mask branch each and cuts the
// basic block the MFMA chains are scheduled in).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ float buf_f32(rsrc_t r, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); }
__device__ __forceinline__ rsrc_t head_rsrc(const float *base, int d, long long T) { return make_uniform_rsrc(base, (int)(d * T * 4)); }
template <int HDP>
__device__ __forceinline__ void fetch_tile(rsrc_t src, long long T, int d, int t0, int tid, float (&v)[HDP / 8]) {
    const int c = tid & 31, t = t0 + c;
#pragma unroll
    for (int u = 0; u < HDP / 8; ++u) {
        const int dd = (tid >> 5) + 8 * u;
        v[u] = buf_f32(src, (t < T && dd < d) ? (unsigned)(dd * (int)T + t) * 4u : kOOB);
    }
}
template <int HDP>
__device__ __forceinline__ void commit_tile(float *dst, int tid, const float (&v)[HDP / 8]) {
#pragma unroll
    for (int u = 0; u < HDP / 8; ++u) {
        const int idx = tid + 256 * u;
        dst[(idx >> 5) * TP + (idx & 31)] = v[u];
    }
}
template <int HDP>
__device__ __forceinline__ void load_tile(rsrc_t src, long long T, int d, int t0, float *dst, int tid) {
    float v[HDP / 8];
    fetch_tile<HDP>(src, T, d, t0, tid, v);
    commit_tile<HDP>(dst, tid, v);
}
// B-operand fragment of a (d x T) matrix for the wave's 32 columns: f[s] = X[2 s + kk][t0 + li]
template <int HDP>
__device__ __forceinline__ void load_frag(rsrc_t src, long long T, int d, int t0, int li, int kk, float (&f)[HDP / 2]) {
    const int t = t0 + li;
#pragma unroll
    for (int s = 0; s < HDP / 2; ++s) {
        const int dd = 2 * s + kk;
        f[s] = buf_f32(src, (t < T && dd < d) ? (unsigned)(dd * (int)T + t) * 4u : kOOB);
    }
}
// 32-bit word: bit i set <=> key t0 + i is masked (padding) or beyond T
__device__ __forceinline__ unsigned key_bits(const unsigned char *mrow, int T, int t0, int lane) {
    const int t = t0 + (lane & 31);
    const bool bad = t >= T || (mrow && mrow[t]);
    return (unsigned)__builtin_amdgcn_ballot_w64(bad && lane < 32);
}
// The words of ALL key tiles, once per workgroup, in LDS.  Evaluated inside the tile loop the mask byte was a dependent global load
// per tile whose wait (vmcnt(0)) also waited for the K / V tile just requested for the NEXT iteration - the prefetch never overlapped.
constexpr int KBITS_MAX = 512;          // tiles a table holds (T <= 16384); beyond that the loop falls back to key_bits
__device__ __forceinline__ void fill_key_bits(unsigned *tab, const unsigned char *mrow, int T, int ntile, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < ntile && i < KBITS_MAX; i += 4) {
        const unsigned w = key_bits(mrow, T, 32 * i, lane);
        if (lane == 0) tab[i] = w;
    }
}
// S tile: acc[i = rows of the LDS tile][j = the fragment's columns] = sum_dd tile[dd][i] * frag[dd][j]
template <int HDP>
__device__ __forceinline__ void mma_tile_frag(const float *tile, const float (&frag)[HDP / 2], int li, int kk, f32x16 &acc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < HDP / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(2 * s + kk) * TP + li], frag[s], acc, 0, 0, 0);
}
// two score-type tiles at once (S and dP of the backward kernels): acc0 = tile0^T frag0, acc1 = tile1^T frag1.  The A operands are read
// from LDS TWO k-steps ahead of the MFMAs that use them and the order is pinned - left alone hipcc reads each operand right in front
// of its MFMA and the pair waits ~100 cycles for it, 32 times per tile.
template <int HDP>
__device__ __forceinline__ void mma_tile_frag2(const float *tile0, const float (&frag0)[HDP / 2], const float *tile1, const float (&frag1)[HDP / 2],
                                               int li, int kk, f32x16 &acc0, f32x16 &acc1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
    constexpr int NS = HDP / 2;
    float a0[3], a1[3];
    a0[0] = tile0[kk * TP + li], a1[0] = tile1[kk * TP + li];
    a0[1] = tile0[(2 + kk) * TP + li], a1[1] = tile1[(2 + kk) * TP + li];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 2 < NS) a0[(s + 2) % 3] = tile0[(2 * (s + 2) + kk) * TP + li], a1[(s + 2) % 3] = tile1[(2 * (s + 2) + kk) * TP + li];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s % 3], frag0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s % 3], frag1[s], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
}
// O[dd][j] += sum over the tile's 32 columns c of tile[dd][c] * P[c][j], P in accumulator layout (register s <-> row rho(s, kk))
template <int HDP>
__device__ __forceinline__ void mma_tile_acc(const float *tile, const f32x16 &P, int li, int kk, f32x16 (&O)[HDP / 32]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int c = rho(s, kk);
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt) O[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(mt * 32 + li) * TP + c], P[s], O[mt], 0, 0, 0);
    }
}

// forward.  One wave = 32 query columns; pass 1: column statistics (max, sum) over all keys; pass 2: probabilities (written to
// `att` when asked for) and out = V P.  grid (ceil(T / 128), H * N)
template <int HDP, bool ATT>
__global__ __launch_bounds__(256, ATT ? 2 : 3) void attn_fwd_kernel(AttnParams p) {
    __shared__ float sK[2][HDP * TP], sV[2][HDP * TP];
    __shared__ unsigned s_kb[KBITS_MAX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kk = lane >> 5;
    int tx, b;
    if (!attn_tile(p, tx, b)) return;
    const int h = b / p.N, n = b - h * p.N;
    const long long T = p.T;
    const float *Kg = p.kvq + ((long long)n * 3 * p.C + h * p.d) * T;
    const rsrc_t Kp = head_rsrc(Kg, p.d, T), Vp = head_rsrc(Kg + (long long)p.C * T, p.d, T), Qp = head_rsrc(Kg + 2 * (long long)p.C * T, p.d, T);
    const unsigned char *mrow = p.mask ? p.mask + (long long)n * T : nullptr;
    const int tq0 = tx * 128 + wave * 32, tq = tq0 + li;
    float qf[HDP / 2];
    load_frag<HDP>(Qp, T, p.d, tq0, li, kk, qf);
    const int ntile = (p.T + 31) / 32;
    // Both passes are software-pipelined inside the wave: the score tile of key tile it + 1 is multiplied (a chain of dependent
    // MFMAs, 64 cycles each, that leaves the VALU idle) WHILE the exponentials of key tile it are evaluated - independent work in
    // one basic block, so the scheduler can fill the MFMA shadows.  K therefore runs one tile ahead of V in LDS:
    //   iteration it: barrier | request K(it+2), V(it+1) | S(it+1) from sK[(it+1)&1]  ||  softmax of S(it) | O += V(it) P from
    //   sV[it&1] | commit K(it+2) -> sK[it&1] (K(it) was consumed in iteration it-1), V(it+1) -> sV[(it+1)&1]
    float pk[HDP / 8], pv[HDP / 8];
    if constexpr (!ATT) {
        // ONE pass when the probabilities are not returned (round 4; see attn_fwd_bf16_kernel): running column maximum, rescaled accumulator
        float mx = -INFINITY, sum = 0.f;
        f32x16 O[HDP / 32];
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[mt][i] = 0.f;
        load_tile<HDP>(Kp, T, p.d, 0, sK[0], tid);
        load_tile<HDP>(Kp, T, p.d, 32, sK[1], tid);
        load_tile<HDP>(Vp, T, p.d, 0, sV[0], tid);
        fill_key_bits(s_kb, mrow, p.T, ntile, tid);
        __syncthreads();
        f32x16 s;
        mma_tile_frag<HDP>(sK[0], qf, li, kk, s);
        for (int it = 0; it < ntile; ++it) {
            __syncthreads();
            fetch_tile<HDP>(Kp, T, p.d, 32 * (it + 2), tid, pk);
            fetch_tile<HDP>(Vp, T, p.d, 32 * (it + 1), tid, pv);
            const unsigned bad = it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane);
            f32x16 sn;
            mma_tile_frag<HDP>(sK[(it + 1) & 1], qf, li, kk, sn);
            float tm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = (bad >> rho(r, kk)) & 1u ? -INFINITY : s[r] * p.scale;
                s[r] = v;
                tm = fmaxf(tm, v);
            }
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float m2 = fmaxf(mx, tm);
            const float m2s = m2 > -INFINITY ? m2 : 0.f;
            const float alpha = __expf(mx - m2s);
            float asum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(s[r] - m2s);
                s[r] = e;
                asum += e;
            }
            sum = sum * alpha + asum;
            mx = m2;
#pragma unroll
            for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[mt][i] *= alpha;
            mma_tile_acc<HDP>(sV[it & 1], s, li, kk, O);
            commit_tile<HDP>(sK[it & 1], tid, pk);
            commit_tile<HDP>(sV[(it + 1) & 1], tid, pv);
            s = sn;
        }
        sum += __shfl_xor(sum, 32, 64);
        const bool qpad = tq < p.T && mrow && mrow[tq];
        const float inv = 1.f / sum;
        const bool nancol = !(sum > 0.f);
        if (tq < p.T && kk == 0) {
            p.stats[((long long)b * T + tq) * 2] = mx;
            p.stats[((long long)b * T + tq) * 2 + 1] = inv;
        }
        if (tq < p.T) {
            float *op = p.out + ((long long)n * p.C + h * p.d) * T + tq;
#pragma unroll
            for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mt * 32 + rho(r, kk) < p.d) op[(long long)(mt * 32 + rho(r, kk)) * T] = qpad ? 0.f : (nancol ? NAN : O[mt][r] * inv);
        }
        return;
    }
    // ---- pass 1
    float mx = -INFINITY, sum = 0.f;
    load_tile<HDP>(Kp, T, p.d, 0, sK[0], tid);
    load_tile<HDP>(Kp, T, p.d, 32, sK[1], tid);
    fill_key_bits(s_kb, mrow, p.T, ntile, tid);
    __syncthreads();
    f32x16 s;
    mma_tile_frag<HDP>(sK[0], qf, li, kk, s);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        fetch_tile<HDP>(Kp, T, p.d, 32 * (it + 2), tid, pk);
        const unsigned bad = it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane);
        f32x16 sn;
        mma_tile_frag<HDP>(sK[(it + 1) & 1], qf, li, kk, sn);       // past the last tile: zeros, never used
        float tm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (bad >> rho(r, kk)) & 1u ? -INFINITY : s[r] * p.scale;
            s[r] = v;
            tm = fmaxf(tm, v);
        }
        const float m2 = fmaxf(mx, tm);
        const float m2s = m2 > -INFINITY ? m2 : 0.f;               // every key so far padded: all terms exp(-inf - 0) = 0
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += __expf(s[r] - m2s);
        sum = sum * __expf(mx - m2s) + a;
        mx = m2;
        commit_tile<HDP>(sK[it & 1], tid, pk);
        s = sn;
    }
    {   // the two lane halves hold different key rows of the same query column
        const float om = __shfl_xor(mx, 32, 64), os = __shfl_xor(sum, 32, 64);
        const float m2 = fmaxf(mx, om);
        sum = (mx > -INFINITY ? sum * __expf(mx - m2) : 0.f) + (om > -INFINITY ? os * __expf(om - m2) : 0.f);
        mx = m2;
    }
    const bool qpad = tq < p.T && mrow && mrow[tq];
    const float inv = 1.f / sum;                 // every key padded: 1 / 0 -> the NaN column torch's softmax of all -inf gives
    if (tq < p.T && kk == 0) {
        p.stats[((long long)b * T + tq) * 2] = mx;
        p.stats[((long long)b * T + tq) * 2 + 1] = inv;
    }
    // ---- pass 2
    f32x16 O[HDP / 32];
#pragma unroll
    for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[mt][i] = 0.f;
    const bool nancol = !(sum > 0.f);            // no unpadded key at all: torch gives NaN for every entry of the column
    const float mxs = nancol ? 0.f : mx;
    __syncthreads();
    load_tile<HDP>(Kp, T, p.d, 0, sK[0], tid);
    load_tile<HDP>(Kp, T, p.d, 32, sK[1], tid);
    load_tile<HDP>(Vp, T, p.d, 0, sV[0], tid);
    __syncthreads();
    mma_tile_frag<HDP>(sK[0], qf, li, kk, s);
    float *attp = ATT ? p.att + (long long)b * T * T + tq : nullptr;      // ATT: the (H N, T, T) tensor is written (its own instances)
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        fetch_tile<HDP>(Kp, T, p.d, 32 * (it + 2), tid, pk);
        fetch_tile<HDP>(Vp, T, p.d, 32 * (it + 1), tid, pv);
        const unsigned bad = it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane);
        f32x16 sn;
        mma_tile_frag<HDP>(sK[(it + 1) & 1], qf, li, kk, sn);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(s[r] * p.scale - mxs) * inv;
            s[r] = ((bad >> rho(r, kk)) & 1u) || qpad ? 0.f : e;
        }
        if (__builtin_amdgcn_ballot_w64(nancol && !qpad) != 0) {   // rare, wave-uniform: the NaN semantics stay out of the pipelined block
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (nancol && !qpad) ? (32 * it + rho(r, kk) < p.T ? NAN : 0.f) : s[r];
        }
        if (attp && tq < p.T) {
            float *ap = attp + (long long)(32 * it + 4 * kk) * T;
            if (32 * it + 32 <= p.T) {           // whole tile: one predicate for all sixteen stores
#pragma unroll
                for (int r = 0; r < 16; ++r) ap[(long long)rho(r, 0) * T] = s[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * it + rho(r, kk) < p.T) ap[(long long)rho(r, 0) * T] = s[r];
            }
        }
        mma_tile_acc<HDP>(sV[it & 1], s, li, kk, O);
        commit_tile<HDP>(sK[it & 1], tid, pk);
        commit_tile<HDP>(sV[(it + 1) & 1], tid, pv);
        s = sn;
    }
    if (tq < p.T) {
        float *op = p.out + ((long long)n * p.C + h * p.d) * T + tq;
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mt * 32 + rho(r, kk) < p.d) op[(long long)(mt * 32 + rho(r, kk)) * T] = O[mt][r];
    }
}

// delta[b][tq] = sum_dd gout[dd][tq] * out[dd][tq]  (+ sum_tk att[tk][tq] * gatt[tk][tq]): the softmax-backward column term
// (ET = unsigned short: out and gout are STORED as bf16 - psnd_mha_bwd with bf16 = 2)
template <typename ET>
__global__ __launch_bounds__(256) void attn_delta_kernel(const ET *out, const ET *gout, const float *att, const float *gatt, int N, int H,
                                                         int C, int d, long long T, float *delta) {
    const int b = blockIdx.y, h = b / N, n = b - h * N;
    const long long tq = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tq >= T) return;
    const ET *o = out + ((long long)n * C + h * d) * T + tq, *g = gout + ((long long)n * C + h * d) * T + tq;
    auto val = [](ET x) __attribute__((always_inline)) -> float {
        if constexpr (sizeof(ET) == 2) return __builtin_bit_cast(float, (unsigned)x << 16);
        else return x;
    };
    float a = 0.f;
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int dd = 0; dd + 3 < d; dd += 4) {          // 8 independent loads in flight
#pragma unroll
        for (int e = 0; e < 4; ++e) a4[e] = __builtin_fmaf(val(o[(dd + e) * T]), val(g[(dd + e) * T]), a4[e]);
    }
    for (int dd = d & ~3; dd < d; ++dd) a = __builtin_fmaf(val(o[dd * T]), val(g[dd * T]), a);
    a += (a4[0] + a4[1]) + (a4[2] + a4[3]);
    if (gatt) {
        const float *pa = att + (long long)b * T * T + tq, *pg = gatt + (long long)b * T * T + tq;
        for (long long tk = 0; tk < T; ++tk) a = __builtin_fmaf(pa[tk * T], pg[tk * T], a);
    }
    delta[(long long)b * T + tq] = a;
}

// backward, key side.  One wave = 32 keys (its K and V fragments stay in registers), loop over query tiles:
//   S'[tq][tk], dP'[tq][tk] (+ gatt) -> p', dS' = scale p' (dP' - delta) -> dV += gO p', dK += Q dS'.   grid (ceil(T / 128), H * N)
template <int HDP, bool GATT>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(AttnParams p) {
    __shared__ float sQ[2][HDP * TP], sG[2][HDP * TP];
    __shared__ __attribute__((aligned(16))) float sSt[2][32 * 4];           // per query of the tile: max, 1 / sum, delta, query padded
    __shared__ float sT[4][32 * TP];                                     // per wave: a gatt tile, transposed through LDS
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kk = lane >> 5;
    int tx, b;
    if (!attn_tile(p, tx, b)) return;
    const int h = b / p.N, n = b - h * p.N;
    const long long T = p.T;
    const float *Kg = p.kvq + ((long long)n * 3 * p.C + h * p.d) * T;
    const rsrc_t Kp = head_rsrc(Kg, p.d, T), Vp = head_rsrc(Kg + (long long)p.C * T, p.d, T), Qp = head_rsrc(Kg + 2 * (long long)p.C * T, p.d, T);
    const rsrc_t Gp = head_rsrc(p.gout + ((long long)n * p.C + h * p.d) * T, p.d, T);
    const unsigned char *mrow = p.mask ? p.mask + (long long)n * T : nullptr;
    const int tk0 = tx * 128 + wave * 32, tk = tk0 + li;
    const bool kbad = tk >= p.T || (mrow && mrow[tk]);
    float kf[HDP / 2], vf[HDP / 2];
    load_frag<HDP>(Kp, T, p.d, tk0, li, kk, kf);
    load_frag<HDP>(Vp, T, p.d, tk0, li, kk, vf);
    f32x16 dK[HDP / 32], dV[HDP / 32];
#pragma unroll
    for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dK[mt][i] = 0.f, dV[mt][i] = 0.f;
    const int ntile = (p.T + 31) / 32;
    float pq[HDP / 8], pg[HDP / 8];
    auto stage = [&](int it, int buf) __attribute__((always_inline)) {
        if (tid < 32) {
            const int t = 32 * it + tid;
            f32x4_t st = {0.f, 0.f, 0.f, 1.f};
            if (t < p.T) {
                st[0] = p.stats[((long long)b * T + t) * 2];
                st[1] = p.stats[((long long)b * T + t) * 2 + 1];
                st[2] = p.delta[(long long)b * T + t];
                st[3] = (mrow && mrow[t]) ? 1.f : 0.f;
            }
            *reinterpret_cast<f32x4_t *>(&sSt[buf][4 * tid]) = st;
        }
    };
    stage(0, 0);
    load_tile<HDP>(Qp, T, p.d, 0, sQ[0], tid);
    load_tile<HDP>(Gp, T, p.d, 0, sG[0], tid);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        if (it + 1 < ntile) {
            stage(it + 1, (it + 1) & 1);
            fetch_tile<HDP>(Qp, T, p.d, 32 * (it + 1), tid, pq);
            fetch_tile<HDP>(Gp, T, p.d, 32 * (it + 1), tid, pg);
        }
        const float *tQ = sQ[it & 1], *tG = sG[it & 1], *tS = sSt[it & 1];
        f32x16 s, dp;
        mma_tile_frag2<HDP>(tQ, kf, tG, vf, li, kk, s, dp);     // rows: queries of the tile, column: this lane's key
        if (it + 1 < ntile) {
            commit_tile<HDP>(sQ[(it + 1) & 1], tid, pq);
            commit_tile<HDP>(sG[(it + 1) & 1], tid, pg);
        }
        if constexpr (GATT) {                                // gatt[tk][tq] is query-contiguous: through LDS, read transposed
            float *tt = sT[wave];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = 2 * u + kk, k2 = tk0 + r, q2 = 32 * it + li;     // row r of the wave's key tile, lanes along the queries
                tt[r * TP + li] = (k2 < p.T && q2 < p.T) ? p.gatt[((long long)b * T + k2) * T + q2] : 0.f;
            }
            __builtin_amdgcn_s_waitcnt(0);           // wave-private region: the wave's own writes are ordered by the LDS queue
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] += tt[li * TP + rho(r, kk)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4_t st = *reinterpret_cast<const f32x4_t *>(&tS[4 * rho(r, kk)]);
            const bool dead = kbad || st[3] > 0.f || 32 * it + rho(r, kk) >= p.T;
            // (selects, not `dead ? 0 : exp(..)`: that form compiles to an exec-mask branch around every one of the 16 exponentials of a tile -
            //  432 -> 388 us per config-4 launch; exp(-inf) is exactly 0)
            const float pr = __expf(dead ? -INFINITY : s[r] * p.scale - st[0]) * (dead ? 0.f : st[1]);
            s[r] = pr;
            dp[r] = p.scale * pr * (dp[r] - st[2]);
        }
        mma_tile_acc<HDP>(tG, s, li, kk, dV);
        mma_tile_acc<HDP>(tQ, dp, li, kk, dK);
    }
    if (tk < p.T) {
        float *gk = p.gkvq + ((long long)n * 3 * p.C + h * p.d) * T + tk, *gv = gk + (long long)p.C * T;
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mt * 32 + rho(r, kk) < p.d) {
                    gk[(long long)(mt * 32 + rho(r, kk)) * T] = dK[mt][r];
                    gv[(long long)(mt * 32 + rho(r, kk)) * T] = dV[mt][r];
                }
    }
}

// backward, query side.  One wave = 32 queries (Q and gO fragments in registers), loop over key tiles:
//   S, dP (+ gatt) -> p, dS = scale p (dP - delta) -> dQ += K dS.    grid (ceil(T / 128), H * N)
template <int HDP, bool GATT>
__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(AttnParams p) {      // (three waves per SIMD: 24 spilled registers)
    __shared__ float sK[2][HDP * TP], sV[2][HDP * TP];
    __shared__ unsigned s_kb[KBITS_MAX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kk = lane >> 5;
    int tx, b;
    if (!attn_tile(p, tx, b)) return;
    const int h = b / p.N, n = b - h * p.N;
    const long long T = p.T;
    const float *Kg = p.kvq + ((long long)n * 3 * p.C + h * p.d) * T;
    const rsrc_t Kp = head_rsrc(Kg, p.d, T), Vp = head_rsrc(Kg + (long long)p.C * T, p.d, T), Qp = head_rsrc(Kg + 2 * (long long)p.C * T, p.d, T);
    const rsrc_t Gp = head_rsrc(p.gout + ((long long)n * p.C + h * p.d) * T, p.d, T);
    const unsigned char *mrow = p.mask ? p.mask + (long long)n * T : nullptr;
    const int tq0 = tx * 128 + wave * 32, tq = tq0 + li;
    float qf[HDP / 2], gf[HDP / 2];
    load_frag<HDP>(Qp, T, p.d, tq0, li, kk, qf);
    load_frag<HDP>(Gp, T, p.d, tq0, li, kk, gf);
    const bool qdead = tq >= p.T || (mrow && mrow[tq]);
    const float mx = tq < p.T ? p.stats[((long long)b * T + tq) * 2] : 0.f, inv = tq < p.T ? p.stats[((long long)b * T + tq) * 2 + 1] : 0.f;
    const float dl = tq < p.T ? p.delta[(long long)b * T + tq] : 0.f;
    f32x16 dQ[HDP / 32];
#pragma unroll
    for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQ[mt][i] = 0.f;
    const int ntile = (p.T + 31) / 32;
    load_tile<HDP>(Kp, T, p.d, 0, sK[0], tid);
    load_tile<HDP>(Vp, T, p.d, 0, sV[0], tid);
    fill_key_bits(s_kb, mrow, p.T, ntile, tid);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        float pk[HDP / 8], pv[HDP / 8];
        if (it + 1 < ntile) {
            fetch_tile<HDP>(Kp, T, p.d, 32 * (it + 1), tid, pk);
            fetch_tile<HDP>(Vp, T, p.d, 32 * (it + 1), tid, pv);
        }
        const unsigned bad = it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane);
        f32x16 s, dp;
        mma_tile_frag2<HDP>(sK[it & 1], qf, sV[it & 1], gf, li, kk, s, dp);
        if (it + 1 < ntile) {
            commit_tile<HDP>(sK[(it + 1) & 1], tid, pk);
            commit_tile<HDP>(sV[(it + 1) & 1], tid, pv);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rho(r, kk);
            const bool dead = qdead || ((bad >> row) & 1u);
            const float pr = __expf(dead ? -INFINITY : s[r] * p.scale - mx) * (dead ? 0.f : inv);           // (selects, not a branch per element)
            float g = dp[r];
            if (GATT && !dead) g += p.gatt[((long long)b * T + 32 * it + row) * T + tq];
            dp[r] = p.scale * pr * (g - dl);
        }
        mma_tile_acc<HDP>(sK[it & 1], dp, li, kk, dQ);
    }
    if (tq < p.T) {
        float *gq = p.gkvq + ((long long)n * 3 * p.C + 2 * p.C + h * p.d) * T + tq;
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mt * 32 + rho(r, kk) < p.d) gq[(long long)(mt * 32 + rho(r, kk)) * T] = dQ[mt][r];
    }
}


// =====================================================================================================================
// The same three attention kernels with bf16 OPERANDS (fp32 scores, softmax statistics and accumulation) on
// v_mfma_f32_32x32x16_bf16: 16 x the matrix rate of the exact-fp32 form above, which is bound by fp32 MFMA issue (~50 % of the
// 157 TFLOP/s fp32 matrix peak at T = 1292).  Chosen by the caller (`bf16` of psnd_mha_fwd / _bwd: the module passes it under
// torch.autocast(bfloat16), as it does for the projections).  Structure, masks, statistics and the software pipelining are those
// of the fp32 kernels; what changes is the operand staging:
//   * a (d x 32) tile of K / V / Q / gO lives in LDS as bf16 in the orientation(s) its products need:
//       tT [32 t][HDP + 8]  (d contiguous)  - A operand of a SCORE product  S[t][j] = sum_dd X[dd][t] f[dd][j]: one ds_read_b128
//       tD [HDP][36]        (t contiguous)  - A operand of an ACCUMULATE product O[dd][j] += sum_c X[dd][c] P[c][j]: two ds_read_b64
//     (a thread converts a 2 x 2 block of fp32 values and writes packed pairs: 32-bit LDS stores in both orientations);
//   * the probabilities / score gradients in the accumulator registers ARE the B operand of the accumulate products after a
//     v_cvt_pk_bf16_f32 per register pair: B slot e of lane half h of the product over keys 16 s .. 16 s + 15 is accumulator
//     register 8 s + e = row 16 s + 8 (e >> 2) + 4 h + (e & 3) - the A operand is read in that same permuted order.
// =====================================================================================================================
template <int HDP>
struct BTile {
    static constexpr int DP = HDP + 8;           // tT row pitch (bf16): 144 B at HDP = 64 - 16-byte aligned rows, conflict-free b128 reads
    static constexpr int TPB = 36;               // tD row pitch (bf16): 72 B - 8-byte aligned, the 32 rows of a read on 64 distinct banks
    static constexpr int TT = 32 * DP, TD = HDP * TPB;
    static constexpr int NP = HDP > 64 ? HDP / 64 : 1;      // passes of the 256 threads over a tile: 64 rows x 32 columns a pass
};
// the thread's 2 x 4 blocks of a (HDP x 32) tile: v[u] = { X[dd][t .. t + 3], X[dd + 1][t .. t + 3] }, t = t0 + 4 (tid & 7),
// dd = 2 (tid >> 3) + 64 u.  A tile INSIDE the matrix (d == HDP, t0 + 32 <= T: all but the last one at the model's sizes) takes two
// 16-byte loads per block whose per-thread offset never changes - the tile's position rides in the instruction's scalar offset, no
// address arithmetic or range selects in the loop (round 6: the loops are bound by their vector ALU work; rows are 4-byte aligned only,
// T = 1292 - gfx9 memory takes that); any other tile element loads that read zeros outside the matrix.
// H: the matrix is STORED as bf16 (kvq written by the projection's epilogue under autocast, round 6): the same 2 x 4 block is two 8-byte loads,
// v[u][0..3] then hold the packed words {row dd: t, t+1 | t+2, t+3}, {row dd + 1: ...} - nothing to convert on the way to LDS.
template <int HDP, bool H = false>
__device__ __forceinline__ void fetch_tile_b(rsrc_t src, long long T, int d, int t0, int tid, float (&v)[BTile<HDP>::NP][8]) {
    const int tl = 4 * (tid & 7);
    const bool inside = d == HDP && t0 + 32 <= (int)T;
#pragma unroll
    for (int u = 0; u < BTile<HDP>::NP; ++u) {
        const int dd = 2 * (tid >> 3) + 64 * u;
        if constexpr (H) {
            unsigned w[4] = {0u, 0u, 0u, 0u};
            if (inside) {
                const unsigned voff = dd < HDP ? (unsigned)(dd * (int)T + tl) * 2u : kOOB;
                const uint2 a = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(src, (int)voff, t0 * 2, 0));
                const uint2 c = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(src, (int)(voff + (unsigned)T * 2u), t0 * 2, 0));
                w[0] = a.x, w[1] = a.y, w[2] = c.x, w[3] = c.y;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int te = t0 + tl + (e & 3), de = dd + (e >> 2);
                    const unsigned x = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(src, (te < T && de < d) ? (unsigned)(de * (int)T + te) * 2u : kOOB, 0, 0);
                    w[e >> 1] |= x << (16 * (e & 1));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = __builtin_bit_cast(float, w[e]);
        } else if (inside) {
            const unsigned voff = dd < HDP ? (unsigned)(dd * (int)T + tl) * 4u : kOOB;
            const f32x4_t a = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(src, (int)voff, t0 * 4, 0));
            const f32x4_t c = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(src, (int)(voff + (unsigned)T * 4u), t0 * 4, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = a[e], v[u][4 + e] = c[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int te = t0 + tl + (e & 3), de = dd + (e >> 2);
                v[u][e] = buf_f32(src, (te < T && de < d) ? (unsigned)(de * (int)T + te) * 4u : kOOB);
            }
        }
    }
}
template <int HDP, bool WT, bool WD, bool H = false>
__device__ __forceinline__ void commit_tile_b(unsigned short *tT, unsigned short *tD, int tid, const float (&v)[BTile<HDP>::NP][8]) {
    using B = BTile<HDP>;
    const int tl = 4 * (tid & 7);
    if (HDP < 64 && tid >= 4 * HDP) return;      // (HDP = 32: the tile has 32 rows, the upper 128 threads hold zeros)
#pragma unroll
    for (int u = 0; u < B::NP; ++u) {
        const int dd = 2 * (tid >> 3) + 64 * u;
        if constexpr (H) {
            const unsigned w0 = __builtin_bit_cast(unsigned, v[u][0]), w1 = __builtin_bit_cast(unsigned, v[u][1]);
            const unsigned w2 = __builtin_bit_cast(unsigned, v[u][2]), w3 = __builtin_bit_cast(unsigned, v[u][3]);
            if constexpr (WT) {                  // (row dd, row dd + 1) at one frame: the low / high halves of the two rows' words
                *reinterpret_cast<unsigned *>(tT + (tl + 0) * B::DP + dd) = __builtin_amdgcn_perm(w2, w0, 0x05040100u);
                *reinterpret_cast<unsigned *>(tT + (tl + 1) * B::DP + dd) = __builtin_amdgcn_perm(w2, w0, 0x07060302u);
                *reinterpret_cast<unsigned *>(tT + (tl + 2) * B::DP + dd) = __builtin_amdgcn_perm(w3, w1, 0x05040100u);
                *reinterpret_cast<unsigned *>(tT + (tl + 3) * B::DP + dd) = __builtin_amdgcn_perm(w3, w1, 0x07060302u);
            }
            if constexpr (WD) {
                *reinterpret_cast<uint2 *>(tD + dd * B::TPB + tl) = make_uint2(w0, w1);
                *reinterpret_cast<uint2 *>(tD + (dd + 1) * B::TPB + tl) = make_uint2(w2, w3);
            }
        } else {
            if constexpr (WT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) *reinterpret_cast<unsigned *>(tT + (tl + e) * B::DP + dd) = pack2_bf16(v[u][e], v[u][4 + e]);
            }
            if constexpr (WD) {
                *reinterpret_cast<uint2 *>(tD + dd * B::TPB + tl) = make_uint2(pack2_bf16(v[u][0], v[u][1]), pack2_bf16(v[u][2], v[u][3]));
                *reinterpret_cast<uint2 *>(tD + (dd + 1) * B::TPB + tl) = make_uint2(pack2_bf16(v[u][4], v[u][5]), pack2_bf16(v[u][6], v[u][7]));
            }
        }
    }
}
template <int HDP, bool WT, bool WD, bool H = false>
__device__ __forceinline__ void load_tile_b(rsrc_t src, long long T, int d, int t0, unsigned short *tT, unsigned short *tD, int tid) {
    float v[BTile<HDP>::NP][8];
    fetch_tile_b<HDP, H>(src, T, d, t0, tid, v);
    commit_tile_b<HDP, WT, WD, H>(tT, tD, tid, v);
}
// the descriptor of one head's (d x T) rows of a matrix stored as fp32 or (H) bf16, `off` elements into `base`
template <bool H>
__device__ __forceinline__ rsrc_t head_rsrc_e(const void *base, long long off, int d, long long T) {
    if constexpr (H) return make_uniform_rsrc(static_cast<const unsigned short *>(base) + off, (int)(d * T * 2));
    else return make_uniform_rsrc(static_cast<const float *>(base) + off, (int)(d * T * 4));
}
// B-operand fragments of a (d x T) matrix for the wave's 32 columns: f[s] slot e = X[16 s + 8 kk + e][t0 + li]
template <int HDP, bool H = false>
__device__ __forceinline__ void load_frag_b(rsrc_t src, long long T, int d, int t0, int li, int kk, bf16x8_t (&f)[HDP / 16]) {
    const int t = t0 + li;
#pragma unroll
    for (int s = 0; s < HDP / 16; ++s) {
        if constexpr (H) {
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int dd = 16 * s + 8 * kk + e;
                const unsigned x = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(src, (t < T && dd < d) ? (unsigned)(dd * (int)T + t) * 2u : kOOB, 0, 0);
                w[e >> 1] |= x << (16 * (e & 1));
            }
            f[s] = __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
        } else {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int dd = 16 * s + 8 * kk + e;
                x[e] = buf_f32(src, (t < T && dd < d) ? (unsigned)(dd * (int)T + t) * 4u : kOOB);
            }
            const uint4 pk = make_uint4(pack2_bf16(x[0], x[1]), pack2_bf16(x[2], x[3]), pack2_bf16(x[4], x[5]), pack2_bf16(x[6], x[7]));
            f[s] = __builtin_bit_cast(bf16x8_t, pk);
        }
    }
}
// S tile: acc[i = rows t of the tile][j] = sum_dd X[dd][i] frag[dd][j]
template <int HDP>
__device__ __forceinline__ void mma_tile_frag_b(const unsigned short *tT, const bf16x8_t (&frag)[HDP / 16], int li, int kk, f32x16 &acc) {
    using B = BTile<HDP>;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < HDP / 16; ++s) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t *>(tT + li * B::DP + 16 * s + 8 * kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag[s], acc, 0, 0, 0);
    }
}
// accumulator tile -> the two B operands (keys 0-15, 16-31) of an accumulate product
__device__ __forceinline__ void pack_acc_b(const f32x16 &P, bf16x8_t (&b)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint4 pk = make_uint4(pack2_bf16(P[8 * s], P[8 * s + 1]), pack2_bf16(P[8 * s + 2], P[8 * s + 3]),
                                    pack2_bf16(P[8 * s + 4], P[8 * s + 5]), pack2_bf16(P[8 * s + 6], P[8 * s + 7]));
        b[s] = __builtin_bit_cast(bf16x8_t, pk);
    }
}
// O[dd][j] += sum over the tile's 32 columns c of X[dd][c] P[c][j]
template <int HDP>
__device__ __forceinline__ void mma_tile_acc_b(const unsigned short *tD, const bf16x8_t (&b)[2], int li, int kk, f32x16 (&O)[HDP / 32]) {
    using B = BTile<HDP>;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt) {
            const unsigned short *row = tD + (mt * 32 + li) * B::TPB + 16 * s + 4 * kk;
            const uint2 lo = *reinterpret_cast<const uint2 *>(row), hi = *reinterpret_cast<const uint2 *>(row + 8);
            const uint4 a4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
            O[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a4), b[s], O[mt], 0, 0, 0);
        }
}

// KH: kvq (and, backward, gkvq) STORED as bf16 (psnd_mha_fwd / _bwd with bf16 = 2)
template <int HDP, bool ATT, bool KH = false>
__global__ __launch_bounds__(256, ATT ? 2 : 3) void attn_fwd_bf16_kernel(AttnParams p) {
    using B = BTile<HDP>;
    __shared__ __attribute__((aligned(16))) unsigned short sK[2][B::TT], sV[2][B::TD];
    __shared__ unsigned s_kb[KBITS_MAX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kk = lane >> 5;
    int tx, b;
    if (!attn_tile(p, tx, b)) return;
    const int h = b / p.N, n = b - h * p.N;
    const long long T = p.T;
    const long long Ko = ((long long)n * 3 * p.C + h * p.d) * T;
    const rsrc_t Kp = head_rsrc_e<KH>(p.kvq, Ko, p.d, T), Vp = head_rsrc_e<KH>(p.kvq, Ko + (long long)p.C * T, p.d, T),
                 Qp = head_rsrc_e<KH>(p.kvq, Ko + 2 * (long long)p.C * T, p.d, T);
    const unsigned char *mrow = p.mask ? p.mask + (long long)n * T : nullptr;
    const int tq0 = tx * 128 + wave * 32, tq = tq0 + li;
    bf16x8_t qf[HDP / 16];
    load_frag_b<HDP, KH>(Qp, T, p.d, tq0, li, kk, qf);
    const int ntile = (p.T + 31) / 32;
    float pk[B::NP][8], pv[B::NP][8];
    if constexpr (!ATT) {
        // ---- ONE pass when the probabilities themselves are not returned (round 4): running column maximum and sum, the accumulator
        // rescaled when the maximum moves (O *= exp(m_old - m_new): 32 multiplies per 32 x 32 tile) - the second K^T Q product of the
        // two-pass form, its exponentials and its K tile loads are gone.  The statistics written for the backward are the same
        // (global maximum, 1 / sum).
        // Round 6: the loop is bound by its vector ALU work (248 instructions next to 8 matrix ones), so: the maximum runs over the RAW
        // scores and `scale` rides in the exponential's FMA (exp2(s c - m c), c = scale log2(e): the form the backward kernels recompute
        // the probabilities in); a tile without a masked key (the key bits are wave-uniform) skips the sixteen bit tests and selects.
        float mx = -INFINITY, sum = 0.f;                          // mx: running maximum of the raw scores
        const float cexp = p.scale * 1.44269504088896341f;
        f32x16 O[HDP / 32];
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[mt][i] = 0.f;
        load_tile_b<HDP, true, false, KH>(Kp, T, p.d, 0, sK[0], nullptr, tid);
        load_tile_b<HDP, true, false, KH>(Kp, T, p.d, 32, sK[1], nullptr, tid);
        load_tile_b<HDP, false, true, KH>(Vp, T, p.d, 0, nullptr, sV[0], tid);
        fill_key_bits(s_kb, mrow, p.T, ntile, tid);
        __syncthreads();
        f32x16 s;
        mma_tile_frag_b<HDP>(sK[0], qf, li, kk, s);
        for (int it = 0; it < ntile; ++it) {
            __syncthreads();
            fetch_tile_b<HDP, KH>(Kp, T, p.d, 32 * (it + 2), tid, pk);
            fetch_tile_b<HDP, KH>(Vp, T, p.d, 32 * (it + 1), tid, pv);
            const unsigned bad = __builtin_amdgcn_readfirstlane(it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane));
            f32x16 sn;
            mma_tile_frag_b<HDP>(sK[(it + 1) & 1], qf, li, kk, sn);
            if (bad != 0) {
                const unsigned badl = bad >> (4 * kk);               // bit rho(r, 0) <-> register r
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (badl & (1u << rho(r, 0))) ? -INFINITY : s[r];
            }
            float tm = fmaxf(s[0], s[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) tm = fmaxf(tm, fmaxf(s[r], s[r + 1]));
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));            // the column's 32 keys sit in the two half-waves
            const float m2 = fmaxf(mx, tm);
            const float m2s = m2 > -INFINITY ? m2 : 0.f;
            const float alpha = __builtin_amdgcn_exp2f((mx - m2s) * cexp);      // (mx = -inf: 0, and O is still 0)
            const float mc = m2s * cexp;
            f32x2_v as2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2_v e = {__builtin_amdgcn_exp2f(__builtin_fmaf(s[r], cexp, -mc)), __builtin_amdgcn_exp2f(__builtin_fmaf(s[r + 1], cexp, -mc))};
                s[r] = e[0], s[r + 1] = e[1];
                as2 += e;
            }
            sum = sum * alpha + (as2[0] + as2[1]);
            mx = m2;
#pragma unroll
            for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[mt][i] *= alpha;
            bf16x8_t pb[2];
            pack_acc_b(s, pb);
            mma_tile_acc_b<HDP>(sV[it & 1], pb, li, kk, O);
            commit_tile_b<HDP, true, false, KH>(sK[it & 1], nullptr, tid, pk);
            commit_tile_b<HDP, false, true, KH>(nullptr, sV[(it + 1) & 1], tid, pv);
            s = sn;
        }
        mx *= p.scale;                                             // the statistics keep the maximum of the SCALED scores
        sum += __shfl_xor(sum, 32, 64);
        const bool qpad = tq < p.T && mrow && mrow[tq];
        const float inv = 1.f / sum;
        const bool nancol = !(sum > 0.f);
        if (tq < p.T && kk == 0) {
            p.stats[((long long)b * T + tq) * 2] = mx;
            p.stats[((long long)b * T + tq) * 2 + 1] = inv;
        }
        if (tq < p.T) {
            const long long oo = ((long long)n * p.C + h * p.d) * T + tq;
            float *op = p.out + oo;
            unsigned short *oph = reinterpret_cast<unsigned short *>(p.out) + oo;
#pragma unroll
            for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (mt * 32 + rho(r, kk) < p.d) {
                        const float ov = qpad ? 0.f : (nancol ? NAN : O[mt][r] * inv);
                        if constexpr (KH) oph[(long long)(mt * 32 + rho(r, kk)) * T] = (unsigned short)(pack2_bf16(ov, 0.f) & 0xffffu);
                        else op[(long long)(mt * 32 + rho(r, kk)) * T] = ov;
                    }
        }
        return;
    }
    // ---- pass 1: column statistics
    float mx = -INFINITY, sum = 0.f;
    load_tile_b<HDP, true, false, KH>(Kp, T, p.d, 0, sK[0], nullptr, tid);
    load_tile_b<HDP, true, false, KH>(Kp, T, p.d, 32, sK[1], nullptr, tid);
    fill_key_bits(s_kb, mrow, p.T, ntile, tid);
    __syncthreads();
    f32x16 s;
    mma_tile_frag_b<HDP>(sK[0], qf, li, kk, s);
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        fetch_tile_b<HDP, KH>(Kp, T, p.d, 32 * (it + 2), tid, pk);
        const unsigned bad = it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane);
        f32x16 sn;
        mma_tile_frag_b<HDP>(sK[(it + 1) & 1], qf, li, kk, sn);
        float tm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (bad >> rho(r, kk)) & 1u ? -INFINITY : s[r] * p.scale;
            s[r] = v;
            tm = fmaxf(tm, v);
        }
        const float m2 = fmaxf(mx, tm);
        const float m2s = m2 > -INFINITY ? m2 : 0.f;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += __expf(s[r] - m2s);
        sum = sum * __expf(mx - m2s) + a;
        mx = m2;
        commit_tile_b<HDP, true, false, KH>(sK[it & 1], nullptr, tid, pk);
        s = sn;
    }
    {
        const float om = __shfl_xor(mx, 32, 64), os = __shfl_xor(sum, 32, 64);
        const float m2 = fmaxf(mx, om);
        sum = (mx > -INFINITY ? sum * __expf(mx - m2) : 0.f) + (om > -INFINITY ? os * __expf(om - m2) : 0.f);
        mx = m2;
    }
    const bool qpad = tq < p.T && mrow && mrow[tq];
    const float inv = 1.f / sum;
    if (tq < p.T && kk == 0) {
        p.stats[((long long)b * T + tq) * 2] = mx;
        p.stats[((long long)b * T + tq) * 2 + 1] = inv;
    }
    // ---- pass 2: probabilities and out = V P
    f32x16 O[HDP / 32];
#pragma unroll
    for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[mt][i] = 0.f;
    const bool nancol = !(sum > 0.f);
    const float mxs = nancol ? 0.f : mx;
    __syncthreads();
    load_tile_b<HDP, true, false, KH>(Kp, T, p.d, 0, sK[0], nullptr, tid);
    load_tile_b<HDP, true, false, KH>(Kp, T, p.d, 32, sK[1], nullptr, tid);
    load_tile_b<HDP, false, true, KH>(Vp, T, p.d, 0, nullptr, sV[0], tid);
    __syncthreads();
    mma_tile_frag_b<HDP>(sK[0], qf, li, kk, s);
    float *attp = ATT ? p.att + (long long)b * T * T + tq : nullptr;      // ATT: the (H N, T, T) tensor is written (its own instances)
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        fetch_tile_b<HDP, KH>(Kp, T, p.d, 32 * (it + 2), tid, pk);
        fetch_tile_b<HDP, KH>(Vp, T, p.d, 32 * (it + 1), tid, pv);
        const unsigned bad = it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane);
        f32x16 sn;
        mma_tile_frag_b<HDP>(sK[(it + 1) & 1], qf, li, kk, sn);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(s[r] * p.scale - mxs) * inv;
            s[r] = ((bad >> rho(r, kk)) & 1u) || qpad ? 0.f : e;
        }
        if (__builtin_amdgcn_ballot_w64(nancol && !qpad) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (nancol && !qpad) ? (32 * it + rho(r, kk) < p.T ? NAN : 0.f) : s[r];
        }
        if (attp && tq < p.T) {
            float *ap = attp + (long long)(32 * it + 4 * kk) * T;
            if (32 * it + 32 <= p.T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ap[(long long)rho(r, 0) * T] = s[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * it + rho(r, kk) < p.T) ap[(long long)rho(r, 0) * T] = s[r];
            }
        }
        bf16x8_t pb[2];
        pack_acc_b(s, pb);
        mma_tile_acc_b<HDP>(sV[it & 1], pb, li, kk, O);
        commit_tile_b<HDP, true, false, KH>(sK[it & 1], nullptr, tid, pk);
        commit_tile_b<HDP, false, true, KH>(nullptr, sV[(it + 1) & 1], tid, pv);
        s = sn;
    }
    if (tq < p.T) {
        float *op = p.out + ((long long)n * p.C + h * p.d) * T + tq;
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mt * 32 + rho(r, kk) < p.d) op[(long long)(mt * 32 + rho(r, kk)) * T] = O[mt][r];
    }
}

// GATT: a gradient arrives for the returned attention tensor too (modules.py:60 returns `att`; training recipes rarely differentiate it).
// The plain instances do not carry that path: 232 / 158 instead of 256 (13 spilled) / 190 registers - the query kernel runs three waves per SIMD.
template <int HDP, bool GATT, bool KH = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_bf16_kernel(AttnParams p) {
    using B = BTile<HDP>;
    __shared__ __attribute__((aligned(16))) unsigned short sQt[2][B::TT], sQd[2][B::TD], sGt[2][B::TT], sGd[2][B::TD];
    __shared__ __attribute__((aligned(16))) float sSt[2][32 * 4];
    __shared__ float sT[4][32 * TP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kk = lane >> 5;
    int tx, b;
    if (!attn_tile(p, tx, b)) return;
    const int h = b / p.N, n = b - h * p.N;
    const long long T = p.T;
    const long long Ko = ((long long)n * 3 * p.C + h * p.d) * T;
    const rsrc_t Kp = head_rsrc_e<KH>(p.kvq, Ko, p.d, T), Vp = head_rsrc_e<KH>(p.kvq, Ko + (long long)p.C * T, p.d, T),
                 Qp = head_rsrc_e<KH>(p.kvq, Ko + 2 * (long long)p.C * T, p.d, T);
    const rsrc_t Gp = head_rsrc_e<KH>(p.gout, ((long long)n * p.C + h * p.d) * T, p.d, T);
    const unsigned char *mrow = p.mask ? p.mask + (long long)n * T : nullptr;
    const int tk0 = tx * 128 + wave * 32, tk = tk0 + li;
    const bool kbad = tk >= p.T || (mrow && mrow[tk]);
    bf16x8_t kf[HDP / 16], vf[HDP / 16];
    load_frag_b<HDP, KH>(Kp, T, p.d, tk0, li, kk, kf);
    load_frag_b<HDP, KH>(Vp, T, p.d, tk0, li, kk, vf);
    f32x16 dK[HDP / 32], dV[HDP / 32];
#pragma unroll
    for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dK[mt][i] = 0.f, dV[mt][i] = 0.f;
    const int ntile = (p.T + 31) / 32;
    const float cexp = p.scale * 1.44269504088896341f;
    float pq[B::NP][8], pg[B::NP][8];
    // Per query row ONE exponent offset e0 = log2(e) max - log2(1 / sum), so that the probability is exp2(s c - e0) with c = scale log2(e)
    // - an FMA and a v_exp_f32 per element (round 5); a dead row (padding, or past T) has e0 = +inf: exp2(-inf) = 0, no select.
    // Round 6: the statistics of a query tile and the tile's Q / dOut elements are REQUESTED a whole iteration before they are written
    // to LDS (registers in between): requested at the top of the iteration that writes them they were waited for on the spot - a memory
    // latency per iteration in front of the barrier every wave of the workgroup stands at.
    float sr[3] = {0.f, 0.f, 0.f};
    int sflag = 0;                                               // 0: past T, 1: live row, 2: padded row
    auto stage_fetch = [&](int it) __attribute__((always_inline)) {
        sflag = 0;
        if (tid < 32) {
            const int t = 32 * it + tid;
            if (t < p.T) {
                sr[0] = p.stats[((long long)b * T + t) * 2], sr[1] = p.stats[((long long)b * T + t) * 2 + 1];
                sr[2] = p.delta[(long long)b * T + t];
                sflag = (mrow && mrow[t]) ? 2 : 1;
            }
        }
    };
    auto stage_commit = [&](int buf) __attribute__((always_inline)) {
        if (tid < 32) {
            f32x4_t st = {INFINITY, 0.f, 0.f, 0.f};
            if (sflag) {
                st[0] = sflag == 2 ? INFINITY : sr[0] * 1.44269504088896341f - __builtin_log2f(sr[1]);
                st[1] = sr[2];
            }
            *reinterpret_cast<f32x4_t *>(&sSt[buf][4 * tid]) = st;
        }
    };
    stage_fetch(0);
    stage_commit(0);
    load_tile_b<HDP, true, true, KH>(Qp, T, p.d, 0, sQt[0], sQd[0], tid);
    load_tile_b<HDP, true, true, KH>(Gp, T, p.d, 0, sGt[0], sGd[0], tid);
    if (ntile > 1) {
        stage_fetch(1);
        fetch_tile_b<HDP, KH>(Qp, T, p.d, 32, tid, pq);
        fetch_tile_b<HDP, KH>(Gp, T, p.d, 32, tid, pg);
    }
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        const float *tS = sSt[it & 1];
        f32x16 s, dp;
        mma_tile_frag_b<HDP>(sQt[it & 1], kf, li, kk, s);        // rows: queries of the tile, column: this lane's key
        mma_tile_frag_b<HDP>(sGt[it & 1], vf, li, kk, dp);
        if (it + 1 < ntile) {                                    // tile it + 1 (requested an iteration ago) -> LDS, tile it + 2 requested
            stage_commit((it + 1) & 1);
            commit_tile_b<HDP, true, true, KH>(sQt[(it + 1) & 1], sQd[(it + 1) & 1], tid, pq);
            commit_tile_b<HDP, true, true, KH>(sGt[(it + 1) & 1], sGd[(it + 1) & 1], tid, pg);
            if (it + 2 < ntile) {
                stage_fetch(it + 2);
                fetch_tile_b<HDP, KH>(Qp, T, p.d, 32 * (it + 2), tid, pq);
                fetch_tile_b<HDP, KH>(Gp, T, p.d, 32 * (it + 2), tid, pg);
            }
        }
        if constexpr (GATT) {
            float *tt = sT[wave];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = 2 * u + kk, k2 = tk0 + r, q2 = 32 * it + li;
                tt[r * TP + li] = (k2 < p.T && q2 < p.T) ? p.gatt[((long long)b * T + k2) * T + q2] : 0.f;
            }
            __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] += tt[li * TP + rho(r, kk)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x2_v st = *reinterpret_cast<const f32x2_v *>(&tS[4 * rho(r, kk)]);
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], cexp, -st[0]));     // (a dead KEY - this lane's column - is cleared at the end)
            s[r] = pr;
            dp[r] = pr * (dp[r] - st[1]);                                                  // (the factor `scale` of the score gradient: on dK, at the end)
        }
        bf16x8_t pb[2], db[2];
        pack_acc_b(s, pb);
        pack_acc_b(dp, db);
        mma_tile_acc_b<HDP>(sGd[it & 1], pb, li, kk, dV);
        mma_tile_acc_b<HDP>(sQd[it & 1], db, li, kk, dK);
    }
    if (tk < p.T) {
        const long long go = ((long long)n * 3 * p.C + h * p.d) * T + tk;
        float *gk = p.gkvq + go, *gv = gk + (long long)p.C * T;
        unsigned short *gkh = reinterpret_cast<unsigned short *>(p.gkvq) + go, *gvh = gkh + (long long)p.C * T;
        const float ks = kbad ? 0.f : p.scale, vs = kbad ? 0.f : 1.f;          // a padded key receives no gradient
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mt * 32 + rho(r, kk) < p.d) {
                    const float a = kbad ? 0.f : dK[mt][r] * ks, c = kbad ? 0.f : dV[mt][r] * vs;
                    if constexpr (KH) {
                        gkh[(long long)(mt * 32 + rho(r, kk)) * T] = (unsigned short)(pack2_bf16(a, 0.f) & 0xffffu);
                        gvh[(long long)(mt * 32 + rho(r, kk)) * T] = (unsigned short)(pack2_bf16(c, 0.f) & 0xffffu);
                    } else {
                        gk[(long long)(mt * 32 + rho(r, kk)) * T] = a;
                        gv[(long long)(mt * 32 + rho(r, kk)) * T] = c;
                    }
                }
    }
}

template <int HDP, bool GATT, bool KH = false>
__global__ __launch_bounds__(256, GATT ? 2 : 3) void attn_bwd_q_bf16_kernel(AttnParams p) {
    using B = BTile<HDP>;
    __shared__ __attribute__((aligned(16))) unsigned short sKt[2][B::TT], sKd[2][B::TD], sVt[2][B::TT];
    __shared__ unsigned s_kb[KBITS_MAX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, kk = lane >> 5;
    int tx, b;
    if (!attn_tile(p, tx, b)) return;
    const int h = b / p.N, n = b - h * p.N;
    const long long T = p.T;
    const long long Ko = ((long long)n * 3 * p.C + h * p.d) * T;
    const rsrc_t Kp = head_rsrc_e<KH>(p.kvq, Ko, p.d, T), Vp = head_rsrc_e<KH>(p.kvq, Ko + (long long)p.C * T, p.d, T),
                 Qp = head_rsrc_e<KH>(p.kvq, Ko + 2 * (long long)p.C * T, p.d, T);
    const rsrc_t Gp = head_rsrc_e<KH>(p.gout, ((long long)n * p.C + h * p.d) * T, p.d, T);
    const unsigned char *mrow = p.mask ? p.mask + (long long)n * T : nullptr;
    const int tq0 = tx * 128 + wave * 32, tq = tq0 + li;
    bf16x8_t qf[HDP / 16], gf[HDP / 16];
    load_frag_b<HDP, KH>(Qp, T, p.d, tq0, li, kk, qf);
    load_frag_b<HDP, KH>(Gp, T, p.d, tq0, li, kk, gf);
    const bool qdead = tq >= p.T || (mrow && mrow[tq]);
    const float mx = tq < p.T ? p.stats[((long long)b * T + tq) * 2] : 0.f, inv = tq < p.T ? p.stats[((long long)b * T + tq) * 2 + 1] : 1.f;
    // delta[tq] = sum_dd out[dd][tq] gout[dd][tq], the softmax-backward column term.  With the operands stored as bf16 (KH) this kernel forms it
    // itself from the fragments it holds anyway - the same bf16 values attn_delta_kernel would read -, writes it for the key / value kernel
    // (launched BEHIND this one then) and the separate pass over out and gout (23 us in front of both kernels at 32 x 1292) is gone (round 6).
    float dl;
    if constexpr (KH && !GATT) {
        bf16x8_t of[HDP / 16];
        load_frag_b<HDP, KH>(head_rsrc_e<KH>(p.out, ((long long)n * p.C + h * p.d) * T, p.d, T), T, p.d, tq0, li, kk, of);
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < HDP / 16; ++s) {
            const uint4 ow = __builtin_bit_cast(uint4, of[s]), gw = __builtin_bit_cast(uint4, gf[s]);
            const unsigned o4[4] = {ow.x, ow.y, ow.z, ow.w}, g4[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                part = __builtin_fmaf(__builtin_bit_cast(float, o4[e] << 16), __builtin_bit_cast(float, g4[e] << 16), part);
                part = __builtin_fmaf(__builtin_bit_cast(float, o4[e] & 0xffff0000u), __builtin_bit_cast(float, g4[e] & 0xffff0000u), part);
            }
        }
        dl = part + __shfl_xor(part, 32, 64);                  // the head's channels sit in the two half-waves
        if (tq < p.T && kk == 0) const_cast<float *>(p.delta)[(long long)b * T + tq] = dl;
    } else {
        dl = tq < p.T ? p.delta[(long long)b * T + tq] : 0.f;
    }
    // (as in attn_bwd_kv_bf16_kernel: probability = exp2(s c - e0); the factor `scale` and a dead query are applied to dQ at the end)
    const float cexp = p.scale * 1.44269504088896341f, e0 = mx * 1.44269504088896341f - __builtin_log2f(inv);
    f32x16 dQ[HDP / 32];
#pragma unroll
    for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQ[mt][i] = 0.f;
    const int ntile = (p.T + 31) / 32;
    load_tile_b<HDP, true, true, KH>(Kp, T, p.d, 0, sKt[0], sKd[0], tid);
    load_tile_b<HDP, true, false, KH>(Vp, T, p.d, 0, sVt[0], nullptr, tid);
    fill_key_bits(s_kb, mrow, p.T, ntile, tid);
    float pk[B::NP][8], pv[B::NP][8];                           // tile it + 1, requested a whole iteration before it is written to LDS (round 6)
    if (ntile > 1) {
        fetch_tile_b<HDP, KH>(Kp, T, p.d, 32, tid, pk);
        fetch_tile_b<HDP, KH>(Vp, T, p.d, 32, tid, pv);
    }
    for (int it = 0; it < ntile; ++it) {
        __syncthreads();
        const unsigned bad = __builtin_amdgcn_readfirstlane(it < KBITS_MAX ? s_kb[it] : key_bits(mrow, p.T, 32 * it, lane));
        f32x16 s, dp;
        mma_tile_frag_b<HDP>(sKt[it & 1], qf, li, kk, s);
        mma_tile_frag_b<HDP>(sVt[it & 1], gf, li, kk, dp);
        if (it + 1 < ntile) {
            commit_tile_b<HDP, true, true, KH>(sKt[(it + 1) & 1], sKd[(it + 1) & 1], tid, pk);
            commit_tile_b<HDP, true, false, KH>(sVt[(it + 1) & 1], nullptr, tid, pv);
            if (it + 2 < ntile) {
                fetch_tile_b<HDP, KH>(Kp, T, p.d, 32 * (it + 2), tid, pk);
                fetch_tile_b<HDP, KH>(Vp, T, p.d, 32 * (it + 2), tid, pv);
            }
        }
        if (bad != 0) {                                              // (wave-uniform: a tile without a masked key skips the bit tests and selects)
            const unsigned badl = bad >> (4 * kk);                   // bit rho(r, 0) <-> register r
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (badl & (1u << rho(r, 0))) ? -INFINITY : s[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], cexp, -e0));     // (a masked key: exp2(-inf) = 0)
            float g = dp[r];
            if constexpr (GATT) {
                const bool kdead = (bad >> rho(r, kk)) & 1u;
                if (!(qdead || kdead)) g += p.gatt[((long long)b * T + 32 * it + rho(r, kk)) * T + tq];
            }
            dp[r] = pr * (g - dl);
        }
        bf16x8_t db[2];
        pack_acc_b(dp, db);
        mma_tile_acc_b<HDP>(sKd[it & 1], db, li, kk, dQ);
    }
    if (tq < p.T) {
        const long long go = ((long long)n * 3 * p.C + 2 * p.C + h * p.d) * T + tq;
        float *gq = p.gkvq + go;
        unsigned short *gqh = reinterpret_cast<unsigned short *>(p.gkvq) + go;
#pragma unroll
        for (int mt = 0; mt < HDP / 32; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mt * 32 + rho(r, kk) < p.d) {
                    const float a = qdead ? 0.f : dQ[mt][r] * p.scale;
                    if constexpr (KH) gqh[(long long)(mt * 32 + rho(r, kk)) * T] = (unsigned short)(pack2_bf16(a, 0.f) & 0xffffu);
                    else gq[(long long)(mt * 32 + rho(r, kk)) * T] = a;
                }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
static int gemm_launch(GemmParams &p, bool a_mcontig, bool b_ncontig, int gz, hipStream_t st, const char *what, bool bf16 = false) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || gz <= 0) return PSND_OK;
    long long tx = (p.N + GBN - 1) / GBN, ty = (p.M + GBM - 1) / GBM, tz = gz;
    if (p.flatT > 0) {           // one column axis over all batches
        if (!b_ncontig || p.zchunk > 0 || p.sAz != 0) PSND_FAIL(PSND_E_ARG, "%s: flat columns need a shared A and an n-contiguous B", what);
        tx = ((long long)p.Z * p.flatT + GBN - 1) / GBN, tz = 1;
    }
    // one-dimensional launch in the tile order of gemm_tile(): outer tiles padded to a multiple of 8 (one per XCD)
    const long long inner = p.zchunk > 0 ? tx * ty : ty, outer = p.zchunk > 0 ? tz : tx * tz;
    const long long total = (outer + 7) / 8 * 8 * inner;
    if (total > 0x7fffffff || tx > 0x7fffffff || ty > 65535) PSND_FAIL(PSND_E_SHAPE, "%s: grid too large", what);
    p.tiles_x = (int)tx, p.tiles_y = (int)ty, p.tiles_z = (int)tz;
    const dim3 grid((unsigned)total);
    if (bf16) {
        const char *abl = PSND_ENV("PSND_GEMM_ABLATE");
        p.ablate = abl ? atoi(abl) : 0;
        const bool masked = p.amask || p.bmask;
        const int dt = (p.a_h ? 1 : 0) | (p.b_h ? 2 : 0) | (p.c_h ? 4 : 0);
        if (dt) {                                // bf16-stored hidden tensor of the feed-forward pair: the six instances that pair launches
            if (masked || (p.c_h && (p.addend || p.zchunk > 0))) PSND_FAIL(PSND_E_ARG, "%s: bf16 storage with an operand mask, or a bf16 output with an addend / slabs", what);
            if (!a_mcontig && b_ncontig && dt == 4) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, false, 4>), grid, dim3(256), 0, st, p);
            else if (!a_mcontig && b_ncontig && dt == 2) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, false, 2>), grid, dim3(256), 0, st, p);
            else if (a_mcontig && b_ncontig && dt == 4) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false, 4>), grid, dim3(256), 0, st, p);
            else if (a_mcontig && b_ncontig && dt == 2) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false, 2>), grid, dim3(256), 0, st, p);
            else if (!a_mcontig && !b_ncontig && dt == 2) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, false, 2>), grid, dim3(256), 0, st, p);
            else if (!a_mcontig && !b_ncontig && dt == 1) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, false, 1>), grid, dim3(256), 0, st, p);
            else PSND_FAIL(PSND_E_UNSUPPORTED, "%s: no instance for bf16 storage %d with this operand layout", what, dt);
        } else if (masked) {
            if (a_mcontig && b_ncontig) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, true>), grid, dim3(256), 0, st, p);
            else if (!a_mcontig && b_ncontig) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, true>), grid, dim3(256), 0, st, p);
            else if (!a_mcontig && !b_ncontig) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_bf16_kernel<true, false, true>), grid, dim3(256), 0, st, p);
        } else if (a_mcontig && b_ncontig) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false>), grid, dim3(256), 0, st, p);
        else if (!a_mcontig && b_ncontig) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, false>), grid, dim3(256), 0, st, p);
        else if (!a_mcontig && !b_ncontig) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, false, false>), grid, dim3(256), 0, st, p);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) PSND_FAIL(PSND_E_HIP, "%s: %s", what, hipGetErrorString(e2));
        return PSND_OK;
    }
    if (p.a_h || p.b_h || p.c_h) PSND_FAIL(PSND_E_ARG, "%s: bf16 storage comes with bf16 operands only", what);
    if (a_mcontig && b_ncontig) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, p);
    else if (!a_mcontig && b_ncontig) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, p);
    else if (!a_mcontig && !b_ncontig) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, p);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) PSND_FAIL(PSND_E_HIP, "%s: %s", what, hipGetErrorString(e_));
    return PSND_OK;
}

extern "C" int psnd_linear1x1_fwd_ex(const void *x, const float *w, const float *bias, int64_t N, int Cin, int Cout, int64_t T, int relu, int bf16,
                                     int io_h, int64_t ld_h, void *y, void *stream);
extern "C" int psnd_linear1x1_fwd(const float *x, const float *w, const float *bias, int64_t N, int Cin, int Cout, int64_t T, int relu, int bf16,
                                  float *y, void *stream) {
    return psnd_linear1x1_fwd_ex(x, w, bias, N, Cin, Cout, T, relu, bf16, 0, 0, y, stream);
}
// io_h (bf16 != 0 only): 1 = x is STORED as bf16, 2 = y is - the hidden tensor of Conv1d -> ReLU -> Conv1d under autocast: the products take
// bf16 operands either way, so the values multiplied are the same and the tensor moves at half the bytes (as torch.autocast's own conv
// output would be).  One of the two per call.  ld_h: the row pitch (elements) of that bf16 tensor, (N, C, ld_h) in memory with T <= ld_h
// frames used (0: T) - it is the caller's own tensor, and rows that start on 128-byte lines (ld_h a multiple of 64) spare the GEMMs the
// straddling loads and stores of 1292-frame rows (-13..20 % at 32 x 1292: tools/r06/time_gemm_h.py).
extern "C" int psnd_linear1x1_fwd_ex(const void *xv, const float *w, const float *bias, int64_t N, int Cin, int Cout, int64_t T, int relu, int bf16,
                                     int io_h, int64_t ld_h, void *yv, void *stream) {
    const float *x = static_cast<const float *>(xv);
    float *y = static_cast<float *>(yv);
    if (!x || !w || !y) PSND_FAIL(PSND_E_ARG, "linear1x1_fwd: null pointer");
    if (io_h && (!bf16 || (io_h != 1 && io_h != 2))) PSND_FAIL(PSND_E_ARG, "linear1x1_fwd: io_h=%d (1 or 2, with bf16 operands)", io_h);
    if (N < 0 || Cin <= 0 || Cout <= 0 || T <= 0 || T >= ((int64_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "linear1x1_fwd: N=%lld Cin=%d Cout=%d T=%lld", (long long)N, Cin, Cout, (long long)T);
    GemmParams p = {};
    p.A = w, p.B = x, p.C = y, p.bias = bias, p.amask = nullptr, p.bmask = nullptr;
    p.M = Cout, p.N = (int)T, p.K = Cin, p.Z = (int)N;
    p.sAm = Cin, p.sAk = 1, p.sAz = 0, p.sBk = T, p.sBn = 1, p.sBz = (long long)Cin * T, p.sCm = T, p.sCz = (long long)Cout * T;
    p.relu = relu, p.zchunk = 0, p.sCslab = 0;
    p.flatT = (T >= 4 && N * T < ((int64_t)1 << 31)) ? (int)T : 0;
    p.b_h = io_h & 1, p.c_h = (io_h & 2) != 0;
    if (ld_h && (!io_h || ld_h < T)) PSND_FAIL(PSND_E_ARG, "linear1x1_fwd: ld_h=%lld (a bf16 tensor's row pitch, >= T)", (long long)ld_h);
    if (ld_h && io_h == 1) p.sBk = ld_h, p.sBz = (long long)Cin * ld_h;
    if (ld_h && io_h == 2) p.sCm = ld_h, p.sCz = (long long)Cout * ld_h;
    return gemm_launch(p, false, true, (int)N, static_cast<hipStream_t>(stream), "linear1x1_fwd", bf16 != 0);
}

// weight-gradient launch of psnd_linear1x1_bwd: batch chunks x K parts (frames of a clip) so that there are ~2 workgroups per CU
static void wgrad_split(int64_t N, int Cin, int Cout, int64_t T, int64_t *zslabs, int *ksplit, int *kpart) {
    const int64_t tiles = (int64_t)((Cout + GBM - 1) / GBM) * ((Cin + GBN - 1) / GBN);
    int64_t target = 512;                        // (round 6: 768 -> 512 workgroups - half the slabs to write and add up: config-4 step 1.776 -> 1.764 ms; 256 the same, 1536 1.836)
    if (const char *e = PSND_ENV("PSND_WGRAD_WGS")) target = atoi(e) > 0 ? atoi(e) : target;      // lab A/B: workgroups the launch aims at
    int64_t want = target / (tiles > 0 ? tiles : 1);
    if (want < 1) want = 1;
    int64_t zs = want > N ? N : want;
    const int64_t chunk = (N + zs - 1) / zs;
    zs = (N + chunk - 1) / chunk;
    int64_t ks = 1;
    if (zs == N && want > N) {                   // one clip per chunk already: cut the frames of a clip, parts of >= 256 frames
        ks = (want + N - 1) / N;
        const int64_t cap = T / 256 > 1 ? T / 256 : 1;
        if (ks > cap) ks = cap;
        if (PSND_ENV("PSND_LINEAR_NO_KSPLIT")) ks = 1;
    }
    int64_t kp = ((T + ks - 1) / ks + 31) / 32 * 32;
    ks = (T + kp - 1) / kp;
    *zslabs = zs, *ksplit = (int)ks, *kpart = (int)kp;
}

extern "C" int64_t psnd_linear1x1_wgrad_slabs(int64_t N, int Cin, int Cout, int64_t T) {
    if (N <= 0) return 0;
    int64_t zs;
    int ks, kp;
    wgrad_split(N, Cin, Cout, T, &zs, &ks, &kp);
    return zs * ks;
}

// gx = W^T gy' (gy' = gy where ymask > 0 when ymask is given), gw = sum gy' x^T (slabs in `gw_part`, summed into gw), gbias = sum gy'
extern "C" int psnd_linear1x1_bwd_ex(const void *gy, const float *ymask, const void *x, const float *w, int64_t N, int Cin, int Cout, int64_t T,
                                     int bf16, int io_h, int64_t ld_h, const float *gx_addend, const void *gx_mask, void *gx, float *gw, float *gw_part,
                                     float *gbias, void *stream);
extern "C" int psnd_linear1x1_bwd(const float *gy, const float *ymask, const float *x, const float *w, int64_t N, int Cin, int Cout, int64_t T,
                                  int bf16, float *gx, float *gw, float *gw_part, float *gbias, void *stream) {
    return psnd_linear1x1_bwd_ex(gy, ymask, x, w, N, Cin, Cout, T, bf16, 0, 0, nullptr, nullptr, gx, gw, gw_part, gbias, stream);
}
// ... with gx = W^T gy' + gx_addend (N, Cin, T): the gradient that reaches x along another branch (a residual connection) rides in the
// GEMM's epilogue instead of a separate accumulation pass over both tensors
extern "C" int psnd_linear1x1_bwd_acc(const float *gy, const float *ymask, const float *x, const float *w, int64_t N, int Cin, int Cout, int64_t T,
                                      int bf16, const float *gx_addend, float *gx, float *gw, float *gw_part, float *gbias, void *stream) {
    return psnd_linear1x1_bwd_ex(gy, ymask, x, w, N, Cin, Cout, T, bf16, 0, 0, gx_addend, nullptr, gx, gw, gw_part, gbias, stream);
}
// ... or with gx = (gx_mask > 0) ? W^T gy' : 0, gx_mask (N, Cin, T): x is the output of a ReLU (gx_mask = x itself, or the ReLU's output
// wherever it is kept) and gx is wanted for the ReLU's INPUT - the layer before the ReLU then takes gx as it is (ymask = null there: its two
// GEMMs and its bias sum read one tensor instead of two).  Not together with gx_addend.
// io_h (bf16 != 0 only; see psnd_linear1x1_fwd_ex): 1 = gy is STORED as bf16 (no ymask then), 2 = x, gx_mask and gx are (no gx_addend then).
// ld_h: the row pitch of the bf16-stored tensors (psnd_linear1x1_fwd_ex), 0 = T.
extern "C" int psnd_linear1x1_bwd_ex(const void *gyv, const float *ymask, const void *xv, const float *w, int64_t N, int Cin, int Cout, int64_t T,
                                     int bf16, int io_h, int64_t ld_h, const float *gx_addend, const void *gx_maskv, void *gxv, float *gw, float *gw_part,
                                     float *gbias, void *stream) {
    const float *gy = static_cast<const float *>(gyv), *x = static_cast<const float *>(xv), *gx_mask = static_cast<const float *>(gx_maskv);
    float *gx = static_cast<float *>(gxv);
    if (gx_addend && gx_mask) PSND_FAIL(PSND_E_ARG, "linear1x1_bwd: gx_addend and gx_mask exclude each other");
    if (io_h && (!bf16 || (io_h != 1 && io_h != 2))) PSND_FAIL(PSND_E_ARG, "linear1x1_bwd: io_h=%d (1 or 2, with bf16 operands)", io_h);
    if ((io_h == 1 && ymask) || (io_h == 2 && gx_addend)) PSND_FAIL(PSND_E_ARG, "linear1x1_bwd: bf16 storage with an operand mask / addend");
    if (ld_h && (!io_h || ld_h < T)) PSND_FAIL(PSND_E_ARG, "linear1x1_bwd: ld_h=%lld (a bf16 tensor's row pitch, >= T)", (long long)ld_h);
    const int64_t ld = ld_h ? ld_h : T;
    if (!gy || !x || !w) PSND_FAIL(PSND_E_ARG, "linear1x1_bwd: null pointer");
    if (N < 0 || Cin <= 0 || Cout <= 0 || T <= 0 || T >= ((int64_t)1 << 31)) PSND_FAIL(PSND_E_SHAPE, "linear1x1_bwd: bad shape");
    if (gw && !gw_part) PSND_FAIL(PSND_E_ARG, "linear1x1_bwd: gw needs the slab buffer gw_part (psnd_linear1x1_wgrad_slabs x Cout x Cin floats)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (N == 0) return PSND_OK;
    int rc = PSND_OK;
    if (gx) {
        GemmParams p = {};
        p.A = w, p.B = gy, p.C = gx, p.bias = nullptr, p.amask = nullptr, p.bmask = ymask, p.addend = gx_addend, p.omask = gx_mask;
        p.b_h = io_h & 1, p.c_h = (io_h & 2) != 0;
        p.M = Cin, p.N = (int)T, p.K = Cout, p.Z = (int)N;
        p.sAm = 1, p.sAk = Cin, p.sAz = 0, p.sBk = T, p.sBn = 1, p.sBz = (long long)Cout * T, p.sCm = T, p.sCz = (long long)Cin * T;
        p.flatT = (T >= 4 && N * T < ((int64_t)1 << 31)) ? (int)T : 0;
        if (io_h == 1) p.sBk = ld, p.sBz = (long long)Cout * ld;
        if (io_h == 2) p.sCm = ld, p.sCz = (long long)Cin * ld;
        rc = gemm_launch(p, true, true, (int)N, st, "linear1x1_bwd(data)", bf16 != 0);
        if (rc != PSND_OK) return rc;
    }
    if (gw) {
        int64_t zslabs;
        int ks, kp;
        wgrad_split(N, Cin, Cout, T, &zslabs, &ks, &kp);
        const int64_t slabs = zslabs * ks;
        GemmParams p = {};
        p.A = gy, p.B = x, p.C = gw_part, p.bias = nullptr, p.amask = ymask, p.bmask = nullptr;
        p.a_h = io_h & 1, p.b_h = (io_h & 2) != 0;
        p.M = Cout, p.N = Cin, p.K = (int)T, p.Z = (int)N;
        p.sAm = T, p.sAk = 1, p.sAz = (long long)Cout * T, p.sBk = 1, p.sBn = T, p.sBz = (long long)Cin * T, p.sCm = Cin, p.sCz = 0;
        p.zchunk = (int)((N + zslabs - 1) / zslabs), p.sCslab = (long long)Cout * Cin, p.ksplit = ks, p.kpart = kp;
        if (io_h == 1) p.sAm = ld, p.sAz = (long long)Cout * ld;
        if (io_h == 2) p.sBn = ld, p.sBz = (long long)Cin * ld;
        rc = gemm_launch(p, false, false, (int)slabs, st, "linear1x1_bwd(weight)", bf16 != 0);
        if (rc != PSND_OK) return rc;
        const long long n = (long long)Cout * Cin;
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, gw_part, (int)slabs, n, gw);
        PSND_CHECK_LAUNCH("linear1x1_bwd(slab sum)");
    }
    if (gbias && (io_h & 1)) {
        hipLaunchKernelGGL(rowsum_h_kernel, dim3(Cout), dim3(512), 0, st, reinterpret_cast<const unsigned short *>(gy), (int)N, Cout, (long long)T, (long long)ld,
                           gbias);
        PSND_CHECK_LAUNCH("linear1x1_bwd(bias)");
    } else if (gbias) {
        const bool vec = T % 4 == 0 && (reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(ymask)) % 16 == 0;
        if (vec && ymask) hipLaunchKernelGGL(rowsum4_kernel<true>, dim3(Cout), dim3(512), 0, st, gy, ymask, (int)N, Cout, (int)(T / 4), gbias);
        else if (vec) hipLaunchKernelGGL(rowsum4_kernel<false>, dim3(Cout), dim3(512), 0, st, gy, gy, (int)N, Cout, (int)(T / 4), gbias);
        else if (ymask) hipLaunchKernelGGL(rowsum_kernel<true>, dim3(Cout), dim3(1024), 0, st, gy, ymask, (int)N, Cout, (long long)T, gbias);
        else hipLaunchKernelGGL(rowsum_kernel<false>, dim3(Cout), dim3(1024), 0, st, gy, ymask, (int)N, Cout, (long long)T, gbias);
        PSND_CHECK_LAUNCH("linear1x1_bwd(bias)");
    }
    return PSND_OK;
}

// head dimension padded to 32 / 64 / 128 (round 5: 128 - MultiHeadAttention(256, 2), modules.py:24-27 takes any hidden_dim / num_head)
#define PSND_ATTN_LAUNCH(kern_, flag_, st_)                                                                          \
    do {                                                                                                            \
        if (p.d <= 32) hipLaunchKernelGGL((kern_<32, flag_>), grid, dim3(256), 0, st_, p);                          \
        else if (p.d <= 64) hipLaunchKernelGGL((kern_<64, flag_>), grid, dim3(256), 0, st_, p);                     \
        else hipLaunchKernelGGL((kern_<128, flag_>), grid, dim3(256), 0, st_, p);                                   \
    } while (0)

// ... the instances that take kvq (and write gkvq) STORED as bf16 (bf16 = 2): without the attention tensor / its gradient only
#define PSND_ATTN_LAUNCH_KH(kern_, st_)                                                                              \
    do {                                                                                                            \
        if (p.d <= 32) hipLaunchKernelGGL((kern_<32, false, true>), grid, dim3(256), 0, st_, p);                    \
        else if (p.d <= 64) hipLaunchKernelGGL((kern_<64, false, true>), grid, dim3(256), 0, st_, p);               \
        else hipLaunchKernelGGL((kern_<128, false, true>), grid, dim3(256), 0, st_, p);                             \
    } while (0)

static int mha_check(const char *what, int64_t N, int H, int C, int64_t T) {
    if (N <= 0 || H <= 0 || C % H != 0 || C / H > 128) PSND_FAIL(PSND_E_UNSUPPORTED, "%s: hidden_dim %d / heads %d: head dimensions up to 128 only", what, C, H);
    if (T <= 0 || T >= ((int64_t)1 << 24) || (int64_t)H * N > 65535) PSND_FAIL(PSND_E_SHAPE, "%s: T=%lld, H*N=%lld", what, (long long)T, (long long)(H * N));
    if (((int64_t)H * N + 7) / 8 * 8 * ((T + 127) / 128) > 0x7fffffff) PSND_FAIL(PSND_E_SHAPE, "%s: T=%lld, H*N=%lld: grid too large", what, (long long)T, (long long)(H * N));
    return PSND_OK;
}

extern "C" int psnd_mha_fwd(const float *kvq, const unsigned char *mask, int64_t N, int H, int C, int64_t T, float *out, float *att, float *stats,
                            int bf16, void *stream) {
    if (!kvq || !out || !stats) PSND_FAIL(PSND_E_ARG, "mha_fwd: null pointer");
    int rc = mha_check("mha_fwd", N, H, C, T);
    if (rc != PSND_OK) return rc;
    AttnParams p = {};
    p.kvq = kvq, p.mask = mask, p.out = out, p.att = att, p.stats = stats;
    p.N = (int)N, p.H = H, p.C = C, p.T = (int)T, p.d = C / H, p.scale = 1.f / __builtin_sqrtf((float)(C / H));
    p.tiles = (int)((T + 127) / 128);
    const dim3 grid((unsigned)((H * N + 7) / 8 * 8 * p.tiles));
    if (bf16 == 2) {
        if (att) PSND_FAIL(PSND_E_UNSUPPORTED, "mha_fwd: kvq stored as bf16 (bf16 = 2) comes without the attention tensor");
        PSND_ATTN_LAUNCH_KH(attn_fwd_bf16_kernel, static_cast<hipStream_t>(stream));
    } else if (bf16) {
        if (att) {
            PSND_ATTN_LAUNCH(attn_fwd_bf16_kernel, true, static_cast<hipStream_t>(stream));
        } else PSND_ATTN_LAUNCH(attn_fwd_bf16_kernel, false, static_cast<hipStream_t>(stream));
    } else if (att) {
        PSND_ATTN_LAUNCH(attn_fwd_kernel, true, static_cast<hipStream_t>(stream));
    } else PSND_ATTN_LAUNCH(attn_fwd_kernel, false, static_cast<hipStream_t>(stream));
    PSND_CHECK_LAUNCH("mha_fwd");
    return PSND_OK;
}

// parts: 1 = delta (the per-query-column dot products both gradient kernels read), 2 = key / value gradients, 4 = query gradients; the two
// gradient kernels write disjoint rows of gkvq and need only `delta`, so a caller may enqueue them on two streams
extern "C" int psnd_mha_bwd_parts(const float *kvq, const unsigned char *mask, const float *out, const float *att, const float *stats,
                                  const float *gout, const float *gatt, int64_t N, int H, int C, int64_t T, float *delta, float *gkvq, int bf16,
                                  int parts, void *stream) {
    if (!kvq || !out || !stats || !gout || !delta || !gkvq) PSND_FAIL(PSND_E_ARG, "mha_bwd: null pointer");
    if (parts < 1 || parts > 7) PSND_FAIL(PSND_E_ARG, "mha_bwd: parts=%d", parts);
    if (gatt && !att) PSND_FAIL(PSND_E_ARG, "mha_bwd: a gradient for `att` needs the att tensor of the forward pass");
    int rc = mha_check("mha_bwd", N, H, C, T);
    if (rc != PSND_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fold_delta = bf16 == 2 && parts == 7 && !gatt;      // the query kernel forms and writes delta itself, and runs first
    if (bf16 == 2 && parts != 7) PSND_FAIL(PSND_E_UNSUPPORTED, "mha_bwd: bf16 = 2 runs as a whole (parts = 7)");
    if ((parts & 1) && !fold_delta) {
        if (bf16 == 2)
            hipLaunchKernelGGL(attn_delta_kernel<unsigned short>, dim3((unsigned)((T + 255) / 256), (unsigned)(H * N)), dim3(256), 0, st,
                               reinterpret_cast<const unsigned short *>(out), reinterpret_cast<const unsigned short *>(gout), att, gatt, (int)N, H, C, C / H,
                               (long long)T, delta);
        else
            hipLaunchKernelGGL(attn_delta_kernel<float>, dim3((unsigned)((T + 255) / 256), (unsigned)(H * N)), dim3(256), 0, st, out, gout, att, gatt, (int)N, H, C,
                               C / H, (long long)T, delta);
        PSND_CHECK_LAUNCH("mha_bwd(delta)");
    }
    AttnParams p = {};
    p.kvq = kvq, p.mask = mask, p.stats = const_cast<float *>(stats), p.gout = gout, p.gatt = gatt, p.delta = delta, p.gkvq = gkvq;
    p.out = const_cast<float *>(out);
    p.N = (int)N, p.H = H, p.C = C, p.T = (int)T, p.d = C / H, p.scale = 1.f / __builtin_sqrtf((float)(C / H));
    p.tiles = (int)((T + 127) / 128);
    const dim3 grid((unsigned)((H * N + 7) / 8 * 8 * p.tiles));
    if (bf16 == 2 && (att || gatt)) PSND_FAIL(PSND_E_UNSUPPORTED, "mha_bwd: kvq stored as bf16 (bf16 = 2) comes without the attention tensor and its gradient");
    if (fold_delta) {
        PSND_ATTN_LAUNCH_KH(attn_bwd_q_bf16_kernel, st);
        PSND_CHECK_LAUNCH("mha_bwd(q)");
    }
    if (!(parts & 2)) {
    } else if (bf16 == 2) {
        PSND_ATTN_LAUNCH_KH(attn_bwd_kv_bf16_kernel, st);
    } else if (bf16) {
        if (p.gatt) {
            PSND_ATTN_LAUNCH(attn_bwd_kv_bf16_kernel, true, st);
        } else PSND_ATTN_LAUNCH(attn_bwd_kv_bf16_kernel, false, st);
    } else if (p.gatt) {
        PSND_ATTN_LAUNCH(attn_bwd_kv_kernel, true, st);
    } else PSND_ATTN_LAUNCH(attn_bwd_kv_kernel, false, st);
    PSND_CHECK_LAUNCH("mha_bwd(kv)");
    if (!(parts & 4) || fold_delta) {
    } else if (bf16 == 2) {
        PSND_ATTN_LAUNCH_KH(attn_bwd_q_bf16_kernel, st);
    } else if (bf16) {
        if (p.gatt) {
            PSND_ATTN_LAUNCH(attn_bwd_q_bf16_kernel, true, st);
        } else PSND_ATTN_LAUNCH(attn_bwd_q_bf16_kernel, false, st);
    } else if (p.gatt) {
        PSND_ATTN_LAUNCH(attn_bwd_q_kernel, true, st);
    } else PSND_ATTN_LAUNCH(attn_bwd_q_kernel, false, st);
    PSND_CHECK_LAUNCH("mha_bwd(q)");
    return PSND_OK;
}

extern "C" int psnd_mha_bwd(const float *kvq, const unsigned char *mask, const float *out, const float *att, const float *stats, const float *gout,
                            const float *gatt, int64_t N, int H, int C, int64_t T, float *delta, float *gkvq, int bf16, void *stream) {
    return psnd_mha_bwd_parts(kvq, mask, out, att, stats, gout, gatt, N, H, C, T, delta, gkvq, bf16, 7, stream);
}
