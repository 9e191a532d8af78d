// Exact-fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32): C[m,n] = sum_k A(m,k) * B(n,k) with a
// fused epilogue.  Used by the full-catalog scoring (recad/workflow/normal.py:57-93 semantics) and
// the NCF tower (recad/model/victim/ncf.py:41-53,122).
//
// * 128x128 tile per 4-wave workgroup, each wave 64x64 = 2x2 accumulators of 32x32.
// * k-chunks of 32 staged in padded LDS (row stride 33 floats: the ds_read_b32 of an MFMA operand
//   is conflict-free); the next chunk's global loads are issued (16-byte loads when the layout
//   allows) before the current chunk's 64 MFMAs and written to LDS afterwards (register
//   prefetch; 33 KiB of LDS => 3 workgroups per CU overlap each other's barriers and epilogues).
// * MFMA f32 is a k-ordered fmaf chain and chunks are visited in order, so a result is
//   bit-identical to  s = fmaf(a[k], b[k], s), k = 0..K-1  (the oracle's orc_score_rows).
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    int M, N, K;
    const float *A; long long a_rs, a_cs;  // A(m,k) = A[row(m)*a_rs + k*a_cs], row(m) = a_ridx ? a_ridx[m] : m
    const int *a_ridx;                     // optional row gather of A (and of row_bias): scoring a block of user ids
    int a_rmod, a_roff;                    // a_rmod > 0 (and a_ridx == NULL): row(m) = (m + a_roff) % a_rmod -- the full-catalog pair
                                           // list (user q / I, item q % I) of the NCF evaluation reads the item table in place
    const float *acc_init; int ld_init, init_base;  // optional: the accumulator of C(m, n) starts at acc_init[((m + a_roff) / a_rmod -
                                           // init_base) * ld_init + n] instead of 0: the k-ordered chain CONTINUES a prefix computed
                                           // once per user (layer 0 of the tower over [user | item]: the user half is shared)
    const float *B; long long b_rs, b_cs;  // B(n,k) = B[n*b_rs + k*b_cs]
    float *C; int ldc;
    const float *col_bias;                 // + col_bias[n]
    const float *row_bias;                 // scoring: ((s + row_bias[row(m)]) + col_bias[n]) + const_add
    float const_add;
    int relu;
    int sigmoid;                           // s = 1 / (1 + exp(-s))  (LightGCN.getUsersRating, lightgcn.py:119)
    unsigned drop_thresh24;                // > 0: nn.Dropout on the output (mf.py:47): keep element (m, n) iff its counter hash
    float drop_scale;                      //      is below keep_prob * 2^24, scaled by 1 / keep_prob
    unsigned long long drop_seed;
    const float *mask; int ldmask;         // keep s only where mask[m,n] > 0
    float *sk_part; long long sk_stride;   // split_k > 1 and sk_part: slice q STORES its partial at sk_part[q*sk_stride + m*ldc + n]
                                           // (the caller adds the slices in order: deterministic, no zeroing, no atomics)
    int split_k;                           // > 1: gridDim.y K-slices, each ADDS its partial into C (C pre-zeroed,
                                           // no bias/relu/mask; float atomics => summation order not fixed)
};

static constexpr int kGK = 32, kGLd = kGK + 1;
template <int BT> constexpr int gemm_lds_bytes(int nbuf) { return nbuf * 2 * BT * kGLd * (int)sizeof(float); }

// Tile rows [r0, r0+128) x k [k0, k0+32) of a strided matrix -> 16 floats per thread.
// mode 0: scalar, bounds-checked.  mode 1: k contiguous (cs==1), float4 along k.
// mode 2: rows contiguous (rs==1), float4 along rows.
template <int BT> struct TileRegs { float v[BT / 8]; };

template <int BT>
__device__ __forceinline__ int tile_mode(const float *src, int n_rows, int n_k, int r0, int k0, long long rs, long long cs,
                                         bool gathered = false)
{
    const bool full = (r0 + BT <= n_rows) && (k0 + kGK <= n_k);
    if (!full) return 0;
    if (cs == 1 && (rs & 3) == 0 && ((uintptr_t)src & 15) == 0) return 1;
    if (!gathered && rs == 1 && (cs & 3) == 0 && ((uintptr_t)src & 15) == 0) return 2;
    return 0;
}

template <int BT, bool GATHER = false>
__device__ __forceinline__ void tile_load(TileRegs<BT> &t, int mode, const float *__restrict__ src, int n_rows, int n_k, int r0,
                                          int k0, long long rs, long long cs, const int *__restrict__ ridx = nullptr, int rmod = 0,
                                          int roff = 0)
{
    auto row_of = [&](int m) -> long long { return ridx ? (long long)ridx[m] : (long long)((m + roff) % rmod); };
    const int tid = threadIdx.x;
    constexpr int NP = BT / 32;  // float4 loads per thread
    if (mode == 1) {  // 8 float4 per row of 32 k: thread -> (row = p*32 + tid/8, k4 = tid%8)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = p * 32 + (tid >> 3), c = (tid & 7) * 4;
            const long long gr = GATHER ? row_of(r0 + r) : r0 + r;
            const float4 x = *reinterpret_cast<const float4 *>(src + gr * rs + (k0 + c));
            t.v[p * 4 + 0] = x.x; t.v[p * 4 + 1] = x.y; t.v[p * 4 + 2] = x.z; t.v[p * 4 + 3] = x.w;
        }
    } else if (mode == 2) {  // BT/4 float4 per k column of BT rows: thread -> (k = p*KP + tid/(BT/4), row4 = tid%(BT/4))
        constexpr int RQ = BT / 4, KP = 256 / RQ;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = p * KP + tid / RQ, r = (tid % RQ) * 4;
            const float4 x = *reinterpret_cast<const float4 *>(src + (long long)(k0 + c) * cs + (r0 + r));
            t.v[p * 4 + 0] = x.x; t.v[p * 4 + 1] = x.y; t.v[p * 4 + 2] = x.z; t.v[p * 4 + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int p = 0; p < BT / 8; ++p) {
            const int idx = p * 256 + tid;
            const int r = idx / kGK, c = idx % kGK;
            const int gr = r0 + r, gc = k0 + c;
            t.v[p] = (gr < n_rows && gc < n_k) ? src[(GATHER ? row_of(gr) : (long long)gr) * rs + (long long)gc * cs] : 0.f;
        }
    }
}

template <int BT, int LD>
__device__ __forceinline__ void tile_store(const TileRegs<BT> &t, int mode, float (*dst)[LD])
{
    const int tid = threadIdx.x;
    constexpr int NP = BT / 32;
    if (mode == 1) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = p * 32 + (tid >> 3), c = (tid & 7) * 4;
            if (LD % 4 == 0) {  // rows are 16-byte aligned: one ds_write_b128
                *reinterpret_cast<float4 *>(&dst[r][c]) = make_float4(t.v[p * 4], t.v[p * 4 + 1], t.v[p * 4 + 2], t.v[p * 4 + 3]);
            } else {
                dst[r][c] = t.v[p * 4]; dst[r][c + 1] = t.v[p * 4 + 1]; dst[r][c + 2] = t.v[p * 4 + 2]; dst[r][c + 3] = t.v[p * 4 + 3];
            }
        }
    } else if (mode == 2) {
        constexpr int RQ = BT / 4, KP = 256 / RQ;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = p * KP + tid / RQ, r = (tid % RQ) * 4;
            dst[r][c] = t.v[p * 4]; dst[r + 1][c] = t.v[p * 4 + 1]; dst[r + 2][c] = t.v[p * 4 + 2]; dst[r + 3][c] = t.v[p * 4 + 3];
        }
    } else {
#pragma unroll
        for (int p = 0; p < BT / 8; ++p) {
            const int idx = p * 256 + tid;
            dst[idx / kGK][idx % kGK] = t.v[p];
        }
    }
}

// position `id` of the tile order -> tile origin.  Order (speed only): strips of 8 tile-rows,
// tile-row fastest, so consecutive tiles share the B tile and the strip's A tiles stay in L2.
template <int BT>
__device__ __forceinline__ void tile_origin(int id, int gx, int gy, int &m0, int &n0)
{
    const int strip = id / (8 * gx), rem = id % (8 * gx);
    const int h = min(8, gy - strip * 8);
    m0 = (strip * 8 + rem % h) * BT;
    n0 = (rem / h) * BT;
}

// BT = tile edge (128: each wave 2x2 accumulators of 32x32; 64: one accumulator per wave, for
// skinny problems that would not fill the chip with 128-tiles).
// GATHER: rows of A (and of row_bias) are taken through g.a_ridx (scoring a block of user ids).
// PLAIN: C = A.B^T with no bias / ReLU / mask / split-K (the LightGCN scoring GEMM) -- a separate
// instantiation because the general epilogue costs this one registers (scratch spills at 3 workgroups/CU).
template <int BT, int NBUF, int MINW, bool GATHER = false, bool PLAIN = false>
static __global__ __launch_bounds__(256, MINW) void gemm_f32_kernel(const GemmArgs g, const int tiles_per_block)
{
    constexpr int kGT = BT, TA = BT / 64, WS = BT / 2;  // TA accumulators per wave and dim, WS = wave sub-tile edge
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS: [buffer 0: A | B][buffer 1: A | B]
    auto tileA = [&](int buf) { return reinterpret_cast<float (*)[kGLd]>(smem + (size_t)buf * 2 * kGT * kGLd); };
    auto tileB = [&](int buf) { return reinterpret_cast<float (*)[kGLd]>(smem + (size_t)buf * 2 * kGT * kGLd + kGT * kGLd); };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int gx = (g.N + kGT - 1) / kGT, gy = (g.M + kGT - 1) / kGT, nwg = gx * gy;
    // Workgroups are dealt round-robin over the 8 XCDs: give every XCD a contiguous range of the
    // tile order (bijective remap), each workgroup a run of tiles_per_block consecutive tiles.
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb / 8, r = nb % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int t_begin = bid * tiles_per_block, t_end = min(nwg, t_begin + tiles_per_block);
    if (t_begin >= t_end) return;
    f32x16 acc[TA][TA];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TA; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    // K-slice of this workgroup (split_k > 1: blockIdx.y picks a contiguous range of k-chunks)
    const int all_chunks = (g.K + kGK - 1) / kGK;
    const int splits = g.split_k > 1 ? g.split_k : 1;
    const int per = (all_chunks + splits - 1) / splits;
    const int c_lo = (int)blockIdx.y * per, c_hi = min(all_chunks, c_lo + per);
    if (c_lo >= c_hi) return;
    const int n_chunks = c_hi - c_lo;
    const int total = (t_end - t_begin) * n_chunks;
    TileRegs<BT> ta, tb;
    int m0, n0;
    tile_origin<BT>(t_begin, gx, gy, m0, n0);
    int ma = tile_mode<BT>(g.A, g.M, g.K, m0, c_lo * kGK, g.a_rs, g.a_cs, GATHER), mb = tile_mode<BT>(g.B, g.N, g.K, n0, c_lo * kGK, g.b_rs, g.b_cs);
    tile_load<BT, GATHER>(ta, ma, g.A, g.M, g.K, m0, c_lo * kGK, g.a_rs, g.a_cs, g.a_ridx, g.a_rmod, g.a_roff);
    tile_load(tb, mb, g.B, g.N, g.K, n0, c_lo * kGK, g.b_rs, g.b_cs);
    tile_store(ta, ma, tileA(0));
    tile_store(tb, mb, tileB(0));
    __syncthreads();
    // ONE software pipeline over all (tile, k-chunk) pairs of this workgroup: the next pair's global
    // loads are in flight under the current chunk's MFMAs, and a finished tile's stores drain under
    // the next tile's MFMAs.
    for (int it = 0; it < total; ++it) {
        const int cur = (NBUF == 2) ? (it & 1) : 0, c = c_lo + it % n_chunks;
        const bool more = it + 1 < total;
        if (more) {
            const int c1 = c_lo + (it + 1) % n_chunks;
            int m1, n1;
            tile_origin<BT>(t_begin + (it + 1) / n_chunks, gx, gy, m1, n1);
            ma = tile_mode<BT>(g.A, g.M, g.K, m1, c1 * kGK, g.a_rs, g.a_cs, GATHER);
            mb = tile_mode<BT>(g.B, g.N, g.K, n1, c1 * kGK, g.b_rs, g.b_cs);
            tile_load<BT, GATHER>(ta, ma, g.A, g.M, g.K, m1, c1 * kGK, g.a_rs, g.a_cs, g.a_ridx, g.a_rmod, g.a_roff);
            tile_load(tb, mb, g.B, g.N, g.K, n1, c1 * kGK, g.b_rs, g.b_cs);
        }
        const int kc = min(kGK, g.K - c * kGK);
        if (GATHER && !PLAIN && g.acc_init && c == c_lo) {   // the tile's chains continue a per-user prefix
            int mi, ni;
            tile_origin<BT>(t_begin + it / n_chunks, gx, gy, mi, ni);
            // rows of a tile are consecutive pairs: one division for the tile, a carry test per row
            const int q0 = (mi + g.a_roff) / g.a_rmod, rem0 = (mi + g.a_roff) % g.a_rmod;
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = wr * WS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, m = mi + dm;
                    const int t = rem0 + dm;
                    const int q = q0 + (t >= g.a_rmod ? (g.a_rmod >= BT ? 1 : t / g.a_rmod) : 0) - g.init_base;
                    const float *src = g.acc_init + (size_t)q * g.ld_init;
#pragma unroll
                    for (int j = 0; j < TA; ++j) {
                        const int n = ni + wc * WS + j * 32 + lr;
                        if (m < g.M && n < g.N) acc[i][j][r] = src[n];
                    }
                }
        }
        float (*Ac)[kGLd] = tileA(cur);
        float (*Bc)[kGLd] = tileB(cur);
        for (int kk = 0; kk < kc; kk += 2) {
            float av[TA], bv[TA];
#pragma unroll
            for (int i = 0; i < TA; ++i) {
                av[i] = Ac[wr * WS + i * 32 + lr][kk + lk];
                bv[i] = Bc[wc * WS + i * 32 + lr][kk + lk];
            }
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < TA; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (c == c_hi - 1) {  // tile finished: epilogue, then clear the accumulators
            tile_origin<BT>(t_begin + it / n_chunks, gx, gy, m0, n0);
            // accumulator register r of a 32x32 block = row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31:
            // one store instruction writes two 128-byte row segments
            if (PLAIN) {
                const bool interior = (m0 + kGT <= g.M) && (n0 + kGT <= g.N);
                float *cbase = g.C + (size_t)(m0 + wr * WS + 4 * lk) * g.ldc + (n0 + wc * WS + lr);
                const size_t ld = (size_t)g.ldc;
                if (interior) {  // no per-element tests, pointer bumps only
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int j = 0; j < TA; ++j) {
                            float *cp = cbase + (size_t)(i * 32) * ld + j * 32;
#pragma unroll
                            for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * ld] = acc[i][j][r];
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int j = 0; j < TA; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                                if (m0 + wr * WS + 4 * lk + row < g.M && n0 + wc * WS + lr + j * 32 < g.N)
                                    cbase[(size_t)row * ld + j * 32] = acc[i][j][r];
                            }
                }
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int j = 0; j < TA; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            } else {
                // general form (biases / ReLU / mask / split-K / edge tiles).  Where the accumulators are cleared
                // is a measured choice (A/B on one box): inside the loop is +2 % for the NCF tower, afterwards is
                // +35 % for the biased scoring GEMM (MF, short K: fewer scratch spills).
                constexpr bool kClearInside = !GATHER;
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int j = 0; j < TA; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                            const int m = m0 + wr * WS + i * 32 + row, n = n0 + wc * WS + j * 32 + lr;
                            if (m < g.M && n < g.N) {
                                float s = acc[i][j][r];
                                if (splits > 1) {
                                    if (g.sk_part) g.sk_part[(size_t)blockIdx.y * g.sk_stride + (size_t)m * g.ldc + n] = s;
                                    else unsafeAtomicAdd(&g.C[(size_t)m * g.ldc + n], s);
                                    if (kClearInside) acc[i][j][r] = 0.f;
                                    continue;
                                }
                                if (g.row_bias) s = ((s + g.row_bias[(GATHER && g.a_ridx) ? g.a_ridx[m] : m]) + g.col_bias[n]) + g.const_add;
                                else if (g.col_bias) s += g.col_bias[n];
                                if (g.relu) s = s > 0.f ? s : 0.f;
                                if (g.sigmoid) s = 1.f / (1.f + expf(-s));
                                if (g.mask) s = g.mask[(size_t)m * g.ldmask + n] > 0.f ? s : 0.f;
                                if (g.drop_thresh24) s = rk_drop_keep(g.drop_seed, (unsigned)((size_t)m * g.N + n), g.drop_thresh24) ? s * g.drop_scale : 0.f;
                                g.C[(size_t)m * g.ldc + n] = s;
                            }
                            if (kClearInside) acc[i][j][r] = 0.f;
                        }
                if (!kClearInside) {
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int j = 0; j < TA; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
            }
        }
        if (NBUF == 1) __syncthreads();  // everyone is done reading the single buffer
        if (more) {
            tile_store(ta, ma, tileA(NBUF == 2 ? cur ^ 1 : 0));
            tile_store(tb, mb, tileB(NBUF == 2 ? cur ^ 1 : 0));
        }
        __syncthreads();
    }
}

// K-slices actually launched for a request of `splits`: every slice holds the same number of k-chunks
// (the last one possibly fewer) and none is empty.
inline int gemm_effective_splits(int K, int splits)
{
    if (splits <= 1) return 1;
    const int chunks = (K + kGK - 1) / kGK, per = (chunks + splits - 1) / splits;
    return (chunks + per - 1) / per;
}

// ---- 64x64-tile form on v_mfma_f32_16x16x4_f32, for problems with too few 128-tiles to fill the chip (the NCF
// tower at batch 1024).  Each wave owns a 32x32 sub-tile as 2x2 INDEPENDENT 16x16 accumulators: with one
// 32x32 accumulator per wave every MFMA waits for the previous one's result (measured: the MFMA phase alone ran
// at half rate -- 30 us of the 47 us layer-0 GEMM with loads and LDS stores switched off); four independent
// chains issue back to back at the same 256 flop/clk/CU.  Same k-ordered accumulation, same operand layouts
// (strided A/B, bounds-checked edges), same general epilogue (column bias, ReLU, mask, split-K by atomics or
// parked slices) as gemm_f32_kernel.  LDS rows are padded to 36 floats: the operand read of a k-step touches
// 16 rows x 4 k's = banks 4*row + k, each exactly twice (the minimum for 64 lanes), and rows stay 16-byte aligned.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment (rows of C with ldc % 4 != 0)
static constexpr int kSLd = kGK + 4;
static __global__ __launch_bounds__(256, 4) void gemm_f32_skinny_kernel(const GemmArgs g)
{
    constexpr int BT = 64;
    __shared__ __attribute__((aligned(16))) float sA[BT][kSLd];
    __shared__ __attribute__((aligned(16))) float sB[BT][kSLd];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int gx = (g.N + BT - 1) / BT;
    const int m0 = ((int)blockIdx.x / gx) * BT, n0 = ((int)blockIdx.x % gx) * BT;
    const int all_chunks = (g.K + kGK - 1) / kGK;
    const int splits = g.split_k > 1 ? g.split_k : 1;
    const int per = (all_chunks + splits - 1) / splits;
    const int c_lo = (int)blockIdx.y * per, c_hi = min(all_chunks, c_lo + per);
    if (c_lo >= c_hi) return;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l16 = lane & 15, lq = lane >> 4;
    TileRegs<BT> ta, tb;
    int ma = tile_mode<BT>(g.A, g.M, g.K, m0, c_lo * kGK, g.a_rs, g.a_cs), mb = tile_mode<BT>(g.B, g.N, g.K, n0, c_lo * kGK, g.b_rs, g.b_cs);
    tile_load(ta, ma, g.A, g.M, g.K, m0, c_lo * kGK, g.a_rs, g.a_cs);
    tile_load(tb, mb, g.B, g.N, g.K, n0, c_lo * kGK, g.b_rs, g.b_cs);
    tile_store(ta, ma, sA);
    tile_store(tb, mb, sB);
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
        const bool more = c + 1 < c_hi;
        if (more) {  // the next chunk's global loads fly under this chunk's MFMAs
            ma = tile_mode<BT>(g.A, g.M, g.K, m0, (c + 1) * kGK, g.a_rs, g.a_cs);
            mb = tile_mode<BT>(g.B, g.N, g.K, n0, (c + 1) * kGK, g.b_rs, g.b_cs);
            tile_load(ta, ma, g.A, g.M, g.K, m0, (c + 1) * kGK, g.a_rs, g.a_cs);
            tile_load(tb, mb, g.B, g.N, g.K, n0, (c + 1) * kGK, g.b_rs, g.b_cs);
        }
        // a partial last chunk is zero-filled by the bounds-checked loads, so whole k-steps of 4 are safe
        const int kc = (min(kGK, g.K - c * kGK) + 3) & ~3;
        auto k_step = [&](const int kk) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = sA[wr * 32 + i * 16 + l16][kk + lq];
                bv[i] = sB[wc * 32 + i * 16 + l16][kk + lq];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        };
        if (kc == kGK) {
#pragma unroll
            for (int kk = 0; kk < kGK; kk += 4) k_step(kk);
        } else {
            for (int kk = 0; kk < kc; kk += 4) k_step(kk);
        }
        __syncthreads();  // everyone is done reading the buffer
        if (more) {
            tile_store(ta, ma, sA);
            tile_store(tb, mb, sB);
        }
        __syncthreads();
    }
    // accumulator register r of a 16x16 block = row 4*(lane>>4) + r, column lane&15
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + 4 * lq + r, n = n0 + wc * 32 + j * 16 + l16;
                if (m < g.M && n < g.N) {
                    float s = acc[i][j][r];
                    if (splits > 1) {
                        if (g.sk_part) g.sk_part[(size_t)blockIdx.y * g.sk_stride + (size_t)m * g.ldc + n] = s;
                        else unsafeAtomicAdd(&g.C[(size_t)m * g.ldc + n], s);
                        continue;
                    }
                    if (g.row_bias) s = ((s + g.row_bias[m]) + g.col_bias[n]) + g.const_add;
                    else if (g.col_bias) s += g.col_bias[n];
                    if (g.relu) s = s > 0.f ? s : 0.f;
                    if (g.sigmoid) s = 1.f / (1.f + expf(-s));
                    if (g.mask) s = g.mask[(size_t)m * g.ldmask + n] > 0.f ? s : 0.f;
                    if (g.drop_thresh24) s = rk_drop_keep(g.drop_seed, (unsigned)((size_t)m * g.N + n), g.drop_thresh24) ? s * g.drop_scale : 0.f;
                    g.C[(size_t)m * g.ldc + n] = s;
                }
            }
}

// ---- the same 64x64 tile with k-chunks of 128 ("deep"): for problems with FEW tiles and LONG K (the whole-K forward
// GEMMs of the NCF tower at batch 1024: 128-256 workgroups, K = 1024-8192).  With 32-deep chunks such a workgroup is
// latency-bound: 1 024 MFMA cycles per chunk against ~1 us of global-load latency + two barriers, and one workgroup per
// CU has nobody to hide behind (measured: 70.5 us for 1024 x 1024 x 2048 = 64 chunks x 1.1 us).  128-deep chunks put
// 4 096 MFMA cycles behind every load / barrier pair, so the chunk loop runs at the MFMA rate.  Same k-ordered chain.
static constexpr int kDK = 128, kDLd = kDK + 4;
struct DeepRegs { f32x4 v[8]; };   // native vectors: stay in registers across the conditional prefetch

// operand form of a deep chunk: 1 = k contiguous (unit column stride), 2 = rows contiguous (unit row stride); 0 = neither.
// The deep kernel only takes whole 64 x 128 chunks of 16-byte aligned operands -- the launcher falls back to the
// 32-deep kernel (bounds-checked loads) for anything else.
inline int deep_form(const float *src, long long rs, long long cs)
{
    if (((uintptr_t)src & 15) != 0) return 0;
    if (cs == 1 && (rs & 3) == 0) return 1;
    if (rs == 1 && (cs & 3) == 0) return 2;
    return 0;
}
template <int MODE, int ROWS>
__device__ __forceinline__ void deep_load(DeepRegs &t, const float *__restrict__ src, int r0, int k0, long long rs, long long cs)
{
    const int tid = threadIdx.x;
    constexpr int NP = ROWS / 8, LPC = ROWS / 4;   // float4 loads per thread; lanes per k column (form 2)
    if (MODE == 1) {   // 32 float4 per row of 128 k: thread -> (row = p*8 + tid/32, k4 = tid%32)
        const float *q = src + (long long)(r0 + (tid >> 5)) * rs + (k0 + (tid & 31) * 4);
#pragma unroll
        for (int p = 0; p < NP; ++p) t.v[p] = *reinterpret_cast<const f32x4 *>(q + (long long)p * 8 * rs);
    } else {           // ROWS/4 float4 per k column: thread -> (k = p*(256/LPC) + tid/LPC, row4 = tid%LPC)
        const float *q = src + (long long)(k0 + tid / LPC) * cs + (r0 + (tid % LPC) * 4);
#pragma unroll
        for (int p = 0; p < NP; ++p) t.v[p] = *reinterpret_cast<const f32x4 *>(q + (long long)p * (256 / LPC) * cs);
    }
}
// LDS image of a chunk.  Form 1 (k contiguous): [row][k], rows padded to 132 floats -- an operand read of a k-step touches
// banks 4*row + k = 0..63, each once (measured: rotating k so that a half-wave covers 32 distinct banks instead is 5 % SLOWER).  Form 2 (rows contiguous): kept as it arrives, [k][row] with
// rows of ROWS+16 floats -- the b128 stores are conflict-free (the transposing scalar stores of the 32-deep kernel are 8-way
// conflicted) and an operand read touches banks 16*k + row, again every bank twice.
template <int MODE, int ROWS> constexpr int deep_floats() { return MODE == 1 ? ROWS * kDLd : kDK * (ROWS + 16); }
template <int MODE, int ROWS>
__device__ __forceinline__ void deep_store(const DeepRegs &t, float *dst)
{
    const int tid = threadIdx.x;
    constexpr int NP = ROWS / 8, LPC = ROWS / 4;
    if (MODE == 1) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4 *>(dst + (p * 8 + (tid >> 5)) * kDLd + (tid & 31) * 4) = t.v[p];
    } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<f32x4 *>(dst + (p * (256 / LPC) + tid / LPC) * (ROWS + 16) + (tid % LPC) * 4) = t.v[p];
    }
}
// operand of k-step `st` for the 16-row block `i` of a wave's sub-tile: base points at (row l16 of the sub-tile, k = lq)
template <int MODE, int ROWS>
__device__ __forceinline__ float deep_operand(const float *base, int i, int st)
{
    return MODE == 1 ? base[i * 16 * kDLd + st * 4] : base[st * 4 * (ROWS + 16) + i * 16];
}

// BM = rows of C per workgroup: 64 (each wave 32x32 as 2x2 accumulators) or 32 (each wave 16x32 as 1x2) -- the latter
// doubles the workgroup count for problems that would leave CUs idle (forward layers with N <= 512 at batch 1024).
template <int MA, int MB, int BM>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_f32_skinny_deep_kernel(const GemmArgs g)
{
    constexpr int BN = 64, TI = BM / 32, HM = BM / 2;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float *sA = dsm, *sB = dsm + deep_floats<MA, BM>();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int gx = g.N / BN;
    const int m0 = ((int)blockIdx.x / gx) * BM, n0 = ((int)blockIdx.x % gx) * BN;
    // K-slice in units of 32-deep chunks, as gemm_effective_splits counts them (the launcher made it a multiple of 4)
    const int all_chunks = g.K / kGK;
    const int splits = g.split_k > 1 ? g.split_k : 1;
    const int per = (all_chunks + splits - 1) / splits;
    const int c_lo = (int)blockIdx.y * per, c_hi = min(all_chunks, c_lo + per);
    if (c_lo >= c_hi) return;
    const int k_lo = c_lo * kGK, k_end = c_hi * kGK;
    f32x4 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l16 = lane & 15, lq = lane >> 4;
    DeepRegs ta, tb;
    deep_load<MA, BM>(ta, g.A, m0, k_lo, g.a_rs, g.a_cs);
    deep_load<MB, BN>(tb, g.B, n0, k_lo, g.b_rs, g.b_cs);
    deep_store<MA, BM>(ta, sA);
    deep_store<MB, BN>(tb, sB);
    __syncthreads();
    const float *pa = MA == 1 ? sA + (wr * HM + l16) * kDLd + lq : sA + lq * (BM + 16) + wr * HM + l16;
    const float *pb = MB == 1 ? sB + (wc * 32 + l16) * kDLd + lq : sB + lq * (BN + 16) + wc * 32 + l16;
    for (int k0 = k_lo; k0 < k_end; k0 += kDK) {
        const bool more = k0 + kDK < k_end;
        if (more) {  // the next chunk's global loads fly under this chunk's 128 MFMAs per wave
            deep_load<MA, BM>(ta, g.A, m0, k0 + kDK, g.a_rs, g.a_cs);
            deep_load<MB, BN>(tb, g.B, n0, k0 + kDK, g.b_rs, g.b_cs);
        }
        // operands of k-step s+1 are read from LDS before the MFMAs of step s issue: with one wave per SIMD nobody else
        // hides the ds_read latency
        float av[2][TI], bv[2][2];
#pragma unroll
        for (int i = 0; i < TI; ++i) av[0][i] = deep_operand<MA, BM>(pa, i, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[0][j] = deep_operand<MB, BN>(pb, j, 0);
#pragma unroll
        for (int st = 0; st < kDK / 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < kDK / 4) {
#pragma unroll
                for (int i = 0; i < TI; ++i) av[nxt][i] = deep_operand<MA, BM>(pa, i, st + 1);
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[nxt][j] = deep_operand<MB, BN>(pb, j, st + 1);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // everyone is done reading the buffer
        if (more) {
            deep_store<MA, BM>(ta, sA);
            deep_store<MB, BN>(tb, sB);
        }
        __syncthreads();
    }
    // accumulator register r of a 16x16 block = row 4*(lane>>4) + r, column lane&15
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * HM + i * 16 + 4 * lq + r, n = n0 + wc * 32 + j * 16 + l16;
                if (m < g.M && n < g.N) {
                    float s = acc[i][j][r];
                    if (splits > 1) {
                        if (g.sk_part) g.sk_part[(size_t)blockIdx.y * g.sk_stride + (size_t)m * g.ldc + n] = s;
                        else unsafeAtomicAdd(&g.C[(size_t)m * g.ldc + n], s);
                        continue;
                    }
                    if (g.row_bias) s = ((s + g.row_bias[m]) + g.col_bias[n]) + g.const_add;
                    else if (g.col_bias) s += g.col_bias[n];
                    if (g.relu) s = s > 0.f ? s : 0.f;
                    if (g.sigmoid) s = 1.f / (1.f + expf(-s));
                    if (g.mask) s = g.mask[(size_t)m * g.ldmask + n] > 0.f ? s : 0.f;
                    if (g.drop_thresh24) s = rk_drop_keep(g.drop_seed, (unsigned)((size_t)m * g.N + n), g.drop_thresh24) ? s * g.drop_scale : 0.f;
                    g.C[(size_t)m * g.ldc + n] = s;
                }
            }
}

// ---- 128x128 tiles with 64-deep chunks ("wide"): the deep form's recipe for problems with MANY tiles (the scoring GEMM).
// Against gemm_f32_kernel: half the barrier pairs per k, b128 LDS stores (rows of 68 floats stay 16-byte aligned; the 33-float
// rows of the 32x32x2 layout force scalar stores), operands of k-step s+1 read before step s issues, and 2 workgroups of
// 256 VGPRs per CU instead of 3 squeezed under 168 (scratch spills).  Each wave owns 64x64 as 4x4 accumulators of
// v_mfma_f32_16x16x4_f32; same k-ordered chain per element.  A is k-contiguous (scoring, tower forward, dX);
// partial last tiles are taken for k-contiguous operands (row loads clamped to the last row, stores guarded); for a
// row-contiguous B the launcher hands the edge strip to gemm_f32_kernel.
static constexpr int kWK = 64, kWLd = kWK + 4, kWLdT = 128 + 16;
template <int MODE> constexpr int wide_floats() { return MODE == 1 ? 128 * kWLd : kWK * kWLdT; }
template <int MODE>
__device__ __forceinline__ void wide_load(DeepRegs &t, const float *__restrict__ src, const long long (&rowoff)[8], int r0, int k0, long long cs)
{
    const int tid = threadIdx.x;
    if (MODE == 1) {   // 16 float4 per row of 64 k: thread -> (row = p*16 + tid/16, k4 = tid%16); rowoff[p] = row(p) * rs
#pragma unroll
        for (int p = 0; p < 8; ++p) t.v[p] = *reinterpret_cast<const f32x4 *>(src + rowoff[p] + (k0 + (tid & 15) * 4));
    } else {           // 32 float4 per k column of 128 rows: thread -> (k = p*8 + tid/32, row4 = tid%32)
        const float *q = src + (long long)(k0 + (tid >> 5)) * cs + (r0 + (tid & 31) * 4);
#pragma unroll
        for (int p = 0; p < 8; ++p) t.v[p] = *reinterpret_cast<const f32x4 *>(q + (long long)p * 8 * cs);
    }
}
template <int MODE>
__device__ __forceinline__ void wide_store(const DeepRegs &t, float *dst)
{
    const int tid = threadIdx.x;
    if (MODE == 1) {
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4 *>(dst + (p * 16 + (tid >> 4)) * kWLd + (tid & 15) * 4) = t.v[p];
    } else {
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<f32x4 *>(dst + (p * 8 + (tid >> 5)) * kWLdT + (tid & 31) * 4) = t.v[p];
    }
}
template <int MODE>
__device__ __forceinline__ float wide_operand(const float *base, int i, int st)
{
    return MODE == 1 ? base[i * 16 * kWLd + st * 4] : base[st * 4 * kWLdT + i * 16];
}

#ifndef GEMM_STAMP   // (scripts/gemm_probe.hip -DGEMM_STAMPS: wall-clock stamps per workgroup and tile)
#define GEMM_STAMP(it, i) do { } while (0)
#endif
// PLAIN: C = A.B^T with no bias / ReLU / sigmoid / mask (the LightGCN scoring GEMM) as a separate instantiation.  With the
// epilogue options tested at run time, every one of a tile's 64 stored elements walked a chain of scalar branches: per-tile
// stamps (scripts/gemm_probe.hip -DGEMM_STAMPS, profiles/r05_gemm_stamps.txt) showed 4.7 us of "stores" after every 8 us MFMA
// phase at 5 893 x 3 702 x 64 -- instruction issue, not memory (half the workgroups storing at a time took just as long).
// GROUPED (with PLAIN; the launcher takes it when C's rows start on 128-byte lines): full tiles finish their last chunk group by
// group with the stores under the next group's MFMAs (below).  A separate instantiation on purpose: as a run-time branch inside one
// kernel the grouped path cost the OTHER path its schedule (8192 x 34474 x 256 with the natural row stride: 1 209 -> 1 264 us).
template <int MA, int MB, bool PLAIN = false, bool GROUPED = false>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_f32_wide_kernel(const GemmArgs g, const int gx, const int gy)
{
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float *sA = dsm, *sB = dsm + wide_floats<MA>();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
    const int wr = w >> 1, wc = w & 1;
    // Persistent workgroups, each a contiguous run of the tile order; workgroups are dealt round-robin over the 8 XCDs, so
    // give every XCD a contiguous range of runs (bijective remap).
    int bid = blockIdx.x;
    const int nb = gridDim.x;
    {
        const int q = nb / 8, r = nb % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int n_tiles = gx * gy;
    const int t_begin = (int)((long long)bid * n_tiles / nb), t_end = (int)((long long)(bid + 1) * n_tiles / nb);
    if (t_begin >= t_end) return;
    const int n_chunks = g.K / kWK, total = (t_end - t_begin) * n_chunks;
    f32x4 acc[4][4];
    const int l16 = lane & 15, lq = lane >> 4;
    long long ra[8], rb[8];
    int m0, n0, m1, n1;   // origin of the tile being accumulated / of the tile whose chunk is being prefetched
    auto row_offsets = [&](int mt, int nt) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // rows past the edge of a partial tile (k-contiguous operands only) re-read the last row; their results are not stored
            const int m = min(mt + p * 16 + (tid >> 4), g.M - 1);
            ra[p] = (long long)(g.a_ridx ? g.a_ridx[m] : g.a_rmod ? (m + g.a_roff) % g.a_rmod : m) * g.a_rs;
            rb[p] = (long long)min(nt + p * 16 + (tid >> 4), g.N - 1) * g.b_rs;
        }
    };
    // accumulators of the tile at (mt, nt): zero, or the per-user prefix the k-ordered chain continues (GemmArgs::acc_init);
    // in the transposed block layout a lane owns C[mt + .. + l16][nt + .. + 4*lq + 0..3]
    auto init_acc = [&](int mt, int nt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *src = nullptr;
            // (never in the PLAIN instantiation -- the launcher takes it only without acc_init: a POSSIBLE global load into the
            //  accumulators makes the compiler drain every memory operation (s_waitcnt vmcnt(0)) in front of a tile's first MFMA --
            //  the prefetch of the next chunk's operands and the stores of the tile before included: nothing overlapped, the
            //  lock step of r05_gemm_stamps.txt)
            if (!PLAIN && g.acc_init) {
                const int m = min(mt + wr * 64 + i * 16 + l16, g.M - 1);
                src = g.acc_init + (size_t)((m + g.a_roff) / g.a_rmod - g.init_base) * g.ld_init;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nt + wc * 64 + j * 16 + 4 * lq;
                if (!src) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                else if (n + 4 <= g.N) acc[i][j] = *reinterpret_cast<const f32x4_u *>(src + n);
                else acc[i][j] = f32x4{n < g.N ? src[n] : 0.f, n + 1 < g.N ? src[n + 1] : 0.f, n + 2 < g.N ? src[n + 2] : 0.f, 0.f};
            }
        }
    };
    GEMM_STAMP(0, 0);
    tile_origin<128>(t_begin, gx, gy, m0, n0);
    m1 = m0; n1 = n0;
    row_offsets(m0, n0);
    init_acc(m0, n0);
    DeepRegs ta, tb;
    wide_load<MA>(ta, g.A, ra, m0, 0, g.a_cs);
    wide_load<MB>(tb, g.B, rb, n0, 0, g.b_cs);
    wide_store<MA>(ta, sA);
    wide_store<MB>(tb, sB);
    __syncthreads();
    const float *pa = MA == 1 ? sA + (wr * 64 + l16) * kWLd + lq : sA + lq * kWLdT + wr * 64 + l16;
    const float *pb = MB == 1 ? sB + (wc * 64 + l16) * kWLd + lq : sB + lq * kWLdT + wc * 64 + l16;
    const bool plain = PLAIN || (!g.row_bias && !g.col_bias && !g.relu && !g.sigmoid && !g.mask);
    // ONE software pipeline over all (tile, k-chunk) pairs of the run: the next pair's global loads fly under this chunk's
    // MFMAs, and a finished tile's stores drain under the next tile's MFMAs.
    int c = 0;
    GEMM_STAMP(0, 1);
    for (int it = 0; it < total; ++it) {
        const bool more = it + 1 < total, last = c == n_chunks - 1;
        GEMM_STAMP(it, 2);
        if (more) {
            if (last) {
                tile_origin<128>(t_begin + (it + 1) / n_chunks, gx, gy, m1, n1);
                row_offsets(m1, n1);
            }
            const int k1 = last ? 0 : (c + 1) * kWK;
            wide_load<MA>(ta, g.A, ra, m1, k1, g.a_cs);
            wide_load<MB>(tb, g.B, rb, n1, k1, g.b_cs);
        }
        float av[2][4], bv[2][4];
        __builtin_amdgcn_s_setprio(1);   // the wave in its MFMA phase wins issue slots over the co-resident one's stores (+1.4 %)
        // PLAIN, a full tile's LAST chunk (k = 64: its only one): the 16 accumulator blocks are finished GROUP BY GROUP -- four groups
        // of 2 x 2 blocks, each walked through all 16 k-steps (the same k-ordered chain per element: same bits) -- and a finished
        // group's four 16-byte stores are issued right behind its last MFMA, so they drain under the next group's 64 MFMAs (2 048
        // pipe cycles) instead of all 16 stores of every wave of the chip hitting the write path at once behind the tile's last MFMA
        // (profiles/r05_gemm_stamps.txt: 4.7 us of stores per 7.2 us MFMA phase at 5 893 x 3 702 x 64, nothing overlapped; a second
        // accumulator set to overlap them with the NEXT tile does not fit beside the operand prefetch).  Costs LDS operand reads:
        // 4 per 4 MFMAs instead of 8 per 16.
        // Only for score rows that start on 128-byte lines (rk_score_topk pads them: plan.ld_scores): with the natural stride of a
        // 34 474- or 3 702-item catalogue every 64-byte store segment straddles two lines and the interleaved stores stall the MFMAs
        // instead of hiding under them (profiles/r06d_gemm_ab.txt: 54 617 x 34 474 x 128 4.47 -> 7.3 ms unaligned, 4.69 -> 4.21 ms
        // aligned; 5 893 x 3 702 x 64 58 -> 82 us unaligned, 44.3 -> 40.0 us aligned) -- unaligned callers keep the round-5 order.
        const bool grouped = GROUPED && PLAIN && last && m0 + 128 <= g.M && n0 + 128 <= g.N;
        if (grouped) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int i0 = (gq >> 1) * 2, j0 = (gq & 1) * 2;
                float ga[2][2], gb[2][2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    ga[0][e] = wide_operand<MA>(pa, i0 + e, 0);
                    gb[0][e] = wide_operand<MB>(pb, j0 + e, 0);
                }
#pragma unroll
                for (int st = 0; st < kWK / 4; ++st) {
                    const int cur = st & 1, nxt = cur ^ 1;
                    if (st + 1 < kWK / 4) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            ga[nxt][e] = wide_operand<MA>(pa, i0 + e, st + 1);
                            gb[nxt][e] = wide_operand<MB>(pb, j0 + e, st + 1);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gb[cur][j], ga[cur][i], acc[i0 + i][j0 + j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float *crow = g.C + (size_t)(m0 + wr * 64 + (i0 + i) * 16 + l16) * g.ldc + (n0 + wc * 64 + 4 * lq);
#pragma unroll
                    for (int j = 0; j < 2; ++j) __builtin_nontemporal_store(acc[i0 + i][j0 + j], reinterpret_cast<f32x4_u *>(crow + (j0 + j) * 16));
                }
                __builtin_amdgcn_sched_barrier(0);   // (the group's stores stay in front of the next group's MFMAs)
            }
            __builtin_amdgcn_s_setprio(0);
            GEMM_STAMP(it, 3);
            m0 = m1; n0 = n1;
            if (more) init_acc(m0, n0);
            c = 0;
            GEMM_STAMP(it, 4);
            __syncthreads();
            GEMM_STAMP(it, 5);
            if (more) {
                wide_store<MA>(ta, sA);
                wide_store<MB>(tb, sB);
            }
            __syncthreads();
            GEMM_STAMP(it, 6);
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[0][i] = wide_operand<MA>(pa, i, 0);
            bv[0][i] = wide_operand<MB>(pb, i, 0);
        }
#pragma unroll
        for (int st = 0; st < kWK / 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < kWK / 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    av[nxt][i] = wide_operand<MA>(pa, i, st + 1);
                    bv[nxt][i] = wide_operand<MB>(pb, i, st + 1);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[cur][j], av[cur][i], acc[i][j], 0, 0, 0);   // the block TRANSPOSED
        }
        __builtin_amdgcn_s_setprio(0);
        GEMM_STAMP(it, 3);
        if (last) {
            // tile finished.  The MFMAs took (B, A), so a 16x16 accumulator block holds C^T: register r of lane (l16, lq) is
            // C[row l16][column 4*lq + r] -- four consecutive columns per lane, one 16-byte store (a quarter of the store
            // instructions of the row-per-register layout; same products, same order, same bits).
            auto emit = [&](auto guarded) {
                constexpr bool G = decltype(guarded)::value;   // partial tile: per-element bounds tests
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + wr * 64 + i * 16 + l16;
                    if (G && m >= g.M) continue;
                    float *crow = g.C + (size_t)m * g.ldc;
                    const float rbias = (!PLAIN && !plain && g.row_bias) ? g.row_bias[g.a_ridx ? g.a_ridx[m] : m] : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wc * 64 + j * 16 + 4 * lq;
                        f32x4 v = acc[i][j];
                        if (!PLAIN && !plain) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if (G && n + r >= g.N) continue;
                                float s = v[r];
                                if (g.row_bias) s = ((s + rbias) + g.col_bias[n + r]) + g.const_add;
                                else if (g.col_bias) s += g.col_bias[n + r];
                                if (g.relu) s = s > 0.f ? s : 0.f;
                                if (g.sigmoid) s = 1.f / (1.f + expf(-s));
                                if (g.mask) s = g.mask[(size_t)m * g.ldmask + n + r] > 0.f ? s : 0.f;
                                v[r] = s;
                            }
                        }
                        if (!G || n + 4 <= g.N) {
                            // the score matrix is written once and read by the NEXT kernel: streaming (nt) stores measured +1-2 % at
                            // the ml1m / d = 256 shapes, +7 % at 54 617 x 34 474 x 128 (profiles/r05_gemm_nt.txt); the tower GEMMs'
                            // outputs are re-read from L2 by the next layer and keep the default policy
                            if (PLAIN) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_u *>(crow + n));
                            else *reinterpret_cast<f32x4_u *>(crow + n) = v;
                        }
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (n + r < g.N) crow[n + r] = v[r];
                        }
                    }
                }
            };
            // The iteration's tail -- barrier, the prefetched operands into LDS, barrier -- stands once per path on purpose: the
            // operand loads were issued in front of this chunk's MFMAs and are OLDER than the tile's stores, so waiting for them
            // is `s_waitcnt vmcnt(stores + younger loads)` -- which the compiler can only count where the number of stores behind
            // the loads is known.  With one shared tail it merged "16 stores", "some stores" (partial tile) and "none" (chunk
            // inside a tile) into vmcnt(15 .. 0): every finished tile's stores were drained before the next operands went to LDS.
            // (GROUPED: full tiles never get here -- the grouped path above stores them and runs its own tail)
            if (PLAIN && !GROUPED && m0 + 128 <= g.M && n0 + 128 <= g.N) {   // (the score-matrix instantiation: the others keep one tail and their register budget)
                emit(std::false_type{});
                m0 = m1; n0 = n1;
                if (more) init_acc(m0, n0);
                c = 0;
                GEMM_STAMP(it, 4);
                __syncthreads();
                GEMM_STAMP(it, 5);
                if (more) {
                    wide_store<MA>(ta, sA);
                    wide_store<MB>(tb, sB);
                }
                __syncthreads();
                GEMM_STAMP(it, 6);
                continue;
            }
            if (!PLAIN && m0 + 128 <= g.M && n0 + 128 <= g.N) emit(std::false_type{});
            else emit(std::true_type{});
            m0 = m1; n0 = n1;
            if (more) init_acc(m0, n0);
            c = 0;
        } else ++c;
        GEMM_STAMP(it, 4);
        __syncthreads();
        GEMM_STAMP(it, 5);
        if (more) {
            wide_store<MA>(ta, sA);
            wide_store<MB>(tb, sB);
        }
        __syncthreads();
        GEMM_STAMP(it, 6);
    }
}

// asynchronous launch; returns the hipError_t of the launch
inline hipError_t gemm_f32_launch(const GemmArgs &g, hipStream_t s)
{
    static RkPerDeviceOnce attr_once;
    static const int variant = RK_TUNE_INT("RK_GEMM_VARIANT", 0);
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f32_kernel<128, 2, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds_bytes<128>(2));
        if (e != hipSuccess) return e;
        attr_once.done(attr_dev);
    }
    const int nwg128 = ((g.N + 127) / 128) * ((g.M + 127) / 128);
    // skinny problem (NCF tower at B=1024): 64-tiles fill 4x more CUs; parked K-slices (sk_part) are only wired for this form and
    // take it at any size (the blocked training forward of large batches)
    if ((nwg128 < 384 || (g.split_k > 1 && g.sk_part)) && variant != 3 && !g.a_ridx && !g.a_rmod) {
        const int nwg = ((g.N + 63) / 64) * ((g.M + 63) / 64);
        GemmArgs g2 = g;
        const int splits = g2.split_k = gemm_effective_splits(g.K, g.split_k);
        // whole 64 x 128 chunks of aligned operands: the 128-deep chunk loop (even a single chunk gains from the half tiles,
        // the b128 LDS stores and the pipelined operand reads: 1024 x 512 x 256 14.1 -> 8.4 us, 1024 x 256 x 128 9.2 -> 5.8 us)
        const int chunks = (g.K + kGK - 1) / kGK, per = (chunks + splits - 1) / splits;
        static const int no_deep = RK_TUNE_INT("RK_GEMM_NO_DEEP", 0);   // A/B only
        static const int deep_min_k = RK_TUNE_INT("RK_GEMM_DEEP_MINK", 128);   // tuning only
        const int fa = deep_form(g.A, g.a_rs, g.a_cs), fb = deep_form(g.B, g.b_rs, g.b_cs);
        // (A row-contiguous with B k-contiguous has no caller: forward / dX / dW are <1,1>, <1,2>, <2,2>)
        const bool deep = !no_deep && variant != 4 && fa && fb && !(fa == 2 && fb == 1) && g.M % 64 == 0 && g.N % 64 == 0 && g.K % kDK == 0 && per % 4 == 0 &&
                          per * kGK >= deep_min_k;
        if (variant == 4) hipLaunchKernelGGL((gemm_f32_kernel<64, 1, 4>), dim3(nwg, splits), dim3(256), gemm_lds_bytes<64>(1), s, g2, 1);
        else if (deep) {
            static const int no_half = RK_TUNE_INT("RK_GEMM_NO_HALF", 0);   // A/B only
            const bool half = nwg * splits < 256 && !no_half;   // fewer workgroups than CUs: 32-row tiles
            const void *fn[6] = {reinterpret_cast<const void *>(gemm_f32_skinny_deep_kernel<1, 1, 64>), reinterpret_cast<const void *>(gemm_f32_skinny_deep_kernel<1, 2, 64>),
                                 reinterpret_cast<const void *>(gemm_f32_skinny_deep_kernel<2, 2, 64>),
                                 reinterpret_cast<const void *>(gemm_f32_skinny_deep_kernel<1, 1, 32>), reinterpret_cast<const void *>(gemm_f32_skinny_deep_kernel<1, 2, 32>),
                                 reinterpret_cast<const void *>(gemm_f32_skinny_deep_kernel<2, 2, 32>)};
            static RkPerDeviceOnce deep_attr;
            int deep_attr_dev;
            if (deep_attr.need(&deep_attr_dev)) {
                for (const void *f : fn) {
                    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * deep_floats<2, 64>() * (int)sizeof(float));
                    if (e != hipSuccess) return e;
                }
                deep_attr.done(deep_attr_dev);
            }
            const int fl_a = half ? (fa == 1 ? deep_floats<1, 32>() : deep_floats<2, 32>()) : (fa == 1 ? deep_floats<1, 64>() : deep_floats<2, 64>());
            const int fl_b = fb == 1 ? deep_floats<1, 64>() : deep_floats<2, 64>();
            const size_t lds = (size_t)(fl_a + fl_b) * sizeof(float);
            const dim3 grid(half ? 2 * nwg : nwg, splits);
            const int which = (half ? 3 : 0) + (fa == 2 ? 2 : fb == 2 ? 1 : 0);
            void *params[1] = {const_cast<GemmArgs *>(&g2)};
            return hipLaunchKernel(fn[which], grid, dim3(256), params, lds, s);
        } else hipLaunchKernelGGL(gemm_f32_skinny_kernel, dim3(nwg, splits), dim3(256), 0, s, g2);
        return hipGetLastError();
    }
    if (g.split_k > 1) return hipErrorInvalidValue;  // split-K is only wired for the 64-tile form
    {
        // many 128-tiles, aligned operands, A k-contiguous (scoring, tower forward, dX), K a multiple of 64: the wide kernel.
        // A k-contiguous B takes a partial last tile (clamped row loads); a row-contiguous B (dX) needs whole column tiles
        // and leaves the edge strip to this function again (fewer than 128 columns: it never comes back here).
        static const int no_wide = RK_TUNE_INT("RK_GEMM_NO_WIDE", 0);   // A/B only
        static const int wide_min_k = RK_TUNE_INT("RK_GEMM_WIDE_MINK", kWK);   // tuning only
        static const int strips = RK_TUNE_INT("RK_GEMM_WIDE_STRIPS", 0);   // A/B only
        static const int wide_wgs = RK_TUNE_INT("RK_GEMM_WIDE_WGS", 512);   // tuning only: 2 per CU
        const int fa = deep_form(g.A, g.a_rs, g.a_cs), fb = deep_form(g.B, g.b_rs, g.b_cs);
        const int Ni = fb == 1 && !strips ? g.N : g.N / 128 * 128;
        if (!no_wide && variant == 0 && fa == 1 && fb && g.M >= 128 && Ni >= 128 && g.K % kWK == 0 && g.K >= wide_min_k &&
            (!g.acc_init || g.a_rmod > 0) && !g.drop_thresh24) {
            const bool plain_w = !g.row_bias && !g.col_bias && !g.relu && !g.sigmoid && !g.mask && !g.acc_init;
            const bool c_lines = (g.ldc & 31) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 127) == 0;   // rows of C start on 128-byte lines
            const void *fn[4] = {reinterpret_cast<const void *>(gemm_f32_wide_kernel<1, 1>), reinterpret_cast<const void *>(gemm_f32_wide_kernel<1, 2>),
                                 reinterpret_cast<const void *>(gemm_f32_wide_kernel<1, 1, true>),
                                 reinterpret_cast<const void *>(gemm_f32_wide_kernel<1, 1, true, true>)};
            static RkPerDeviceOnce wide_attr;
            int wide_attr_dev;
            if (wide_attr.need(&wide_attr_dev)) {
                for (const void *f : fn) {
                    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * wide_floats<2>() * (int)sizeof(float));
                    if (e != hipSuccess) return e;
                }
                wide_attr.done(wide_attr_dev);
            }
            int gx = (Ni + 127) / 128, gy = (g.M + 127) / 128;
            const size_t lds = (size_t)(wide_floats<1>() + (fb == 1 ? wide_floats<1>() : wide_floats<2>())) * sizeof(float);
            void *params[3] = {const_cast<GemmArgs *>(&g), &gx, &gy};
            hipError_t e = hipLaunchKernel(fn[fb == 2 ? 1 : plain_w ? (c_lines ? 3 : 2) : 0], dim3(std::min(gx * gy, wide_wgs)), dim3(256), params, lds, s);
            if (e != hipSuccess) return e;
            if (Ni < g.N) {   // right strip: all rows, columns [Ni, N)
                GemmArgs e1 = g;
                e1.N = g.N - Ni; e1.B = g.B + (long long)Ni * g.b_rs; e1.C = g.C + Ni;
                if (g.col_bias) e1.col_bias = g.col_bias + Ni;
                if (g.mask) e1.mask = g.mask + Ni;
                if (g.acc_init) e1.acc_init = g.acc_init + Ni;
                e = gemm_f32_launch(e1, s);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    // short-K problems are epilogue-bound: run several tiles per workgroup so stores drain under MFMAs
    int tpb = (g.K <= 64) ? nwg128 / 1024 : 1;
    static const int tpb_env = RK_TUNE_INT("RK_GEMM_TPB", 0);   // tuning only
    if (tpb_env > 0) tpb = tpb_env;
    tpb = tpb < 1 ? 1 : (tpb > 64 ? 64 : tpb);
    const dim3 grid((nwg128 + tpb - 1) / tpb);
    // default: single LDS buffer + register prefetch, 3 workgroups per CU (measured 101 TF/s at K=256
    // vs 93 for the double-buffered 2-per-CU form, RK_GEMM_VARIANT=1)
    const bool plain = !g.row_bias && !g.col_bias && !g.relu && !g.mask && !g.sigmoid && !g.drop_thresh24;
    if (g.a_ridx && plain && !g.acc_init) hipLaunchKernelGGL((gemm_f32_kernel<128, 1, 3, true, true>), grid, dim3(256), gemm_lds_bytes<128>(1), s, g, tpb);
    else if (g.a_ridx || g.a_rmod) hipLaunchKernelGGL((gemm_f32_kernel<128, 1, 3, true, false>), grid, dim3(256), gemm_lds_bytes<128>(1), s, g, tpb);
    else if (variant == 1) hipLaunchKernelGGL((gemm_f32_kernel<128, 2, 2>), grid, dim3(256), gemm_lds_bytes<128>(2), s, g, tpb);
    else hipLaunchKernelGGL((gemm_f32_kernel<128, 1, 3>), grid, dim3(256), gemm_lds_bytes<128>(1), s, g, tpb);
    return hipGetLastError();
}
