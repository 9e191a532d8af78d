// Fused full-catalog scoring + selection for gfx950: the [users, items] score matrix of
// Normal.user_item_model_generate (recad/workflow/normal.py:57-93) is never materialised.
//
// One workgroup owns RB user rows (their embedding rows stay in LDS for the whole kernel) and sweeps the
// catalogue in tiles of 128 items:
//   * scores of the RB x 128 tile on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, k-ordered fmaf chain =>
//     bit-identical to the oracle's scalar loop), item rows streamed through a double-buffered LDS tile in
//     k-chunks of 32 with register prefetch of the next chunk;
//   * epilogue straight from the accumulators: bias terms, seen-item mask (a 128-bit LDS bitmap per row and
//     tile, filled from the user's sorted train list by a running cursor whose loads fly under the MFMAs),
//     rank of up to 4 targets (#(s > s_t) + #(s == s_t, id < t) in per-lane registers), and a THRESHOLDED
//     append of (key, ~id) composites to the row's candidate list in global memory (512 slots per row):
//     a score enters only if its key is above tau_row, the K-th best key seen so far.  Items arrive in
//     increasing id order, so once K candidates with key >= tau are held a later item with key == tau can
//     never enter the top K (it loses the id tie-break): "key > tau" is exact, ties included;
//   * when a row's list could overflow with the next tile, one wave compacts it to its exact top K
//     (bit-wise binary search for the K-th largest 64-bit composite: 8 entries per lane in registers,
//     v_cmp_ge_u64 + s_bcnt1 per entry and bit) and raises tau_row.  Expected appends after the first
//     compaction are ~K ln(tiles/3): one or two compactions per row for catalogues up to millions of items;
//   * after the last tile every row is compacted once more, rank-sorted through LDS by (score desc, id asc)
//     and written out with the target ranks.
// Nothing but the K results per row, the target scores / ranks and the (L2-resident) candidate lists
// touches memory: 22 M scores of the ml1m evaluation were 88 MB written and re-read before.
#pragma once
#include <algorithm>
#include <stdlib.h>

#include "common.h"

typedef float sel_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned score_key(float s)
{
    // monotone float -> uint; 0 is reserved for "excluded" (seen item, encoded as -inf).  -0.0 is
    // folded onto +0.0 so that key equality is float equality (the oracle compares floats).
    if (s == -INFINITY) return 0u;
    unsigned u = __float_as_uint(s);
    if (u == 0x80000000u) u = 0u;
    const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return k == 0u ? 1u : k;
}

__device__ __forceinline__ float key_score(unsigned k)
{
    // inverse of score_key for every finite score and +inf (-0.0 comes back as +0.0)
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

static constexpr int kSelC = 512;     // candidate slots per user row
static constexpr int kSelTN = 128;    // items per tile
static constexpr int kSelKC = 32;     // k per LDS chunk
static constexpr int kSelLdB = 36;    // padded chunk row: operand reads hit every bank exactly twice, rows 16-byte aligned
static constexpr int kSelMaxT = 4;    // targets ranked in the sweep
static constexpr int kSelMaxK = 256;
static constexpr int kSelMaxSplits = 8;

struct SelArgs {
    int nb, n_items, d, K;
    const float *utab;
    const int *user_ids;
    const float *itab;
    const float *ubias, *ibias;   // both or neither: s = ((dot + ubias[u]) + ibias[i]) + mean
    float mean;
    const int *seen_ptr, *seen_idx;
    const int *targets;
    int n_targets;
    int *top_ids;
    float *top_scores;
    float *target_score;
    int *target_rank;
    unsigned long long *cand;     // [nb][n_splits][kSelC] scratch
    int *cand_cnt;                // [nb][n_splits] scratch
    int *rank_part;               // [nb][n_splits][n_targets] scratch: per-split target rank counts, summed by the finalize kernel
    int n_splits;                 // the catalogue is swept in n_splits contiguous tile ranges, one workgroup per (row block, range)
    int id_bits;                  // item ids < 2^id_bits
    unsigned long long *stamps;   // diagnostic build (RK_SEL_STAMPS) only: [grid][8] cycle sums of wave 0
};

__host__ __device__ inline int sel_row_stride(int d) { return ((d + kSelKC - 1) / kSelKC) * kSelKC + 4; }

// Threshold search over the (distinct) composites held 8 per lane (unused slots 0), n_valid >= K.  Bit-wise binary
// search from the top: T grows while at least K composites stay >= T.  After `min_bits` bits it stops as soon as at
// most `limit` composites are >= T (a COARSE threshold: cheap, and any T with count >= K is a valid pruning bound);
// with limit == K it runs to the exact K-th composite (early exit once the count is exactly K).  Key bits are
// searched on the high words alone (32-bit compares); the id bits above id_bits are ones in every entry.
// Returns T; *cnt_out = #(composite >= T).
__device__ __forceinline__ unsigned long long wave_threshold(const unsigned long long (&c)[8], int n, int K, int limit, int min_bits,
                                                             int id_bits, int *cnt_out)
{
    unsigned hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) hi[j] = (unsigned)(c[j] >> 32);
    unsigned Tk = 0u;
    int cntT = n, done_bits = 0;
    for (int bit = 31; bit >= 0; --bit) {
        if (done_bits >= min_bits && cntT <= limit) { *cnt_out = cntT; return (unsigned long long)Tk << 32; }
        const unsigned trial = Tk | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += __popcll(__ballot(hi[j] >= trial));
        if (cnt >= K) { Tk = trial; cntT = cnt; }
        ++done_bits;
    }
    unsigned long long T = (unsigned long long)Tk << 32;
    if (cntT <= limit) { *cnt_out = cntT; return T; }
    // more than `limit` entries share the K-th key: resolve the tie on the (inverted) item ids
    if (id_bits < 32) T |= (0xffffffffULL >> id_bits) << id_bits;
    for (int bit = (id_bits < 32 ? id_bits : 32) - 1; bit >= 0; --bit) {
        const unsigned long long trial = T | (1ULL << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += __popcll(__ballot(c[j] >= trial));
        if (cnt >= K) { T = trial; cntT = cnt; }
        if (cntT <= limit) break;
    }
    *cnt_out = cntT;
    return T;
}

// One wave: shrink the candidate list of a row (n entries in global memory, kSelC - kSelTN < n <= kSelC) in place
// to the entries at or above a threshold T that at least K of them reach (at most K + 64 unless more than that are
// tied at the K-th key and id); returns the new count, the new pruning bound (T's key) through *tau_out.
__device__ __forceinline__ int wave_compact_row(unsigned long long *__restrict__ row, int n, int K, int id_bits, int lane,
                                                unsigned *tau_out)
{
    unsigned long long c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = (j * 64 + lane < n) ? row[j * 64 + lane] : 0ULL;
    if (n <= K) { *tau_out = 0u; return n; }
    int cnt = n;
    const int limit = K + 64 < kSelC - 2 * kSelTN ? K + 64 : K;   // leave room for two more tiles when K is large
    const unsigned long long T = wave_threshold(c, n, K, limit, 12, id_bits, &cnt);
    int base = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool keep = c[j] >= T && c[j] != 0ULL;
        const unsigned long long m = __ballot(keep);
        if (keep) row[base + __popcll(m & ((1ULL << lane) - 1ULL))] = c[j];
        base += __popcll(m);
    }
    *tau_out = (unsigned)(T >> 32);
    return base;
}

// WM rows per wave (16 or 32), WAVES_M waves along the rows; 4 waves along the 128 items of a tile.
// NTG: targets ranked inside the sweep (1 or kSelMaxT: a separate instantiation keeps the common one-target
// evaluation free of three dead compare chains per score).  NCH: k-chunks of 32 (dim <= 32 * NCH).
//
// Operand traffic: the workgroup's user rows never change, so every lane keeps ITS A operands of the whole
// sweep in registers (A[row = lane&15][k = (lane>>4) + 4j], 8*NCH floats per 16-row block) -- no LDS reads for
// A at all.  The item tile is stored k-permuted ((k%4)*8 + k/4 inside a 32-chunk) so that the 8 B operands a
// lane needs for one chunk are 32 contiguous bytes: two ds_read_b128 per 16-column block and chunk.
template <int WM, int WAVES_M, int NTG, int NCH>
__global__ __launch_bounds__(WAVES_M * 256) void score_select_kernel(const SelArgs a)
{
    constexpr int RB = WM * WAVES_M, NW = WAVES_M * 4, NT = NW * 64, BM = WM / 16, RPL = BM * 4;  // rows per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int sCnt[RB], sCur[RB], sEnd[RB], sUid[RB], sRank[RB][kSelMaxT], sNeed;
    __shared__ unsigned sTau[RB], sTkey[RB][kSelMaxT], sMask[2][RB][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int l16 = lane & 15, lq = lane >> 4;
    const int SA = sel_row_stride(a.d);
    float *sB = smem;                                   // [2][128][kSelLdB], k-permuted chunks
    float *sA = smem;                                   // prologue only: [RB][SA] user rows, then the target rows
    const int row0 = blockIdx.x * RB;
    const int n_in = a.n_targets;                       // <= kSelMaxT (host-checked)
    const int n_tiles_all = (a.n_items + kSelTN - 1) / kSelTN;
    // item-range split (blockIdx.y): with few row blocks (ml1m: 93 of 64 rows) one workgroup per row block leaves most CUs idle
    // behind a serial chain of 29 tiles; the ranges keep their own exact top-K lists, merged by sel_finalize_kernel
    const int split = blockIdx.y, n_splits = gridDim.y;
    const int tile_begin = (int)((long long)n_tiles_all * split / n_splits), n_tiles = (int)((long long)n_tiles_all * (split + 1) / n_splits);

    // ---- prologue: row metadata, user rows -> LDS (zero-padded), target scores
    for (int r = tid; r < RB; r += NT) {
        const int g = row0 + r;
        const int u = g < a.nb ? a.user_ids[g] : -1;
        sUid[r] = u;
        int cur = u >= 0 ? a.seen_ptr[u] : 0;
        const int endp = u >= 0 ? a.seen_ptr[u + 1] : 0;
        if (tile_begin > 0) {   // first seen id of this range (the list is sorted)
            int lo = cur, hi = endp;
            const int first = tile_begin * kSelTN;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.seen_idx[mid] < first) lo = mid + 1; else hi = mid; }
            cur = lo;
        }
        sCur[r] = cur;
        sEnd[r] = endp;
        sCnt[r] = 0;
        sTau[r] = 0u;
#pragma unroll
        for (int t = 0; t < kSelMaxT; ++t) sRank[r][t] = 0;
    }
    for (int i = tid; i < 2 * RB * 4; i += NT) (&sMask[0][0][0])[i] = 0u;
    if (tid == 0) sNeed = 0;
    __syncthreads();
    for (int i = tid; i < RB * SA; i += NT) {
        const int r = i / SA, k = i % SA;
        const int u = sUid[r];
        sA[i] = (u >= 0 && k < a.d) ? a.utab[(size_t)u * a.d + k] : 0.f;
    }
    float *sT = sA + (size_t)RB * SA;                   // target item rows [n_in][d]
    for (int i = tid; i < n_in * a.d; i += NT) sT[i] = a.itab[(size_t)a.targets[i / a.d] * a.d + i % a.d];
    __syncthreads();
    // this lane's A operands for the whole sweep
    float areg[BM][NCH * 8];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int q = 0; q < NCH * 8; ++q) areg[i][q] = sA[(wm * WM + i * 16 + l16) * SA + (q >> 3) * kSelKC + lq + 4 * (q & 7)];
    // target scores (before masking, normal.py:83-85): one thread per (row, target) runs the MFMA's k-ordered chain
    for (int i = tid; i < RB * kSelMaxT; i += NT) {
        const int r = i / kSelMaxT, t = i % kSelMaxT;
        unsigned key = 0xffffffffu;
        if (t < n_in && sUid[r] >= 0) {
            const int tg = a.targets[t];
            const float *iv = sT + t * a.d, *av = sA + r * SA;
            float s = 0.f;
            for (int k = 0; k < a.d; ++k) s = fmaf(av[k], iv[k], s);
            if (a.ubias) s = ((s + a.ubias[sUid[r]]) + a.ibias[tg]) + a.mean;
            if (split == 0) a.target_score[(size_t)(row0 + r) * a.n_targets + t] = s;
            key = score_key(s);
        }
        sTkey[r][t] = key;
    }
    __syncthreads();   // sA / sT are dead from here on: the region becomes the item tile

    // per-lane constants: the rows this lane's accumulator registers belong to
    unsigned tkey[RPL][NTG];
    int cntr[RPL][NTG];
    int tgt[NTG];
    float ubr[RPL];
    bool rok[RPL];
#pragma unroll
    for (int t = 0; t < NTG; ++t) tgt[t] = t < n_in ? a.targets[t] : -1;
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int r = wm * WM + (q >> 2) * 16 + 4 * lq + (q & 3);
        rok[q] = sUid[r] >= 0;
        ubr[q] = (a.ubias && rok[q]) ? a.ubias[sUid[r]] : 0.f;
#pragma unroll
        for (int t = 0; t < NTG; ++t) { tkey[q][t] = sTkey[r][t]; cntr[q][t] = 0; }
    }

    // B tile loader: 128 rows x 32 k = 1024 float4, NT threads; LDS image k-permuted
    constexpr int NLD = 1024 / NT;   // float4 per thread (4 or 2)
    float4 breg[NLD];
    const bool vec_ok = (a.d % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.itab) & 15) == 0);
    const int ld_r = tid >> 3, ld_k4 = (tid & 7) * 4;            // first float4 of this thread; the others NT/8 rows on
    auto load_b = [&](int tile, int chunk) {
        const bool fast = vec_ok && (tile + 1) * kSelTN <= a.n_items && (chunk + 1) * kSelKC <= a.d;   // workgroup-uniform
        if (fast) {
            const float *src = a.itab + (size_t)(tile * kSelTN + ld_r) * a.d + chunk * kSelKC + ld_k4;
#pragma unroll
            for (int p = 0; p < NLD; ++p) breg[p] = *reinterpret_cast<const float4 *>(src + (size_t)p * (NT / 8) * a.d);
        } else {
#pragma unroll
            for (int p = 0; p < NLD; ++p) {
                const int item = tile * kSelTN + ld_r + p * (NT / 8), k = chunk * kSelKC + ld_k4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (item < a.n_items) {
                    const float *src = a.itab + (size_t)item * a.d + k;
                    if (k < a.d) v.x = src[0];
                    if (k + 1 < a.d) v.y = src[1];
                    if (k + 2 < a.d) v.z = src[2];
                    if (k + 3 < a.d) v.w = src[3];
                }
                breg[p] = v;
            }
        }
    };
    auto store_b = [&](int buf) {
        // element k of a row goes to (k % 4) * 8 + k / 4: the float4 (k4 .. k4+3) lands at k4/4, 8 + k4/4, 16 + ..., 24 + ...
        float *dst = sB + (size_t)buf * kSelTN * kSelLdB + ld_r * kSelLdB + (ld_k4 >> 2);
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            float *q = dst + p * (NT / 8) * kSelLdB;
            q[0] = breg[p].x; q[8] = breg[p].y; q[16] = breg[p].z; q[24] = breg[p].w;
        }
    };

    sel_f32x4 acc[BM][2];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = sel_f32x4{0.f, 0.f, 0.f, 0.f};

    load_b(tile_begin, 0);
    store_b(0);
    __syncthreads();
#ifdef RK_SEL_STAMPS
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_t0 = __builtin_amdgcn_s_memtime(), st_prev = st_t0;
#define RK_STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[k] += t_ - st_prev; st_prev = t_; }
#else
#define RK_STAMP(k)
#endif
    int buf = 0;
    for (int tile = tile_begin; tile < n_tiles; ++tile) {
        const int n0 = tile * kSelTN;
        // seen ids of this tile: loads issued before the MFMAs, consumed after them
        int seen_id = 0x7fffffff;
        const bool marker = tid < RB * 8;
        if (marker) {
            const int r = tid >> 3, pos = sCur[r] + (tid & 7);
            if (pos < sEnd[r]) seen_id = a.seen_idx[pos];
        }
#pragma unroll
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const bool more = chunk + 1 < NCH || tile + 1 < n_tiles;
            if (more) load_b(chunk + 1 < NCH ? tile : tile + 1, chunk + 1 < NCH ? chunk + 1 : 0);
            const float *Bc = sB + (size_t)buf * kSelTN * kSelLdB + (wn * 32 + l16) * kSelLdB + lq * 8;
            float4 b0[2], b1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b0[j] = *reinterpret_cast<const float4 *>(Bc + j * 16 * kSelLdB);
                b1[j] = *reinterpret_cast<const float4 *>(Bc + j * 16 * kSelLdB + 4);
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 &bb = jj < 4 ? b0[j] : b1[j];
                    const float bv = (jj & 3) == 0 ? bb.x : (jj & 3) == 1 ? bb.y : (jj & 3) == 2 ? bb.z : bb.w;
#pragma unroll
                    for (int i = 0; i < BM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[i][chunk * 8 + jj], bv, acc[i][j], 0, 0, 0);
                }
            }
            if (chunk == NCH - 1 && marker) {
                // mark this tile's seen items; the ids are sorted, so the consumed lanes of a row form a prefix
                const int r = tid >> 3, j8 = tid & 7;
                int curp = sCur[r];
                const int endp = sEnd[r];
                const int sub = (lane >> 3) * 8;
                for (;;) {
                    const bool in = seen_id < n0 + kSelTN;
                    if (in) atomicOr(&sMask[tile & 1][r][(seen_id - n0) >> 5], 1u << ((seen_id - n0) & 31));
                    const unsigned long long m = __ballot(in);
                    const int c8 = __popc((unsigned)((m >> sub) & 0xffULL));
                    curp += c8;
                    if (c8 < 8) break;
                    seen_id = (curp + j8 < endp) ? a.seen_idx[curp + j8] : 0x7fffffff;
                }
                if (j8 == 0) sCur[r] = curp;
            }
            RK_STAMP(0)
            if (more) store_b(buf ^ 1);
            RK_STAMP(1)
            __syncthreads();
            RK_STAMP(2)
            buf ^= 1;
        }

        // ---- epilogue of the tile, straight from the accumulators (branch-free: one LDS atomic per row group)
        // accumulator register r of block (i, j) = row 4*(lane>>4) + r, column lane&15 of the 16x16 block
#pragma unroll
        for (int i = 0; i < BM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = i * 4 + r;
                const int rl = wm * WM + i * 16 + 4 * lq + r;
                const unsigned mask = sMask[tile & 1][rl][wn];
                const unsigned tau = sTau[rl];
                unsigned key[2];
                bool app[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wn * 32 + j * 16 + l16;
                    float s = acc[i][j][r];
                    acc[i][j][r] = 0.f;
                    if (a.ubias) s = ((s + ubr[q]) + (col < a.n_items ? a.ibias[col] : 0.f)) + a.mean;
                    const bool valid = rok[q] && col < a.n_items && !((mask >> (j * 16 + l16)) & 1u);
                    key[j] = valid ? score_key(s) : 0u;
#pragma unroll
                    for (int t = 0; t < NTG; ++t)
                        cntr[q][t] += (key[j] != 0u && col != tgt[t] && (key[j] > tkey[q][t] || (key[j] == tkey[q][t] && col < tgt[t]))) ? 1 : 0;
                    app[j] = key[j] > tau;
                }
                // the 16 lanes of a row group reserve their slots with ONE atomic (lane l16 == 0); past the first tiles
                // almost no score beats its row's threshold, so the whole block is skipped wave-uniformly
                const unsigned long long m0 = __ballot(app[0]), m1 = __ballot(app[1]);
                if ((m0 | m1) != 0ULL) {
                    const unsigned g0 = (unsigned)(m0 >> (lq * 16)) & 0xffffu, g1 = (unsigned)(m1 >> (lq * 16)) & 0xffffu;
                    const int n0g = __popc(g0), tot = n0g + __popc(g1);
                    int base = 0;
                    if (l16 == 0 && tot) base = atomicAdd(&sCnt[rl], tot);
                    base = __shfl(base, lq * 16, 64);
                    unsigned long long *dst = a.cand + ((size_t)(row0 + rl) * n_splits + split) * kSelC + base;
                    const unsigned below = (1u << l16) - 1u;
                    if (app[0]) dst[__popc(g0 & below)] = ((unsigned long long)key[0] << 32) | (unsigned)(~(unsigned)(n0 + wn * 32 + l16));
                    if (app[1]) dst[n0g + __popc(g1 & below)] = ((unsigned long long)key[1] << 32) | (unsigned)(~(unsigned)(n0 + wn * 32 + 16 + l16));
                }
            }
        }
        RK_STAMP(3)
        __syncthreads();
        RK_STAMP(4)
        // ---- rows that could overflow with the next tile are shrunk to (about) their top K
        for (int i = tid; i < RB * 4; i += NT) sMask[tile & 1][i >> 2][i & 3] = 0u;
        bool need = false;
        for (int r = tid; r < RB; r += NT) need |= sCnt[r] > kSelC - kSelTN;
        if (need) sNeed = 1;
        __syncthreads();
        if (sNeed) {
            for (int r = w; r < RB; r += NW) {
                const int n = sCnt[r];
                if (n > kSelC - kSelTN) {   // wave-uniform
                    unsigned tau;
                    const int m = wave_compact_row(a.cand + ((size_t)(row0 + r) * n_splits + split) * kSelC, n, a.K, a.id_bits, lane, &tau);
                    if (lane == 0) { sCnt[r] = m; sTau[r] = tau; }
                }
            }
            __syncthreads();
            if (tid == 0) sNeed = 0;
            __syncthreads();
        }
        RK_STAMP(5)
    }
#ifdef RK_SEL_STAMPS
    if (tid == 0 && a.stamps) { for (int k = 0; k < 6; ++k) a.stamps[blockIdx.x * 8 + k] = st_sum[k]; a.stamps[blockIdx.x * 8 + 6] = st_t0; a.stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime(); }
#endif

    // ---- target ranks: per-lane counters -> 16-lane groups -> LDS (the 4 waves along the items add up)
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int rl = wm * WM + (q >> 2) * 16 + 4 * lq + (q & 3);
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            int v = cntr[q][t];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (l16 == 0 && t < n_in) atomicAdd(&sRank[rl][t], v);
        }
    }
    __syncthreads();
    for (int i = tid; i < RB * n_in; i += NT) {
        const int r = i / n_in, t = i % n_in;
        if (sUid[r] >= 0) a.rank_part[((size_t)(row0 + r) * n_splits + split) * a.n_targets + t] = sRank[r][t];
    }
    // ---- the candidate lists (at most kSelC entries per range, at least min(K, #unseen of the range)) go to sel_finalize_kernel
    for (int r = tid; r < RB; r += NT)
        if (row0 + r < a.nb) a.cand_cnt[(size_t)(row0 + r) * n_splits + split] = sCnt[r];
}

// One radix digit of a 256-bin histogram, scanned by wave 0 alone (4 bins per lane, shuffles only).  Finds the bin
// holding the `need`-th largest entry: out[0] = bin, out[1] = how many entries of that bin are needed, out[2] =
// entries in or above the bin.  Called by every thread; ends with a workgroup barrier.
__device__ __forceinline__ void sel_find_bin(const int *hist, int need, int tid, int *out)
{
    if (tid < 64) {
        const int4 v = *reinterpret_cast<const int4 *>(hist + 4 * tid);
        const int own = v.x + v.y + v.z + v.w;
        int inc = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_down(inc, o, 64);
            if (tid + o < 64) inc += n;
        }
        const int above = inc - own;
        if (above < need && need <= above + own) {
            int cum = above, bin = 4 * tid + 3, h = v.w;
            if (need > cum + h) { cum += h; bin = 4 * tid + 2; h = v.z; }
            if (need > cum + h) { cum += h; bin = 4 * tid + 1; h = v.y; }
            if (need > cum + h) { cum += h; bin = 4 * tid; h = v.x; }
            out[0] = bin; out[1] = need - cum; out[2] = cum + h;
        }
    }
    __syncthreads();
}

// Final selection: one 256-thread workgroup per user row.  The row's candidates (distinct 64-bit composites
// key << 32 | ~id, n <= kSelC) are staged in LDS; while more than 256 of them could still belong to the top K,
// 8-bit radix rounds over the composite (most significant byte first) narrow the set; then every thread
// rank-sorts one survivor by counting the composites above it -- (score desc, id asc) exactly like the oracle.
__global__ __launch_bounds__(256) void sel_finalize_kernel(const unsigned long long *__restrict__ cand, const int *__restrict__ cand_cnt,
                                                           int K, int *__restrict__ top_ids, float *__restrict__ top_scores, int n_splits,
                                                           const int *__restrict__ rank_part, int n_targets, int *__restrict__ target_rank)
{
    __shared__ unsigned long long ent[kSelC * kSelMaxSplits];
    __shared__ unsigned long long sel[256];
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ int sh[4], sOff[kSelMaxSplits + 1];
    const int tid = threadIdx.x, b = blockIdx.x;
    // the ranges' lists (each its range's exact top candidates) back to back; the target ranks are the sums of the ranges' counts
    if (tid == 0) {
        int o = 0;
        for (int q = 0; q < n_splits; ++q) { sOff[q] = o; o += min(cand_cnt[(size_t)b * n_splits + q], kSelC); }
        sOff[n_splits] = o;
        sh[3] = 0;
    }
    if (tid < n_targets) {
        int r = 0;
        for (int q = 0; q < n_splits; ++q) r += rank_part[((size_t)b * n_splits + q) * n_targets + tid];
        target_rank[(size_t)b * n_targets + tid] = r;
    }
    __syncthreads();
    const int n = sOff[n_splits];
    for (int q = 0; q < n_splits; ++q)
        for (int i = tid; i < sOff[q + 1] - sOff[q]; i += 256) ent[sOff[q] + i] = cand[((size_t)b * n_splits + q) * kSelC + i];
    __syncthreads();
    unsigned long long lower = 0ULL;   // survivors: composites >= lower
    int n_sel = n;
    if (n > 256) {
        unsigned long long prefix = 0ULL, mask = 0ULL;
        int need = K, shift = 64, n_cand = n;
        while (n_cand > 256 && shift > 0) {
            shift -= 8;
            hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                const unsigned long long c = ent[i];
                if ((c & mask) == prefix) atomicAdd(&hist[(int)((c >> shift) & 255ULL)], 1);
            }
            __syncthreads();
            sel_find_bin(hist, need, tid, sh);
            const int n_gt = K - need;          // composites strictly above the old prefix range
            prefix |= (unsigned long long)sh[0] << shift;
            mask |= 255ULL << shift;
            n_cand = n_gt + sh[2];
            need = sh[1];
            __syncthreads();
        }
        lower = prefix;
        n_sel = n_cand;   // <= 256 (composites are distinct: with all 64 bits resolved exactly K remain)
    }
    if (n <= 256) {
        if (tid < n) sel[tid] = ent[tid];
        if (tid == 0) sh[3] = n;
    } else {
        for (int i = tid; i < n; i += 256) {
            const unsigned long long c = ent[i];
            if (c >= lower && c != 0ULL) sel[atomicAdd(&sh[3], 1)] = c;
        }
    }
    __syncthreads();
    n_sel = sh[3];
    if (tid < n_sel) {
        const unsigned long long mine = sel[tid];
        int rank = 0;
        for (int j = 0; j < n_sel; ++j) rank += sel[j] > mine ? 1 : 0;
        if (rank < K) {
            top_ids[(size_t)b * K + rank] = (int)(~(unsigned)(mine & 0xffffffffULL));
            top_scores[(size_t)b * K + rank] = key_score((unsigned)(mine >> 32));
        }
    }
    for (int k = n_sel + tid; k < K; k += 256) {
        top_ids[(size_t)b * K + k] = -1;
        top_scores[(size_t)b * K + k] = -INFINITY;
    }
}

inline size_t sel_lds_bytes(int rb, int d)
{
    // the sweep needs the double-buffered item tile; the prologue aliases it with the user rows + target rows
    const size_t tile = 2 * (size_t)kSelTN * kSelLdB, pro = (size_t)rb * sel_row_stride(d) + (size_t)kSelMaxT * d;
    return sizeof(float) * (tile > pro ? tile : pro);
}

// Item ranges per row block (blockIdx.y): a function of the call's (nb, n_items) only, so that the scratch sizing and the
// launch agree.  ONE by default: measured on ml1m (5893 x 3702 x 64, round 3) the ranges do not pay -- 64-row workgroups with
// 1 / 2 / 3 / 8 ranges 330 / 238 / 266 / 281 us against 198 us for the 16-row form and 97 us for GEMM + selection: the sweep is
// bound by its per-tile epilogue and barriers, not by the number of workgroups.  RK_SEL_SPLITS=n (with RK_SEL_CONFIG=2) for tuning.
inline int sel_splits(int nb, int n_items)
{
    const char *es = getenv("RK_SEL_SPLITS");   // read per call (tests switch it)
    const int env = es ? atoi(es) : 0;
    const int n_tiles = (n_items + kSelTN - 1) / kSelTN;
    (void)nb;
    long long sp = env > 0 ? env : 1;
    sp = std::min<long long>(sp, std::max(1, n_tiles / 4));
    return (int)std::max<long long>(1, std::min<long long>(sp, kSelMaxSplits));
}
// floats of scratch per call: candidate slots (8 bytes each) + counts + per-range target rank counts
inline long long sel_scratch_floats(int nb, int n_items, int n_targets)
{
    const long long sp = sel_splits(nb, n_items);
    return (long long)nb * sp * (kSelC * 2 + 1 + std::max(n_targets, 1)) + 2;
}

// whether the fused path applies to a request (dim <= 128: the A operands of the whole sweep live in registers)
inline bool sel_supported(int n_items, int d, int K, int n_targets)
{
    return K >= 1 && K <= kSelMaxK && n_targets <= kSelMaxT && d >= 1 && d <= 128 && n_items >= 1;
}

template <int WM, int WAVES_M, int NTG>
inline void sel_launch_nch(const SelArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t s)
{
    const int nch = (a.d + kSelKC - 1) / kSelKC;
    if (nch <= 1) hipLaunchKernelGGL((score_select_kernel<WM, WAVES_M, NTG, 1>), grid, block, lds, s, a);
    else if (nch == 2) hipLaunchKernelGGL((score_select_kernel<WM, WAVES_M, NTG, 2>), grid, block, lds, s, a);
    else if (nch == 3) hipLaunchKernelGGL((score_select_kernel<WM, WAVES_M, NTG, 3>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((score_select_kernel<WM, WAVES_M, NTG, 4>), grid, block, lds, s, a);
}

inline hipError_t score_select_launch(SelArgs a, hipStream_t s)
{
    int bits = 1;
    while (bits < 32 && (1LL << bits) < (long long)a.n_items) ++bits;
    a.id_bits = bits;
    const char *fe = getenv("RK_SEL_CONFIG");   // tuning / tests: 1 = 16-row, 2 = 64-row workgroups (read per call)
    const int force = fe ? atoi(fe) : 0;
    // small user blocks: 16 rows per workgroup so that the chip is filled; large ones: 64 rows (4x less item traffic per flop)
    const bool small = force ? force == 1 : ((long long)(a.nb + 63) / 64 < 512);
    const bool one = a.n_targets <= 1;
    if (a.n_splits < 1 || a.n_splits > kSelMaxSplits) return hipErrorInvalidValue;
    if (small) {
        const dim3 grid((a.nb + 15) / 16, a.n_splits), block(256);
        if (one) sel_launch_nch<16, 1, 1>(a, grid, block, sel_lds_bytes(16, a.d), s);
        else sel_launch_nch<16, 1, kSelMaxT>(a, grid, block, sel_lds_bytes(16, a.d), s);
    } else {
        const dim3 grid((a.nb + 63) / 64, a.n_splits), block(512);
        if (one) sel_launch_nch<32, 2, 1>(a, grid, block, sel_lds_bytes(64, a.d), s);
        else sel_launch_nch<32, 2, kSelMaxT>(a, grid, block, sel_lds_bytes(64, a.d), s);
    }
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(sel_finalize_kernel, dim3(a.nb), dim3(256), 0, s, a.cand, a.cand_cnt, a.K, a.top_ids, a.top_scores, a.n_splits,
                       a.rank_part, a.n_targets, a.target_rank);
    return hipGetLastError();
}
