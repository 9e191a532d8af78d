// Full-catalog scoring + top-K + target rank (recad/workflow/normal.py:57-93) on gfx950.
//   pass 1: scores[nb, I] = U_b . Items^T with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); the
//           instruction is a k-ordered fmaf chain, so a score is bit-identical to the scalar
//           loop  s = fmaf(u[k], v[k], s), k = 0..d-1  (oracle: orc_score_rows).
//   pass 2: one workgroup per user: mask seen items, rank of each target, radix-select of
//           the K-th largest score, ordered collection, bitonic sort by (score desc, id asc).
#include <algorithm>

#include "gemm.h"

// ---------------------------------------------------------------- pass 2
__device__ __forceinline__ unsigned score_key(float s)
{
    // monotone float -> uint; 0 is reserved for "excluded" (seen item, encoded as -inf)
    if (s == -INFINITY) return 0u;
    const unsigned u = __float_as_uint(s);
    const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return k == 0u ? 1u : k;
}

static constexpr int kMaxK = 256;

// LDS_ROW: the user's score row is staged once in (dynamic) LDS and every pass reads it from
// there; the scores matrix is then left untouched.  Otherwise passes stream the row from global
// memory / L2 and seen items are overwritten with -inf in place.
// NT = threads per user row: 256 for long rows, 64 (one wave, barriers degenerate) for short ones.
template <bool LDS_ROW, int NT>
__global__ __launch_bounds__(NT) void topk_rows_kernel(float *__restrict__ scores, int n_items, const int *__restrict__ user_ids,
                                                        const int *__restrict__ seen_ptr, const int *__restrict__ seen_idx, int K,
                                                        int *__restrict__ top_ids, float *__restrict__ top_scores,
                                                        const int *__restrict__ targets, int n_targets,
                                                        float *__restrict__ target_score, int *__restrict__ target_rank)
{
    __shared__ int hist[256];
    __shared__ unsigned long long sel[kMaxK];
    __shared__ int sh_i[8];
    constexpr int NW = NT / 64;
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    extern __shared__ __attribute__((aligned(16))) float lds_row[];
    const int b = blockIdx.x;
    float *grow = scores + (size_t)b * n_items;
    float *row = LDS_ROW ? lds_row : grow;
    const int u = user_ids[b];
    // target scores before masking (normal.py:83-85)
    for (int t = tid; t < n_targets; t += NT) target_score[(size_t)b * n_targets + t] = grow[targets[t]];
    if (LDS_ROW) {
        if ((((uintptr_t)grow) & 15) == 0) {
            for (int i = tid * 4; i + 3 < n_items; i += NT * 4) *reinterpret_cast<float4 *>(row + i) = *reinterpret_cast<const float4 *>(grow + i);
            for (int i = (n_items & ~3) + tid; i < n_items; i += NT) row[i] = grow[i];
        } else {
            for (int i = tid; i < n_items; i += NT) row[i] = grow[i];
        }
    }
    __syncthreads();
    for (int k = seen_ptr[u] + tid; k < seen_ptr[u + 1]; k += NT) row[seen_idx[k]] = -INFINITY;
    __threadfence_block();
    __syncthreads();
    // rank of every target among the unseen items: #(s > st) + #(s == st and id < target)
    for (int t = 0; t < n_targets; ++t) {
        const int tg = targets[t];
        const float st = target_score[(size_t)b * n_targets + t];
        int c = 0;
        for (int i = tid; i < n_items; i += NT) {
            const float s = row[i];
            if (s == -INFINITY || i == tg) continue;
            c += (s > st || (s == st && i < tg)) ? 1 : 0;
        }
        c = (int)wave_sum((float)c);  // counts < 2^24: exact in fp32
        if (lane == 0) wtot[w] = c;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int q = 0; q < NW; ++q) tot += wtot[q];
            target_rank[(size_t)b * n_targets + t] = tot;
        }
        __syncthreads();
    }
    // radix select: K-th largest key among the valid ones
    unsigned prefix = 0u, mask = 0u;
    int need = K;
    bool take_all = false;
    for (int round = 0; round < 4; ++round) {
        const int shift = 24 - 8 * round;
        for (int q = tid; q < 256; q += NT) hist[q] = 0;
        __syncthreads();
        for (int i = tid; i < n_items; i += NT) {
            const unsigned k = score_key(row[i]);
            if (k != 0u && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int cum = 0, digit = -1, nd = need;
            for (int bin = 255; bin >= 0; --bin) {
                if (cum + hist[bin] >= need) { digit = bin; nd = need - cum; break; }
                cum += hist[bin];
            }
            sh_i[0] = digit;
            sh_i[1] = nd;
        }
        __syncthreads();
        if (sh_i[0] < 0) { take_all = true; break; }  // fewer than K valid items
        prefix |= (unsigned)sh_i[0] << shift;
        mask |= 255u << shift;
        need = sh_i[1];
        __syncthreads();
    }
    const unsigned T = take_all ? 1u : prefix;
    const int n_gt_slots = take_all ? K : K - need;
    // collect: keys > T anywhere in [0, n_gt_slots), keys == T (lowest ids first) after them
    if (tid == 0) { sh_i[2] = 0; sh_i[3] = 0; }
    for (int k = tid; k < kMaxK; k += NT) sel[k] = 0ULL;
    __syncthreads();
    for (int base = 0; base < n_items; base += NT) {
        const int i = base + tid;
        const unsigned k = i < n_items ? score_key(row[i]) : 0u;
        const bool gt = take_all ? (k != 0u) : (k > T);
        const bool eq = !take_all && k == T && k != 0u;
        if (gt) {
            const int p = atomicAdd(&sh_i[2], 1);
            if (p < kMaxK) sel[p] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)i);
        }
        const unsigned long long m = __ballot(eq);
        if (lane == 0) wtot[w] = __popcll(m);
        __syncthreads();
        int pre = __popcll(m & ((1ULL << lane) - 1ULL));
        for (int ww = 0; ww < w; ++ww) pre += wtot[ww];
        const int eq_base = sh_i[3];
        if (eq) {
            const int idx = eq_base + pre;
            if (idx < need) sel[n_gt_slots + idx] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)i);
        }
        __syncthreads();
        if (tid == 0) {
            int tot = eq_base;
            for (int q = 0; q < NW; ++q) tot += wtot[q];
            sh_i[3] = tot;
        }
        __syncthreads();
    }
    // bitonic sort of 256 composite keys, descending => score desc, item id asc
    for (int size = 2; size <= kMaxK; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int q = tid; q < kMaxK; q += NT) {
                const int partner = q ^ stride;
                if (partner > q) {
                    const bool desc = (q & size) == 0;
                    const unsigned long long x = sel[q], y = sel[partner];
                    if (desc ? (x < y) : (x > y)) { sel[q] = y; sel[partner] = x; }
                }
            }
            __syncthreads();
        }
    for (int k = tid; k < K; k += NT) {
        const unsigned long long e = sel[k];
        if (e == 0ULL) {
            top_ids[(size_t)b * K + k] = -1;
            top_scores[(size_t)b * K + k] = -INFINITY;
        } else {
            const int id = (int)(~(unsigned)(e & 0xffffffffULL));
            top_ids[(size_t)b * K + k] = id;
            top_scores[(size_t)b * K + k] = row[id];
        }
    }
}

extern "C" int rk_topk_rows_impl(float *scores, int nb, int n_items, const int *user_ids, const int *seen_ptr,
                                 const int *seen_idx, int K, int *top_ids, float *top_scores, const int *targets,
                                 int n_targets, float *target_score, int *target_rank, hipStream_t s)
{
    if (K <= 0 || K > kMaxK) RK_FAIL(RK_EINVAL, "top-K: K must be in [1,%d]", kMaxK);
    if (n_targets < 0 || n_targets > 256 || (n_targets > 0 && (!targets || !target_score || !target_rank)))
        RK_FAIL(RK_EINVAL, "top-K: bad targets");
    const size_t row_bytes = ((size_t)n_items * sizeof(float) + 15) & ~(size_t)15;
    // LDS staging only pays while several workgroups still fit per CU (measured: a 138 KB row in LDS
    // is 2.8x SLOWER than streaming it from L2 -- one 4-wave workgroup per CU); ml1m-size rows tie.
    if (row_bytes <= 16 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            RK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(topk_rows_kernel<true, 256>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_set = true;
        }
        // (one wave per short row -- NT = 64 -- was measured too: 333 vs 250 us for 5950 x 3702)
        hipLaunchKernelGGL((topk_rows_kernel<true, 256>), dim3(nb), dim3(256), row_bytes, s, scores, n_items, user_ids, seen_ptr, seen_idx,
                           K, top_ids, top_scores, targets, n_targets, target_score, target_rank);
    } else {
        hipLaunchKernelGGL((topk_rows_kernel<false, 256>), dim3(nb), dim3(256), 0, s, scores, n_items, user_ids, seen_ptr, seen_idx, K,
                           top_ids, top_scores, targets, n_targets, target_score, target_rank);
    }
    RK_CHECK_LAUNCH();
    return RK_OK;
}

RK_EXPORT int rk_score_topk(int32_t dim, const float *urows, int32_t nb, const int32_t *user_ids, const float *itab,
                            int32_t n_items, const float *ubias_rows, const float *ibias, float mean,
                            const int32_t *seen_ptr, const int32_t *seen_idx, int32_t K, int32_t *top_ids,
                            float *top_scores, const int32_t *targets, int32_t n_targets, float *target_score,
                            int32_t *target_rank, float *scratch, void *stream)
{
    if (nb <= 0) return RK_OK;
    if (dim <= 0 || n_items <= 0 || !urows || !itab || !user_ids || !seen_ptr || !seen_idx || !scratch || !top_ids || !top_scores)
        RK_FAIL(RK_EINVAL, "rk_score_topk: bad arguments");
    if ((ubias_rows == nullptr) != (ibias == nullptr)) RK_FAIL(RK_EINVAL, "rk_score_topk: give both biases or neither");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = nb; g.N = n_items; g.K = dim;
    g.A = urows; g.a_rs = dim; g.a_cs = 1;
    g.B = itab; g.b_rs = dim; g.b_cs = 1;
    g.C = scratch; g.ldc = n_items;
    g.row_bias = ubias_rows; g.col_bias = ibias; g.const_add = mean;
    RK_HIP(gemm_f32_launch(g, s));
    return rk_topk_rows_impl(scratch, nb, n_items, user_ids, seen_ptr, seen_idx, K, top_ids, top_scores, targets, n_targets,
                             target_score, target_rank, s);
}
