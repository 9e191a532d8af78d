// Device-side samplers: the BPR triplet sampler and the pointwise
// negative sampler of recad/dataset/implicit.py:50-91 without the per-draw Python loop.
// Counter-based RNG (splitmix64 of seed/draw/attempt): reproducible for a given seed, the
// stream differs from numpy's (distributional parity is what the tests check).
#include "common.h"

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) { return rk_mix64(z); }
__device__ __forceinline__ unsigned long long rnd(unsigned long long seed, unsigned long long draw, unsigned k)
{
    return mix64(mix64(seed ^ (draw * 0xD1342543DE82EF95ULL)) + k);
}
// uniform integer in [0, n) from 64 random bits (multiply-shift, no modulo bias to speak of)
__device__ __forceinline__ unsigned bounded(unsigned long long r, unsigned n) { return (unsigned)(((r >> 32) * (unsigned long long)n) >> 32); }

__device__ __forceinline__ int lower_bound(const int *__restrict__ a, int lo, int hi, int x)
{
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// r-th (0-based) item NOT in the sorted list idx[b, e): exact, O(log deg).  idx[b+k] - k is the number of free
// items below positive k and is non-decreasing in k, so j = #{k : idx[b+k] - k <= r} positives precede the answer.
__device__ __forceinline__ int rth_free_item(const int *__restrict__ idx, int b, int e, int r)
{
    int lo = 0, hi = e - b;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (idx[b + mid] - mid <= r) lo = mid + 1; else hi = mid; }
    return r + lo;
}

// pairwise_sample (implicit.py:50-74): uniform user with replacement, skipped when the user has no
// positives (valid=0), uniform positive, negative rejection-sampled outside the positives.
__global__ void bpr_sample_kernel(int n_users, int n_items, const int *__restrict__ ptr, const int *__restrict__ idx,
                                  long long n_draws, unsigned long long seed, int64_t *__restrict__ users,
                                  int64_t *__restrict__ pos, int64_t *__restrict__ neg, int *__restrict__ valid)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n_draws; t += (long long)gridDim.x * blockDim.x) {
        const int u = (int)bounded(rnd(seed, (unsigned long long)t, 0), (unsigned)n_users);
        const int b = ptr[u], e = ptr[u + 1], deg = e - b;
        users[t] = u;
        if (deg == 0 || deg >= n_items) { valid[t] = 0; pos[t] = 0; neg[t] = 0; continue; }
        pos[t] = idx[b + (int)bounded(rnd(seed, (unsigned long long)t, 1), (unsigned)deg)];
        int ng = 0;
        bool ok = false;
        for (unsigned k = 2; k < 66 && !ok; ++k) {
            ng = (int)bounded(rnd(seed, (unsigned long long)t, k), (unsigned)n_items);
            const int p = lower_bound(idx, b, e, ng);
            ok = !(p < e && idx[p] == ng);
        }
        if (!ok) {  // dense user: pick the r-th free item directly
            const int r = (int)bounded(rnd(seed, (unsigned long long)t, 66), (unsigned)(n_items - deg));
            ng = rth_free_item(idx, b, e, r);
        }
        neg[t] = ng;
        valid[t] = 1;
    }
}

RK_EXPORT int rk_bpr_sample(int32_t n_users, int32_t n_items, const int32_t *pos_ptr, const int32_t *pos_idx,
                            int64_t n_draws, uint64_t seed, int64_t *users, int64_t *pos, int64_t *neg, int32_t *valid,
                            void *stream)
{
    if (n_draws <= 0) return RK_OK;
    if (n_users <= 0 || n_items <= 0 || !pos_ptr || !pos_idx || !users || !pos || !neg || !valid)
        RK_FAIL(RK_EINVAL, "rk_bpr_sample: bad arguments");
    const int grid = (int)((n_draws + 255) / 256 < 4096 ? (n_draws + 255) / 256 : 4096);
    hipLaunchKernelGGL(bpr_sample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n_users, n_items, pos_ptr, pos_idx,
                       (long long)n_draws, (unsigned long long)seed, users, pos, neg, valid);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// pointwise_sample (implicit.py:77-91): for every train edge (u, i): the row (u, i, 1) and `ratio`
// rows (u, j, 0) with j uniform (with replacement) over the items u has NOT interacted with.
// Output rows: edge e -> positions e*(ratio+1) .. +ratio.
__global__ void pointwise_sample_kernel(int n_users, int n_items, const int *__restrict__ ptr, const int *__restrict__ idx,
                                        int ratio, unsigned long long seed, int64_t *__restrict__ users,
                                        int64_t *__restrict__ items, int64_t *__restrict__ labels)
{
    const long long E = ptr[n_users];
    const long long total = E * (ratio + 1);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long e = t / (ratio + 1);
        const int k = (int)(t % (ratio + 1));
        // user of edge e: last u with ptr[u] <= e
        int lo = 0, hi = n_users;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ptr[mid] <= e) lo = mid; else hi = mid - 1; }
        const int u = lo, b = ptr[u], en = ptr[u + 1], deg = en - b;
        users[t] = u;
        if (k == 0) { items[t] = idx[e]; labels[t] = 1; continue; }
        const int free_items = n_items - deg;
        int ng = 0;
        if (free_items > 0) {
            const int r = (int)bounded(rnd(seed, (unsigned long long)t, 0), (unsigned)free_items);
            ng = rth_free_item(idx, b, en, r);
        }
        items[t] = ng;
        labels[t] = 0;
    }
}

RK_EXPORT int rk_pointwise_sample(int32_t n_users, int32_t n_items, const int32_t *train_ptr, const int32_t *train_idx,
                                  int64_t n_edges, int32_t negative_ratio, uint64_t seed, int64_t *users, int64_t *items,
                                  int64_t *labels, void *stream)
{
    if (n_edges <= 0) return RK_OK;
    if (n_users <= 0 || n_items <= 0 || negative_ratio < 0 || !train_ptr || !train_idx || !users || !items || !labels)
        RK_FAIL(RK_EINVAL, "rk_pointwise_sample: bad arguments");
    const long long total = (long long)n_edges * (negative_ratio + 1);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pointwise_sample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n_users, n_items, train_ptr,
                       train_idx, negative_ratio, (unsigned long long)seed, users, items, labels);
    RK_CHECK_LAUNCH();
    return RK_OK;
}
