// Graph construction (COO->CSR, degree schedule, normalised adjacency), pairwise scores and
// the stand-alone SpMM entry point.
#include <algorithm>
#include <numeric>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "spmm.h"

RK_EXPORT int rk_abi_version(void) { return RK_ABI_VERSION; }
RK_EXPORT const char *rk_last_error(void) { return rk_err_buf; }

RK_EXPORT int rk_device_info(char *name, int32_t name_len, int32_t *cu_count)
{
    int dev = 0;
    RK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    RK_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (!strstr(p.gcnArchName, "gfx950")) RK_FAIL(RK_EINVAL, "device %s is not gfx950", p.gcnArchName);
    return RK_OK;
}

// rowptr[r] = first e with coo_row[e] >= r  (coo_row non-decreasing)
__global__ void csr_rowptr_kernel(int n_rows, long long nnz, const int64_t *__restrict__ row, int *__restrict__ rowptr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    long long lo = 0, hi = nnz;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (row[mid] < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = (int)lo;
}

__global__ void csr_copy_kernel(long long nnz, const int64_t *__restrict__ c64, const float *__restrict__ v, int *__restrict__ c32,
                                float *__restrict__ vout)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (long long)gridDim.x * blockDim.x) {
        c32[e] = (int)c64[e];
        vout[e] = v[e];
    }
}

RK_EXPORT int rk_coo_to_csr(int32_t n_rows, int64_t nnz, const int64_t *coo_row, const int64_t *coo_col,
                            const float *coo_val, int32_t *rowptr, int32_t *col, float *val, void *stream)
{
    if (n_rows <= 0 || nnz < 0 || nnz > 0x7fffffffLL) RK_FAIL(RK_EINVAL, "rk_coo_to_csr: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3((n_rows + 1 + 255) / 256), dim3(256), 0, s, n_rows, (long long)nnz, coo_row, rowptr);
    RK_CHECK_LAUNCH();
    if (nnz > 0) {
        const int grid = (int)std::min<long long>((nnz + 255) / 256, 4096);
        hipLaunchKernelGGL(csr_copy_kernel, dim3(grid), dim3(256), 0, s, (long long)nnz, coo_col, coo_val, col, val);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}

#include "host/csr_schedule_host.h"

RK_EXPORT int rk_csr_schedule_build_host(int32_t n_rows, const int32_t *rowptr, int32_t class_split, int32_t dim,
                                         rk_schedule_t *out, int32_t *n_blocks, int64_t *n_words, int64_t *scratch_words)
{
    return csr_schedule_build_host_impl(n_rows, rowptr, class_split, dim, out, n_blocks, n_words, scratch_words);
}

RK_EXPORT int rk_csr_schedule_build(int32_t n_rows, const int32_t *rowptr, int32_t class_split, int32_t dim, void *stream,
                                    rk_schedule_t *out, int32_t *n_blocks, int64_t *n_words, int64_t *scratch_words)
{
    if (n_rows <= 0 || !rowptr) RK_FAIL(RK_EINVAL, "rk_csr_schedule_build: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    std::vector<int32_t> rp((size_t)n_rows + 1);
    RK_HIP(hipMemcpyAsync(rp.data(), rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, s));
    RK_HIP(hipStreamSynchronize(s));
    return csr_schedule_build_host_impl(n_rows, rp.data(), class_split, dim, out, n_blocks, n_words, scratch_words);
}

// the schedule's words as _upload lays them out: [wave descriptors][header {n_long, n_slots, dim, n_packed}][workgroup metas][packed rows]
static std::vector<int32_t> schedule_words(const rk_schedule *sched)
{
    std::vector<int32_t> head(sched->desc);
    const int32_t hdr[4] = {sched->n_long, sched->n_slots, sched->dim, (int32_t)(sched->packed.size() / 4)};
    head.insert(head.end(), hdr, hdr + 4);
    head.insert(head.end(), sched->bmeta.begin(), sched->bmeta.end());
    head.insert(head.end(), sched->packed.begin(), sched->packed.end());
    return head;
}

RK_EXPORT int rk_csr_schedule_words(rk_schedule_t sched, int32_t *host_out)
{
    if (!sched || !host_out) RK_FAIL(RK_EINVAL, "rk_csr_schedule_words: bad arguments");
    const std::vector<int32_t> w = schedule_words(sched);
    memcpy(host_out, w.data(), sizeof(int32_t) * w.size());
    return RK_OK;
}

RK_EXPORT int rk_csr_schedule_upload(rk_schedule_t sched, int32_t *wave_desc, void *stream)
{
    if (!sched || !wave_desc) RK_FAIL(RK_EINVAL, "rk_csr_schedule_upload: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const std::vector<int32_t> head = schedule_words(sched);   // (read-only on the device)
    RK_HIP(hipMemcpyAsync(wave_desc, head.data(), sizeof(int32_t) * head.size(), hipMemcpyHostToDevice, s));
    RK_HIP(hipStreamSynchronize(s));
    return RK_OK;
}

RK_EXPORT int rk_csr_schedule_destroy(rk_schedule_t sched)
{
    delete sched;
    return RK_OK;
}

// ---- D^-1/2 A D^-1/2 on device (implicit.py:259-277, fp32 like numpy>=2 computes it)
__global__ void item_count_kernel(long long E, const int *__restrict__ ridx, int *__restrict__ icnt)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x)
        atomicAdd(&icnt[ridx[e]], 1);
}

// single-block exclusive scan of icnt[0..I) into rowptr[U+1 .. U+I], on top of the user part
__global__ void adj_rowptr_kernel(int U, int I, const int *__restrict__ rptr, const int *__restrict__ icnt, int *__restrict__ rowptr)
{
    __shared__ int carry;
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int u = tid; u <= U; u += blockDim.x) rowptr[u] = rptr[u];
    if (tid == 0) carry = rptr[U];
    __syncthreads();
    for (int base = 0; base < I; base += blockDim.x) {
        const int i = base + tid;
        int v = i < I ? icnt[i] : 0, x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int pre = 0;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        int tot = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += wsum[k];
        if (i < I) rowptr[U + i + 1] = carry + pre + x;
        __syncthreads();
        if (tid == 0) carry += tot;
        __syncthreads();
    }
}

__device__ __forceinline__ float dinv_f(int deg)
{
    const float v = powf((float)deg + 1e-14f, -0.5f);
    return isinf(v) ? 0.f : v;
}

// user rows: direct; item rows: each item row collects its users in increasing user order.
__global__ void adj_user_rows_kernel(int U, const int *__restrict__ rptr, const int *__restrict__ ridx, const int *__restrict__ rowptr,
                                     int *__restrict__ col, float *__restrict__ val)
{
    const int u = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= U) return;
    const int b = rptr[u], e = rptr[u + 1];
    const float du = dinv_f(e - b);
    for (int k = b + lane; k < e; k += 64) {
        const int i = ridx[k];
        const int ideg = rowptr[U + i + 1] - rowptr[U + i];
        col[k] = U + i;
        val[k] = (du * 1.0f) * dinv_f(ideg);
    }
}

// item rows = transpose of the user rows: radix-sort the edge keys (item << 32 | user); edge e of
// the sorted list is entry e of the item block (users ascending inside an item => deterministic).
__global__ void adj_edge_keys_kernel(int U, const int *__restrict__ rptr, const int *__restrict__ ridx, unsigned long long *__restrict__ keys)
{
    const int u = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= U) return;
    for (int k = rptr[u] + lane; k < rptr[u + 1]; k += 64) keys[k] = ((unsigned long long)(unsigned)ridx[k] << 32) | (unsigned)u;
}

__global__ void adj_item_rows_kernel(int U, long long E, const unsigned long long *__restrict__ keys, const int *__restrict__ rptr,
                                     const int *__restrict__ rowptr, int *__restrict__ col, float *__restrict__ val)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long long)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[e];
        const int i = (int)(k >> 32), u = (int)(k & 0xffffffffULL);
        col[E + e] = u;
        val[E + e] = (dinv_f(rowptr[U + i + 1] - rowptr[U + i]) * 1.0f) * dinv_f(rptr[u + 1] - rptr[u]);
    }
}

RK_EXPORT int rk_build_norm_adj(int32_t n_users, int32_t n_items, const int32_t *r_ptr, const int32_t *r_idx,
                                int32_t *rowptr, int32_t *col, float *val, int32_t *tmp, void *stream)
{
    if (n_users <= 0 || n_items <= 0 || !r_ptr || !r_idx || !rowptr || !col || !val || !tmp)
        RK_FAIL(RK_EINVAL, "rk_build_norm_adj: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int32_t E = 0;
    RK_HIP(hipMemcpyAsync(&E, r_ptr + n_users, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    RK_HIP(hipStreamSynchronize(s));
    RK_HIP(hipMemsetAsync(tmp, 0, sizeof(int32_t) * ((size_t)n_items + 1), s));
    if (E > 0) {
        const int grid = (int)std::min<long long>(((long long)E + 255) / 256, 4096);
        hipLaunchKernelGGL(item_count_kernel, dim3(grid), dim3(256), 0, s, (long long)E, r_idx, tmp);
        RK_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(adj_rowptr_kernel, dim3(1), dim3(1024), 0, s, n_users, n_items, r_ptr, tmp, rowptr);
    RK_CHECK_LAUNCH();
    hipLaunchKernelGGL(adj_user_rows_kernel, dim3((n_users + 3) / 4), dim3(256), 0, s, n_users, r_ptr, r_idx, rowptr, col, val);
    RK_CHECK_LAUNCH();
    if (E > 0) {
        unsigned long long *keys = nullptr, *sorted = nullptr;
        void *tmp_sort = nullptr;
        size_t tmp_bytes = 0;
        int end_bit = 32;
        while (end_bit < 64 && (1LL << (end_bit - 32)) < (long long)n_items) ++end_bit;
        RK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys, sorted, (int)E, 0, end_bit, s));
        RK_HIP(hipMallocAsync((void **)&keys, sizeof(unsigned long long) * (size_t)E, s));
        RK_HIP(hipMallocAsync((void **)&sorted, sizeof(unsigned long long) * (size_t)E, s));
        RK_HIP(hipMallocAsync(&tmp_sort, tmp_bytes, s));
        hipLaunchKernelGGL(adj_edge_keys_kernel, dim3((n_users + 3) / 4), dim3(256), 0, s, n_users, r_ptr, r_idx, keys);
        RK_CHECK_LAUNCH();
        RK_HIP(hipcub::DeviceRadixSort::SortKeys(tmp_sort, tmp_bytes, keys, sorted, (int)E, 0, end_bit, s));
        const int grid = (int)std::min<long long>(((long long)E + 255) / 256, 8192);
        hipLaunchKernelGGL(adj_item_rows_kernel, dim3(grid), dim3(256), 0, s, n_users, (long long)E, sorted, r_ptr, rowptr, col, val);
        RK_CHECK_LAUNCH();
        RK_HIP(hipFreeAsync(keys, s));
        RK_HIP(hipFreeAsync(sorted, s));
        RK_HIP(hipFreeAsync(tmp_sort, s));
    }
    return RK_OK;
}

// ---- pairwise scores
__global__ void pair_scores_kernel(int d, const float *__restrict__ utab, const float *__restrict__ itab, const float *ubias,
                                   const float *ibias, float mean, const int64_t *__restrict__ users,
                                   const int64_t *__restrict__ items, long long n, float *__restrict__ out,
                                   unsigned drop_thresh24, float drop_scale, unsigned long long drop_seed)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long b = wave; b < n; b += n_waves) {
        const long long u = users[b], i = items[b];
        const float *pu = utab + (size_t)u * d, *pi = itab + (size_t)i * d;
        float s = 0.f;
        for (int k = lane; k < d; k += 64) s += pu[k] * pi[k];
        s = wave_sum(s);
        if (lane == 0) {
            if (ubias) s = ((s + ubias[u]) + ibias[i]) + mean;
            if (drop_thresh24) s = rk_drop_keep(drop_seed, (unsigned)b, drop_thresh24) ? s * drop_scale : 0.f;
            out[b] = s;
        }
    }
}

RK_EXPORT int rk_pair_scores(int32_t dim, const float *utab, const float *itab, const float *ubias, const float *ibias,
                             float mean, const int64_t *users, const int64_t *items, int64_t n, float *out, float dropout,
                             uint64_t drop_seed, void *stream)
{
    if (n <= 0) return RK_OK;
    if (!(dropout >= 0.f) || dropout >= 1.f) RK_FAIL(RK_EINVAL, "rk_pair_scores: dropout must be in [0, 1)");
    if (dim <= 0 || !utab || !itab || !users || !items || !out) RK_FAIL(RK_EINVAL, "rk_pair_scores: bad arguments");
    if ((ubias == nullptr) != (ibias == nullptr)) RK_FAIL(RK_EINVAL, "rk_pair_scores: give both biases or neither");
    const int grid = (int)std::min<long long>((n + 3) / 4, 4096);
    hipLaunchKernelGGL(pair_scores_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dim, utab, itab, ubias, ibias, mean,
                       users, items, (long long)n, out, dropout > 0.f ? (unsigned)((1.0 - (double)dropout) * 16777216.0) : 0u,
                       dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.f, (unsigned long long)drop_seed);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// ---- row gather / row zero (the row-sharded trainer's per-step index work)
__global__ void rows_gather_masked_kernel(int d, const float *__restrict__ src, const int64_t *__restrict__ idx,
                                          const float *__restrict__ mask, long long n, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long i = wave; i < n; i += n_waves) {
        const float m = mask ? mask[i] : 1.f;
        const float *row = src + (size_t)idx[i] * d;
        for (int k = lane; k < d; k += 64) out[(size_t)i * d + k] = m != 0.f ? row[k] * m : 0.f;
    }
}

__global__ void rows_zero_kernel(int d, float *__restrict__ a, float *__restrict__ b, const int64_t *__restrict__ idx, long long n)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long i = wave; i < n; i += n_waves) {
        const size_t o = (size_t)idx[i] * d;
        for (int k = lane; k < d; k += 64) { a[o + k] = 0.f; if (b) b[o + k] = 0.f; }
    }
}

__global__ void rows_mark_bits_kernel(unsigned *bits, const int64_t *__restrict__ idx, long long n, int set)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = idx[i];
        if (set) atomicOr(&bits[r >> 5], 1u << (r & 31)); else bits[r >> 5] = 0u;
    }
}

RK_EXPORT int rk_rows_mark_bits(uint32_t *bits, const int64_t *idx, int64_t n, int32_t set, void *stream)
{
    if (n <= 0) return RK_OK;
    if (!bits || !idx) RK_FAIL(RK_EINVAL, "rk_rows_mark_bits: bad arguments");
    hipLaunchKernelGGL(rows_mark_bits_kernel, dim3((int)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, bits, idx,
                       (long long)n, (int)set);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

RK_EXPORT int rk_rows_gather_masked(int32_t dim, const float *src, const int64_t *idx, const float *mask, int64_t n, float *out, void *stream)
{
    if (n <= 0) return RK_OK;
    if (dim <= 0 || !src || !idx || !out) RK_FAIL(RK_EINVAL, "rk_rows_gather_masked: bad arguments");
    const int grid = (int)std::min<long long>((n + 3) / 4, 4096);
    hipLaunchKernelGGL(rows_gather_masked_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dim, src, idx, mask, (long long)n, out);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

RK_EXPORT int rk_rows_zero(int32_t dim, float *a, float *b, const int64_t *idx, int64_t n, void *stream)
{
    if (n <= 0) return RK_OK;
    if (dim <= 0 || !a || !idx) RK_FAIL(RK_EINVAL, "rk_rows_zero: bad arguments");
    const int grid = (int)std::min<long long>((n + 3) / 4, 4096);
    hipLaunchKernelGGL(rows_zero_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dim, a, b, idx, (long long)n);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

RK_EXPORT int rk_spmm_csr(int32_t n_rows, const int32_t *rowptr, const int32_t *col, const float *val,
                          const int32_t *wave_desc, int32_t n_blocks, int32_t *scratch, int32_t dim, const float *x,
                          const float *add, float *y, void *stream)
{
    if (n_rows <= 0 || dim <= 0 || dim > 256 || !rowptr || !col || !val || !wave_desc || n_blocks <= 0 || !x || !y)
        RK_FAIL(RK_EINVAL, "rk_spmm_csr: bad arguments");
    SpmmArgs a;
    memset(&a, 0, sizeof(a));
    a.n_rows = n_rows; a.rowptr = rowptr; a.col = col; a.val = val; a.wave_desc = reinterpret_cast<const int4 *>(wave_desc); a.n_blocks = n_blocks; a.d = dim;
    if ((size_t)n_rows * dim * sizeof(float) >= (1ULL << 32)) RK_FAIL(RK_EINVAL, "rk_spmm_csr: n_rows*dim*4 must be < 4 GiB");
    if ((n_blocks & kSchedLongFlag) && !scratch) RK_FAIL(RK_EINVAL, "rk_spmm_csr: this schedule has long rows and needs its scratch block");
    a.scratch = scratch;
    a.x = x;
    a.e.add = add;
    a.e.y = y;
    RK_HIP(spmm_launch(a, (hipStream_t)stream));
    return RK_OK;
}
