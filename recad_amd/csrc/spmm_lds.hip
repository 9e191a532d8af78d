// Plan builder and entry points of the LDS-resident sliced SpMM (spmm_lds.h).
#include <algorithm>
#include <vector>

#include "spmm_lds.h"

#include "host/lds_plan_host.h"

RK_EXPORT int rk_lds_plan_build_host(int32_t n_users, int32_t n_items, const int32_t *rowptr, const int32_t *col, const float *val,
                                     int32_t dim, int32_t n_cu, rk_lds_plan_t *out, int64_t *n_words, rk_lds_info *info)
{
    return lds_plan_build_host_impl(n_users, n_items, rowptr, col, val, dim, n_cu, out, n_words, info);
}

RK_EXPORT int rk_lds_plan_build(int32_t n_users, int32_t n_items, const int32_t *rowptr, const int32_t *col, const float *val,
                                int32_t dim, void *stream, rk_lds_plan_t *out, int64_t *n_words, rk_lds_info *info)
{
    if (n_users <= 0 || n_items <= 0 || !rowptr || !col || !out || !n_words || !info) RK_FAIL(RK_EINVAL, "rk_lds_plan_build: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const size_t N = (size_t)n_users + (size_t)n_items;
    std::vector<int32_t> rp(N + 1);
    RK_HIP(hipMemcpyAsync(rp.data(), rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, s));
    RK_HIP(hipStreamSynchronize(s));
    const size_t nnz = (size_t)rp[N];
    std::vector<int32_t> c(std::max<size_t>(nnz, 1));
    std::vector<float> v;
    if (nnz) RK_HIP(hipMemcpyAsync(c.data(), col, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, s));
    if (val && nnz) { v.resize(nnz); RK_HIP(hipMemcpyAsync(v.data(), val, sizeof(float) * nnz, hipMemcpyDeviceToHost, s)); }
    RK_HIP(hipStreamSynchronize(s));
    int dev = 0;
    RK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    RK_HIP(hipGetDeviceProperties(&p, dev));
    return rk_lds_plan_build_host(n_users, n_items, rp.data(), c.data(), val ? v.data() : nullptr, dim, p.multiProcessorCount, out, n_words, info);
}

RK_EXPORT int rk_lds_plan_words(rk_lds_plan_t plan, int32_t *host_out)
{
    if (!plan || !host_out) RK_FAIL(RK_EINVAL, "rk_lds_plan_words: bad arguments");
    memcpy(host_out, plan->words.data(), sizeof(int32_t) * plan->words.size());
    return RK_OK;
}

RK_EXPORT int rk_lds_plan_upload(rk_lds_plan_t plan, int32_t *dev, void *stream)
{
    if (!plan || !dev) RK_FAIL(RK_EINVAL, "rk_lds_plan_upload: bad arguments");
    if (reinterpret_cast<uintptr_t>(dev) & 15) RK_FAIL(RK_EINVAL, "rk_lds_plan_upload: the device buffer must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    RK_HIP(hipMemcpyAsync(dev, plan->words.data(), sizeof(int32_t) * plan->words.size(), hipMemcpyHostToDevice, s));
    RK_HIP(hipStreamSynchronize(s));
    return RK_OK;
}

RK_EXPORT int rk_lds_plan_destroy(rk_lds_plan_t plan)
{
    delete plan;
    return RK_OK;
}

__global__ void lds_set_coef_kernel(float *coef, float step_size, float bc2s)
{
    coef[0] = step_size;
    coef[1] = bc2s;
}

static int lds_info_of(const rk_lds_info *fi, LdsInfo *o, const char *who)
{
    if (!fi || fi->n_wg <= 0 || fi->lds_bytes <= 0 || fi->lds_bytes > kLdsMaxBytes - 64 || fi->dim <= 0) RK_FAIL(RK_EINVAL, "%s: bad plan info", who);
    o->n_wg = fi->n_wg; o->lds_bytes = fi->lds_bytes; o->lpa = fi->lpa; o->lpb = fi->lpb;
    o->U = fi->n_users; o->I = fi->n_items; o->d = fi->dim; o->lsu = fi->lsu; o->lsi = fi->lsi;
    o->wgx_ofs = fi->wgx_ofs; o->dinv_ofs = fi->dinv_ofs; o->perm0 = fi->perm0_ofs; o->perm1 = fi->perm1_ofs; o->mq_ofs = fi->mq_ofs;
    if (fi->wgx_ofs <= 0 || fi->dinv_ofs <= 0) RK_FAIL(RK_EINVAL, "%s: plan info without section offsets (built by an older library?)", who);
    return RK_OK;
}

static int lds_repack(const rk_lds_info *fi, float *rm, float *sl, int32_t n_arrays, int64_t stride, int to_sliced, void *stream, const char *who)
{
    LdsInfo li;
    int rc = lds_info_of(fi, &li, who);
    if (rc) return rc;
    if (!rm || !sl || n_arrays <= 0) RK_FAIL(RK_EINVAL, "%s: bad arguments", who);
    const LdsDims g{li.U, li.I, li.d, li.lsu, li.lsi};
    const long long n = (long long)(li.U + li.I) * (li.d / 4);
    const int grid = (int)std::min<long long>((n + 255) / 256, 2048);
    for (int32_t q0 = 0; q0 < n_arrays; q0 += 3) {
        LdsPackJob job;
        memset(&job, 0, sizeof(job));
        job.n = std::min<int32_t>(3, n_arrays - q0);
        for (int q = 0; q < job.n; ++q) { job.rm[q] = rm + (size_t)(q0 + q) * (size_t)stride; job.sl[q] = sl + (size_t)(q0 + q) * (size_t)stride; }
        hipLaunchKernelGGL(lds_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, job, to_sliced);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}

RK_EXPORT int rk_lds_pack(const rk_lds_info *info, const float *row_major, float *sliced, int32_t n_arrays, int64_t stride, void *stream)
{
    return lds_repack(info, const_cast<float *>(row_major), sliced, n_arrays, stride, 1, stream, "rk_lds_pack");
}

RK_EXPORT int rk_lds_unpack(const rk_lds_info *info, const float *sliced, float *row_major, int32_t n_arrays, int64_t stride, void *stream)
{
    return lds_repack(info, row_major, const_cast<float *>(sliced), n_arrays, stride, 0, stream, "rk_lds_unpack");
}

RK_EXPORT int rk_spmm_lds(const rk_lds_info *info, const int32_t *plan, const float *x, const rk_lds_epilogue *epi, void *stream)
{
    LdsInfo li;
    int rc = lds_info_of(info, &li, "rk_spmm_lds");
    if (rc) return rc;
    if (!plan || !x || !epi) RK_FAIL(RK_EINVAL, "rk_spmm_lds: bad arguments");
    if (epi->sum_out && !epi->sum_in) RK_FAIL(RK_EINVAL, "rk_spmm_lds: sum_out needs sum_in");
    hipStream_t s = (hipStream_t)stream;
    LdsArgs a;
    memset(&a, 0, sizeof(a));
    a.plan = plan; a.x = x; a.h = li.hdr();
    a.e.add = epi->add; a.e.y = epi->y; a.e.y_rm = epi->y_row_major; a.e.sum_in = epi->sum_in; a.e.sum_out = epi->sum_out;
    a.e.sum_rm = epi->sum_out_row_major; a.e.sum_scale = epi->sum_scale; a.e.zero1 = epi->zero1; a.e.zero2 = epi->zero2;
    a.e.stamps = reinterpret_cast<unsigned long long *>(epi->stamps);
    if (epi->adam_t > 0) {
        if (!epi->adam_p || !epi->adam_m || !epi->adam_v || !epi->coef_scratch) RK_FAIL(RK_EINVAL, "rk_spmm_lds: Adam pointers missing");
        const AdamCoef c = adam_coef(epi->adam_t, epi->lr, epi->beta1, epi->beta2);
        hipLaunchKernelGGL(lds_set_coef_kernel, dim3(1), dim3(1), 0, s, epi->coef_scratch, c.step_size, c.bc2s);
        RK_CHECK_LAUNCH();
        a.e.adam = 1; a.e.adam_rm = 1; a.e.p = epi->adam_p; a.e.m = epi->adam_m; a.e.v = epi->adam_v; a.e.shadow = epi->adam_shadow; a.e.coef = epi->coef_scratch;
        a.e.b1 = epi->beta1; a.e.b2 = epi->beta2; a.e.eps = epi->eps;
    }
    RK_HIP(spmm_lds_launch(li, a, s));
    return RK_OK;
}
