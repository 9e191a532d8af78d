// CSR SpMM for gfx950 with fused epilogues (layer sum, scaled output, addend, self-cleaning
// gradient buffers, dense Adam).  Replaces torch.sparse.mm at
// recad/model/victim/lightgcn.py:107 and its autograd twin.
//
// Mapping (DESIGN.md "SpMM"): 1024-thread workgroups = 16 waves.  A host-built schedule
// (rk_csr_schedule_*) cuts every row into segments of <= 64 nonzeros and packs segments of
// whole rows into workgroups of 16 (first-fit decreasing), one segment per wave, so every
// wave has one short dependent chain: 64 column/value entries read coalesced, broadcast with
// ds_bpermute, and up to 16 independent 16-byte gathers in flight.  Segments of one row sit
// in consecutive waves; they combine through LDS in fixed order and the row's first wave runs
// the epilogue.  Rows longer than 1024 nonzeros get a workgroup of their own and its waves
// loop.  Inside a wave a row of D floats is covered by G = D/4 lanes with one 16-byte load
// each, so one global_load_dwordx4 gathers 64/G different X rows.  Summation order is fixed
// by the schedule => bit-reproducible.
#pragma once
#include <stdlib.h>

#include "common.h"

struct SpmmEpi {
    // v = acc (+ add[r])
    const float *add;
    float *y;  // nullable: y[r] = v
    // sum_out[r] = (sum_in[r] + v) * sum_scale   (nullable sum_out)
    const float *sum_in;
    float *sum_out;
    float sum_scale;
    float *zero1, *zero2;  // nullable: rows set to 0 after the addend was read
    // Adam on p[r] with gradient v
    int adam;
    float *p, *m, *v;
    const float *coef;  // {step_size, bc2s}
    float b1, b2, eps;
    int *state;  // bump words ST_STEP_BASE / ST_ADAM_T by `bump` (last kernel of a chunk)
    int bump;
};

struct SpmmArgs {
    int n_rows;
    const int *rowptr, *col;
    const float *val;
    const int4 *wave_desc;  // per wave: {row, e_begin, e_end, n_segments if row leader else 0}
    int n_blocks;
    int d;
    const float *x;  // [n_rows, d] row-major; n_rows*d*4 < 4 GiB (32-bit byte offsets)
    int dbg;
    // LDS hot-row variant (spmm_csr_hot_kernel); hot_H == 0 disables it
    const int *col_tagged;  // col with bit 31 set => low bits are an LDS slot of the row's class
    const int *hot_rows;    // [2][hot_H] node ids staged in LDS by class-0 / class-1 workgroups
    const int *pblocks;     // schedule workgroup indices: class 0 first (nb_class0), then class 1
    int hot_H, nb_class0, two_classes;
    SpmmEpi e;
};

static constexpr int kSpmmWavesMax = 16;
// waves per workgroup (schedule and launch must agree); RK_SPMM_WAVES overrides for tuning
inline int spmm_waves()
{
    static const int w = getenv("RK_SPMM_WAVES") ? atoi(getenv("RK_SPMM_WAVES")) : 8;
    return (w == 4 || w == 8 || w == 16) ? w : 8;
}

static constexpr int kSegNnz = 64;  // default nonzeros per schedule segment (RK_SEG_NNZ overrides, tuning only)

__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc)
{
    acc.x += a * x.x; acc.y += a * x.y; acc.z += a * x.z; acc.w += a * x.w;
    return acc;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int D, int UN>
__device__ __forceinline__ float4 gather_round(float4 acc, int c, float a, int n, int t0, const float *__restrict__ x,
                                               int grp, int sub)
{
    constexpr int NG = 64 / (D / 4);
    float4 xv[UN];
    float av[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const int src = (t0 + j) * NG + grp;
        int cc = __shfl(c, src & 63, 64);
        float aa = __shfl(a, src & 63, 64);
        const bool ok = src < n;
        cc = ok ? cc : 0;
        av[j] = ok ? aa : 0.f;
        xv[j] = *reinterpret_cast<const float4 *>(x + (unsigned)(cc * D + sub * 4));
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) acc = f4_fma(av[j], xv[j], acc);
    return acc;
}

// Partial sum of row segment [eb, ee) for the D/4 lanes that share `sub`; after the
// cross-group reduction every lane holds the total for its float4 slot.
template <int D, int UNMAX>
__device__ __forceinline__ float4 spmm_segment(const int *__restrict__ col, const float *__restrict__ val, int eb, int ee,
                                               const float *__restrict__ x, int lane)
{
    constexpr int G = D / 4, NG = 64 / G;
    const int grp = lane / G, sub = lane % G;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c_next = 0;
    float a_next = 0.f;
    if (eb + lane < ee) { c_next = col[eb + lane]; a_next = val[eb + lane]; }
    for (int base = eb; base < ee; base += 64) {
        const int n = min(64, ee - base);
        const int c = c_next;
        const float a = a_next;
        c_next = 0; a_next = 0.f;
        if (base + 64 + lane < ee) { c_next = col[base + 64 + lane]; a_next = val[base + 64 + lane]; }
        const int iters = (n + NG - 1) / NG;
        int t = 0;
        for (; t + UNMAX <= iters; t += UNMAX) acc = gather_round<D, UNMAX>(acc, c, a, n, t, x, grp, sub);
        const int rem = iters - t;
        if (UNMAX > 8 && rem > 8) acc = gather_round<D, UNMAX>(acc, c, a, n, t, x, grp, sub);
        else if (rem > 4) acc = gather_round<D, 8>(acc, c, a, n, t, x, grp, sub);
        else if (rem > 0) acc = gather_round<D, 4>(acc, c, a, n, t, x, grp, sub);
    }
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    return acc;
}

// ---- LDS hot-row variant: tagged columns (bit 31) are served from the workgroup's LDS copy of the
// hottest X rows ("LDS staging of embedding tiles"), everything else is gathered from L2 as above.
template <int D, int UN>
__device__ __forceinline__ float4 gather_round_hot(float4 acc, int c, float a, int n, int t0, const float *__restrict__ x,
                                                   const float *hot, int grp, int sub)
{
    constexpr int NG = 64 / (D / 4);
    float4 xv[UN];
    float av[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const int src = (t0 + j) * NG + grp;
        int cc = __shfl(c, src & 63, 64);
        float aa = __shfl(a, src & 63, 64);
        const bool ok = src < n;
        cc = ok ? cc : 0;
        av[j] = ok ? aa : 0.f;
        if (cc < 0) xv[j] = *reinterpret_cast<const float4 *>(hot + (unsigned)((cc & 0x7fffffff) * D + sub * 4));
        else xv[j] = *reinterpret_cast<const float4 *>(x + (unsigned)(cc * D + sub * 4));
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) acc = f4_fma(av[j], xv[j], acc);
    return acc;
}

template <int D, int UNMAX>
__device__ __forceinline__ float4 spmm_segment_hot(const int *__restrict__ col, const float *__restrict__ val, int eb, int ee,
                                                   const float *__restrict__ x, const float *hot, int lane)
{
    constexpr int G = D / 4, NG = 64 / G;
    const int grp = lane / G, sub = lane % G;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c_next = 0;
    float a_next = 0.f;
    if (eb + lane < ee) { c_next = col[eb + lane]; a_next = val[eb + lane]; }
    for (int base = eb; base < ee; base += 64) {
        const int n = min(64, ee - base);
        const int c = c_next;
        const float a = a_next;
        c_next = 0; a_next = 0.f;
        if (base + 64 + lane < ee) { c_next = col[base + 64 + lane]; a_next = val[base + 64 + lane]; }
        const int iters = (n + NG - 1) / NG;
        int t = 0;
        for (; t + UNMAX <= iters; t += UNMAX) acc = gather_round_hot<D, UNMAX>(acc, c, a, n, t, x, hot, grp, sub);
        const int rem = iters - t;
        if (rem > 4) acc = gather_round_hot<D, 8>(acc, c, a, n, t, x, hot, grp, sub);
        else if (rem > 0) acc = gather_round_hot<D, 4>(acc, c, a, n, t, x, hot, grp, sub);
    }
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    return acc;
}

template <int D>
__device__ __forceinline__ void spmm_epilogue(const SpmmEpi &e, int r, int sub, float4 v, float4 addv, float4 sumv)
{
    const size_t off = (size_t)r * D + (size_t)sub * 4;
    if (e.add) v = f4_add(v, addv);
    if (e.zero1) *reinterpret_cast<float4 *>(e.zero1 + off) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.zero2) *reinterpret_cast<float4 *>(e.zero2 + off) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.y) *reinterpret_cast<float4 *>(e.y + off) = v;
    if (e.sum_out) {
        float4 s = f4_add(sumv, v);
        s.x *= e.sum_scale; s.y *= e.sum_scale; s.z *= e.sum_scale; s.w *= e.sum_scale;
        *reinterpret_cast<float4 *>(e.sum_out + off) = s;
    }
    if (e.adam) {
        const float step_size = e.coef[0], bc2s = e.coef[1];
        const float w1 = (float)(1.0 - (double)e.b1), w2 = (float)(1.0 - (double)e.b2);
        float4 *pp = reinterpret_cast<float4 *>(e.p + off), *mp = reinterpret_cast<float4 *>(e.m + off);
        float4 *vp = reinterpret_cast<float4 *>(e.v + off);
        float4 p = *pp, m = *mp, vv = *vp;
        adam_elem(p.x, m.x, vv.x, v.x, w1, e.b2, w2, step_size, bc2s, e.eps);
        adam_elem(p.y, m.y, vv.y, v.y, w1, e.b2, w2, step_size, bc2s, e.eps);
        adam_elem(p.z, m.z, vv.z, v.z, w1, e.b2, w2, step_size, bc2s, e.eps);
        adam_elem(p.w, m.w, vv.w, v.w, w1, e.b2, w2, step_size, bc2s, e.eps);
        *pp = p; *mp = m; *vp = vv;
    }
}

template <int D, int UNMAX, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void spmm_csr_kernel(const SpmmArgs a)
{
    constexpr int G = D / 4;
    __shared__ float4 part[WAVES][G];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (a.e.bump && blockIdx.x == 0 && threadIdx.x == 0) {
        a.e.state[ST_STEP_BASE] += a.e.bump;
        a.e.state[ST_ADAM_T] += a.e.bump;
    }
    const int4 ds = a.wave_desc[(size_t)blockIdx.x * WAVES + w];  // {row, eb, ee, nseg}
    const bool lead = ds.w > 0 && lane < G;
    // epilogue operands do not depend on the gather: fetch them first so their latency hides
    float4 addv = make_float4(0.f, 0.f, 0.f, 0.f), sumv = addv;
    const size_t eoff = (size_t)max(ds.x, 0) * D + (size_t)(lane % G) * 4;
    if (lead && a.e.add) addv = *reinterpret_cast<const float4 *>(a.e.add + eoff);
    if (lead && a.e.sum_out) sumv = *reinterpret_cast<const float4 *>(a.e.sum_in + eoff);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ds.z > ds.y && !(a.dbg & 1)) acc = spmm_segment<D, UNMAX>(a.col, a.val, ds.y, ds.z, a.x, lane);
    if (lane < G) part[w][lane] = acc;
    __syncthreads();
    if (lead && !(a.dbg & 2)) {
        for (int k = 1; k < ds.w; ++k) acc = f4_add(acc, part[w + k][lane]);
        spmm_epilogue<D>(a.e, ds.x, lane, acc, addv, sumv);
    }
}

// Persistent form: one 16-wave workgroup per CU keeps the hot X rows of its row class in LDS
// (hot_H * D floats) and walks the schedule workgroups of that class, two (8-wave) at a time.
static constexpr int kHotWaves = 16;
static constexpr int kHotLdsBytes = 128 * 1024;
__host__ __device__ inline int hot_rows_for_dim(int d) { return kHotLdsBytes / (4 * d); }

template <int D>
__global__ __launch_bounds__(kHotWaves * 64, 4) void spmm_csr_hot_kernel(const SpmmArgs a)
{
    constexpr int G = D / 4, NG = 64 / G, SW = 8;  // SW = waves per schedule workgroup
    extern __shared__ __attribute__((aligned(16))) float hot[];  // [hot_H][D] then partials
    float4 (*part)[G] = reinterpret_cast<float4 (*)[G]>(hot + (size_t)a.hot_H * D);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (a.e.bump && blockIdx.x == 0 && threadIdx.x == 0) {
        a.e.state[ST_STEP_BASE] += a.e.bump;
        a.e.state[ST_ADAM_T] += a.e.bump;
    }
    // class of this workgroup: XCD halves (blockIdx % 8 < 4 -> class 0), see rk_csr_schedule_build
    const int pb = blockIdx.x, P = gridDim.x;
    int cls = 0, ord = pb, Pc = P;
    if (a.two_classes) {
        cls = (pb & 7) >= 4;
        ord = (pb >> 3) * 4 + (pb & 3);
        Pc = P / 2;
    }
    const int *list = a.pblocks + (cls ? a.nb_class0 : 0);
    const int n_list = cls ? a.n_blocks - a.nb_class0 : a.nb_class0;
    // stage the class's hot rows: 64/G rows per wave step, 16-byte loads
    {
        const int *hr = a.hot_rows + (size_t)cls * a.hot_H;
        const int grp = lane / G, sub = lane % G;
        for (int h = w * NG + grp; h < a.hot_H; h += kHotWaves * NG) {
            const int r = hr[h];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r >= 0) v = *reinterpret_cast<const float4 *>(a.x + (unsigned)(r * D + sub * 4));
            *reinterpret_cast<float4 *>(hot + (size_t)h * D + sub * 4) = v;
        }
    }
    __syncthreads();
    const int iters = (n_list + Pc * 2 - 1) / (Pc * 2);
    for (int it = 0; it < iters; ++it) {
        const int j = (it * Pc + ord) * 2 + (w >> 3);
        int4 ds = make_int4(-1, 0, 0, 0);
        if (j < n_list) ds = a.wave_desc[(size_t)list[j] * SW + (w & 7)];
        const bool lead = ds.w > 0 && lane < G;
        float4 addv = make_float4(0.f, 0.f, 0.f, 0.f), sumv = addv;
        const size_t eoff = (size_t)max(ds.x, 0) * D + (size_t)(lane % G) * 4;
        if (lead && a.e.add) addv = *reinterpret_cast<const float4 *>(a.e.add + eoff);
        if (lead && a.e.sum_out) sumv = *reinterpret_cast<const float4 *>(a.e.sum_in + eoff);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ds.z > ds.y) acc = spmm_segment_hot<D, 8>(a.col_tagged, a.val, ds.y, ds.z, a.x, hot, lane);
        if (lane < G) part[w][lane] = acc;
        __syncthreads();
        if (lead) {
            for (int k = 1; k < ds.w; ++k) acc = f4_add(acc, part[w + k][lane]);
            spmm_epilogue<D>(a.e, ds.x, lane, acc, addv, sumv);
        }
        __syncthreads();
    }
}

// Any d (<= 512): one X row per wave step, lanes stride over the row.  Same schedule and
// epilogue semantics; used for dims without a vector instantiation.
static constexpr int kGenMaxC = 8;
static __global__ __launch_bounds__(1024) void spmm_csr_generic_kernel(const SpmmArgs a)
{
    __shared__ float part[kSpmmWavesMax][kGenMaxC * 64];
    const int kSpmmWaves = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int d = a.d;
    if (a.e.bump && blockIdx.x == 0 && threadIdx.x == 0) {
        a.e.state[ST_STEP_BASE] += a.e.bump;
        a.e.state[ST_ADAM_T] += a.e.bump;
    }
    const int4 ds = a.wave_desc[(size_t)blockIdx.x * kSpmmWaves + w];
    const int r = ds.x;
    float acc[kGenMaxC];
#pragma unroll
    for (int k = 0; k < kGenMaxC; ++k) acc[k] = 0.f;
    for (int e = ds.y; e < ds.z; ++e) {
        const float *x = a.x + (size_t)a.col[e] * d;
        const float av = a.val[e];
#pragma unroll
        for (int k = 0; k < kGenMaxC; ++k)
            if (k * 64 + lane < d) acc[k] += av * x[k * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < kGenMaxC; ++k) part[w][k * 64 + lane] = acc[k];
    __syncthreads();
    if (ds.w <= 0) return;
#pragma unroll
    for (int k = 0; k < kGenMaxC; ++k)
        for (int ww = 1; ww < ds.w; ++ww) acc[k] += part[w + ww][k * 64 + lane];
    const SpmmEpi &e = a.e;
    const float w1 = (float)(1.0 - (double)e.b1), w2 = (float)(1.0 - (double)e.b2);
#pragma unroll
    for (int k = 0; k < kGenMaxC; ++k) {
        const int c = k * 64 + lane;
        if (c >= d) continue;
        float v = acc[k];
        const size_t o = (size_t)r * d + c;
        if (e.add) v += e.add[o];
        if (e.zero1) e.zero1[o] = 0.f;
        if (e.zero2) e.zero2[o] = 0.f;
        if (e.y) e.y[o] = v;
        if (e.sum_out) e.sum_out[o] = (e.sum_in[o] + v) * e.sum_scale;
        if (e.adam) {
            float p = e.p[o], m = e.m[o], vv = e.v[o];
            adam_elem(p, m, vv, v, w1, e.b2, w2, e.coef[0], e.coef[1], e.eps);
            e.p[o] = p; e.m[o] = m; e.v[o] = vv;
        }
    }
}

// Host-side launch (asynchronous).  Returns a hipError_t from the launch.
inline hipError_t spmm_launch(const SpmmArgs &a, hipStream_t s)
{
    if (a.n_rows <= 0) return hipSuccess;
    const int W = spmm_waves();
    static const int hot_off = getenv("RK_SPMM_NO_HOT") ? atoi(getenv("RK_SPMM_NO_HOT")) : 0;
    if (a.hot_H > 0 && !hot_off && W == 8 && (a.d == 32 || a.d == 64 || a.d == 128 || a.d == 256)) {
        static int n_cu = 0;
        static bool attr_set = false;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t p;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return hipErrorUnknown;
            n_cu = (p.multiProcessorCount / 8) * 8;
            if (n_cu < 8) n_cu = 8;
        }
        const size_t lds = (size_t)a.hot_H * a.d * 4 + (size_t)kHotWaves * (a.d / 4) * 16;
#define RK_HOT_CASE(D)                                                                                          \
    {                                                                                                           \
        if (!attr_set) {                                                                                        \
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_csr_hot_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); \
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_csr_hot_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); \
            hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_csr_hot_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); \
            hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_csr_hot_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); \
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) return hipErrorUnknown; \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        hipLaunchKernelGGL(spmm_csr_hot_kernel<D>, dim3(n_cu), dim3(kHotWaves * 64), lds, s, a);               \
    }
        switch (a.d) {
            case 32: RK_HOT_CASE(32) break;
            case 64: RK_HOT_CASE(64) break;
            case 128: RK_HOT_CASE(128) break;
            default: RK_HOT_CASE(256) break;
        }
#undef RK_HOT_CASE
        return hipGetLastError();
    }
    const dim3 grid(a.n_blocks), block(W * 64);
    static const int variant = getenv("RK_SPMM_VARIANT") ? atoi(getenv("RK_SPMM_VARIANT")) : 0;
    static const int dbg = getenv("RK_SPMM_DEBUG") ? atoi(getenv("RK_SPMM_DEBUG")) : 0;
    const_cast<SpmmArgs &>(a).dbg = dbg;
#define RK_SPMM_CASE(D, UN, WV, MW) hipLaunchKernelGGL((spmm_csr_kernel<D, UN, WV, MW>), grid, block, 0, s, a)
#define RK_SPMM_D(D)                                                                         \
    if (W == 16) { if (variant == 1) RK_SPMM_CASE(D, 16, 16, 4); else RK_SPMM_CASE(D, 8, 16, 8); } \
    else if (W == 8) { if (variant == 1) RK_SPMM_CASE(D, 8, 8, 8); else if (variant == 2) RK_SPMM_CASE(D, 16, 8, 4); else RK_SPMM_CASE(D, 8, 8, 6); } \
    else { if (variant == 1) RK_SPMM_CASE(D, 8, 4, 8); else RK_SPMM_CASE(D, 8, 4, 6); }
    switch (a.d) {
        case 32: RK_SPMM_D(32) break;
        case 64: RK_SPMM_D(64) break;
        case 128: RK_SPMM_D(128) break;
        case 256: RK_SPMM_D(256) break;
        default: hipLaunchKernelGGL(spmm_csr_generic_kernel, grid, block, 0, s, a); break;
    }
#undef RK_SPMM_D
#undef RK_SPMM_CASE
    return hipGetLastError();
}
