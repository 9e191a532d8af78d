// CSR SpMM for gfx950 with fused epilogues (layer sum, scaled output, addend, self-cleaning
// gradient buffers, dense Adam).  Replaces torch.sparse.mm at
// recad/model/victim/lightgcn.py:107 and its autograd twin.
//
// Mapping (DESIGN.md "SpMM"): 1024-thread workgroups = 16 waves.  Rows are visited in a
// degree-descending schedule (row_perm).  The first n_long rows get a whole workgroup each
// (16 waves split the row's nonzeros, combine through LDS in fixed order); every other
// workgroup takes 16 consecutive schedule entries, one row per wave, so the waves of a
// workgroup carry similar work.  Inside a wave a row of D floats is covered by G = D/4
// lanes with one 16-byte load each, so one global_load_dwordx4 gathers 64/G different
// X rows; the column/value stream is read coalesced (64 entries per wave) and broadcast
// with ds_bpermute.  Summation order is fixed by the schedule => bit-reproducible.
#pragma once
#include "common.h"

struct SpmmEpi {
    // v = acc (+ add[r])
    const float *add_lo, *add_hi;
    int add_split;
    float *y;  // nullable: y[r] = v
    // sum_out[r] = (sum_in[r] + v) * sum_scale   (nullable sum_out)
    const float *sum_lo, *sum_hi;
    int sum_split;
    float *sum_out;
    float sum_scale;
    float *zero1, *zero2;  // nullable: rows set to 0 after the addend was read
    // Adam on p[r] with gradient v
    int adam;
    float *p_lo, *p_hi, *m_lo, *m_hi, *v_lo, *v_hi;
    int p_split;
    const float *coef;  // {step_size, bc2s}
    float b1, b2, eps;
    int *state;  // bump words ST_STEP_BASE / ST_ADAM_T by `bump` (last kernel of a chunk)
    int bump;
};

struct SpmmArgs {
    int n_rows;
    const int *rowptr, *col;
    const float *val;
    const int *perm;
    int n_long;
    int d;
    const float *x_lo, *x_hi;
    int x_split;
    SpmmEpi e;
};

static constexpr int kSpmmWaves = 16;
static constexpr int kSpmmThreads = kSpmmWaves * kWave;

__host__ inline int spmm_grid(int n_rows, int n_long) { return n_long + (n_rows - n_long + kSpmmWaves - 1) / kSpmmWaves; }

template <int VEC>
struct Acc;
template <>
struct Acc<4> {
    float4 v;
    __device__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
};

__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc)
{
    acc.x += a * x.x; acc.y += a * x.y; acc.z += a * x.z; acc.w += a * x.w;
    return acc;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Partial sum of row segment [eb, ee) for the D/4 lanes that share `sub`; after the
// cross-group reduction every lane holds the total for its float4 slot.
template <int D>
__device__ __forceinline__ float4 spmm_segment(const int *__restrict__ col, const float *__restrict__ val, int eb, int ee,
                                               const float *x_lo, const float *x_hi, int split, int lane)
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
        for (int t = 0; t < iters; t += 4) {
            float4 x[4];
            float av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int src = (t + j) * NG + grp;
                int cc = __shfl(c, src & 63, 64);
                float aa = __shfl(a, src & 63, 64);
                const bool ok = src < n;
                cc = ok ? cc : 0;
                av[j] = ok ? aa : 0.f;
                x[j] = *reinterpret_cast<const float4 *>(row2(x_lo, x_hi, split, cc, D) + sub * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = f4_fma(av[j], x[j], acc);
        }
    }
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    return acc;
}

template <int D>
__device__ __forceinline__ void spmm_epilogue(const SpmmEpi &e, int r, int sub, float4 v)
{
    const size_t off = (size_t)sub * 4;
    if (e.add_lo) {
        float *ap = const_cast<float *>(row2(e.add_lo, e.add_hi, e.add_split, r, D)) + off;
        v = f4_add(v, *reinterpret_cast<const float4 *>(ap));
    }
    if (e.zero1) *reinterpret_cast<float4 *>(e.zero1 + (size_t)r * D + off) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.zero2) *reinterpret_cast<float4 *>(e.zero2 + (size_t)r * D + off) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.y) *reinterpret_cast<float4 *>(e.y + (size_t)r * D + off) = v;
    if (e.sum_out) {
        float4 s = *reinterpret_cast<const float4 *>(row2(e.sum_lo, e.sum_hi, e.sum_split, r, D) + off);
        s = f4_add(s, v);
        s.x *= e.sum_scale; s.y *= e.sum_scale; s.z *= e.sum_scale; s.w *= e.sum_scale;
        *reinterpret_cast<float4 *>(e.sum_out + (size_t)r * D + off) = s;
    }
    if (e.adam) {
        const float step_size = e.coef[0], bc2s = e.coef[1];
        const float w1 = (float)(1.0 - (double)e.b1), w2 = (float)(1.0 - (double)e.b2);
        float4 *pp = reinterpret_cast<float4 *>(row2(e.p_lo, e.p_hi, e.p_split, r, D) + off);
        float4 *mp = reinterpret_cast<float4 *>(row2(e.m_lo, e.m_hi, e.p_split, r, D) + off);
        float4 *vp = reinterpret_cast<float4 *>(row2(e.v_lo, e.v_hi, e.p_split, r, D) + off);
        float4 p = *pp, m = *mp, vv = *vp;
        adam_elem(p.x, m.x, vv.x, v.x, w1, e.b2, w2, step_size, bc2s, e.eps);
        adam_elem(p.y, m.y, vv.y, v.y, w1, e.b2, w2, step_size, bc2s, e.eps);
        adam_elem(p.z, m.z, vv.z, v.z, w1, e.b2, w2, step_size, bc2s, e.eps);
        adam_elem(p.w, m.w, vv.w, v.w, w1, e.b2, w2, step_size, bc2s, e.eps);
        *pp = p; *mp = m; *vp = vv;
    }
}

template <int D>
__global__ __launch_bounds__(kSpmmThreads) void spmm_csr_kernel(const SpmmArgs a)
{
    constexpr int G = D / 4;
    __shared__ float4 part[kSpmmWaves][G];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x;
    if (a.e.bump && b == 0 && threadIdx.x == 0) {
        a.e.state[ST_STEP_BASE] += a.e.bump;
        a.e.state[ST_ADAM_T] += a.e.bump;
    }
    if (b < a.n_long) {
        const int r = a.perm[b];
        const int rb = a.rowptr[r], re = a.rowptr[r + 1];
        int chunk = (re - rb + kSpmmWaves - 1) / kSpmmWaves;
        chunk = (chunk + 3) & ~3;
        const int eb = min(re, rb + w * chunk), ee = min(re, eb + chunk);
        float4 acc = spmm_segment<D>(a.col, a.val, eb, ee, a.x_lo, a.x_hi, a.x_split, lane);
        if (lane < G) part[w][lane] = acc;
        __syncthreads();
        if (w == 0 && lane < G) {
            float4 t = part[0][lane];
#pragma unroll
            for (int k = 1; k < kSpmmWaves; ++k) t = f4_add(t, part[k][lane]);
            spmm_epilogue<D>(a.e, r, lane, t);
        }
    } else {
        const int idx = a.n_long + (b - a.n_long) * kSpmmWaves + w;
        if (idx >= a.n_rows) return;
        const int r = a.perm[idx];
        float4 acc = spmm_segment<D>(a.col, a.val, a.rowptr[r], a.rowptr[r + 1], a.x_lo, a.x_hi, a.x_split, lane);
        if (lane < G) spmm_epilogue<D>(a.e, r, lane, acc);
    }
}

// Any d (<= 512): one X row per wave step, lanes stride over the row.  Same schedule and
// epilogue semantics; used for dims without a vector instantiation.
static constexpr int kGenMaxC = 8;
static __global__ __launch_bounds__(kSpmmThreads) void spmm_csr_generic_kernel(const SpmmArgs a)
{
    __shared__ float part[kSpmmWaves][kGenMaxC * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x, d = a.d;
    if (a.e.bump && b == 0 && threadIdx.x == 0) {
        a.e.state[ST_STEP_BASE] += a.e.bump;
        a.e.state[ST_ADAM_T] += a.e.bump;
    }
    int r, eb, ee;
    const bool is_long = b < a.n_long;
    if (is_long) {
        r = a.perm[b];
        const int rb = a.rowptr[r], re = a.rowptr[r + 1];
        const int chunk = (re - rb + kSpmmWaves - 1) / kSpmmWaves;
        eb = min(re, rb + w * chunk);
        ee = min(re, eb + chunk);
    } else {
        const int idx = a.n_long + (b - a.n_long) * kSpmmWaves + w;
        if (idx >= a.n_rows) return;
        r = a.perm[idx];
        eb = a.rowptr[r];
        ee = a.rowptr[r + 1];
    }
    float acc[kGenMaxC];
#pragma unroll
    for (int k = 0; k < kGenMaxC; ++k) acc[k] = 0.f;
    for (int e = eb; e < ee; ++e) {
        const float *x = row2(a.x_lo, a.x_hi, a.x_split, a.col[e], d);
        const float av = a.val[e];
#pragma unroll
        for (int k = 0; k < kGenMaxC; ++k)
            if (k * 64 + lane < d) acc[k] += av * x[k * 64 + lane];
    }
    if (is_long) {
#pragma unroll
        for (int k = 0; k < kGenMaxC; ++k) part[w][k * 64 + lane] = acc[k];
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int k = 0; k < kGenMaxC; ++k) {
            float t = part[0][k * 64 + lane];
            for (int ww = 1; ww < kSpmmWaves; ++ww) t += part[ww][k * 64 + lane];
            acc[k] = t;
        }
    }
    const SpmmEpi &e = a.e;
    const float w1 = (float)(1.0 - (double)e.b1), w2 = (float)(1.0 - (double)e.b2);
#pragma unroll
    for (int k = 0; k < kGenMaxC; ++k) {
        const int c = k * 64 + lane;
        if (c >= d) continue;
        float v = acc[k];
        if (e.add_lo) v += row2(e.add_lo, e.add_hi, e.add_split, r, d)[c];
        if (e.zero1) e.zero1[(size_t)r * d + c] = 0.f;
        if (e.zero2) e.zero2[(size_t)r * d + c] = 0.f;
        if (e.y) e.y[(size_t)r * d + c] = v;
        if (e.sum_out) e.sum_out[(size_t)r * d + c] = (row2(e.sum_lo, e.sum_hi, e.sum_split, r, d)[c] + v) * e.sum_scale;
        if (e.adam) {
            float *pp = row2(e.p_lo, e.p_hi, e.p_split, r, d) + c;
            float *mp = row2(e.m_lo, e.m_hi, e.p_split, r, d) + c;
            float *vp = row2(e.v_lo, e.v_hi, e.p_split, r, d) + c;
            float p = *pp, m = *mp, vv = *vp;
            adam_elem(p, m, vv, v, w1, e.b2, w2, e.coef[0], e.coef[1], e.eps);
            *pp = p; *mp = m; *vp = vv;
        }
    }
}

// Host-side launch (asynchronous).  Returns a hipError_t from the launch.
inline hipError_t spmm_launch(const SpmmArgs &a, hipStream_t s)
{
    if (a.n_rows <= 0) return hipSuccess;
    const dim3 grid(spmm_grid(a.n_rows, a.n_long)), block(kSpmmThreads);
    switch (a.d) {
        case 32: hipLaunchKernelGGL(spmm_csr_kernel<32>, grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL(spmm_csr_kernel<64>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(spmm_csr_kernel<128>, grid, block, 0, s, a); break;
        case 256: hipLaunchKernelGGL(spmm_csr_kernel<256>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(spmm_csr_generic_kernel, grid, block, 0, s, a); break;
    }
    return hipGetLastError();
}
