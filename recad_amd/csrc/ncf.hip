// NCF victim hot path on gfx950 (model = NeuMF-end / NeuMF-pre / MLP / GMF, ncf.py:49-52,112-131) (recad/model/victim/ncf.py:112-153): embedding
// gathers, the MLP tower as exact-fp32 MFMA GEMMs (forward, dX and dW), predict layer +
// BCE-with-logits, embedding-gradient scatter-add and one multi-tensor dense Adam launch.
#include <algorithm>

#include "gemm.h"

static int gemm(hipStream_t s, int M, int N, int K, const float *A, long long a_rs, long long a_cs, const float *B,
                long long b_rs, long long b_cs, float *C, int ldc, const float *bias, int relu, const float *mask, int ldmask,
                int split_k = 1)
{
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.split_k = split_k;
    g.M = M; g.N = N; g.K = K; g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.B = B; g.b_rs = b_rs; g.b_cs = b_cs;
    g.C = C; g.ldc = ldc; g.col_bias = bias; g.relu = relu; g.mask = mask; g.ldmask = ldmask;
    RK_HIP(gemm_f32_launch(g, s));
    return RK_OK;
}

// K-slices for a GEMM whose whole-K 64-tiles would not fill the chip (batch 1024 against a few hundred
// outputs): aim at ~2-3 workgroups per CU, keep >= 2 k-chunks per slice, stay inside the workspace.
static int pick_splits(int M, int N, int K, long long cap_floats)
{
    const long long nwg128 = (long long)((N + 127) / 128) * ((M + 127) / 128);
    if (nwg128 >= 384) return 1;  // the 128-tile form takes it
    // whole-K on the deep 64-tile kernel beats K-slices + the slice-adding pass wherever that kernel applies (measured,
    // scripts/gemm_probe.hip, dX forms at batch 1024: 1024 x 512 x 256 8.4 us against 8.5 + 4.7; 1024 x 1024 x 512 18.4
    // against 15.4 + 5; 1024 x 256 x 128 5.8 against 6.3 + 4.7)
    if (M % 64 == 0 && N % 64 == 0 && K % 128 == 0) return 1;
    const long long tiles = (long long)((N + 63) / 64) * ((M + 63) / 64);
    long long sp = std::min<long long>(512 / std::max<long long>(tiles, 1), (K + 63) / 64);
    sp = std::min<long long>(sp, cap_floats / std::max<long long>((long long)M * N, 1));
    return gemm_effective_splits(K, (int)std::max<long long>(1, std::min<long long>(sp, 32)));
}

// y = epilogue(sum_q part[q]) in slice order q = 0..sp-1: bias + ReLU (forward) or the ReLU mask (dX)
__global__ void slices_epilogue_kernel(long long n4, int N, int sp, const float *__restrict__ part, float *__restrict__ y,
                                       const float *__restrict__ bias, int relu, const float *__restrict__ mask, int pairwise)
{
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4 *>(part)[q];
        if (pairwise && sp == 8) {   // ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)): the blocked training forward
            float4 t[8];
            t[0] = v;
#pragma unroll
            for (int k = 1; k < 8; ++k) t[k] = reinterpret_cast<const float4 *>(part)[(long long)k * n4 + q];
            auto add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
            v = add(add(add(t[0], t[1]), add(t[2], t[3])), add(add(t[4], t[5]), add(t[6], t[7])));
        } else {
            for (int k = 1; k < sp; ++k) {
                const float4 t = reinterpret_cast<const float4 *>(part)[(long long)k * n4 + q];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
        }
        if (bias) {
            const float4 bb = *reinterpret_cast<const float4 *>(bias + (q * 4) % N);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (mask) {
            const float4 mm = reinterpret_cast<const float4 *>(mask)[q];
            v.x = mm.x > 0.f ? v.x : 0.f; v.y = mm.y > 0.f ? v.y : 0.f; v.z = mm.z > 0.f ? v.z : 0.f; v.w = mm.w > 0.f ? v.w : 0.f;
        }
        reinterpret_cast<float4 *>(y)[q] = v;
    }
}

// C[M,N] (contiguous: ldc == N == ldmask, N % 4 == 0) = epilogue(A.B^T).  Whole-K tiles with the fused epilogue
// when they fill the chip or scratch == NULL; otherwise K-slices park their partial products in the workspace
// and one elementwise pass adds them IN SLICE ORDER and applies the epilogue (deterministic).
// Used for the dX GEMMs only.  The FORWARD GEMMs stay whole-K on purpose: their k-ordered fmaf chain is the
// oracle's summation order, and the ReLU gates they decide are where ulp differences turn macroscopic -- a
// unit that is dead for the whole batch has an exactly-zero weight-gradient row, and one flipped gate makes
// Adam move that row by a full +-lr.  Splitting the forward K (ordered slices or atomics, both measured:
// 2.85 -> 3.2 M samples/s) made tests/test_gpu_parity.py::test_ncf_train_golden fail in 3-7 of 14 runs with
// 4e-4 differences on a third of MLP_layers.1.weight; with the dX GEMMs split only: 0 of 14.
static int gemm_auto(hipStream_t s, int M, int N, int K, const float *A, long long a_rs, long long a_cs, const float *B,
                     long long b_rs, long long b_cs, float *C, const float *bias, int relu, const float *mask,
                     float *scratch, long long cap_floats)
{
    const int sp = (N % 4 == 0 && scratch) ? pick_splits(M, N, K, cap_floats) : 1;
    if (sp <= 1) return gemm(s, M, N, K, A, a_rs, a_cs, B, b_rs, b_cs, C, N, bias, relu, mask, N);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.split_k = sp; g.sk_part = scratch; g.sk_stride = (long long)M * N;
    g.M = M; g.N = N; g.K = K; g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.B = B; g.b_rs = b_rs; g.b_cs = b_cs;
    g.C = C; g.ldc = N;
    RK_HIP(gemm_f32_launch(g, s));
    const long long n4 = (long long)M * N / 4;
    hipLaunchKernelGGL(slices_epilogue_kernel, dim3((int)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, s, n4, N, sp,
                       scratch, C, bias, relu, mask, 0);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// The TRAINING forward of a tower layer whose input width K is a multiple of 256: y = relu(sum + bias) with the products summed
// in kFwdBlocks consecutive k-blocks (each the MFMA's k-ordered chain) combined PAIRWISE -- the oracle's ncf_forward_one_ex
// (blocked) order.  Why: a layer's ReLU gates are the masks of its backward, and a pre-activation that is zero to within the
// summation noise can land on either side; one flipped gate changes every lower dW by a rank-1 term of weight 1/B (1e-3 of
// the largest entry at factor 256).  One 2048-long chain is 9e-6 of the layer's rms off fp64, ATen's blocked sgemm 2.5e-6,
// this form 2e-6: on the reference's golden batch the single chain flips one of 1.8 M gates (|z| = 7e-10), this form none, and
// the step-1 gradients go from 9e-4 to 1e-6 of the reference's.  K-slices of the 64-tile kernels park their partial products in
// `scratch` ([8][rows, N], rows in chunks that fit); they also fill the chip better than whole-K tiles at batch 1024.
static constexpr int kFwdBlocks = 8;
static inline bool fwd_blocked(int K) { return K % 256 == 0; }
static int gemm_fwd_blocked(hipStream_t s, int M, int N, int K, const float *A, const float *W, float *Y, const float *bias,
                            float *scratch, long long cap_floats)
{
    if (!scratch || cap_floats < (long long)kFwdBlocks * 64 * N || N % 4) RK_FAIL(RK_EINVAL, "ncf: the blocked forward needs desc.gemm_scratch of >= %lld floats", (long long)kFwdBlocks * 64 * N);
    const int rows_cap = (int)std::min<long long>(M, cap_floats / ((long long)kFwdBlocks * N) / 64 * 64);
    for (int m0 = 0; m0 < M; m0 += rows_cap) {
        const int mc = std::min(rows_cap, M - m0);
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.split_k = kFwdBlocks; g.sk_part = scratch; g.sk_stride = (long long)mc * N;
        g.M = mc; g.N = N; g.K = K; g.A = A + (size_t)m0 * K; g.a_rs = K; g.a_cs = 1; g.B = W; g.b_rs = K; g.b_cs = 1;
        g.C = Y + (size_t)m0 * N; g.ldc = N;
        RK_HIP(gemm_f32_launch(g, s));
        const long long n4 = (long long)mc * N / 4;
        hipLaunchKernelGGL(slices_epilogue_kernel, dim3((int)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, s, n4, N, kFwdBlocks,
                           scratch, Y + (size_t)m0 * N, bias, 1, (const float *)nullptr, 1);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}

// ---------------------------------------------------------------- element kernels
// pair b of this chunk: explicit (users[b], items[b]) or full-catalog (user_ids[b / I], b % I)
struct PairSrc {
    const int64_t *users, *items;
    const int32_t *user_ids;
    int n_items;
    long long off;
};
__device__ __forceinline__ void pair_at(const PairSrc &p, long long b, long long &u, long long &i)
{
    if (p.user_ids) { const long long q = p.off + b; u = p.user_ids[q / p.n_items]; i = q % p.n_items; }
    else { u = p.users[p.off + b]; i = p.items[p.off + b]; }
}

// X0[b] = [um[u] | im[i]]   (ncf.py:119-121)
// nn.Dropout in front of a tower layer (ncf.py:44): element `id` of the layer input is kept iff its counter hash under the
// (call, layer) seed is below keep_prob * 2^24, and scaled by 1 / keep_prob.  The backward applies the same function
// to dX of that layer (same ids => same mask).
struct DropSpec {
    unsigned thresh24;   // 0 = off
    float scale;
    unsigned long long seed;
};
__device__ __forceinline__ float drop_apply(const DropSpec &d, unsigned id, float v)
{
    return d.thresh24 ? (rk_drop_keep(d.seed, id, d.thresh24) ? v * d.scale : 0.f) : v;
}
__global__ void dropout_apply_kernel(long long n, float *x, DropSpec d)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[i] = drop_apply(d, (unsigned)i, x[i]);
}

__global__ void ncf_gather_kernel(PairSrc p, int nb, int E, const float *__restrict__ um, const float *__restrict__ im,
                                  float *__restrict__ x0, DropSpec drop)
{
    const int lane = threadIdx.x & 63;
    for (long long b = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < nb; b += (long long)gridDim.x * (blockDim.x >> 6)) {
        long long u, i;
        pair_at(p, b, u, i);
        const float *su = um + (size_t)u * E, *si = im + (size_t)i * E;
        float *d = x0 + (size_t)b * 2 * E;
        for (int k = lane * 4; k < E; k += 256) {
            float4 a = *reinterpret_cast<const float4 *>(su + k), c = *reinterpret_cast<const float4 *>(si + k);
            if (drop.thresh24) {
                const unsigned ia = (unsigned)((size_t)b * 2 * E + k), ic = ia + (unsigned)E;
                a.x = drop_apply(drop, ia, a.x); a.y = drop_apply(drop, ia + 1, a.y); a.z = drop_apply(drop, ia + 2, a.z); a.w = drop_apply(drop, ia + 3, a.w);
                c.x = drop_apply(drop, ic, c.x); c.y = drop_apply(drop, ic + 1, c.y); c.z = drop_apply(drop, ic + 2, c.z); c.w = drop_apply(drop, ic + 3, c.w);
            }
            *reinterpret_cast<float4 *>(d + k) = a;
            *reinterpret_cast<float4 *>(d + E + k) = c;
        }
    }
}

// logit = pw . concat + pb with concat = [ug*ig | xL] (NeuMF), xL (MLP) or ug*ig (GMF) (ncf.py:114-131); optional BCE
// loss + d0 = (sigmoid - y)/nb.  mode: RK_NCF_NEUMF / RK_NCF_MLP / RK_NCF_GMF.
__global__ __launch_bounds__(256) void ncf_predict_kernel(PairSrc p, int nb, int f, int mode, const float *__restrict__ ug,
                                                          const float *__restrict__ ig, const float *__restrict__ xl,
                                                          const float *__restrict__ pw, const float *__restrict__ pb,
                                                          float *__restrict__ logits, const int64_t *labels, float *d0,
                                                          float *loss_partials, float *__restrict__ dxl, float *g_ug, float *g_ig)
{
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float invB = 1.f / (float)nb;
    float lsum = 0.f;
    for (int b = blockIdx.x * 4 + w; b < nb; b += gridDim.x * 4) {
        long long u, i;
        pair_at(p, b, u, i);
        const float *pu = ug + (size_t)u * f, *pi = ig + (size_t)i * f, *px = xl + (size_t)b * f;
        float s = 0.f;
        if (mode == RK_NCF_NEUMF) { for (int k = lane; k < f; k += 64) s += pw[k] * (pu[k] * pi[k]) + pw[f + k] * px[k]; }
        else if (mode == RK_NCF_MLP) { for (int k = lane; k < f; k += 64) s += pw[k] * px[k]; }
        else { for (int k = lane; k < f; k += 64) s += pw[k] * (pu[k] * pi[k]); }
        s = wave_sum(s) + pb[0];
        float dd = 0.f;
        if (lane == 0) {
            if (logits) logits[p.off + b] = s;
            if (labels) {
                const float y = (float)labels[p.off + b];
                lsum += fmaxf(s, 0.f) - s * y + log1pf(expf(-fabsf(s)));
                dd = (1.f / (1.f + expf(-s)) - y) * invB;
                d0[b] = dd;
            }
        }
        if (dxl) {
            // training: the predict layer's backward for this pair straight away -- dXL[b,k] = d0[b]*pw[off+k], masked by the top
            // tower layer's own ReLU (xL > 0); GMF table gradients by atomics.  (Was two more launches of ~5 us each.)
            dd = __shfl(dd, 0);
            float *pd = dxl + (size_t)b * f;
            const int off = mode == RK_NCF_NEUMF ? f : 0;
            for (int k = lane; k < f; k += 64) {
                if (mode != RK_NCF_GMF) pd[k] = px[k] > 0.f ? dd * pw[off + k] : 0.f;
                if (mode != RK_NCF_MLP) {
                    unsafeAtomicAdd(g_ug + (size_t)u * f + k, dd * pw[k] * pi[k]);
                    unsafeAtomicAdd(g_ig + (size_t)i * f + k, dd * pw[k] * pu[k]);
                }
            }
        }
    }
    if (loss_partials) {
        if (lane == 0) red[w] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) loss_partials[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * invB;
    }
}

// Predict-layer weight gradient: gpw[k] += sum_b d0[b] * z[b,k], z = [ug*ig | xL]; gpb += sum_b d0[b].
// Deterministic two-stage sum (ulp noise in a weight gradient is amplified by Adam wherever the gradient is
// otherwise exactly zero -- dead ReLU units -- so every reduction that can be ordered is): stage 1, one
// workgroup per (64 outputs, slab of kWgradSlab batch rows), 16 row groups combined through LDS in fixed
// order -> part[slab][k]; stage 2 adds the slabs in order.  (One workgroup per 64 outputs walking all 1024
// rows' dependent gathers was 57 us of a 417 us step.)
static constexpr int kWgradSlab = 64;
__global__ __launch_bounds__(1024) void ncf_predict_wgrad_kernel(PairSrc p, int nb, int f, int mode, const float *__restrict__ ug,
                                                                const float *__restrict__ ig, const float *__restrict__ xl,
                                                                const float *__restrict__ d0, float *__restrict__ part)
{
    __shared__ float red[16][64];
    const int kc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kc;
    const int b_lo = blockIdx.y * kWgradSlab, b_hi = min(nb, b_lo + kWgradSlab);
    const int PS = mode == RK_NCF_NEUMF ? 2 * f : f;   // predict_layer input width (ncf.py:49-52)
    float s = 0.f;
    if (k <= PS) {
        float z[kWgradSlab / 16], dd[kWgradSlab / 16];
#pragma unroll
        for (int j = 0; j < kWgradSlab / 16; ++j) {
            const int b = b_lo + rg + 16 * j;
            z[j] = 0.f; dd[j] = 0.f;
            if (b < b_hi) {
                dd[j] = d0[b];
                if (k == PS) z[j] = 1.f;
                else if (mode == RK_NCF_MLP) z[j] = xl[(size_t)b * f + k];
                else if (k < f) { long long u, i; pair_at(p, b, u, i); z[j] = ug[(size_t)u * f + k] * ig[(size_t)i * f + k]; }
                else z[j] = xl[(size_t)b * f + (k - f)];
            }
        }
#pragma unroll
        for (int j = 0; j < kWgradSlab / 16; ++j) s += dd[j] * z[j];
    }
    red[rg][kc] = s;
    __syncthreads();
    if (rg == 0 && k <= PS) {
        float t = red[0][kc];
        for (int q = 1; q < 16; ++q) t += red[q][kc];
        part[(size_t)blockIdx.y * (PS + 1) + k] = t;
    }
}
// Column sums of one train step in ONE launch (job = blockIdx.y): db_l[n] += sum_m dY_l[m, n] for every tower layer -- the
// dY's of all layers are still in the activation workspace after the backward sweep -- and, as one more job, the slab
// partials of the predict-layer weight gradient (gpw[k] / gpb += sum_slab part[slab, k]).  Per job: 64 columns x 16 row
// groups per workgroup, fixed-order LDS combine (deterministic, see above), eight independent loads in flight per
// thread.  (One launch per layer plus the finish kernel were 5-6 launches of ~5 us each.)
struct ColsumJobs {
    int n_jobs;
    int M[RK_NCF_MAX_LAYERS + 1], N[RK_NCF_MAX_LAYERS + 1];
    const float *src[RK_NCF_MAX_LAYERS + 1];
    float *dst[RK_NCF_MAX_LAYERS + 1];
    float *dst_last[RK_NCF_MAX_LAYERS + 1];   // optional: column N-1 goes here instead of dst[N-1] (gpb)
};
__global__ __launch_bounds__(1024) void colsum_jobs_kernel(const ColsumJobs j)
{
    __shared__ float red[16][64];
    const int job = blockIdx.y;
    const int M = j.M[job], N = j.N[job];
    if ((int)blockIdx.x * 64 >= N) return;
    const float *__restrict__ dY = j.src[job];
    const int nc = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nc;
    float s = 0.f;
    if (n < N) {
        for (int m0 = rg; m0 < M; m0 += 16 * 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = m0 + 16 * q;
                v[q] = m < M ? dY[(size_t)m * N + n] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
    }
    red[rg][nc] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = red[0][nc];
        for (int q = 1; q < 16; ++q) t += red[q][nc];
        float *out = (j.dst_last[job] && n == N - 1) ? j.dst_last[job] : j.dst[job] + n;
        *out += t;
    }
}

// g_um[u] += dX0[b, :E], g_im[i] += dX0[b, E:]
__global__ void ncf_scatter_kernel(PairSrc p, int nb, int E, const float *__restrict__ dx0, float *g_um, float *g_im)
{
    const int lane = threadIdx.x & 63;
    for (long long b = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < nb; b += (long long)gridDim.x * (blockDim.x >> 6)) {
        long long u, i;
        pair_at(p, b, u, i);
        const float *s = dx0 + (size_t)b * 2 * E;
        for (int k = lane; k < E; k += 64) {
            unsafeAtomicAdd(g_um + (size_t)u * E + k, s[k]);
            unsafeAtomicAdd(g_im + (size_t)i * E + k, s[E + k]);
        }
    }
}

// dense Adam over every tensor in one launch; gradients are re-zeroed after use
struct MultiAdam {
    int n_tensors;
    float *p[RK_NCF_MAX_TENSORS], *g[RK_NCF_MAX_TENSORS], *m[RK_NCF_MAX_TENSORS], *v[RK_NCF_MAX_TENSORS];
    long long n[RK_NCF_MAX_TENSORS];
    float step_size, bc2s, b1, b2, eps;
};
// 16 bytes per lane and two independent float4 quads in flight per thread: with one float per lane a wave had 1 KB in flight per
// iteration and the launch ran at 4.6 TB/s (485 us for the 80 M parameters of factor 256 / 5 layers); tensors whose four arrays
// are not all 16-byte aligned take the scalar loop.
__global__ __launch_bounds__(256) void multi_adam_kernel(const MultiAdam a)
{
    const int t = blockIdx.y;
    const float w1 = (float)(1.0 - (double)a.b1), w2 = (float)(1.0 - (double)a.b2);
    float *p = a.p[t], *g = a.g[t], *m = a.m[t], *v = a.v[t];
    const long long n = a.n[t];
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    long long done = 0;
    if (vec) {
        const long long n4 = n >> 2;
        float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
        auto upd = [&](float4 &pp, float4 &mm, float4 &vv, const float4 &gg) {
            adam_elem(pp.x, mm.x, vv.x, gg.x, w1, a.b2, w2, a.step_size, a.bc2s, a.eps);
            adam_elem(pp.y, mm.y, vv.y, gg.y, w1, a.b2, w2, a.step_size, a.bc2s, a.eps);
            adam_elem(pp.z, mm.z, vv.z, gg.z, w1, a.b2, w2, a.step_size, a.bc2s, a.eps);
            adam_elem(pp.w, mm.w, vv.w, gg.w, w1, a.b2, w2, a.step_size, a.bc2s, a.eps);
        };
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        long long i = tid;
        for (; i + nth < n4; i += 2 * nth) {
            const long long j = i + nth;
            const float4 g0 = g4[i], g1 = g4[j];
            float4 p0 = p4[i], m0 = m4[i], v0 = v4[i], p1 = p4[j], m1 = m4[j], v1 = v4[j];
            upd(p0, m0, v0, g0);
            upd(p1, m1, v1, g1);
            g4[i] = z; p4[i] = p0; m4[i] = m0; v4[i] = v0;
            g4[j] = z; p4[j] = p1; m4[j] = m1; v4[j] = v1;
        }
        if (i < n4) {
            const float4 g0 = g4[i];
            float4 p0 = p4[i], m0 = m4[i], v0 = v4[i];
            upd(p0, m0, v0, g0);
            g4[i] = z; p4[i] = p0; m4[i] = m0; v4[i] = v0;
        }
        done = n4 << 2;
    }
    for (long long i = done + tid; i < n; i += nth) {
        const float gg = g[i];
        g[i] = 0.f;
        float pp = p[i], mm = m[i], vv = v[i];
        adam_elem(pp, mm, vv, gg, w1, a.b2, w2, a.step_size, a.bc2s, a.eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// ---------------------------------------------------------------- host orchestration
static int check_ncf(const rk_ncf_desc &d)
{
    if (d.n_users <= 0 || d.n_items <= 0 || d.factor <= 0 || d.n_layers < 1 || d.n_layers > RK_NCF_MAX_LAYERS)
        RK_FAIL(RK_EINVAL, "ncf: bad sizes");
    if ((d.factor << (d.n_layers - 1)) % 4) RK_FAIL(RK_EINVAL, "ncf: factor*2^(L-1) must be a multiple of 4");
    if (!d.ug || !d.ig || !d.um || !d.im || !d.pw || !d.pb || !d.acts || !d.dacts || !d.d0 || d.max_batch <= 0)
        RK_FAIL(RK_EINVAL, "ncf: null pointer");
    for (int l = 0; l < d.n_layers; ++l)
        if (!d.W[l] || !d.b[l]) RK_FAIL(RK_EINVAL, "ncf: tower pointer missing");
    if (d.mode != RK_NCF_NEUMF && d.mode != RK_NCF_MLP && d.mode != RK_NCF_GMF) RK_FAIL(RK_EINVAL, "ncf: unknown mode %d", d.mode);
    if (!(d.dropout >= 0.f) || d.dropout >= 1.f) RK_FAIL(RK_EINVAL, "ncf: dropout must be in [0, 1)");
    return RK_OK;
}

static inline int in_of(const rk_ncf_desc &d, int l) { return d.factor << (d.n_layers - l); }  // input width of layer l
static size_t act_off(const rk_ncf_desc &d, int l, int nbmax)
{
    size_t o = 0;
    for (int k = 0; k < l; ++k) o += (size_t)in_of(d, k) * nbmax;
    return o;
}

static DropSpec drop_spec(const rk_ncf_desc &d, unsigned long long call, int layer)
{
    DropSpec ds;
    ds.thresh24 = d.dropout > 0.f ? (unsigned)((1.0 - (double)d.dropout) * 16777216.0) : 0u;
    ds.scale = d.dropout > 0.f ? 1.0f / (1.0f - d.dropout) : 1.f;
    ds.seed = rk_drop_step_seed((unsigned long long)d.drop_seed, call * 16ULL + (unsigned long long)layer);
    return ds;
}

// forward for one chunk of nb pairs; acts[l] = input of layer l (after its dropout), acts[L] = tower output [nb, f].
// `call` numbers the dropout masks (train: the global step; scoring: desc.drop_call and the chunk).
static int ncf_forward_chunk(const rk_ncf_desc &d, const PairSrc &p, int nb, unsigned long long call, hipStream_t s,
                             const float *prefix = nullptr,   // prefix: [users of the call][out0] layer-0 user halves
                             bool train = false)              // train: layers of width % 256 == 0 use the blocked sums (gemm_fwd_blocked)
{
    if (d.mode == RK_NCF_GMF) return RK_OK;   // no tower (ncf.py:118-127)
    const int L = d.n_layers, E = d.factor << (L - 1);
    float *x0 = d.acts;
    int l_first = 0;
    const int out0 = in_of(d, 0) / 2;
    if (prefix) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.M = nb; g.N = out0; g.K = E;
        g.A = d.im; g.a_rs = E; g.a_cs = 1; g.a_rmod = p.n_items; g.a_roff = (int)(p.off % p.n_items);
        g.acc_init = prefix; g.ld_init = out0; g.init_base = -(int)(p.off / p.n_items);   // row = user index within the call
        g.B = d.W[0] + E; g.b_rs = 2 * E; g.b_cs = 1;
        g.C = d.acts + act_off(d, 1, d.max_batch); g.ldc = out0;
        g.col_bias = d.b[0]; g.relu = 1;
        RK_HIP(gemm_f32_launch(g, s));
        l_first = 1;
    } else {
        hipLaunchKernelGGL(ncf_gather_kernel, dim3(std::min(2048, (nb + 3) / 4)), dim3(256), 0, s, p, nb, E, d.um, d.im, x0, drop_spec(d, call, 0));
        RK_CHECK_LAUNCH();
    }
    for (int l = l_first; l < L; ++l) {
        const int in = in_of(d, l), out = in / 2;
        float *y = d.acts + act_off(d, l + 1, d.max_batch);
        int rc;
        if (train && fwd_blocked(in))
            rc = gemm_fwd_blocked(s, nb, out, in, d.acts + act_off(d, l, d.max_batch), d.W[l], y, d.b[l], d.gemm_scratch, d.gemm_scratch_floats);
        else
            rc = gemm_auto(s, nb, out, in, d.acts + act_off(d, l, d.max_batch), in, 1, d.W[l], in, 1,
                           y, d.b[l], 1, nullptr, nullptr, 0);  // whole-K: see gemm_auto
        if (rc) return rc;
        if (d.dropout > 0.f && l + 1 < L) {   // Dropout in front of layer l+1 (none in front of predict_layer)
            const long long n = (long long)nb * out;
            hipLaunchKernelGGL(dropout_apply_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, s, n, y, drop_spec(d, call, l + 1));
            RK_CHECK_LAUNCH();
        }
    }
    return RK_OK;
}

RK_EXPORT int rk_ncf_forward(const rk_ncf_desc *desc, const int64_t *users, const int64_t *items,
                             const int32_t *user_ids, int32_t n_items_catalog, int64_t n, float *out, void *stream)
{
    if (!desc || !out || n < 0) RK_FAIL(RK_EINVAL, "rk_ncf_forward: bad arguments");
    if (!user_ids && (!users || !items)) RK_FAIL(RK_EINVAL, "rk_ncf_forward: give (users, items) or user_ids");
    int rc = check_ncf(*desc);
    if (rc) return rc;
    const rk_ncf_desc &d = *desc;
    hipStream_t s = (hipStream_t)stream;
    // Full-catalog scoring (pair q = (user_ids[q / I], q % I)), no dropout: layer 0 over [um[u] | im[i]] is the SAME k-ordered
    // chain for every item of a user up to k = E.  That prefix is computed once per user of the call (one GEMM over the user
    // half of W0); layer 0 then runs over the ITEM half only with its accumulators starting at the user's prefix: bit-identical
    // results, half the layer-0 flops (layer 0 is 3/4 of the tower), and neither the gather nor the [pairs, 2E] input exist.
    const float *prefix = nullptr;
    if (user_ids && d.mode != RK_NCF_GMF && !(d.dropout > 0.f) && d.gemm_scratch && n_items_catalog > 0) {
        const int E = d.factor << (d.n_layers - 1), out0 = in_of(d, 0) / 2;
        const long long n_u = (n + n_items_catalog - 1) / n_items_catalog;
        if (n_u * out0 <= d.gemm_scratch_floats && E % 4 == 0) {
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.M = (int)n_u; g.N = out0; g.K = E;
            g.A = d.um; g.a_rs = E; g.a_cs = 1; g.a_ridx = user_ids;
            g.B = d.W[0]; g.b_rs = 2 * E; g.b_cs = 1;
            g.C = d.gemm_scratch; g.ldc = out0;
            RK_HIP(gemm_f32_launch(g, s));
            prefix = d.gemm_scratch;
        }
    }
    for (long long off = 0; off < n; off += d.max_batch) {
        const int nb = (int)std::min<long long>(d.max_batch, n - off);
        PairSrc p{users, items, user_ids, n_items_catalog, off};
        rc = ncf_forward_chunk(d, p, nb, ((unsigned long long)d.drop_call << 24) + (unsigned long long)(off / d.max_batch) + (1ULL << 60), s, prefix);
        if (rc) return rc;
        hipLaunchKernelGGL(ncf_predict_kernel, dim3(std::min(1024, (nb + 3) / 4)), dim3(256), 0, s, p, nb, d.factor, d.mode, d.ug, d.ig,
                           d.acts + act_off(d, d.n_layers, d.max_batch), d.pw, d.pb, out, (const int64_t *)nullptr,
                           (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}

RK_EXPORT int rk_ncf_train_epoch(const rk_ncf_desc *desc, const int64_t *users, const int64_t *items,
                                 const int64_t *labels, int64_t n, int32_t batch, int32_t adam_t0,
                                 float *loss_partials, int32_t apply_update, void *stream)
{
    if (!desc || !users || !items || !labels || !loss_partials || n <= 0 || batch <= 0)
        RK_FAIL(RK_EINVAL, "rk_ncf_train_epoch: bad arguments");
    int rc = check_ncf(*desc);
    if (rc) return rc;
    const rk_ncf_desc &d = *desc;
    if (batch > d.max_batch) RK_FAIL(RK_EINVAL, "rk_ncf_train_epoch: batch %d > desc.max_batch %d", batch, d.max_batch);
    const int L = d.n_layers, f = d.factor, E = f << (L - 1), T = 4 + 2 * L + 2;
    const int PS = d.mode == RK_NCF_NEUMF ? 2 * f : f;
    for (int t = 0; t < T; ++t)
        if (!d.grad[t] || !d.m[t] || !d.v[t]) RK_FAIL(RK_EINVAL, "ncf: grad/moment pointer %d missing", t);
    if (!d.wgrad_part) RK_FAIL(RK_EINVAL, "ncf: desc.wgrad_part missing");
    hipStream_t s = (hipStream_t)stream;
    // tensor order: ug, ig, um, im, W0.., b0.., pw, pb
    float *P[RK_NCF_MAX_TENSORS];
    long long NN[RK_NCF_MAX_TENSORS];
    P[0] = d.ug; NN[0] = (long long)d.n_users * f;
    P[1] = d.ig; NN[1] = (long long)d.n_items * f;
    P[2] = d.um; NN[2] = (long long)d.n_users * E;
    P[3] = d.im; NN[3] = (long long)d.n_items * E;
    for (int l = 0; l < L; ++l) {
        const int in = in_of(d, l);
        P[4 + l] = d.W[l]; NN[4 + l] = (long long)in * (in / 2);
        P[4 + L + l] = d.b[l]; NN[4 + L + l] = in / 2;
    }
    P[4 + 2 * L] = d.pw; NN[4 + 2 * L] = PS;
    P[5 + 2 * L] = d.pb; NN[5 + 2 * L] = 1;
    for (int t = 0; t < T; ++t) RK_HIP(hipMemsetAsync(d.grad[t], 0, sizeof(float) * (size_t)NN[t], s));
    const int n_steps = (int)((n + batch - 1) / batch);
    RK_HIP(hipMemsetAsync(loss_partials, 0, sizeof(float) * (size_t)n_steps * RK_LOSS_PARTIALS, s));
    const int wgrid = [](int nb) { return std::min(2048, (nb + 3) / 4); }(batch);
    for (int step = 0; step < n_steps; ++step) {
        const long long off = (long long)step * batch;
        const int nb = (int)std::min<long long>(batch, n - off);
        PairSrc p{users, items, nullptr, 0, off};
        const unsigned long long call = (unsigned long long)(adam_t0 + step);
        rc = ncf_forward_chunk(d, p, nb, call, s, nullptr, true);
        if (rc) return rc;
        float *xl = d.acts + act_off(d, L, d.max_batch), *dxl = d.dacts + act_off(d, L, d.max_batch);
        hipLaunchKernelGGL(ncf_predict_kernel, dim3(std::min(RK_LOSS_PARTIALS, (nb + 3) / 4)), dim3(256), 0, s, p, nb, f, d.mode, d.ug, d.ig,
                           xl, d.pw, d.pb, (float *)nullptr, labels, d.d0, loss_partials + (size_t)step * RK_LOSS_PARTIALS, dxl, d.grad[0], d.grad[1]);
        RK_CHECK_LAUNCH();
        {
            const int n_slabs = (nb + kWgradSlab - 1) / kWgradSlab;
            hipLaunchKernelGGL(ncf_predict_wgrad_kernel, dim3((PS + 1 + 63) / 64, n_slabs), dim3(1024), 0, s, p, nb, f, d.mode, d.ug, d.ig, xl,
                               d.d0, d.wgrad_part);
            RK_CHECK_LAUNCH();
        }
        // tower backward.  dY of the top layer was masked by its own ReLU in the predict kernel; for the layers below
        // the mask (x > 0, x = the previous layer's ReLU output) is applied in the dX GEMM's epilogue.
        for (int l = (d.mode == RK_NCF_GMF ? -1 : L - 1); l >= 0; --l) {
            const int in = in_of(d, l), out = in / 2;
            float *x = d.acts + act_off(d, l, d.max_batch);
            float *dy = d.dacts + act_off(d, l + 1, d.max_batch), *dx = d.dacts + act_off(d, l, d.max_batch);
            // dW[out,in] += dY^T X : A(m=o,k=b) = dy[b*out+o], B(n=i,k=b) = x[b*in+i].  K = batch is long and
            // the tile count small: split K over workgroups that add into the (zeroed) gradient.
            const int tiles = ((out + 63) / 64) * ((in + 63) / 64);
            const int splits = ((out + 127) / 128) * ((in + 127) / 128) < 384 ? std::max(1, std::min((nb + 31) / 32, 512 / std::max(tiles, 1))) : 1;
            rc = gemm(s, out, in, nb, dy, 1, out, x, 1, in, d.grad[4 + l], in, nullptr, 0, nullptr, 0, splits);
            if (rc) return rc;
            // dX[nb,in] = dY W, masked by (x > 0) for l >= 1 (x is the previous layer's ReLU output)
            rc = gemm_auto(s, nb, in, out, dy, out, 1, d.W[l], 1, in, dx, nullptr, 0, l >= 1 ? x : nullptr, d.gemm_scratch, d.gemm_scratch_floats);
            if (rc) return rc;
            if (d.dropout > 0.f) {
                // back through the Dropout in front of layer l: the forward's mask again (same ids), times 1 / keep_prob.
                // (For l >= 1 the ReLU mask above already zeroed the dropped entries: x is the post-dropout activation.)
                const long long n = (long long)nb * in;
                hipLaunchKernelGGL(dropout_apply_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, s, n, dx, drop_spec(d, call, l));
                RK_CHECK_LAUNCH();
            }
        }
        if (d.mode != RK_NCF_GMF) {
            hipLaunchKernelGGL(ncf_scatter_kernel, dim3(wgrid), dim3(256), 0, s, p, nb, E, d.dacts, d.grad[2], d.grad[3]);
            RK_CHECK_LAUNCH();
        }
        {
            // bias gradients of every layer + the predict layer's weight-gradient slabs: one launch
            ColsumJobs cj;
            memset(&cj, 0, sizeof(cj));
            int widest = 0;
            if (d.mode != RK_NCF_GMF)
                for (int l = 0; l < L; ++l) {
                    const int out = in_of(d, l) / 2, q = cj.n_jobs++;
                    cj.M[q] = nb; cj.N[q] = out; cj.src[q] = d.dacts + act_off(d, l + 1, d.max_batch); cj.dst[q] = d.grad[4 + L + l];
                    widest = std::max(widest, out);
                }
            const int q = cj.n_jobs++;
            cj.M[q] = (nb + kWgradSlab - 1) / kWgradSlab; cj.N[q] = PS + 1; cj.src[q] = d.wgrad_part;
            cj.dst[q] = d.grad[4 + 2 * L]; cj.dst_last[q] = d.grad[5 + 2 * L];
            widest = std::max(widest, PS + 1);
            hipLaunchKernelGGL(colsum_jobs_kernel, dim3((widest + 63) / 64, cj.n_jobs), dim3(1024), 0, s, cj);
            RK_CHECK_LAUNCH();
        }
        if (!apply_update) break;
        const AdamCoef c = adam_coef(adam_t0 + step + 1, d.lr, d.beta1, d.beta2);
        MultiAdam a;
        a.n_tensors = T;
        long long maxn = 0;
        for (int t = 0; t < T; ++t) { a.p[t] = P[t]; a.g[t] = d.grad[t]; a.m[t] = d.m[t]; a.v[t] = d.v[t]; a.n[t] = NN[t]; maxn = std::max(maxn, NN[t]); }
        a.step_size = c.step_size; a.bc2s = c.bc2s; a.b1 = d.beta1; a.b2 = d.beta2; a.eps = d.eps;
        hipLaunchKernelGGL(multi_adam_kernel, dim3((int)std::min<long long>((maxn / 4 + 255) / 256 + 1, 1024), T), dim3(256), 0, s, a);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}

// ---------------------------------------------------------------- top-K over a ready score matrix
extern "C" int rk_topk_rows_impl(float *scores, long long ld, int nb, int n_items, const int *user_ids, const int *seen_ptr,
                                 const int *seen_idx, int K, int *top_ids, float *top_scores, const int *targets,
                                 int n_targets, float *target_score, int *target_rank, hipStream_t s);

RK_EXPORT int rk_topk_rows(float *scores, int32_t nb, int32_t n_items, const int32_t *user_ids, const int32_t *seen_ptr,
                           const int32_t *seen_idx, int32_t K, int32_t *top_ids, float *top_scores,
                           const int32_t *targets, int32_t n_targets, float *target_score, int32_t *target_rank,
                           void *stream)
{
    if (nb <= 0) return RK_OK;
    if (!scores || n_items <= 0 || !user_ids || !seen_ptr || !seen_idx || !top_ids || !top_scores)
        RK_FAIL(RK_EINVAL, "rk_topk_rows: bad arguments");
    return rk_topk_rows_impl(scores, n_items, nb, n_items, user_ids, seen_ptr, seen_idx, K, top_ids, top_scores, targets, n_targets,
                             target_score, target_rank, (hipStream_t)stream);
}
