// LightGCN victim hot path on gfx950: propagate (lightgcn.py:82-113), BPR train step
// (lightgcn.py:137-169) as 2L+1 launches per step, optionally replayed from a hipGraph.
#include <algorithm>

#include <hipcub/hipcub.hpp>

#include "spmm.h"

thread_local char rk_err_buf[512] = "";

struct rk_lightgcn {
    rk_lightgcn_desc d;
    hipStream_t cap_stream = nullptr;
    // hipGraphs of graph_steps, graph_steps/2, graph_steps/4, ... (>= 2) train steps: an epoch is replayed as the longest
    // chunks that fit, so only a last single step is ever launched kernel by kernel
    static constexpr int kExecSlots = 4;
    hipGraphExec_t exec[kExecSlots] = {nullptr, nullptr, nullptr, nullptr};
    int exec_steps[kExecSlots] = {0, 0, 0, 0};
    int exec_update = -1;
    const void *cap_key[4] = {nullptr, nullptr, nullptr, nullptr};  // pointers baked into exec
    // ordered scatter (rk_lightgcn_set_deterministic): the epoch's 3n (row, triplet, role) incidences sorted by
    // (step, row, 3b + role).  Owned by the handle.
    int deterministic = 0;
    unsigned long long *plan_keys[2] = {nullptr, nullptr};
    size_t plan_cap = 0;
    void *plan_tmp = nullptr;
    size_t plan_tmp_bytes = 0;
    const unsigned long long *plan_sorted = nullptr;   // baked into exec
    const void *cap_plan = nullptr;
    int cap_det = 0, cap_batch = 0;
};

__global__ void state_init_kernel(int *state, int step_base, int adam_t, long long n, int batch)
{
    state[ST_STEP_BASE] = step_base;
    state[ST_ADAM_T] = adam_t;
    state[ST_NTRIP_LO] = (int)(unsigned)(n & 0xffffffffLL);
    state[ST_NTRIP_HI] = (int)(n >> 32);
    state[ST_BATCH] = batch;
}

// ---------------------------------------------------------------- BPR forward+backward
struct BprArgs {
    int U, d, L;
    float lam;
    const float *light, *emb;  // emb = [users; items] contiguous
    float *gprop, *gego;
    const int64_t *users, *pos, *neg;
    float *loss_partials;
    const int *state;
    float *coef;  // coef[2k], coef[2k+1] for step k of the chunk
    int k;
    float lr, b1, b2;
    int nb_direct;  // state == nullptr: one batch of nb_direct triplets starting at users[0]
    int light_compact;  // light is a compact [3*nb, d] block: rows b, nb+b, 2nb+b of triplet b (row-sharded trainer)
    // ordered mode (bpr_rows_kernel): the epoch plan, every row's incidences in the order they are added
    const unsigned long long *keys;
};

// incidence key of the epoch plan: step (20 bits) | node row (24 bits) | 3*b + role (20 bits)
static constexpr int kPlanIncBits = 20, kPlanRowBits = 24, kPlanStepBits = 20;
__device__ __forceinline__ unsigned plan_row(unsigned long long key) { return (unsigned)(key >> kPlanIncBits) & ((1u << kPlanRowBits) - 1u); }

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// One wave per triplet, lanes stride over the embedding; 6 row gathers (lightgcn.py:124-129),
// dot products by wave shuffle, softplus / reg (lightgcn.py:149-165), and the scatter-add of
// d(loss)/d(light) and d(reg)/d(E0) with no-return float atomics.
__global__ __launch_bounds__(256) void bpr_kernel(const BprArgs a)
{
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int step = 0, nb = a.nb_direct;
    long long off = 0;
    if (a.state) {
        step = a.state[ST_STEP_BASE] + a.k;
        const long long ntrip = ((long long)(unsigned)a.state[ST_NTRIP_LO]) | ((long long)a.state[ST_NTRIP_HI] << 32);
        const int B = a.state[ST_BATCH];
        off = (long long)step * B;
        nb = (int)max(0LL, min((long long)B, ntrip - off));
    }
    if (a.state && blockIdx.x == 0 && threadIdx.x == 0) {
        const AdamCoef c = adam_coef(a.state[ST_ADAM_T] + a.k + 1, a.lr, a.b1, a.b2);
        a.coef[2 * a.k] = c.step_size;
        a.coef[2 * a.k + 1] = c.bc2s;
    }
    const float invB = nb > 0 ? 1.0f / (float)nb : 0.f;
    const float inv_layers = 1.0f / (float)(a.L + 1);
    const float creg = a.lam * invB;
    const int d = a.d;
    float sp_sum = 0.f, reg_sum = 0.f;
    const int wave_id = blockIdx.x * 4 + w, n_waves = gridDim.x * 4;
    for (int b = wave_id; b < nb; b += n_waves) {
        const long long u = a.users[off + b], p = a.pos[off + b], n = a.neg[off + b];
        const float *lu = a.light + (size_t)u * d, *lp = a.light + (size_t)(a.U + p) * d, *ln = a.light + (size_t)(a.U + n) * d;
        if (a.light_compact) { lu = a.light + (size_t)b * d; lp = a.light + (size_t)(nb + b) * d; ln = a.light + (size_t)(2 * nb + b) * d; }
        const float *eu = a.emb + (size_t)u * d, *ep = a.emb + (size_t)(a.U + p) * d, *en = a.emb + (size_t)(a.U + n) * d;
        float ps = 0.f, ns = 0.f, r = 0.f;
        for (int k = lane; k < d; k += 64) {
            const float xu = lu[k];
            ps += xu * lp[k];
            ns += xu * ln[k];
            const float a0 = eu[k], a1 = ep[k], a2 = en[k];
            r += a0 * a0 + a1 * a1 + a2 * a2;
        }
        ps = wave_sum(ps); ns = wave_sum(ns); r = wave_sum(r);
        const float x = ns - ps;
        sp_sum += softplus_f(x);
        reg_sum += r;
        const float dx = (x > 20.f ? 1.f : 1.f / (1.f + expf(-x))) * invB * inv_layers;
        float *gu = a.gprop + (size_t)u * d, *gp = a.gprop + (size_t)(a.U + p) * d, *gn = a.gprop + (size_t)(a.U + n) * d;
        float *hu = a.gego + (size_t)u * d, *hp = a.gego + (size_t)(a.U + p) * d, *hn = a.gego + (size_t)(a.U + n) * d;
        for (int k = lane; k < d; k += 64) {
            const float xu = lu[k];
            const float du = dx * (ln[k] - lp[k]), dp = -dx * xu, dn = dx * xu;
            unsafeAtomicAdd(gu + k, du);
            unsafeAtomicAdd(gp + k, dp);
            unsafeAtomicAdd(gn + k, dn);
            unsafeAtomicAdd(hu + k, du + creg * eu[k]);
            unsafeAtomicAdd(hp + k, dp + creg * ep[k]);
            unsafeAtomicAdd(hn + k, dn + creg * en[k]);
        }
    }
    if (lane == 0) { red[0][w] = sp_sum; red[1][w] = reg_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float r = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        a.loss_partials[(size_t)step * RK_LOSS_PARTIALS + blockIdx.x] = s * invB + a.lam * (0.5f * r * invB);
    }
}

// Ordered form of bpr_kernel (rk_lightgcn_set_deterministic): one wave per node row that occurs in the minibatch; the row's
// incidences are added in the order of the plan -- triplet index, then role (user, positive, negative): the order a
// sequential loop over the batch produces (oracle: orc_lightgcn_step_general) -- and the row is written with plain stores
// (gprop / gego are zero outside the minibatch's rows).  No atomics: gradients, and with them the trained tables, are
// reproducible bit for bit.
// A popular item occurs 10-30 times in a 1024-triplet batch, and a wave that walks its run one incidence at a time
// (ids -> rows -> two 64-lane reductions -> add) makes the launch 22 us instead of bpr_kernel's 6.  So, per run of up to
// 64 incidences: (1) lane i loads key and triplet ids of incidence i (two dependent round trips for the whole run);
// (2) the triplets' coefficients, sixteen at a time, one per 4-lane group (each triplet is recomputed by the up to three
// rows it touches: cheaper than a second launch handing the coefficients over); (3) the adds, in plan order, LA row
// loads in flight.  The loss terms are added by the user-role incidences.
static constexpr int kRowsGL = 4, kRowsNG = 64 / kRowsGL;   // lanes per coefficient group, groups per wave (16 / 8 / 4 lanes measured: 104.0 / 103.4 / 103.1 us per step)
static constexpr int kRowsWaves = 16;   // 256 workgroups (one loss partial each) x 16 waves: a wave per incidence of a 1024-triplet batch
// Q = ceil(dim / 64) floats per lane and row; 16 / Q rows in flight in phase (3) (more spills at the 128-VGPR cap of a 1024-thread workgroup)
// DIRECT: the op-level form (rk_bpr_rows_ordered, row-sharded trainer): one batch of nb_direct triplets whose node ids are
// gathered positions, `keys` is that batch's sorted plan, light is the compact [3*nb, d] block (rows b, nb+b, 2nb+b).
template <int Q, bool DIRECT = false>
__global__ __launch_bounds__(kRowsWaves * 64) void bpr_rows_kernel(const BprArgs a)
{
    __shared__ float red[2][kRowsWaves];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int step = 0, nb = a.nb_direct;
    long long off = 0;
    if (!DIRECT) {
        step = a.state[ST_STEP_BASE] + a.k;
        const long long ntrip = ((long long)(unsigned)a.state[ST_NTRIP_LO]) | ((long long)a.state[ST_NTRIP_HI] << 32);
        const int B = a.state[ST_BATCH];
        off = (long long)step * B;
        nb = (int)max(0LL, min((long long)B, ntrip - off));
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const AdamCoef c = adam_coef(a.state[ST_ADAM_T] + a.k + 1, a.lr, a.b1, a.b2);
            a.coef[2 * a.k] = c.step_size;
            a.coef[2 * a.k + 1] = c.bc2s;
        }
    }
    const float invB = nb > 0 ? 1.0f / (float)nb : 0.f;
    const float inv_layers = 1.0f / (float)(a.L + 1);
    const float creg = a.lam * invB;
    const int d = a.d, cnt = 3 * nb;
    const unsigned long long *keys = a.keys + 3 * off;
    const unsigned inc_mask = (1u << kPlanIncBits) - 1u;
    const int l4 = lane % kRowsGL, grp = lane / kRowsGL;
    const bool vec4 = (d & 3) == 0;   // rows are 16-byte aligned
    float sp_sum = 0.f, reg_sum = 0.f;   // per 4-lane group; combined in group order at the end
    for (int j = blockIdx.x * kRowsWaves + w; j < cnt; j += gridDim.x * kRowsWaves) {
        // lane i's key of the first 64 incidences from j on, and the one before j: one round trip decides "head of a run"
        const unsigned long long kk0 = (j + lane < cnt) ? keys[j + lane] : ~0ULL;
        const unsigned long long kprev = keys[max(j - 1, 0)];
        const unsigned row = __shfl(plan_row(kk0), 0, 64);
        if (j > 0 && plan_row(kprev) == row) continue;   // not the head of its row's run
        float g[Q], h[Q], e[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { g[q] = 0.f; h[q] = 0.f; }
#pragma unroll
        for (int q = 0; q < Q; ++q) e[q] = (lane + 64 * q < d) ? a.emb[(size_t)row * d + lane + 64 * q] : 0.f;
        for (int j0 = j; j0 < cnt; j0 += 64) {
            // (1) lane i = incidence j0 + i
            const unsigned long long kk = j0 == j ? kk0 : ((j0 + lane < cnt) ? keys[j0 + lane] : ~0ULL);
            const unsigned long long same = __ballot(j0 + lane < cnt && plan_row(kk) == row);
            const int len = (~same == 0ULL) ? 64 : (__ffsll((long long)~same) - 1);   // leading lanes of this row
            if (len == 0) break;
            const unsigned inc = (unsigned)kk & inc_mask;
            const int b = (int)(inc / 3u), my_role = (int)(inc - 3u * (unsigned)b);
            int iu = 0, ip = 0, in_ = 0;   // node rows of the triplet (emb / gradient rows)
            if (lane < len) { iu = (int)a.users[off + b]; ip = a.U + (int)a.pos[off + b]; in_ = a.U + (int)a.neg[off + b]; }
            const int xu = DIRECT ? b : iu, xp = DIRECT ? nb + b : ip, xn = DIRECT ? 2 * nb + b : in_;   // their light rows
            // what lane i adds: coef * (A - B) with role user: dx * (ln - lp); positive: -dx * lu; negative: dx * lu
            const int idx_a = my_role == 0 ? xn : xu, idx_b = my_role == 0 ? xp : -1;
            constexpr int LA = 16 / Q;
            float xa[LA][Q], xb[LA][Q];
            auto load_rows = [&](int c0) {
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    const int src = min(c0 + i, len - 1);   // past the run: re-read its last incidence (unconditional loads issue back to back)
                    const int ra = __shfl(idx_a, src, 64), rb = __shfl(idx_b, src, 64);
                    const float *pa = a.light + (size_t)ra * d, *pb = a.light + (size_t)max(rb, 0) * d;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const int k = lane + 64 * q;
                        xa[i][q] = k < d ? pa[k] : 0.f;
                        xb[i][q] = (k < d && rb >= 0) ? pb[k] : 0.f;
                    }
                }
            };
            load_rows(0);   // independent of the coefficients: in flight under phase (2)
            // (2) coefficients: sixteen 4-lane groups, group `grp` takes incidence r + grp of every round r = 0, 16, 32, 48
            float my_dx = 0.f;
            for (int r = 0; r < len; r += kRowsNG) {
                const int src = min(r + grp, len - 1);
                const bool live = r + grp < len;
                const int ru = __shfl(xu, src, 64), rp = __shfl(xp, src, 64), rn = __shfl(xn, src, 64), ro = __shfl(my_role, src, 64);
                const float *lu = a.light + (size_t)ru * d, *lp = a.light + (size_t)rp * d, *ln = a.light + (size_t)rn * d;
                float ps = 0.f, ns = 0.f;
                if (vec4) {
                    for (int k = l4 * 4; k < d; k += kRowsGL * 4) {
                        const float4 xu = *reinterpret_cast<const float4 *>(lu + k), xp = *reinterpret_cast<const float4 *>(lp + k),
                                     xn = *reinterpret_cast<const float4 *>(ln + k);
                        ps += xu.x * xp.x; ps += xu.y * xp.y; ps += xu.z * xp.z; ps += xu.w * xp.w;
                        ns += xu.x * xn.x; ns += xu.y * xn.y; ns += xu.z * xn.z; ns += xu.w * xn.w;
                    }
                } else {
                    for (int k = l4; k < d; k += kRowsGL) { const float xu = lu[k]; ps += xu * lp[k]; ns += xu * ln[k]; }
                }
#pragma unroll
                for (int o = kRowsGL / 2; o > 0; o >>= 1) { ps += __shfl_xor(ps, o, 64); ns += __shfl_xor(ns, o, 64); }
                const float x = ns - ps;
                const float dx = (x > 20.f ? 1.f : 1.f / (1.f + expf(-x))) * invB * inv_layers;
                if (live && ro == 0) {   // this triplet's loss terms, once
                    const int nu = DIRECT ? __shfl(iu, src, 64) : ru, np_ = DIRECT ? __shfl(ip, src, 64) : rp, nn = DIRECT ? __shfl(in_, src, 64) : rn;
                    const float *eu = a.emb + (size_t)nu * d, *ep = a.emb + (size_t)np_ * d, *en = a.emb + (size_t)nn * d;
                    float rr = 0.f;
                    if (vec4) {
                        for (int k = l4 * 4; k < d; k += kRowsGL * 4) {
                            const float4 a0 = *reinterpret_cast<const float4 *>(eu + k), a1 = *reinterpret_cast<const float4 *>(ep + k),
                                         a2 = *reinterpret_cast<const float4 *>(en + k);
                            rr += a0.x * a0.x + a1.x * a1.x + a2.x * a2.x; rr += a0.y * a0.y + a1.y * a1.y + a2.y * a2.y;
                            rr += a0.z * a0.z + a1.z * a1.z + a2.z * a2.z; rr += a0.w * a0.w + a1.w * a1.w + a2.w * a2.w;
                        }
                    } else {
                        for (int k = l4; k < d; k += kRowsGL) { const float a0 = eu[k], a1 = ep[k], a2 = en[k]; rr += a0 * a0 + a1 * a1 + a2 * a2; }
                    }
#pragma unroll
                    for (int o = kRowsGL / 2; o > 0; o >>= 1) rr += __shfl_xor(rr, o, 64);
                    reg_sum += rr;
                    sp_sum += softplus_f(x);
                }
                const float t = __shfl(dx, ((lane - r) & (kRowsNG - 1)) * kRowsGL, 64);   // incidence `lane` was computed by group lane - r
                if (lane >= r && lane < r + kRowsNG) my_dx = t;
            }
            const float coef = my_role == 1 ? -my_dx : my_dx;
            // (3) the adds, in plan order (the first LA incidences' rows were requested before phase (2))
            for (int c0 = 0; c0 < len; c0 += LA) {
                if (c0 > 0) load_rows(c0);
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    if (c0 + i >= len) break;
                    const float cf = __shfl(coef, c0 + i, 64);
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const float c = cf * (xa[i][q] - xb[i][q]);
                        g[q] += c;
                        h[q] += c + creg * e[q];
                    }
                }
            }
            if (len < 64) break;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int k = lane + 64 * q;
            if (k < d) { a.gprop[(size_t)row * d + k] = g[q]; a.gego[(size_t)row * d + k] = h[q]; }
        }
    }
    // the sixteen groups' loss terms in group order, then the waves' in wave order
    float sp_w = 0.f, rg_w = 0.f;
#pragma unroll
    for (int q = 0; q < kRowsNG; ++q) { sp_w += __shfl(sp_sum, q * kRowsGL, 64); rg_w += __shfl(reg_sum, q * kRowsGL, 64); }
    if (lane == 0) { red[0][w] = sp_w; red[1][w] = rg_w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f, r = 0.f;
        for (int q = 0; q < kRowsWaves; ++q) { s += red[0][q]; r += red[1][q]; }
        a.loss_partials[(size_t)step * RK_LOSS_PARTIALS + blockIdx.x] = s * invB + a.lam * (0.5f * r * invB);
    }
}

// incidence keys of a whole epoch, three per triplet (unsorted)
__global__ void bpr_plan_keys_kernel(const int64_t *users, const int64_t *pos, const int64_t *neg, long long n, int batch, int U,
                                     unsigned long long *keys)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const unsigned long long step = (unsigned long long)(t / batch), b3 = 3ULL * (unsigned long long)(t % batch);
        const unsigned long long hi = step << (kPlanIncBits + kPlanRowBits);
        keys[3 * t + 0] = hi | ((unsigned long long)users[t] << kPlanIncBits) | (b3 + 0);
        keys[3 * t + 1] = hi | ((unsigned long long)(U + pos[t]) << kPlanIncBits) | (b3 + 1);
        keys[3 * t + 2] = hi | ((unsigned long long)(U + neg[t]) << kPlanIncBits) | (b3 + 2);
    }
}

__global__ void zero_f4_kernel(float4 *p4, long long n4, float *p, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = t; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = n4 * 4 + t; i < n; i += stride) p[i] = 0.f;
}

// standalone dense Adam (L == 0 and the MF/NCF tables)
__global__ void adam_kernel(long long n, float *p, const float *g, float *m, float *v, float step_size, float bc2s,
                            float b1, float b2, float eps)
{
    const float w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_elem(pp, mm, vv, g[i], w1, b2, w2, step_size, bc2s, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// n_layers == 0 (valid in the reference: light = E0, lightgcn.py:97-113 with an empty layer loop): the whole
// gradient is gego.  Dense Adam with the step's device-resident coefficients, the gradient buffers cleaned for the
// next step, optional copy of the gradient, state bump -- the epilogue work the last backward SpMM does when L >= 1.
__global__ void adam_l0_kernel(long long n, float *p, float *gego, float *gprop, float *m, float *v, const float *coef,
                               float b1, float b2, float eps, int apply_update, float *grad_out, int *state, int bump)
{
    if (bump && blockIdx.x == 0 && threadIdx.x == 0) {
        state[ST_STEP_BASE] += bump;
        state[ST_ADAM_T] += bump;
    }
    const float w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float g = gego[i];
        gego[i] = 0.f;
        gprop[i] = 0.f;
        if (grad_out) grad_out[i] = g;
        if (apply_update) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam_elem(pp, mm, vv, g, w1, b2, w2, coef[0], coef[1], eps);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

RK_EXPORT int rk_adam_step(int64_t n, float *param, const float *grad, float *m, float *v, int32_t t, float lr,
                           float beta1, float beta2, float eps, void *stream)
{
    if (n <= 0) return RK_OK;
    if (t < 1) RK_FAIL(RK_EINVAL, "rk_adam_step: t must be >= 1");
    const AdamCoef c = adam_coef(t, lr, beta1, beta2);
    const int grid = (int)std::min<long long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (long long)n, param, grad, m, v,
                       c.step_size, c.bc2s, beta1, beta2, eps);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// ---------------------------------------------------------------- handle
static int check_desc(const rk_lightgcn_desc &d)
{
    if (d.n_users <= 0 || d.n_items <= 0 || d.dim <= 0 || d.n_layers < 0) RK_FAIL(RK_EINVAL, "lightgcn: bad sizes");
    if (d.dim > 256) RK_FAIL(RK_EINVAL, "lightgcn: dim %d > 256 unsupported (long-row scratch slots are 256 floats)", d.dim);
    if (!d.rowptr || !d.col || !d.val || !d.wave_desc || d.n_blocks <= 0) RK_FAIL(RK_EINVAL, "lightgcn: graph pointers missing");
    if ((d.n_blocks & kSchedLongFlag) && !d.spmm_scratch) RK_FAIL(RK_EINVAL, "lightgcn: the schedule has long rows: desc.spmm_scratch is required");
    if (!d.user_emb || !d.item_emb || !d.m_user || !d.v_user || !d.m_item || !d.v_item)
        RK_FAIL(RK_EINVAL, "lightgcn: parameter/moment pointers missing");
    if (!d.buf_a || !d.buf_b || !d.light || !d.gprop || !d.gego || !d.state || !d.coef)
        RK_FAIL(RK_EINVAL, "lightgcn: workspace pointers missing");
    const size_t ud = (size_t)d.n_users * d.dim;
    if (d.item_emb != d.user_emb + ud || d.m_item != d.m_user + ud || d.v_item != d.v_user + ud)
        RK_FAIL(RK_EINVAL, "lightgcn: the item table/moments must directly follow the user table/moments in memory "
                           "(one [U+I, dim] allocation; the kernels address E0 with a single base)");
    if (d.keep_prob != 0.f && (!(d.keep_prob > 0.f) || d.keep_prob > 1.f || !d.tpos))
        RK_FAIL(RK_EINVAL, "lightgcn: graph dropout needs 0 < keep_prob <= 1 and the transpose index tpos");
    if (((size_t)d.n_users + d.n_items) * d.dim * sizeof(float) >= (1ULL << 32))
        RK_FAIL(RK_EINVAL, "lightgcn: (U+I)*dim*4 must be < 4 GiB (32-bit gather offsets)");
    return RK_OK;
}

RK_EXPORT int rk_lightgcn_create(const rk_lightgcn_desc *desc, rk_lightgcn_t *out)
{
    if (!desc || !out) RK_FAIL(RK_EINVAL, "rk_lightgcn_create: null argument");
    int rc = check_desc(*desc);
    if (rc) return rc;
    rk_lightgcn *h = new rk_lightgcn();
    h->d = *desc;
    *out = h;
    return RK_OK;
}

RK_EXPORT int rk_lightgcn_destroy(rk_lightgcn_t h)
{
    if (!h) return RK_OK;
    for (auto &e : h->exec) if (e) (void)hipGraphExecDestroy(e);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    for (int i = 0; i < 2; ++i) if (h->plan_keys[i]) (void)hipFree(h->plan_keys[i]);
    if (h->plan_tmp) (void)hipFree(h->plan_tmp);
    delete h;
    return RK_OK;
}

static SpmmArgs base_args(const rk_lightgcn_desc &d)
{
    SpmmArgs a;
    memset(&a, 0, sizeof(a));
    a.n_rows = d.n_users + d.n_items;
    a.rowptr = d.rowptr; a.col = d.col; a.val = d.val; a.wave_desc = reinterpret_cast<const int4 *>(d.wave_desc); a.n_blocks = d.n_blocks;
    a.d = d.dim;
    a.scratch = d.spmm_scratch;
    return a;
}


// graph dropout of one launch (spmm.h, SpmmArgs::drop_*): mode 1/2 = train step k (first forward layer /
// every later launch of the step), mode 3 = explicit mask seed
static void set_dropout(SpmmArgs &a, const rk_lightgcn_desc &d, int mode, int k, unsigned long long seed, bool transposed)
{
    a.drop_thresh24 = (unsigned)((double)d.keep_prob * 16777216.0);
    a.drop_inv_keep = 1.0f / d.keep_prob;
    a.drop_mode = mode; a.drop_k = k; a.drop_seed = seed; a.drop_state = d.state;
    a.drop_tpos = transposed ? d.tpos : nullptr;
}

// forward: light = mean_l A^l E0 ; uses buf_a/buf_b as ping-pong
struct BatchRef {
    const int64_t *users, *pos, *neg;
    int k;
};

// forward: light = mean_l A^l E0.  With a BatchRef (training) the first layer marks the minibatch's
// rows in d.row_bits and the last layer computes only those rows of `light`.
// drop: 0 = none, 1 = the train step's mask (batch->k), 3 = the mask of mask_seed
static int launch_forward(const rk_lightgcn_desc &d, hipStream_t s, const BatchRef *batch = nullptr, int drop = 0,
                          unsigned long long mask_seed = 0ULL)
{
    const int L = d.n_layers;
    const float inv = 1.0f / (float)(L + 1);
    if (L == 0) {
        RK_HIP(hipMemcpyAsync(d.light, d.user_emb, sizeof(float) * (size_t)d.n_users * d.dim, hipMemcpyDeviceToDevice, s));
        RK_HIP(hipMemcpyAsync(d.light + (size_t)d.n_users * d.dim, d.item_emb, sizeof(float) * (size_t)d.n_items * d.dim,
                              hipMemcpyDeviceToDevice, s));
        return RK_OK;
    }
    float *bufs[2] = {d.buf_a, d.buf_b};
    for (int l = 1; l <= L; ++l) {
        SpmmArgs a = base_args(d);
        a.x = (l == 1) ? d.user_emb : bufs[l & 1];
        a.e.y = (l < L) ? bufs[(l + 1) & 1] : nullptr;
        a.e.sum_in = (l == 1) ? d.user_emb : d.light;
        a.e.sum_out = d.light;
        a.e.sum_scale = (l == L) ? inv : 1.0f;
        if (batch && d.row_bits && L >= 2) {
            if (l == 1) {
                a.mark_bits = d.row_bits; a.mark_U = d.n_users; a.mark_k = batch->k; a.mark_state = d.state;
                a.mark_users = batch->users; a.mark_pos = batch->pos; a.mark_neg = batch->neg;
            }
            if (l == L) a.row_filter = d.row_bits;
        }
        if (drop == 1) set_dropout(a, d, l == 1 ? 1 : 2, batch ? batch->k : 0, d.drop_seed, false);
        else if (drop == 3) set_dropout(a, d, 3, 0, rk_drop_step_seed(d.drop_seed, mask_seed), false);
        RK_HIP(spmm_launch(a, s));
    }
    return RK_OK;
}

// backward + Adam for chunk step k; gprop/gego hold the BPR scatter
static int launch_backward(const rk_lightgcn_desc &d, int k, int apply_update, int bump, hipStream_t s)
{
    const int N = d.n_users + d.n_items, L = d.n_layers;
    float *bufs[2] = {d.buf_a, d.buf_b};
    auto fill_adam = [&](SpmmEpi &e) {
        if (apply_update) {
            e.adam = 1;
            e.p = d.user_emb; e.m = d.m_user; e.v = d.v_user;
            e.coef = d.coef + 2 * k;
            e.b1 = d.beta1; e.b2 = d.beta2; e.eps = d.eps;
        }
        e.y = d.grad;  // nullable
        e.state = d.state;
        e.bump = bump;
    };
    if (L == 0) {
        const long long n = (long long)N * d.dim;
        hipLaunchKernelGGL(adam_l0_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, s, n, d.user_emb, d.gego,
                           d.gprop, d.m_user, d.v_user, d.coef + 2 * k, d.beta1, d.beta2, d.eps, apply_update, d.grad, d.state, bump);
        RK_CHECK_LAUNCH();
        return RK_OK;
    }
    for (int j = 1; j <= L; ++j) {
        SpmmArgs a = base_args(d);
        a.x = (j == 1) ? d.gprop : bufs[j & 1];
        const bool last = (j == L);
        if (d.row_bits && L >= 2) {
            if (last) { a.clear_bits = d.row_bits; a.n_words = (N + 31) / 32; }
        }
        if (d.keep_prob > 0.f) set_dropout(a, d, 2, k, d.drop_seed, true);
        a.e.add = last ? d.gego : d.gprop;
        if (last) {
            a.e.zero1 = d.gego;
            a.e.zero2 = (L >= 2) ? d.gprop : nullptr;
            fill_adam(a.e);
        } else {
            a.e.y = bufs[(j + 1) & 1];
        }
        RK_HIP(spmm_launch(a, s));
    }
    if (L == 1) {
        // gprop is the gather operand of the only backward SpMM, so it cannot clean itself there.  A
        // kernel (not hipMemsetAsync): as the trailing node of a replayed hipGraph a memset node was
        // observed to race with the next step's scatter-add on the stream (wrong gradients from the
        // second epoch on, tests/test_gpu_parity.py::test_lightgcn_vs_oracle_shapes[*-1-*]).
        const long long n4 = (long long)N * d.dim / 4;
        hipLaunchKernelGGL(zero_f4_kernel, dim3((int)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, s,
                           reinterpret_cast<float4 *>(d.gprop), n4, d.gprop, (long long)N * d.dim);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}

struct OrderedRef {   // non-null keys: ordered scatter
    const unsigned long long *keys;
    int batch;
};

static int launch_step(const rk_lightgcn_desc &d, const int64_t *users, const int64_t *pos, const int64_t *neg,
                       float *loss_partials, int k, int apply_update, int bump, hipStream_t s, const OrderedRef &ord = OrderedRef{nullptr, 0})
{
    const BatchRef br{users, pos, neg, k};
    int rc = launch_forward(d, s, &br, d.keep_prob > 0.f ? 1 : 0);
    if (rc) return rc;
    BprArgs b;
    b.U = d.n_users; b.d = d.dim; b.L = d.n_layers; b.lam = d.lambda;
    b.light = d.light; b.emb = d.user_emb;
    b.gprop = d.gprop; b.gego = d.gego;
    b.users = users; b.pos = pos; b.neg = neg;
    b.loss_partials = loss_partials;
    b.state = d.state; b.coef = d.coef; b.k = k;
    b.lr = d.lr; b.b1 = d.beta1; b.b2 = d.beta2;
    b.nb_direct = 0; b.light_compact = 0;
    b.keys = ord.keys;
    if (ord.keys) {
        if (d.dim <= 64) hipLaunchKernelGGL(bpr_rows_kernel<1>, dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
        else if (d.dim <= 128) hipLaunchKernelGGL(bpr_rows_kernel<2>, dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
        else hipLaunchKernelGGL(bpr_rows_kernel<4>, dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
    }
    else hipLaunchKernelGGL(bpr_kernel, dim3(RK_LOSS_PARTIALS), dim3(256), 0, s, b);
    RK_CHECK_LAUNCH();
    return launch_backward(d, k, apply_update, bump, s);
}

// hipGraph of `graph_steps` train steps with users/pos/neg/loss_partials baked in; (re)captured when any of
// them, the chunk length or the update flag changed.
static OrderedRef ordered_ref(const rk_lightgcn *h, int batch)
{
    return h->deterministic ? OrderedRef{h->plan_sorted, batch} : OrderedRef{nullptr, 0};
}

static int ensure_exec(rk_lightgcn *h, const int64_t *users, const int64_t *pos, const int64_t *neg, float *loss_partials,
                       int apply_update, int graph_steps, int batch, hipGraphExec_t *out)
{
    const rk_lightgcn_desc &d = h->d;
    const void **cap_key = h->cap_key;
    const OrderedRef ord = ordered_ref(h, batch);
    const bool same_key = h->exec_update == apply_update &&
                          cap_key[0] == users && cap_key[1] == pos && cap_key[2] == neg && cap_key[3] == loss_partials &&
                          h->cap_det == h->deterministic && (!h->deterministic || (h->cap_plan == ord.keys && h->cap_batch == batch));
    if (!same_key) {   // everything baked into the graphs changed: drop them all
        for (int i = 0; i < rk_lightgcn::kExecSlots; ++i) {
            if (h->exec[i]) (void)hipGraphExecDestroy(h->exec[i]);
            h->exec[i] = nullptr; h->exec_steps[i] = 0;
        }
        h->exec_update = apply_update;
        cap_key[0] = users; cap_key[1] = pos; cap_key[2] = neg; cap_key[3] = loss_partials;
        h->cap_det = h->deterministic; h->cap_plan = ord.keys; h->cap_batch = batch;
    }
    int slot = -1;
    for (int i = 0; i < rk_lightgcn::kExecSlots; ++i) {
        if (h->exec[i] && h->exec_steps[i] == graph_steps) { *out = h->exec[i]; return RK_OK; }
        if (!h->exec[i] && slot < 0) slot = i;
    }
    if (slot < 0) {   // all slots taken by other chunk lengths (graph_steps changed between calls): recycle the last
        slot = rk_lightgcn::kExecSlots - 1;
        (void)hipGraphExecDestroy(h->exec[slot]);
        h->exec[slot] = nullptr;
    }
    if (!h->cap_stream) RK_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    hipGraph_t g = nullptr;
    RK_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = RK_OK;
    for (int k = 0; k < graph_steps && rc == RK_OK; ++k)
        rc = launch_step(d, users, pos, neg, loss_partials, k, apply_update, k == graph_steps - 1 ? graph_steps : 0, h->cap_stream, ord);
    hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    RK_HIP(e);
    RK_HIP(hipGraphInstantiate(&h->exec[slot], g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    h->exec_steps[slot] = graph_steps;
    *out = h->exec[slot];
    return RK_OK;
}

// every chunk length an epoch can be cut into: graph_steps, then halves down to 2
static int ensure_all_execs(rk_lightgcn *h, const int64_t *users, const int64_t *pos, const int64_t *neg, float *loss_partials,
                            int apply_update, int graph_steps, int batch, hipStream_t upload_stream)
{
    int n = 0;
    for (int c = graph_steps; c >= 2 && n < rk_lightgcn::kExecSlots; c /= 2, ++n) {
        hipGraphExec_t ex = nullptr;
        int rc = ensure_exec(h, users, pos, neg, loss_partials, apply_update, c, batch, &ex);
        if (rc) return rc;
        if (upload_stream) RK_HIP(hipGraphUpload(ex, upload_stream));
    }
    return RK_OK;
}

// Ordered mode: (re)build the epoch's incidence plan.  Buffers are the handle's own, grown on demand (the first epoch of a
// size pays a hipMalloc; rk_lightgcn_prepare with the epoch's n does it ahead of time).
static int ensure_plan_capacity(rk_lightgcn *h, long long n)
{
    const size_t need = (size_t)3 * (size_t)n;
    if (need > h->plan_cap) {
        for (int i = 0; i < 2; ++i) { if (h->plan_keys[i]) (void)hipFree(h->plan_keys[i]); h->plan_keys[i] = nullptr; }
        if (h->plan_tmp) { (void)hipFree(h->plan_tmp); h->plan_tmp = nullptr; }
        h->plan_cap = 0;
        RK_HIP(hipMalloc(&h->plan_keys[0], need * sizeof(unsigned long long)));
        RK_HIP(hipMalloc(&h->plan_keys[1], need * sizeof(unsigned long long)));
        hipcub::DoubleBuffer<unsigned long long> db(h->plan_keys[0], h->plan_keys[1]);
        size_t bytes = 0;
        RK_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, db, (long long)need, 0, 64, (hipStream_t) nullptr));
        RK_HIP(hipMalloc(&h->plan_tmp, std::max<size_t>(bytes, 16)));
        h->plan_tmp_bytes = bytes;
        h->plan_cap = need;
    }
    return RK_OK;
}

static int build_plan(rk_lightgcn *h, const int64_t *users, const int64_t *pos, const int64_t *neg, long long n, int batch, hipStream_t s)
{
    const rk_lightgcn_desc &d = h->d;
    const long long n_steps = (n + batch - 1) / batch;
    if (n_steps >= (1LL << kPlanStepBits) || (long long)d.n_users + d.n_items >= (1LL << kPlanRowBits) || 3LL * batch >= (1LL << kPlanIncBits))
        RK_FAIL(RK_EINVAL, "deterministic scatter: an epoch of < 2^20 steps, < 2^24 node rows and batches of < 349525 triplets is supported");
    int rc = ensure_plan_capacity(h, n);
    if (rc) return rc;
    hipLaunchKernelGGL(bpr_plan_keys_kernel, dim3((int)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, s, users, pos, neg, n, batch,
                       d.n_users, h->plan_keys[0]);
    RK_CHECK_LAUNCH();
    hipcub::DoubleBuffer<unsigned long long> db(h->plan_keys[0], h->plan_keys[1]);
    size_t bytes = h->plan_tmp_bytes;
    RK_HIP(hipcub::DeviceRadixSort::SortKeys(h->plan_tmp, bytes, db, 3 * n, 0, 64, s));
    h->plan_sorted = db.Current();
    return RK_OK;
}

RK_EXPORT int rk_lightgcn_set_deterministic(rk_lightgcn_t h, int32_t on)
{
    if (!h) RK_FAIL(RK_EINVAL, "rk_lightgcn_set_deterministic: null handle");
    h->deterministic = on ? 1 : 0;
    return RK_OK;
}

RK_EXPORT int rk_lightgcn_prepare(rk_lightgcn_t h, const int64_t *users, const int64_t *pos, const int64_t *neg,
                                  float *loss_partials, int32_t apply_update, int32_t graph_steps, void *stream)
{
    if (!h) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: null handle");
    if (!users || !pos || !neg || !loss_partials) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: bad arguments");
    if (!apply_update && !h->d.grad) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: apply_update=0 needs desc.grad");
    if (graph_steps > RK_MAX_GRAPH_STEPS) graph_steps = RK_MAX_GRAPH_STEPS;
    if (graph_steps <= 1) return RK_OK;
    if (h->deterministic) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: in deterministic mode the graph depends on the epoch's plan; the first rk_lightgcn_train_epoch captures it");
    hipStream_t s = (hipStream_t)stream;
    hipStream_t up = nullptr;
    if (!s) { if (!h->cap_stream) RK_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking)); up = h->cap_stream; } else up = s;
    return ensure_all_execs(h, users, pos, neg, loss_partials, apply_update, graph_steps, 0, up);
}

RK_EXPORT int rk_lightgcn_propagate(rk_lightgcn_t h, void *stream)
{
    if (!h) RK_FAIL(RK_EINVAL, "rk_lightgcn_propagate: null handle");
    return launch_forward(h->d, (hipStream_t)stream);
}

RK_EXPORT int rk_lightgcn_propagate_dropout(rk_lightgcn_t h, uint64_t mask_seed, void *stream)
{
    if (!h) RK_FAIL(RK_EINVAL, "rk_lightgcn_propagate_dropout: null handle");
    if (!(h->d.keep_prob > 0.f)) RK_FAIL(RK_EINVAL, "rk_lightgcn_propagate_dropout: desc.keep_prob is 0");
    return launch_forward(h->d, (hipStream_t)stream, nullptr, 3, (unsigned long long)mask_seed);
}

RK_EXPORT int rk_lightgcn_train_epoch(rk_lightgcn_t h, const int64_t *users, const int64_t *pos, const int64_t *neg,
                                      int64_t n, int32_t batch, int32_t adam_t0, float *loss_partials,
                                      int32_t apply_update, int32_t graph_steps, void *stream)
{
    if (!h) RK_FAIL(RK_EINVAL, "rk_lightgcn_train_epoch: null handle");
    if (n <= 0 || batch <= 0 || !users || !pos || !neg || !loss_partials)
        RK_FAIL(RK_EINVAL, "rk_lightgcn_train_epoch: bad arguments");
    if (!apply_update && !h->d.grad) RK_FAIL(RK_EINVAL, "rk_lightgcn_train_epoch: apply_update=0 needs desc.grad");
    hipStream_t s = (hipStream_t)stream;
    const rk_lightgcn_desc &d = h->d;
    const int n_steps = (int)((n + batch - 1) / batch);
    const int N = d.n_users + d.n_items;
    // scatter targets start (and, by the self-cleaning epilogues, stay) zero
    RK_HIP(hipMemsetAsync(d.gprop, 0, sizeof(float) * (size_t)N * d.dim, s));
    RK_HIP(hipMemsetAsync(d.gego, 0, sizeof(float) * (size_t)N * d.dim, s));
    if (d.row_bits) RK_HIP(hipMemsetAsync(d.row_bits, 0, sizeof(uint32_t) * (size_t)((N + 31) / 32), s));
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(1), 0, s, d.state, 0, adam_t0, (long long)n, batch);
    RK_CHECK_LAUNCH();

    if (h->deterministic) {
        int rc = build_plan(h, users, pos, neg, n, batch, s);
        if (rc) return rc;
    }
    int done = 0;
    if (graph_steps > RK_MAX_GRAPH_STEPS) graph_steps = RK_MAX_GRAPH_STEPS;
    if (graph_steps > 1 && n_steps >= 2) {
        int slots = 0;
        for (int c = graph_steps; c >= 2 && slots < rk_lightgcn::kExecSlots; c /= 2, ++slots) {
            if (done + c > n_steps) continue;
            hipGraphExec_t ex = nullptr;
            int rc = ensure_exec(h, users, pos, neg, loss_partials, apply_update, c, batch, &ex);
            if (rc) return rc;
            for (; done + c <= n_steps; done += c) RK_HIP(hipGraphLaunch(ex, s));
        }
    }
    const OrderedRef ord = ordered_ref(h, batch);
    for (; done < n_steps; ++done) {
        int rc = launch_step(d, users, pos, neg, loss_partials, 0, apply_update, 1, s, ord);
        if (rc) return rc;
    }
    return RK_OK;
}


// ---------------------------------------------------------------- op-level entry points
// (the row-sharded multi-GPU trainer composes a step from these between its collectives)
__global__ void set_coef_kernel(float *coef, float step_size, float bc2s)
{
    coef[0] = step_size;
    coef[1] = bc2s;
}

RK_EXPORT int rk_spmm_csr_ex(int32_t n_rows, const int32_t *rowptr, const int32_t *col, const float *val,
                             const int32_t *wave_desc, int32_t n_blocks, int32_t *scratch, int32_t dim, const float *x,
                             int64_t x_rows, const rk_spmm_epilogue *epi, void *stream)
{
    if (n_rows <= 0 || dim <= 0 || dim > 256 || !rowptr || !col || !val || !wave_desc || n_blocks <= 0 || !x || !epi)
        RK_FAIL(RK_EINVAL, "rk_spmm_csr_ex: bad arguments");
    if ((size_t)x_rows * dim * sizeof(float) >= (1ULL << 32)) RK_FAIL(RK_EINVAL, "rk_spmm_csr_ex: x_rows*dim*4 must be < 4 GiB");
    hipStream_t s = (hipStream_t)stream;
    SpmmArgs a;
    memset(&a, 0, sizeof(a));
    a.n_rows = n_rows; a.rowptr = rowptr; a.col = col; a.val = val;
    a.wave_desc = reinterpret_cast<const int4 *>(wave_desc); a.n_blocks = n_blocks; a.d = dim; a.x = x;
    if ((n_blocks & kSchedLongFlag) && !scratch) RK_FAIL(RK_EINVAL, "rk_spmm_csr_ex: this schedule has long rows and needs its scratch block");
    a.scratch = scratch;
    a.e.add = epi->add; a.e.y = epi->y; a.e.sum_in = epi->sum_in; a.e.sum_out = epi->sum_out; a.e.sum_scale = epi->sum_scale;
    a.e.zero1 = epi->zero1; a.e.zero2 = epi->zero2;
    if (epi->sum_out && !epi->sum_in) RK_FAIL(RK_EINVAL, "rk_spmm_csr_ex: sum_out needs sum_in");
    if (epi->adam_t > 0) {
        if (!epi->adam_p || !epi->adam_m || !epi->adam_v || !epi->coef_scratch) RK_FAIL(RK_EINVAL, "rk_spmm_csr_ex: Adam pointers missing");
        const AdamCoef c = adam_coef(epi->adam_t, epi->lr, epi->beta1, epi->beta2);
        hipLaunchKernelGGL(set_coef_kernel, dim3(1), dim3(1), 0, s, epi->coef_scratch, c.step_size, c.bc2s);
        RK_CHECK_LAUNCH();
        a.e.adam = 1; a.e.p = epi->adam_p; a.e.m = epi->adam_m; a.e.v = epi->adam_v; a.e.coef = epi->coef_scratch;
        a.e.b1 = epi->beta1; a.e.b2 = epi->beta2; a.e.eps = epi->eps;
    }
    RK_HIP(spmm_launch(a, s));
    return RK_OK;
}

RK_EXPORT int rk_bpr_rows(int32_t dim, int32_t n_layers, float lambda, const float *light, int32_t light_compact, const float *emb,
                          float *gprop, float *gego, const int64_t *rows_u, const int64_t *rows_p,
                          const int64_t *rows_n, int32_t nb, float *loss_partials, void *stream)
{
    if (dim <= 0 || n_layers < 0 || !light || !emb || !gprop || !gego || !rows_u || !rows_p || !rows_n || nb <= 0 || !loss_partials)
        RK_FAIL(RK_EINVAL, "rk_bpr_rows: bad arguments");
    BprArgs b;
    memset(&b, 0, sizeof(b));
    b.U = 0; b.d = dim; b.L = n_layers; b.lam = lambda;
    b.light = light; b.emb = emb; b.gprop = gprop; b.gego = gego;
    b.users = rows_u; b.pos = rows_p; b.neg = rows_n;
    b.loss_partials = loss_partials; b.state = nullptr; b.coef = nullptr; b.k = 0; b.nb_direct = nb; b.light_compact = light_compact ? 1 : 0;
    b.keys = nullptr;
    hipLaunchKernelGGL(bpr_kernel, dim3(RK_LOSS_PARTIALS), dim3(256), 0, (hipStream_t)stream, b);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

RK_EXPORT int rk_bpr_rows_ordered(int32_t dim, int32_t n_layers, float lambda, const float *light, const float *emb, float *gprop,
                                  float *gego, const int64_t *rows_u, const int64_t *rows_p, const int64_t *rows_n, int32_t nb,
                                  const uint64_t *keys, float *loss_partials, void *stream)
{
    if (dim <= 0 || dim > 256 || n_layers < 0 || !light || !emb || !gprop || !gego || !rows_u || !rows_p || !rows_n || nb <= 0 || !keys || !loss_partials)
        RK_FAIL(RK_EINVAL, "rk_bpr_rows_ordered: bad arguments");
    if (3LL * nb >= (1LL << kPlanIncBits)) RK_FAIL(RK_EINVAL, "rk_bpr_rows_ordered: batch of < 349525 triplets");
    BprArgs b;
    memset(&b, 0, sizeof(b));
    b.U = 0; b.d = dim; b.L = n_layers; b.lam = lambda;
    b.light = light; b.emb = emb; b.gprop = gprop; b.gego = gego;
    b.users = rows_u; b.pos = rows_p; b.neg = rows_n;
    b.loss_partials = loss_partials; b.state = nullptr; b.coef = nullptr; b.k = 0; b.nb_direct = nb; b.light_compact = 1;
    b.keys = reinterpret_cast<const unsigned long long *>(keys);
    hipStream_t s = (hipStream_t)stream;
    if (dim <= 64) hipLaunchKernelGGL((bpr_rows_kernel<1, true>), dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
    else if (dim <= 128) hipLaunchKernelGGL((bpr_rows_kernel<2, true>), dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
    else hipLaunchKernelGGL((bpr_rows_kernel<4, true>), dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
    RK_CHECK_LAUNCH();
    return RK_OK;
}
