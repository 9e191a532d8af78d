// LightGCN victim hot path on gfx950: propagate (lightgcn.py:82-113), BPR train step
// (lightgcn.py:137-169) as 2L+1 launches per step, optionally replayed from a hipGraph.
#include <algorithm>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "spmm.h"
#include "spmm_lds.h"

thread_local char rk_err_buf[512] = "";
static_assert(LS_WORDS <= RK_LDS_SYNC_WORDS && LS_ERR == RK_LDS_SYNC_ERR, "include/recad_hip.h and spmm_lds.h disagree about the sync words");

struct rk_lightgcn {
    rk_lightgcn_desc d;
    hipStream_t cap_stream = nullptr;
    // Captured train steps.  Nothing the caller passes per epoch is baked in (the triplet / loss pointers travel through the
    // device state block), so an exec stays valid for the life of the handle.  `whole` graphs are one complete
    // rk_lightgcn_train_epoch call of n_steps steps (zeroing of the scatter targets and -- LDS path -- the layout
    // conversions included); the others are chunks of a longer epoch.  LRU over kExecSlots.
    static constexpr int kExecSlots = 8;
    struct Exec {
        hipGraphExec_t exec = nullptr;
        int n_steps = 0, whole = 0, update = 0, det = 0, batch = 0;
        const void *plan = nullptr;
        unsigned long long stamp = 0;
    } exec[kExecSlots];
    unsigned long long clock = 0;
    // ordered scatter (rk_lightgcn_set_deterministic): the epoch's 3n (row, triplet, role) incidences sorted by
    // (step, row, 3b + role).  Owned by the handle.
    int deterministic = 0;
    unsigned long long *plan_keys[2] = {nullptr, nullptr};
    size_t plan_cap = 0;
    void *plan_tmp = nullptr;
    size_t plan_tmp_bytes = 0;
    const unsigned long long *plan_sorted = nullptr;   // baked into exec
};

__global__ void state_init_kernel(int *state, int step_base, int adam_t, long long n, int batch, const int64_t *users,
                                  const int64_t *pos, const int64_t *neg, float *loss_partials)
{
    state[ST_STEP_BASE] = step_base;
    state[ST_ADAM_T] = adam_t;
    state[ST_NTRIP_LO] = (int)(unsigned)(n & 0xffffffffLL);
    state[ST_NTRIP_HI] = (int)(n >> 32);
    state[ST_BATCH] = batch;
    unsigned long long *q = reinterpret_cast<unsigned long long *>(state);
    q[ST_PTR_USERS / 2] = reinterpret_cast<unsigned long long>(users);
    q[ST_PTR_POS / 2] = reinterpret_cast<unsigned long long>(pos);
    q[ST_PTR_NEG / 2] = reinterpret_cast<unsigned long long>(neg);
    q[ST_PTR_LOSS / 2] = reinterpret_cast<unsigned long long>(loss_partials);
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
    // gprop / gego in the sliced layout of spmm_lds.h (the LDS-resident propagation gathers them by slice)
    int sliced;
    LdsDims sl;
    // LDS path: the reg gradient is applied in closed form by the last backward launch (LdsEpi::cnt): only gprop is
    // scattered, every node's incidences are counted, and the step's lambda / nb goes to creg_out
    int *cnt;
    float *creg_out;
};

// float offset of element (row, k) of a gradient buffer
__device__ __forceinline__ size_t grad_off(const BprArgs &a, int row, int k)
{
    return a.sliced ? sl_off(a.sl, row, k) : (size_t)row * a.d + k;
}

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
    const int64_t *users = a.users, *pos = a.pos, *neg = a.neg;
    float *loss_partials = a.loss_partials;
    if (a.state) {   // a train step of an epoch: batch window and buffers come from the device state block
        step = a.state[ST_STEP_BASE] + a.k;
        const long long ntrip = ((long long)(unsigned)a.state[ST_NTRIP_LO]) | ((long long)a.state[ST_NTRIP_HI] << 32);
        const int B = a.state[ST_BATCH];
        off = (long long)step * B;
        nb = (int)max(0LL, min((long long)B, ntrip - off));
        users = st_ptr<const int64_t>(a.state, ST_PTR_USERS); pos = st_ptr<const int64_t>(a.state, ST_PTR_POS);
        neg = st_ptr<const int64_t>(a.state, ST_PTR_NEG); loss_partials = st_ptr<float>(a.state, ST_PTR_LOSS);
    }
    if (a.state && blockIdx.x == 0 && threadIdx.x == 0) {
        const AdamCoef c = adam_coef(a.state[ST_ADAM_T] + a.k + 1, a.lr, a.b1, a.b2);
        a.coef[2 * a.k] = c.step_size;
        a.coef[2 * a.k + 1] = c.bc2s;
    }
    const float invB = nb > 0 ? 1.0f / (float)nb : 0.f;
    const float inv_layers = 1.0f / (float)(a.L + 1);
    const float creg = a.lam * invB;
    if (a.creg_out && blockIdx.x == 0 && threadIdx.x == 0) a.creg_out[0] = creg;
    const int d = a.d;
    float sp_sum = 0.f, reg_sum = 0.f;
    const int wave_id = blockIdx.x * 4 + w, n_waves = gridDim.x * 4;
    for (int b = wave_id; b < nb; b += n_waves) {
        const long long u = users[off + b], p = pos[off + b], n = neg[off + b];
        const float *lu = a.light + (size_t)u * d, *lp = a.light + (size_t)(a.U + p) * d, *ln = a.light + (size_t)(a.U + n) * d;
        if (a.light_compact) { lu = a.light + (size_t)b * d; lp = a.light + (size_t)(nb + b) * d; ln = a.light + (size_t)(2 * nb + b) * d; }
        float ps = 0.f, ns = 0.f, r = 0.f;
        for (int k = lane; k < d; k += 64) {
            const float xu = lu[k];
            ps += xu * lp[k];
            ns += xu * ln[k];
            const float a0 = a.emb[grad_off(a, (int)u, k)], a1 = a.emb[grad_off(a, a.U + (int)p, k)], a2 = a.emb[grad_off(a, a.U + (int)n, k)];
            r += a0 * a0 + a1 * a1 + a2 * a2;
        }
        ps = wave_sum(ps); ns = wave_sum(ns); r = wave_sum(r);
        const float x = ns - ps;
        sp_sum += softplus_f(x);
        reg_sum += r;
        const float dx = (x > 20.f ? 1.f : 1.f / (1.f + expf(-x))) * invB * inv_layers;
        for (int k = lane; k < d; k += 64) {
            const float xu = lu[k];
            const float du = dx * (ln[k] - lp[k]), dp = -dx * xu, dn = dx * xu;
            const size_t ou = grad_off(a, (int)u, k), op = grad_off(a, a.U + (int)p, k), on = grad_off(a, a.U + (int)n, k);
            unsafeAtomicAdd(a.gprop + ou, du);
            unsafeAtomicAdd(a.gprop + op, dp);
            unsafeAtomicAdd(a.gprop + on, dn);
            if (!a.cnt) {
                unsafeAtomicAdd(a.gego + ou, du + creg * a.emb[ou]);
                unsafeAtomicAdd(a.gego + op, dp + creg * a.emb[op]);
                unsafeAtomicAdd(a.gego + on, dn + creg * a.emb[on]);
            }
        }
        if (a.cnt && lane < 3) atomicAdd(a.cnt + (lane == 0 ? (int)u : a.U + (int)(lane == 1 ? p : n)), 1);
    }
    if (lane == 0) { red[0][w] = sp_sum; red[1][w] = reg_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float r = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        loss_partials[(size_t)step * RK_LOSS_PARTIALS + blockIdx.x] = s * invB + a.lam * (0.5f * r * invB);
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
    const int64_t *users = a.users, *pos = a.pos, *neg = a.neg;
    float *loss_partials = a.loss_partials;
    if (!DIRECT) {
        users = st_ptr<const int64_t>(a.state, ST_PTR_USERS); pos = st_ptr<const int64_t>(a.state, ST_PTR_POS);
        neg = st_ptr<const int64_t>(a.state, ST_PTR_NEG); loss_partials = st_ptr<float>(a.state, ST_PTR_LOSS);
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
        for (int q = 0; q < Q; ++q) e[q] = (lane + 64 * q < d) ? a.emb[grad_off(a, (int)row, lane + 64 * q)] : 0.f;
        for (int j0 = j; j0 < cnt; j0 += 64) {
            // (1) lane i = incidence j0 + i
            const unsigned long long kk = j0 == j ? kk0 : ((j0 + lane < cnt) ? keys[j0 + lane] : ~0ULL);
            const unsigned long long same = __ballot(j0 + lane < cnt && plan_row(kk) == row);
            const int len = (~same == 0ULL) ? 64 : (__ffsll((long long)~same) - 1);   // leading lanes of this row
            if (len == 0) break;
            const unsigned inc = (unsigned)kk & inc_mask;
            const int b = (int)(inc / 3u), my_role = (int)(inc - 3u * (unsigned)b);
            int iu = 0, ip = 0, in_ = 0;   // node rows of the triplet (emb / gradient rows)
            if (lane < len) { iu = (int)users[off + b]; ip = a.U + (int)pos[off + b]; in_ = a.U + (int)neg[off + b]; }
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
                    float rr = 0.f;
                    if (vec4) {   // (four consecutive columns from a multiple of 4 are contiguous in the sliced layout too)
                        for (int k = l4 * 4; k < d; k += kRowsGL * 4) {
                            const float4 a0 = *reinterpret_cast<const float4 *>(a.emb + grad_off(a, nu, k)), a1 = *reinterpret_cast<const float4 *>(a.emb + grad_off(a, np_, k)),
                                         a2 = *reinterpret_cast<const float4 *>(a.emb + grad_off(a, nn, k));
                            rr += a0.x * a0.x + a1.x * a1.x + a2.x * a2.x; rr += a0.y * a0.y + a1.y * a1.y + a2.y * a2.y;
                            rr += a0.z * a0.z + a1.z * a1.z + a2.z * a2.z; rr += a0.w * a0.w + a1.w * a1.w + a2.w * a2.w;
                        }
                    } else {
                        for (int k = l4; k < d; k += kRowsGL) { const float a0 = a.emb[grad_off(a, nu, k)], a1 = a.emb[grad_off(a, np_, k)], a2 = a.emb[grad_off(a, nn, k)]; rr += a0 * a0 + a1 * a1 + a2 * a2; }
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
            if (k < d) { const size_t o = grad_off(a, (int)row, k); a.gprop[o] = g[q]; a.gego[o] = h[q]; }
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
        loss_partials[(size_t)step * RK_LOSS_PARTIALS + blockIdx.x] = s * invB + a.lam * (0.5f * r * invB);
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

// the same with the step's coefficients {lr / (1 - b1^t), sqrt(1 - b2^t)} read from device memory (rk_adam_coef_advance): a
// captured step cannot take t from the host
__global__ __launch_bounds__(256) void adam_dev_kernel(long long n, float *p, const float *g, float *m, float *v, const float *coef, float b1, float b2, float eps)
{
    const float w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
    const float step_size = coef[0], bc2s = coef[1];
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    // 16 bytes per lane when the four arrays allow it (one float per lane keeps too few bytes in flight for the HBM rate)
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    long long done = 0;
    if (vec) {
        const long long n4 = n >> 2;
        float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        for (long long i = tid; i < n4; i += nth) {
            const float4 gg = g4[i];
            float4 pp = p4[i], mm = m4[i], vv = v4[i];
            adam_elem(pp.x, mm.x, vv.x, gg.x, w1, b2, w2, step_size, bc2s, eps);
            adam_elem(pp.y, mm.y, vv.y, gg.y, w1, b2, w2, step_size, bc2s, eps);
            adam_elem(pp.z, mm.z, vv.z, gg.z, w1, b2, w2, step_size, bc2s, eps);
            adam_elem(pp.w, mm.w, vv.w, gg.w, w1, b2, w2, step_size, bc2s, eps);
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
        }
        done = n4 << 2;
    }
    for (long long i = done + tid; i < n; i += nth) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_elem(pp, mm, vv, g[i], w1, b2, w2, step_size, bc2s, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

RK_EXPORT int rk_adam_step_dev(int64_t n, float *param, const float *grad, float *m, float *v, const float *coef, float beta1, float beta2,
                               float eps, void *stream)
{
    if (n <= 0) return RK_OK;
    if (!param || !grad || !m || !v || !coef) RK_FAIL(RK_EINVAL, "rk_adam_step_dev: bad arguments");
    const int grid = (int)std::min<long long>((n / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (long long)n, param, grad, m, v, coef, beta1, beta2, eps);
    RK_CHECK_LAUNCH();
    return RK_OK;
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
    const bool lds_only = d.lds_plan && d.n_layers >= 1 && !(d.keep_prob > 0.f);   // every SpMM of the handle takes the LDS kernel
    if (!d.rowptr || !d.col || !d.val) RK_FAIL(RK_EINVAL, "lightgcn: graph pointers missing");
    if (!lds_only && (!d.wave_desc || d.n_blocks <= 0)) RK_FAIL(RK_EINVAL, "lightgcn: the SpMM schedule (wave_desc, n_blocks) is missing");
    if (!lds_only && (d.n_blocks & kSchedLongFlag) && !d.spmm_scratch) RK_FAIL(RK_EINVAL, "lightgcn: the schedule has long rows: desc.spmm_scratch is required");
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
    if (d.lds_plan) {
        if (!d.lsum || !d.e0s || !d.ms || !d.vs) RK_FAIL(RK_EINVAL, "lightgcn: lds_plan needs the lsum, e0s, ms and vs work buffers");
        if (d.lds_info.n_users != d.n_users || d.lds_info.n_items != d.n_items || d.lds_info.dim != d.dim || d.lds_info.n_wg <= 0 ||
            d.lds_info.wgx_ofs <= 0 || d.lds_info.dinv_ofs <= 0)
            RK_FAIL(RK_EINVAL, "lightgcn: lds_info does not describe this graph / dim");
        if (reinterpret_cast<uintptr_t>(d.lds_plan) & 15) RK_FAIL(RK_EINVAL, "lightgcn: lds_plan must be 16-byte aligned");
        if (d.lds_sync && (reinterpret_cast<uintptr_t>(d.lds_sync) & 127)) RK_FAIL(RK_EINVAL, "lightgcn: lds_sync must be 128-byte aligned");
    }
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

RK_EXPORT int rk_lightgcn_sync_status(rk_lightgcn_t h, int32_t *status, void *stream)
{
    if (!h || !status) RK_FAIL(RK_EINVAL, "rk_lightgcn_sync_status: bad arguments");
    *status = 0;
    if (!h->d.lds_sync) return RK_OK;
    RK_HIP(hipMemcpyAsync(status, h->d.lds_sync + LS_ERR, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    RK_HIP(hipStreamSynchronize((hipStream_t)stream));
    // read-and-clear: the word is sticky across launches and call prologues (nothing else resets it), so a reported timeout
    // does not poison the calls after it
    if (*status) RK_HIP(hipMemsetAsync(h->d.lds_sync + LS_ERR, 0, sizeof(int32_t), (hipStream_t)stream));
    return RK_OK;
}

RK_EXPORT int rk_lightgcn_destroy(rk_lightgcn_t h)
{
    if (!h) return RK_OK;
    for (auto &e : h->exec) if (e.exec) (void)hipGraphExecDestroy(e.exec);
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

// ---------------------------------------------------------------- LDS-resident propagation (spmm_lds.h)
// Used when desc.lds_plan is given, the graph is not dropped out and L >= 1.  buf_a / buf_b / gprop / gego / lsum / e0s are
// SLICED; E0 / m / v / light / grad stay row-major.
static bool use_lds(const rk_lightgcn_desc &d) { return d.lds_plan && d.n_layers >= 1 && !(d.keep_prob > 0.f); }

static LdsInfo lds_info(const rk_lightgcn_desc &d)
{
    LdsInfo li;
    li.n_wg = d.lds_info.n_wg; li.lds_bytes = d.lds_info.lds_bytes; li.lpa = d.lds_info.lpa; li.lpb = d.lds_info.lpb;
    li.U = d.n_users; li.I = d.n_items; li.d = d.dim; li.lsu = d.lds_info.lsu; li.lsi = d.lds_info.lsi;
    li.wgx_ofs = d.lds_info.wgx_ofs; li.dinv_ofs = d.lds_info.dinv_ofs; li.perm0 = d.lds_info.perm0_ofs; li.perm1 = d.lds_info.perm1_ofs;
    li.mq_ofs = d.lds_info.mq_ofs;
    return li;
}

// Sliced working copies.  During a train_epoch call E0 and the Adam moments LIVE in e0s / ms / vs (the fused Adam's
// 32-byte pieces of row-major rows cost the last backward launch 7 of 17.6 us); the row-major tensors the caller owns
// are read once at the start of the call (with_moments) and written back once at its end.  A propagate call only needs e0s.
static int lds_sync(const rk_lightgcn_desc &d, bool with_moments, int to_sliced, hipStream_t s, bool clear_scatter = false, bool clear_gego = true)
{
    const LdsInfo li = lds_info(d);
    const LdsDims g{li.U, li.I, li.d, li.lsu, li.lsi};
    const long long n = (long long)(li.U + li.I) * (li.d / 4);
    LdsPackJob job;
    memset(&job, 0, sizeof(job));
    job.n = with_moments ? 3 : 1;
    job.rm[0] = d.user_emb; job.sl[0] = d.e0s;
    job.rm[1] = d.m_user; job.sl[1] = d.ms;
    job.rm[2] = d.v_user; job.sl[2] = d.vs;
    if (clear_scatter) {   // a train call's prologue: the scatter targets start (and, by the self-cleaning epilogues, stay) zero
        job.zero[0] = d.gprop;
        job.zero[1] = clear_gego ? d.gego : nullptr;   // (only the ordered scatter writes gego on this path)
        job.zero_i = d.cnt;
    }
    if (to_sliced) job.zero_sync = d.lds_sync;   // (nullable) a call starts from clean hand-off counters
    hipLaunchKernelGGL(lds_pack_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, s, g, job, to_sliced);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// phases [p0, p1) of a pass as ONE multi-phase launch (spmm_lds_multi_kernel); every phase but the last publishes
static int launch_lds_multi(const rk_lightgcn_desc &d, const LdsInfo &li, const LdsArgs *ph, int n, hipStream_t s)
{
    for (int p0 = 0; p0 < n; p0 += kLdsMaxPhases) {
        const int m = std::min(kLdsMaxPhases, n - p0);
        if (m == 1) { RK_HIP(spmm_lds_launch(li, ph[p0], s)); continue; }
        LdsMultiArgs ma;
        memset(&ma, 0, sizeof(ma));
        ma.plan = d.lds_plan; ma.sync = d.lds_sync; ma.n_phases = m; ma.h = li.hdr();
        // tuning builds: RK_LDS_MSTAMPS=1 puts per-item wall-clock stamps behind the sync words (the probe over-allocates them)
        if (RK_TUNE_INT("RK_LDS_MSTAMPS", 0) && d.lds_sync) ma.stamps = reinterpret_cast<unsigned long long *>(d.lds_sync + RK_LDS_SYNC_WORDS);
        for (int k = 0; k < m; ++k) {
            ma.ph[k].x = ph[p0 + k].x;
            ma.ph[k].e = ph[p0 + k].e;
            ma.ph[k].e.publish = (k + 1 < m) ? 1 : 0;
            if (ma.ph[k].e.bump && k + 1 < m) RK_FAIL(RK_EINVAL, "internal: only the last phase of a launch may bump the step counter");
        }
        RK_HIP(spmm_lds_multi_launch(li, ma, s));
    }
    return RK_OK;
}

static bool lds_fused(const rk_lightgcn_desc &d) { return d.lds_sync != nullptr && d.n_layers >= 2; }

static int launch_forward_lds(const rk_lightgcn_desc &d, hipStream_t s, bool training)
{
    const int L = d.n_layers;
    const LdsInfo li = lds_info(d);
    float *bufs[2] = {d.buf_a, d.buf_b};
    std::vector<LdsArgs> ph((size_t)L);
    for (int l = 1; l <= L; ++l) {
        LdsArgs &a = ph[(size_t)l - 1];
        memset(&a, 0, sizeof(a));
        a.plan = d.lds_plan; a.h = li.hdr();
        a.x = (l == 1) ? d.e0s : bufs[l & 1];
        a.e.y = (l < L) ? bufs[(l + 1) & 1] : nullptr;
        a.e.sum_in = (l == 1) ? d.e0s : d.lsum;
        a.e.sum_out = (l == L) ? d.light : d.lsum;
        a.e.sum_rm = (l == L) ? 1 : 0;
        a.e.sum_scale = (l == L) ? 1.0f / (float)(L + 1) : 1.0f;
        if (training && l == 1) a.e.zero_cnt = d.cnt;   // (nullable) the incidence counts of the previous step
        if (!lds_fused(d)) { a.e.y_staged = (l < L) ? 1 : 0; a.e.x_staged = (l > 1) ? 1 : 0; }   // (the intermediate layers: spmm_lds.h LdsEpi)
    }
    if (lds_fused(d)) return launch_lds_multi(d, li, ph.data(), L, s);
    for (int l = 0; l < L; ++l) RK_HIP(spmm_lds_launch(li, ph[(size_t)l], s));
    return RK_OK;
}

static int launch_backward_lds(const rk_lightgcn_desc &d, int k, int apply_update, int bump, hipStream_t s, bool counted)
{
    const int N = d.n_users + d.n_items, L = d.n_layers;
    const LdsInfo li = lds_info(d);
    float *bufs[2] = {d.buf_a, d.buf_b};
    std::vector<LdsArgs> ph((size_t)L);
    for (int j = 1; j <= L; ++j) {
        LdsArgs &a = ph[(size_t)j - 1];
        memset(&a, 0, sizeof(a));
        a.plan = d.lds_plan; a.h = li.hdr();
        a.x = (j == 1) ? d.gprop : bufs[j & 1];
        const bool last = (j == L);
        a.e.add = (last && !counted) ? d.gego : d.gprop;
        a.e.sum_scale = 1.0f;
        if (last) {
            a.e.zero1 = counted ? nullptr : d.gego;
            a.e.zero2 = (L >= 2) ? d.gprop : nullptr;
            if (counted) { a.e.cnt = d.cnt; a.e.creg = d.coef + 2 * RK_MAX_GRAPH_STEPS + k; a.e.reg_p = d.e0s; }
            if (apply_update) {
                a.e.adam = 1;
                a.e.p = d.e0s; a.e.m = d.ms; a.e.v = d.vs;   // sliced working copies (lds_sync)
                a.e.coef = d.coef + 2 * k;
                a.e.b1 = d.beta1; a.e.b2 = d.beta2; a.e.eps = d.eps;
            }
            a.e.y = d.grad;  // nullable, row-major
            a.e.y_rm = 1;
            a.e.state = d.state;
            a.e.bump = bump;
        } else {
            a.e.y = bufs[(j + 1) & 1];
            if (!lds_fused(d)) a.e.y_staged = 1;
        }
        if (!lds_fused(d) && j > 1) a.e.x_staged = 1;
    }
    if (lds_fused(d)) { int rc = launch_lds_multi(d, li, ph.data(), L, s); if (rc) return rc; }
    else for (int j = 0; j < L; ++j) RK_HIP(spmm_lds_launch(li, ph[(size_t)j], s));
    if (L == 1) {   // gprop is the gather operand of the only backward SpMM: cleaned by a kernel (see launch_backward)
        const long long n4 = (long long)N * d.dim / 4;
        hipLaunchKernelGGL(zero_f4_kernel, dim3((int)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, s,
                           reinterpret_cast<float4 *>(d.gprop), n4, d.gprop, (long long)N * d.dim);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}

// forward: light = mean_l A^l E0 ; uses buf_a/buf_b as ping-pong
struct BatchRef {
    int k;       // step of the chunk (the triplets themselves are found through the state block)
    int batch;   // minibatch size the step was set up for (bounds the marked-block list: <= 3 * batch rows)
};

// forward: light = mean_l A^l E0.  With a BatchRef (training) the first layer marks the minibatch's
// rows in d.row_bits and the last layer computes only those rows of `light`.
// drop: 0 = none, 1 = the train step's mask (batch->k), 3 = the mask of mask_seed
static int launch_forward(const rk_lightgcn_desc &d, hipStream_t s, const BatchRef *batch = nullptr, int drop = 0,
                          unsigned long long mask_seed = 0ULL)
{
    const int L = d.n_layers;
    const float inv = 1.0f / (float)(L + 1);
    if (use_lds(d) && drop == 0) return launch_forward_lds(d, s, batch != nullptr);
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
            }
            if (l == L) a.row_filter = d.row_bits;
            // L >= 3: the launch between marking and filtering lists the workgroups that hold a marked row, and the filtered
            // launch starts only those (spmm.h, SpmmArgs::blk_mode)
            // -- where that at least halves the filtered launch: on a small schedule (ml1m: 2 413 workgroups against 3 072 minibatch
            // rows) nearly every workgroup holds a marked row, and the list's two dependent loads in front of the descriptor plus
            // the appends cost the step 17-24 % (profiles/r05_bench_asis.json before / after)
            const long long cap = 3LL * batch->batch + (long long)std::max(d.row_blocks_extra, 0);
            const long long n_plain = (long long)(d.n_blocks & ~(kSchedPackedFlag | kSchedWavesMask | kSchedLongFlag));
            if (d.row_blocks && L >= 3 && batch->batch > 0 && 2 * cap <= n_plain) {
                a.blk_count = d.row_blocks; a.blk_list = d.row_blocks + 4; a.blk_bits = d.row_bits;
                a.blk_cap = (int)std::min<long long>(0x7fffffff, cap);
                a.blk_mode = l == 1 ? 1 : l == L - 1 ? 2 : l == L ? 3 : 0;
            }
        }
        if (drop == 1) set_dropout(a, d, l == 1 ? 1 : 2, batch ? batch->k : 0, d.drop_seed, false);
        else if (drop == 3) set_dropout(a, d, 3, 0, rk_drop_step_seed(d.drop_seed, mask_seed), false);
        RK_HIP(spmm_launch(a, s));
    }
    return RK_OK;
}

// backward + Adam for chunk step k; gprop/gego hold the BPR scatter
static int launch_backward(const rk_lightgcn_desc &d, int k, int apply_update, int bump, hipStream_t s, bool ordered = false)
{
    const int N = d.n_users + d.n_items, L = d.n_layers;
    if (use_lds(d)) return launch_backward_lds(d, k, apply_update, bump, s, d.cnt != nullptr && !ordered);
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
            // the bitmap of the minibatch's rows (set by the first forward layer) is still up: gprop is zero elsewhere
            if (j == 1) a.src_filter = d.row_bits;
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

static int launch_step(const rk_lightgcn_desc &d, int k, int apply_update, int bump, hipStream_t s, int batch, const OrderedRef &ord = OrderedRef{nullptr, 0})
{
    const BatchRef br{k, batch};
    int rc = launch_forward(d, s, &br, d.keep_prob > 0.f ? 1 : 0);
    if (rc) return rc;
    BprArgs b;
    memset(&b, 0, sizeof(b));
    b.U = d.n_users; b.d = d.dim; b.L = d.n_layers; b.lam = d.lambda;
    b.light = d.light; b.emb = use_lds(d) ? d.e0s : d.user_emb;   // (LDS path: E0 lives in its sliced copy during an epoch)
    b.gprop = d.gprop; b.gego = d.gego;
    b.state = d.state; b.coef = d.coef; b.k = k;
    b.lr = d.lr; b.b1 = d.beta1; b.b2 = d.beta2;
    b.nb_direct = 0; b.light_compact = 0;
    b.keys = ord.keys;
    b.sliced = use_lds(d) ? 1 : 0;
    b.sl = LdsDims{d.n_users, d.n_items, d.dim, d.lds_info.lsu, d.lds_info.lsi};
    if (use_lds(d) && d.cnt && !ord.keys) { b.cnt = d.cnt; b.creg_out = d.coef + 2 * RK_MAX_GRAPH_STEPS + k; }
    if (ord.keys) {
        if (d.dim <= 64) hipLaunchKernelGGL(bpr_rows_kernel<1>, dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
        else if (d.dim <= 128) hipLaunchKernelGGL(bpr_rows_kernel<2>, dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
        else hipLaunchKernelGGL(bpr_rows_kernel<4>, dim3(RK_LOSS_PARTIALS), dim3(kRowsWaves * 64), 0, s, b);
    }
    else hipLaunchKernelGGL(bpr_kernel, dim3(RK_LOSS_PARTIALS), dim3(256), 0, s, b);
    RK_CHECK_LAUNCH();
    return launch_backward(d, k, apply_update, bump, s, ord.keys != nullptr);
}

static OrderedRef ordered_ref(const rk_lightgcn *h, int batch)
{
    return h->deterministic ? OrderedRef{h->plan_sorted, batch} : OrderedRef{nullptr, 0};
}

// what every train_epoch call does before its first step: scatter targets start (and, by the self-cleaning epilogues, stay)
// zero; LDS path: sliced working copies of E0 / m / v.  Kernels, not memset nodes (see common.h rk_zero_async).
static int launch_prologue(const rk_lightgcn_desc &d, int apply_update, hipStream_t s, bool ordered)
{
    const int N = d.n_users + d.n_items;
    if (use_lds(d)) return lds_sync(d, apply_update != 0, 1, s, true, ordered || !d.cnt);   // ONE launch: layout conversions + clears
    RK_HIP(rk_zero_async(d.gprop, sizeof(float) * (size_t)N * d.dim, s));
    RK_HIP(rk_zero_async(d.gego, sizeof(float) * (size_t)N * d.dim, s));
    if (d.row_bits) RK_HIP(rk_zero_async(d.row_bits, sizeof(uint32_t) * (size_t)((N + 31) / 32), s));
    return RK_OK;
}

static int launch_epilogue(const rk_lightgcn_desc &d, int apply_update, hipStream_t s)
{
    if (use_lds(d) && apply_update) return lds_sync(d, true, 0, s);   // back to the caller's row-major tensors
    return RK_OK;
}

// hipGraph of n_steps train steps (whole: a complete epoch call, prologue and epilogue included)
static int ensure_exec(rk_lightgcn *h, int n_steps, int whole, int apply_update, int batch, hipStream_t upload_stream, hipGraphExec_t *out)
{
    const rk_lightgcn_desc &d = h->d;
    const OrderedRef ord = ordered_ref(h, batch);
    int slot = -1;
    for (int i = 0; i < rk_lightgcn::kExecSlots; ++i) {
        rk_lightgcn::Exec &e = h->exec[i];
        if (e.exec && e.n_steps == n_steps && e.whole == whole && e.update == apply_update && e.det == h->deterministic &&
            e.batch == batch &&   // (the batch size is baked into a capture: the marked-block launch's grid, spmm.h blk_cap)
            (!h->deterministic || e.plan == ord.keys)) {
            e.stamp = ++h->clock;
            *out = e.exec;
            return RK_OK;
        }
    }
    for (int i = 0; i < rk_lightgcn::kExecSlots; ++i) {   // a free slot, else the least recently used
        if (!h->exec[i].exec) { slot = i; break; }
        if (slot < 0 || h->exec[i].stamp < h->exec[slot].stamp) slot = i;
    }
    rk_lightgcn::Exec &e = h->exec[slot];
    if (e.exec) { (void)hipGraphExecDestroy(e.exec); e.exec = nullptr; }
    if (!h->cap_stream) RK_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    hipGraph_t g = nullptr;
    RK_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = RK_OK;
    if (whole) rc = launch_prologue(d, apply_update, h->cap_stream, h->deterministic != 0);
    for (int k = 0; k < n_steps && rc == RK_OK; ++k)
        rc = launch_step(d, k, apply_update, k == n_steps - 1 ? n_steps : 0, h->cap_stream, batch, ord);
    if (whole && rc == RK_OK) rc = launch_epilogue(d, apply_update, h->cap_stream);
    hipError_t err = hipStreamEndCapture(h->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    RK_HIP(err);
    RK_HIP(hipGraphInstantiate(&e.exec, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    e.n_steps = n_steps; e.whole = whole; e.update = apply_update; e.det = h->deterministic; e.plan = ord.keys; e.batch = batch;
    e.stamp = ++h->clock;
    if (upload_stream) RK_HIP(hipGraphUpload(e.exec, upload_stream));
    *out = e.exec;
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

// How an epoch of n_steps is cut: n_steps <= RK_MAX_GRAPH_STEPS (and graph replay wanted): ONE whole-call graph;
// otherwise chunks of graph_steps steps and one remainder graph.
static bool whole_call(int n_steps, int graph_steps) { return graph_steps > 1 && n_steps >= 1 && n_steps <= RK_MAX_GRAPH_STEPS; }

RK_EXPORT int rk_lightgcn_prepare(rk_lightgcn_t h, int64_t n, int32_t batch, int32_t apply_update, int32_t graph_steps, void *stream)
{
    if (!h) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: null handle");
    if (n <= 0 || batch <= 0) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: bad arguments");
    if (!apply_update && !h->d.grad) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: apply_update=0 needs desc.grad");
    if (graph_steps > RK_MAX_GRAPH_STEPS) graph_steps = RK_MAX_GRAPH_STEPS;
    if (graph_steps <= 1) return RK_OK;
    if (h->deterministic) RK_FAIL(RK_EINVAL, "rk_lightgcn_prepare: in deterministic mode the graph depends on the epoch's plan; the first rk_lightgcn_train_epoch captures it");
    hipStream_t s = (hipStream_t)stream;
    hipStream_t up = nullptr;
    if (!s) { if (!h->cap_stream) RK_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking)); up = h->cap_stream; } else up = s;
    const int n_steps = (int)std::min<long long>((n + batch - 1) / batch, 1LL << 30);
    hipGraphExec_t ex = nullptr;
    if (whole_call(n_steps, graph_steps)) return ensure_exec(h, n_steps, 1, apply_update, batch, up, &ex);
    int rc = ensure_exec(h, graph_steps, 0, apply_update, batch, up, &ex);
    if (rc) return rc;
    const int rem = n_steps % graph_steps;
    if (rem >= 2) rc = ensure_exec(h, rem, 0, apply_update, batch, up, &ex);
    return rc;
}

RK_EXPORT int rk_lightgcn_propagate(rk_lightgcn_t h, void *stream)
{
    if (!h) RK_FAIL(RK_EINVAL, "rk_lightgcn_propagate: null handle");
    if (use_lds(h->d)) { int rc = lds_sync(h->d, false, 1, (hipStream_t)stream); if (rc) return rc; }
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
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(1), 0, s, d.state, 0, adam_t0, (long long)n, batch, users, pos, neg, loss_partials);
    RK_CHECK_LAUNCH();
    if (h->deterministic) {
        int rc = build_plan(h, users, pos, neg, n, batch, s);
        if (rc) return rc;
    }
    if (graph_steps > RK_MAX_GRAPH_STEPS) graph_steps = RK_MAX_GRAPH_STEPS;
    if (whole_call(n_steps, graph_steps)) {   // the whole call is one replay
        hipGraphExec_t ex = nullptr;
        int rc = ensure_exec(h, n_steps, 1, apply_update, batch, nullptr, &ex);
        if (rc) return rc;
        RK_HIP(hipGraphLaunch(ex, s));
        return RK_OK;
    }
    int rc = launch_prologue(d, apply_update, s, h->deterministic != 0);
    if (rc) return rc;
    int done = 0;
    if (graph_steps > 1) {
        hipGraphExec_t ex = nullptr;
        if (n_steps >= graph_steps) {
            rc = ensure_exec(h, graph_steps, 0, apply_update, batch, nullptr, &ex);
            if (rc) return rc;
            for (; done + graph_steps <= n_steps; done += graph_steps) RK_HIP(hipGraphLaunch(ex, s));
        }
        const int rem = n_steps - done;
        if (rem >= 2) {
            rc = ensure_exec(h, rem, 0, apply_update, batch, nullptr, &ex);
            if (rc) return rc;
            RK_HIP(hipGraphLaunch(ex, s));
            done += rem;
        }
    }
    const OrderedRef ord = ordered_ref(h, batch);
    for (; done < n_steps; ++done) {
        rc = launch_step(d, 0, apply_update, 1, s, batch, ord);
        if (rc) return rc;
    }
    return launch_epilogue(d, apply_update, s);
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
    a.src_filter = epi->src_filter;
    if (epi->sum_out && !epi->sum_in) RK_FAIL(RK_EINVAL, "rk_spmm_csr_ex: sum_out needs sum_in");
    if (epi->adam_t != 0) {
        if (!epi->adam_p || !epi->adam_m || !epi->adam_v || !epi->coef_scratch) RK_FAIL(RK_EINVAL, "rk_spmm_csr_ex: Adam pointers missing");
        if (epi->adam_t > 0) {   // (adam_t < 0: coef_scratch already holds this step's coefficients -- rk_adam_coef_advance)
            const AdamCoef c = adam_coef(epi->adam_t, epi->lr, epi->beta1, epi->beta2);
            hipLaunchKernelGGL(set_coef_kernel, dim3(1), dim3(1), 0, s, epi->coef_scratch, c.step_size, c.bc2s);
            RK_CHECK_LAUNCH();
        }
        a.e.adam = 1; a.e.p = epi->adam_p; a.e.m = epi->adam_m; a.e.v = epi->adam_v; a.e.coef = epi->coef_scratch;
        a.e.b1 = epi->beta1; a.e.b2 = epi->beta2; a.e.eps = epi->eps;
    }
    RK_HIP(spmm_launch(a, s));
    return RK_OK;
}

// Adam coefficients of the NEXT step from a device-resident step counter (a captured step graph cannot take the step number
// from the host): counter += 1; coef = {lr / (1 - b1^t), sqrt(1 - b2^t)}.
__global__ void adam_coef_advance_kernel(float *coef, int *counter, float lr, float b1, float b2)
{
    const int t = counter[0] + 1;
    counter[0] = t;
    const AdamCoef c = adam_coef(t, lr, b1, b2);
    coef[0] = c.step_size;
    coef[1] = c.bc2s;
}

RK_EXPORT int rk_adam_coef_advance(float *coef, int32_t *counter, float lr, float beta1, float beta2, void *stream)
{
    if (!coef || !counter) RK_FAIL(RK_EINVAL, "rk_adam_coef_advance: bad arguments");
    hipLaunchKernelGGL(adam_coef_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, coef, counter, lr, beta1, beta2);
    RK_CHECK_LAUNCH();
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
