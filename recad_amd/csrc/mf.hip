// MF victim hot path (recad/model/victim/mf.py:40-69): per step one fused
// gather + logit + BCE-with-logits + scatter-add kernel and one dense Adam pass over the
// four tables that also re-zeroes the gradient buffer.
#include <algorithm>

#include "common.h"

struct MfArgs {
    int U, I, d;
    const float *ue, *ie, *ub, *ib;
    float mean;
    float *grads;  // [U*d | I*d | U | I]
    const int64_t *users, *items, *labels;
    long long off;
    int nb;
    float *loss_partials;  // this step's RK_LOSS_PARTIALS slots
    unsigned drop_thresh24;  // > 0: nn.Dropout on the logit (mf.py:27,47), one counter-hash mask per (step, sample)
    float drop_scale;
    unsigned long long drop_seed;
};

// One wave per sample; lanes stride over the embedding (mf.py:40-47), wave-shuffle dot,
// BCEWithLogits (mf.py:32,59-60) and its gradient scattered with no-return float atomics.
__global__ __launch_bounds__(256) void mf_step_kernel(const MfArgs a)
{
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int d = a.d;
    const float invB = 1.0f / (float)a.nb;
    float *gue = a.grads, *gie = gue + (size_t)a.U * d, *gub = gie + (size_t)a.I * d, *gib = gub + a.U;
    float lsum = 0.f;
    for (int b = blockIdx.x * 4 + w; b < a.nb; b += gridDim.x * 4) {
        const long long u = a.users[a.off + b], i = a.items[a.off + b];
        const float y = (float)a.labels[a.off + b];
        const float *pu = a.ue + (size_t)u * d, *pi = a.ie + (size_t)i * d;
        float s = 0.f;
        for (int k = lane; k < d; k += 64) s += pu[k] * pi[k];
        s = wave_sum(s);
        float x = ((s + a.ub[u]) + a.ib[i]) + a.mean;
        float keep = 1.f;
        if (a.drop_thresh24) {
            keep = rk_drop_keep(a.drop_seed, (unsigned)b, a.drop_thresh24) ? a.drop_scale : 0.f;
            x *= keep;
        }
        lsum += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        const float dx = (1.f / (1.f + expf(-x)) - y) * invB * keep;
        for (int k = lane; k < d; k += 64) {
            unsafeAtomicAdd(gue + (size_t)u * d + k, dx * pi[k]);
            unsafeAtomicAdd(gie + (size_t)i * d + k, dx * pu[k]);
        }
        if (lane == 0) {
            unsafeAtomicAdd(gub + u, dx);
            unsafeAtomicAdd(gib + i, dx);
        }
    }
    if (lane == 0) red[w] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) a.loss_partials[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * invB;
}

struct MfAdamArgs {
    long long n_ue, n_ie, n_ub, n_ib;
    float *ue, *ie, *ub, *ib;
    float *m, *v, *grads;
    float step_size, bc2s, b1, b2, eps;
    int apply;
};

__global__ void mf_adam_kernel(const MfAdamArgs a)
{
    const long long tot = a.n_ue + a.n_ie + a.n_ub + a.n_ib;
    const float w1 = (float)(1.0 - (double)a.b1), w2 = (float)(1.0 - (double)a.b2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        float *p;
        long long j = i;
        if (j < a.n_ue) p = a.ue + j;
        else if ((j -= a.n_ue) < a.n_ie) p = a.ie + j;
        else if ((j -= a.n_ie) < a.n_ub) p = a.ub + j;
        else p = a.ib + (j - a.n_ub);
        const float g = a.grads[i];
        a.grads[i] = 0.f;
        float pp = *p, mm = a.m[i], vv = a.v[i];
        adam_elem(pp, mm, vv, g, w1, a.b2, w2, a.step_size, a.bc2s, a.eps);
        *p = pp; a.m[i] = mm; a.v[i] = vv;
    }
}

RK_EXPORT int rk_mf_train_epoch(int32_t n_users, int32_t n_items, int32_t dim, float *user_emb, float *item_emb,
                                float *user_bias, float *item_bias, float mean, float *m, float *v, float *grads,
                                const int64_t *users, const int64_t *items, const int64_t *labels, int64_t n,
                                int32_t batch, int32_t adam_t0, float lr, float beta1, float beta2, float eps,
                                float *loss_partials, int32_t apply_update, float dropout, uint64_t drop_seed, void *stream)
{
    if (n_users <= 0 || n_items <= 0 || dim <= 0 || n <= 0 || batch <= 0) RK_FAIL(RK_EINVAL, "rk_mf_train_epoch: bad sizes");
    if (!(dropout >= 0.f) || dropout >= 1.f) RK_FAIL(RK_EINVAL, "rk_mf_train_epoch: dropout must be in [0, 1)");
    if (!user_emb || !item_emb || !user_bias || !item_bias || !m || !v || !grads || !users || !items || !labels || !loss_partials)
        RK_FAIL(RK_EINVAL, "rk_mf_train_epoch: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const long long tot = (long long)n_users * dim + (long long)n_items * dim + n_users + n_items;
    RK_HIP(hipMemsetAsync(grads, 0, sizeof(float) * (size_t)tot, s));
    const int n_steps = (int)((n + batch - 1) / batch);
    RK_HIP(hipMemsetAsync(loss_partials, 0, sizeof(float) * (size_t)n_steps * RK_LOSS_PARTIALS, s));
    for (int step = 0; step < n_steps; ++step) {
        MfArgs a;
        a.U = n_users; a.I = n_items; a.d = dim;
        a.ue = user_emb; a.ie = item_emb; a.ub = user_bias; a.ib = item_bias; a.mean = mean;
        a.grads = grads;
        a.users = users; a.items = items; a.labels = labels;
        a.off = (long long)step * batch;
        a.nb = (int)std::min<long long>(batch, n - a.off);
        a.loss_partials = loss_partials + (size_t)step * RK_LOSS_PARTIALS;
        a.drop_thresh24 = dropout > 0.f ? (unsigned)((1.0 - (double)dropout) * 16777216.0) : 0u;
        a.drop_scale = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.f;
        a.drop_seed = rk_drop_step_seed((unsigned long long)drop_seed, (unsigned long long)(adam_t0 + step));
        const int grid = std::min(RK_LOSS_PARTIALS, (a.nb + 3) / 4);
        hipLaunchKernelGGL(mf_step_kernel, dim3(grid), dim3(256), 0, s, a);
        RK_CHECK_LAUNCH();
        if (!apply_update) break;  // testing: leave the step-1 gradients in `grads`
        const AdamCoef c = adam_coef(adam_t0 + step + 1, lr, beta1, beta2);
        MfAdamArgs b;
        b.n_ue = (long long)n_users * dim; b.n_ie = (long long)n_items * dim; b.n_ub = n_users; b.n_ib = n_items;
        b.ue = user_emb; b.ie = item_emb; b.ub = user_bias; b.ib = item_bias;
        b.m = m; b.v = v; b.grads = grads;
        b.step_size = c.step_size; b.bc2s = c.bc2s; b.b1 = beta1; b.b2 = beta2; b.eps = eps;
        b.apply = 1;
        const int agrid = (int)std::min<long long>((tot + 255) / 256, 2048);
        hipLaunchKernelGGL(mf_adam_kernel, dim3(agrid), dim3(256), 0, s, b);
        RK_CHECK_LAUNCH();
    }
    return RK_OK;
}
