/*
 * recad_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C, single-thread, fp32 restatement of the arithmetic on RecAD's victim-model
 * hot path, written from the reference's behaviour (gusye1234/recad v0.0.2).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (recad_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
 * against golden vectors produced by importing the reference itself in the build
 * container (tests/golden/make_golden.py; datasets dev and game, seed 2023, 1 thread).
 * The path's arithmetic lives in ATen (PyTorch 2.10.0, not vendored in the reference);
 * summation order is therefore not bit-identical to ATen, tolerances are stated in
 * the tests (<=1e-5 rel on losses, <=1e-4 rel on tables).
 *
 * Each function cites the reference lines it restates (paths under /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * Graph: coalesced COO (row-major sorted, as torch .coalesce() returns it,
 * recad/dataset/implicit.py:295-296,320-326) -> CSR with int32 columns.
 * ---------------------------------------------------------------------------------- */
API void orc_coo_to_csr(int32_t n, int64_t nnz, const int32_t *row, int32_t *rowptr)
{
    memset(rowptr, 0, sizeof(int32_t) * (size_t)(n + 1));
    for (int64_t e = 0; e < nnz; ++e) rowptr[row[e] + 1]++;
    for (int32_t r = 0; r < n; ++r) rowptr[r + 1] += rowptr[r];
}

/* D^-1/2 A D^-1/2 of the bipartite graph, recad/dataset/implicit.py:259-277:
 * A = [[0,R],[R^T,0]], rowsum + 1e-14, power -0.5, inf -> 0, values cast to fp32
 * (implicit.py:321).  R is given as CSR (user -> sorted unique item ids).
 * Output: CSR over N=U+I nodes (rowptr[N+1], col[2*E], val[2*E]), columns sorted. */
API void orc_build_norm_adj(int32_t U, int32_t I, const int32_t *rptr, const int32_t *ridx,
                            int32_t *rowptr, int32_t *col, float *val)
{
    int32_t N = U + I;
    int64_t E = rptr[U];
    double *deg = (double *)calloc((size_t)N, sizeof(double));
    int32_t *icnt = (int32_t *)calloc((size_t)I + 1, sizeof(int32_t));
    for (int32_t u = 0; u < U; ++u) {
        deg[u] = (double)(rptr[u + 1] - rptr[u]);
        for (int32_t e = rptr[u]; e < rptr[u + 1]; ++e) { deg[U + ridx[e]] += 1.0; icnt[ridx[e] + 1]++; }
    }
    /* All of it happens in fp32 under numpy>=2 promotion rules: adj_mat is a float32
     * dok, so rowsum is float32, `rowsum + 1e-14` stays float32 (python scalars are
     * weak), np.power(...,-0.5) is the float32 loop, sp.diags(d_inv) is float32 and
     * both .dot() products are float32.  numpy's SIMD float32 pow is not correctly
     * rounded, so values agree with the reference's to a few ulp (measured 2.8e-7 rel), not bit-for-bit
     * (tests/test_oracle_golden.py states rtol 4e-7). */
    float *dinv = (float *)malloc(sizeof(float) * (size_t)N);
    for (int32_t r = 0; r < N; ++r) {
        float v = powf((float)deg[r] + 1e-14f, -0.5f);
        dinv[r] = isinf(v) ? 0.0f : v;
    }
    rowptr[0] = 0;
    for (int32_t u = 0; u < U; ++u) rowptr[u + 1] = rowptr[u] + (rptr[u + 1] - rptr[u]);
    for (int32_t i = 0; i < I; ++i) rowptr[U + i + 1] = rowptr[U + i] + icnt[i + 1];
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)I);
    for (int32_t i = 0; i < I; ++i) fill[i] = rowptr[U + i];
    for (int32_t u = 0; u < U; ++u)
        for (int32_t e = rptr[u]; e < rptr[u + 1]; ++e) {
            int32_t i = ridx[e];
            col[e] = U + i;
            val[e] = (dinv[u] * 1.0f) * dinv[U + i];
            int32_t p = fill[i]++;
            col[p] = u; /* users visited in increasing order -> item rows sorted */
            val[p] = (dinv[U + i] * 1.0f) * dinv[u];
        }
    (void)E;
    free(deg); free(icnt); free(dinv); free(fill);
}

/* Y = A X, torch.sparse.mm(g, all_emb) at recad/model/victim/lightgcn.py:107 */
API void orc_spmm(int32_t n, const int32_t *rowptr, const int32_t *col, const float *val,
                  int32_t d, const float *X, float *Y)
{
    for (int32_t r = 0; r < n; ++r) {
        float *y = Y + (size_t)r * d;
        for (int32_t k = 0; k < d; ++k) y[k] = 0.f;
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float *x = X + (size_t)col[e] * d;
            float a = val[e];
            for (int32_t k = 0; k < d; ++k) y[k] += a * x[k];
        }
    }
}

/* LightGCN.computer(), recad/model/victim/lightgcn.py:82-113:
 * E0=[U;I]; E(l+1)=A E(l); light = mean(E0..EL).  layers (optional, may be NULL)
 * receives E0..EL back to back for the backward pass. */
API void orc_lightgcn_propagate(int32_t U, int32_t I, int32_t d, int32_t L,
                                const int32_t *rowptr, const int32_t *col, const float *val,
                                const float *user, const float *item, float *light)
{
    int32_t N = U + I;
    size_t nd = (size_t)N * d;
    float *cur = (float *)malloc(sizeof(float) * nd), *nxt = (float *)malloc(sizeof(float) * nd);
    memcpy(cur, user, sizeof(float) * (size_t)U * d);
    memcpy(cur + (size_t)U * d, item, sizeof(float) * (size_t)I * d);
    memcpy(light, cur, sizeof(float) * nd);
    for (int32_t l = 0; l < L; ++l) {
        orc_spmm(N, rowptr, col, val, d, cur, nxt);
        for (size_t k = 0; k < nd; ++k) light[k] += nxt[k];
        float *t = cur; cur = nxt; nxt = t;
    }
    float inv = 1.0f / (float)(L + 1);
    for (size_t k = 0; k < nd; ++k) light[k] *= inv;
    free(cur); free(nxt);
}

static float softplus_f(float x) /* torch softplus, beta=1, threshold=20 */
{
    return x > 20.f ? x : log1pf(expf(x));
}
static float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

/* torch.optim.Adam defaults (betas .9/.999, eps 1e-8, wd 0, amsgrad off), one dense
 * tensor, recad/utils.py:181-183 + recad/model/victim/lightgcn.py:17-19,168.  t is
 * the 1-based step count. */
API void orc_adam(int64_t n, float *p, const float *g, float *m, float *v, int32_t t,
                  float lr, float b1, float b2, float eps)
{
    double bc1 = 1.0 - pow((double)b1, (double)t);
    double bc2 = 1.0 - pow((double)b2, (double)t);
    float step_size = (float)((double)lr / bc1);
    float bc2s = (float)sqrt(bc2);
    float w1 = (float)(1.0 - (double)b1), w2 = (float)(1.0 - (double)b2);
    for (int64_t k = 0; k < n; ++k) {
        m[k] = m[k] + w1 * (g[k] - m[k]);          /* exp_avg.lerp_(grad, 1-beta1) */
        v[k] = v[k] * b2 + w2 * g[k] * g[k];       /* mul_(beta2).addcmul_(g,g,1-beta2) */
        float denom = sqrtf(v[k]) / bc2s + eps;
        p[k] = p[k] - step_size * (m[k] / denom);  /* addcdiv_(exp_avg, denom, -step_size) */
    }
}

/* One minibatch of LightGCN.train_step, recad/model/victim/lightgcn.py:137-169:
 * forward propagate, 6 gathers (:122-130), reg (:149-157), BPR softplus (:158-163),
 * loss = mean softplus + lambda*reg (:165), backward, dense Adam on both tables (:168).
 * grad_user/grad_item (optional) receive dLoss/dE0.  Returns the step loss. */
/* General form: the forward graph G (rowptr/col/val) and its transpose (rowptr_t/col_t/val_t) are given
 * separately.  With graph dropout (lightgcn.py:62-80,91-95: every stored entry of the normalised adjacency is
 * kept with probability keep_prob and divided by it, a fresh draw per training forward) the propagated
 * graph is no longer symmetric, and autograd of torch.sparse.mm applies G^T in the backward. */
API float orc_lightgcn_step_general(int32_t U, int32_t I, int32_t d, int32_t L,
                            const int32_t *rowptr, const int32_t *col, const float *val,
                            const int32_t *rowptr_t, const int32_t *col_t, const float *val_t,
                            float *user, float *item, float *m_user, float *v_user, float *m_item, float *v_item,
                            int32_t t, const int64_t *users, const int64_t *pos, const int64_t *neg, int32_t B,
                            float lam, float lr, float b1, float b2, float eps,
                            float *grad_user, float *grad_item, int32_t apply_update);

API float orc_lightgcn_step(int32_t U, int32_t I, int32_t d, int32_t L,
                            const int32_t *rowptr, const int32_t *col, const float *val,
                            float *user, float *item, float *m_user, float *v_user, float *m_item, float *v_item,
                            int32_t t, const int64_t *users, const int64_t *pos, const int64_t *neg, int32_t B,
                            float lam, float lr, float b1, float b2, float eps,
                            float *grad_user, float *grad_item, int32_t apply_update)
{
    /* D^-1/2 A D^-1/2 is symmetric: G^T = G */
    return orc_lightgcn_step_general(U, I, d, L, rowptr, col, val, rowptr, col, val, user, item, m_user, v_user, m_item, v_item,
                                     t, users, pos, neg, B, lam, lr, b1, b2, eps, grad_user, grad_item, apply_update);
}

API float orc_lightgcn_step_general(int32_t U, int32_t I, int32_t d, int32_t L,
                            const int32_t *rowptr, const int32_t *col, const float *val,
                            const int32_t *rowptr_t, const int32_t *col_t, const float *val_t,
                            float *user, float *item, float *m_user, float *v_user, float *m_item, float *v_item,
                            int32_t t, const int64_t *users, const int64_t *pos, const int64_t *neg, int32_t B,
                            float lam, float lr, float b1, float b2, float eps,
                            float *grad_user, float *grad_item, int32_t apply_update)
{
    int32_t N = U + I;
    size_t nd = (size_t)N * d;
    float *light = (float *)malloc(sizeof(float) * nd);
    orc_lightgcn_propagate(U, I, d, L, rowptr, col, val, user, item, light);
    float *g = (float *)calloc(nd, sizeof(float));   /* dLoss/dlight */
    float *ge = (float *)calloc(nd, sizeof(float));  /* dLoss/dE0 through the reg term */
    double loss = 0.0, reg = 0.0;
    float invB = 1.0f / (float)B;
    for (int32_t b = 0; b < B; ++b) {
        size_t u = (size_t)users[b], p = (size_t)U + (size_t)pos[b], n = (size_t)U + (size_t)neg[b];
        const float *lu = light + u * d, *lp = light + p * d, *ln = light + n * d;
        const float *eu = user + u * d, *ep = item + (size_t)pos[b] * d, *en = item + (size_t)neg[b] * d;
        float ps = 0.f, ns = 0.f, r = 0.f;
        for (int32_t k = 0; k < d; ++k) {
            ps += lu[k] * lp[k];
            ns += lu[k] * ln[k];
            r += eu[k] * eu[k] + ep[k] * ep[k] + en[k] * en[k];
        }
        float x = ns - ps;
        loss += (double)softplus_f(x);
        reg += (double)r;
        float dx = (x > 20.f ? 1.f : sigmoid_f(x)) * invB;
        float *gu = g + u * d, *gp = g + p * d, *gn = g + n * d;
        float *geu = ge + u * d, *gep = ge + p * d, *gen = ge + n * d;
        float c = lam * invB;
        for (int32_t k = 0; k < d; ++k) {
            gu[k] += dx * (ln[k] - lp[k]);
            gp[k] -= dx * lu[k];
            gn[k] += dx * lu[k];
            geu[k] += c * eu[k];
            gep[k] += c * ep[k];
            gen[k] += c * en[k];
        }
    }
    float floss = (float)(loss / B) + lam * (float)(0.5 * reg / B);
    /* backward of mean-of-layers + L propagations: t = g; t = g + G^T t (L times) */
    float *tcur = (float *)malloc(sizeof(float) * nd), *tnxt = (float *)malloc(sizeof(float) * nd);
    memcpy(tcur, g, sizeof(float) * nd);
    for (int32_t l = 0; l < L; ++l) {
        orc_spmm(N, rowptr_t, col_t, val_t, d, tcur, tnxt);
        for (size_t k = 0; k < nd; ++k) tnxt[k] += g[k];
        float *s = tcur; tcur = tnxt; tnxt = s;
    }
    float inv = 1.0f / (float)(L + 1);
    for (size_t k = 0; k < nd; ++k) tcur[k] = tcur[k] * inv + ge[k];
    if (grad_user) memcpy(grad_user, tcur, sizeof(float) * (size_t)U * d);
    if (grad_item) memcpy(grad_item, tcur + (size_t)U * d, sizeof(float) * (size_t)I * d);
    if (apply_update) {
        orc_adam((int64_t)U * d, user, tcur, m_user, v_user, t, lr, b1, b2, eps);
        orc_adam((int64_t)I * d, item, tcur + (size_t)U * d, m_item, v_item, t, lr, b1, b2, eps);
    }
    free(light); free(g); free(ge); free(tcur); free(tnxt);
    return floss;
}

/* Pairwise scores, LightGCN.forward (lightgcn.py:174-183) given the propagated
 * tables, and MF.forward (mf.py:40-47) when bias pointers are non-NULL. */
API void orc_pair_scores(int32_t d, const float *utab, const float *itab, const float *ubias, const float *ibias,
                         float mean, const int64_t *users, const int64_t *items, int64_t n, float *out)
{
    for (int64_t b = 0; b < n; ++b) {
        const float *u = utab + (size_t)users[b] * d, *i = itab + (size_t)items[b] * d;
        float s = 0.f;
        for (int32_t k = 0; k < d; ++k) s += u[k] * i[k];
        if (ubias) s = ((s + ubias[users[b]]) + ibias[items[b]]) + mean;
        out[b] = s;
    }
}

static float bce_logits(float x, float y) /* nn.BCEWithLogitsLoss element, mf.py:32, ncf.py:58 */
{
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

/* One minibatch of MF.train_step, recad/model/victim/mf.py:49-69 (Dropout p=0). */
API float orc_mf_step(int32_t U, int32_t I, int32_t d, float *ue, float *ie, float *ub, float *ib, float mean,
                      float *mom /* m,v for ue,ie,ub,ib back to back: 2*(U*d+I*d+U+I) */,
                      int32_t t, const int64_t *users, const int64_t *items, const int64_t *labels, int32_t B,
                      float lr, float b1, float b2, float eps, float *grads /* optional, same layout as params */,
                      int32_t apply_update)
{
    size_t nue = (size_t)U * d, nie = (size_t)I * d, tot = nue + nie + (size_t)U + (size_t)I;
    float *g = (float *)calloc(tot, sizeof(float));
    float *gue = g, *gie = g + nue, *gub = gie + nie, *gib = gub + U;
    double loss = 0.0;
    float invB = 1.f / (float)B;
    for (int32_t b = 0; b < B; ++b) {
        size_t u = (size_t)users[b], i = (size_t)items[b];
        const float *pu = ue + u * d, *pi = ie + i * d;
        float s = 0.f;
        for (int32_t k = 0; k < d; ++k) s += pu[k] * pi[k];
        float x = ((s + ub[u]) + ib[i]) + mean, y = (float)labels[b];
        loss += (double)bce_logits(x, y);
        float dx = (sigmoid_f(x) - y) * invB;
        for (int32_t k = 0; k < d; ++k) { gue[u * d + k] += dx * pi[k]; gie[i * d + k] += dx * pu[k]; }
        gub[u] += dx;
        gib[i] += dx;
    }
    if (grads) memcpy(grads, g, sizeof(float) * tot);
    if (apply_update) {
        float *m = mom, *v = mom + tot;
        orc_adam((int64_t)nue, ue, gue, m, v, t, lr, b1, b2, eps);
        orc_adam((int64_t)nie, ie, gie, m + nue, v + nue, t, lr, b1, b2, eps);
        orc_adam((int64_t)U, ub, gub, m + nue + nie, v + nue + nie, t, lr, b1, b2, eps);
        orc_adam((int64_t)I, ib, gib, m + nue + nie + U, v + nue + nie + U, t, lr, b1, b2, eps);
    }
    free(g);
    return (float)(loss / B);
}

/* NCF (NeuMF-end) forward for n pairs, recad/model/victim/ncf.py:112-131.
 * Tower layer l: Linear(in_l -> in_l/2) + ReLU, in_0 = 2E, E = f*2^(L-1) (ncf.py:41-47).
 * W[l] is [out,in] row-major (nn.Linear), pw is [2f] (gmf part first, ncf.py:129), pb scalar.
 * acts (optional) receives per pair: x0[2E], then each layer's post-ReLU output. */
/* blocked != 0: the TRAINING forward of the HIP path (csrc/ncf.hip, ncf_forward_chunk(train)): a tower layer whose input width
 * is a multiple of 256 sums its products in 8 consecutive k-blocks (each a k-ordered chain) combined pairwise,
 * ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)) -- the error of ATen's blocked sgemm (2e-6 of the layer's rms at K = 2048 instead of
 * 9e-6 for one chain), so that four times fewer numerically-zero pre-activations land on the other side of their ReLU gate
 * than fp64 / the reference puts them (one flipped gate changes every lower dW by a rank-1 term of weight 1/B).  Narrower
 * layers and the evaluation forward keep the single chain. */
static void ncf_forward_one_ex(int32_t f, int32_t L, const float *ug, const float *ig, const float *um, const float *im,
                               const float *const *W, const float *const *bias, const float *pw, float pb,
                               float *acts, float *out, int blocked)
{
    int32_t E = f << (L - 1);
    float *x = acts;
    memcpy(x, um, sizeof(float) * (size_t)E);
    memcpy(x + E, im, sizeof(float) * (size_t)E);
    int32_t in = 2 * E;
    for (int32_t l = 0; l < L; ++l) {
        int32_t o = in / 2;
        float *y = x + in;
        for (int32_t r = 0; r < o; ++r) {
            const float *w = W[l] + (size_t)r * in;
            float s = 0.f;
            if (blocked && in % 256 == 0) {
                float p[8];
                const int32_t blk = in / 8;
                for (int32_t q = 0; q < 8; ++q) {
                    float t = 0.f;
                    for (int32_t k = q * blk; k < (q + 1) * blk; ++k) t += w[k] * x[k];
                    p[q] = t;
                }
                s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            } else {
                for (int32_t k = 0; k < in; ++k) s += w[k] * x[k];
            }
            s += bias[l][r];
            y[r] = s > 0.f ? s : 0.f;
        }
        x = y;
        in = o;
    }
    float s = 0.f;
    for (int32_t k = 0; k < f; ++k) s += pw[k] * (ug[k] * ig[k]);
    for (int32_t k = 0; k < f; ++k) s += pw[f + k] * x[k];
    *out = s + pb;
}

static void ncf_forward_one(int32_t f, int32_t L, const float *ug, const float *ig, const float *um, const float *im,
                            const float *const *W, const float *const *bias, const float *pw, float pb,
                            float *acts, float *out)
{
    ncf_forward_one_ex(f, L, ug, ig, um, im, W, bias, pw, pb, acts, out, 0);
}

static size_t ncf_act_floats(int32_t f, int32_t L)
{
    int32_t E = f << (L - 1);
    size_t n = 0;
    int32_t in = 2 * E;
    for (int32_t l = 0; l <= L; ++l) { n += (size_t)in; in /= 2; }
    return n;
}

API void orc_ncf_forward(int32_t f, int32_t L, const float *ugt, const float *igt, const float *umt, const float *imt,
                         const float *const *W, const float *const *bias, const float *pw, float pb,
                         const int64_t *users, const int64_t *items, int64_t n, float *out)
{
    int32_t E = f << (L - 1);
    float *acts = (float *)malloc(sizeof(float) * ncf_act_floats(f, L));
    for (int64_t b = 0; b < n; ++b)
        ncf_forward_one(f, L, ugt + (size_t)users[b] * f, igt + (size_t)items[b] * f, umt + (size_t)users[b] * E,
                        imt + (size_t)items[b] * E, W, bias, pw, pb, acts, out + b);
    free(acts);
}

/* One minibatch of NCF.train_step, recad/model/victim/ncf.py:133-153: gradients of
 * mean BCE-with-logits wrt every parameter.  Gradients are RETURNED (the caller
 * applies orc_adam per tensor): g_ug[U*f], g_ig[I*f], g_um[U*E], g_im[I*E], gW[l], gb[l], gpw[2f], gpb[1]. */
API float orc_ncf_grads(int32_t U, int32_t I, int32_t f, int32_t L, const float *ugt, const float *igt,
                        const float *umt, const float *imt, const float *const *W, const float *const *bias,
                        const float *pw, float pb, const int64_t *users, const int64_t *items, const int64_t *labels,
                        int32_t B, float *g_ug, float *g_ig, float *g_um, float *g_im, float *const *gW,
                        float *const *gb, float *gpw, float *gpb)
{
    int32_t E = f << (L - 1);
    size_t na = ncf_act_floats(f, L);
    float *acts = (float *)malloc(sizeof(float) * na);
    float *dx = (float *)malloc(sizeof(float) * (size_t)2 * E), *dy = (float *)malloc(sizeof(float) * (size_t)2 * E);
    memset(g_ug, 0, sizeof(float) * (size_t)U * f); memset(g_ig, 0, sizeof(float) * (size_t)I * f);
    memset(g_um, 0, sizeof(float) * (size_t)U * E); memset(g_im, 0, sizeof(float) * (size_t)I * E);
    { int32_t in = 2 * E; for (int32_t l = 0; l < L; ++l) { memset(gW[l], 0, sizeof(float) * (size_t)in * (in / 2)); memset(gb[l], 0, sizeof(float) * (size_t)(in / 2)); in /= 2; } }
    memset(gpw, 0, sizeof(float) * (size_t)2 * f); *gpb = 0.f;
    double loss = 0.0;
    float invB = 1.f / (float)B;
    for (int32_t b = 0; b < B; ++b) {
        size_t u = (size_t)users[b], i = (size_t)items[b];
        const float *ug = ugt + u * f, *ig = igt + i * f;
        float x;
        ncf_forward_one_ex(f, L, ug, ig, umt + u * E, imt + i * E, W, bias, pw, pb, acts, &x, 1);
        float y = (float)labels[b];
        loss += (double)bce_logits(x, y);
        float d0 = (sigmoid_f(x) - y) * invB;
        /* predict layer */
        const float *xl = acts + na - f; /* last tower output */
        for (int32_t k = 0; k < f; ++k) {
            gpw[k] += d0 * (ug[k] * ig[k]);
            gpw[f + k] += d0 * xl[k];
            g_ug[u * f + k] += d0 * pw[k] * ig[k];
            g_ig[i * f + k] += d0 * pw[k] * ug[k];
            dy[k] = d0 * pw[f + k];
        }
        *gpb += d0;
        /* tower backward */
        size_t off = na - f; /* start of layer L output */
        int32_t o = f;
        for (int32_t l = L - 1; l >= 0; --l) {
            int32_t in = o * 2;
            const float *yout = acts + off;
            const float *xin = acts + off - in;
            for (int32_t k = 0; k < in; ++k) dx[k] = 0.f;
            for (int32_t r = 0; r < o; ++r) {
                float gr = yout[r] > 0.f ? dy[r] : 0.f;
                if (gr == 0.f) continue;
                const float *w = W[l] + (size_t)r * in;
                float *gw = gW[l] + (size_t)r * in;
                for (int32_t k = 0; k < in; ++k) { gw[k] += gr * xin[k]; dx[k] += gr * w[k]; }
                gb[l][r] += gr;
            }
            float *s = dx; dx = dy; dy = s;
            off -= in;
            o = in;
        }
        for (int32_t k = 0; k < E; ++k) { g_um[u * E + k] += dy[k]; g_im[i * E + k] += dy[E + k]; }
    }
    free(acts); free(dx); free(dy);
    return (float)(loss / B);
}

/* Full-catalog scoring + top-K for ONE user, the semantic content of
 * Normal.user_item_model_generate, recad/workflow/normal.py:57-93: candidates are all
 * items NOT in the user's train list (normal.py:133-143), sorted by score descending
 * (normal.py:86-88; tie-break defined here as lower item id first -- the reference's
 * is unspecified).  scores[I] are the user's scores for every item.
 * Outputs top_ids/top_scores[K] (padded with -1/-inf) and, for each target, its score
 * and rank among the candidates (hit@k <=> rank < k). */
API void orc_topk_row(int32_t I, const float *scores, const int32_t *seen, int32_t n_seen, int32_t K,
                      int32_t *top_ids, float *top_scores, const int32_t *targets, int32_t n_targets,
                      float *target_score, int32_t *target_rank)
{
    uint8_t *mask = (uint8_t *)calloc((size_t)I, 1);
    for (int32_t k = 0; k < n_seen; ++k) mask[seen[k]] = 1;
    for (int32_t k = 0; k < K; ++k) { top_ids[k] = -1; top_scores[k] = -INFINITY; }
    int32_t cnt = 0;
    for (int32_t i = 0; i < I; ++i) {
        if (mask[i]) continue;
        float s = scores[i];
        /* insertion into the sorted top-K; strict > keeps lower ids first among ties */
        if (cnt < K || s > top_scores[K - 1]) {
            int32_t p = cnt < K ? cnt : K - 1;
            while (p > 0 && s > top_scores[p - 1]) { top_scores[p] = top_scores[p - 1]; top_ids[p] = top_ids[p - 1]; --p; }
            top_scores[p] = s; top_ids[p] = i;
            if (cnt < K) ++cnt;
        }
    }
    for (int32_t t = 0; t < n_targets; ++t) {
        int32_t tg = targets[t];
        float st = scores[tg];
        int32_t rank = 0;
        for (int32_t i = 0; i < I; ++i) {
            if (mask[i] || i == tg) continue;
            if (scores[i] > st || (scores[i] == st && i < tg)) ++rank;
        }
        target_score[t] = st;
        target_rank[t] = rank;
    }
    free(mask);
}

/* scores[b, :] = U_b . Items^T (+ MF biases): the batched form of forward() that the
 * HIP scoring GEMM replaces; k-ordered fp32 fma chain per pair like v_mfma_f32. */
API void orc_score_rows(int32_t d, const float *urows, int32_t nb, const float *itab, int32_t I,
                        const float *ubias_rows, const float *ibias, float mean, float *out)
{
    for (int32_t b = 0; b < nb; ++b)
        for (int32_t i = 0; i < I; ++i) {
            const float *u = urows + (size_t)b * d, *v = itab + (size_t)i * d;
            float s = 0.f;
            for (int32_t k = 0; k < d; ++k) s = fmaf(u[k], v[k], s);
            if (ibias) s = ((s + ubias_rows[b]) + ibias[i]) + mean;
            out[(size_t)b * I + i] = s;
        }
}
