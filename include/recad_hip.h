/*
 * recad_hip.h -- C ABI of librecad_hip.so, the MI355X (gfx950) implementation of RecAD's
 * victim-model hot path.  Plain pointers and sizes only; every pointer marked "device"
 * is a HIP device pointer owned by the caller (e.g. torch.Tensor.data_ptr()); `stream`
 * is a hipStream_t passed as void*.  Nothing here allocates caller-visible memory.
 *
 * The reference (gusye1234/recad v0.0.2) is pure Python on ATen: it has NO FFI for this
 * path.  Its boundary is the duck-typed victim class; each entry point
 * below names the reference code it replaces (paths under /root/reference), and
 * INTEGRATION.md shows the ctypes binding a maintainer would add to the reference.
 *
 * Return value: 0 on success, negative on error (RK_E*); rk_last_error() gives the text.
 * Thread-safety: one handle per host thread; entry points are asynchronous on `stream`
 * unless stated otherwise.
 */
#ifndef RECAD_HIP_H
#define RECAD_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RK_ABI_VERSION 9
#define RK_OK 0
#define RK_EINVAL (-22)   /* bad argument / unsupported shape */
#define RK_EHIP (-5)      /* a HIP runtime call failed */
#define RK_ENOMEM (-12)

/* number of float partials rk_*_train_epoch writes per step */
#define RK_LOSS_PARTIALS 256

int rk_abi_version(void);
const char *rk_last_error(void);
/* Device sanity: returns 0 and fills name (<=255 chars) when a gfx950 device is current. */
int rk_device_info(char *name, int32_t name_len, int32_t *cu_count);

/* ---------------------------------------------------------------- graph ------------ */
/* Coalesced COO -> CSR.  Replaces the layout ATen's sparse.mm consumes: the int64 [2,nnz]
 * index tensor + fp32 values built at recad/dataset/implicit.py:320-326,295-296.
 * coo_row must be non-decreasing (torch .coalesce()).  All device pointers. */
int rk_coo_to_csr(int32_t n_rows, int64_t nnz, const int64_t *coo_row, const int64_t *coo_col,
                  const float *coo_val, int32_t *rowptr /*[n_rows+1]*/, int32_t *col /*[nnz]*/,
                  float *val /*[nnz]*/, void *stream);

/* Work schedule for rk_spmm_csr (load balance for power-law rows): every row is cut into
 * segments of <= 64 (or 128, chosen per graph) nonzeros; whole rows are packed into workgroups of 4 or 8
 * waves (best-fit decreasing), one segment per wave; a row with more segments than a workgroup has waves
 * is cut into pieces, one workgroup each, whose partial sums meet in scratch slots -- the last workgroup
 * of a row to arrive (agent-scope ticket) adds them in piece order and runs the epilogue.
 * Rows that fit one lane-group chunk (<= dim/4 nonzeros, dim in {32,64,128}) are packed 256/dim per
 * wave, one row per lane group.  Built on the host once per (graph, dim) (reads rowptr back:
 * synchronous) and only valid for SpMMs of that `dim`.  _build returns
 *   n_blocks       an opaque launch parameter (workgroup count plus flag bits) to hand back to the SpMM
 *                  entry points unchanged,
 *   n_words        the size in int32 words of the READ-ONLY device buffer `wave_desc` that _upload fills
 *                  (wave descriptors, workgroup metas, packed-row table): shareable by any number of
 *                  handles, models and streams,
 *   scratch_words  the size in int32 words (0 when the graph has no long row) of the MUTABLE scratch block
 *                  the SpMM entry points take: arrival counters + partial-sum slots of the long rows.
 *                  Caller-allocated, zero-filled once (the counters reset themselves), one block per
 *                  stream / handle: two SpMMs in flight on one schedule must not share it.
 * class_split > 0 (= n_users for the bipartite adjacency): rows < split and rows >= split are
 * scheduled separately and interleaved 4:4 over the 8 XCDs so each XCD L2 holds one table. */
typedef struct rk_schedule *rk_schedule_t;
int rk_csr_schedule_build(int32_t n_rows, const int32_t *rowptr, int32_t class_split, int32_t dim, void *stream,
                          rk_schedule_t *out, int32_t *n_blocks, int64_t *n_words, int64_t *scratch_words);
/* the same from a HOST rowptr, no HIP call (ABI 8): what the CPU tests and the sanitizer builds drive; _words copies the
 * n_words schedule words (what _upload sends to the device) into a host array */
int rk_csr_schedule_build_host(int32_t n_rows, const int32_t *rowptr_host, int32_t class_split, int32_t dim,
                               rk_schedule_t *out, int32_t *n_blocks, int64_t *n_words, int64_t *scratch_words);
int rk_csr_schedule_words(rk_schedule_t sched, int32_t *host_out /*[n_words]*/);
int rk_csr_schedule_upload(rk_schedule_t sched, int32_t *wave_desc, void *stream);
int rk_csr_schedule_destroy(rk_schedule_t sched);

/* D^-1/2 A D^-1/2 of the bipartite user-item graph straight into CSR, on device.
 * Replaces ImplicitData.getSparseGraph, recad/dataset/implicit.py:243-298 (scipy dok/lil).
 * r_ptr/r_idx: user->item CSR (device, item ids sorted ascending within a user).
 * Outputs: rowptr[U+I+1], col[2E], val[2E] (device).  `tmp`: int32[I+1] device scratch; the
 * transpose uses a radix sort of the edge keys with stream-ordered temporaries (hipMallocAsync). */
int rk_build_norm_adj(int32_t n_users, int32_t n_items, const int32_t *r_ptr, const int32_t *r_idx,
                      int32_t *rowptr, int32_t *col, float *val, int32_t *tmp, void *stream);

/* Y = A.X (+ add).  Replaces torch.sparse.mm(g, all_emb), recad/model/victim/lightgcn.py:107.
 * X, add (nullable), Y: device float[n_rows*dim], row-major; n_rows*dim*4 < 4 GiB.
 * scratch: device int32[scratch_words] of rk_csr_schedule_build (NULL when that was 0). */
int rk_spmm_csr(int32_t n_rows, const int32_t *rowptr, const int32_t *col, const float *val,
                const int32_t *wave_desc, int32_t n_blocks, int32_t *scratch, int32_t dim, const float *x,
                const float *add, float *y, void *stream);

/* rk_spmm_csr with every fused epilogue the LightGCN step uses, for callers that compose a step
 * themselves (the row-sharded multi-GPU trainer runs collectives between the launches).  Row r of
 * the CSR slab addresses row r of add/y/sum_in/sum_out/zero1/zero2/adam_*; x has x_rows rows.
 *   v = A.x (+ add);  y = v;  sum_out = (sum_in + v) * sum_scale;  zero1 = zero2 = 0;
 *   adam_t > 0: torch.optim.Adam step t on adam_p/m/v with gradient v (coef_scratch: device float[2]);
 *   adam_t < 0: the same with the coefficients already in coef_scratch (rk_adam_coef_advance: a device-resident step
 *   counter, for callers that replay a captured step). */
typedef struct rk_spmm_epilogue {
    const float *add;
    float *y;
    const float *sum_in;
    float *sum_out;
    float sum_scale;
    int32_t adam_t;
    float *zero1, *zero2;
    float *adam_p, *adam_m, *adam_v, *coef_scratch;
    float lr, beta1, beta2, eps;
    /* optional frontier of the gather operand: device uint32 bitmap over x's rows, bit c clear => x[c] is all zeros and
     * the entries (r, c) are skipped (first backward layer of a train step: x = dL/dlight is non-zero on the minibatch's
     * rows only).  The surviving terms keep their CSR order but are dealt to the wave's lane groups anew, so a filtered sum
     * equals the unfiltered one up to summation order (rounding, ~1e-7 relative) and is deterministic for a fixed bitmap;
     * NULL = gather everything.  rk_rows_mark_bits sets / clears the bits.  When `add` is the SAME pointer as x (t = A x + x, the
     * first backward layer), the addend of a row whose bit is clear is not read: it is zero by this contract. */
    const uint32_t *src_filter;
} rk_spmm_epilogue;
int rk_spmm_csr_ex(int32_t n_rows, const int32_t *rowptr, const int32_t *col, const float *val,
                   const int32_t *wave_desc, int32_t n_blocks, int32_t *scratch, int32_t dim, const float *x,
                   int64_t x_rows, const rk_spmm_epilogue *epi, void *stream);

/* ---- LDS-resident sliced SpMM (recad_amd/csrc/spmm_lds.h): the fast form of the same torch.sparse.mm
 * (lightgcn.py:107) for bipartite D^-1/2 A D^-1/2 graphs whose class tables fit a CU's 160 KB LDS (ml1m / Amazon-game
 * size: <= ~9 K rows per class at 4 floats per slice).  Y[:, s] = A . X[:, s] is computed per column slice: a workgroup
 * stages the source class's whole slice in LDS once and every nonzero becomes one ds_read_b128; the matrix shrinks to a
 * 16-bit column stream because A is binary: y[r] = dinv[r] * sum_c dinv[c] * x[c], dinv = deg^-1/2 (implicit.py:259-277;
 * differs from the stored-value product by rounding only, ~1e-7 relative).
 * Gathered operands use a SLICED layout: users block [dim/Su][U][Su] followed by the items block [dim/Si][I][Si]
 * (Su = 1 << lsu, Si = 1 << lsi floats); rk_lds_pack / rk_lds_unpack convert from / to row-major [U+I, dim].
 * _build reads rowptr / col / val back (synchronous), checks that the graph qualifies and returns *n_words == 0 when it
 * does not (callers then keep rk_spmm_csr); otherwise the plan goes into a 16-byte aligned READ-ONLY device buffer of
 * n_words int32 (_upload), shareable by any number of handles and streams -- the kernel has no global scratch. */
typedef struct rk_lds_info {
    int32_t n_wg, lds_bytes, lpa, lpb;        /* launch shape: workgroups, dynamic LDS bytes, lanes per entry of the two halves */
    int32_t n_users, n_items, dim, lsu, lsi;  /* sliced-layout parameters */
    int32_t chunk;                            /* chunk caps (half 0 | half 1 << 16), informational */
    int32_t wgx_ofs, dinv_ofs, perm0_ofs, perm1_ofs, mq_ofs;   /* word offsets of the plan's sections: the launches pass them as kernel arguments (ABI 8) */
    int32_t reserved;
} rk_lds_info;
typedef struct rk_lds_plan *rk_lds_plan_t;
int rk_lds_plan_build(int32_t n_users, int32_t n_items, const int32_t *rowptr, const int32_t *col, const float *val /*nullable*/,
                      int32_t dim, void *stream, rk_lds_plan_t *out, int64_t *n_words, rk_lds_info *info);
/* the same from HOST arrays, no HIP call (n_cu = compute units to fill): what the CPU tests drive */
int rk_lds_plan_build_host(int32_t n_users, int32_t n_items, const int32_t *rowptr, const int32_t *col, const float *val,
                           int32_t dim, int32_t n_cu, rk_lds_plan_t *out, int64_t *n_words, rk_lds_info *info);
int rk_lds_plan_words(rk_lds_plan_t plan, int32_t *host_out /*[n_words]*/);
int rk_lds_plan_upload(rk_lds_plan_t plan, int32_t *dev /*[n_words], 16-byte aligned*/, void *stream);
int rk_lds_plan_destroy(rk_lds_plan_t plan);
/* n_arrays arrays of (U+I)*dim floats, `stride` floats apart, converted in one launch */
int rk_lds_pack(const rk_lds_info *info, const float *row_major, float *sliced, int32_t n_arrays, int64_t stride, void *stream);
int rk_lds_unpack(const rk_lds_info *info, const float *sliced, float *row_major, int32_t n_arrays, int64_t stride, void *stream);
/* v = A.x (+ add); y = v; sum_out = (sum_in + v) * sum_scale; zero1 = zero2 = 0; Adam as in rk_spmm_csr_ex.
 * x, add, sum_in, zero1, zero2, adam_shadow: sliced.  y / sum_out: sliced unless their *_row_major flag is set.
 * adam_p / m / v: row-major; adam_shadow (nullable) receives a sliced copy of the updated parameters. */
typedef struct rk_lds_epilogue {
    const float *add;
    float *y;
    const float *sum_in;
    float *sum_out;
    float sum_scale;
    int32_t y_row_major, sum_out_row_major, adam_t;
    float *zero1, *zero2;
    float *adam_p, *adam_m, *adam_v, *adam_shadow, *coef_scratch;
    float lr, beta1, beta2, eps;
    uint64_t *stamps;   /* diagnostic, nullable: device uint64[4 * n_wg], wall-clock stamps (start, staged, gathered, done) per workgroup */
} rk_lds_epilogue;
int rk_spmm_lds(const rk_lds_info *info, const int32_t *plan, const float *x, const rk_lds_epilogue *epi, void *stream);

/* The row-sharded trainer's per-step index work (the reference's getEmbedding gathers, lightgcn.py:122-130, split over ranks):
 * out[i, :] = mask[i] * src[idx[i], :] (mask NULL = 1; a zero mask entry gives exact zeros), and a[idx[i], :] = b[idx[i], :] = 0
 * (b nullable).  idx: device int64[n]. */
int rk_rows_gather_masked(int32_t dim, const float *src, const int64_t *idx, const float *mask, int64_t n, float *out, void *stream);
int rk_rows_zero(int32_t dim, float *a, float *b, const int64_t *idx, int64_t n, void *stream);
/* bits[idx[i] >> 5] |= 1 << (idx[i] & 31) (set != 0), or the words holding those bits = 0 (set == 0) */
int rk_rows_mark_bits(uint32_t *bits, const int64_t *idx, int64_t n, int32_t set, void *stream);

/* counter[0] += 1; coef = {lr / (1 - beta1^t), sqrt(1 - beta2^t)} for t = the new counter value (device int32 / float[2]). */
int rk_adam_coef_advance(float *coef, int32_t *counter, float lr, float beta1, float beta2, void *stream);

/* BPR forward+backward of ONE minibatch on explicit node rows (lightgcn.py:122-165): rows_u/p/n
 * index emb/gprop/gego directly (item rows already offset), and light too unless light_compact != 0:
 * then light is a compact [3*nb, dim] block holding the propagated rows of the minibatch in the order
 * users, positives, negatives (triplet b reads rows b, nb+b, 2nb+b) -- what the row-sharded trainer
 * assembles with one small all-reduce instead of all-gathering the whole light table.
 * gprop += dL/dlight / (L+1), gego += that + the L2-reg gradient; loss_partials: device
 * float[RK_LOSS_PARTIALS] (every entry is written). */
int rk_bpr_rows(int32_t dim, int32_t n_layers, float lambda, const float *light, int32_t light_compact,
                const float *emb, float *gprop, float *gego, const int64_t *rows_u, const int64_t *rows_p,
                const int64_t *rows_n, int32_t nb, float *loss_partials, void *stream);

/* The same minibatch with the ordered scatter of rk_lightgcn_set_deterministic (no atomics: plain stores, one wave per
 * touched row; gprop / gego must be zero on the minibatch's rows): light is always the compact [3*nb, dim] block, and
 * keys = device uint64[3*nb], the batch's incidences (row << 20) | (3*b + role) with row < 2^24, role 0/1/2 = user/positive/negative,
 * SORTED ascending -- the caller's sort (the row-sharded trainer sorts a whole epoch with one batched torch.sort).
 * Every rank that runs it on the same triplets gets bit-identical gradient rows. */
int rk_bpr_rows_ordered(int32_t dim, int32_t n_layers, float lambda, const float *light, const float *emb, float *gprop,
                        float *gego, const int64_t *rows_u, const int64_t *rows_p, const int64_t *rows_n, int32_t nb,
                        const uint64_t *keys, float *loss_partials, void *stream);

/* ---------------------------------------------------------------- LightGCN --------- */
typedef struct rk_lightgcn_desc {
    int32_t n_users, n_items, dim, n_layers;
    float lambda, lr, beta1, beta2, eps;      /* default.py:105-115; torch.optim.Adam defaults */
    int32_t reserved0;
    /* normalised adjacency, CSR over N = n_users + n_items nodes (device) */
    const int32_t *rowptr, *col;
    const float *val;
    const int32_t *wave_desc;                 /* from rk_csr_schedule_upload */
    int32_t n_blocks, reserved1;
    /* parameters and Adam moments, row-major fp32 (device); updated in place.  The item
     * block must directly follow the user block (item_emb == user_emb + n_users*dim, same for
     * m/v): E0 = [users; items] is then ONE [N,dim] matrix and torch.cat (lightgcn.py:88) is
     * a no-op. */
    float *user_emb, *item_emb;               /* embedding_user/.item.weight, lightgcn.py:40-45 */
    float *m_user, *v_user, *m_item, *v_item; /* exp_avg / exp_avg_sq of torch.optim.Adam */
    /* workspace, each float[N*dim] (device) */
    float *buf_a, *buf_b, *light, *gprop, *gego;
    float *grad;                              /* nullable: receives dLoss/dE0 [N*dim] */
    int32_t *state;                           /* device int32[16], 8-byte aligned, owned by the handle's user */
    float *coef;                              /* device float[2*RK_MAX_GRAPH_STEPS] (3*RK_MAX_GRAPH_STEPS with cnt) */
    int32_t *spmm_scratch;                    /* int32[scratch_words] of rk_csr_schedule_build, zero-filled, owned by this
                                               * handle's user (NULL when scratch_words == 0) */
    /* optional: device uint32[(N+31)/32] bitmap of the current minibatch's rows; when given (and
     * n_layers >= 2) the last forward layer computes only those rows of `light` */
    uint32_t *row_bits;
    /* optional graph dropout of the TRAINING propagation (lightgcn.py:62-80,91-95; config keys dropout /
     * keep_prob): keep_prob in (0,1) enables it (0 = off).  Every stored entry of the adjacency is kept with
     * probability keep_prob and divided by it, one fresh mask per train step (counter-based RNG of drop_seed,
     * the step index and the entry's position: the stream differs from torch.rand, the distribution does
     * not).  The mask is drawn per stored entry, so the propagated graph is not symmetric and the backward
     * applies its transpose: tpos[e] = position of the entry (col[e], row(e)) (device int32[nnz]). */
    float keep_prob;
    int32_t reserved3;
    uint64_t drop_seed;
    const int32_t *tpos;
    /* optional LDS-resident propagation (rk_lds_plan_*): lds_plan != NULL switches every SpMM of this handle to
     * rk_spmm_lds's kernel (not with graph dropout).  buf_a / buf_b / gprop / gego then hold SLICED data (same sizes),
     * lsum, e0s, ms, vs are four more float[N*dim] work buffers: the running layer sum and the SLICED working copies of
     * E0 and the Adam moments -- refreshed from user_emb / m_user / v_user at the start of every train_epoch call (e0s
     * also by propagate), updated by the fused Adam, written back to the row-major tensors at the end of the call.
     * row_bits is unused. */
    const int32_t *lds_plan;
    rk_lds_info lds_info;
    float *lsum, *e0s, *ms, *vs;
    int32_t *cnt;                             /* optional, int32[N]: per-node incidence counts of the minibatch; with it the L2-reg
                                               * gradient is applied in closed form (lambda / nb * count * E0) and the BPR kernel
                                               * scatters three rows per triplet instead of six; coef must then hold
                                               * 3 * RK_MAX_GRAPH_STEPS floats */
    int32_t *row_blocks;                      /* optional (ABI 9, row-gather path, with row_bits and n_layers >= 3): int32[4 + number of
                                               * schedule blocks], this handle's own: word 0 counts, words 4.. list the workgroups of
                                               * the schedule that hold a row of the current minibatch, so that the row-filtered last
                                               * forward layer starts min(n_blocks, 3 * batch + row_blocks_extra) workgroups instead of
                                               * all of them (csrc/spmm.h SpmmArgs::blk_mode) */
    int32_t row_blocks_extra, reserved4;      /* upper bound of the schedule's long-row pieces (scratch_words / dim is one) */
    int32_t *lds_sync;                        /* optional (ABI 8), int32[RK_LDS_SYNC_WORDS], zero-initialised, this handle's own: with
                                               * it the L propagation layers of a forward / backward pass run as ONE launch (<= 4
                                               * layers per launch) whose workgroups hand the layers over to each other per column
                                               * group through agent-scope counters kept here (recad_amd/csrc/spmm_lds.h:
                                               * spmm_lds_multi_kernel; same sums, same bits as one launch per layer).  Word
                                               * RK_LDS_SYNC_ERR is set if a wait ever gave up and stays set until rk_lightgcn_sync_status reads it. */
} rk_lightgcn_desc;
#define RK_LDS_SYNC_WORDS 2560
#define RK_LDS_SYNC_ERR 2336
#define RK_MAX_GRAPH_STEPS 64

typedef struct rk_lightgcn *rk_lightgcn_t;

int rk_lightgcn_create(const rk_lightgcn_desc *desc, rk_lightgcn_t *out);
/* *status = desc.lds_sync[RK_LDS_SYNC_ERR] (0 = every in-launch hand-off completed; no lds_sync: 0).  Synchronises the stream.
 * The word is STICKY: no launch and no call prologue clears it (the caller's zero-initialisation and this call are the only
 * resets), so a time-out in any propagate / train call since the last status read is reported; a non-zero status is
 * cleared by the read. */
int rk_lightgcn_sync_status(rk_lightgcn_t h, int32_t *status, void *stream);
int rk_lightgcn_destroy(rk_lightgcn_t h);

/* LightGCN.computer(), recad/model/victim/lightgcn.py:82-113: desc.light[N,dim] =
 * mean_l(A^l [U;I]).  Users are rows [0,U), items rows [U,N). */
int rk_lightgcn_propagate(rk_lightgcn_t h, void *stream);
/* The same through a dropped-out graph (computer() of a module in training mode with dropout > 0,
 * lightgcn.py:91-95): mask_seed picks the mask; requires desc.keep_prob > 0. */
int rk_lightgcn_propagate_dropout(rk_lightgcn_t h, uint64_t mask_seed, void *stream);

/* One epoch of LightGCN.train_step, recad/model/victim/lightgcn.py:137-169, over the
 * pre-sampled triplets (device int64[n], the tensors ImplicitData.generate_batch yields,
 * recad/dataset/implicit.py:416-437): ceil(n/batch) steps of forward, BPR loss + L2 reg,
 * backward, dense Adam.  adam_t0 = optimizer steps already taken.  loss_partials: device
 * float[ceil(n/batch)*RK_LOSS_PARTIALS]; step s's loss is the sum of its RK_LOSS_PARTIALS
 * entries (fixed order => reproducible).  apply_update=0 leaves parameters untouched and
 * only fills desc.grad (testing).
 * graph_steps > 1 replays captured hipGraphs: an epoch of <= RK_MAX_GRAPH_STEPS steps is ONE replay of a whole-call graph
 * (scatter-target zeroing and, on the LDS path, the layout conversions included); a longer one is replayed in chunks of
 * graph_steps steps plus one remainder graph (a single trailing step is launched kernel by kernel).  The triplet and loss
 * pointers reach the kernels through desc.state, not through the captured launches, so the graphs (cached per handle, 8
 * LRU slots) serve any buffers; they must stay valid until the stream has run the call. */
int rk_lightgcn_train_epoch(rk_lightgcn_t h, const int64_t *users, const int64_t *pos, const int64_t *neg,
                            int64_t n, int32_t batch, int32_t adam_t0, float *loss_partials,
                            int32_t apply_update, int32_t graph_steps, void *stream);

/* Optional: capture, instantiate and upload ahead of time the hipGraphs that an epoch of n triplets in batches of `batch`
 * will replay, so that no epoch (or timed region) pays for it.  The reference has no counterpart (its step is eager ATen,
 * lightgcn.py:137-169). */
int rk_lightgcn_prepare(rk_lightgcn_t h, int64_t n, int32_t batch, int32_t apply_update, int32_t graph_steps, void *stream);

/* Ordered (reproducible) gradient scatter for rk_lightgcn_train_epoch.  The reference scatters the minibatch's embedding
 * gradients through autograd's index_select backward (lightgcn.py:124-129,166): a sequential sum on CPU, float atomics on a
 * GPU.  Default here: float atomics too (sum order not fixed; the loss is reproducible, gradients to ~1e-7).  on != 0:
 * train_epoch first sorts the epoch's 3n (node row, triplet, role) incidences once (handle-owned buffers, 48 bytes per
 * triplet) and every step adds each row's contributions in (triplet index, role) order with plain stores, one wave per
 * touched row instead of one per triplet -- bit-identical gradients and parameters from run to run (+5 % per ml1m step).  Limits: < 2^20 steps per epoch, < 2^24 node
 * rows, batch < 349 525.  Not used by rk_bpr_rows (the row-sharded trainer). */
int rk_lightgcn_set_deterministic(rk_lightgcn_t h, int32_t on);

/* ---------------------------------------------------------------- shared ops ------- */
/* out[b] = <utab[users[b]], itab[items[b]]> (+ ubias[users[b]] + ibias[items[b]] + mean when
 * ubias != NULL).  LightGCN.forward tail (lightgcn.py:179-182) and MF.forward (mf.py:40-47). */
int rk_pair_scores(int32_t dim, const float *utab, const float *itab, const float *ubias, const float *ibias,
                   float mean, const int64_t *users, const int64_t *items, int64_t n, float *out, float dropout,
                   uint64_t drop_seed, void *stream);   /* dropout > 0: nn.Dropout on the score (mf.py:47), mask of drop_seed */

/* Dense torch.optim.Adam step on one tensor (recad/utils.py:181-183): t = 1-based step. */
int rk_adam_step(int64_t n, float *param, const float *grad, float *m, float *v, int32_t t, float lr,
                 float beta1, float beta2, float eps, void *stream);
/* the same, coefficients {lr / (1 - beta1^t), sqrt(1 - beta2^t)} from device float[2] (rk_adam_coef_advance): for captured steps (ABI 8) */
int rk_adam_step_dev(int64_t n, float *param, const float *grad, float *m, float *v, const float *coef, float beta1, float beta2, float eps,
                     void *stream);

/* Full-catalog scoring + top-K + target rank for a block of users: replaces the per-user
 * loop of Normal.user_item_model_generate, recad/workflow/normal.py:57-93.
 *   scores[b,i] = <utab[user_ids[b]], itab[i]> (+ ubias[user_ids[b]] + ibias[i] + mean when ibias != NULL)
 * computed with fp32 MFMA (utab/ubias are the whole user tables: the block's rows are gathered on the way
 * in); items in the user's seen list (CSR seen_ptr/seen_idx indexed by user_ids[b], ids sorted ascending)
 * are excluded (normal.py:133-143).  Outputs per user: top_ids/top_scores[K] sorted by (score desc, item id
 * asc), padded with -1/-inf; for each target t its score and rank among the unseen items (hit@k <=> rank < k).
 * Two paths, identical results (ids, scores, target scores and ranks bit for bit):
 *   RK_SCORE_GEMM  -- GEMM into a [nb, n_items] matrix (row stride plan.ld_scores) in `scratch` + a selection pass: small user blocks, catalogues of < 16 384
 *                     items, and everything the panel form does not take;
 *   RK_SCORE_PANEL -- the register-resident panel form, recad_amd/csrc/score_panel.h: a workgroup holds the scores of 16 / 32
 *                     users x 1920 items in its registers, selects from there and never writes a score (K <= 256, n_targets <= 4,
 *                     dim <= 256; scratch = a k-permuted copy of the item table, n_items * 16 * ceil(dim / 16) floats).  It
 *                     parallelises over the users only, so it is the default from 16 384 items on for blocks of >= 8192 users
 *                     (>= 4096 at dim <= 64).
 * The path is an ARGUMENT, not process state (ABI 8): rk_score_topk_plan() fills a plan for one (nb, n_items, dim, K, n_targets) --
 * the library's choice when request is NULL / request->path == RK_SCORE_AUTO, the requested path and shape knobs otherwise
 * (RK_EINVAL if that path does not support the request) -- with the scratch it needs, and rk_score_topk() runs exactly that plan:
 * sizing and launch cannot disagree.  (Round 2's fused sweep, score_select.h, is gone: measured slower than both paths on every
 * shape, its last default corner included -- profiles/r04_score_corner.txt.) */
#define RK_SCORE_AUTO 0
#define RK_SCORE_GEMM 1
#define RK_SCORE_PANEL 2
typedef struct rk_score_plan {
    int32_t path;             /* RK_SCORE_GEMM / RK_SCORE_PANEL (RK_SCORE_AUTO only in a request) */
    int32_t panel_rows;       /* PANEL: 16 or 32 user rows per workgroup (request: 0 = by shape) */
    int32_t panel_ntw;        /* PANEL: 16-item tiles per wave and panel: 8 (1024-item panels) or 15 (1920) (request: 0 = by catalogue size) */
    int32_t panel_safe;       /* PANEL: 1 = every panel through the exact safe form (tests) */
    int32_t nb, n_items, dim, K, n_targets;   /* what the plan was made for: rk_score_topk refuses anything else */
    int32_t ld_scores;        /* GEMM (ABI 9): row stride of the score matrix in floats = n_items rounded up to 32, so that every row and every
                               * 16-column store segment of the GEMM's tiles starts on a 128- / 64-byte line (3 702 -> 3 712); PANEL: 0 */
    int32_t reserved[2];
    int64_t scratch_floats;   /* device float[scratch_floats], 16-byte aligned (GEMM: nb * ld_scores) */
} rk_score_plan;
int rk_score_topk_plan(int32_t nb, int32_t n_items, int32_t dim, int32_t K, int32_t n_targets, const rk_score_plan *request /*nullable*/,
                       rk_score_plan *out);
int rk_score_topk(int32_t dim, const float *utab, int32_t nb, const int32_t *user_ids, const float *itab,
                  int32_t n_items, const float *ubias, const float *ibias, float mean,
                  const int32_t *seen_ptr, const int32_t *seen_idx, int32_t K, int32_t *top_ids,
                  float *top_scores, const int32_t *targets, int32_t n_targets, float *target_score,
                  int32_t *target_rank, const rk_score_plan *plan, float *scratch, void *stream);

/* ---------------------------------------------------------------- MF --------------- */
/* One epoch of MF.train_step, recad/model/victim/mf.py:49-69: logits (mf.py:40-47),
 * BCE-with-logits (mf.py:32,59-60), backward, dense Adam on the four tables.
 * grads: device float[U*d + I*d + U + I] scratch (zeroed by the call).
 * moments: m then v, same layout as grads, each [U*d + I*d + U + I]. */
int rk_mf_train_epoch(int32_t n_users, int32_t n_items, int32_t dim, float *user_emb, float *item_emb,
                      float *user_bias, float *item_bias, float mean, float *m, float *v, float *grads,
                      const int64_t *users, const int64_t *items, const int64_t *labels, int64_t n,
                      int32_t batch, int32_t adam_t0, float lr, float beta1, float beta2, float eps,
                      float *loss_partials, int32_t apply_update, float dropout, uint64_t drop_seed, void *stream);
/* dropout > 0: nn.Dropout on the logit (mf.py:27,47), one counter-based mask per optimizer step and sample. */

/* ---------------------------------------------------------------- samplers ---------- */
/* BPR triplets with the semantics of pairwise_sample, recad/dataset/implicit.py:50-74: n_draws
 * uniform user draws with replacement; a user without positives yields valid=0 (the reference
 * skips the draw); positive uniform over the user's list; negative uniform over the items NOT in
 * it.  pos_ptr/pos_idx: user->sorted item ids (device int32).  Counter-based RNG of `seed`. */
int rk_bpr_sample(int32_t n_users, int32_t n_items, const int32_t *pos_ptr, const int32_t *pos_idx,
                  int64_t n_draws, uint64_t seed, int64_t *users, int64_t *pos, int64_t *neg, int32_t *valid,
                  void *stream);

/* (user, item, label) rows of pointwise_sample, recad/dataset/implicit.py:77-91: each train edge
 * once with label 1, followed by negative_ratio rows with label 0 whose item is uniform (with
 * replacement) over the user's non-interacted items.  Outputs: int64[n_edges*(negative_ratio+1)]. */
int rk_pointwise_sample(int32_t n_users, int32_t n_items, const int32_t *train_ptr, const int32_t *train_idx,
                        int64_t n_edges, int32_t negative_ratio, uint64_t seed, int64_t *users, int64_t *items,
                        int64_t *labels, void *stream);

/* Pass 2 of rk_score_topk on a ready score matrix scores[nb, n_items] (rows too long to stage
 * in LDS are modified in place: seen items are overwritten with -inf).  Used for victims whose
 * scores are not a dot product (NCF). */
int rk_topk_rows(float *scores, int32_t nb, int32_t n_items, const int32_t *user_ids, const int32_t *seen_ptr,
                 const int32_t *seen_idx, int32_t K, int32_t *top_ids, float *top_scores,
                 const int32_t *targets, int32_t n_targets, float *target_score, int32_t *target_rank,
                 void *stream);

/* HR@k numerators (normal.py:86-92,150-156: hit@k <=> the target's rank among the unseen items < k):
 * counts[t*nk + q] = #{b < n : target_rank[b*n_targets + t] < ks[q]}.  All pointers on the device. */
int rk_hit_counts(const int32_t *target_rank, int64_t n, int32_t n_targets, const int32_t *ks, int32_t nk,
                  int32_t *counts, void *stream);

/* Users the reference evaluates (normal.py:133-143): every user whose train list (CSR seen_ptr/seen_idx, item ids
 * sorted) is non-empty and holds none of the targets.  user_ids[0..count) = their ids in ascending order; all
 * pointers on the device; flags_scratch: int32[n_users]; count: int32[1]. */
int rk_eligible_users(int32_t n_users, const int32_t *seen_ptr, const int32_t *seen_idx, const int32_t *targets,
                      int32_t n_targets, int32_t *flags_scratch, int32_t *user_ids, int32_t *count, void *stream);

/* pred_shift (normal.py:147-149): out[0] = mean(score_after[i] - score_before[i]), out[1] = the sum, i < n, in
 * double, fixed summation order.  out: device double[2]. */
int rk_pred_shift(const float *score_before, const float *score_after, int64_t n, double *out, void *stream);

/* out[b*n_items + i] = <utab[user_ids[b]], itab[i]> (+ ubias[user_ids[b]] + ibias[i] + mean), optionally through
 * nn.Dropout(dropout) on the score -- MF.forward of a module in training mode (mf.py:40-47); feeds rk_topk_rows. */
int rk_score_matrix(int32_t dim, const float *utab, int32_t nb, const int32_t *user_ids, const float *itab,
                    int32_t n_items, const float *ubias, const float *ibias, float mean, float dropout,
                    uint64_t drop_seed, float *out, void *stream);

/* LightGCN.getUsersRating (lightgcn.py:115-120): out[b*n_items + i] = sigmoid(<utab[user_ids[b]], itab[i]>), the
 * dot product on the fp32 MFMA GEMM (bit-identical to the k-ordered fmaf chain). */
int rk_users_rating(int32_t dim, const float *utab, int32_t nb, const int32_t *user_ids, const float *itab,
                    int32_t n_items, float *out, void *stream);

/* ---------------------------------------------------------------- NCF -------------- */
#define RK_NCF_MAX_LAYERS 8
#define RK_NCF_MAX_TENSORS 24
/* NCF (recad/model/victim/ncf.py:9-58; variants: `mode` below).  E = factor * 2^(n_layers-1) is the MLP embedding
 * width; tower layer l is Linear(in_l -> in_l/2) + ReLU with in_l = factor * 2^(n_layers-l)
 * (ncf.py:41-47); W[l] is nn.Linear's [out, in] row-major weight.  Tensor order of grad/m/v:
 * ug, ig, um, im, W[0..L-1], b[0..L-1], pw, pb. */
typedef struct rk_ncf_desc {
    int32_t n_users, n_items, factor, n_layers;
    float lr, beta1, beta2, eps;
    float *ug, *ig, *um, *im;                  /* embed_{user,item}_{GMF,MLP}.weight */
    float *W[RK_NCF_MAX_LAYERS], *b[RK_NCF_MAX_LAYERS];
    float *pw, *pb;                            /* predict_layer.weight [2*factor] (GMF part first), .bias [1] */
    float *grad[RK_NCF_MAX_TENSORS], *m[RK_NCF_MAX_TENSORS], *v[RK_NCF_MAX_TENSORS];
    float *acts, *dacts;                       /* each float[max_batch * sum_{l=0..L} in_l] */
    float *d0;                                 /* float[max_batch] */
    int32_t max_batch, reserved;
    /* K-slice workspace.  (1) backward (dX) tower GEMMs whose whole-K tiling would not fill the chip (batch 1024 against a
     * few hundred outputs): K-slices park partial products here and are added in slice order (deterministic); optional
     * for that.  (2) REQUIRED by rk_ncf_train_epoch when a tower layer's input width is a multiple of 256: the training
     * forward of such a layer sums its products in 8 consecutive k-blocks combined pairwise (ATen-like error: the ReLU
     * gates it decides are the masks of the backward; csrc/ncf.hip gemm_fwd_blocked, oracle ncf_forward_one_ex), the blocks
     * parked here as [8][rows, out] -- 8 * max_batch * (widest layer output) floats run every layer in one piece, less
     * makes row chunks (>= 8 * 64 * out).  rk_ncf_forward (scoring) keeps the single k-ordered chain. */
    float *gemm_scratch;
    int64_t gemm_scratch_floats;
    float *wgrad_part;                         /* training: float[ceil(max_batch/64) * (2*factor + 1)] */
    /* model variant (ncf.py:49-52,112-131): RK_NCF_NEUMF = 'NeuMF-end' / 'NeuMF-pre' (they differ in initialisation
     * only, ncf.py:78-110), RK_NCF_MLP = tower only, RK_NCF_GMF = element-wise product only; pw has 2*factor entries
     * for NeuMF and factor entries otherwise.  Tensors a variant does not use receive zero gradients. */
    int32_t mode;
    /* nn.Dropout(p) in front of every tower layer (ncf.py:44); 0 = off.  Counter-based masks of drop_seed, the call
     * (train: optimizer step; rk_ncf_forward: drop_call and the chunk) and the layer: the stream differs from
     * torch's, the distribution does not.  The reference's workflows score WITHOUT .eval() (normal.py:61-67), so a
     * module in training mode passes its dropout here for rk_ncf_forward too. */
    float dropout;
    uint64_t drop_seed;
    int32_t drop_call, reserved2;
} rk_ncf_desc;
#define RK_NCF_NEUMF 0
#define RK_NCF_MLP 1
#define RK_NCF_GMF 2

/* NCF.forward (ncf.py:112-131) for n pairs -> out[n].  Pairs are (users[b], items[b]), or -- when
 * user_ids != NULL -- the full catalog of each listed user: pair q = (user_ids[q / n_items_catalog],
 * q % n_items_catalog), which is how the evaluation fills a [users, items] score matrix. */
int rk_ncf_forward(const rk_ncf_desc *desc, const int64_t *users, const int64_t *items,
                   const int32_t *user_ids, int32_t n_items_catalog, int64_t n, float *out, void *stream);

/* One epoch of NCF.train_step (ncf.py:133-153): forward, BCE-with-logits, backward through the
 * tower (fp32 MFMA GEMMs), embedding scatter-add, dense Adam on every tensor. */
int rk_ncf_train_epoch(const rk_ncf_desc *desc, const int64_t *users, const int64_t *items,
                       const int64_t *labels, int64_t n, int32_t batch, int32_t adam_t0,
                       float *loss_partials, int32_t apply_update, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RECAD_HIP_H */
