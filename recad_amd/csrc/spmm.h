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
#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include "host/layout.h"

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
    // long rows (> waves-per-workgroup segments): arrival counters int[n_long] (zero between launches: the last
    // arriver resets its counter) followed, 16-byte aligned, by the partial-sum slots float[n_slots][d].
    // Caller-owned and PER STREAM: two SpMMs in flight on the same schedule need two scratch blocks.
    int *scratch;
    // Batch sparsity of a train step (bitmap over node rows, bit r = row r is a user/pos/neg row of
    // this step's minibatch): the last forward layer only needs those rows of `light`.  (Skipping
    // the gathers of all-zero gradient rows in the first backward layer was measured too: the
    // predicated loads cost more than they save, 26.0 vs 18.1 us.)
    const unsigned *row_filter;  // skip rows whose bit is clear (outputs not written)
    // Frontier-sparse gather operand (first backward layer of a train step: x = dL/dlight is non-zero only on the
    // minibatch's <= 3B rows): bit c clear => x[c] is all zeros and entry (r, c) is skipped.  Every 64-entry chunk of
    // the column stream is tested against the bitmap and the hits are compacted (ds_permute) before the row gathers,
    // so the launch moves sum_{s in batch} deg(s) rows instead of nnz.  The surviving terms keep their order in the
    // stream but are dealt to the lane groups anew (gather_round takes entry t*NG+grp of the COMPACTED chunk), so a
    // filtered sum equals the unfiltered one up to summation order -- deterministic for a fixed bitmap, not bit-equal.
    const unsigned *src_filter;
    // The row-filtered launch (last forward layer of a train step) used to start EVERY workgroup of the schedule to find the
    // few that hold a minibatch row: at the yelp shape 22 K workgroups took 67 us to compute <= 3 072 rows (dispatch, one
    // descriptor load, one bitmap test, exit).  With a block list the launch before it (blk_mode 2) appends the ids of the
    // workgroups that hold a marked row (one __syncthreads_or + one atomic per marked workgroup), and the filtered launch
    // (blk_mode 3) starts only min(n_blocks, blk_cap) workgroups that take their ids from the list.  blk_mode 1 (the
    // marking launch) zeroes the count.  List order varies run to run; every workgroup's work is independent of it.
    int blk_mode, blk_cap;
    int *blk_count, *blk_list;
    const unsigned *blk_bits;
    unsigned *mark_bits;         // set the bits of the current minibatch (first forward layer)
    unsigned *clear_bits;        // zero the bitmap (last backward layer)
    int n_words, mark_U, mark_k;
    const int *mark_state;
    // Graph dropout of a training forward/backward (lightgcn.py:62-80): drop_thresh24 = keep_prob * 2^24
    // (0 = off); every stored entry e is kept iff rk_drop_keep(seed_step, id, thresh) and scaled by
    // drop_inv_keep.  id = e in the forward, drop_tpos[e] (the position of the transposed entry) in the
    // backward, which applies G^T through the same symmetric-structure CSR.
    // drop_mode 1: seed_step from the global step index drop_state[ST_ADAM_T] + drop_k (first forward layer;
    // block 0 also stashes it in drop_state[ST_DROP_LO/HI]); 2: read the stash (the last backward launch of a
    // chunk bumps ST_ADAM_T itself); 3: drop_seed is the step seed (inference call in training mode).
    unsigned drop_thresh24;
    float drop_inv_keep;
    int drop_mode, drop_k;
    unsigned long long drop_seed;
    int *drop_state;
    const int *drop_tpos;
    SpmmEpi e;
};

struct DropCtx {
    unsigned long long seed_step;
    unsigned thresh24;
    float inv_keep;
    const int *tpos;
};
__device__ __forceinline__ DropCtx drop_ctx(const SpmmArgs &a)
{
    DropCtx c;
    c.thresh24 = a.drop_thresh24; c.inv_keep = a.drop_inv_keep; c.tpos = a.drop_tpos; c.seed_step = 0ULL;
    if (a.drop_thresh24) {
        if (a.drop_mode == 1) {
            c.seed_step = rk_drop_step_seed(a.drop_seed, (unsigned long long)(unsigned)(a.drop_state[ST_ADAM_T] + a.drop_k));
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                a.drop_state[ST_DROP_LO] = (int)(unsigned)c.seed_step;
                a.drop_state[ST_DROP_HI] = (int)(unsigned)(c.seed_step >> 32);
            }
        } else if (a.drop_mode == 2) {
            c.seed_step = (unsigned long long)(unsigned)a.drop_state[ST_DROP_LO] | ((unsigned long long)(unsigned)a.drop_state[ST_DROP_HI] << 32);
        } else {
            c.seed_step = a.drop_seed;
        }
    }
    return c;
}
// value of stored entry e under the step's dropout mask
__device__ __forceinline__ float drop_val(const DropCtx &c, int e, float v)
{
    if (!c.thresh24) return v;
    const unsigned id = (unsigned)(c.tpos ? c.tpos[e] : e);
    return rk_drop_keep(c.seed_step, id, c.thresh24) ? v * c.inv_keep : 0.f;
}

__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc)
{
    acc.x += a * x.x; acc.y += a * x.y; acc.z += a * x.z; acc.w += a * x.w;
    return acc;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// RK_NT_STREAMS (tuning build): the once-read col / val streams bypass the cache hierarchy's retention
#ifdef RK_NT_STREAMS
__device__ __forceinline__ int ld_col(const int *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ld_val(const float *p) { return __builtin_nontemporal_load(p); }
#else
__device__ __forceinline__ int ld_col(const int *p) { return *p; }
__device__ __forceinline__ float ld_val(const float *p) { return *p; }
#endif

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
#ifdef RK_SPMM_L1TEST
        cc &= RK_SPMM_L1TEST;  // diagnostic build: confine the gather to a few rows (L1/L2-hit ceiling)
#endif
        av[j] = ok ? aa : 0.f;
        xv[j] = *reinterpret_cast<const float4 *>(x + (unsigned)(cc * D + sub * 4));
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) acc = f4_fma(av[j], xv[j], acc);
    return acc;
}

// Partial sum of row segment [eb, ee) for the D/4 lanes that share `sub`; after the
// cross-group reduction every lane holds the total for its float4 slot.
template <int D, int UNMAX, bool DROP, bool FILT = false>
__device__ __forceinline__ float4 spmm_segment(const int *__restrict__ col, const float *__restrict__ val, int eb, int ee,
                                               const float *__restrict__ x, int lane, const DropCtx &dc,
                                               const unsigned *__restrict__ filt = nullptr)
{
    constexpr int G = D / 4, NG = 64 / G;
    const int grp = lane / G, sub = lane % G;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int c_next = 0;
    float a_next = 0.f;
    if (eb + lane < ee) { c_next = ld_col(col + eb + lane); a_next = DROP ? drop_val(dc, eb + lane, val[eb + lane]) : ld_val(val + eb + lane); }
    for (int base = eb; base < ee; base += 64) {
        int n = min(64, ee - base);
        int c = c_next;
        float a = a_next;
        c_next = 0; a_next = 0.f;
        if (base + 64 + lane < ee) { c_next = ld_col(col + base + 64 + lane); a_next = DROP ? drop_val(dc, base + 64 + lane, val[base + 64 + lane]) : ld_val(val + base + 64 + lane); }
        if (FILT) {
            // keep the entries whose source row is in the frontier, compacted to the low lanes in their original order
            const bool hit = lane < n && ((filt[(unsigned)c >> 5] >> (c & 31)) & 1u);
            const unsigned long long mask = __ballot(hit);
            n = __popcll(mask);
            if (n == 0) continue;
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            const int dst = hit ? rank : 63;   // lanes that miss park on the last lane: only read when all 64 hit, and then nobody misses
            c = __builtin_amdgcn_ds_permute(dst << 2, c);
            a = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(a)));
        }
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

// ---- long rows: pieces (one workgroup each) meet in scratch slots; the last to arrive combines.
// Hand-off form (cdna_hip_programming.md G16 / MI355X_MICROARCH.md "Valid forms"): the payload is
// written with 8-byte agent-scope atomic stores (write-through, sc1) by ONE wave, that wave drains
// vmcnt, one lane draws a ticket with a relaxed agent-scope fetch_add, and the wave whose add
// returned n_pieces-1 reads every slot with 8-byte agent-scope atomic loads (sc1) after the add
// has returned.  No fence, no placement assumption; summation in piece order => deterministic.
struct PieceRef {
    int4 meta;            // {n_pieces, piece index, first slot, counter index}
    const int4 *packed;   // {row, e_begin, e_end, 0} of the packed short rows
    float *partials;      // slot s at partials + s * stride
    int *counters;
    int stride;
    int n_slots;          // (bounds the buffer descriptor of the last arriver's slot reads)
};

__device__ __forceinline__ PieceRef piece_ref(const int4 *wave_desc, int n_blocks, int waves, int block, int *scratch)
{
    const int4 *hdr = wave_desc + (size_t)n_blocks * waves;  // {n_long, n_slots, dim, n_packed}
    PieceRef p;
    p.meta = hdr[1 + block];
    const int4 h = hdr[0];
    p.packed = hdr + 1 + n_blocks;
    p.counters = scratch;
    p.partials = reinterpret_cast<float *>(scratch + ((h.x + 3) & ~3));
    p.stride = h.z;
    p.n_slots = h.y;
    return p;
}

typedef unsigned long long u64_t;
__device__ __forceinline__ void slot_store(float *slot, int sub, float4 v)
{
    u64_t *q = reinterpret_cast<u64_t *>(slot + sub * 4);
    const u64_t lo = ((u64_t)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    const u64_t hi = ((u64_t)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
    __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 slot_load(const float *slot, int sub)
{
    u64_t *q = reinterpret_cast<u64_t *>(const_cast<float *>(slot) + sub * 4);
    const u64_t lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64_t hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                       __uint_as_float((unsigned)(hi >> 32)));
}

// Called by the piece's leader wave (all 64 lanes; lanes >= g_lanes carry no data).  Returns true
// in the wave that arrived last, with `acc` replaced by the row's total.
__device__ __forceinline__ bool piece_arrive(const PieceRef &p, int lane, int g_lanes, float4 &acc)
{
    if (lane < g_lanes) slot_store(p.partials + (size_t)(p.meta.z + p.meta.y) * p.stride, lane, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(p.counters + p.meta.w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0, 64);
    if (ticket != p.meta.x - 1) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the slot reads below the ticket)
    if (lane == 0) __hip_atomic_store(p.counters + p.meta.w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    const size_t slot_bytes = (size_t)p.n_slots * (size_t)p.stride * 4u;
    if (lane < g_lanes && slot_bytes > 0x7fffffffULL) {   // (a slot area beyond a buffer descriptor's 32-bit range: one atomic load per slot half)
        float4 t = slot_load(p.partials + (size_t)p.meta.z * p.stride, lane);
        for (int q = 1; q < p.meta.x; ++q) t = f4_add(t, slot_load(p.partials + (size_t)(p.meta.z + q) * p.stride, lane));
        acc = t;
    } else if (lane < g_lanes) {
        // The last arriver adds the pieces' slots in piece order.  EIGHT slot reads in flight per round (sc1 buffer loads: the
        // hand-off's load form, MI355X_MICROARCH.md visibility table row 1 -- every handed-off byte was stored write-through and is
        // read with sc1 after the ticket has returned): as a loop of agent-scope atomic loads each slot was its own round trip, and
        // a popular item's row is 50-100 pieces -- the critical path of the row-filtered last forward layer, where nothing else
        // is left to hide it (yelp-shaped f3: 44 us for 9 % of a layer's nonzeros).  Same order of additions, same bits.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.partials, 0, (int)(unsigned)slot_bytes, 0x00020000);
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        auto ld = [&](int q) -> float4 {
            const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((unsigned)(p.meta.z + q) * (unsigned)p.stride + (unsigned)lane * 4u) * 4u), 0, 16);
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        };
        float4 t = ld(0);
        int q = 1;
        for (; q + 8 <= p.meta.x; q += 8) {   // (sixteen in flight measured the same -- 31.6 against 32.0 us -- and cost 32 more registers)
            float4 u[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = ld(q + k);
#pragma unroll
            for (int k = 0; k < 8; ++k) t = f4_add(t, u[k]);
        }
        for (; q < p.meta.x; ++q) t = f4_add(t, ld(q));
        acc = t;
    }
    return true;
}

// DROP: graph dropout compiled in (a separate instantiation: carrying the mask state through the default
// kernel cost 7 % of a train step even with dropout switched off at run time)
template <int D, int UNMAX, int WAVES, int MINW, bool PACKED = false, bool DROP = false, bool FILT = false>
__global__ __launch_bounds__(WAVES * 64, MINW) void spmm_csr_kernel(const SpmmArgs a)
{
    constexpr int G = D / 4;
    __shared__ float4 part[WAVES][G];
    __shared__ int blk_flag[WAVES];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (a.e.bump && blockIdx.x == 0 && threadIdx.x == 0) {
        a.e.state[ST_STEP_BASE] += a.e.bump;
        a.e.state[ST_ADAM_T] += a.e.bump;
    }
    if (a.clear_bits)
        for (int i = blockIdx.x * (WAVES * 64) + threadIdx.x; i < a.n_words; i += gridDim.x * WAVES * 64) a.clear_bits[i] = 0u;
    if (a.mark_bits) {  // rows of this step's minibatch; consumed two launches later
        const int step = a.mark_state[ST_STEP_BASE] + a.mark_k;
        const long long ntrip = ((long long)(unsigned)a.mark_state[ST_NTRIP_LO]) | ((long long)a.mark_state[ST_NTRIP_HI] << 32);
        const int B = a.mark_state[ST_BATCH];
        const long long off = (long long)step * B;
        const int nb = (int)max(0LL, min((long long)B, ntrip - off));
        const int64_t *mu = st_ptr<const int64_t>(a.mark_state, ST_PTR_USERS), *mp = st_ptr<const int64_t>(a.mark_state, ST_PTR_POS),
                      *mn = st_ptr<const int64_t>(a.mark_state, ST_PTR_NEG);
        for (int b = blockIdx.x * (WAVES * 64) + threadIdx.x; b < nb; b += gridDim.x * WAVES * 64) {
            const int r0 = (int)mu[off + b], r1 = a.mark_U + (int)mp[off + b], r2 = a.mark_U + (int)mn[off + b];
            atomicOr(&a.mark_bits[r0 >> 5], 1u << (r0 & 31));
            atomicOr(&a.mark_bits[r1 >> 5], 1u << (r1 & 31));
            atomicOr(&a.mark_bits[r2 >> 5], 1u << (r2 & 31));
        }
    }
    DropCtx dc{};
    if (DROP) dc = drop_ctx(a);
    int bid = blockIdx.x;   // the schedule block this workgroup runs
    if (a.blk_mode == 1 && blockIdx.x == 0 && threadIdx.x == 0) a.blk_count[0] = 0;
    if (a.blk_mode == 3) {
        if ((int)blockIdx.x >= a.blk_count[0]) return;
        bid = a.blk_list[blockIdx.x];
    }
    int4 ds = a.wave_desc[(size_t)bid * WAVES + w];  // {row, eb, ee, nseg}
    const PieceRef pr = piece_ref(a.wave_desc, a.n_blocks, WAVES, bid, a.scratch);  // scalar loads, in flight under the gather
    if (a.blk_mode == 2) {   // does this workgroup hold a row of the minibatch?  (every piece of a long row names the row)
        bool mk = false;
        if (ds.w >= 0) { if (ds.x >= 0 && lane == 0) mk = (a.blk_bits[(unsigned)ds.x >> 5] >> (ds.x & 31)) & 1u; }
        else if (PACKED && ds.x >= 0) {
            if (lane % G == 0 && lane / G < ds.y) { const int r = pr.packed[ds.x + lane / G].x; mk = (a.blk_bits[(unsigned)r >> 5] >> (r & 31)) & 1u; }
        }
        // (no barrier of its own -- that cost the launch 10 us at the yelp shape: the waves' flags ride on the workgroup's one
        // existing barrier, blk_append() below)
        const bool any = __ballot(mk) != 0ULL;
        if (lane == 0) blk_flag[w] = any ? 1 : 0;
    }
    auto blk_append = [&]() {   // after the workgroup's barrier, by its first thread
        if (a.blk_mode == 2 && threadIdx.x == 0) {
            int any = 0;
#pragma unroll
            for (int k = 0; k < WAVES; ++k) any |= blk_flag[k];
            if (any) a.blk_list[atomicAdd(a.blk_count, 1)] = (int)blockIdx.x;
        }
    };
    if (a.row_filter && ds.w >= 0 && ds.x >= 0 && !((a.row_filter[(unsigned)ds.x >> 5] >> (ds.x & 31)) & 1u)) ds = make_int4(-1, 0, 0, 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PACKED && ds.w < 0) {
        // packed wave: lane group g owns short row packed[ds.x + g] (<= G nonzeros): no cross-group
        // reduction, no LDS combine, 64/G rows finished per gather instruction
        const int grp = lane / G, sub = lane % G;
        int4 pk = make_int4(-1, 0, 0, 0);
        if (grp < ds.y) pk = pr.packed[ds.x + grp];
        if (a.row_filter && pk.x >= 0 && !((a.row_filter[(unsigned)pk.x >> 5] >> (pk.x & 31)) & 1u)) pk = make_int4(-1, 0, 0, 0);
        const int n = pk.x >= 0 ? pk.z - pk.y : 0;
        int c = 0;
        float av = 0.f;
        if (sub < n) { c = a.col[pk.y + sub]; av = DROP ? drop_val(dc, pk.y + sub, a.val[pk.y + sub]) : a.val[pk.y + sub]; }
        float4 addv = make_float4(0.f, 0.f, 0.f, 0.f), sumv = addv;
        if (pk.x >= 0) {
            const size_t eoff = (size_t)pk.x * D + (size_t)sub * 4;
            // (frontier-filtered launch whose addend IS the filtered operand -- the first backward layer: t1 = A g + g, g zero outside
            //  the frontier -- reads the addend only for frontier rows: the other rows' 512 bytes are zeros by construction.  Same bits.)
            const bool add_row = !(FILT && a.e.add == a.x) || ((a.src_filter[(unsigned)pk.x >> 5] >> (pk.x & 31)) & 1u);
            if (a.e.add && add_row) addv = *reinterpret_cast<const float4 *>(a.e.add + eoff);
            if (a.e.sum_out) sumv = *reinterpret_cast<const float4 *>(a.e.sum_in + eoff);
        }
        constexpr int PU = UNMAX < 8 ? UNMAX : 8;   // (gathers in flight per round: the filtered launch's instantiation keeps four)
        for (int t0 = 0; t0 < ds.z; t0 += PU) {  // ds.z = longest row of this wave
            float4 xv[PU];
            float aw[PU];
#pragma unroll
            for (int j = 0; j < PU; ++j) {
                const int t = t0 + j;
                int cc = __shfl(c, (grp * G + t) & 63, 64);
                float aa = __shfl(av, (grp * G + t) & 63, 64);
                bool ok = t < n;
                if (FILT) ok = ok && ((a.src_filter[(unsigned)cc >> 5] >> (cc & 31)) & 1u);
                cc = ok ? cc : 0;
                aw[j] = ok ? aa : 0.f;
                xv[j] = *reinterpret_cast<const float4 *>(a.x + (unsigned)(cc * D + sub * 4));
            }
#pragma unroll
            for (int j = 0; j < PU; ++j) acc = f4_fma(aw[j], xv[j], acc);
        }
        __syncthreads();  // keep the workgroup's barrier count uniform
        blk_append();
        if (pk.x >= 0 && !(a.dbg & 2)) spmm_epilogue<D>(a.e, pk.x, sub, acc, addv, sumv);
        return;
    }
    if (ds.z > ds.y && !(a.dbg & 1)) {
        acc = spmm_segment<D, UNMAX, DROP, FILT>(a.col, a.val, ds.y, ds.z, a.x, lane, dc, a.src_filter);
    }
    if (lane < G) part[w][lane] = acc;
    __syncthreads();
    blk_append();
    if (ds.w > 0 && !(a.dbg & 2)) {  // leader wave of a row (or of a long row's piece), whole wave
        if (lane < G)
            for (int k = 1; k < ds.w; ++k) acc = f4_add(acc, part[w + k][lane]);
        bool fin = true;
        if (pr.meta.x > 0) fin = piece_arrive(pr, lane, G, acc);
        if (fin && lane < G) {
            float4 addv = make_float4(0.f, 0.f, 0.f, 0.f), sumv = addv;
            const size_t eoff = (size_t)ds.x * D + (size_t)lane * 4;
            const bool add_row = !(FILT && a.e.add == a.x) || ((a.src_filter[(unsigned)ds.x >> 5] >> (ds.x & 31)) & 1u);   // (see the packed path)
            if (a.e.add && add_row) addv = *reinterpret_cast<const float4 *>(a.e.add + eoff);
            if (a.e.sum_out) sumv = *reinterpret_cast<const float4 *>(a.e.sum_in + eoff);
            spmm_epilogue<D>(a.e, ds.x, lane, acc, addv, sumv);
        }
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
    const DropCtx dc = drop_ctx(a);
    const int4 ds = a.wave_desc[(size_t)blockIdx.x * kSpmmWaves + w];
    const int r = ds.x;
    float acc[kGenMaxC];
#pragma unroll
    for (int k = 0; k < kGenMaxC; ++k) acc[k] = 0.f;
    for (int e = ds.y; e < ds.z; ++e) {
        const float *x = a.x + (size_t)a.col[e] * d;
        const float av = drop_val(dc, e, a.val[e]);
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
    {
        const PieceRef pr = piece_ref(a.wave_desc, a.n_blocks, kSpmmWaves, blockIdx.x, a.scratch);
        if (pr.meta.x > 0) {  // piece of a long row: same hand-off as the vector kernel, 4 bytes per lane and k
            float *slot = pr.partials + (size_t)(pr.meta.z + pr.meta.y) * pr.stride;
#pragma unroll
            for (int k = 0; k < kGenMaxC; ++k)
                if (k * 64 + lane < d) __hip_atomic_store(reinterpret_cast<unsigned *>(slot) + k * 64 + lane, __float_as_uint(acc[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(pr.counters + pr.meta.w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = __shfl(ticket, 0, 64);
            if (ticket != pr.meta.x - 1) return;
            if (lane == 0) __hip_atomic_store(pr.counters + pr.meta.w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < kGenMaxC; ++k) {
                if (k * 64 + lane >= d) continue;
                float t = 0.f;
                for (int q = 0; q < pr.meta.x; ++q)
                    t += __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned *>(pr.partials + (size_t)(pr.meta.z + q) * pr.stride) + k * 64 + lane,
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                acc[k] = t;
            }
        }
    }
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
inline hipError_t spmm_launch(const SpmmArgs &a_in, hipStream_t s)
{
    if (a_in.n_rows <= 0) return hipSuccess;
    SpmmArgs a = a_in;
    const bool packed = (a.n_blocks & kSchedPackedFlag) != 0;
    const int W = sched_waves(a.n_blocks);
    if ((a.n_blocks & kSchedLongFlag) && !a.scratch) return hipErrorInvalidValue;  // long rows need their scratch block
    a.n_blocks &= ~(kSchedPackedFlag | kSchedWavesMask | kSchedLongFlag);
    const bool vec = a.d == 32 || a.d == 64 || a.d == 128 || a.d == 256;
    if (!vec || !a.blk_count || !a.blk_list) a.blk_mode = 0;   // (the generic kernel walks the whole schedule)
    const dim3 grid(a.blk_mode == 3 ? std::max(1, std::min(a.n_blocks, a.blk_cap)) : a.n_blocks), block(W * 64);
    static const int variant = RK_TUNE_INT("RK_SPMM_VARIANT", 0);
    static const int dbg = RK_TUNE_INT("RK_SPMM_DEBUG", 0);
    a.dbg = dbg;
    static const int no_filt = RK_TUNE_INT("RK_SPMM_NO_FRONTIER", 0);   // A/B only
    if (no_filt || a.drop_thresh24 || !(a.d == 32 || a.d == 64 || a.d == 128 || a.d == 256)) a.src_filter = nullptr;
    const bool filt = a.src_filter != nullptr;
    static const int no_filt8 = RK_TUNE_INT("RK_SPMM_NO_FILT8", 0);   // A/B only
    (void)no_filt8;
    // The frontier-filtered launch keeps one or two of a segment's 64 entries: four gathers in flight are plenty, and without the
    // other four's registers the kernel fits 64 VGPRs -- eight waves per SIMD instead of six for a launch that is bound by its waves'
    // dependent round trips (descriptor -> columns -> bitmap -> rows), not by bytes.
#define RK_SPMM_CASE(D, UN, WV, MW)                                                              \
    do {                                                                                         \
        if (filt && (WV) <= 8 && !no_filt8) {                                                    \
            if (packed) hipLaunchKernelGGL((spmm_csr_kernel<D, 4, WV, 8, true, false, true>), grid, block, 0, s, a);   \
            else hipLaunchKernelGGL((spmm_csr_kernel<D, 4, WV, 8, false, false, true>), grid, block, 0, s, a);         \
        } else if (filt) {                                                                       \
            if (packed) hipLaunchKernelGGL((spmm_csr_kernel<D, UN, WV, MW, true, false, true>), grid, block, 0, s, a);  \
            else hipLaunchKernelGGL((spmm_csr_kernel<D, UN, WV, MW, false, false, true>), grid, block, 0, s, a);        \
        } else if (packed) hipLaunchKernelGGL((spmm_csr_kernel<D, UN, WV, MW, true>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((spmm_csr_kernel<D, UN, WV, MW, false>), grid, block, 0, s, a);  \
    } while (0)
#define RK_SPMM_DROP(D)                                                                                \
    do {                                                                                               \
        if (W == 4) {                                                                                  \
            if (packed) hipLaunchKernelGGL((spmm_csr_kernel<D, 8, 4, 6, true, true>), grid, block, 0, s, a);  \
            else hipLaunchKernelGGL((spmm_csr_kernel<D, 8, 4, 6, false, true>), grid, block, 0, s, a);        \
        } else {                                                                                       \
            if (packed) hipLaunchKernelGGL((spmm_csr_kernel<D, 8, 8, 6, true, true>), grid, block, 0, s, a);  \
            else hipLaunchKernelGGL((spmm_csr_kernel<D, 8, 8, 6, false, true>), grid, block, 0, s, a);        \
        }                                                                                              \
    } while (0)
    if (a.drop_thresh24 && (a.d == 32 || a.d == 64 || a.d == 128 || a.d == 256)) {
        // graph dropout: instantiated for the two automatic workgroup shapes
        if (W != 8 && W != 4) return hipErrorInvalidValue;
        switch (a.d) {
            case 32: RK_SPMM_DROP(32); break;
            case 64: RK_SPMM_DROP(64); break;
            case 128: RK_SPMM_DROP(128); break;
            default: RK_SPMM_DROP(256); break;
        }
        return hipGetLastError();
    }
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
#undef RK_SPMM_DROP
    return hipGetLastError();
}
