// LDS-resident, d-sliced SpMM for bipartite D^-1/2 A D^-1/2 graphs whose class tables fit a CU's LDS
// (ml1m / Amazon-game sized).  Replaces torch.sparse.mm at recad/model/victim/lightgcn.py:107 (and its
// autograd twin) on the graphs where spmm.h's row-gather kernel sits at the L2 gather rate.
//
// Idea: Y[:, s] = A . X[:, s] for every column slice s independently.  A workgroup owns
// (class half, slice s of S floats, contiguous block of output rows); it stages the WHOLE source-class
// slice -- n_src x S floats, pre-multiplied by D^-1/2 of the source rows -- into LDS once (coalesced), then
// every nonzero is one ds_read_b128 instead of a 256-byte row gather from L2, and the sum is multiplied by
// D^-1/2 of the output row:  y[r] = dinv[r] * sum_c (dinv[c] * x[c])  (A is binary, implicit.py:259-277:
// val = dinv[r]*dinv[c]; the factored form differs from the stored-value form by rounding only, ~1e-7 rel).
// The matrix therefore shrinks to a 16-bit column stream, laid out SELL-style in the order the lanes consume
// it: rows are cut into chunks of <= C nonzeros, chunks sorted by length, 64/LP chunks form a task that one
// wave walks (LP = S/4 lanes per entry, 8 entries per 16-byte stream load); waves pop tasks longest-first
// from an LDS counter; chunk partial sums meet in LDS and a second phase adds each row's chunks in CSR order
// and runs the fused epilogue (spmm.h's, plus layouts).  Every sum has a fixed order => bit-reproducible.
//
// Buffers that are gathered (E_l, the backward's t_l, gprop) live in a SLICED layout: per class block,
// [d/Sc][n_c][Sc] floats (Sc = slice width used when THAT class is the source), so a slice table is one
// contiguous run.  E0 / m / v / light stay row-major (they are the caller's tensors).
#pragma once
#include "common.h"
#include "host/layout.h"

struct LdsDims {
    int U, I, d;
    int lsu, lsi;  // log2 of the users / items block slice width (floats)
};
// float offset of element (node r, column k) of a sliced [U+I, d] buffer
__host__ __device__ __forceinline__ size_t sl_off(const LdsDims &g, int r, int k)
{
    if (r < g.U) return ((((size_t)(k >> g.lsu)) * (size_t)g.U + (size_t)r) << g.lsu) + (size_t)(k & ((1 << g.lsu) - 1));
    return (size_t)g.U * (size_t)g.d + ((((size_t)(k >> g.lsi)) * (size_t)g.I + (size_t)(r - g.U)) << g.lsi) + (size_t)(k & ((1 << g.lsi) - 1));
}

struct LdsEpi {
    // v = dinv[r] * acc (+ add[r])          add: sliced
    const float *add;
    float *y;            // nullable; sliced unless y_rm
    const float *sum_in; // sliced
    float *sum_out;      // sliced unless sum_rm:  sum_out[r] = (sum_in[r] + v) * sum_scale
    float sum_scale;
    int y_rm, sum_rm;
    float *zero1, *zero2;  // sliced, nullable: set to 0 after the addend was read
    int adam;              // Adam on p/m/v with gradient v (sliced, or row-major when adam_rm); shadow (sliced, nullable)
    int adam_rm;           // receives the new p
    float *p, *m, *v, *shadow;
    const float *coef;
    float b1, b2, eps;
    int *state;
    int bump;
    unsigned long long *stamps;  // diagnostic, nullable: 4 wall-clock stamps per workgroup (start, staged, gathered, done)
    // L2-regularisation gradient in closed form: v += creg[0] * cnt[node] * reg_p[r] (cnt = how often the node occurs in the
    // minibatch, reg_p = E0 sliced) -- the BPR kernel then scatters three gradient rows per triplet instead of six
    const int *cnt;
    const float *creg, *reg_p;
    int *zero_cnt;               // nullable: cnt is cleared here (a launch in which nobody reads it)
    // multi-phase launch (spmm_lds_multi_kernel): a later phase of the SAME launch gathers / reads y and sum_out -- they are
    // stored write-through (sc1) and the workgroup's arrival is counted
    int publish;
    // Between two launches of a layer chain the intermediate y needs no natural order: y_staged -- y is written the way its ONLY
    // consumer stages it, row r at the LDS-table position perm[r] of the other half and already multiplied by dinv[r]; x_staged --
    // x was written like that, so staging is a plain copy: one 16-byte load per row instead of three loads (row, permutation,
    // dinv).  The stage phase runs at the 64 B/clk a CU reads from L2 (1.04 us + 40 B/clk of table by the stamps), a third of
    // its bytes were those two words per row.  Same products (dinv[c] * x[c], one rounding each), same bits.  Sliced y only.
    int y_staged, x_staged;
};

// what the plan's header words say, as KERNEL ARGUMENTS (the host knows them: rk_lds_info): a workgroup's only dependent load
// before its staging loads is then its own 64-byte record (LW_*) -- the header -> table -> block descriptor chain of dependent
// loads cost every launch ~1 us before the first staging load was issued
struct LdsHdr {
    int U, I, d, lsu, lsi, wgx_ofs, dinv_ofs, perm0, perm1, mq_ofs;
};
struct LdsRec {   // a workgroup's record, in SGPRs
    int half, slice, rb, grp, row0, n_rows, n_part, n_tasks, task_ofs, dst_ofs, pp_ofs, stream_ofs;
};
__device__ __forceinline__ LdsRec lds_load_rec(const int *__restrict__ plan, int wgx_ofs, int b)
{
    const int4 *p = reinterpret_cast<const int4 *>(plan + wgx_ofs) + (size_t)b * (LW_WORDS / 4);
    const int4 a = p[0], c = p[1], g = p[2];
    LdsRec r;
    r.half = __builtin_amdgcn_readfirstlane(a.x); r.slice = __builtin_amdgcn_readfirstlane(a.y);
    r.rb = __builtin_amdgcn_readfirstlane(a.z); r.grp = __builtin_amdgcn_readfirstlane(a.w);
    r.row0 = __builtin_amdgcn_readfirstlane(c.x); r.n_rows = __builtin_amdgcn_readfirstlane(c.y);
    r.n_part = __builtin_amdgcn_readfirstlane(c.z); r.n_tasks = __builtin_amdgcn_readfirstlane(c.w);
    r.task_ofs = __builtin_amdgcn_readfirstlane(g.x); r.dst_ofs = __builtin_amdgcn_readfirstlane(g.y);
    r.pp_ofs = __builtin_amdgcn_readfirstlane(g.z); r.stream_ofs = __builtin_amdgcn_readfirstlane(g.w);
    return r;
}

struct LdsArgs {
    const int *plan;
    const float *x;  // sliced [N, d]
    LdsHdr h;
    LdsEpi e;
};

typedef float lds_f4n __attribute__((ext_vector_type(4)));
// (through the ext-vector type: two v_pk_add_f32 / v_pk_mul_f32 instead of four scalar operations, the same value per component)
__device__ __forceinline__ float4 f4_scale(float4 a, float s)
{
    const lds_f4n x = {a.x, a.y, a.z, a.w};
    const lds_f4n r = x * s;
    return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ float4 f4_plus(float4 a, float4 b)
{
    const lds_f4n x = {a.x, a.y, a.z, a.w}, y = {b.x, b.y, b.z, b.w};
    const lds_f4n r = x + y;
    return make_float4(r.x, r.y, r.z, r.w);
}

// LDS (address space 3) pointer to a float4: lets an absolute LDS byte address be dereferenced without a base add
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"   // (host pass only: LDS pointers are 32-bit on the device)
typedef const lds_f4n __attribute__((address_space(3))) *lds_f4_ptr;

// ---- in-launch hand-off traffic (multi-phase launch): every byte a later phase reads was stored write-through (sc1) and
// is loaded with sc1 buffer loads (L1 bypassed; cdna_hip_programming.md Guideline 16 R1, MI355X_MICROARCH.md visibility table
// row 1: one lane per storing workgroup adds to an agent-scope counter behind every wave's vmcnt(0) + the workgroup barrier,
// the consumer polls it with an sc1 load, a workgroup barrier, then ONLY sc1 loads of the handed-off bytes).
typedef unsigned lds_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lds_rsrc(const float *p, size_t n_floats)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, p ? (int)(unsigned)(n_floats * 4) : 0, 0x00020000);
}
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t r, size_t float_off)
{
    const lds_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(unsigned)(float_off * 4), 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, size_t float_off, float4 v)
{
    lds_u4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)(unsigned)(float_off * 4), 0, 16);
}

// Workgroup barrier between two phases that only exchange data through LDS: release / acquire fences on the LOCAL address space
// only, so global loads already requested (epilogue operands, the next stream blocks) stay in flight across it -- __syncthreads()
// drains vmcnt.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// epilogue operands of one (row, 4-column piece), requested before the gather so that they are in registers when
// the row's sum is ready
struct LdsRowOps {
    int p0, p1;
    float dr;
    float4 addv, sumv;
    size_t so, ro;
    int rp;   // y_staged: the row's LDS-table position in the half that stages this class (else the row itself)
};

// HAND: a phase of the multi-phase launch -- x / add / sum_in may have been written by OTHER workgroups of this launch (sc1
// loads only), y / sliced sum_out are published write-through when e.publish is set.
template <int LP, bool HAND = false>
__device__ __forceinline__ void lds_body(const float *ax, const LdsEpi &e, const int *__restrict__ plan, const LdsHdr &hd, const LdsRec &rec, float4 *lds,
                                         unsigned long long *hst = nullptr /* HAND, diagnostic: {staged, gathered} wall-clock stamps */)
{
    constexpr int SL = 64 / LP, S = 4 * LP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane / LP, pj = lane % LP;
    const int half = rec.half, slice = rec.slice;
    const int U = hd.U, I = hd.I, d = hd.d;
    const int n_src = half ? U : I, n_dst = half ? I : U;
    const float *dinv = reinterpret_cast<const float *>(plan + hd.dinv_ofs);
    const float *dsrc = dinv + (half ? 0 : U), *ddst = dinv + (half ? U : 0);
    // the source class's slice table: n_src * S contiguous floats
    const float4 *s4 = reinterpret_cast<const float4 *>(ax + (half ? (size_t)0 : (size_t)U * d) + (size_t)slice * n_src * S);
    const int n4 = n_src * LP;
    // LDS row of source row c: sources sorted by degree are dealt round-robin over the 16 / LP bank classes, so the hot
    // columns (an item half the users rated) do not pile up in one class of every lane group
    const int *perm = plan + (half ? hd.perm1 : hd.perm0);
    const int n_tasks = rec.n_tasks;
    const int2 *tasks = reinterpret_cast<const int2 *>(plan + rec.task_ofs);
    const int *dstv = plan + rec.dst_ofs;
    const uint4 *stream = reinterpret_cast<const uint4 *>(plan) + rec.stream_ofs;
    const int n_rows = rec.n_rows, row0 = rec.row0;
    const int *pp = plan + rec.pp_ofs;
    const size_t nd = (size_t)(U + I) * (size_t)d;
    const __amdgpu_buffer_rsrc_t rs_x = lds_rsrc(HAND ? ax : nullptr, nd), rs_add = lds_rsrc(HAND ? e.add : nullptr, nd),
                                 rs_sum = lds_rsrc(HAND && e.sum_out ? e.sum_in : nullptr, nd);
    const size_t x_off = (half ? (size_t)0 : (size_t)U * d) + (size_t)slice * n_src * S;   // first float of the slice table
    const int lso = half ? hd.lsi : hd.lsu;  // slice width (log2) of the OUTPUT class's block
    const size_t cls_base = half ? (size_t)U * d : 0;
    const int node0 = half ? U : 0;
    if (!HAND && e.stamps && tid == 0) e.stamps[blockIdx.x * 4 + 0] = wall_clock64();
    // LDS layout: [slice table][16 / LP zero rows, one per bank class][chunk partials][task descriptors (int2)][queue head]
    // (the table sits at LDS address 0, so an entry's address is its stream word shifted)
    float4 *tab = lds;
    float4 *part = tab + n4 + 16;
    int2 *ltask = reinterpret_cast<int2 *>(part + rec.n_part * LP);
    int *qhead = reinterpret_cast<int *>(ltask + n_tasks);
    if (HAND) { if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) *qhead = 0; }   // (wave-uniform branch: see spmm_lds_multi_kernel)
    else if (tid == 0) *qhead = 0;
    auto row_ops = [&](int i) {
        LdsRowOps o;
        const int lr = i / LP, j = i % LP;
        const int r = row0 + lr;            // class-local output row
        const int k0 = slice * S + 4 * j;   // first of this thread's four columns
        o.so = cls_base + ((((size_t)(k0 >> lso)) * (size_t)n_dst + (size_t)r) << lso) + (size_t)(k0 & ((1 << lso) - 1));
        o.rp = r;
        if (!HAND && e.y_staged) o.rp = (plan + (half ? hd.perm0 : hd.perm1))[r];   // (the other half stages this class)
        o.ro = (size_t)(node0 + r) * d + k0;
        o.p0 = pp[lr]; o.p1 = pp[lr + 1];
        o.dr = ddst[r];
        o.addv = make_float4(0.f, 0.f, 0.f, 0.f); o.sumv = o.addv;
        if (HAND) {
            if (e.add) o.addv = ld16_sc1(rs_add, o.so);
            if (e.sum_out) o.sumv = ld16_sc1(rs_sum, o.so);
        } else {
            if (e.add) o.addv = *reinterpret_cast<const float4 *>(e.add + o.so);
            if (e.sum_out) o.sumv = *reinterpret_cast<const float4 *>(e.sum_in + o.so);
        }
        return o;
    };
    // ---- phase 1: stage the slice table, pre-scaled by dinv of the source rows
    constexpr int UN = 8;
    if (!HAND && e.x_staged) {   // (workgroup-uniform) the producer wrote the table: a copy
        for (int i0 = tid; i0 < n4; i0 += kLdsThreads * UN) {
            float4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * kLdsThreads;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);   // (defined on both paths: left undefined the array went to scratch)
                if (i < n4) v[u] = s4[i];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = i0 + u * kLdsThreads;
                if (i < n4) tab[i] = v[u];
            }
        }
    } else
    for (int i0 = tid; i0 < n4; i0 += kLdsThreads * UN) {
        float4 v[UN];
        float s[UN];
        int pr[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * kLdsThreads;
            if (i < n4) { v[u] = HAND ? ld16_sc1(rs_x, x_off + (size_t)i * 4) : s4[i]; s[u] = dsrc[i / LP]; pr[u] = perm[i / LP]; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * kLdsThreads;
            if (i < n4) tab[pr[u] * LP + i % LP] = f4_scale(v[u], s[u]);
        }
    }
    if (tid < 16) tab[n4 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // the padding entries' rows
    for (int t = tid; t < n_tasks; t += kLdsThreads) ltask[t] = tasks[t];   // (after the staging loads: a load that feeds an LDS write is waited for)
    // this thread's first epilogue row: its operand loads fly under the barrier and the gather.  (Requested AFTER the staging
    // loads: in front of them the compiler drained them -- s_waitcnt vmcnt(0) -- before the first staging load was issued.)
    LdsRowOps ops0{};
    if (tid < n_rows * LP) ops0 = row_ops(tid);
    lds_barrier();   // (LDS-only fences: the operand loads above stay in flight across it)
    if (!HAND && e.stamps && tid == 0) e.stamps[blockIdx.x * 4 + 1] = wall_clock64();
    if (HAND && hst && __builtin_amdgcn_readfirstlane(tid >> 6) == 0) hst[0] = wall_clock64();
    // ---- phase 2: tasks, longest first, popped from an LDS counter; the next task's descriptor, destination and
    // first stream block are requested while the current one is walked
    auto pop = [&]() {
        int v = 0;
        if (lane == 0) v = atomicAdd(qhead, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // byte address of the table row named by the low / high 16 bits of a stream word.  LP == 1: the table sits at LDS
    // address 0 and a row is 16 bytes, so the address is the word's half shifted by 4 -- ONE VALU instruction with SDWA
    // operand selection (the compiler emits v_and / v_bfe + v_lshl_add: the gather loop is VALU-bound, SQ_INSTS_VALU,
    // and the address arithmetic was half of it).
    auto row_lo = [&](unsigned w) -> lds_f4_ptr {
        if (LP == 1) {
            unsigned r;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "s"(4u), "v"(w));
            return (lds_f4_ptr)r;
        }
        return (lds_f4_ptr)(tab + (w & 0xffffu) * LP + pj);
    };
    auto row_hi = [&](unsigned w) -> lds_f4_ptr {
        if (LP == 1) {
            unsigned r;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "s"(4u), "v"(w));
            return (lds_f4_ptr)r;
        }
        return (lds_f4_ptr)(tab + (w >> 16) * LP + pj);
    };
    // (the sums stay ext-vector typed from the LDS read to the partial's store: the adds are v_pk_add_f32, two per table row
    // instead of four v_add_f32 -- the gather phase is bound by VALU issue, not by the LDS array; same order per component)
    auto read8 = [&](lds_f4n (&xv)[8], uint4 c) {
        const unsigned wv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xv[2 * k] = *row_lo(wv[k]);
            xv[2 * k + 1] = *row_hi(wv[k]);
        }
    };
    auto add8 = [&](lds_f4n acc, const lds_f4n (&xv)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = acc + xv[k];
        return acc;
    };
    int t = pop();
    int2 tk = make_int2(0, 0);
    int my_dst = -1;
    const uint4 *st = stream;
    uint4 c0 = make_uint4(0u, 0u, 0u, 0u), c1 = c0;
    if (t < n_tasks) {
        tk = ltask[t];
        st = stream + tk.x + q;
        my_dst = dstv[t * SL + q];
        c0 = st[0];
        if (tk.y > 1) c1 = st[SL];
    }
    while (t < n_tasks) {
        const int tn = pop();
        int2 tkn = make_int2(0, 0);
        int dstn = -1;
        const uint4 *stn = stream;
        uint4 n0 = make_uint4(0u, 0u, 0u, 0u), n1 = n0;
        if (tn < n_tasks) {
            tkn = ltask[tn];
            stn = stream + tkn.x + q;
            dstn = dstv[tn * SL + q];
            n0 = stn[0];
            if (tkn.y > 1) n1 = stn[SL];
        }
        // Two 8-entry blocks per round: all sixteen table reads are issued before the first add, so a wave keeps the LDS
        // pipe fed while it adds (with one block per round LDS-array and VALU time of the sixteen waves added up: the
        // array was busy 55 % of the gather phase).  The stream words of the next round are already in flight.
        lds_f4n acc = {0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 2 <= tk.y; b += 2) {
            uint4 d0 = c0, d1 = c1;
            if (b + 2 < tk.y) d0 = st[(size_t)(b + 2) * SL];
            if (b + 3 < tk.y) d1 = st[(size_t)(b + 3) * SL];
            lds_f4n xa[8], xb[8];
            read8(xa, c0);
            read8(xb, c1);
            acc = add8(acc, xa);
            acc = add8(acc, xb);
            c0 = d0; c1 = d1;
        }
        if (b < tk.y) {
            lds_f4n xa[8];
            read8(xa, c0);
            acc = add8(acc, xa);
        }
        if (my_dst >= 0) part[my_dst * LP + pj] = make_float4(acc.x, acc.y, acc.z, acc.w);
        t = tn; tk = tkn; st = stn; my_dst = dstn; c0 = n0; c1 = n1;
    }
    lds_barrier();
    if (!HAND && e.stamps && tid == 0) e.stamps[blockIdx.x * 4 + 2] = wall_clock64();
    if (HAND && hst && __builtin_amdgcn_readfirstlane(tid >> 6) == 0) hst[1] = wall_clock64();
    // ---- phase 3: per output row, chunk partials in CSR order, dinv of the row, fused epilogue
    for (int i = tid; i < n_rows * LP; i += kLdsThreads) {
        const int j = i % LP;
        const LdsRowOps o = (i == tid) ? ops0 : row_ops(i);
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f), mw = pw, vw = pw;
        if (e.adam) {   // (requested here, not before the gather: twelve registers the gather loop needs)
            const size_t ao = e.adam_rm ? o.ro : o.so;
            pw = *reinterpret_cast<const float4 *>(e.p + ao);
            mw = *reinterpret_cast<const float4 *>(e.m + ao);
            vw = *reinterpret_cast<const float4 *>(e.v + ao);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int p = o.p0;
        for (; p + 8 <= o.p1; p += 8) {   // long rows: eight partials in flight, added in chunk order
            float4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = part[(p + k) * LP + j];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = f4_plus(acc, t[k]);
        }
        for (; p < o.p1; ++p) acc = f4_plus(acc, part[p * LP + j]);
        float4 v = f4_scale(acc, o.dr);
        if (e.add) v = f4_plus(v, o.addv);
        if (e.cnt) {
            const float c = e.creg[0] * (float)e.cnt[node0 + row0 + i / LP];
            const float4 pr = e.adam && !e.adam_rm ? pw : *reinterpret_cast<const float4 *>(e.reg_p + o.so);
            v.x += c * pr.x; v.y += c * pr.y; v.z += c * pr.z; v.w += c * pr.w;
        }
        if (e.zero_cnt && slice == 0 && j == 0) e.zero_cnt[node0 + row0 + i / LP] = 0;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e.zero1) *reinterpret_cast<float4 *>(e.zero1 + o.so) = z;
        if (e.zero2) *reinterpret_cast<float4 *>(e.zero2 + o.so) = z;
        if (HAND && e.publish) {
            // (one descriptor per destination; a null pointer gives zero records: the store is dropped)
            if (e.y) { if (e.y_rm) *reinterpret_cast<float4 *>(e.y + o.ro) = v; else st16_sc1(lds_rsrc(e.y, nd), o.so, v); }
            if (e.sum_out) {
                const float4 sv = f4_scale(f4_plus(o.sumv, v), e.sum_scale);
                if (e.sum_rm) *reinterpret_cast<float4 *>(e.sum_out + o.ro) = sv; else st16_sc1(lds_rsrc(e.sum_out, nd), o.so, sv);
            }
        } else {
            if (e.y) {
                if (e.y_staged) {   // so with the row replaced by its table position
                    const int lr = i / LP, j2 = i % LP, k0 = slice * S + 4 * j2;
                    const size_t sy = o.so + (((size_t)o.rp - (size_t)(row0 + lr)) << lso);
                    (void)k0;
                    *reinterpret_cast<float4 *>(e.y + sy) = f4_scale(v, o.dr);
                }
                else *reinterpret_cast<float4 *>(e.y + (e.y_rm ? o.ro : o.so)) = v;
            }
            if (e.sum_out) *reinterpret_cast<float4 *>(e.sum_out + (e.sum_rm ? o.ro : o.so)) = f4_scale(f4_plus(o.sumv, v), e.sum_scale);
        }
        if (e.adam) {
            const float step_size = e.coef[0], bc2s = e.coef[1];
            const float w1 = (float)(1.0 - (double)e.b1), w2 = (float)(1.0 - (double)e.b2);
            adam_elem(pw.x, mw.x, vw.x, v.x, w1, e.b2, w2, step_size, bc2s, e.eps);
            adam_elem(pw.y, mw.y, vw.y, v.y, w1, e.b2, w2, step_size, bc2s, e.eps);
            adam_elem(pw.z, mw.z, vw.z, v.z, w1, e.b2, w2, step_size, bc2s, e.eps);
            adam_elem(pw.w, mw.w, vw.w, v.w, w1, e.b2, w2, step_size, bc2s, e.eps);
            const size_t ao = e.adam_rm ? o.ro : o.so;
            *reinterpret_cast<float4 *>(e.p + ao) = pw;
            *reinterpret_cast<float4 *>(e.m + ao) = mw;
            *reinterpret_cast<float4 *>(e.v + ao) = vw;
            if (e.shadow) *reinterpret_cast<float4 *>(e.shadow + o.so) = pw;
        }
    }
    if (!HAND && e.stamps) {
        __syncthreads();
        if (tid == 0) e.stamps[blockIdx.x * 4 + 3] = wall_clock64();
    }
}

// LPA / LPB: lanes per entry (= slice width / 4) of the user-row half (gathers the items table) and of the
// item-row half (gathers the users table)
template <int LPA, int LPB>
__global__ __launch_bounds__(kLdsThreads) void spmm_lds_kernel(const LdsArgs a)
{
    extern __shared__ float4 lds_dyn[];
    const int *__restrict__ plan = a.plan;
    if (a.e.bump && blockIdx.x == 0 && threadIdx.x == 0) {
        a.e.state[ST_STEP_BASE] += a.e.bump;
        a.e.state[ST_ADAM_T] += a.e.bump;
    }
    const LdsRec rec = lds_load_rec(plan, a.h.wgx_ofs, blockIdx.x);
    if (rec.half == 0) lds_body<LPA>(a.x, a.e, plan, a.h, rec, lds_dyn);
    else lds_body<LPB>(a.x, a.e, plan, a.h, rec, lds_dyn);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Multi-phase launch: the L propagation layers of a forward (or backward) pass in ONE launch.
//
// Y[:, s] = A . X[:, s] holds per column slice for every layer, so a workgroup (half, s, block) of layer l+1 depends only on
// the workgroups of the same column GROUP (the 2 x n_blk workgroups that produce and consume columns [g G, (g+1) G), G = the
// wider of the two halves' slice widths) of layer l -- not on the grid.  Work items = (phase, half, slice, block); the plan
// deals the groups to up to 8 QUEUES (a group never straddles queues) and lists every queue's items; a queue's items are
// handed out in phase-major order by one agent-scope ticket counter, and an item of phase p starts when its group's arrival
// counter shows that all members finished phase p-1 (that one wait also covers every buffer that is re-used two phases
// later: within a group all hazards are between consecutive phases).
// Deadlock-free WITHOUT assuming that the whole grid is resident: an item only ever waits for items with smaller tickets of
// its own queue, and those have been taken by workgroups that are running.  Placement-independent: the hand-off is
// "payload sc1 + agent-scope counter" (see ld16_sc1 above); workgroup b pulls from queue (b % 8) % n_queues, which keeps a
// group on one XCD under round-robin dispatch -- speed only (the re-staged slice tables are then same-XCD traffic).
// sync words (int32, caller-owned, one launch at a time, zero before the first launch; the last workgroup to leave zeroes
// the counters again, and every train / propagate call's prologue does too -- but never LS_ERR: a timed-out wait stays
// recorded until rk_lightgcn_sync_status reads (and then clears) it):
enum { LS_HEAD = 0 /* 8 queue heads, one per 128-byte line */, LS_ARRIVE = 8 * 32 /* per group, one line each */,
       LS_MAX_GROUPS = 64, LS_DONE = LS_ARRIVE + LS_MAX_GROUPS * 32, LS_ERR = LS_DONE + 32, LS_WORDS = LS_ERR + 32 };
// plan words at LP_MQ_OFS: {n_queues, n_groups, G, 0}, then n_queues x {n_items, first int4 of its list (in int4 units from
// the plan's start)}, then n_groups x members; the lists hold int4 {half, slice, block, group}
static constexpr int kLdsMaxPhases = 4;
static constexpr unsigned kLdsSpinLimit = 1u << 21;   // polls before a waiter gives up and sets LS_ERR (a poll is a ~1.5 us round trip: ~3 s)

struct LdsPhase {
    const float *x;
    LdsEpi e;
};
struct LdsMultiArgs {
    const int *plan;
    int *sync;
    LdsHdr h;
    int n_phases, bc_ofs;   // bc_ofs: byte offset of a 16-byte broadcast slot behind the body's LDS layout
    unsigned long long *stamps;   // diagnostic, nullable: 8 wall-clock stamps per (workgroup, item): ticket known, wait over, staged, gathered, rows done, drained
    LdsPhase ph[kLdsMaxPhases];
};

template <int LPA, int LPB>
__global__ __launch_bounds__(kLdsThreads) void spmm_lds_multi_kernel(const LdsMultiArgs a)
{
    extern __shared__ float4 lds_dyn[];
    const int *__restrict__ plan = a.plan;
    const int tid = threadIdx.x;
    const LdsEpi &el = a.ph[a.n_phases - 1].e;
    if (el.bump && blockIdx.x == 0 && tid == 0) {
        el.state[ST_STEP_BASE] += el.bump;
        el.state[ST_ADAM_T] += el.bump;
    }
    const int *mq = plan + a.h.mq_ofs;
    const int n_queues = mq[0], n_groups = mq[1];
    const int q = (int)(blockIdx.x & 7) % n_queues;
    const int n_items = mq[4 + 2 * q];
    const int4 *list = reinterpret_cast<const int4 *>(plan) + mq[4 + 2 * q + 1];
    const int *members = mq + 4 + 2 * n_queues;
    volatile int *bc = reinterpret_cast<volatile int *>(reinterpret_cast<char *>(lds_dyn) + a.bc_ofs);
    int *head = a.sync + LS_HEAD + q * 32;
    const int total = n_items * a.n_phases;
#ifdef RK_LDS_DEBUG   // progress markers behind the sync words (debug builds; the probe over-allocates): {stage, ticket, iterations, spins}
#define LDS_MARK(k, v) do { if (tid == 0) __hip_atomic_store(a.sync + 2560 + blockIdx.x * 4 + (k), (int)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define LDS_MARK(k, v) do { } while (0)
#endif
    // Everything one "leader" does around the workgroup barriers is done by the WHOLE first wave under a wave-uniform
    // (scalar) branch, never by `if (tid == 0)`: with a single-lane branch at the bottom AND the top of this loop the
    // compiler rotated the loop so that lane 0 left it alone (arrival add, next ticket, LDS store) while lanes 1-63 of its
    // wave went on to the barrier -- s_barrier counts waves, not lanes, so the workgroup read a stale ticket and never
    // terminated (ROCm 7.2, gfx950).  The counters advance in units of 64: lane 0 of that wave adds 64, lanes 1-63 add 0,
    // and the first lane's return value is the ticket.  Correct WITHOUT relying on the compiler's atomic optimizer: folded
    // (one atomic of 64 per wave, what ROCm 7.2 emits) or not (63 no-op adds interleaving with other workgroups' adds of
    // 64), lane 0 always receives a distinct multiple of 64.
    const bool w0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const int one64 = (tid & 63) == 0 ? 64 : 0;
    int iters = 0;
    int t_next = 0;   // (first wave) the NEXT ticket, drawn while this item's stores drain
    if (w0) t_next = __hip_atomic_fetch_add(head, one64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        LDS_MARK(0, 1); LDS_MARK(2, iters); ++iters;
        if (w0) bc[0] = __builtin_amdgcn_readfirstlane(t_next) >> 6;
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(bc[0]);   // uniform: the phase's arguments below are scalar loads
        LDS_MARK(0, 2); LDS_MARK(1, t);
        if (t >= total) break;
        const int phase = t / n_items;
        const int4 wg = list[t - phase * n_items];   // {record index, -, -, group}
        const LdsRec rec = lds_load_rec(plan, a.h.wgx_ofs, __builtin_amdgcn_readfirstlane(wg.x));   // (in flight under the poll)
        int *arrive = a.sync + LS_ARRIVE + wg.w * 32;
        if (phase > 0 && w0) {
            // every member of the group has finished the previous phase (its stores drained before its add)
            const int need = members[wg.w] * phase * 64;
            unsigned spins = 0;
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
                LDS_MARK(3, spins);
                if (++spins > kLdsSpinLimit) { __hip_atomic_store(a.sync + LS_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
        }
#ifdef RK_TUNING   // per-item wall-clock stamps (tuning builds only; written by the whole first wave: no single-lane branch, see above)
        unsigned long long *st = a.stamps ? a.stamps + ((size_t)blockIdx.x * kLdsMaxPhases + (size_t)(iters - 1)) * 8 : nullptr;
#define LDS_STAMP(k) do { if (st && w0) st[k] = wall_clock64(); } while (0)
        if (st && w0) st[6] = (unsigned long long)t;
#else
        unsigned long long *const st = nullptr;
#define LDS_STAMP(k) do { } while (0)
#endif
        LDS_STAMP(0);
        __syncthreads();   // after the poll, before EVERY load of handed-off bytes; also: bc has been read by everybody
        LDS_STAMP(1);
        LDS_MARK(0, 3);
        const LdsPhase &ph = a.ph[phase];   // (kernel-argument memory: uniform loads at a uniform offset, no copy)
        if (rec.half == 0) lds_body<LPA, true>(ph.x, ph.e, plan, a.h, rec, lds_dyn, st ? st + 2 : nullptr);
        else lds_body<LPB, true>(ph.x, ph.e, plan, a.h, rec, lds_dyn, st ? st + 2 : nullptr);
        LDS_STAMP(4);
        LDS_MARK(0, 4);
        if (w0) t_next = __hip_atomic_fetch_add(head, one64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // in flight under the drain
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores ...
        __syncthreads();                                    // ... before the ONE wave that signals for all of them
        LDS_STAMP(5);
        LDS_MARK(0, 5);
        if (w0) __hip_atomic_fetch_add(arrive, one64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    LDS_MARK(0, 6);
    // leave the sync words zero for the next launch: the last workgroup out (nobody reads or adds after its own done-add)
    if (w0) bc[1] = __builtin_amdgcn_readfirstlane(__hip_atomic_fetch_add(a.sync + LS_DONE, one64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 6;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(bc[1]) == (int)gridDim.x - 1) {
        if (tid < 8) __hip_atomic_store(a.sync + LS_HEAD + tid * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < n_groups) __hip_atomic_store(a.sync + LS_ARRIVE + tid * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) __hip_atomic_store(a.sync + LS_DONE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// row-major [N, d] <-> sliced; one float4 per thread
struct LdsPackJob {
    float *rm[3], *sl[3];   // up to three (row-major, sliced) pairs converted by one launch
    int n;
    float *zero[2];         // nullable: [N, d] buffers cleared by the same launch (a train call's scatter targets)
    int *zero_i;            // nullable: int32[N] cleared too (incidence counts)
    int *zero_sync;         // nullable: the multi-phase launch's sync words: the counters [0, LS_ERR) are cleared (a call starts from clean counters whatever came before)
};
static __global__ void lds_pack_kernel(LdsDims g, LdsPackJob job, int to_sliced)
{
    const int d4 = g.d / 4;
    const long long n = (long long)(g.U + g.I) * d4;
    if (job.zero_sync && blockIdx.x == 0)
        for (int i = threadIdx.x; i < LS_ERR; i += blockDim.x) job.zero_sync[i] = 0;   // heads / arrivals / done only: LS_ERR is sticky until rk_lightgcn_sync_status has read it
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / d4), k = (int)(i % d4) * 4;
        const size_t so = sl_off(g, r, k), ro = (size_t)r * g.d + k;
        if (job.zero[0]) *reinterpret_cast<float4 *>(job.zero[0] + ro) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (job.zero[1]) *reinterpret_cast<float4 *>(job.zero[1] + ro) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (job.zero_i && k == 0) job.zero_i[r] = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q >= job.n) break;
            if (to_sliced) *reinterpret_cast<float4 *>(job.sl[q] + so) = *reinterpret_cast<const float4 *>(job.rm[q] + ro);
            else *reinterpret_cast<float4 *>(job.rm[q] + ro) = *reinterpret_cast<const float4 *>(job.sl[q] + so);
        }
    }
}

// Host-side description of an uploaded plan (what the launch needs without reading the device buffer)
struct LdsInfo {
    int n_wg, lds_bytes, lpa, lpb, U, I, d, lsu, lsi;
    int wgx_ofs, dinv_ofs, perm0, perm1, mq_ofs;
    LdsHdr hdr() const { return LdsHdr{U, I, d, lsu, lsi, wgx_ofs, dinv_ofs, perm0, perm1, mq_ofs}; }
};

inline hipError_t spmm_lds_launch(const LdsInfo &info, const LdsArgs &a, hipStream_t s)
{
    const dim3 grid(info.n_wg), block(kLdsThreads);
#define RK_LDS_CASE(A, B)                                                                                                  \
    do {                                                                                                                    \
        static RkPerDeviceOnce attr_once;                                                                                   \
        int attr_dev_;                                                                                                      \
        if (attr_once.need(&attr_dev_)) {                                                                                   \
            /* the gather addresses table rows by ABSOLUTE LDS address (table at 0): the kernel must own no static LDS */  \
            hipFuncAttributes fa_;                                                                                          \
            hipError_t e_ = hipFuncGetAttributes(&fa_, reinterpret_cast<const void *>(&spmm_lds_kernel<A, B>));            \
            if (e_ != hipSuccess) return e_;                                                                                \
            if (fa_.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;                                              \
            e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_lds_kernel<A, B>),                      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMaxBytes - 64);             \
            if (e_ != hipSuccess) return e_;                                                                                \
            attr_once.done(attr_dev_);                                                                                      \
        }                                                                                                                   \
        hipLaunchKernelGGL((spmm_lds_kernel<A, B>), grid, block, (size_t)info.lds_bytes, s, a);                              \
    } while (0)
    if (info.lpa == 2 && info.lpb == 1) RK_LDS_CASE(2, 1);
    else if (info.lpa == 1 && info.lpb == 2) RK_LDS_CASE(1, 2);
    else if (info.lpa == 1 && info.lpb == 1) RK_LDS_CASE(1, 1);
    else if (info.lpa == 2 && info.lpb == 2) RK_LDS_CASE(2, 2);
    else if (info.lpa == 4 && info.lpb == 4) RK_LDS_CASE(4, 4);
    else if (info.lpa == 4 && info.lpb == 2) RK_LDS_CASE(4, 2);
    else if (info.lpa == 2 && info.lpb == 4) RK_LDS_CASE(2, 4);
    else return hipErrorInvalidValue;
#undef RK_LDS_CASE
    return hipGetLastError();
}

// n_phases <= kLdsMaxPhases phases in one launch; `a.ph[*]`, `a.plan`, `a.sync`, `a.n_phases` filled by the caller
inline hipError_t spmm_lds_multi_launch(const LdsInfo &info, LdsMultiArgs &a, hipStream_t s)
{
    const dim3 grid(info.n_wg), block(kLdsThreads);
    a.bc_ofs = (info.lds_bytes + 15) & ~15;
    const size_t lds = (size_t)a.bc_ofs + 16;
    if (lds > (size_t)(kLdsMaxBytes - 64)) return hipErrorInvalidValue;
#define RK_LDS_MCASE(A, B)                                                                                                 \
    do {                                                                                                                    \
        static RkPerDeviceOnce attr_once;                                                                                   \
        int attr_dev_;                                                                                                      \
        if (attr_once.need(&attr_dev_)) {                                                                                   \
            hipFuncAttributes fa_;                                                                                          \
            hipError_t e_ = hipFuncGetAttributes(&fa_, reinterpret_cast<const void *>(&spmm_lds_multi_kernel<A, B>));      \
            if (e_ != hipSuccess) return e_;                                                                                \
            if (fa_.sharedSizeBytes != 0) return hipErrorInvalidConfiguration;   /* table at LDS address 0 */               \
            e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&spmm_lds_multi_kernel<A, B>),                          \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMaxBytes - 64);                        \
            if (e_ != hipSuccess) return e_;                                                                                \
            attr_once.done(attr_dev_);                                                                                      \
        }                                                                                                                   \
        hipLaunchKernelGGL((spmm_lds_multi_kernel<A, B>), grid, block, lds, s, a);                                           \
    } while (0)
    if (info.lpa == 2 && info.lpb == 1) RK_LDS_MCASE(2, 1);
    else if (info.lpa == 1 && info.lpb == 2) RK_LDS_MCASE(1, 2);
    else if (info.lpa == 1 && info.lpb == 1) RK_LDS_MCASE(1, 1);
    else if (info.lpa == 2 && info.lpb == 2) RK_LDS_MCASE(2, 2);
    else if (info.lpa == 4 && info.lpb == 4) RK_LDS_MCASE(4, 4);
    else if (info.lpa == 4 && info.lpb == 2) RK_LDS_MCASE(4, 2);
    else if (info.lpa == 2 && info.lpb == 4) RK_LDS_MCASE(2, 4);
    else return hipErrorInvalidValue;
#undef RK_LDS_MCASE
    return hipGetLastError();
}
