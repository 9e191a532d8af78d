// HOST-ONLY builder of the LDS-resident SpMM's plan (spmm_lds.h): pure C++ (no HIP call, no HIP header), so that it also
// builds under -fsanitize=address,undefined and -fsanitize=thread (make -C recad_amd/csrc host-asan host-tsan;
// tests/test_host_sanitizers.py) -- it runs a std::thread pool and sits on the perturb-retrain loop (one plan per injected graph).
#pragma once
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

#include "../../../include/recad_hip.h"
#include "layout.h"

struct rk_lds_plan {
    std::vector<int32_t> words;
    rk_lds_info info;
};

namespace rk_plan_detail {

struct Chunk {
    int32_t padded, len, e_begin, pidx;
};

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// contiguous row blocks with about equal nonzero counts
static std::vector<int32_t> balanced_blocks(const int32_t *rp, int row_lo, int row_hi, int n_blk)
{
    std::vector<int32_t> b((size_t)n_blk + 1, row_hi);
    b[0] = row_lo;
    const long long base = rp[row_lo], total = (long long)rp[row_hi] - base;
    int r = row_lo;
    for (int k = 1; k < n_blk; ++k) {
        const long long want = total * k / n_blk;
        while (r < row_hi && (long long)rp[r] - base < want) ++r;
        // keep at least one row per block on both sides
        r = std::max(r, b[(size_t)k - 1] + 1);
        r = std::min(r, row_hi - (n_blk - k));
        b[(size_t)k] = r;
    }
    return b;
}

// ds_read_b128 services a wave in four groups of 16 lanes (MI355X_MICROARCH.md, LDS table); only lanes of one group can
// conflict, and they do when their 16-byte pieces share a bank quad = (address / 16) mod 16.
static const int kB128Group[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

// Scratch of colour_group / build_block, one per builder thread and reused across groups and blocks: the builder used to make
// ~5 000 small allocations per row block, and concurrent malloc / free from the pool's threads serialised the "parallel" section
// (measured: 16 blocks of 2 ms each took 33 ms on 8 threads; 8 threads of allocation-only work ran no faster than one).
struct ColourEdge { int u, v, i, c; };
struct PlanScratch {
    std::vector<ColourEdge> es;
    std::vector<int> atL, atR, ccount, path;
    std::vector<unsigned long long> useL, useR;
    std::vector<int> cls, pos, len;          // [K][stride] classes / positions of a lane group's chunks, their lengths
    std::vector<int> slot_pos, slot_len;     // [SL][stride] positions of every slot of the group
    std::vector<Chunk> chunks, placed;
};

// Conflict-free order of the entries of the K chunks that share a lane group.  Table row c occupies bank class c mod K
// (K = 16 / LP rows cover the 64 banks once), so a wave-instruction is conflict-free iff the K chunks read K different
// classes.  That is an edge colouring of the bipartite multigraph chunks x classes (one edge per entry, colour =
// position in the chunk's walk): Delta = max(longest chunk, largest class count) colours always suffice (Koenig); built
// with alternating-path flips.  cls[j][i] = class of entry i of chunk j; pos[j][i] receives its position.  Returns Delta.
// cls / pos: n rows of `stride` ints; row j holds len[j] entries
static int colour_group(const int *cls, const int *len, int n, int stride, int K, int *pos, PlanScratch &ws)
{
    std::vector<int> &ccount = ws.ccount;
    ccount.assign((size_t)K, 0);
    int delta = 0;
    for (int j = 0; j < n; ++j) {
        delta = std::max(delta, len[j]);
        for (int i = 0; i < len[j]; ++i) ++ccount[(size_t)cls[(size_t)j * stride + i]];
    }
    for (int v = 0; v < K; ++v) delta = std::max(delta, ccount[(size_t)v]);
    if (delta == 0) return 0;
    typedef ColourEdge Edge;
    std::vector<Edge> &es = ws.es;
    es.clear();
    const size_t D = (size_t)delta, W = (D + 63) / 64;
    std::vector<int> &atL = ws.atL, &atR = ws.atR;
    atL.assign((size_t)n * D, -1);
    atR.assign((size_t)K * D, -1);
    // used-colour bitmaps per node (bits >= delta preset): the first free colour is one ctz away
    std::vector<unsigned long long> &useL = ws.useL, &useR = ws.useR;
    useL.assign((size_t)n * W, 0ULL);
    useR.assign((size_t)K * W, 0ULL);
    auto preset = [&](std::vector<unsigned long long> &m, size_t nodes) {
        for (size_t q = 0; q < nodes; ++q)
            for (size_t c = D; c < W * 64; ++c) m[q * W + c / 64] |= 1ULL << (c % 64);
    };
    preset(useL, (size_t)n); preset(useR, (size_t)K);
    auto first_free = [&](const unsigned long long *a, const unsigned long long *b) {   // first colour free in a (and b)
        for (size_t q = 0; q < W; ++q) {
            const unsigned long long m = ~(a[q] | (b ? b[q] : 0ULL));
            if (m) return (int)(q * 64 + (size_t)__builtin_ctzll(m));
        }
        return -1;
    };
    auto set_bit = [&](std::vector<unsigned long long> &m, size_t node, int c, bool on) {
        if (on) m[node * W + (size_t)c / 64] |= 1ULL << (c % 64); else m[node * W + (size_t)c / 64] &= ~(1ULL << (c % 64));
    };
    std::vector<int> &path = ws.path;
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < len[j]; ++i) {
            const int u = j, v = cls[(size_t)j * stride + i];
            int a = first_free(&useL[(size_t)u * W], &useR[(size_t)v * W]);   // free at both ends: no flip needed
            if (a < 0) {
                a = first_free(&useL[(size_t)u * W], nullptr);
                const int b = first_free(&useR[(size_t)v * W], nullptr);
                // a is taken at v: flip the a/b alternating path that starts there (it cannot reach u, which has no a edge)
                path.clear();
                int node = v, c = a, o = b;
                bool right = true;
                for (;;) {
                    const int eid = right ? atR[(size_t)node * D + (size_t)c] : atL[(size_t)node * D + (size_t)c];
                    if (eid < 0) break;
                    path.push_back(eid);
                    node = right ? es[(size_t)eid].u : es[(size_t)eid].v;
                    right = !right;
                    std::swap(c, o);
                }
                for (int eid : path) {
                    Edge &e = es[(size_t)eid];
                    atL[(size_t)e.u * D + (size_t)e.c] = -1; set_bit(useL, (size_t)e.u, e.c, false);
                    atR[(size_t)e.v * D + (size_t)e.c] = -1; set_bit(useR, (size_t)e.v, e.c, false);
                }
                for (int eid : path) {
                    Edge &e = es[(size_t)eid];
                    e.c = (e.c == a) ? b : a;
                    atL[(size_t)e.u * D + (size_t)e.c] = eid; set_bit(useL, (size_t)e.u, e.c, true);
                    atR[(size_t)e.v * D + (size_t)e.c] = eid; set_bit(useR, (size_t)e.v, e.c, true);
                }
            }
            const int eid = (int)es.size();
            es.push_back({u, v, i, a});
            atL[(size_t)u * D + (size_t)a] = eid; set_bit(useL, (size_t)u, a, true);
            atR[(size_t)v * D + (size_t)a] = eid; set_bit(useR, (size_t)v, a, true);
        }
    }
    for (const Edge &e : es) pos[(size_t)e.u * stride + (size_t)e.i] = e.c;
    return delta;
}

// The pool's threads live for a few milliseconds.  Measured on this image's hosts: the scheduler leaves such short-lived
// threads on the CPU that created them, and the "parallel" section ran serially (16 row blocks of 1.7 ms each: 28 ms on 8
// threads, 16 x a 7 ms spin loop: 122 ms).  So every worker places ITSELF on its own CPU of the caller's affinity mask (never
// outside it; the caller's own thread and CPU are left alone).  With it the section takes 7-10 ms.
// worker_cpus(): the CPUs the caller may use, without the one it is running on (empty: leave the workers where they are);
// place_self(cpu): first statement of a worker.  (The caller must not set a worker's affinity from outside: a worker that
// has already finished has thread id 0, and the call would then pin the CALLER.)
static std::vector<int> worker_cpus()
{
    std::vector<int> cpus;
#if defined(__linux__)
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return cpus;
    const int self = sched_getcpu();
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &allowed) && c != self) cpus.push_back(c);
    // start behind the caller's own CPU: several callers (one process per GPU building their plans at the same time, concurrent
    // victims) then take different neighbours instead of all piling onto the first CPUs of the mask
    size_t first = 0;
    while (first < cpus.size() && cpus[first] < self) ++first;
    std::rotate(cpus.begin(), cpus.begin() + (long)(first % std::max<size_t>(cpus.size(), 1)), cpus.end());
#endif
    return cpus;
}
static void place_self(int cpu)
{
#if defined(__linux__)
    if (cpu < 0) return;
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(cpu, &one);
    sched_setaffinity(0, sizeof(one), &one);   // (the calling thread; best effort: a refusal leaves it where it is)
#else
    (void)cpu;
#endif
}

struct HalfPlan {
    int S = 0, lp = 0, n_slices = 0, n_blk = 0, chunk = 0;
    std::vector<int32_t> bounds;
    int lds_bytes = 0;
};

// partial-sum slots the largest block of this half needs with chunk cap C
static int max_partials(const int32_t *rp, const std::vector<int32_t> &bounds, int C)
{
    int best = 0;
    for (size_t k = 0; k + 1 < bounds.size(); ++k) {
        int n = 0;
        for (int r = bounds[k]; r < bounds[k + 1]; ++r) n += (rp[r + 1] - rp[r] + C - 1) / C;
        best = std::max(best, n);
    }
    return best;
}

static bool choose_half(const int32_t *rp, int row_lo, int row_hi, int n_src, int dim, int n_cu_half, int force_s, int force_c, HalfPlan *hp)
{
    // 4-float slices first: one lane per entry lets the kernel form an LDS address with one SDWA shift; 8-float slices
    // (two lanes per entry, half the stream traffic) measured the same before that and are kept as RK_LDS_SA/SB=8
    static const int kS[] = {4, 8};
    static const int kC[] = {64, 96, 128, 256, 512};   // 64: 10.8 us per ml1m launch against 12.1 (32) and 11.4 (128)
    const int limit = kLdsMaxBytes - 256;
    for (int S : kS) {
        if (force_s && S != force_s) continue;
        if (dim % S) continue;
        const int n_slices = dim / S;
        int n_blk = std::max(1, n_cu_half / n_slices);
        n_blk = std::min(n_blk, std::max(1, row_hi - row_lo));
        const std::vector<int32_t> bounds = balanced_blocks(rp, row_lo, row_hi, n_blk);
        const long long table = (long long)(n_src + 64 / S) * S * 4;   // + one zero row per bank class
        if (table >= limit) continue;
        for (int C : kC) {
            if (force_c && C != force_c) continue;
            // every chunk gets one partial slot per piece of its group
            static const int sub2 = RK_TUNE_INT("RK_LDS_SUB", 0);
            const int mp = std::max(1, max_partials(rp, bounds, C)) * (sub2 == 2 ? 2 : 1);
            const int SL = 256 / S;                                        // chunks per task
            const long long task_bytes = (((mp + SL - 1) / SL + 1) / 2 + 1) * 16;   // descriptors staged in front of the table
            const long long need = task_bytes + table + (long long)mp * S * 4;
            if (need > limit) continue;
            hp->S = S; hp->lp = S / 4; hp->n_slices = n_slices; hp->n_blk = n_blk; hp->chunk = C; hp->bounds = bounds;
            hp->lds_bytes = (int)need;
            return true;
        }
    }
    return false;
}

}  // namespace rk_plan_detail
using namespace rk_plan_detail;

// Host-only builder (no HIP call): every array is host memory.  *n_words == 0 on return: the graph does not qualify
// (not bipartite / not the normalised binary adjacency / a class table does not fit a CU's LDS) -- use spmm.h's kernel.
inline int lds_plan_build_host_body(int32_t n_users, int32_t n_items, const int32_t *rowptr, const int32_t *col, const float *val,
                                    int32_t dim, int32_t n_cu, rk_lds_plan_t *out, int64_t *n_words, rk_lds_info *info);
// The C-ABI entry points call THIS: no C++ exception (std::bad_alloc of a builder vector, std::system_error of the thread
// pool) may unwind through extern "C" -- the perturb-retrain loop calls the builder once per injected graph, and a transient
// resource limit must come back as RK_E*, not std::terminate.
inline int lds_plan_build_host_impl(int32_t n_users, int32_t n_items, const int32_t *rowptr, const int32_t *col, const float *val,
                                    int32_t dim, int32_t n_cu, rk_lds_plan_t *out, int64_t *n_words, rk_lds_info *info)
{
    try {
        return lds_plan_build_host_body(n_users, n_items, rowptr, col, val, dim, n_cu, out, n_words, info);
    } catch (const std::bad_alloc &) {
        if (out) *out = nullptr;
        if (n_words) *n_words = 0;
        RK_FAIL(RK_ENOMEM, "rk_lds_plan_build_host: out of host memory");
    } catch (const std::exception &e) {
        if (out) *out = nullptr;
        if (n_words) *n_words = 0;
        RK_FAIL(RK_EINVAL, "rk_lds_plan_build_host: %s", e.what());
    }
}
inline int lds_plan_build_host_body(int32_t n_users, int32_t n_items, const int32_t *rowptr, const int32_t *col, const float *val,
                                    int32_t dim, int32_t n_cu, rk_lds_plan_t *out, int64_t *n_words, rk_lds_info *info)
{
    if (n_users <= 0 || n_items <= 0 || !rowptr || !col || dim <= 0 || !out || !n_words || !info)
        RK_FAIL(RK_EINVAL, "rk_lds_plan_build_host: bad arguments");
    *out = nullptr;
    *n_words = 0;
    memset(info, 0, sizeof(*info));
    const int U = n_users, I = n_items, N = U + I;
    if (dim % 4 || dim > 256 || n_cu < 2) return RK_OK;
    if (U + 17 > 65535 || I + 17 > 65535) return RK_OK;  // 16-bit column stream
    const int32_t *rp = rowptr;
    // bipartite structure, binary normalised values
    std::vector<float> dinv((size_t)N);
    for (int r = 0; r < N; ++r) {
        const int deg = rp[r + 1] - rp[r];
        dinv[(size_t)r] = deg > 0 ? (float)(1.0 / std::sqrt((double)deg)) : 0.f;
    }
    if ((long long)rp[U] * 2 != (long long)rp[N]) return RK_OK;
    for (int r = 0; r < N; ++r) {
        const int lo = r < U ? U : 0, hi = r < U ? N : U;
        for (int e = rp[r]; e < rp[r + 1]; ++e) {
            const int c = col[e];
            if (c < lo || c >= hi) return RK_OK;
            if (val) {
                const float want = dinv[(size_t)r] * dinv[(size_t)c];
                if (std::fabs(val[e] - want) > 2e-6f * std::fabs(want)) return RK_OK;
            }
        }
    }
    static const int force_sa = RK_TUNE_INT("RK_LDS_SA", 0);   // tuning: slice width of the items table
    static const int force_sb = RK_TUNE_INT("RK_LDS_SB", 0);   // ... of the users table
    static const int force_c = RK_TUNE_INT("RK_LDS_CHUNK", 0);
    HalfPlan hp[2];
    // half 0: user rows gather the items table; half 1: item rows gather the users table
    if (!choose_half(rp, 0, U, I, dim, n_cu / 2, force_sa, force_c, &hp[0])) return RK_OK;
    if (!choose_half(rp, U, N, U, dim, n_cu - n_cu / 2, force_sb, force_c, &hp[1])) return RK_OK;

    std::unique_ptr<rk_lds_plan> pl_owner(new rk_lds_plan());   // freed if anything below throws
    rk_lds_plan *pl = pl_owner.get();
    std::vector<int32_t> &w = pl->words;
    const int n_wg = hp[0].n_slices * hp[0].n_blk + hp[1].n_slices * hp[1].n_blk;
    w.assign(LP_HDR_WORDS, 0);
    w[LP_MAGIC] = kLdsMagic; w[LP_NWG] = n_wg; w[LP_U] = U; w[LP_I] = I; w[LP_D] = dim;
    // layout slice widths: the items block is sliced the way half 0 gathers it, the users block the way half 1 does
    const int lsi = ilog2(hp[0].S), lsu = ilog2(hp[1].S);
    w[LP_LSU] = lsu; w[LP_LSI] = lsi; w[LP_NBLK0] = hp[0].n_blk; w[LP_NBLK1] = hp[1].n_blk;
    w[LP_LDS_BYTES] = std::max(hp[0].lds_bytes, hp[1].lds_bytes);
    w[LP_CHUNK] = hp[0].chunk | (hp[1].chunk << 16);
    // ---- workgroup table, XCD-aware: workgroup b runs on XCD b % 8 (round-robin dispatch; speed only).  All
    // workgroups of an 8-float column range share an XCD, so the slice tables a launch writes are re-read on the XCD
    // that wrote them and each L2 sees one copy of the column stream.
    {
        std::vector<std::vector<int32_t>> queue(8);
        for (int h = 0; h < 2; ++h)
            for (int rb = 0; rb < hp[h].n_blk; ++rb)
                for (int s = 0; s < hp[h].n_slices; ++s) {
                    const int col0 = s * hp[h].S;
                    static const int map_rows = RK_TUNE_INT("RK_LDS_MAP", 0);   // tuning: 1 = a row block's slices share an XCD
                    std::vector<int32_t> &qv = queue[(size_t)(map_rows ? (rb + h * 4) % 8 : (col0 / 8) % 8)];
                    qv.insert(qv.end(), {h, s, rb, 0});
                }
        w[LP_WG_OFS] = (int32_t)w.size();
        size_t taken[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < n_wg; ++b) {
            int x = b % 8;
            if (taken[x] * 4 >= queue[(size_t)x].size()) {  // this XCD's queue is empty: steal from the fullest
                size_t best = 0;
                for (int y = 0; y < 8; ++y) {
                    const size_t left = queue[(size_t)y].size() / 4 - taken[y];
                    if (left > best) { best = left; x = y; }
                }
            }
            const int32_t *src = &queue[(size_t)x][taken[x] * 4];
            w.insert(w.end(), src, src + 4);
            ++taken[x];
        }
    }
    // ---- work-item queues of the multi-phase launch (spmm_lds.h, spmm_lds_multi_kernel): column groups of G floats (the
    // wider slice width) are dealt to min(8, n_groups) queues, a group never straddles queues; inside a queue the items are
    // interleaved by row block so that both halves of every group advance together.
    {
        const int G = std::max(hp[0].S, hp[1].S), n_groups = dim / G, n_queues = std::min(8, n_groups);
        while (w.size() & 3) w.push_back(0);
        w[LP_MQ_OFS] = (int32_t)w.size();
        const size_t hdr = w.size();
        w.resize(hdr + 4 + 2 * (size_t)n_queues + (size_t)n_groups, 0);
        w[hdr] = n_queues; w[hdr + 1] = n_groups; w[hdr + 2] = G;
        for (int g = 0; g < n_groups; ++g)
            w[hdr + 4 + 2 * (size_t)n_queues + (size_t)g] = (G / hp[0].S) * hp[0].n_blk + (G / hp[1].S) * hp[1].n_blk;
        for (int q = 0; q < n_queues; ++q) {
            while (w.size() & 3) w.push_back(0);
            const size_t first = w.size();
            int n_items = 0;
            const int rounds = std::max(hp[0].n_blk, hp[1].n_blk);
            for (int k = 0; k < rounds; ++k)
                for (int g = q; g < n_groups; g += n_queues)
                    for (int h = 0; h < 2; ++h) {
                        if (k >= hp[h].n_blk) continue;
                        for (int sl = g * G / hp[h].S; sl < (g + 1) * G / hp[h].S; ++sl) {
                            int rec = -1;   // the launch-order record of this (half, slice, block)
                            for (int b = 0; b < n_wg && rec < 0; ++b) {
                                const int32_t *e = &w[(size_t)w[LP_WG_OFS] + (size_t)b * 4];
                                if (e[0] == h && e[1] == sl && e[2] == k) rec = b;
                            }
                            w.insert(w.end(), {rec, 0, 0, g});   // {record index (LP_WGX_OFS), -, -, group}
                            ++n_items;
                        }
                    }
            w[hdr + 4 + 2 * (size_t)q] = n_items;
            w[hdr + 4 + 2 * (size_t)q + 1] = (int32_t)(first / 4);
        }
    }
    // ---- LDS row of every source row.  Identity, except that the kHot highest-degree sources are dealt round-robin over
    // the 16 / LP bank classes (each swaps rows with a cold source that sits in the wanted class): an item that half the
    // users rated would otherwise load its class in every lane group.  Cold rows keep their natural order, so the staging
    // writes of consecutive rows stay conflict-free.
    std::vector<int32_t> perm[2];
    static const int no_perm = RK_TUNE_INT("RK_LDS_NOPERM", 0);   // tuning: identity
    static const int hot_env = RK_TUNE_INT("RK_LDS_HOT", 0);
    for (int h = 0; h < 2; ++h) {
        const int n_src = h ? U : I, src0 = h ? 0 : U, K = 16 / hp[h].lp;
        perm[h].resize((size_t)n_src);
        for (int c = 0; c < n_src; ++c) perm[h][(size_t)c] = c;
        const int kHot = no_perm ? 0 : std::min(n_src / 2, hot_env > 0 ? hot_env : 512);
        if (kHot > 0) {
            std::vector<int32_t> order((size_t)n_src);
            for (int c = 0; c < n_src; ++c) order[(size_t)c] = c;
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return rp[src0 + x + 1] - rp[src0 + x] > rp[src0 + y + 1] - rp[src0 + y]; });
            std::vector<char> hot((size_t)n_src, 0);
            for (int k = 0; k < kHot; ++k) hot[(size_t)order[(size_t)k]] = 1;
            std::vector<int32_t> owner((size_t)n_src);   // source currently stored in LDS row q
            for (int c = 0; c < n_src; ++c) owner[(size_t)c] = c;
            std::vector<int> cursor((size_t)K, 0);        // next candidate row of each class (rows q = class, class + K, ...)
            for (int k = 0; k < kHot; ++k) {
                const int c = order[(size_t)k], want = k % K;
                if (perm[h][(size_t)c] % K == want) continue;
                int q = want + cursor[(size_t)want] * K;
                while (q < n_src && hot[(size_t)owner[(size_t)q]]) { ++cursor[(size_t)want]; q = want + cursor[(size_t)want] * K; }
                if (q >= n_src) continue;
                ++cursor[(size_t)want];
                const int other = owner[(size_t)q], mine = perm[h][(size_t)c];
                perm[h][(size_t)c] = q; owner[(size_t)q] = c;
                perm[h][(size_t)other] = mine; owner[(size_t)mine] = other;
            }
        }
        w[h ? LP_PERM1 : LP_PERM0] = (int32_t)w.size();
        w.insert(w.end(), perm[h].begin(), perm[h].end());
    }
    w[LP_BLK_OFS] = (int32_t)w.size();
    const size_t n_blocks_total = (size_t)hp[0].n_blk + (size_t)hp[1].n_blk;
    w.resize(w.size() + n_blocks_total * LB_WORDS, 0);
    // ---- per (half, row block): chunks, tasks, conflict-free column stream.  Blocks are independent: built by a few host
    // threads (the plan sits on the perturb-retrain loop: one per injected graph), merged in block order afterwards.
    struct BlockOut {
        int32_t row0 = 0, n_rows = 0, n_part = 0, n_tasks = 0;
        std::vector<int32_t> tasks, dst, pp;
        std::vector<uint16_t> stream;
    };
    std::vector<BlockOut> blocks(n_blocks_total);
    static const int no_colour = RK_TUNE_INT("RK_LDS_NOCOLOUR", 0);   // tuning: CSR order
    auto build_block = [&](size_t bi, PlanScratch &ws) {
        const int h = bi < (size_t)hp[0].n_blk ? 0 : 1;
        const int rb = (int)(bi - (h ? (size_t)hp[0].n_blk : 0));
        const int SL = 64 / hp[h].lp, C = hp[h].chunk;
        const int cls0 = h ? U : 0;           // node id of the first output row of this class
        const int src0 = h ? 0 : U;           // node id of the first source row
        const int n_src = h ? U : I;
        const int r_lo = hp[h].bounds[(size_t)rb], r_hi = hp[h].bounds[(size_t)rb + 1];
        const int n_rows = r_hi - r_lo;
        BlockOut &o = blocks[bi];
        std::vector<int32_t> &pp = o.pp;
        pp.assign((size_t)n_rows + 1, 0);
        std::vector<Chunk> &chunks = ws.chunks;
        chunks.clear();
        for (int r = r_lo; r < r_hi; ++r) {
            const int b = rp[r], n = rp[r + 1] - b;
            // ceil(n / C) chunks of (almost) equal length: fewer short leftovers than full chunks + a remainder
            static const int even_env = RK_TUNE_INT("RK_LDS_EVEN", 1);
            const int n_ch = (n + C - 1) / C;
            int ci = 0;
            for (int k = 0; ci < n_ch; ++ci) {
                const int len = even_env ? (int)(((long long)n * (ci + 1)) / n_ch - ((long long)n * ci) / n_ch) : std::min(C, n - k);
                chunks.push_back({(len + 7) & ~7, len, b + k, pp[(size_t)(r - r_lo)] + ci});
                k += len;
            }
            pp[(size_t)(r - r_lo) + 1] = pp[(size_t)(r - r_lo)] + ci;
        }
        std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk &x, const Chunk &y) { return x.padded > y.padded; });
        // A "group" = SL chunks walked together (one per lane slot) and coloured jointly; it is executed as pieces of at
        // most half its 8-entry blocks -- the tasks the waves pop -- each with its own partial-sum slot per chunk, so that the
        // sixteen waves of the workgroup finish within a few blocks of each other (whole groups as tasks left a quarter
        // of the waves idle behind the last long ones).
        static const int sub_env = RK_TUNE_INT("RK_LDS_SUB", 0);   // tuning: 2 = split groups in two
        const int kMaxPieces = sub_env == 2 ? 2 : 1;   // (measured: pieces cost the row phase more than the balance gains -- off)
        const int n_groups = (int)((chunks.size() + (size_t)SL - 1) / (size_t)SL);
        o.row0 = r_lo - cls0; o.n_rows = n_rows;
        std::vector<uint16_t> &stream = o.stream;
        size_t unit = 0;   // 16-byte units since the block's stream began
        const int LPh = hp[h].lp, K = 16 / LPh;   // lanes per entry, bank classes (= chunks per 16-lane group)
        auto zero_row = [&](int klass) { return n_src + ((klass - n_src % K) % K + K) % K; };
        std::vector<int> gslots[4];
        for (int gi = 0; gi < 4; ++gi)
            for (int l = 0; l < 16; ++l) {
                const int sl = kB128Group[gi][l] / LPh;
                if (gslots[gi].empty() || gslots[gi].back() != sl) gslots[gi].push_back(sl);
            }
        struct Piece { int32_t unit, nb, group, part; };
        std::vector<Piece> pieces;
        std::vector<int> group_parts((size_t)n_groups, 0);
        for (int t = 0; t < n_groups; ++t) {
            const size_t c0 = (size_t)t * SL, c1 = std::min(chunks.size(), c0 + (size_t)SL);
            // Which of the group's chunks share a 16-lane group is free: deal them greedily so that no bank class of a lane
            // group collects many more entries than the chunks are long (the colouring needs max(longest chunk, fullest
            // class) positions).  The chunks are swapped inside [c0, c1): slot = position in the sorted list.
            if (!no_colour && c1 - c0 == (size_t)SL) {
                int hist[64 * 16], load[4 * 16], n_mem[4] = {0, 0, 0, 0};
                size_t members[4][16];
                memset(hist, 0, sizeof(hist));
                memset(load, 0, sizeof(load));
                for (size_t c = c0; c < c1; ++c)
                    for (int k = 0; k < chunks[c].len; ++k) ++hist[(c - c0) * 16 + (size_t)(perm[h][(size_t)(col[chunks[c].e_begin + k] - src0)] % K)];
                for (size_t c = c0; c < c1; ++c) {
                    int best = -1, best_cost = 0;
                    for (int gi = 0; gi < 4; ++gi) {
                        if (n_mem[gi] >= K) continue;
                        int cost = 0;
                        for (int k = 0; k < K; ++k) cost = std::max(cost, load[gi * 16 + k] + hist[(c - c0) * 16 + (size_t)k]);
                        if (best < 0 || cost < best_cost) { best = gi; best_cost = cost; }
                    }
                    members[best][n_mem[best]++] = c;
                    for (int k = 0; k < K; ++k) load[best * 16 + k] += hist[(c - c0) * 16 + (size_t)k];
                }
                std::vector<Chunk> &placed = ws.placed;
                placed.resize(c1 - c0);
                for (int gi = 0; gi < 4; ++gi)
                    for (int j = 0; j < K; ++j) placed[(size_t)gslots[gi][(size_t)j]] = chunks[members[gi][j]];
                std::copy(placed.begin(), placed.end(), chunks.begin() + (long)c0);
            }
            // per 16-lane group: its K slots' entries, ordered so that every wave-instruction reads K different classes
            // (flat scratch: row j of cls / pos = slot gslots[gi][j]; slot_pos row = slot of the group)
            const int stride = C;
            ws.slot_pos.resize((size_t)SL * stride);
            ws.slot_len.assign((size_t)SL, 0);
            ws.cls.resize((size_t)K * stride);
            ws.pos.resize((size_t)K * stride);
            ws.len.resize((size_t)K);
            int longest = 0;
            for (int gi = 0; gi < 4; ++gi) {
                int *cls = ws.cls.data(), *pos = ws.pos.data(), *len = ws.len.data();
                for (int j = 0; j < K; ++j) {
                    const size_t c = c0 + (size_t)gslots[gi][(size_t)j];
                    len[j] = 0;
                    if (c >= c1) continue;
                    const Chunk &ck = chunks[c];
                    len[j] = ck.len;
                    for (int k = 0; k < ck.len; ++k) cls[(size_t)j * stride + k] = perm[h][(size_t)(col[ck.e_begin + k] - src0)] % K;
                }
                if (no_colour) {
                    for (int j = 0; j < K; ++j) { for (int k = 0; k < len[j]; ++k) pos[(size_t)j * stride + k] = k; longest = std::max(longest, len[j]); }
                } else {
                    longest = std::max(longest, colour_group(cls, len, K, stride, K, pos, ws));
                }
                for (int j = 0; j < K; ++j) {
                    const int sl = gslots[gi][(size_t)j];
                    ws.slot_len[(size_t)sl] = len[j];
                    std::copy(pos + (size_t)j * stride, pos + (size_t)j * stride + len[j], ws.slot_pos.begin() + (long)((size_t)sl * stride));
                }
            }
            const int nb = std::max(1, (longest + 7) / 8);
            const size_t sbase = stream.size();
            stream.resize(sbase + (size_t)nb * SL * 8, (uint16_t)0xffff);
            auto at = [&](int slot, int p) -> uint16_t & { return stream[sbase + ((size_t)(p / 8) * SL + (size_t)slot) * 8 + (size_t)(p % 8)]; };
            for (size_t c = c0; c < c1; ++c) {
                const Chunk &ck = chunks[c];
                const int slot = (int)(c - c0);
                for (int k = 0; k < ck.len; ++k) at(slot, ws.slot_pos[(size_t)slot * stride + (size_t)k]) = (uint16_t)perm[h][(size_t)(col[ck.e_begin + k] - src0)];
            }
            // padding: the zero row of a class nobody else in the lane group reads at that position
            for (int gi = 0; gi < 4; ++gi)
                for (int p = 0; p < nb * 8; ++p) {
                    unsigned used = 0;
                    for (int sl : gslots[gi]) if (at(sl, p) != 0xffff) used |= 1u << (at(sl, p) % K);
                    for (int sl : gslots[gi]) {
                        if (at(sl, p) != 0xffff) continue;
                        int k = 0;
                        while (k < K - 1 && (used & (1u << k))) ++k;
                        used |= 1u << k;
                        at(sl, p) = (uint16_t)zero_row(k);
                    }
                }
            // two pieces of about equal length once a group is at least four blocks long
            const int np = (nb >= 4) ? kMaxPieces : 1;
            group_parts[(size_t)t] = np;
            for (int q = 0; q < np; ++q) {
                const int b0 = (int)((long long)nb * q / np), b1 = (int)((long long)nb * (q + 1) / np);
                pieces.push_back({(int32_t)(unit + (size_t)b0 * SL), b1 - b0, t, q});
            }
            unit += (size_t)nb * SL;
        }
        // partial-sum slots: row by row, chunk by chunk (CSR order), piece by piece -- the order phase 3 adds them in
        std::vector<int32_t> chunk_first(chunks.size(), 0);   // first slot of chunk (sorted index)
        {
            std::vector<int32_t> by_pidx(chunks.size(), 0);    // sorted index of the chunk with row-order index pidx
            for (size_t c = 0; c < chunks.size(); ++c) by_pidx[(size_t)chunks[c].pidx] = (int32_t)c;
            int32_t next = 0;
            size_t ci = 0;
            for (int lr = 0; lr < n_rows; ++lr) {
                const int32_t n_ch = pp[(size_t)lr + 1] - pp[(size_t)lr];
                pp[(size_t)lr] = next;
                for (int32_t k = 0; k < n_ch; ++k, ++ci) {
                    const size_t c = (size_t)by_pidx[ci];
                    chunk_first[c] = next;
                    next += group_parts[c / (size_t)SL];
                }
            }
            pp[(size_t)n_rows] = next;
            o.n_part = next;
        }
        std::stable_sort(pieces.begin(), pieces.end(), [](const Piece &x, const Piece &y) { return x.nb > y.nb; });   // longest first
        const int n_tasks = (int)pieces.size();
        o.n_tasks = n_tasks;
        o.tasks.assign((size_t)n_tasks * 2, 0);
        o.dst.assign((size_t)n_tasks * SL, -1);
        for (int t = 0; t < n_tasks; ++t) {
            const Piece &pc = pieces[(size_t)t];
            o.tasks[(size_t)t * 2] = pc.unit;
            o.tasks[(size_t)t * 2 + 1] = pc.nb;
            const size_t c0 = (size_t)pc.group * SL, c1 = std::min(chunks.size(), c0 + (size_t)SL);
            for (size_t c = c0; c < c1; ++c) o.dst[(size_t)t * SL + (c - c0)] = chunk_first[c] + pc.part;
        }
    };
    {
        static const int env_threads = RK_TUNE_INT("RK_LDS_PLAN_THREADS", 0);
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t n_threads = std::min<size_t>(n_blocks_total, env_threads > 0 ? (size_t)env_threads : std::min<unsigned>(hw, 16u));
        std::atomic<size_t> next(0);
        // a worker never lets an exception escape its thread (that would be std::terminate): the first one is parked and
        // rethrown on the caller's thread after the join; the others stop taking blocks
        std::atomic<bool> failed(false);
        std::exception_ptr first_error;
        std::mutex error_mutex;
        auto worker = [&](int cpu) {
            try {
                place_self(cpu);
                PlanScratch ws;
                for (size_t bi = next++; bi < n_blocks_total && !failed.load(std::memory_order_relaxed); bi = next++) build_block(bi, ws);
            } catch (...) {
                std::lock_guard<std::mutex> lock(error_mutex);
                if (!first_error) first_error = std::current_exception();
                failed.store(true, std::memory_order_relaxed);
            }
        };
        static const int pin = RK_TUNE_INT("RK_LDS_PLAN_PIN", 1);   // tuning: 0 = leave the workers to the scheduler
        const std::vector<int> cpus = pin ? worker_cpus() : std::vector<int>();
        std::vector<std::thread> pool;
        pool.reserve(n_threads);
        for (size_t k = 1; k < n_threads; ++k) {
            // thread / pids limit reached (std::system_error): go on with the workers that did start -- the caller's thread
            // drains the queue anyway
            try { pool.emplace_back(worker, cpus.empty() ? -1 : cpus[(k - 1) % cpus.size()]); } catch (const std::system_error &) { break; }
        }
        worker(-1);
        for (auto &th : pool) th.join();
        if (first_error) std::rethrow_exception(first_error);
    }
    std::vector<uint16_t> stream;   // all blocks' column streams, 16-byte units
    std::vector<size_t> stream_ofs16(n_blocks_total, 0);
    for (size_t bi = 0; bi < n_blocks_total; ++bi) {
        const BlockOut &o = blocks[bi];
        if (w.size() & 1) w.push_back(0);   // the int2 task array must be 8-byte aligned
        const size_t task_ofs = w.size();
        w.insert(w.end(), o.tasks.begin(), o.tasks.end());
        const size_t dst_ofs = w.size();
        w.insert(w.end(), o.dst.begin(), o.dst.end());
        const size_t pp_ofs = w.size();
        w.insert(w.end(), o.pp.begin(), o.pp.end());
        int32_t *bd = &w[(size_t)w[LP_BLK_OFS] + bi * LB_WORDS];
        bd[LB_ROW0] = o.row0; bd[LB_NROWS] = o.n_rows; bd[LB_NPART] = o.n_part; bd[LB_NTASKS] = o.n_tasks;
        bd[LB_TASK_OFS] = (int32_t)task_ofs; bd[LB_DST_OFS] = (int32_t)dst_ofs; bd[LB_PP_OFS] = (int32_t)pp_ofs;
        stream_ofs16[bi] = stream.size() / 8;
        stream.insert(stream.end(), o.stream.begin(), o.stream.end());
    }
    // ---- one 64-byte record per workgroup (LW_*): table entry + its block descriptor, in launch order
    {
        while (w.size() & 15) w.push_back(0);
        w[LP_WGX_OFS] = (int32_t)w.size();
        const int G = std::max(hp[0].S, hp[1].S);
        for (int b = 0; b < n_wg; ++b) {
            const int32_t *e = &w[(size_t)w[LP_WG_OFS] + (size_t)b * 4];
            const int h = e[0], sl = e[1], rb = e[2];
            const int32_t *bd = &w[(size_t)w[LP_BLK_OFS] + ((size_t)(h ? hp[0].n_blk : 0) + (size_t)rb) * LB_WORDS];
            int32_t rec[LW_WORDS] = {h, sl, rb, sl * hp[h].S / G, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < LB_WORDS; ++k) rec[LW_BD + k] = bd[k];
            w.insert(w.end(), rec, rec + LW_WORDS);
        }
    }
    w[LP_DINV_OFS] = (int32_t)w.size();
    w.resize(w.size() + (size_t)N);
    memcpy(&w[(size_t)w[LP_DINV_OFS]], dinv.data(), sizeof(float) * (size_t)N);
    while (w.size() & 3) w.push_back(0);   // the stream is read with 16-byte loads
    const size_t stream_base16 = w.size() / 4;
    for (size_t bi = 0; bi < n_blocks_total; ++bi) w[(size_t)w[LP_BLK_OFS] + bi * LB_WORDS + LB_STREAM_OFS] = (int32_t)(stream_base16 + stream_ofs16[bi]);
    for (int b = 0; b < n_wg; ++b) {   // (the records were written before the stream had its place)
        int32_t *rec = &w[(size_t)w[LP_WGX_OFS] + (size_t)b * LW_WORDS];
        const size_t bi = (size_t)(rec[LW_HALF] ? hp[0].n_blk : 0) + (size_t)rec[LW_BLOCK];
        rec[LW_BD + LB_STREAM_OFS] = w[(size_t)w[LP_BLK_OFS] + bi * LB_WORDS + LB_STREAM_OFS];
    }
    w.resize(w.size() + stream.size() / 2);
    memcpy(&w[stream_base16 * 4], stream.data(), stream.size() * sizeof(uint16_t));
    w[LP_NWORDS] = (int32_t)w.size();
    rk_lds_info &fi = pl->info;
    memset(&fi, 0, sizeof(fi));
    fi.n_wg = n_wg; fi.lds_bytes = w[LP_LDS_BYTES]; fi.lpa = hp[0].lp; fi.lpb = hp[1].lp;
    fi.n_users = U; fi.n_items = I; fi.dim = dim; fi.lsu = lsu; fi.lsi = lsi; fi.chunk = w[LP_CHUNK];
    fi.wgx_ofs = w[LP_WGX_OFS]; fi.dinv_ofs = w[LP_DINV_OFS]; fi.perm0_ofs = w[LP_PERM0]; fi.perm1_ofs = w[LP_PERM1]; fi.mq_ofs = w[LP_MQ_OFS];
    *info = fi;
    *n_words = (int64_t)w.size();
    *out = pl_owner.release();
    return RK_OK;
}

