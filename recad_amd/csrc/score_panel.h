// Full-catalog scoring + selection with the scores held in REGISTERS (gfx950), for Normal.user_item_model_generate
// (recad/workflow/normal.py:57-93).  No score matrix, no per-tile bookkeeping.
//
// One workgroup (8 waves, 128 VGPRs each, two workgroups per CU) owns 16 user rows and walks the catalogue in PANELS of
// 128 * NTW items.  Inside a panel wave w owns the 16-item tiles w, w + 8, ...; a tile's 16 x 16 scores are one
// accumulator block of v_mfma_f32_16x16x4_f32 taken as (items, users), so a lane holds, for ONE user row (lane & 15),
// the scores of four consecutive items per tile -- 4 * NTW scores per lane, the whole 16 x (128 NTW) panel in the
// register file.  The MFMA is the exact-fp32 k-ordered fmaf chain, so a score has the bits of the oracle's scalar loop.
//   * operands: the item rows come straight from global memory as 16-byte loads of a K-PERMUTED copy of the item table
//     (position 16c + 4g + s holds k = 16c + 4s + g: the four floats a lane group g feeds into the MFMAs s = 0..3 of a
//     16-chunk are contiguous; rk_score_topk makes the copy, one pass over the table); the 16 user rows are permuted the
//     same way into registers (dim <= 128) or LDS (dim <= 256).  No LDS traffic and no barrier inside a panel's MFMAs.
//   * a panel's epilogue runs on the registers: seen items (a per-row LDS bitmap filled from the sorted train lists) become
//     -inf; target ranks are per-lane counters (#(s > s_t) + #(s == s_t, id < t)); the candidates are the scores at or
//     above a per-row bound tau and go to a 384-slot list per row in LDS as (key << 32 | ~id) composites.
//   * tau: in the first panel the K-th largest of 256 group maxima per row (each maximum is a distinct item, so it is a
//     lower bound of the K-th largest score; with groups of 8 scores about 1.25 K scores lie above it); afterwards the
//     (coarse) K-th largest key of the row's list.  Later panels hold larger item ids, so "s >= tau" keeps a superset of
//     the final top K whatever the ties.
//   * a row whose list would overflow in a panel (scores that grow with the item id, a panel of systematically better items)
//     gets its bound RAISED first: a bisection on the key space that counts straight from the registers finds a key T with
//     K <= #(list and panel entries >= T) <= 256, the lists are cut to it and the panel is collected again.  Rows that no
//     bound can separate (tie-heavy or constant rows, K near 256) send the workgroup through the SAFE form of the panel:
//     computed again, one tile per wave and round (128 consecutive ids), every row cut back to its exact top K composites
//     after each round -- exact by construction, slow, and only taken by degenerate rows.
//   * at the end a wave per row cuts the list to <= 256 entries, sorts them (bitonic, 4 per lane) and writes the K
//     results: (score desc, id asc) exactly like the oracle's scan.
#pragma once
#include <float.h>

#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "common.h"

typedef float sel_f32x4 __attribute__((ext_vector_type(4)));
typedef float sel_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned score_key(float s)
{
    // monotone float -> uint, branch-free; 0 is reserved for "excluded" (a seen item, encoded as -inf).  s + 0.0f folds -0.0 onto
    // +0.0 so that key equality is float equality (the oracle compares floats) and changes no other number.  NaN scores (a diverged
    // poisoned retrain; the oracle's float compares give them no place at all): a NaN with the sign bit set is excluded like -inf,
    // a positive NaN keeps a key above +inf's, a signalling NaN is quieted by the addition -- the SAME key in every kernel that
    // ranks scores (topk_wave_kernel, topk_rows_kernel, the panel form's final sort): round 5 had two encoders that disagreed there.
    const unsigned u = __float_as_uint(s + 0.0f);
    const unsigned k = u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
    return k <= 0x007fffffu ? 0u : k;   // k <= 0x007fffff: -inf (exactly 0x007fffff) or a negative NaN
}

__device__ __forceinline__ float key_score(unsigned k)
{
    // inverse of score_key for every finite score and +inf (-0.0 comes back as +0.0)
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// Threshold search over the (distinct) composites held 8 per lane (unused slots 0), n_valid >= K.  Bit-wise binary
// search from the top: T grows while at least K composites stay >= T.  After `min_bits` bits it stops as soon as at
// most `limit` composites are >= T (a COARSE threshold: cheap, and any T with count >= K is a valid pruning bound);
// with limit == K it runs to the exact K-th composite (early exit once the count is exactly K).  Key bits are
// searched on the high words alone (32-bit compares); the id bits above id_bits are ones in every entry.
// Returns T; *cnt_out = #(composite >= T).
__device__ __forceinline__ unsigned long long wave_threshold(const unsigned long long (&c)[8], int n, int K, int limit, int min_bits,
                                                             int id_bits, int *cnt_out)
{
    unsigned hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) hi[j] = (unsigned)(c[j] >> 32);
    unsigned Tk = 0u;
    int cntT = n, done_bits = 0;
    for (int bit = 31; bit >= 0; --bit) {
        if (done_bits >= min_bits && cntT <= limit) { *cnt_out = cntT; return (unsigned long long)Tk << 32; }
        const unsigned trial = Tk | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += __popcll(__ballot(hi[j] >= trial));
        if (cnt >= K) { Tk = trial; cntT = cnt; }
        ++done_bits;
    }
    unsigned long long T = (unsigned long long)Tk << 32;
    if (cntT <= limit) { *cnt_out = cntT; return T; }
    // more than `limit` entries share the K-th key: resolve the tie on the (inverted) item ids
    if (id_bits < 32) T |= (0xffffffffULL >> id_bits) << id_bits;
    for (int bit = (id_bits < 32 ? id_bits : 32) - 1; bit >= 0; --bit) {
        const unsigned long long trial = T | (1ULL << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += __popcll(__ballot(c[j] >= trial));
        if (cnt >= K) { T = trial; cntT = cnt; }
        if (cntT <= limit) break;
    }
    *cnt_out = cntT;
    return T;
}

static constexpr int kPanMaxRows = 32;       // user rows per workgroup: 16 or 32
static constexpr int kPanWaves = 8;
static constexpr int kPanNT = kPanWaves * 64;
static constexpr int kPanCap = 384;          // candidate slots per row: K (<= 256) + one safe round (128)
static constexpr int kPanMaxT = 4;
static constexpr int kPanDefaultMinItems = 16384;   // rk_score_topk_plan takes this form by default from this catalogue size on

struct PanArgs {
    int nb, n_items, d, K;
    const float *utab;
    const int *user_ids;
    const float *itab;            // natural layout (target rows)
    const float *itabp;           // k-permuted copy, rows of 16 * DC floats
    const float *ubias, *ibias;   // both or neither: s = ((dot + ubias[u]) + ibias[i]) + mean
    float mean;
    const int *seen_ptr, *seen_idx;
    const int *targets;
    int n_targets;
    int *top_ids;
    float *top_scores;
    float *target_score;
    int *target_rank;
    int id_bits;
    int force_safe;               // tests: every panel through the safe form
    unsigned long long *stamps;   // diagnostic, nullable (RK_PAN_STAMPS=1): [grid][36] wall-clock stamps of thread 0
};

// out[item][16c + 4g + s] = in[item][16c + 4s + g]  (zero beyond d): one thread per float4 of the copy
__global__ __launch_bounds__(256) void pan_permute_kernel(const float *__restrict__ in, int n_items, int d, int dc, float *__restrict__ out)
{
    const long long n4 = (long long)n_items * dc * 4;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n4; t += (long long)gridDim.x * 256) {
        const long long item = t / (dc * 4);
        const int q = (int)(t % (dc * 4)), c = q >> 2, g = q & 3;
        const float *src = in + item * d + 16 * c + g;
        float4 v;
        v.x = 16 * c + g < d ? src[0] : 0.f;
        v.y = 16 * c + 4 + g < d ? src[4] : 0.f;
        v.z = 16 * c + 8 + g < d ? src[8] : 0.f;
        v.w = 16 * c + 12 + g < d ? src[12] : 0.f;
        reinterpret_cast<float4 *>(out)[t] = v;
    }
}

// The candidate lists hold RAW entries (score bits << 32 | ~id): the panels' epilogues append without the key transform;
// the wave-level consumers below turn an entry into its (key << 32 | ~id) composite when they load it.
__device__ __forceinline__ unsigned long long pan_raw(float s, unsigned id) { return ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(~id); }
__device__ __forceinline__ unsigned long long pan_comp(unsigned long long raw)
{
    return ((unsigned long long)score_key(__uint_as_float((unsigned)(raw >> 32))) << 32) | (raw & 0xffffffffULL);
}

__device__ __forceinline__ float pan_bound(unsigned T)
{
    // the smallest float whose key is >= T (T: a key prefix with zero low bits); keys below those of finite floats: everything
    return T < 0x00800000u ? -FLT_MAX : key_score(T);
}

// Two sets of 256 composites, 4 per lane each (element 4 * lane + q), sorted DESCENDING across the wave; the two sets are
// independent, so their lane exchanges overlap
__device__ __forceinline__ void pan_sort256x2(unsigned long long (&c0)[4], unsigned long long (&c1)[4], int lane)
{
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long o0 = __shfl_xor(c0[q], j >> 2, 64), o1 = __shfl_xor(c1[q], j >> 2, 64);
                    const int e = lane * 4 + q;
                    const bool take_max = ((e & j) == 0) == ((e & k) == 0);
                    c0[q] = ((c0[q] > o0) == take_max) ? c0[q] : o0;
                    c1[q] = ((c1[q] > o1) == take_max) ? c1[q] : o1;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q & j) continue;
                    const int e = lane * 4 + q;
                    const bool up = (e & k) == 0;
                    {
                        const unsigned long long x = c0[q], y = c0[q | j];
                        const unsigned long long hi = x > y ? x : y, lo = x > y ? y : x;
                        c0[q] = up ? hi : lo;
                        c0[q | j] = up ? lo : hi;
                    }
                    {
                        const unsigned long long x = c1[q], y = c1[q | j];
                        const unsigned long long hi = x > y ? x : y, lo = x > y ? y : x;
                        c1[q] = up ? hi : lo;
                        c1[q | j] = up ? lo : hi;
                    }
                }
            }
        }
    }
}

// One wave: cut a row's list (n <= 512 entries in LDS) down to the entries at or above a threshold that at least K of
// them reach and at most `limit` do (limit >= K; ties beyond that are resolved on the ids; limit == K: the exact top K).
// Returns the new count; *kmin = the smallest key kept (n >= K: an inclusive lower bound of the row's K-th largest key,
// the K-th key itself when K entries are kept), 0 when n < K (list untouched).
__device__ __noinline__ int pan_prune(unsigned long long *row, int n, int K, int limit, int min_bits, int id_bits, int lane, unsigned *kmin)
{
    *kmin = 0u;
    if (n < K) return n;
    unsigned long long c[8], raw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        raw[j] = (j * 64 + lane < n) ? row[j * 64 + lane] : 0ULL;
        c[j] = (j * 64 + lane < n) ? pan_comp(raw[j]) : 0ULL;
    }
    unsigned long long T = 0ULL;
    if (n > limit) {
        int cnt = n;
        T = wave_threshold(c, n, K, limit, min_bits, id_bits, &cnt);
    }
    int base = 0;
    unsigned km = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool keep = c[j] >= T && c[j] != 0ULL;
        const unsigned long long m = __ballot(keep);
        if (keep) {
            if (n > limit) row[base + __popcll(m & ((1ULL << lane) - 1ULL))] = raw[j];
            const unsigned k = (unsigned)(c[j] >> 32);
            km = k < km ? k : km;
        }
        base += __popcll(m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)km, o, 64);
        km = other < km ? other : km;
    }
    *kmin = km;
    return base;
}

template <int NTW>
struct PanGeom {
    static constexpr int PI = 128 * NTW;                                  // items per panel
    // The seen bitmap of a panel is kept TRANSPOSED: the 4 * NTW bits a lane tests (its four columns of each of its wave's tiles)
    // are the two dwords [row][wave slot wq][g]: bit 4 * i + r of the pair = column 16 * (wq + 8 i) + 4 g + r.  One 8-byte read per
    // row block and panel, immediate bit offsets per tile (a dword per tile read where it was used cost every tile an LDS round trip).
    static constexpr int RS = 68;                                         // row stride in dwords (64 + 4: the 16 rows x two g of a half-wave read 64 distinct banks)
    static constexpr int TPG = (NTW + 7) / 8;                             // tiles per maxima group (8 groups per lane)
};

template <int NTW, int DC, int RB>
inline size_t pan_lds_bytes()
{
    return (size_t)(16 * RB) * kPanCap * 8 + (size_t)DC * 1024 * RB + 2 * (size_t)(16 * RB) * PanGeom<NTW>::RS * 4;
}

// RB: blocks of 16 user rows per workgroup.  RB == 1: 128 registers per wave, two workgroups per CU (small user blocks: more
// workgroups); RB == 2: 32 rows, 256 registers, one workgroup per CU -- every item operand loaded from L2 feeds 8 MFMAs instead
// of 4 (at 16 rows the sweep is bound by the L2 -> L1 operand traffic: 8 flop per byte), and the lane-parallel epilogue work per
// workgroup barrier doubles.
template <int NTW, int DC, int NTG, int RB>
__global__ __launch_bounds__(kPanNT, RB == 1 ? 4 : 2) void score_panel_kernel(const PanArgs a)
{
    using G = PanGeom<NTW>;
    constexpr int PI = G::PI, RS = G::RS, TPG = G::TPG, NIT = NTW * DC, PF = RB == 1 ? 3 : 4, R = 16 * RB;
    extern __shared__ __attribute__((aligned(16))) unsigned char pan_smem[];
    unsigned long long *sList = reinterpret_cast<unsigned long long *>(pan_smem);                 // [R][kPanCap]
    unsigned *sMax = reinterpret_cast<unsigned *>(pan_smem);                                      // [R][256], first panel only (lists still empty)
    float *sA = reinterpret_cast<float *>(pan_smem + (size_t)R * kPanCap * 8);                     // the user rows, k-permuted, as float2 halves (a_pos below)
    unsigned *sBits = reinterpret_cast<unsigned *>(pan_smem + (size_t)R * kPanCap * 8 + (size_t)DC * 1024 * RB);   // [2][R][RS]
    __shared__ int sCnt[kPanMaxRows], sCnt0[kPanMaxRows], sUid[kPanMaxRows], sCur[kPanMaxRows], sEnd[kPanMaxRows], sRank[kPanMaxRows][kPanMaxT],
        sFlag[2], sStrict[kPanMaxRows], sTot[kPanMaxRows], sCLo[kPanMaxRows], sAct[2], sFail;
    __shared__ unsigned sLo[kPanMaxRows], sHi[kPanMaxRows];
    __shared__ float sTau[kPanMaxRows];
    const int tid = threadIdx.x, lane = tid & 63, w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = lane & 15, g0 = lane >> 4;
    int w = w0, u = u0, g = g0;
    const int row0 = blockIdx.x * R;
    const int n_in = a.n_targets;
    const int dp = 16 * DC;
    const int n_panels = (a.n_items + PI - 1) / PI;
    const unsigned lds_list = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)pan_smem;   // LDS byte address of the lists

    // diagnostic stamps (thread 0): [0] start, [1] after the prologue, [2 + 8 * min(p, 3) + k] end of phase k of panel p, [34] ranks, [35] end
#define PAN_STAMP(slot) if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 36 + (slot)] = wall_clock64();
    PAN_STAMP(0)
    // ---- prologue
    if (tid < R) {
        const int gr = row0 + tid;
        const int uid = gr < a.nb ? a.user_ids[gr] : -1;
        sUid[tid] = uid;
        sCur[tid] = uid >= 0 ? a.seen_ptr[uid] : 0;
        sEnd[tid] = uid >= 0 ? a.seen_ptr[uid + 1] : 0;
        sCnt[tid] = 0;
        sCnt0[tid] = 0;
        sTau[tid] = -FLT_MAX;
#pragma unroll
        for (int t = 0; t < kPanMaxT; ++t) sRank[tid][t] = 0;
    }
    if (tid == 0) { sFlag[0] = 0; sFlag[1] = 0; }
    for (int i = tid; i < 2 * R * RS; i += kPanNT) sBits[i] = 0u;
    __syncthreads();
    // the user rows, k-permuted, into LDS: A[row][16c + 4s + g], s = 0..3, as the float4 of (c, g, row)
    const bool full_k = a.d == 16 * DC;   // no k padding: unguarded loads
    auto load_a = [&](int uid, int c, int gg) -> sel_f32x4 {
        sel_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (uid >= 0) {
            const float *src = a.utab + (size_t)uid * a.d + 16 * c + gg;
            if (full_k) { v.x = src[0]; v.y = src[4]; v.z = src[8]; v.w = src[12]; }
            else {
                if (16 * c + gg < a.d) v.x = src[0];
                if (16 * c + 4 + gg < a.d) v.y = src[4];
                if (16 * c + 8 + gg < a.d) v.z = src[8];
                if (16 * c + 12 + gg < a.d) v.w = src[12];
            }
        }
        return v;
    };
    // LDS layout of the user operand: float2 HALVES {x, y} / {z, w} of the float4 of (c, g, row) at [(c * 2 + h) * 4 + g][row'],
    // row' = row ^ 16 for odd g at 32 rows -- an 8-byte read of 32 lanes (16 rows x two g) then covers all 64 banks.  The sweep
    // reads the halves one MFMA pair ahead (below); the float4 readers (target tile, safe form) put them together.
    auto a_pos = [&](int c, int h, int gg, int row) -> int { return (((c * 2 + h) * 4 + gg) * R + (row ^ ((R == 32 && (gg & 1)) ? 16 : 0))) * 2; };
    for (int i = tid; i < DC * 4 * R; i += kPanNT) {
        const int row = i % R, gg = (i / R) & 3, c = i / (4 * R);
        const sel_f32x4 v = load_a(sUid[row], c, gg);
        *reinterpret_cast<sel_f32x2 *>(sA + a_pos(c, 0, gg, row)) = sel_f32x2{v.x, v.y};
        *reinterpret_cast<sel_f32x2 *>(sA + a_pos(c, 1, gg, row)) = sel_f32x2{v.z, v.w};
    }
    __syncthreads();
    auto a_half = [&](int c, int rb, int h) -> sel_f32x2 { return *reinterpret_cast<const sel_f32x2 *>(sA + a_pos(c, h, g, rb * 16 + u)); };
    auto a_op = [&](int c, int rb) -> sel_f32x4 {
        const sel_f32x2 lo = a_half(c, rb, 0), hi = a_half(c, rb, 1);
        return sel_f32x4{lo.x, lo.y, hi.x, hi.y};
    };
    // target scores before masking (normal.py:83-85): one extra MFMA tile whose "items" are the targets -- the same k-ordered
    // chain as every other score; every wave computes it (no LDS hand-off), lanes 0..15 of wave 0 write it out
    float ts[RB][NTG], ub[RB];
    int tgt[NTG], cntr[RB][NTG];
#pragma unroll
    for (int t = 0; t < NTG; ++t) tgt[t] = t < n_in ? a.targets[t] : -1;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int my_uid = sUid[rb * 16 + u];
        ub[rb] = (a.ubias && my_uid >= 0) ? a.ubias[my_uid] : 0.f;
        sel_f32x4 tacc = {0.f, 0.f, 0.f, 0.f};
        if (n_in > 0) {
            const int tj = a.targets[u < n_in ? u : 0];
            const float *trow = a.itabp + (size_t)tj * dp + 4 * g;
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const sel_f32x4 b = *reinterpret_cast<const sel_f32x4 *>(trow + 16 * c);
                const sel_f32x4 av = a_op(c, rb);
                tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, av.x, tacc, 0, 0, 0);
                tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, av.y, tacc, 0, 0, 0);
                tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, av.z, tacc, 0, 0, 0);
                tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, av.w, tacc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            float v = tacc[t];                       // lanes 0..15 (g == 0): target t of row u
            if (a.ubias && t < n_in) v = ((v + ub[rb]) + a.ibias[tgt[t]]) + a.mean;
            if (w == 0 && g == 0 && t < n_in && my_uid >= 0) a.target_score[(size_t)(row0 + rb * 16 + u) * a.n_targets + t] = v;
            ts[rb][t] = __shfl(v, u, 64);
            cntr[rb][t] = 0;
        }
    }
    PAN_STAMP(1)
    // Per panel: MFMAs | seen bits of the panel -> bitmap | barrier A | one pass over the registers (mask, target counts, hits)
    // | slot reservation, writes | barrier B | the rows' owners refresh tau (no barrier: the next MFMAs touch none of it).
    // The bitmap is double-buffered: panel p reads buffer p & 1, and zeroes the other one for panel p + 1 between A and B.
    // Rows rb * 16 + {2w, 2w + 1} belong to wave w for everything that is done per row.
    sel_f32x4 acc[RB][NTW];
#pragma unroll 1
    for (int p = 0; p < n_panels; ++p) {
        const int pbase = p * PI;
        // (opaque copies: per-tile addresses and shifts derived from them are recomputed where they are used instead of being
        //  hoisted out of the panel loop -- 15 tiles' worth of invariants do not fit the register budget)
        w = w0; u = u0; g = g0;
        asm volatile("" : "+s"(w), "+v"(u), "+v"(g));
        unsigned *bits = sBits + (p & 1) * (R * RS), *bits_next = sBits + ((p + 1) & 1) * (R * RS);
        // seen ids of this panel: first loads issued before the MFMAs, consumed after them (32 lanes per row, 16 rows per round)
        const int sj = tid & 31;
        int seen_cur[RB], seen_id[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int sr = rb * 16 + (tid >> 5);
            seen_cur[rb] = sCur[sr];
            seen_id[rb] = seen_cur[rb] + sj < sEnd[sr] ? a.seen_idx[seen_cur[rb] + sj] : 0x7fffffff;
        }

        const bool force_safe = a.force_safe != 0;
        if (!force_safe) {
            // ---- the panel's scores: NTW tiles x DC chunks of 4 MFMAs per row block, item operands PF chunks ahead
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int i = 0; i < NTW; ++i) acc[rb][i] = sel_f32x4{0.f, 0.f, 0.f, 0.f};
            auto b_ptr = [&](int n) -> const sel_f32x4 * {
                const int i = n / DC, c = n % DC;
                int item = pbase + 16 * (w + 8 * i) + u;
                item = item < a.n_items ? item : a.n_items - 1;
                return reinterpret_cast<const sel_f32x4 *>(a.itabp + (size_t)item * dp + 16 * c + 4 * g);
            };
            sel_f32x4 bq[PF];
#pragma unroll
            for (int n = 0; n < PF; ++n) bq[n] = *b_ptr(n < NIT ? n : NIT - 1);
            // The user operand comes from LDS one MFMA PAIR ahead: the {z, w} halves of chunk n are requested in front of its
            // {x, y} MFMAs and the {x, y} halves of chunk n + 1 in front of its {z, w} MFMAs, into the registers the pair before
            // has just released -- 64 cycles of MFMAs per row block cover each read.  (Read as one float4 right in front of its
            // four MFMAs, every chunk waited out an LDS round trip: one s_waitcnt per four MFMAs, the pipe 0.69 busy.)
            sel_f32x2 axy[RB], azw[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) axy[rb] = a_half(0, rb, 0);
#pragma unroll
            for (int n = 0; n < NIT; ++n) {
                const int i = n / DC, c = n % DC;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) azw[rb] = a_half(c, rb, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[n % PF].x, axy[rb].x, acc[rb][i], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[n % PF].y, axy[rb].y, acc[rb][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (n + 1 < NIT) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) axy[rb] = a_half((n + 1) % DC, rb, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[n % PF].z, azw[rb].x, acc[rb][i], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[n % PF].w, azw[rb].y, acc[rb][i], 0, 0, 0);
                // the slot just consumed takes the operands of PF chunks ahead; the fences keep the compiler from sinking the
                // load down to its use (it does, to save registers, and then every chunk waits for its own load)
                __builtin_amdgcn_sched_barrier(0);
                if (n + PF < NIT) bq[n % PF] = *b_ptr(n + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        PAN_STAMP(2 + 8 * (p < 3 ? p : 3) + 0)
        // ---- the panel's bitmap: seen items (sorted lists: the consumed lanes of a row form a prefix), columns past the
        //      catalogue, rows past the block
        {
            const int sub = lane & 32;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int sr = rb * 16 + (tid >> 5);
                const int seen_end = sEnd[sr];
                int cur = seen_cur[rb], id = seen_id[rb];
                for (;;) {
                    const bool in = id < pbase + PI;
                    if (in) {
                        const int col = id - pbase, tile = col >> 4, bit = ((tile >> 3) << 2) | (col & 3);
                        atomicOr(&bits[sr * RS + ((((tile & 7) << 2) | ((col >> 2) & 3)) << 1) + (bit >> 5)], 1u << (bit & 31));
                    }
                    const int c32 = __popc((unsigned)(__ballot(in) >> sub));
                    cur += c32;
                    if (c32 < 32) break;
                    id = cur + sj < seen_end ? a.seen_idx[cur + sj] : 0x7fffffff;
                }
                if (sj == 0) sCur[sr] = cur;
            }
            if (pbase + PI > a.n_items || row0 + R > a.nb) {   // (workgroup-uniform)
                for (int i = tid; i < R * 64; i += kPanNT) {
                    const int r = i >> 6, e = i & 63, wq = e >> 3, gg = (e >> 1) & 3, half = e & 1;
                    unsigned m = 0u;
                    if (sUid[r] < 0) m = 0xffffffffu;
                    else {
#pragma unroll
                        for (int ii = 0; ii < 8; ++ii) {
                            const int lo = pbase + 16 * (wq + 8 * (half * 8 + ii)) + 4 * gg;   // first of the four columns of bits 4 ii .. 4 ii + 3
                            if (lo + 4 > a.n_items) m |= (lo >= a.n_items ? 0xfu : (0xfu << (a.n_items - lo)) & 0xfu) << (4 * ii);
                        }
                    }
                    if (m) atomicOr(&bits[r * RS + e], m);
                }
            }
        }
        __syncthreads();   // ---- barrier A

        PAN_STAMP(2 + 8 * (p < 3 ? p : 3) + 1)
        for (int i = tid; i < R * RS; i += kPanNT) bits_next[i] = 0u;   // (last read before barrier B of the panel before)
        if (tid == 0) sFlag[(p + 1) & 1] = 0;
        bool safe = force_safe, counted = false;   // counted: this panel's target counts are in
        const bool first = p == 0;
        if (!force_safe) {
            // ---- one pass over the registers: bias, mask, target counts; group maxima in the first panel (one running maximum,
            //      stored to LDS as a key when its group of TPG tiles is complete), hits against tau in the others
            float tau[RB];
            int nh[RB];
            // The pass is specialised per PANEL, not per tile: the tile loop of the common panels (not the first, the single
            // target wholly behind or wholly in front of the panel) is straight-line code -- mask, four compares against the
            // target score, four against tau.  Decided per tile, the same work walked six scalar branches per tile and row block
            // (first panel? biases? target before / inside / behind the tile?) and took 5.8 us per panel instead of the ~2.5 us its
            // vector instructions need.  TM: 0 = decide per tile (first panel, several targets, the panel that holds the target),
            // 1 = every tile lies in front of the target (ties count: >=), 2 = every tile lies behind it (>).
            // (the masking, which rewrites the accumulators, is ONE loop in front of the specialised ones: four copies of a loop that
            //  writes 120 registers met in 120 phi copies and spilled 200 of them)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const uint2 seen = *reinterpret_cast<const uint2 *>(bits + (rb * 16 + u) * RS + ((w * 4 + g) << 1));   // this lane's 4 * NTW bits
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    if (a.ubias) {   // (workgroup-uniform, decided per tile: a loop of its own over all tiles merged 120 registers at its end)
                        const int id0 = pbase + 16 * (w + 8 * i) + 4 * g;
                        float ib[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) ib[r] = id0 + r < a.n_items ? a.ibias[id0 + r] : 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[rb][i][r] = ((acc[rb][i][r] + ub[rb]) + ib[r]) + a.mean;
                    }
                    const unsigned sw = i < 8 ? seen.x : seen.y;
#pragma unroll
                    for (int r = 0; r < 4; ++r)   // seen / out of range: all ones = a NaN, which loses every comparison and every fmaxf below
                        acc[rb][i][r] = __uint_as_float(__float_as_uint(acc[rb][i][r]) | (unsigned)(((int)(sw << (31 - 4 * (i & 7) - r))) >> 31));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            auto pass1 = [&](auto tm_c, auto first_c) __attribute__((always_inline)) {
                constexpr int TM = decltype(tm_c)::value;
                constexpr bool FIRST = decltype(first_c)::value;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int row = rb * 16 + u;
                    float gm = -INFINITY;
                    if (FIRST) {   // groups that do not exist (NTW < 8 * TPG)
#pragma unroll
                        for (int q = (NTW + TPG - 1) / TPG; q < 8; ++q) sMax[row * 256 + (w * 4 + g) * 8 + q] = 0u;
                    }
                    tau[rb] = FIRST ? INFINITY : sTau[row];
                    nh[rb] = 0;
#pragma unroll
                    for (int i = 0; i < NTW; ++i) {
                        if (TM == 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) cntr[rb][0] += acc[rb][i][r] >= ts[rb][0] ? 1 : 0;
                        } else if (TM == 2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) cntr[rb][0] += acc[rb][i][r] > ts[rb][0] ? 1 : 0;
                        } else {
                            const int id0 = pbase + 16 * (w + 8 * i) + 4 * g;
#pragma unroll
                            for (int t = 0; t < NTG; ++t) {
                                const int tile0 = pbase + 16 * (w + 8 * i);       // wave-uniform
                                if (tile0 + 16 <= tgt[t]) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) cntr[rb][t] += acc[rb][i][r] >= ts[rb][t] ? 1 : 0;
                                } else if (tile0 > tgt[t]) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) cntr[rb][t] += acc[rb][i][r] > ts[rb][t] ? 1 : 0;
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        cntr[rb][t] += (id0 + r != tgt[t] && (acc[rb][i][r] > ts[rb][t] || (acc[rb][i][r] == ts[rb][t] && id0 + r < tgt[t]))) ? 1 : 0;
                                }
                            }
                        }
                        if (FIRST) {
                            gm = fmaxf(gm, fmaxf(fmaxf(acc[rb][i][0], acc[rb][i][1]), fmaxf(acc[rb][i][2], acc[rb][i][3])));   // (masked scores are NaNs: ignored)
                            if ((i + 1) % TPG == 0 || i + 1 == NTW) {
                                sMax[row * 256 + (w * 4 + g) * 8 + i / TPG] = gm == -INFINITY ? 0u : score_key(gm);   // (a group of masked scores only: no maximum)
                                gm = -INFINITY;
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) nh[rb] += acc[rb][i][r] >= tau[rb] ? 1 : 0;
                        }
                        __builtin_amdgcn_sched_barrier(0);   // (keeps the tiles' work from being interleaved: 15 tiles of temporaries do not fit)
                    }
                }
            };
            using pan_c0 = std::integral_constant<int, 0>;
            using pan_c1 = std::integral_constant<int, 1>;
            using pan_c2 = std::integral_constant<int, 2>;
            if (first) pass1(pan_c0{}, std::true_type{});
            else if (NTG == 1 && tgt[0] >= pbase + PI) pass1(pan_c1{}, std::false_type{});
            else if (NTG == 1 && tgt[0] < pbase) pass1(pan_c2{}, std::false_type{});
            else pass1(pan_c0{}, std::false_type{});
            counted = true;
            PAN_STAMP(2 + 8 * (p < 3 ? p : 3) + 2)
            if (first) {
                // ---- tau of the first panel: K-th largest of the 256 group maxima of a row, to 16 bits
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int ra = rb * 16 + 2 * w;
                    const uint4 m0 = *reinterpret_cast<const uint4 *>(sMax + ra * 256 + lane * 4);
                    const uint4 m1 = *reinterpret_cast<const uint4 *>(sMax + (ra + 1) * 256 + lane * 4);
                    unsigned T0 = 0u, T1 = 0u;   // two rows side by side: two independent chains of ballots
                    for (int bit = 31; bit >= 16; --bit) {
                        const unsigned t0 = T0 | (1u << bit), t1 = T1 | (1u << bit);
                        const int c0 = __popcll(__ballot(m0.x >= t0)) + __popcll(__ballot(m0.y >= t0)) + __popcll(__ballot(m0.z >= t0)) +
                                       __popcll(__ballot(m0.w >= t0));
                        const int c1 = __popcll(__ballot(m1.x >= t1)) + __popcll(__ballot(m1.y >= t1)) + __popcll(__ballot(m1.z >= t1)) +
                                       __popcll(__ballot(m1.w >= t1));
                        if (c0 >= a.K) T0 = t0;
                        if (c1 >= a.K) T1 = t1;
                    }
                    if (lane == 0) { sTau[ra] = pan_bound(T0); sTau[ra + 1] = pan_bound(T1); }
                }
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    tau[rb] = sTau[rb * 16 + u];
#pragma unroll
                    for (int i = 0; i < NTW; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) nh[rb] += acc[rb][i][r] >= tau[rb] ? 1 : 0;
                }
            }
            PAN_STAMP(2 + 8 * (p < 3 ? p : 3) + 3)
            // ---- candidates at or above tau -> the row's list.  One reservation per lane and row block; a lane whose range would
            //      pass the end of the list writes nothing and raises the panel's overflow flag.
            //      The writes are straight-line: per score one compare, skipped when no lane of the wave matches, else an exec-masked
            //      8-byte LDS store of (~id, score bits) and the bump of the lane's slot address.  (Written as asm: left to the
            //      compiler this loop keeps the compare masks of the counting loop alive in SGPRs and spills accumulators to scratch
            //      to build 64-bit store operands.)
#pragma unroll 1
            for (int attempt = 0;; ++attempt) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rb * 16 + u;
                int k = -1;
                if (nh[rb]) {
                    k = atomicAdd(&sCnt[row], nh[rb]);
                    if (k + nh[rb] > kPanCap) { k = -1; sFlag[p & 1] = 1; }
                }
                if (__ballot(k >= 0)) {
                    const float tau_w = k >= 0 ? tau[rb] : INFINITY;      // lanes without a reservation match nothing
                    unsigned addr = lds_list + (unsigned)(row * kPanCap + (k >= 0 ? k : 0)) * 8u;
#pragma unroll
                    for (int i = 0; i < NTW; ++i) {
                        unsigned nid = ~(unsigned)(pbase + 16 * (w + 8 * i) + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            unsigned long long sv;
                            asm volatile("v_cmp_ge_f32 vcc, %3, %4\n\t"
                                         "s_cbranch_vccz .Lpan_skip%=\n\t"
                                         "s_and_saveexec_b64 %2, vcc\n\t"
                                         "ds_write2_b32 %0, %1, %3 offset1:1\n\t"
                                         "v_add_u32 %0, 8, %0\n\t"
                                         "s_mov_b64 exec, %2\n"
                                         ".Lpan_skip%=:\n\t"
                                         "v_add_u32 %1, -1, %1"
                                         : "+v"(addr), "+v"(nid), "=&s"(sv)
                                         : "v"(acc[rb][i][r]), "v"(tau_w)
                                         : "vcc", "memory");
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();   // ---- barrier B
            safe = sFlag[p & 1] != 0;
            if (!safe || attempt == 1) break;
            // ---- a list would overflow: the bound was too low for this panel (scores that grow with the item id, a panel of
            //      systematically better items).  Raise it, per row, to a key T with K <= #(list and panel entries >= T) <= 256 --
            //      a bisection on the key space that counts straight from the registers -- cut the lists to it and collect again.
            //      Rows tied beyond that at their K-th key are left to the safe form.
            if (tid < R) {
                const unsigned klo = score_key(sTau[tid]);
                sCLo[tid] = sCnt[tid] > kPanCap ? 0x7fffffff : 0;      // (sCnt: every lane's hits were added, so it is the exact count at tau)
                sCnt[tid] = sCnt0[tid];
                sLo[tid] = klo < 0x00800000u ? 0x00800000u : klo;
                sHi[tid] = 0u;
                sTot[tid] = 0;
            }
            if (tid == 0) { sFlag[p & 1] = 0; sFail = 0; }
            __syncthreads();
            // the search ends at the largest key in sight: the panel's maximum per row (registers) and the list's (its owner)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int row = rb * 16 + u;
                if (sCLo[row] > 256) {
                    float m = -INFINITY;
#pragma unroll
                    for (int i = 0; i < NTW; ++i) m = fmaxf(m, fmaxf(fmaxf(acc[rb][i][0], acc[rb][i][1]), fmaxf(acc[rb][i][2], acc[rb][i][3])));
                    if (m != -INFINITY) atomicMax(&sHi[row], score_key(m));
                }
            }
            for (int q = 0; q < 2 * RB; ++q) {
                const int r = (q >> 1) * 16 + 2 * w + (q & 1);
                if (sCLo[r] > 256) {   // (wave-uniform)
                    const int n0 = sCnt0[r];
                    unsigned km = 0u;
#pragma unroll
                    for (int j = 0; j < kPanCap / 64; ++j)
                        if (j * 64 + lane < n0) { const unsigned k = (unsigned)(pan_comp(sList[r * kPanCap + j * 64 + lane]) >> 32); km = k > km ? k : km; }
                    if (km) atomicMax(&sHi[r], km);
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int it = 0; it < 40; ++it) {
                if (tid == 0) sAct[it & 1] = 0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int row = rb * 16 + u;
                    const unsigned lo = sLo[row], hi = sHi[row];
                    if (lo < hi && sCLo[row] > 256) {
                        const unsigned mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);
                        const float fm = key_score(mid);   // (a NaN pattern above +inf: nothing compares >= it, count 0)
                        int c = 0;
#pragma unroll
                        for (int i = 0; i < NTW; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) c += acc[rb][i][r] >= fm ? 1 : 0;
                        if (c) atomicAdd(&sTot[row], c);
                    }
                }
                __syncthreads();
                for (int q = 0; q < 2 * RB; ++q) {
                    const int r = (q >> 1) * 16 + 2 * w + (q & 1);
                    const unsigned lo = sLo[r], hi = sHi[r];
                    if (lo < hi && sCLo[r] > 256) {   // (wave-uniform)
                        const unsigned mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);
                        const int n0 = sCnt0[r];
                        int c = sTot[r];
#pragma unroll
                        for (int j = 0; j < kPanCap / 64; ++j) {
                            const bool ge = j * 64 + lane < n0 && (unsigned)(pan_comp(sList[r * kPanCap + j * 64 + lane]) >> 32) >= mid;
                            c += __popcll(__ballot(ge));
                        }
                        if (lane == 0) {
                            if (c >= a.K) { sLo[r] = mid; sCLo[r] = c; } else sHi[r] = mid - 1u;
                            sTot[r] = 0;
                            const unsigned nlo = c >= a.K ? mid : lo, nhi = c >= a.K ? hi : mid - 1u;
                            if (nlo < nhi && (c >= a.K ? c : sCLo[r]) > 256) sAct[it & 1] = 1;
                        }
                    }
                }
                __syncthreads();
                if (!sAct[it & 1]) break;
            }
            // the lists cut to the new bounds (rows that did not search keep theirs)
            for (int q = 0; q < 2 * RB; ++q) {
                const int r = (q >> 1) * 16 + 2 * w + (q & 1);
                if (sCLo[r] == 0) continue;               // this row fitted
                if (sCLo[r] > 256) { if (lane == 0) sFail = 1; continue; }
                const unsigned T = sLo[r];
                const int n0 = sCnt0[r];
                unsigned long long e[kPanCap / 64];
                bool keep[kPanCap / 64];
#pragma unroll
                for (int j = 0; j < kPanCap / 64; ++j) {
                    e[j] = j * 64 + lane < n0 ? sList[r * kPanCap + j * 64 + lane] : 0ULL;
                    keep[j] = j * 64 + lane < n0 && (unsigned)(pan_comp(e[j]) >> 32) >= T;
                }
                int base = 0;
#pragma unroll
                for (int j = 0; j < kPanCap / 64; ++j) {
                    const unsigned long long m = __ballot(keep[j]);
                    if (keep[j]) sList[r * kPanCap + base + __popcll(m & ((1ULL << lane) - 1ULL))] = e[j];
                    base += __popcll(m);
                }
                if (lane == 0) { sCnt[r] = base; sCnt0[r] = base; sTau[r] = key_score(T); }   // (sCnt0: what the safe form would restore)
            }
            __syncthreads();
            if (sFail) break;                             // (safe is still set: the safe form takes the panel)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                tau[rb] = sTau[rb * 16 + u];
                nh[rb] = 0;
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nh[rb] += acc[rb][i][r] >= tau[rb] ? 1 : 0;
            }
            }
        }
        PAN_STAMP(2 + 8 * (p < 3 ? p : 3) + 4)
        if (safe) {
            // ---- safe form: the panel is computed AGAIN, one tile per wave and round (128 consecutive ids per round), every row
            //      cut back to its exact top K after each round.  Runtime loops, one accumulator block: nothing of it lives in
            //      the fast path's registers.
            if (tid < R) sCnt[tid] = sCnt0[tid];   // the slots written past the old count are dropped
            __syncthreads();
            auto exact_cut = [&]() {
                for (int q = 0; q < 2 * RB; ++q) {
                    const int r = (q >> 1) * 16 + 2 * w + (q & 1);
                    unsigned km;
                    const int m = pan_prune(sList + r * kPanCap, sCnt[r], a.K, a.K, 0, a.id_bits, lane, &km);
                    if (lane == 0) {
                        sCnt[r] = m;
                        sStrict[r] = m >= a.K;
                        if (m >= a.K) sTau[r] = key_score(km);
                    }
                }
            };
            exact_cut();
            __syncthreads();
#pragma unroll 1
            for (int i = 0; i < NTW; ++i) {
                const int id0 = pbase + 16 * (w + 8 * i) + 4 * g;
                int item = pbase + 16 * (w + 8 * i) + u;
                item = item < a.n_items ? item : a.n_items - 1;
                const float *brow = a.itabp + (size_t)item * dp + 4 * g;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int row = rb * 16 + u;
                    sel_f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                    for (int c = 0; c < DC; ++c) {
                        const sel_f32x4 b = *reinterpret_cast<const sel_f32x4 *>(brow + 16 * c);
                        const sel_f32x4 av = a_op(c, rb);
                        t = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, av.x, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, av.y, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, av.z, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, av.w, t, 0, 0, 0);
                    }
                    const unsigned nib = bits[row * RS + ((w * 4 + g) << 1) + (i >> 3)] >> (4 * (i & 7));
                    // exact: once K entries are held (tau = the K-th key) a later id with s == tau loses the tie -> strict compare;
                    // with fewer than K held everything unmasked so far is in the list and everything unmasked enters
                    const float tau = sTau[row];
                    const bool strict = sStrict[row] != 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float sc = t[r];
                        if (a.ubias) sc = ((sc + ub[rb]) + (id0 + r < a.n_items ? a.ibias[id0 + r] : 0.f)) + a.mean;
                        const bool masked = (nib >> r) & 1u;
                        if (!counted && !masked) {
#pragma unroll
                            for (int q = 0; q < NTG; ++q)
                                cntr[rb][q] += (id0 + r != tgt[q] && (sc > ts[rb][q] || (sc == ts[rb][q] && id0 + r < tgt[q]))) ? 1 : 0;
                        }
                        if (!masked && (!strict || sc > tau)) {
                            const int slot = atomicAdd(&sCnt[row], 1);
                            sList[row * kPanCap + slot] = pan_raw(sc, (unsigned)(id0 + r));
                        }
                    }
                }
                __syncthreads();
                exact_cut();
                __syncthreads();
            }
        } else if (p + 1 < n_panels) {
            // ---- tau for the next panel, by the rows' owners: the (coarse) K-th key of a list that has grown long.  No barrier:
            //      nothing of this is touched before barrier A of the next panel
            for (int q = 0; q < 2 * RB; ++q) {
                const int r = (q >> 1) * 16 + 2 * w + (q & 1);
                const int n = sCnt[r];
                if (n >= a.K && n > kPanCap / 2) {   // (short lists: the old bound is good enough for the next panel)
                    unsigned km;
                    const int limit = a.K + 32 < 256 ? a.K + 32 : 256;
                    const int m = pan_prune(sList + r * kPanCap, n, a.K, limit, 12, a.id_bits, lane, &km);
                    if (lane == 0) {
                        sCnt[r] = m;
                        const float nb = key_score(km);
                        if (nb > sTau[r]) sTau[r] = nb;
                    }
                }
            }
        }
        // the counts the next panel starts from (restored if it overflows), by the rows' owners
        if (lane < 2 * RB) { const int r = (lane >> 1) * 16 + 2 * w + (lane & 1); sCnt0[r] = sCnt[r]; }
        PAN_STAMP(2 + 8 * (p < 3 ? p : 3) + 5)
    }
    __syncthreads();

    // ---- target ranks: the 4 lane groups of a wave, then the 8 waves
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            int v = cntr[rb][t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0 && t < n_in) atomicAdd(&sRank[rb * 16 + u][t], v);
        }
    __syncthreads();
    for (int i = tid; i < R * n_in; i += kPanNT) {
        const int r = i / n_in, t = i % n_in;
        if (sUid[r] >= 0) a.target_rank[(size_t)(row0 + r) * a.n_targets + t] = sRank[r][t];
    }
    PAN_STAMP(34)
    if (a.stamps && tid == 0) {   // diagnostic: candidates held at the end (sum and maximum over the rows)
        int sm = 0, mx = 0;
        for (int r = 0; r < R; ++r) { sm += sCnt[r]; mx = sCnt[r] > mx ? sCnt[r] : mx; }
        a.stamps[(size_t)blockIdx.x * 36 + 32] = (unsigned long long)sm;
        a.stamps[(size_t)blockIdx.x * 36 + 33] = (unsigned long long)mx;
    }
    // ---- results: a wave per row, two rows side by side (two independent chains of lane exchanges)
#pragma unroll 1
    for (int rb = 0; rb < RB; ++rb) {
        unsigned long long c[2][4];
        int nn[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = rb * 16 + 2 * w + h;
            int n = row0 + r < a.nb ? sCnt[r] : 0;
            unsigned long long *row = sList + r * kPanCap;
            if (n > 256) {
                unsigned km;
                n = pan_prune(row, n, a.K, 256, 0, a.id_bits, lane, &km);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) c[h][q] = lane * 4 + q < n ? pan_comp(row[lane * 4 + q]) : 0ULL;
            nn[h] = n;
        }
        pan_sort256x2(c[0], c[1], lane);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = rb * 16 + 2 * w + h;
            if (row0 + r >= a.nb) continue;
            int *oid = a.top_ids + (size_t)(row0 + r) * a.K;
            float *osc = a.top_scores + (size_t)(row0 + r) * a.K;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = lane * 4 + q;
                if (e < a.K) {
                    const bool ok = e < nn[h] && c[h][q] != 0ULL;
                    oid[e] = ok ? (int)(~(unsigned)(c[h][q] & 0xffffffffULL)) : -1;
                    osc[e] = ok ? key_score((unsigned)(c[h][q] >> 32)) : -INFINITY;
                }
            }
        }
    }
    PAN_STAMP(35)
#undef PAN_STAMP
}

// ---------------------------------------------------------------- host side
inline int pan_dc(int d) { return d <= 32 ? 2 : d <= 64 ? 4 : d <= 128 ? 8 : 16; }
inline bool pan_supported(int n_items, int d, int K, int n_targets)
{
    return K >= 1 && K <= 256 && n_targets <= kPanMaxT && d >= 1 && d <= 256 && n_items >= 1;
}
// floats of scratch: the k-permuted item table (rows of 16 * DC floats), 16-byte aligned by the caller
inline long long pan_scratch_floats(int n_items, int d) { return (long long)n_items * 16 * pan_dc(d) + 4; }
// user rows per workgroup: 32 (one workgroup per CU, every item operand feeds 8 MFMAs) pays from 8192 users on when the sweep is
// operand-bound -- dim > 64 or catalogues of >= 65 536 items; measured 8 192 x 34 474 x 128 1.05 vs 1.18 ms, 16 384 x 131 072 x 64
// 4.43 vs 4.72, but 4 096 x 34 474 x 64 0.69 vs 0.47 and 8 192 x 34 474 x 64 0.73 vs 0.70 (rk_score_plan.panel_rows overrides)
inline int pan_rows(int nb, int n_items, int d, int n_targets = 1)
{
    if (n_targets > 1) return 16;   // (the multi-target instantiation spills at 32 rows: 2.3 vs 1.6 ms at 16384 x 34474 x 64)
    return nb >= 8192 && (d > 64 || n_items >= 65536) ? 32 : 16;
}
// 16-item tiles per wave and panel: the narrow form (8: 1024-item panels) while the catalogue fits one or two narrow panels
// (less padding), else 15 (1920 items)
inline int pan_ntw(int n_items) { return n_items <= 1024 ? 8 : 15; }

template <int NTW, int DC, int NTG, int RB>
inline hipError_t pan_launch_one(const PanArgs &a, hipStream_t s)
{
    static RkPerDeviceOnce attr_once;
    const size_t lds = pan_lds_bytes<NTW, DC, RB>();
    int attr_dev;
    if (attr_once.need(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(score_panel_kernel<NTW, DC, NTG, RB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_once.done(attr_dev);
    }
    hipLaunchKernelGGL((score_panel_kernel<NTW, DC, NTG, RB>), dim3((a.nb + 16 * RB - 1) / (16 * RB)), dim3(kPanNT), lds, s, a);
    return hipGetLastError();
}
template <int NTW, int DC, int RB>
inline hipError_t pan_launch_tg(const PanArgs &a, hipStream_t s)
{
    return a.n_targets <= 1 ? pan_launch_one<NTW, DC, 1, RB>(a, s) : pan_launch_one<NTW, DC, kPanMaxT, RB>(a, s);
}
template <int NTW, int RB>
inline hipError_t pan_launch_dc(const PanArgs &a, hipStream_t s)
{
#ifdef PAN_DEV   // development builds: one instantiation per row count (1920-item panels, dim <= 64, one target)
    return pan_launch_one<15, 4, 1, RB>(a, s);
#else
    switch (pan_dc(a.d)) {
    case 2: return pan_launch_tg<NTW, 2, RB>(a, s);
    case 4: return pan_launch_tg<NTW, 4, RB>(a, s);
    case 8: return pan_launch_tg<NTW, 8, RB>(a, s);
    default: return pan_launch_tg<NTW, 16, RB>(a, s);
    }
#endif
}

// scratch: pan_scratch_floats(n_items, d) floats, 16-byte aligned; rows / ntw / safe: the plan's shape knobs (rk_score_plan)
inline hipError_t score_panel_launch(PanArgs a, float *scratch, hipStream_t s, int rows, int ntw, int safe)
{
    int bits = 1;
    while (bits < 32 && (1LL << bits) < (long long)a.n_items) ++bits;
    a.id_bits = bits;
    a.force_safe = safe;
    // tuning builds, RK_PAN_STAMPS=1: the stamps go behind the scratch (the probe over-allocates it by 36 * 8 bytes per workgroup + 64)
    a.stamps = RK_TUNE_INT("RK_PAN_STAMPS", 0) ? reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(scratch + pan_scratch_floats(a.n_items, a.d)) + 63) & ~(uintptr_t)63) : nullptr;
    const int dc = pan_dc(a.d);
    const long long n4 = (long long)a.n_items * dc * 4;
    hipLaunchKernelGGL(pan_permute_kernel, dim3((unsigned)std::min<long long>(2048, (n4 + 255) / 256)), dim3(256), 0, s, a.itab, a.n_items, a.d, dc,
                       scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    a.itabp = scratch;
    if (ntw == 8) return pan_launch_dc<8, 1>(a, s);   // (narrow panels: 16-row workgroups only)
    return rows == 32 ? pan_launch_dc<15, 2>(a, s) : pan_launch_dc<15, 1>(a, s);
}
