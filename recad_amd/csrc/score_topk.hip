// Full-catalog scoring + top-K + target rank (recad/workflow/normal.py:57-93) on gfx950.
//   pass 1: scores[nb, I] = U_b . Items^T with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); the
//           instruction is a k-ordered fmaf chain, so a score is bit-identical to the scalar
//           loop  s = fmaf(u[k], v[k], s), k = 0..d-1  (oracle: orc_score_rows).
//   pass 2: one workgroup per user: mask seen items, rank of each target, radix-select of
//           the K-th largest score, ordered collection, bitonic sort by (score desc, id asc).
#include <algorithm>
#include <climits>

#include "gemm.h"
#include "score_panel.h"

// ---------------------------------------------------------------- pass 2
static constexpr int kMaxK = 256;
static constexpr int kTopkNT = 256;      // threads per user row
static constexpr int kInPassTargets = 4; // targets ranked inside the histogram pass

// One radix digit of a 256-bin histogram, scanned by wave 0 alone (4 bins per lane, shuffles only).
// Finds the bin holding the `need`-th largest entry: out[0] = bin (untouched when the histogram holds
// fewer than `need` entries), out[1] = how many entries of that bin are needed, out[2] = entries in or
// above the bin.  Called by every thread; ends with a workgroup barrier.
__device__ __forceinline__ void find_bin(const int *hist, int need, int tid, int *out)
{
    if (tid < 64) {
        const int4 v = *reinterpret_cast<const int4 *>(hist + 4 * tid);
        const int own = v.x + v.y + v.z + v.w;
        int inc = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_down(inc, o, 64);
            if (tid + o < 64) inc += n;
        }
        const int above = inc - own;
        if (above < need && need <= above + own) {
            int cum = above, bin = 4 * tid + 3, h = v.w;
            if (need > cum + h) { cum += h; bin = 4 * tid + 2; h = v.z; }
            if (need > cum + h) { cum += h; bin = 4 * tid + 1; h = v.y; }
            if (need > cum + h) { cum += h; bin = 4 * tid; h = v.x; }
            out[0] = bin; out[1] = need - cum; out[2] = cum + h;
        }
    }
    __syncthreads();
}

// One 256-thread workgroup per user row.
//   1. keys (monotone uint of the score, 0 = seen) -- staged in LDS when the row is short (LDS_ROW),
//      otherwise recomputed from the score row in L2 (seen items overwritten with -inf in place);
//   2. one pass: every thread takes the maximum of its (strided) keys and counts the rank of up to 4
//      targets.  The K-th largest of the 256 thread maxima is a lower bound of the K-th largest key
//      (each maximum is a distinct item), and for K around 100 only ~1.2 K keys lie above it.  Two
//      8-bit radix digits over the MAXIMA ONLY (256 LDS atomics per digit instead of one per item --
//      LDS atomics retire about one lane per clock) with a parallel suffix scan give that bound to 16 bits;
//   3. every key at or above the bound is collected unordered; if at most 256 of them, they are
//      rank-sorted (each thread counts the candidates above its own (key, ~id) composite) -- ties
//      resolve to the lowest item id exactly like the oracle's scan;
//   4. otherwise (large K, heavy ties, near-constant rows) the exact path: 8-bit radix rounds over the
//      candidates refine the threshold, leaving early as soon as the candidate set fits; with all 32
//      bits resolved the ties are taken in id order.
// LDS: 3 KB + the key row, so 8 workgroups per CU hide the dependent loads of the prologue.
template <bool LDS_ROW>
__global__ __launch_bounds__(kTopkNT) void topk_rows_kernel(float *__restrict__ scores, long long ld, int n_items, const int *__restrict__ user_ids,
                                                            const int *__restrict__ seen_ptr, const int *__restrict__ seen_idx, int K,
                                                            int *__restrict__ top_ids, float *__restrict__ top_scores,
                                                            const int *__restrict__ targets, int n_targets,
                                                            float *__restrict__ target_score, int *__restrict__ target_rank)
{
    constexpr int NT = kTopkNT, NW = NT / 64;
    __shared__ __attribute__((aligned(16))) int hist[256];
    __shared__ __attribute__((aligned(16))) int hist2[256];
    __shared__ unsigned long long sel[kMaxK];
    __shared__ int sh_i[8];
    __shared__ int wtot[NW];
    __shared__ int tcount[kInPassTargets][NW];
    extern __shared__ __attribute__((aligned(16))) unsigned lds_keys[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x;
    float *grow = scores + (size_t)b * (size_t)ld;
    auto key_at = [&](int i) -> unsigned { return LDS_ROW ? lds_keys[i] : score_key(grow[i]); };
    // f(item, key) over this thread's share of the row: four consecutive items per 16-byte LDS read
    // (the LDS row is padded with excluded keys to a multiple of 4), or a strided walk of the row in L2.
    // Any disjoint partition of the items serves the thread-maxima bound.
    auto for_keys = [&](auto f) {
        if (LDS_ROW) {
            const int n4 = (n_items + 3) >> 2;
            for (int q = tid; q < n4; q += NT) {
                const uint4 v = *reinterpret_cast<const uint4 *>(lds_keys + 4 * q);
                f(4 * q, v.x); f(4 * q + 1, v.y); f(4 * q + 2, v.z); f(4 * q + 3, v.w);
            }
        } else {
            // sixteen strided loads in flight per thread (the row is streamed from L2 / the Infinity Cache)
            for (int i0 = tid; i0 < n_items; i0 += 16 * NT) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int i = i0 + j * NT;
                    // (an UNCONDITIONAL load of a clamped index: written as `i < n_items ? grow[i] : -inf` the load sat under the same
                    //  condition as its use below, and with the branch-free score_key the compiler merged the two -- every load inside
                    //  its use's block, one round trip per item instead of one per sixteen: 16 384 x 34 474 rows 0.9 -> 1.5 ms,
                    //  profiles/r06l_unfused_ab.txt)
                    v[j] = grow[min(i, n_items - 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int i = i0 + j * NT;
                    if (i < n_items) f(i, score_key(v[j]));
                }
            }
        }
    };

    // the seen list hangs off two dependent loads: start them before the row is streamed
    const int u = user_ids[b];
    const int seen_b = seen_ptr[u], seen_e = seen_ptr[u + 1];
    const int seen_first = seen_b + tid < seen_e ? seen_idx[seen_b + tid] : -1;
    // target scores before masking (normal.py:83-85); the first four are ranked inside pass A, so
    // their (wave-uniform) loads are started here as well
    for (int t = tid; t < n_targets; t += NT) target_score[(size_t)b * n_targets + t] = grow[targets[t]];
    const int n_in = min(n_targets, kInPassTargets);
    unsigned tkey[kInPassTargets];
    int tgt[kInPassTargets];
#pragma unroll
    for (int t = 0; t < kInPassTargets; ++t) {
        tgt[t] = t < n_in ? targets[t] : -1;
        tkey[t] = t < n_in ? score_key(grow[tgt[t]]) : 0xffffffffu;
    }
    hist[tid] = 0;
    hist2[tid] = 0;
    if (tid == 0) { sh_i[0] = -1; sh_i[1] = 0; sh_i[2] = 0; sh_i[3] = 0; sh_i[4] = 0; }
    if (LDS_ROW) {
        if ((((uintptr_t)grow) & 15) == 0) {
            // four 16-byte loads per thread in flight
            for (int i0 = tid * 4; i0 + 3 < n_items; i0 += NT * 16) {
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + j * NT * 4;
                    v[j] = (i + 3 < n_items) ? *reinterpret_cast<const float4 *>(grow + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + j * NT * 4;
                    if (i + 3 < n_items)
                        *reinterpret_cast<uint4 *>(lds_keys + i) =
                            make_uint4(score_key(v[j].x), score_key(v[j].y), score_key(v[j].z), score_key(v[j].w));
                }
            }
            for (int i = (n_items & ~3) + tid; i < n_items; i += NT) lds_keys[i] = score_key(grow[i]);
        } else {
            for (int i = tid; i < n_items; i += NT) lds_keys[i] = score_key(grow[i]);
        }
        if (tid < ((4 - (n_items & 3)) & 3)) lds_keys[n_items + tid] = 0u;  // pad to a multiple of 4
        __syncthreads();
        if (seen_first >= 0) lds_keys[seen_first] = 0u;
        for (int k = seen_b + NT + tid; k < seen_e; k += NT) lds_keys[seen_idx[k]] = 0u;
    } else {
        // every thread's reads of the target scores (above) must have completed before any thread overwrites a
        // seen item in place: a target may itself be in the user's seen list (full_catalog_topk takes any users)
        __syncthreads();
        if (seen_first >= 0) grow[seen_first] = -INFINITY;
        for (int k = seen_b + NT + tid; k < seen_e; k += NT) grow[seen_idx[k]] = -INFINITY;
        __threadfence_block();
    }
    __syncthreads();

    // ---- pass A: thread maxima + rank of the first targets:  #(s > st) + #(s == st and id < target)
    unsigned mx = 0u;
    if (n_in == 0) {
        for_keys([&](int, unsigned k) { mx = max(mx, k); });
    } else if (n_in == 1) {
        const int tg = tgt[0];
        const unsigned kt = tkey[0];
        int c = 0;
        for_keys([&](int i, unsigned k) {
            mx = max(mx, k);
            c += (k != 0u && i != tg && (k > kt || (k == kt && i < tg))) ? 1 : 0;
        });
        c = (int)wave_sum((float)c);  // counts < 2^24: exact in fp32
        if (lane == 0) tcount[0][w] = c;
    } else {
        int tc[kInPassTargets] = {0, 0, 0, 0};
        for_keys([&](int i, unsigned k) {
            mx = max(mx, k);
#pragma unroll
            for (int t = 0; t < kInPassTargets; ++t)
                tc[t] += (k != 0u && i != tgt[t] && (k > tkey[t] || (k == tkey[t] && i < tgt[t]))) ? 1 : 0;
        });
#pragma unroll
        for (int t = 0; t < kInPassTargets; ++t) {
            const int c = (int)wave_sum((float)tc[t]);
            if (lane == 0) tcount[t][w] = c;
        }
    }
    if (mx != 0u) atomicAdd(&hist[mx >> 24], 1);
    __syncthreads();
    if (tid < n_in) {
        int tot = 0;
        for (int q = 0; q < NW; ++q) tot += tcount[tid][q];
        target_rank[(size_t)b * n_targets + tid] = tot;
    }
    // remaining targets: one pass each
    for (int t = kInPassTargets; t < n_targets; ++t) {
        const int tg = targets[t];
        const unsigned kt = score_key(target_score[(size_t)b * n_targets + t]);
        int c = 0;
        for (int i = tid; i < n_items; i += NT) {
            const unsigned k = key_at(i);
            c += (k != 0u && i != tg && (k > kt || (k == kt && i < tg))) ? 1 : 0;
        }
        c = (int)wave_sum((float)c);
        __syncthreads();
        if (lane == 0) wtot[w] = c;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int q = 0; q < NW; ++q) tot += wtot[q];
            target_rank[(size_t)b * n_targets + t] = tot;
        }
    }

    // ---- 16-bit lower bound of the K-th largest thread maximum (fewer than K non-empty threads:
    //      every valid key is a candidate)
    unsigned cthr = 1u;  // candidates: keys >= cthr (1 = every valid key)
    find_bin(hist, K, tid, sh_i);
    if (sh_i[0] >= 0) {
        const unsigned d1 = (unsigned)sh_i[0];
        const int need1 = sh_i[1];
        if (mx != 0u && (mx >> 24) == d1) atomicAdd(&hist2[(mx >> 16) & 255u], 1);
        __syncthreads();
        find_bin(hist2, need1, tid, sh_i);
        cthr = max((d1 << 24) | ((unsigned)sh_i[0] << 16), 1u);
    }

    // ---- optimistic unordered collection of the candidates
    for_keys([&](int i, unsigned k) {
        if (k >= cthr) {
            const int p = atomicAdd(&sh_i[3], 1);
            if (p < kMaxK) sel[p] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)i);
        }
    });
    __syncthreads();
    int n_cand = sh_i[3];  // >= min(K, valid keys) by construction
    int n_sel = n_cand;
    if (n_cand > kMaxK) {
        // ---- exact path: radix rounds over the candidates, most significant digit first
        unsigned prefix = 0u, mask = 0u;
        int need = K, shift = 32;
        __syncthreads();
        if (tid == 0) sh_i[3] = 0;
        while (n_cand > kMaxK && shift > 0) {
            shift -= 8;
            hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n_items; i += NT) {
                const unsigned k = key_at(i);
                if (k >= cthr && (k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
            }
            __syncthreads();
            find_bin(hist, need, tid, sh_i);  // more than 256 >= K candidates: always found
            const int n_gt = K - need;  // keys strictly above the old prefix range
            prefix |= (unsigned)sh_i[0] << shift;
            mask |= 255u << shift;
            n_cand = n_gt + sh_i[2];
            need = sh_i[1];
            __syncthreads();
        }
        if (n_cand <= kMaxK) {
            for (int i = tid; i < n_items; i += NT) {
                const unsigned k = key_at(i);
                if (k >= cthr && (k & mask) >= prefix) {
                    const int p = atomicAdd(&sh_i[3], 1);
                    sel[p] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)i);
                }
            }
            n_sel = n_cand;
        } else {
            // all 32 bits resolved: prefix is the K-th key itself and more than 256 keys are >= it:
            // take the K - need larger ones in any order and the `need` lowest-id ties in id order
            const unsigned T = prefix;
            const int n_gt_slots = K - need;
            for (int base = 0; base < n_items; base += NT) {
                const int i = base + tid;
                const unsigned k = i < n_items ? key_at(i) : 0u;
                if (k > T) {
                    const int p = atomicAdd(&sh_i[3], 1);
                    sel[p] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)i);
                }
                const bool eq = k == T && k != 0u;
                const unsigned long long m = __ballot(eq);
                if (lane == 0) wtot[w] = __popcll(m);
                __syncthreads();
                int pre = __popcll(m & ((1ULL << lane) - 1ULL));
                for (int ww = 0; ww < w; ++ww) pre += wtot[ww];
                const int eq_base = sh_i[4];
                if (eq) {
                    const int idx = eq_base + pre;
                    if (idx < need) sel[n_gt_slots + idx] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)i);
                }
                __syncthreads();
                if (tid == 0) {
                    int tot = eq_base;
                    for (int q = 0; q < NW; ++q) tot += wtot[q];
                    sh_i[4] = tot;
                }
                __syncthreads();
                if (sh_i[4] >= need && base + NT < n_items) {
                    // ties are complete; only larger keys remain to be collected
                    for (int i2 = base + NT + tid; i2 < n_items; i2 += NT) {
                        const unsigned k2 = key_at(i2);
                        if (k2 > T) {
                            const int p = atomicAdd(&sh_i[3], 1);
                            sel[p] = ((unsigned long long)k2 << 32) | (unsigned)(~(unsigned)i2);
                        }
                    }
                    break;
                }
            }
            n_sel = K;
        }
        __syncthreads();
    }
    // rank sort: composites are distinct (item id in the low word), so ranks are a permutation
    if (tid < n_sel) {
        const unsigned long long mine = sel[tid];
        int rank = 0;
        for (int j = 0; j < n_sel; ++j) rank += sel[j] > mine ? 1 : 0;
        if (rank < K) {
            top_ids[(size_t)b * K + rank] = (int)(~(unsigned)(mine & 0xffffffffULL));
            top_scores[(size_t)b * K + rank] = key_score((unsigned)(mine >> 32));
        }
    }
    for (int k = n_sel + tid; k < K; k += NT) {
        top_ids[(size_t)b * K + k] = -1;
        top_scores[(size_t)b * K + k] = -INFINITY;
    }
}

// ---------------------------------------------------------------- pass 2, short rows: ONE WAVE per user row, the row in registers
// Rows of up to NQ x 64 floats (NQ = 16 ... 96: <= 6 144 items -- ml1m, Amazon-game, dev): no workgroup barrier, no LDS
// atomics on the item path, every reduction a wave reduction.  A workgroup is four independent waves.
//   0. one wave per row, six waves per SIMD (79 VGPRs at NQ = 58), so all 5 893 rows of the headline evaluation are resident at
//      once.  Measured alternatives (scripts/topk_wave_probe.sh, profiles/r05_topk_wave_probe*.txt): the kernel is bound by
//      instruction ISSUE -- ~3 600 wave instructions per row (2 250 vector, 900 scalar, 200 LDS, 100 memory), 41 us at any
//      occupancy from 3 to 6 -- so what pays is fewer instructions per item, not more waves: persistent waves that walk two rows and
//      load the next row under the current one (two rows' registers: 3 waves per SIMD) took 52 us; the dispatcher starts ~3 400
//      waves in the first two microseconds and the rest at ~400 per microsecond, which is where 12 of the 41 us go;
//   1. lane l holds items l, l + 64, l + 128, ... (NQ coalesced dword loads through a buffer descriptor of the row: any row
//      alignment, one address register, past-the-end lanes read 0); the excluded items are a TRANSPOSED per-wave LDS bitmap --
//      bit q of lane l's word = item 64 q + l -- so a lane reads its NQ bits as NQ / 32 dwords and tests them with immediate
//      masks; positions past the row's end start out excluded.  Keys: monotone uint of the score, 0 = excluded;
//   2. target ranks: #(key > kt) + #(key == kt and id < target): two compares per key, ties through a rare slow path;
//   3. lower bound of the K-th largest key: the K-th largest of the 256 GROUP MAXIMA (group = lane x (q mod 4); every maximum
//      is a distinct item), found to 20 bits by a bitwise search with ballots -- ~125 keys lie at or above it for K = 100,
//      whatever the row length;
//   4. candidates (keys >= bound) are compacted into a per-wave list (per-lane counts, wave prefix sum); if more than 256
//      (large K, heavy ties, near-constant rows): the EXACT K-th key by a 32-step bitwise search over all keys, everything above
//      it in any order and the needed ties in ascending item id (ballot order);
//   5. rank sort of the <= 256 composites (key << 32 | ~id: distinct, so ranks are a permutation; ties resolve to the lowest
//      item id exactly like the oracle's scan), permuted in LDS, written out coalesced.
// The score matrix is only READ (the row-per-workgroup kernel below overwrites seen items in place).
// a copy of a per-lane value the compiler cannot hoist or merge with other copies (item ids are formed where they are used and
// die there: left alone, all NQ of them stay alive across the passes)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// a compile-time constant as a scalar register the compiler cannot fold, hoist or merge: `x - scalar_here(c)` is ONE vector
// instruction formed where it stands (NQ item ids precomputed ahead of the branches they are used under cost NQ registers)
__device__ __forceinline__ unsigned scalar_here(unsigned c) { unsigned r; asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "i"(c)); return r; }
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(v, o, 64);
        if (lane >= o) v += n;
    }
    return v;
}
// score_key() (score_panel.h: the ONE encoder of every selection kernel) with the seen bit folded in: 0 for an excluded item.
__device__ __forceinline__ unsigned score_key_sel(float s, bool excluded) { return excluded ? 0u : score_key(s); }
// The wave kernel's per-item form: score_key() WITHOUT its final clamp, the exclusion as an OR.  `ones` is 0 or 0xffffffff (the
// item's seen bit sign-extended: one v_bfe_i32); OR-ed into the score's bits it makes them all ones, whose key is 0.  What score_key
// clamps to 0 -- -inf (0x007fffff) and negative NaNs (below) -- stays where the monotone map puts it: below kKeyValid, the key of
// -FLT_MAX.  Every compare of the kernel that means "a rankable item" is against kKeyValid instead of 1, so the results are the
// clamped encoder's, at 4.5 vector instructions per item instead of 6.5 + a scalar OR (the ISA of the round-5 loop: v_and + v_cmp_ne
// for the bit, v_cmp_gt against 0x800000, s_or, v_cndmask).
// c += (a < b) / (a <= b) / (a == b) per lane, a wave-uniform (an SGPR), b the lane's key: a compare into VCC and ONE add-with-carry
// (the compiler's form of `c += cond ? 1 : 0` is v_cmp + v_cndmask + v_add: three instructions at each of the kernel's three
// counting sites per item -- target greater / equal, candidates -- i.e. 174 of the ~2 250 vector instructions of a 3 702-item row)
__device__ __forceinline__ void count_lt(int &c, unsigned a, unsigned b) { asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(c) : "s"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void count_le(int &c, unsigned a, unsigned b) { asm("v_cmp_le_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(c) : "s"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void count_eq(int &c, unsigned a, unsigned b) { asm("v_cmp_eq_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(c) : "s"(a), "v"(b) : "vcc"); }
static constexpr unsigned kKeyValid = 0x00800000u;
__device__ __forceinline__ unsigned score_key_raw(float s, unsigned ones)
{
    const unsigned u = __float_as_uint(s + 0.0f) | ones;
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}

#ifndef TOPK_STAMP   // (scripts/topk_wave_probe.sh builds this kernel alone with per-phase cycle stamps)
#define TOPK_STAMP(i) do { } while (0)
#endif
#ifndef TOPK_WAVES_EU   // (probe builds: occupancy A/B)
#define TOPK_WAVES_EU(NQ) (NQ <= 58 ? 6 : NQ <= 72 ? 5 : 4), 8
#endif
#ifndef TOPK_WAVE_WG    // waves per workgroup
#define TOPK_WAVE_WG 4
#endif
template <int NQ>
__global__ __launch_bounds__(64 * TOPK_WAVE_WG) __attribute__((amdgpu_waves_per_eu(TOPK_WAVES_EU(NQ)))) void topk_wave_kernel(
    const float *__restrict__ scores, long long ld, int nb, int n_items, const int *__restrict__ user_ids, const int *__restrict__ seen_ptr,
    const int *__restrict__ seen_idx, int K, int *__restrict__ top_ids, float *__restrict__ top_scores, const int *__restrict__ targets,
    int n_targets, float *__restrict__ target_score, int *__restrict__ target_rank)
{
    constexpr int NW = (NQ + 31) / 32;   // bitmap dwords per lane
    __shared__ unsigned bm_all[TOPK_WAVE_WG][NW * 64];
    __shared__ unsigned long long sel_all[TOPK_WAVE_WG][kMaxK + 4];   // (+4: the rank loop reads whole groups of four slots)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = (int)blockIdx.x * TOPK_WAVE_WG + w;
    if (b >= nb) return;   // (waves are independent: no workgroup barrier anywhere below)
    TOPK_STAMP(0);
    unsigned *bm = bm_all[w];
    // composite of a candidate = key << 32 | ~item id, one 8-byte LDS slot each, read back as 64-bit words.  WRITTEN as two dword
    // stores whose adjacency the compiler cannot see (the upper half's index goes through `hi`, an opaque 1): one 64-bit store wants
    // the key in the upper half of an aligned register PAIR -- every key register was then allocated as half of a pair, doubling
    // the keys' footprint, and a private copy of the key is coalesced away where the store is the key's last use
    unsigned long long *sel = sel_all[w];
    unsigned *sel32 = reinterpret_cast<unsigned *>(sel_all[w]);
    const int hi = opaque(1);
    auto sel_put = [&](int slot, unsigned not_id, unsigned k) { sel32[2 * slot] = not_id; sel32[2 * slot + hi] = k; };
    // the candidates' form: the slot as an LDS BYTE ADDRESS that the lane bumps by 8, both dwords through it (immediate offset 4) --
    // with a slot index each of the two stores formed its own address (v_lshl_add_u32 twice, the index's v_add: 6 vector instructions
    // per executed candidate branch against 2 here, and that branch runs for ~50 of a 3 702-item row's 58 item positions).  The
    // wave_lds_sync() in front of the list's first read drains lgkmcnt (its fence), so the compiler need not see these stores.
    const unsigned sel_base = (unsigned)reinterpret_cast<unsigned long long>(sel32);   // (low half of the flat address = the LDS offset)
    auto sel_put_at = [&](unsigned addr, unsigned not_id, unsigned k) {
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4" : : "v"(addr), "v"(not_id), "v"(k) : "memory");
    };

    const float *grow = scores + (size_t)b * (size_t)ld;
    // the row through a buffer descriptor of exactly its bytes: lanes past the row's end read 0 instead of faulting (those positions
    // are excluded).  The whole displacement 4 * lane + 256 q is the VECTOR offset (the compiler keeps what fits -- 4 095 -- in the
    // instruction's offset field and bumps the address register every 16 loads): the range check of a raw buffer covers vector
    // offset + instruction offset and NOT the scalar offset, where the 256 q used to go -- rows then read up to NQ * 256 bytes whatever
    // their length, the last row of a matrix that ends on a mapping boundary past its allocation (round-5 review)
    const unsigned long long gaddr = reinterpret_cast<unsigned long long>(grow);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gaddr >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)gaddr)),
        (short)0, n_items * 4, 0x00020000);
    float v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4 + q * 256, 0, 0));
    TOPK_STAMP(1);
    // the seen list hangs off two dependent (scalar) loads; the row is already in flight
    const int u = user_ids[b];
    const int seen_b = seen_ptr[u], seen_e = seen_ptr[u + 1];
    {
        const int nq = (n_items - lane + 63) >> 6;   // this lane's items: q < nq
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int n = nq - 32 * i;
            bm[i * 64 + lane] = n >= 32 ? 0u : n <= 0 ? 0xffffffffu : (0xffffffffu << n);
        }
    }
    wave_lds_sync();
    for (int k = seen_b + lane; k < seen_e; k += 64) {
        const unsigned i = (unsigned)seen_idx[k];
        if (i < (unsigned)(NQ * 64)) atomicOr(&bm[(i >> 11) * 64 + (i & 63u)], 1u << ((i >> 6) & 31u));
    }
    // target scores before masking (normal.py:83-85); the first target's id and score are wanted right after the keys: their
    // (dependent) loads start here, under the row's
    for (int t = lane; t < n_targets; t += 64) target_score[(size_t)b * n_targets + t] = grow[targets[t]];
    const int tg0 = n_targets > 0 ? targets[0] : 0;
    const float ts0 = n_targets > 0 ? grow[tg0] : 0.f;
    wave_lds_sync();
    unsigned excl[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) excl[i] = bm[i * 64 + lane];
    TOPK_STAMP(2);
    unsigned key[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) key[q] = score_key_raw(v[q], (unsigned)__builtin_amdgcn_sbfe((int)excl[q >> 5], q & 31, 1));
    TOPK_STAMP(3);
    {
        // ---- target ranks:  #(s > st) + #(s == st and id < target), the target itself and excluded items not counted.
        for (int t = 0; t < n_targets; ++t) {
            const int tg = t == 0 ? tg0 : targets[t];
            // (score_key's clamped key: 0 for a target whose own score is -inf / a negative NaN.  Then every rankable item counts:
            //  compared against the last unrankable key, since the row's keys are unclamped)
            const unsigned kt0 = score_key(t == 0 ? ts0 : grow[tg]);
            const unsigned kt = (unsigned)__builtin_amdgcn_readfirstlane((int)(kt0 != 0u ? kt0 : kKeyValid - 1u));   // (uniform: an SGPR operand below)
            // per-lane counters (compare + add-with-carry), one wave sum of both at the end: counting ballots on the scalar side kept
            // 2 NQ mask registers alive (the sums are re-associated) and spilled
            // (per-lane counters, one wave sum of both at the end.  Counting on the scalar side instead -- v_cmp, s_bcnt1, s_add per
            //  item: one vector instruction instead of two -- measured the same on one box, alternating: 36.5-38.4 us in all three
            //  forms, profiles/r06h_topk_count_ab.txt; the kernel is not bound by these counters)
            int c_gt = 0, c_eq = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                count_lt(c_gt, kt, key[q]);
                count_eq(c_eq, kt, key[q]);
            }
            const int packed = wave_sum_i(c_gt + (c_eq << 16));   // (both < 2^13 per wave)
            int cnt = packed & 0xffff;
            const int n_eq = packed >> 16;
            // the target itself is one of the keys equal to kt unless it is excluded: only OTHER items with its score need the id test
            const int self = (unsigned)tg < (unsigned)(NQ * 64) && !((bm[(tg >> 11) * 64 + (tg & 63)] >> ((tg >> 6) & 31)) & 1u) ? 1 : 0;
            if (n_eq > self && kt0 != 0u) {
                int ce = 0;
                const int lane_here = opaque(lane);   // (loop-invariant otherwise: NQ hoisted `lane | 64 q` registers live across the target loop)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    ce += (key[q] == kt && lane_here + 64 * q < tg) ? 1 : 0;
                    if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
                cnt += wave_sum_i(ce);
            }
            if (lane == 0) target_rank[(size_t)b * n_targets + t] = cnt;
        }
        TOPK_STAMP(4);
        // ---- lower bound of the K-th largest key from the 256 group maxima
        unsigned m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < NQ; q += 8)   // (two keys per v_max3_u32)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (q + 4 + j < NQ) m[j] = max(max(m[j], key[q + j]), key[q + 4 + j]);
                else if (q + j < NQ) m[j] = max(m[j], key[q + j]);
            }
        auto count_ge = [&](unsigned x) -> int {
            int c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) c += __popcll(__ballot(m[j] >= x));
            return c;
        };
        unsigned cthr = kKeyValid;   // candidates: keys >= cthr (kKeyValid = every rankable item)
        if (count_ge(kKeyValid) >= K) {
            unsigned T = 0u;
#pragma unroll 1
            for (int bit = 31; bit >= 12; --bit) {
                const unsigned cand = T | (1u << bit);
                if (count_ge(cand) >= K) T = cand;
            }
            cthr = max(T, kKeyValid);
        }
        TOPK_STAMP(5);
        // ---- candidates: per-lane counts, wave prefix sum, compaction into the wave's list
        auto compact = [&](unsigned thr0) -> int {   // the list <- composites of the keys >= thr0, if they fit; returns how many
            int cl = 0;
            const unsigned thr_s = (unsigned)__builtin_amdgcn_readfirstlane((int)thr0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                count_le(cl, thr_s, key[q]);
                if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            const int incl = wave_incl_scan_i(cl, lane);
            const int total = __shfl(incl, 63, 64);
            if (total <= kMaxK) {
                unsigned addr = sel_base + 8u * (unsigned)(incl - cl);   // this lane's next list slot
                // (a private copy of the threshold: sharing the first pass's NQ compare results kept 2 NQ scalar registers alive -- spills)
                const unsigned thr = (unsigned)__builtin_amdgcn_readfirstlane(opaque((int)thr0));
                const unsigned not_lane = ~(unsigned)lane;   // ~(lane + 64 q) = ~lane - 64 q
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (key[q] >= thr) {
                        sel_put_at(addr, not_lane - scalar_here(64 * q), key[q]);
                        addr += 8u;
                    }
            }
            return total;
        };
        int n_sel = compact(cthr);
        if (n_sel > kMaxK) {
            // ---- exact path: the K-th largest key itself (at least K keys are >= cthr here), bit by bit over ALL keys
            auto count_all_ge = [&](unsigned x) -> int {
                int c = 0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    c += __popcll(__ballot(key[q] >= x));
                    if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
                return c;
            };
            unsigned T = 0u;
#pragma unroll 1
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned cand = T | (1u << bit);
                if (count_all_ge(cand) >= K) T = cand;
            }
            // T >= cthr >= kKeyValid.  Everything above T in any order (fewer than K), then the ties in ascending item id = (q, lane) order
            const int n_gt = T == 0xffffffffu ? 0 : compact(T + 1u);   // keys > T
            const int need = K - n_gt;
            int before = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned long long eq = __ballot(key[q] == T);
                const int pos = before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(eq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)eq, 0u));
                if (key[q] == T && pos < need) sel_put(n_gt + pos, ~(unsigned)lane - scalar_here(64 * q), T);
                before += __popcll(eq);
            }
            n_sel = K;
        }
        n_sel = __builtin_amdgcn_readfirstlane(n_sel);   // (uniform by construction: the loops below run on the scalar side)
        if (lane < 4) sel[n_sel + lane] = 0ULL;          // pad to a whole group of four
        wave_lds_sync();
        TOPK_STAMP(6);
        // ---- rank sort: composites are distinct (item id in the low word), so ranks are a permutation
        unsigned long long mine[4];
        int rk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) mine[e] = lane + 64 * e < n_sel ? sel[lane + 64 * e] : ~0ULL;
        auto rank_all = [&](auto ne) {
            constexpr int NE = decltype(ne)::value;
            // four slots per trip (the list is padded with zero composites: never greater than anything), their reads issued
            // together ahead of the compares: one slot per trip exposed an LDS round trip per comparison
#pragma clang loop unroll(disable)
            for (int j = 0; j < n_sel; j += 4) {
                unsigned long long c[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = sel[j + i];   // (uniform addresses: broadcast reads)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < NE; ++e) rk[e] += c[i] > mine[e] ? 1 : 0;
            }
        };
        if (n_sel <= 64) rank_all(std::integral_constant<int, 1>{});
        else if (n_sel <= 128) rank_all(std::integral_constant<int, 2>{});
        else rank_all(std::integral_constant<int, 4>{});
        TOPK_STAMP(7);
        wave_lds_sync();   // every lane has read the list: permute it in place
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (lane + 64 * e < n_sel) sel_put(rk[e], (unsigned)mine[e], (unsigned)(mine[e] >> 32));
        wave_lds_sync();
        for (int k = lane; k < K; k += 64) {
            int id = -1;
            float sc = -INFINITY;
            if (k < n_sel) {
                const unsigned long long c = sel[k];
                id = (int)(~(unsigned)c);
                sc = key_score((unsigned)(c >> 32));
            }
            top_ids[(size_t)b * K + k] = id;
            top_scores[(size_t)b * K + k] = sc;
        }
        TOPK_STAMP(8);
    }
}

extern "C" int rk_topk_rows_impl(float *scores, long long ld, int nb, int n_items, const int *user_ids, const int *seen_ptr,
                                 const int *seen_idx, int K, int *top_ids, float *top_scores, const int *targets,
                                 int n_targets, float *target_score, int *target_rank, hipStream_t s)
{
    if (K <= 0 || K > kMaxK) RK_FAIL(RK_EINVAL, "top-K: K must be in [1,%d]", kMaxK);
    if (n_targets < 0 || n_targets > 256 || (n_targets > 0 && (!targets || !target_score || !target_rank)))
        RK_FAIL(RK_EINVAL, "top-K: bad targets");
    if (ld < n_items) RK_FAIL(RK_EINVAL, "top-K: row stride %lld < n_items %d", ld, n_items);
    // short rows (ml1m 3 702, Amazon-game 5 600 items): one wave per row, the row in registers.  Measured on 5 893 x 3 702
    // (the headline evaluation): 37.9-39.6 us against 52.4 us for the row-per-workgroup kernel below (profiles/r05b_topk_ab.txt).
    static const int no_wave = RK_TUNE_INT("RK_TOPK_NO_WAVE", 0);   // A/B only
    if (!no_wave && n_items <= 96 * 64) {
        const int nq = (n_items + 63) / 64;
        const dim3 grid((nb + TOPK_WAVE_WG - 1) / TOPK_WAVE_WG), block(64 * TOPK_WAVE_WG);
#define RK_TOPK_WAVE(NQ)                                                                                                              \
    hipLaunchKernelGGL((topk_wave_kernel<NQ>), grid, block, 0, s, scores, ld, nb, n_items, user_ids, seen_ptr, seen_idx, K, top_ids, \
                       top_scores, targets, n_targets, target_score, target_rank)
        // NQ = ceil(n_items / 64) rounded up to an instantiated size (58 = the ml1m catalogue, 3 702 items, exactly)
        if (nq <= 16) RK_TOPK_WAVE(16);
        else if (nq <= 32) RK_TOPK_WAVE(32);
        else if (nq <= 48) RK_TOPK_WAVE(48);
        else if (nq <= 58) RK_TOPK_WAVE(58);
        else if (nq <= 64) RK_TOPK_WAVE(64);
        else if (nq <= 80) RK_TOPK_WAVE(80);
        else if (nq <= 88) RK_TOPK_WAVE(88);
        else RK_TOPK_WAVE(96);
#undef RK_TOPK_WAVE
        RK_CHECK_LAUNCH();
        return RK_OK;
    }
    const size_t row_bytes = ((size_t)n_items * sizeof(float) + 15) & ~(size_t)15;
    // LDS staging only pays while several workgroups still fit per CU (measured: a 138 KB row in LDS
    // is 2.8x SLOWER than streaming it from L2 -- one 4-wave workgroup per CU); ml1m-size rows tie.
    // measured on 8192 rows: staging wins up to ~40 KB rows (6000 items: 90 vs 99 us, 9000: 132 vs 140), loses
    // beyond (14000 items, 56 KB: 285 vs 202 us -- two workgroups per CU)
    static const size_t lds_row_max = (size_t)RK_TUNE_INT("RK_TOPK_LDS_KB", 40) * 1024;
    if (row_bytes <= lds_row_max) {
        static RkPerDeviceOnce attr_once;
        int attr_dev;
        if (attr_once.need(&attr_dev)) {
            RK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(topk_rows_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            attr_once.done(attr_dev);
        }
        // (one wave per short row -- NT = 64 -- was measured too: 333 vs 250 us for 5950 x 3702)
        hipLaunchKernelGGL((topk_rows_kernel<true>), dim3(nb), dim3(kTopkNT), row_bytes, s, scores, ld, n_items, user_ids, seen_ptr, seen_idx,
                           K, top_ids, top_scores, targets, n_targets, target_score, target_rank);
    } else {
        hipLaunchKernelGGL((topk_rows_kernel<false>), dim3(nb), dim3(kTopkNT), 0, s, scores, ld, n_items, user_ids, seen_ptr, seen_idx, K,
                           top_ids, top_scores, targets, n_targets, target_score, target_rank);
    }
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// The register-resident panel form (score_panel.h): K <= 256, at most 4 targets, dim <= 256.  It parallelises over the USERS only
// (16 or 32 per workgroup, the catalogue swept panel by panel), so it is the default for catalogues of >= 16 384 items AND enough
// users to fill the chip: >= 8192, or >= 4096 at dim <= 64.  Measured on MI355X against GEMM + selection (ms): 16 384 x 34 474 x 64
// 1.36 vs 1.91, 16 384 x 131 072 x 64 4.43 vs 6.76, 54 617 x 34 474 x 128 6.65 vs 7.77, 8 192 x 131 072 x 256 5.71 vs 6.49,
// 4 096 x 34 474 x 64 0.47 vs 0.51, 4 096 x 500 000 x 64 5.56 vs 6.44, 4 000 x 300 000 x 64 3.43 vs 3.89; and where it is NOT
// taken: 5 893 x 3 702 x 64 0.105 vs 0.098, 2 048 x 131 072 x 64 1.55 vs 1.02, 2 048 x 300 000 x 64 3.39 vs 2.19, 4 096 x 34 474 x
// 128 0.72 vs 0.64 (profiles/r04_score_corner.txt).
// Row stride of the GEMM path's score matrix: n_items rounded up to 32 floats.  Rows then start on 128-byte lines and every
// 16-column (64-byte) segment a wave of the GEMM stores is one aligned half line -- with the natural stride (3 702 floats: rows
// 8-byte aligned) each of those segments straddled two lines.  Costs 0.3 % more scratch at the ml1m size.
static inline long long score_ld(int n_items) { return ((long long)n_items + 31) & ~31LL; }

static bool panel_by_default(int nb, int n_items, int dim, int K, int n_targets)
{
    if (!pan_supported(n_items, dim, K, n_targets)) return false;
    // measured (profiles/r05i_score_thresholds.txt): with a full machine of workgroups (>= 8192 user rows) the panel form wins from 8192
    // items on (8192 x 8192 x 64: 219 vs 241 us; 8192 x 6144 x 64: 181 vs 161); with 4096 rows it only ties at dim <= 64 from 16384 items on
    // several targets: the per-target counters push the 32-row form over its register budget (r05i_score_targets.txt: 4 targets,
    // 8192 x 34474 x 256 1.97 ms against 1.83 for GEMM + selection; 16384 x 34474 x 64 with 16-row workgroups 1.63 against 1.92)
    if (n_targets > 1 && dim > 64) return false;
    if (nb >= 8192) return n_items >= kPanDefaultMinItems / 2;
    return n_items >= kPanDefaultMinItems && nb >= 4096 && dim <= 64;
}

RK_EXPORT int rk_score_topk_plan(int32_t nb, int32_t n_items, int32_t dim, int32_t K, int32_t n_targets, const rk_score_plan *request,
                                 rk_score_plan *out)
{
    if (!out || nb <= 0 || n_items <= 0 || dim <= 0) RK_FAIL(RK_EINVAL, "rk_score_topk_plan: bad arguments");
    if (K <= 0 || K > kMaxK) RK_FAIL(RK_EINVAL, "top-K: K must be in [1,%d]", kMaxK);
    if (n_targets < 0 || n_targets > 256) RK_FAIL(RK_EINVAL, "top-K: bad targets");
    rk_score_plan pl;
    memset(&pl, 0, sizeof(pl));
    pl.nb = nb; pl.n_items = n_items; pl.dim = dim; pl.K = K; pl.n_targets = n_targets;
    const int want = request ? request->path : RK_SCORE_AUTO;
    if (want != RK_SCORE_AUTO && want != RK_SCORE_GEMM && want != RK_SCORE_PANEL) RK_FAIL(RK_EINVAL, "rk_score_topk_plan: unknown path %d", want);
    if (want == RK_SCORE_PANEL && !pan_supported(n_items, dim, K, n_targets))
        RK_FAIL(RK_EINVAL, "rk_score_topk_plan: the panel form needs K <= 256, n_targets <= %d, dim <= 256", kPanMaxT);
    pl.path = want != RK_SCORE_AUTO ? want : (panel_by_default(nb, n_items, dim, K, n_targets) ? RK_SCORE_PANEL : RK_SCORE_GEMM);
    if (pl.path == RK_SCORE_PANEL) {
        const int rows = request ? request->panel_rows : 0, ntw = request ? request->panel_ntw : 0;
        if ((rows != 0 && rows != 16 && rows != 32) || (ntw != 0 && ntw != 8 && ntw != 15)) RK_FAIL(RK_EINVAL, "rk_score_topk_plan: panel_rows in {16, 32}, panel_ntw in {8, 15}");
        pl.panel_ntw = ntw ? ntw : pan_ntw(n_items);
        pl.panel_rows = pl.panel_ntw == 8 ? 16 : (rows ? rows : pan_rows(nb, n_items, dim, n_targets));
        pl.panel_safe = request && request->panel_safe ? 1 : 0;
        pl.scratch_floats = (int64_t)pan_scratch_floats(n_items, dim);   // the k-permuted item table
    } else {
        pl.ld_scores = (int32_t)score_ld(n_items);
        pl.scratch_floats = (int64_t)nb * pl.ld_scores;                 // the score matrix
    }
    *out = pl;
    return RK_OK;
}

RK_EXPORT int rk_score_topk(int32_t dim, const float *utab, int32_t nb, const int32_t *user_ids, const float *itab,
                            int32_t n_items, const float *ubias, const float *ibias, float mean,
                            const int32_t *seen_ptr, const int32_t *seen_idx, int32_t K, int32_t *top_ids,
                            float *top_scores, const int32_t *targets, int32_t n_targets, float *target_score,
                            int32_t *target_rank, const rk_score_plan *plan, float *scratch, void *stream)
{
    if (nb <= 0) return RK_OK;
    if (dim <= 0 || n_items <= 0 || !utab || !itab || !user_ids || !seen_ptr || !seen_idx || !scratch || !top_ids || !top_scores || !plan)
        RK_FAIL(RK_EINVAL, "rk_score_topk: bad arguments");
    if ((ubias == nullptr) != (ibias == nullptr)) RK_FAIL(RK_EINVAL, "rk_score_topk: give both biases or neither");
    if (K <= 0 || K > kMaxK) RK_FAIL(RK_EINVAL, "top-K: K must be in [1,%d]", kMaxK);
    if (n_targets < 0 || n_targets > 256 || (n_targets > 0 && (!targets || !target_score || !target_rank)))
        RK_FAIL(RK_EINVAL, "top-K: bad targets");
    // the plan sized `scratch`: it must be the plan of THIS request (a smaller block of the same plan is fine for the panel form,
    // whose scratch does not depend on nb, and for the matrix, which only shrinks)
    if (plan->n_items != n_items || plan->dim != dim || plan->K != K || plan->n_targets != n_targets || nb > plan->nb)
        RK_FAIL(RK_EINVAL, "rk_score_topk: the plan was made for another request (rk_score_topk_plan)");
    hipStream_t s = (hipStream_t)stream;
    if (plan->path == RK_SCORE_PANEL) {
        if (!pan_supported(n_items, dim, K, n_targets)) RK_FAIL(RK_EINVAL, "rk_score_topk: the panel form does not take this request");
        if (reinterpret_cast<uintptr_t>(scratch) & 15) RK_FAIL(RK_EINVAL, "rk_score_topk: scratch must be 16-byte aligned (panel form)");
        PanArgs a;
        memset(&a, 0, sizeof(a));
        a.nb = nb; a.n_items = n_items; a.d = dim; a.K = K;
        a.utab = utab; a.user_ids = user_ids; a.itab = itab; a.ubias = ubias; a.ibias = ibias; a.mean = mean;
        a.seen_ptr = seen_ptr; a.seen_idx = seen_idx; a.targets = targets; a.n_targets = n_targets;
        a.top_ids = top_ids; a.top_scores = top_scores; a.target_score = target_score; a.target_rank = target_rank;
        RK_HIP(score_panel_launch(a, scratch, s, plan->panel_rows, plan->panel_ntw, plan->panel_safe));
        return RK_OK;
    }
    if (plan->path != RK_SCORE_GEMM) RK_FAIL(RK_EINVAL, "rk_score_topk: plan->path %d", plan->path);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = nb; g.N = n_items; g.K = dim;
    g.A = utab; g.a_rs = dim; g.a_cs = 1; g.a_ridx = user_ids;  // the user rows are gathered by the tile loads
    g.B = itab; g.b_rs = dim; g.b_cs = 1;
    if (plan->ld_scores != (int32_t)score_ld(n_items)) RK_FAIL(RK_EINVAL, "rk_score_topk: plan->ld_scores %d (rk_score_topk_plan sets it)", plan->ld_scores);
    g.C = scratch; g.ldc = plan->ld_scores;
    g.row_bias = ubias; g.col_bias = ibias; g.const_add = mean;
    RK_HIP(gemm_f32_launch(g, s));
    return rk_topk_rows_impl(scratch, plan->ld_scores, nb, n_items, user_ids, seen_ptr, seen_idx, K, top_ids, top_scores, targets, n_targets,
                             target_score, target_rank, s);
}

// ---------------------------------------------------------------- HR@k reduction
// counts[t*nk + q] = #{b : target_rank[b*T + t] < ks[q]}   (hit@k <=> rank < k, normal.py:86-92,150-156)
__global__ __launch_bounds__(1024) void hit_counts_kernel(const int *__restrict__ rank, long long n, int T, const int *__restrict__ ks,
                                                          int nk, int *__restrict__ counts, int use_atomics)
{
    constexpr int KG = 4;  // thresholds counted per pass over the ranks
    __shared__ int wtot[KG][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int t = 0; t < T; ++t)
        for (int q0 = 0; q0 < nk; q0 += KG) {
            int kk[KG], c[KG];
#pragma unroll
            for (int q = 0; q < KG; ++q) { kk[q] = q0 + q < nk ? ks[q0 + q] : INT_MIN; c[q] = 0; }
            for (long long b = (long long)blockIdx.x * 1024 + tid; b < n; b += (long long)gridDim.x * 1024) {
                const int r = rank[b * T + t];
#pragma unroll
                for (int q = 0; q < KG; ++q) c[q] += r < kk[q] ? 1 : 0;
            }
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                const int s = (int)wave_sum((float)c[q]);  // per-wave partial < 2^24: exact in fp32
                if (lane == 0) wtot[q][w] = s;
            }
            __syncthreads();
            if (tid < KG && q0 + tid < nk) {
                int tot = 0;
                for (int i = 0; i < 16; ++i) tot += wtot[tid][i];
                if (use_atomics) atomicAdd(&counts[t * nk + q0 + tid], tot);
                else counts[t * nk + q0 + tid] = tot;
            }
            __syncthreads();
        }
}

RK_EXPORT int rk_hit_counts(const int32_t *target_rank, int64_t n, int32_t n_targets, const int32_t *ks, int32_t nk,
                            int32_t *counts, void *stream)
{
    if (n_targets <= 0 || nk <= 0) return RK_OK;
    if (n < 0 || !target_rank || !ks || !counts) RK_FAIL(RK_EINVAL, "rk_hit_counts: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)std::max<long long>(1, std::min<long long>(64, (n + 16383) / 16384));
    if (grid > 1) RK_HIP(rk_zero_async(counts, sizeof(int32_t) * (size_t)n_targets * nk, s));
    hipLaunchKernelGGL(hit_counts_kernel, dim3(grid), dim3(1024), 0, s, target_rank, (long long)n, n_targets, ks, nk, counts, grid > 1);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// ---------------------------------------------------------------- evaluation plumbing on the device
// Users the reference evaluates (normal.py:133-143): a non-empty train list that holds none of the targets.
__global__ void eligible_flags_kernel(int n_users, const int *__restrict__ ptr, const int *__restrict__ idx,
                                      const int *__restrict__ targets, int n_targets, int *__restrict__ flags)
{
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n_users; u += gridDim.x * blockDim.x) {
        const int b = ptr[u], e = ptr[u + 1];
        int ok = e > b ? 1 : 0;
        for (int t = 0; t < n_targets && ok; ++t) {
            const int x = targets[t];
            int lo = b, hi = e;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (idx[mid] < x) lo = mid + 1; else hi = mid; }
            if (lo < e && idx[lo] == x) ok = 0;
        }
        flags[u] = ok;
    }
}

// ordered stream compaction by ONE workgroup: out[k] = k-th user with flag set (ascending user id), count[0] = total
__global__ __launch_bounds__(1024) void compact_flags_kernel(int n, const int *__restrict__ flags, int *__restrict__ out,
                                                             int *__restrict__ count)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int f = i < n ? flags[i] : 0;
        const unsigned long long m = __ballot(f != 0);
        const int pre = __popcll(m & ((1ULL << lane) - 1ULL));
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = carry, tot = 0;
        for (int k = 0; k < 16; ++k) { if (k < w) off += wsum[k]; tot += wsum[k]; }
        if (f) out[off + pre] = i;
        __syncthreads();
        if (tid == 0) carry += tot;
        __syncthreads();
    }
    if (tid == 0) count[0] = carry;
}

RK_EXPORT int rk_eligible_users(int32_t n_users, const int32_t *seen_ptr, const int32_t *seen_idx, const int32_t *targets,
                                int32_t n_targets, int32_t *flags_scratch, int32_t *user_ids, int32_t *count, void *stream)
{
    if (n_users <= 0) return RK_OK;
    if (!seen_ptr || !seen_idx || n_targets < 0 || (n_targets > 0 && !targets) || !flags_scratch || !user_ids || !count)
        RK_FAIL(RK_EINVAL, "rk_eligible_users: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(eligible_flags_kernel, dim3(std::min(2048, (n_users + 255) / 256)), dim3(256), 0, s, n_users, seen_ptr, seen_idx,
                       targets, n_targets, flags_scratch);
    RK_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, s, n_users, flags_scratch, user_ids, count);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// pred_shift = mean(score_after - score_before) over all (user, target) rows (normal.py:147-149), accumulated in
// double in a fixed order (one workgroup, strided partial sums, tree) => reproducible.
__global__ __launch_bounds__(1024) void pred_shift_kernel(long long n, const float *__restrict__ before, const float *__restrict__ after,
                                                          double *__restrict__ out)
{
    __shared__ double part[1024];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) s += (double)after[i] - (double)before[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = n > 0 ? part[0] / (double)n : 0.0; out[1] = part[0]; }
}

RK_EXPORT int rk_pred_shift(const float *score_before, const float *score_after, int64_t n, double *out, void *stream)
{
    if (n < 0 || !out || (n > 0 && (!score_before || !score_after))) RK_FAIL(RK_EINVAL, "rk_pred_shift: bad arguments");
    hipLaunchKernelGGL(pred_shift_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (long long)n, score_before, score_after, out);
    RK_CHECK_LAUNCH();
    return RK_OK;
}

// Materialised score block out[b, i] = <utab[user_ids[b]], itab[i]> (+ biases + mean), optionally through nn.Dropout on
// the score (MF.forward of a module in training mode with dropout > 0, mf.py:47: the workflows score without .eval()).
RK_EXPORT int rk_score_matrix(int32_t dim, const float *utab, int32_t nb, const int32_t *user_ids, const float *itab,
                              int32_t n_items, const float *ubias, const float *ibias, float mean, float dropout,
                              uint64_t drop_seed, float *out, void *stream)
{
    if (nb <= 0) return RK_OK;
    if (dim <= 0 || n_items <= 0 || !utab || !itab || !user_ids || !out) RK_FAIL(RK_EINVAL, "rk_score_matrix: bad arguments");
    if ((ubias == nullptr) != (ibias == nullptr)) RK_FAIL(RK_EINVAL, "rk_score_matrix: give both biases or neither");
    if (!(dropout >= 0.f) || dropout >= 1.f) RK_FAIL(RK_EINVAL, "rk_score_matrix: dropout must be in [0, 1)");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = nb; g.N = n_items; g.K = dim;
    g.A = utab; g.a_rs = dim; g.a_cs = 1; g.a_ridx = user_ids;
    g.B = itab; g.b_rs = dim; g.b_cs = 1;
    g.C = out; g.ldc = n_items;
    g.row_bias = ubias; g.col_bias = ibias; g.const_add = mean;
    if (dropout > 0.f) {
        g.drop_thresh24 = (unsigned)((1.0 - (double)dropout) * 16777216.0);
        g.drop_scale = 1.0f / (1.0f - dropout);
        g.drop_seed = (unsigned long long)drop_seed;
    }
    RK_HIP(gemm_f32_launch(g, (hipStream_t)stream));
    return RK_OK;
}

// LightGCN.getUsersRating (lightgcn.py:115-120): out[b, i] = sigmoid(<utab[user_ids[b]], itab[i]>) on the fp32-MFMA GEMM
RK_EXPORT int rk_users_rating(int32_t dim, const float *utab, int32_t nb, const int32_t *user_ids, const float *itab,
                              int32_t n_items, float *out, void *stream)
{
    if (nb <= 0) return RK_OK;
    if (dim <= 0 || n_items <= 0 || !utab || !itab || !user_ids || !out) RK_FAIL(RK_EINVAL, "rk_users_rating: bad arguments");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = nb; g.N = n_items; g.K = dim;
    g.A = utab; g.a_rs = dim; g.a_cs = 1; g.a_ridx = user_ids;
    g.B = itab; g.b_rs = dim; g.b_cs = 1;
    g.C = out; g.ldc = n_items;
    g.sigmoid = 1;
    RK_HIP(gemm_f32_launch(g, (hipStream_t)stream));
    return RK_OK;
}
