// Fused full-catalog scoring + selection for gfx950: the [users, items] score matrix of
// Normal.user_item_model_generate (recad/workflow/normal.py:57-93) is never materialised.
//
// One workgroup owns RB user rows (their embedding rows stay in LDS for the whole kernel) and sweeps the
// catalogue in tiles of 128 items:
//   * scores of the RB x 128 tile on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, k-ordered fmaf chain =>
//     bit-identical to the oracle's scalar loop), item rows streamed through a double-buffered LDS tile in
//     k-chunks of 32 with register prefetch of the next chunk;
//   * epilogue straight from the accumulators: bias terms, seen-item mask (a 128-bit LDS bitmap per row and
//     tile, filled from the user's sorted train list by a running cursor whose loads fly under the MFMAs),
//     rank of up to 4 targets (#(s > s_t) + #(s == s_t, id < t) in per-lane registers), and a THRESHOLDED
//     append of (key, ~id) composites to the row's candidate list in global memory (512 slots per row):
//     a score enters only if its key is above tau_row, the K-th best key seen so far.  Items arrive in
//     increasing id order, so once K candidates with key >= tau are held a later item with key == tau can
//     never enter the top K (it loses the id tie-break): "key > tau" is exact, ties included;
//   * when a row's list could overflow with the next tile, one wave compacts it to its exact top K
//     (bit-wise binary search for the K-th largest 64-bit composite: 8 entries per lane in registers,
//     v_cmp_ge_u64 + s_bcnt1 per entry and bit) and raises tau_row.  Expected appends after the first
//     compaction are ~K ln(tiles/3): one or two compactions per row for catalogues up to millions of items;
//   * after the last tile every row is compacted once more, rank-sorted through LDS by (score desc, id asc)
//     and written out with the target ranks.
// Nothing but the K results per row, the target scores / ranks and the (L2-resident) candidate lists
// touches memory: 22 M scores of the ml1m evaluation were 88 MB written and re-read before.
#pragma once
#include "common.h"

typedef float sel_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned score_key(float s)
{
    // monotone float -> uint; 0 is reserved for "excluded" (seen item, encoded as -inf).  -0.0 is
    // folded onto +0.0 so that key equality is float equality (the oracle compares floats).
    if (s == -INFINITY) return 0u;
    unsigned u = __float_as_uint(s);
    if (u == 0x80000000u) u = 0u;
    const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return k == 0u ? 1u : k;
}

__device__ __forceinline__ float key_score(unsigned k)
{
    // inverse of score_key for every finite score and +inf (-0.0 comes back as +0.0)
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

static constexpr int kSelC = 512;     // candidate slots per user row
static constexpr int kSelTN = 128;    // items per tile
static constexpr int kSelKC = 32;     // k per LDS chunk
static constexpr int kSelLdB = 36;    // padded chunk row: operand reads hit every bank exactly twice, rows 16-byte aligned
static constexpr int kSelMaxT = 4;    // targets ranked in the sweep
static constexpr int kSelMaxK = 256;

struct SelArgs {
    int nb, n_items, d, K;
    const float *utab;
    const int *user_ids;
    const float *itab;
    const float *ubias, *ibias;   // both or neither: s = ((dot + ubias[u]) + ibias[i]) + mean
    float mean;
    const int *seen_ptr, *seen_idx;
    const int *targets;
    int n_targets;
    int *top_ids;
    float *top_scores;
    float *target_score;
    int *target_rank;
    unsigned long long *cand;     // [nb][kSelC] scratch
    int id_bits;                  // item ids < 2^id_bits
};

__host__ __device__ inline int sel_row_stride(int d) { return ((d + kSelKC - 1) / kSelKC) * kSelKC + 4; }

// K-th largest of the (distinct) composites held 8 per lane (unused slots 0), n_valid >= K: returns T with
// exactly K composites >= T.  The 32 key bits are always resolved (T >> 32 is the exact K-th key); the id
// bits above id_bits are ones in every entry.
__device__ __forceinline__ unsigned long long wave_kth_composite(const unsigned long long (&c)[8], int K, int id_bits)
{
    unsigned long long T = 0ULL;
    for (int bit = 63; bit >= 32; --bit) {
        const unsigned long long trial = T | (1ULL << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += __popcll(__ballot(c[j] >= trial));
        if (cnt >= K) T = trial;
    }
    if (id_bits < 32) T |= (0xffffffffULL >> id_bits) << id_bits;   // ~id has these bits set in every entry
    for (int bit = (id_bits < 32 ? id_bits : 32) - 1; bit >= 0; --bit) {
        const unsigned long long trial = T | (1ULL << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cnt += __popcll(__ballot(c[j] >= trial));
        if (cnt >= K) T = trial;
        if (cnt == K) break;
    }
    return T;
}

// One wave: compact the candidate list of a row (n entries in global memory, n <= kSelC) to its top K
// composites, in place; returns the new count (min(n, K)) and the new threshold key through *tau_out.
__device__ __forceinline__ int wave_compact_row(unsigned long long *__restrict__ row, int n, int K, int id_bits, int lane,
                                                unsigned *tau_out, unsigned long long (&c)[8])
{
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = (j * 64 + lane < n) ? row[j * 64 + lane] : 0ULL;
    if (n <= K) { *tau_out = 0u; return n; }
    const unsigned long long T = wave_kth_composite(c, K, id_bits);
    int base = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool keep = c[j] >= T;
        const unsigned long long m = __ballot(keep);
        if (keep) row[base + __popcll(m & ((1ULL << lane) - 1ULL))] = c[j];
        base += __popcll(m);
    }
    *tau_out = (unsigned)(T >> 32);
    return K;
}

// WM rows per wave (16 or 32), WAVES_M waves along the rows; 4 waves along the 128 items of a tile.
// NTG: targets ranked inside the sweep (1 or kSelMaxT: a separate instantiation keeps the common one-target
// evaluation free of three dead compare chains per score).
template <int WM, int WAVES_M, int NTG>
__global__ __launch_bounds__(WAVES_M * 256) void score_select_kernel(const SelArgs a)
{
    constexpr int RB = WM * WAVES_M, NW = WAVES_M * 4, NT = NW * 64, BM = WM / 16, RPL = BM * 4;  // rows per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int sCnt[RB], sCur[RB], sEnd[RB], sUid[RB], sRank[RB][kSelMaxT], sNeed;
    __shared__ unsigned sTau[RB], sTkey[RB][kSelMaxT], sMask[2][RB][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int l16 = lane & 15, lq = lane >> 4;
    const int SA = sel_row_stride(a.d);
    float *sA = smem;                                   // [RB][SA]
    float *sB = smem + (size_t)RB * SA;                 // [2][128][kSelLdB]
    const int row0 = blockIdx.x * RB;
    const int n_in = a.n_targets;                       // <= kSelMaxT (host-checked)
    const int n_chunks = (a.d + kSelKC - 1) / kSelKC;
    const int n_tiles = (a.n_items + kSelTN - 1) / kSelTN;

    // ---- prologue: row metadata, user rows -> LDS (zero-padded), target scores
    for (int r = tid; r < RB; r += NT) {
        const int g = row0 + r;
        const int u = g < a.nb ? a.user_ids[g] : -1;
        sUid[r] = u;
        sCur[r] = u >= 0 ? a.seen_ptr[u] : 0;
        sEnd[r] = u >= 0 ? a.seen_ptr[u + 1] : 0;
        sCnt[r] = 0;
        sTau[r] = 0u;
#pragma unroll
        for (int t = 0; t < kSelMaxT; ++t) sRank[r][t] = 0;
    }
    for (int i = tid; i < 2 * RB * 4; i += NT) (&sMask[0][0][0])[i] = 0u;
    if (tid == 0) sNeed = 0;
    __syncthreads();
    for (int i = tid; i < RB * SA; i += NT) {
        const int r = i / SA, k = i % SA;
        const int u = sUid[r];
        sA[i] = (u >= 0 && k < a.d) ? a.utab[(size_t)u * a.d + k] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < RB * kSelMaxT; i += NT) {
        const int r = i / kSelMaxT, t = i % kSelMaxT;
        unsigned key = 0xffffffffu;
        if (t < n_in && sUid[r] >= 0) {
            const int tg = a.targets[t];
            const float *iv = a.itab + (size_t)tg * a.d;
            float s = 0.f;
            for (int k = 0; k < a.d; ++k) s = fmaf(sA[r * SA + k], iv[k], s);   // the MFMA's k-ordered chain
            if (a.ubias) s = ((s + a.ubias[sUid[r]]) + a.ibias[tg]) + a.mean;
            a.target_score[(size_t)(row0 + r) * a.n_targets + t] = s;            // before masking (normal.py:83-85)
            key = score_key(s);
        }
        sTkey[r][t] = key;
    }
    __syncthreads();

    // per-lane constants: the rows this lane's accumulator registers belong to
    unsigned tkey[RPL][NTG];
    int cntr[RPL][NTG];
    int tgt[NTG];
    float ubr[RPL];
#pragma unroll
    for (int t = 0; t < NTG; ++t) tgt[t] = t < n_in ? a.targets[t] : -1;
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int r = wm * WM + (q >> 2) * 16 + 4 * lq + (q & 3);
        ubr[q] = (a.ubias && sUid[r] >= 0) ? a.ubias[sUid[r]] : 0.f;
#pragma unroll
        for (int t = 0; t < NTG; ++t) { tkey[q][t] = sTkey[r][t]; cntr[q][t] = 0; }
    }

    // B tile loader: 128 rows x 32 k = 1024 float4, NT threads
    constexpr int NLD = 1024 / NT;   // float4 per thread (4 or 2)
    float4 breg[NLD];
    const bool vec_ok = (a.d % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.itab) & 15) == 0);
    auto load_b = [&](int tile, int chunk) {
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int idx = p * NT + tid, r = idx >> 3, k4 = (idx & 7) * 4;
            const int item = tile * kSelTN + r, k = chunk * kSelKC + k4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < a.n_items) {
                const float *src = a.itab + (size_t)item * a.d + k;
                if (vec_ok && k + 3 < a.d) v = *reinterpret_cast<const float4 *>(src);
                else {
                    if (k < a.d) v.x = src[0];
                    if (k + 1 < a.d) v.y = src[1];
                    if (k + 2 < a.d) v.z = src[2];
                    if (k + 3 < a.d) v.w = src[3];
                }
            }
            breg[p] = v;
        }
    };
    auto store_b = [&](int buf) {
        float *dst = sB + (size_t)buf * kSelTN * kSelLdB;
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
            const int idx = p * NT + tid, r = idx >> 3, k4 = (idx & 7) * 4;
            *reinterpret_cast<float4 *>(dst + r * kSelLdB + k4) = breg[p];
        }
    };

    sel_f32x4 acc[BM][2];
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = sel_f32x4{0.f, 0.f, 0.f, 0.f};

    load_b(0, 0);
    store_b(0);
    __syncthreads();
    const int total = n_tiles * n_chunks;
    for (int it = 0; it < total; ++it) {
        const int tile = it / n_chunks, chunk = it % n_chunks, cur = it & 1;
        const bool more = it + 1 < total, last_chunk = chunk == n_chunks - 1;
        const int n0 = tile * kSelTN;
        // seen ids of this tile: loads issued before the MFMAs, consumed after them
        int seen_id = 0x7fffffff;
        const bool marker = chunk == 0 && tid < RB * 8;
        if (marker) {
            const int r = tid >> 3, pos = sCur[r] + (tid & 7);
            if (pos < sEnd[r]) seen_id = a.seen_idx[pos];
        }
        if (more) load_b((it + 1) / n_chunks, (it + 1) % n_chunks);
        const float *Bc = sB + (size_t)cur * kSelTN * kSelLdB;
        const float *Ak = sA + chunk * kSelKC;
#pragma unroll
        for (int kk = 0; kk < kSelKC; kk += 4) {
            float av[BM], bv[2];
#pragma unroll
            for (int i = 0; i < BM; ++i) av[i] = Ak[(wm * WM + i * 16 + l16) * SA + kk + lq];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bc[(wn * 32 + j * 16 + l16) * kSelLdB + kk + lq];
#pragma unroll
            for (int i = 0; i < BM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (marker) {
            // mark this tile's seen items; the ids are sorted, so the consumed lanes of a row form a prefix
            const int r = tid >> 3, j8 = tid & 7;
            int curp = sCur[r];
            const int endp = sEnd[r];
            const int sub = (lane >> 3) * 8;
            for (;;) {
                const bool in = seen_id < n0 + kSelTN;
                if (in) atomicOr(&sMask[tile & 1][r][(seen_id - n0) >> 5], 1u << ((seen_id - n0) & 31));
                const unsigned long long m = __ballot(in);
                const int c8 = __popc((unsigned)((m >> sub) & 0xffULL));
                curp += c8;
                if (c8 < 8) break;
                seen_id = (curp + j8 < endp) ? a.seen_idx[curp + j8] : 0x7fffffff;
            }
            if (j8 == 0) sCur[r] = curp;
        }
        if (more) store_b(cur ^ 1);
        __syncthreads();
        if (!last_chunk) continue;

        // ---- epilogue of the tile, straight from the accumulators
        // accumulator register r of block (i, j) = row 4*(lane>>4) + r, column lane&15 of the 16x16 block
#pragma unroll
        for (int i = 0; i < BM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = i * 4 + r;
                const int rl = wm * WM + i * 16 + 4 * lq + r;
                const int uid = sUid[rl];
                const unsigned mask = sMask[tile & 1][rl][wn];
                const unsigned tau = sTau[rl];
                const float ub = ubr[q];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + wn * 32 + j * 16 + l16;
                    float s = acc[i][j][r];
                    acc[i][j][r] = 0.f;
                    if (a.ubias) s = ((s + ub) + (col < a.n_items ? a.ibias[col] : 0.f)) + a.mean;
                    const bool valid = uid >= 0 && col < a.n_items && !((mask >> (j * 16 + l16)) & 1u);
                    const unsigned key = valid ? score_key(s) : 0u;
#pragma unroll
                    for (int t = 0; t < NTG; ++t)
                        cntr[q][t] += (key != 0u && col != tgt[t] && (key > tkey[q][t] || (key == tkey[q][t] && col < tgt[t]))) ? 1 : 0;
                    if (key > tau) {
                        const int p = atomicAdd(&sCnt[rl], 1);
                        a.cand[(size_t)(row0 + rl) * kSelC + p] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)col);
                    }
                }
            }
        }
        __syncthreads();
        // ---- rows that could overflow with the next tile are compacted to their exact top K
        for (int i = tid; i < RB * 4; i += NT) sMask[tile & 1][i >> 2][i & 3] = 0u;
        bool need = false;
        for (int r = tid; r < RB; r += NT) need |= sCnt[r] > kSelC - kSelTN;
        if (need) sNeed = 1;
        __syncthreads();
        if (sNeed) {
            for (int r = w; r < RB; r += NW) {
                const int n = sCnt[r];
                if (n > kSelC - kSelTN) {   // wave-uniform
                    unsigned long long c[8];
                    unsigned tau;
                    const int m = wave_compact_row(a.cand + (size_t)(row0 + r) * kSelC, n, a.K, a.id_bits, lane, &tau, c);
                    if (lane == 0) { sCnt[r] = m; sTau[r] = tau; }
                }
            }
            __syncthreads();
            if (tid == 0) sNeed = 0;
            __syncthreads();
        }
    }

    // ---- target ranks: per-lane counters -> 16-lane groups -> LDS (the 4 waves along the items add up)
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        const int rl = wm * WM + (q >> 2) * 16 + 4 * lq + (q & 3);
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            int v = cntr[q][t];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (l16 == 0 && t < n_in) atomicAdd(&sRank[rl][t], v);
        }
    }
    __syncthreads();
    for (int i = tid; i < RB * n_in; i += NT) {
        const int r = i / n_in, t = i % n_in;
        if (sUid[r] >= 0) a.target_rank[(size_t)(row0 + r) * a.n_targets + t] = sRank[r][t];
    }

    // ---- final selection per row: exact top K, rank-sorted by (score desc, id asc) through LDS (sB is free now)
    unsigned long long *sSort = reinterpret_cast<unsigned long long *>(sB) + (size_t)w * kSelMaxK;
    for (int r = w; r < RB; r += NW) {
        if (sUid[r] < 0) continue;   // wave-uniform
        const unsigned long long *rowp = a.cand + (size_t)(row0 + r) * kSelC;
        const int n = sCnt[r];
        unsigned long long c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = (j * 64 + lane < n) ? rowp[j * 64 + lane] : 0ULL;
        unsigned long long T = 1ULL;   // every real composite is >= 2^32
        if (n > a.K) T = wave_kth_composite(c, a.K, a.id_bits);
        int m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool keep = c[j] >= T;
            const unsigned long long mk = __ballot(keep);
            if (keep) sSort[m + __popcll(mk & ((1ULL << lane) - 1ULL))] = c[j];
            m += __popcll(mk);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        int *out_ids = a.top_ids + (size_t)(row0 + r) * a.K;
        float *out_sc = a.top_scores + (size_t)(row0 + r) * a.K;
        for (int e = lane; e < m; e += 64) {
            const unsigned long long mine = sSort[e];
            int rank = 0;
            for (int j = 0; j < m; ++j) rank += sSort[j] > mine ? 1 : 0;
            out_ids[rank] = (int)(~(unsigned)(mine & 0xffffffffULL));
            out_sc[rank] = key_score((unsigned)(mine >> 32));
        }
        for (int e = m + lane; e < a.K; e += 64) { out_ids[e] = -1; out_sc[e] = -INFINITY; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

inline size_t sel_lds_bytes(int rb, int d) { return sizeof(float) * ((size_t)rb * sel_row_stride(d) + 2 * (size_t)kSelTN * kSelLdB); }

// whether the fused path applies to a request
inline bool sel_supported(int n_items, int d, int K, int n_targets)
{
    return K >= 1 && K <= kSelMaxK && n_targets <= kSelMaxT && d >= 1 && d <= 256 && n_items >= 1;
}

inline hipError_t score_select_launch(SelArgs a, hipStream_t s)
{
    int bits = 1;
    while (bits < 32 && (1LL << bits) < (long long)a.n_items) ++bits;
    a.id_bits = bits;
    const char *fe = getenv("RK_SEL_CONFIG");   // tuning / tests: 1 = 16-row, 2 = 64-row workgroups (read per call)
    const int force = fe ? atoi(fe) : 0;
    // small user blocks: 16 rows per workgroup so that the chip is filled; large ones: 64 rows (4x less item traffic per flop)
    const bool small = force ? force == 1 : ((long long)(a.nb + 63) / 64 < 512);
    static bool attr_set = false;
    if (!attr_set) {
        const void *fns[4] = {reinterpret_cast<const void *>(score_select_kernel<16, 1, 1>), reinterpret_cast<const void *>(score_select_kernel<16, 1, kSelMaxT>),
                              reinterpret_cast<const void *>(score_select_kernel<32, 2, 1>), reinterpret_cast<const void *>(score_select_kernel<32, 2, kSelMaxT>)};
        for (const void *f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    const bool one = a.n_targets <= 1;
    if (small) {
        const dim3 grid((a.nb + 15) / 16), block(256);
        if (one) hipLaunchKernelGGL((score_select_kernel<16, 1, 1>), grid, block, sel_lds_bytes(16, a.d), s, a);
        else hipLaunchKernelGGL((score_select_kernel<16, 1, kSelMaxT>), grid, block, sel_lds_bytes(16, a.d), s, a);
    } else {
        const dim3 grid((a.nb + 63) / 64), block(512);
        if (one) hipLaunchKernelGGL((score_select_kernel<32, 2, 1>), grid, block, sel_lds_bytes(64, a.d), s, a);
        else hipLaunchKernelGGL((score_select_kernel<32, 2, kSelMaxT>), grid, block, sel_lds_bytes(64, a.d), s, a);
    }
    return hipGetLastError();
}
