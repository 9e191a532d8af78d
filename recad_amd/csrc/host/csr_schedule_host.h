// HOST-ONLY builder of the row-gather SpMM's work schedule (spmm.h): pure C++, also built under the sanitizers
// (make -C recad_amd/csrc host-asan host-tsan).
#pragma once
#include <algorithm>
#include <exception>
#include <memory>
#include <new>
#include <numeric>
#include <vector>

#include "../../../include/recad_hip.h"
#include "layout.h"

// ---- SpMM work schedule (see spmm.h)
struct rk_schedule {
    std::vector<int32_t> desc;   // int4 per wave: {row, e_begin, e_end, n_segments if leader else 0}
    std::vector<int32_t> bmeta;  // int4 per workgroup: {n_pieces, piece index, first slot, counter index}; zeros = whole rows
    std::vector<int32_t> packed; // int4 per packed short row: {row, e_begin, e_end, 0}
    int32_t n_blocks = 0, n_long = 0, n_slots = 0, dim = 0;
    size_t words() const { return desc.size() + 4 + bmeta.size() + packed.size(); }
    // per-stream scratch of the long rows: arrival counters (padded to 16 bytes) + partial-sum slots
    size_t scratch_words() const { return n_long ? (size_t)((n_long + 3) & ~3) + (size_t)n_slots * (size_t)dim : 0; }
};

// rowptr: HOST array of n_rows + 1 entries
inline int csr_schedule_build_host_body(int32_t n_rows, const int32_t *rowptr, int32_t class_split, int32_t dim,
                                        rk_schedule_t *out, int32_t *n_blocks, int64_t *n_words, int64_t *scratch_words);
// what the C-ABI entry points call: no C++ exception unwinds through extern "C" (see host/lds_plan_host.h)
inline int csr_schedule_build_host_impl(int32_t n_rows, const int32_t *rowptr, int32_t class_split, int32_t dim,
                                        rk_schedule_t *out, int32_t *n_blocks, int64_t *n_words, int64_t *scratch_words)
{
    try {
        return csr_schedule_build_host_body(n_rows, rowptr, class_split, dim, out, n_blocks, n_words, scratch_words);
    } catch (const std::bad_alloc &) {
        RK_FAIL(RK_ENOMEM, "rk_csr_schedule_build: out of host memory");
    } catch (const std::exception &e) {
        RK_FAIL(RK_EINVAL, "rk_csr_schedule_build: %s", e.what());
    }
}
inline int csr_schedule_build_host_body(int32_t n_rows, const int32_t *rowptr, int32_t class_split, int32_t dim,
                                        rk_schedule_t *out, int32_t *n_blocks, int64_t *n_words, int64_t *scratch_words)
{
    if (n_rows <= 0 || !rowptr || !out || !n_blocks || !n_words || !scratch_words || class_split < 0 || class_split > n_rows || dim <= 0 || dim > 256)
        RK_FAIL(RK_EINVAL, "rk_csr_schedule_build: bad arguments");
    // short rows are packed one per lane group (dim/4 lanes): only the vector kernels with >= 2 groups do that
    static const int no_pack = RK_TUNE_INT("RK_SPMM_NO_PACK", 0);
    int pack_groups = (!no_pack && (dim == 32 || dim == 64 || dim == 128)) ? 256 / dim : 1;
    int pack_max = pack_groups > 1 ? dim / 4 : -1;  // nonzeros a lane group reads in one chunk
    const int32_t *rp = rowptr;
    // rows by degree, descending, stable in row id (counting sort) => deterministic schedule
    int32_t maxdeg = 0;
    for (int32_t r = 0; r < n_rows; ++r) maxdeg = std::max(maxdeg, rp[r + 1] - rp[r]);
    std::vector<int32_t> cnt((size_t)maxdeg + 2, 0), order((size_t)n_rows);
    for (int32_t r = 0; r < n_rows; ++r) cnt[(size_t)(maxdeg - (rp[r + 1] - rp[r])) + 1]++;
    for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
    for (int32_t r = 0; r < n_rows; ++r) order[(size_t)cnt[(size_t)(maxdeg - (rp[r + 1] - rp[r]))]++] = r;

    if (pack_groups > 1) {
        // packing uses a separate kernel instantiation (the packed path costs the plain one ~2 %): only
        // worth it when a good share of the rows is short (the reference's as-is test-edge graphs)
        int32_t n_short = 0;
        for (int32_t r = 0; r < n_rows; ++r) n_short += (rp[r + 1] - rp[r] <= pack_max) ? 1 : 0;
        if (4LL * n_short < n_rows) { pack_groups = 1; pack_max = -1; }
    }
    const int kSpmmWaves = spmm_waves_for((long long)rp[n_rows]);
    // Nonzeros per segment (= per wave).  A row of 65..128 nonzeros is two waves + an LDS combine at 64 but one
    // wave at 128: when most of the nonzeros sit in such rows (ml1m's train graph: 97 per row on average) 128
    // is worth 7 % of a train step (106.4 -> 99.2 us); on short-row graphs it only lengthens the tail (the
    // reference's as-is graph: 46.7 -> 50.0 us), and above 8 M nonzeros it measured flat.  RK_SEG_NNZ overrides.
    static const int seg_env = RK_TUNE_INT("RK_SEG_NNZ", 0) ? std::max(16, RK_TUNE_INT("RK_SEG_NNZ", 0)) : 0;
    int seg_nnz = seg_env ? seg_env : kSegNnz;
    if (!seg_env && kSpmmWaves == 4) {
        long long over = 0;
        for (int32_t r = 0; r < n_rows; ++r) { const int32_t k = rp[r + 1] - rp[r]; if (k > kSegNnz) over += k; }
        if (2 * over >= (long long)rp[n_rows]) seg_nnz = 2 * kSegNnz;
    }
    std::unique_ptr<rk_schedule> sc_owner(new rk_schedule());   // freed if anything below throws
    rk_schedule *sc = sc_owner.get();
    // Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an L2).  With
    // class_split > 0 the rows < split (users) and >= split (items) are scheduled into separate
    // workgroup lists that are then interleaved 4:4 per group of 8, so an XCD's L2 only ever
    // fetches ONE of the two embedding tables (speed only; any placement is correct).
    std::vector<int32_t> cls[2], clsm[2];
    std::vector<int32_t> &packed = sc->packed;
    int32_t n_long = 0, n_slots = 0;
    for (int pass = 0; pass < 2; ++pass) {
    std::vector<int32_t> &d = cls[pass];
    std::vector<int32_t> &dm = clsm[pass];
    auto new_block = [&]() {
        const size_t base = d.size();
        d.resize(base + (size_t)kSpmmWaves * 4, 0);
        for (int w = 0; w < kSpmmWaves; ++w) d[base + (size_t)w * 4] = -1;
        dm.resize(dm.size() + 4, 0);
        return base;
    };
    // open workgroups by free wave count: free_list[k] = blocks with exactly k free waves
    std::vector<std::vector<size_t>> free_list((size_t)kSpmmWaves + 1);
    auto take_wave = [&](int nseg, int &used) {  // best fit: the open workgroup with the fewest free waves that still fits
        size_t base = (size_t)-1;
        used = 0;
        for (int k = nseg; k <= kSpmmWaves && base == (size_t)-1; ++k)
            if (!free_list[(size_t)k].empty()) {
                base = free_list[(size_t)k].back();
                free_list[(size_t)k].pop_back();
                used = kSpmmWaves - k;
            }
        if (base == (size_t)-1) { base = new_block(); used = 0; }
        const int left = kSpmmWaves - used - nseg;
        if (left > 0) free_list[(size_t)left].push_back(base);
        return base;
    };
    // short rows (<= one lane-group chunk) are packed pack_groups per wave, one row per lane group
    std::vector<int32_t> pending;
    auto flush_packed = [&]() {
        if (pending.empty()) return;
        int used = 0;
        const size_t base = take_wave(1, used);
        const size_t o = base + (size_t)used * 4;
        int32_t maxn = 0;
        d[o + 0] = (int32_t)(packed.size() / 4);
        for (int32_t r : pending) {
            packed.insert(packed.end(), {r, rp[r], rp[r + 1], 0});
            maxn = std::max(maxn, rp[r + 1] - rp[r]);
        }
        d[o + 1] = (int32_t)pending.size();
        d[o + 2] = maxn;
        d[o + 3] = -1;  // packed wave
        pending.clear();
    };
    for (int32_t oi = 0; oi < n_rows; ++oi) {
        const int32_t r = order[(size_t)oi];
        if ((class_split > 0 && r >= class_split) != (pass == 1)) continue;
        const int32_t b = rp[r], e = rp[r + 1], nnz = e - b;
        if (nnz <= pack_max) {
            pending.push_back(r);
            if ((int)pending.size() == pack_groups) flush_packed();
            continue;
        }
        int32_t nseg = std::max(1, (nnz + seg_nnz - 1) / seg_nnz);
        if (nseg > kSpmmWaves) {
            // long row: ceil(nseg / W) workgroups, each a "piece" of W segments; the pieces' partial sums
            // meet in scratch slots and the last workgroup to arrive adds them in piece order
            const int32_t np = (nseg + kSpmmWaves - 1) / kSpmmWaves;
            for (int32_t p = 0; p < np; ++p) {
                const size_t base = new_block();
                const int32_t s0 = p * kSpmmWaves, s1 = std::min(nseg, s0 + kSpmmWaves);
                for (int32_t sgi = s0; sgi < s1; ++sgi) {
                    const size_t o = base + (size_t)(sgi - s0) * 4;
                    d[o + 0] = r;
                    d[o + 1] = std::min(e, b + sgi * seg_nnz);
                    d[o + 2] = std::min(e, b + (sgi + 1) * seg_nnz);
                    d[o + 3] = (sgi == s0) ? (s1 - s0) : 0;
                }
                int32_t *m = &dm[dm.size() - 4];
                m[0] = np; m[1] = p; m[2] = n_slots; m[3] = n_long;
            }
            n_slots += np;
            ++n_long;
            continue;
        }
        int used = 0;
        const size_t base = take_wave(nseg, used);
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const size_t o = base + (size_t)(used + sgi) * 4;
            d[o + 0] = r;
            d[o + 1] = std::min(e, b + sgi * seg_nnz);
            d[o + 2] = std::min(e, b + (sgi + 1) * seg_nnz);
            d[o + 3] = (sgi == 0) ? nseg : 0;
        }
    }
    flush_packed();
    }
    const size_t bw = (size_t)kSpmmWaves * 4;
    const size_t nb0 = cls[0].size() / bw, nb1 = cls[1].size() / bw;
    std::vector<int32_t> &d = sc->desc;
    d.reserve(cls[0].size() + cls[1].size());
    size_t i0 = 0, i1 = 0;
    for (size_t b = 0; i0 < nb0 || i1 < nb1; ++b) {
        bool want1 = (b % 8) >= 4;
        if (want1 && i1 >= nb1) want1 = false;
        if (!want1 && i0 >= nb0) want1 = true;
        const std::vector<int32_t> &src = want1 ? cls[1] : cls[0];
        const std::vector<int32_t> &srcm = want1 ? clsm[1] : clsm[0];
        size_t &idx = want1 ? i1 : i0;
        d.insert(d.end(), src.begin() + (long)(idx * bw), src.begin() + (long)((idx + 1) * bw));
        sc->bmeta.insert(sc->bmeta.end(), srcm.begin() + (long)(idx * 4), srcm.begin() + (long)((idx + 1) * 4));
        ++idx;
    }
    sc->n_blocks = (int32_t)(d.size() / bw);
    sc->n_long = n_long; sc->n_slots = n_slots; sc->dim = dim;
    *out = sc_owner.release();
    *n_blocks = sc->n_blocks | (sc->packed.empty() ? 0 : kSchedPackedFlag) | sched_waves_code(kSpmmWaves) |
                (n_long ? kSchedLongFlag : 0);  // opaque launch parameter
    *n_words = (int64_t)sc->words();
    *scratch_words = (int64_t)sc->scratch_words();
    return RK_OK;
}

