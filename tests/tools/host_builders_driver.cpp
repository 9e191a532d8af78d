// Sanitizer driver for the HOST-ONLY builders (recad_amd/csrc/host/*.h): reads graphs from a file, runs the LDS plan builder
// (a std::thread pool) and the CSR schedule builder on each -- several graphs concurrently from several caller threads,
// like independent victims on their own streams would -- and writes the words they produce, which the test compares bit for bit
// with the product library's.  Built by `make -C recad_amd/csrc host-asan host-tsan`; never part of librecad_hip.so.
//
// input file : int32 n_graphs, then per graph: int32 U, I, dim, n_cu, class_split_flag, nnz; int32 rowptr[U+I+1]; int32 col[nnz]; float val[nnz]
// output file: per graph: int64 n_plan_words, int32 plan_words[...], int64 n_sched_words, int32 n_blocks, int64 scratch_words, int32 sched_words[...]
#include <stdio.h>

#include <mutex>
#include <string>

#include "../../recad_amd/csrc/host/csr_schedule_host.h"
#include "../../recad_amd/csrc/host/lds_plan_host.h"

thread_local char rk_err_buf[512] = "";

struct Graph {
    int32_t U, I, dim, n_cu, split, nnz;
    std::vector<int32_t> rowptr, col;
    std::vector<float> val;
    std::vector<int32_t> plan, sched;
    int32_t n_blocks = 0;
    int64_t scratch_words = 0;
    int rc = 0;
};

static bool read_all(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

static void run_one(Graph &g)
{
    rk_lds_plan_t plan = nullptr;
    int64_t n_words = 0;
    rk_lds_info info;
    g.rc = lds_plan_build_host_impl(g.U, g.I, g.rowptr.data(), g.col.data(), g.val.data(), g.dim, g.n_cu, &plan, &n_words, &info);
    if (g.rc) return;
    if (plan) {
        g.plan = plan->words;
        if ((int64_t)g.plan.size() != n_words) g.rc = -100;
        delete plan;
    }
    rk_schedule_t sc = nullptr;
    int64_t sw = 0;
    g.rc = csr_schedule_build_host_impl(g.U + g.I, g.rowptr.data(), g.split ? g.U : 0, g.dim, &sc, &g.n_blocks, &sw, &g.scratch_words);
    if (g.rc) return;
    g.sched = sc->desc;
    const int32_t hdr[4] = {sc->n_long, sc->n_slots, sc->dim, (int32_t)(sc->packed.size() / 4)};
    g.sched.insert(g.sched.end(), hdr, hdr + 4);
    g.sched.insert(g.sched.end(), sc->bmeta.begin(), sc->bmeta.end());
    g.sched.insert(g.sched.end(), sc->packed.begin(), sc->packed.end());
    if ((int64_t)g.sched.size() != sw) g.rc = -101;
    delete sc;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s graphs.bin out.bin [caller threads]\n", argv[0]); return 2; }
    const int n_callers = argc > 3 ? atoi(argv[3]) : 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t n = 0;
    if (!read_all(f, &n, 4) || n <= 0 || n > 100000) { fprintf(stderr, "bad header\n"); return 2; }
    std::vector<Graph> gs((size_t)n);
    for (Graph &g : gs) {
        int32_t h[6];
        if (!read_all(f, h, sizeof(h))) { fprintf(stderr, "truncated input\n"); return 2; }
        g.U = h[0]; g.I = h[1]; g.dim = h[2]; g.n_cu = h[3]; g.split = h[4]; g.nnz = h[5];
        g.rowptr.resize((size_t)g.U + g.I + 1); g.col.resize((size_t)g.nnz); g.val.resize((size_t)g.nnz);
        if (!read_all(f, g.rowptr.data(), 4 * g.rowptr.size()) || !read_all(f, g.col.data(), 4 * g.col.size()) || !read_all(f, g.val.data(), 4 * g.val.size())) {
            fprintf(stderr, "truncated input\n");
            return 2;
        }
    }
    fclose(f);
    std::atomic<size_t> next(0);
    auto caller = [&]() { for (size_t i = next++; i < gs.size(); i = next++) run_one(gs[i]); };
    std::vector<std::thread> pool;
    for (int k = 1; k < n_callers; ++k) pool.emplace_back(caller);
    caller();
    for (auto &t : pool) t.join();
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    for (const Graph &g : gs) {
        if (g.rc) { fprintf(stderr, "builder failed: rc %d (%s)\n", g.rc, rk_err_buf); return 3; }
        const int64_t np = (int64_t)g.plan.size(), ns = (int64_t)g.sched.size();
        fwrite(&np, 8, 1, o); fwrite(g.plan.data(), 4, g.plan.size(), o);
        fwrite(&ns, 8, 1, o); fwrite(&g.n_blocks, 4, 1, o); fwrite(&g.scratch_words, 8, 1, o); fwrite(g.sched.data(), 4, g.sched.size(), o);
    }
    fclose(o);
    printf("ok %d graphs\n", n);
    return 0;
}
