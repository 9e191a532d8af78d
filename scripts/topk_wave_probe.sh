#!/bin/bash
# Per-phase cycle stamps of topk_wave_kernel (csrc/score_topk.hip), built ALONE from the product source with TOPK_STAMP defined:
#   scripts/topk_wave_probe.sh [nb n_items NQ]      (default 5893 3702 58: the headline evaluation)
# phases: 0 start | 1 loads issued | 2 seen chain + bitmap read | 3 keys converted (row landed) | 4 target ranks | 5 bound |
#         6 compaction | 7 rank sort | 8 outputs written
nb=${1:-5893}; ni=${2:-3702}; nq=${3:-58}
root=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
w=$(mktemp -d /tmp/tkprobe.XXXX)
python3 - "$root" "$w" "$nq" <<'PY'
import sys
root, w, nq = sys.argv[1:4]
s = open(root + "/recad_amd/csrc/score_topk.hip").read()
a = s.index("// a copy of a per-lane value the compiler cannot hoist")
b = s.index('extern "C" int rk_topk_rows_impl(float *scores')
hdr = r'''#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <type_traits>
#include <vector>
#ifndef PROBE_MAXK
#define PROBE_MAXK 256
#endif
static constexpr int kMaxK = PROBE_MAXK;
#ifndef TOPK_WAVE_WG
#define TOPK_WAVE_WG 4
#endif
__device__ unsigned long long *g_stamps;
#define TOPK_STAMP(i) do { if (g_stamps && (threadIdx.x & 63) == 0) { unsigned long long *p_ = g_stamps + ((size_t)blockIdx.x * TOPK_WAVE_WG + (threadIdx.x >> 6)) * 12; p_[i] = clock64(); if ((i) == 0) p_[9] = wall_clock64(); if ((i) == 8) p_[10] = wall_clock64(); } } while (0)
__device__ __forceinline__ unsigned score_key(float s)
{
    if (s == -INFINITY) return 0u;
    unsigned u = __float_as_uint(s);
    if (u == 0x80000000u) u = 0u;
    const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return k == 0u ? 1u : k;
}
__device__ __forceinline__ float key_score(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
'''
tail = r'''
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const int nb = atoi(argv[1]), ni = atoi(argv[2]), K = 100, deg = 80;
    std::vector<float> h((size_t)nb * ni);
    unsigned long long z = 12345;
    auto rnd = [&]() { z = z * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((z >> 40) & 0xffffff) / 16777216.0f; };
    for (auto &x : h) { float s = 0; for (int i = 0; i < 6; ++i) s += rnd(); x = (s - 3.0f) * 0.8f; }
    std::vector<int> uid(nb), sp(nb + 1), si((size_t)nb * deg), tg(1, 0);
    for (int b = 0; b < nb; ++b) { uid[b] = b; sp[b] = b * deg; for (int k = 0; k < deg; ++k) si[(size_t)b * deg + k] = (int)(rnd() * ni) % ni; }
    sp[nb] = nb * deg;
    float *ds, *dts, *dtop; int *duid, *dsp, *dsi, *dtg, *dids, *dtr; unsigned long long *dst;
    CK(hipMalloc(&ds, h.size() * 4)); CK(hipMemcpy(ds, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&duid, nb * 4)); CK(hipMemcpy(duid, uid.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dsp, (nb + 1) * 4)); CK(hipMemcpy(dsp, sp.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dsi, si.size() * 4)); CK(hipMemcpy(dsi, si.data(), si.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dtg, 4)); CK(hipMemcpy(dtg, tg.data(), 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dids, (size_t)nb * K * 4)); CK(hipMalloc(&dtop, (size_t)nb * K * 4)); CK(hipMalloc(&dts, nb * 4)); CK(hipMalloc(&dtr, nb * 4));
    const int nwave = nb;
    CK(hipMalloc(&dst, (size_t)nwave * 12 * 8)); CK(hipMemset(dst, 0, (size_t)nwave * 12 * 8));
    unsigned long long *null_p = nullptr;
    auto launch = [&]() { hipLaunchKernelGGL((topk_wave_kernel<NQV>), dim3((nb + TOPK_WAVE_WG - 1) / TOPK_WAVE_WG), dim3(64 * TOPK_WAVE_WG), 0, 0, ds, (long long)ni, nb, ni, duid, dsp, dsi, K, dids, dtop, dtg, 1, dts, dtr); };
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &null_p, sizeof(null_p)));
    for (int i = 0; i < 5; ++i) launch();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("topk_wave_kernel<%d> on %d x %d: %.2f us per launch (no stamps)\n", NQV, nb, ni, ms * 1e3 / 20);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
    launch(); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nwave * 12);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    double sum[9] = {0}; unsigned long long t_min = ~0ULL, t_max = 0; int n = 0;
    unsigned long long s_min = ~0ULL, s_max = 0, e_min = ~0ULL, e_max = 0; double life = 0;
    for (int w = 0; w < nb; ++w) { const unsigned long long *p = &st[(size_t)w * 12]; if (!p[8]) continue; ++n; for (int i = 1; i < 9; ++i) sum[i] += (double)(p[i] - p[i - 1]); if (p[0] < t_min) t_min = p[0]; if (p[8] > t_max) t_max = p[8];
        if (p[9] < s_min) s_min = p[9]; if (p[9] > s_max) s_max = p[9]; if (p[10] < e_min) e_min = p[10]; if (p[10] > e_max) e_max = p[10]; life += (double)(p[10] - p[9]); }
    const char *names[9] = {"", "loads issued", "seen chain + bitmap", "keys converted", "target ranks", "bound", "compaction", "rank sort", "outputs"};
    double tot = 0; for (int i = 1; i < 9; ++i) tot += sum[i] / n;
    printf("per-wave cycles (clock64, mean over %d waves): total %.0f\n", n, tot);
    for (int i = 1; i < 9; ++i) printf("  %-22s %8.0f  (%4.1f %%)\n", names[i], sum[i] / n, 100.0 * sum[i] / n / tot);
    printf("wall clock (100 MHz): first start -> last start %.2f us, first end %.2f us, last end %.2f us; mean wave lifetime %.2f us => clock64 at %.0f MHz\n",
           (s_max - s_min) / 100.0, (e_min - s_min) / 100.0, (e_max - s_min) / 100.0, life / n / 100.0, tot / (life / n / 100.0));
    { std::vector<int> hist(64, 0); for (int w = 0; w < nb; ++w) { const unsigned long long *p = &st[(size_t)w * 12]; if (p[8]) { int b_ = (int)((p[9] - s_min) / 100); if (b_ < 64) hist[b_]++; } }
      printf("wave starts per microsecond:"); for (int i = 0; i < 48; ++i) printf(" %d", hist[i]); printf("\n"); }
    return 0;
}
'''
open(w + "/probe.hip", "w").write(hdr + s[a:b].replace("#ifndef TOPK_STAMP", "#if 0").replace("#define TOPK_STAMP(i) do { } while (0)\n#endif", "#endif") + tail.replace("NQV", nq))
PY
# PROBE_DEFS: extra -D flags, e.g. PROBE_DEFS='-DTOPK_WAVES_EU(NQ)=3,3' (occupancy A/B)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $PROBE_DEFS $w/probe.hip -o $w/probe 2>&1 | grep -E "error|VGPRs:|Occupancy" ; if [ -n "$PROBE_PMC" ]; then
  # PROBE_PMC="SQ_INSTS_VALU SQ_INSTS_SALU ...": per-dispatch counters of the same binary (rocprofv3 --pmc, its own pass)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc $PROBE_PMC --output-format csv -d $w/pmc -- $w/probe $nb $ni > /dev/null 2>&1
  python3 - $w/pmc <<'PY'
import csv, glob, sys, collections
v = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "topk_wave" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in sorted(v.items()):
    print("PMC %-28s mean %14.1f  (%d dispatches)" % (k, sum(x) / len(x), len(x)))
PY
else
  $w/probe $nb $ni
fi
rm -rf $w
