// Review item 8, measured (probe, not product): the LDS SpMM "one size up" -- the source class streams through LDS in PANELS
// (<= 8 800 rows of a 4-float slice = 138 KB) and a workgroup's output rows are REGISTER accumulators: a lane owns J rows
// (slots), walks slot j of every panel in lock step with the other 63 lanes of its wave (pair-steps: two 16-bit panel-local
// columns per 32-bit stream word, padding = the panel's zero row), and writes its J rows once at the end.  No Y read-modify-
// write, no partial sums in LDS.  One workgroup = (row block, slice); both classes: half 0 = user rows <- item panels, half 1 =
// item rows <- user panels.  Plan built by scripts/spmm_panel_probe2.py (numpy).  Tables slice-major [d/4][N][4].
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -munsafe-fp-atomics scripts/spmm_panel_probe2.hip -o recad_amd/lib/libspmm_panel_probe2.so
#include <hip/hip_runtime.h>

struct Unit {           // one (half, row block): 16 ints
    int src0, n_src;    // first node row / rows of the source class
    int PR, P;          // panel rows, panels
    int J;              // slots in use (<= JMAX)
    int hdr_ofs;        // ints: [P][16 waves][1 + JMAX] = {first stream word / 64, pair-steps of slot 0..}
    int row_ofs;        // ints: [JMAX][1024] output node row (-1 none; bit 30: atomic add, the row is split)
    int pad[9];
};

struct PanelArgs {
    const Unit *units;
    const int *tables;            // hdr / row tables
    const unsigned *stream;       // [pair-step][64 lanes]
    const float *dinv;            // [N]
    const float4 *x;              // [n_slices][N]
    float4 *y;
    int N, n_units, n_slices;
    unsigned long long *stamps;   // nullable: [grid][4] start / staged(last panel) / gathered / done
};

template <int JMAX>
__global__ __launch_bounds__(1024) void spmm_panel_kernel(const PanelArgs a)
{
    extern __shared__ float4 tab[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ui = blockIdx.x % a.n_units, slice = blockIdx.x / a.n_units;
    const Unit u = a.units[ui];
    const float4 *xs = a.x + (size_t)slice * a.N + u.src0;
    const float *dsrc = a.dinv + u.src0;
    if (a.stamps && tid == 0) a.stamps[blockIdx.x * 4 + 0] = wall_clock64();
    float4 acc[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < u.P; ++p) {
        const int r0 = p * u.PR, rows = min(u.PR, u.n_src - r0);
        __syncthreads();
        constexpr int UN = 8;
        for (int i0 = tid; i0 < rows; i0 += 1024 * UN) {
            float4 v[UN];
            float s[UN];
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int i = i0 + k * 1024;
                if (i < rows) { v[k] = xs[r0 + i]; s[k] = dsrc[r0 + i]; }
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int i = i0 + k * 1024;
                if (i < rows) tab[i] = make_float4(v[k].x * s[k], v[k].y * s[k], v[k].z * s[k], v[k].w * s[k]);
            }
        }
        if (tid == 0) tab[u.PR] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (a.stamps && tid == 0 && p == u.P - 1) a.stamps[blockIdx.x * 4 + 1] = wall_clock64();
        const int *hd = a.tables + u.hdr_ofs + (p * 16 + w) * (1 + JMAX);
        // the (panel, wave) stream: pair-steps of slot 0, slot 1, ... back to back, padded to whole chunks of 4 pair-steps at its
        // END only; a lane reads a chunk as one 16-byte word (8 entries), the next chunk is requested before this one is walked
        const uint4 *st = reinterpret_cast<const uint4 *>(a.stream) + (size_t)hd[0] * 16 + lane;
        int total = 0;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) total += hd[1 + j];
        if (total == 0) continue;
        const int n_chunks = (total + 3) >> 2;
        uint4 cur = st[0], n1 = cur, n2 = cur;
        if (n_chunks > 1) n1 = st[64];
        if (n_chunks > 2) n2 = st[128];
        int pos = 0;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int n = hd[1 + j];
            float4 s = acc[j];
            for (int t = 0; t < n; ++t) {
                const int k = pos & 3;   // (wave-uniform)
                const unsigned wd = k == 0 ? cur.x : k == 1 ? cur.y : k == 2 ? cur.z : cur.w;
                const float4 x0 = tab[wd & 0xffffu], x1 = tab[wd >> 16];
                s.x += x0.x; s.y += x0.y; s.z += x0.z; s.w += x0.w;
                s.x += x1.x; s.y += x1.y; s.z += x1.z; s.w += x1.w;
                ++pos;
                if ((pos & 3) == 0) {
                    cur = n1; n1 = n2;
                    const int c = (pos >> 2) + 2;
                    if (c < n_chunks) n2 = st[(size_t)c * 64];
                }
            }
            acc[j] = s;
        }
    }
    if (a.stamps && tid == 0) a.stamps[blockIdx.x * 4 + 2] = wall_clock64();
    const int *rowtab = a.tables + u.row_ofs;
    float4 *ys = a.y + (size_t)slice * a.N;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        if (j >= u.J) break;
        const int rr = rowtab[j * 1024 + tid];
        if (rr < 0) continue;
        const int r = rr & 0x3fffffff;
        const float dr = a.dinv[r];
        const float4 v = make_float4(acc[j].x * dr, acc[j].y * dr, acc[j].z * dr, acc[j].w * dr);
        if (rr & 0x40000000) {
            float *yp = reinterpret_cast<float *>(ys + r);
            atomicAdd(yp, v.x); atomicAdd(yp + 1, v.y); atomicAdd(yp + 2, v.z); atomicAdd(yp + 3, v.w);
        } else {
            ys[r] = v;
        }
    }
    if (a.stamps && tid == 0) a.stamps[blockIdx.x * 4 + 3] = wall_clock64();
}

__global__ void zero_rows4_kernel(const int *rows, int n, int n_slices, int N, float4 *y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n_slices) return;
    y[(size_t)(i / n) * N + rows[i % n]] = make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" int panel_spmm(int jmax, const Unit *units, int n_units, const int *tables, const unsigned *stream, const float *dinv, int N, int n_slices,
                          const float *x, float *y, const int *split_rows, int n_split, int lds_bytes, unsigned long long *stamps, hipStream_t s)
{
    PanelArgs a{units, tables, stream, dinv, reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y), N, n_units, n_slices, stamps};
    if (n_split > 0) {
        hipLaunchKernelGGL(zero_rows4_kernel, dim3((n_split * n_slices + 255) / 256), dim3(256), 0, s, split_rows, n_split, n_slices, N, a.y);
    }
    const dim3 grid(n_units * n_slices), block(1024);
#define LAUNCH(JM)                                                                                                      \
    do {                                                                                                                \
        static bool attr = false;                                                                                       \
        if (!attr) { hipFuncSetAttribute((const void *)spmm_panel_kernel<JM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); attr = true; } \
        hipLaunchKernelGGL((spmm_panel_kernel<JM>), grid, block, lds_bytes, s, a);                                      \
    } while (0)
    if (jmax <= 8) LAUNCH(8);
    else if (jmax <= 12) LAUNCH(12);
    else if (jmax <= 16) LAUNCH(16);
    else if (jmax <= 20) LAUNCH(20);
    else return -1;
    return (int)hipGetLastError();
}
