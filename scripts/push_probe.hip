// Probe (not part of the library): how fast is the first backward layer of a LightGCN train step as a PUSH -- for every frontier
// row r (the minibatch's <= 3 B rows, the only non-zero rows of the gradient) and every neighbour c: t[c] += val * g[r] with float
// atomics -- against the frontier-filtered PULL of spmm.h (74 us at the yelp shape)?   push_probe N d n_frontier avg_deg pop_skew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int D>
__global__ __launch_bounds__(256) void push_kernel(const int2 *__restrict__ units, int n_units, int CH, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                  const float *__restrict__ val, const float *__restrict__ g, float *__restrict__ t)
{
    constexpr int G = D / 4, NG = 64 / G;   // lanes per row, neighbour rows per wave step
    const int lane = threadIdx.x & 63, grp = lane / G, sub = lane % G;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int u = wave; u < n_units; u += n_waves) {
        const int2 un = units[u];
        const int r = un.x, e0 = rowptr[r] + un.y * CH, e1 = min(rowptr[r + 1], e0 + CH);
        const float4 x = *reinterpret_cast<const float4 *>(g + (size_t)r * D + sub * 4);
        for (int e = e0 + grp; e < e1; e += NG) {
            const int c = col[e];
            const float v = val[e];
            float *dst = t + (size_t)c * D + sub * 4;
            unsafeAtomicAdd(dst, v * x.x); unsafeAtomicAdd(dst + 1, v * x.y); unsafeAtomicAdd(dst + 2, v * x.z); unsafeAtomicAdd(dst + 3, v * x.w);
        }
    }
}
__global__ void copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, long long n4)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}
int main(int argc, char **argv)
{
    const int N = atoi(argv[1]), d = atoi(argv[2]), nf = atoi(argv[3]);
    const double avg = atof(argv[4]), hot = argc > 5 ? atof(argv[5]) : 200.0;   // a third of the frontier rows are "popular": degree `hot`
    const int CH = 256;
    std::vector<int> rp(N + 1, 0), deg(N);
    unsigned long long z = 99;
    auto rnd = [&]() { z = z * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((z >> 33) & 0x7fffffff) / 2147483648.0; };
    for (int r = 0; r < N; ++r) deg[r] = std::max(1, (int)(avg * (0.25 + 1.5 * rnd())));
    std::vector<int> fr(nf);
    for (int i = 0; i < nf; ++i) { fr[i] = (int)(rnd() * N) % N; if (i % 3 == 1) deg[fr[i]] = (int)(hot * (0.5 + rnd())); }
    for (int r = 0; r < N; ++r) rp[r + 1] = rp[r] + deg[r];
    const size_t nnz = (size_t)rp[N];
    std::vector<int> col(nnz); std::vector<float> val(nnz, 0.01f);
    for (size_t e = 0; e < nnz; ++e) col[e] = (int)(rnd() * N) % N;
    std::vector<int2> units;
    long long fdeg = 0;
    for (int i = 0; i < nf; ++i) { const int r = fr[i]; fdeg += deg[r]; for (int k = 0; k * CH < deg[r]; ++k) units.push_back(make_int2(r, k)); }
    int *drp, *dcol; float *dval, *g, *t; int2 *du;
    CK(hipMalloc(&drp, (N + 1) * 4)); CK(hipMalloc(&dcol, nnz * 4)); CK(hipMalloc(&dval, nnz * 4)); CK(hipMalloc(&g, (size_t)N * d * 4)); CK(hipMalloc(&t, (size_t)N * d * 4));
    CK(hipMalloc(&du, units.size() * 8));
    CK(hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dval, val.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(du, units.data(), units.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(g, 0, (size_t)N * d * 4)); CK(hipMemset(t, 0, (size_t)N * d * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 2048;
    auto push = [&]() {
        if (d == 64) hipLaunchKernelGGL(push_kernel<64>, dim3(grid), dim3(256), 0, 0, du, (int)units.size(), CH, drp, dcol, dval, g, t);
        else hipLaunchKernelGGL(push_kernel<128>, dim3(grid), dim3(256), 0, 0, du, (int)units.size(), CH, drp, dcol, dval, g, t);
    };
    auto copy = [&]() { hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, (const float4 *)g, (float4 *)t, (long long)N * d / 4); };
    for (int i = 0; i < 3; ++i) { copy(); push(); }
    float ms;
    CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) copy(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_copy = ms * 1e3 / 20;
    CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) push(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_push = ms * 1e3 / 20;
    printf("N %d d %d frontier %d rows, %lld neighbour rows, %zu units: copy %.1f us, push %.1f us (%.1f G dword atomics/s)\n", N, d, nf, fdeg, units.size(), us_copy, us_push,
           fdeg * (double)d / us_push / 1e3);
    return 0;
}
