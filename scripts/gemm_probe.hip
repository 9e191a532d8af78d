// Timing probe for gemm.h (not part of the library): gemm_probe M N K [pad_a] [pad_b] [split] [forms: 0 = A[M,K] B[N,K]; 1 = dX form; 2 = dW form] [gather: 1 = rows of A through an index list] [ldc: 1 = rows of C padded to 32 floats]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifdef GEMM_STAMPS   // per-workgroup, per-tile wall-clock stamps of gemm_f32_wide_kernel (100 MHz): [wg][8 iterations][8 stamps]
__device__ unsigned long long *g_gemm_stamps;
#define GEMM_STAMP(it, i) do { if (g_gemm_stamps && threadIdx.x == 0 && (it) < 8) g_gemm_stamps[((size_t)blockIdx.x * 8 + (it)) * 8 + (i)] = wall_clock64(); } while (0)
#endif
#include "../recad_amd/csrc/gemm.h"
int main(int argc, char **argv)
{
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int pa = argc > 4 ? atoi(argv[4]) : 0, pb = argc > 5 ? atoi(argv[5]) : 0, split = argc > 6 ? atoi(argv[6]) : 1;
    const int form = argc > 7 ? atoi(argv[7]) : 0;
    float *A, *B, *C, *P;
    const size_t na = (size_t)(form == 2 ? K : M) * ((form == 2 ? M : K) + pa), nb = (size_t)(form == 0 ? N : K) * ((form == 0 ? K : N) + pb);
    hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&P, (size_t)M * N * 4 * (split > 1 ? split : 1));
    std::vector<float> h(na > nb ? na : nb);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 255) / 256.f - 0.5f;
    hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), nb * 4, hipMemcpyHostToDevice);
    GemmArgs g{};
    // argv[9] = 1: rows of C padded to 32 floats (128-byte lines), as rk_score_topk's score matrix is (plan.ld_scores)
    const long long ldc = (argc > 9 && atoi(argv[9])) ? (((long long)N + 31) & ~31LL) : N;
    if (ldc != N) { hipFree(C); hipMalloc(&C, (size_t)M * ldc * 4); }
    g.M = M; g.N = N; g.K = K; g.A = A; g.B = B; g.C = C; g.ldc = ldc;
    if (form == 2) { g.a_rs = 1; g.a_cs = M + pa; } else { g.a_rs = K + pa; g.a_cs = 1; }
    if (form == 0) { g.b_rs = K + pb; g.b_cs = 1; } else { g.b_rs = 1; g.b_cs = N + pb; }
    if (argc > 8 && atoi(argv[8])) {
        std::vector<int> idx(M);
        for (int i = 0; i < M; ++i) idx[i] = (int)((i * 7919LL) % M);
        int *d; hipMalloc(&d, M * 4); hipMemcpy(d, idx.data(), M * 4, hipMemcpyHostToDevice);
        g.a_ridx = d;
    }
    g.split_k = split; if (split > 1) { g.sk_part = P; g.sk_stride = (long long)M * N; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) gemm_f32_launch(g, 0);
    hipEventRecord(e0, 0);
    const int it = 50;
    for (int i = 0; i < it; ++i) gemm_f32_launch(g, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> c(16); hipMemcpy(c.data(), C, 64, hipMemcpyDeviceToHost);
    printf("M %d N %d K %d pad %d %d split %d form %d: %.2f us  %.1f TF/s  (c0 %g, err %d)\n", M, N, K, pa, pb, split, form, ms * 1e3 / it,
           2.0 * M * N * K / (ms * 1e-3 / it) / 1e12, c[0], (int)hipGetLastError());
#ifdef GEMM_STAMPS
    {
        const int nwg = 512;
        unsigned long long *dst; hipMalloc(&dst, (size_t)nwg * 64 * 8); hipMemset(dst, 0, (size_t)nwg * 64 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &dst, sizeof(dst));
        hipDeviceSynchronize();
        gemm_f32_launch(g, 0); hipDeviceSynchronize();
        std::vector<unsigned long long> st((size_t)nwg * 64);
        hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ULL, t1 = 0;
        for (int w = 0; w < nwg; ++w) { if (st[(size_t)w * 64]) t0 = std::min(t0, st[(size_t)w * 64]); for (int k = 0; k < 64; ++k) t1 = std::max(t1, st[(size_t)w * 64 + k]); }
        printf("stamps (us from the first workgroup's start; mean over workgroups that ran the tile):\n  kernel span %.2f us\n", (t1 - t0) / 100.0);
        double s0 = 0, s1 = 0; int n0 = 0;
        for (int w = 0; w < nwg; ++w) { const unsigned long long *p = &st[(size_t)w * 64]; if (!p[0]) continue; s0 += (p[0] - t0) / 100.0; s1 += (p[1] - p[0]) / 100.0; ++n0; }
        printf("  workgroup start %.2f, prologue (index gather, first tile's loads, LDS image, barrier) %.2f\n", s0 / n0, s1 / n0);
        for (int itn = 0; itn < 4; ++itn) {
            double a[7] = {0}; int n = 0; double beg = 0;
            for (int w = 0; w < nwg; ++w) { const unsigned long long *p = &st[((size_t)w * 8 + itn) * 8]; if (!p[6] || !p[2]) continue; ++n; beg += (p[2] - t0) / 100.0;
                a[3] += (p[3] - p[2]) / 100.0; a[4] += (p[4] - p[3]) / 100.0; a[5] += (p[5] - p[4]) / 100.0; a[6] += (p[6] - p[5]) / 100.0; }
            if (n) printf("  tile %d (%d workgroups): begins %.2f | prefetch + MFMA phase %.2f | epilogue stores %.2f | barrier %.2f | LDS image + barrier %.2f\n", itn, n, beg / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n);
        }
    }
#endif
    return 0;
}
