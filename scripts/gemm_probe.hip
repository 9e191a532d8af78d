// Timing probe for gemm.h (not part of the library): gemm_probe M N K [pad_a] [pad_b] [split] [forms: 0 = A[M,K] B[N,K]; 1 = dX form; 2 = dW form] [gather: 1 = rows of A through an index list]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
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
    g.M = M; g.N = N; g.K = K; g.A = A; g.B = B; g.C = C; g.ldc = N;
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
    return 0;
}
