// Probe (tuning only): do the fp32 MFMAs of one wave and the VALU work of another wave of the SAME SIMD execute together on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_probe scripts/mfma_valu_probe.hip && /tmp/mfma_valu_probe
// One 512-thread workgroup per CU (waves w and w + 4 share a SIMD).  Modes:
//   0  waves 0-3 run N dependency-free v_mfma_f32_16x16x4_f32 (4 accumulators round robin), waves 4-7 exit
//   1  waves 4-7 run M dependent-chain VALU instructions (8 independent v_fma chains), waves 0-3 exit
//   2  both at once
//   3  ONE wave per SIMD runs the MFMAs with F VALU instructions after each MFMA (waves 4-7 exit)
//   4  both kinds of waves run the mixed stream of mode 3 (half the work each)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int F>
__global__ __launch_bounds__(512) void probe(float *out, int n_iter)
{
    const int w = threadIdx.x >> 6;
    const bool mf = w < 4;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 1.0f + 1e-7f * (threadIdx.x + j);
    const float a = 1.0f + 1e-6f * threadIdx.x, b = 0.5f;
    if (MODE == 0 || MODE == 2) {
        if (mf) {
            for (int it = 0; it < n_iter; ++it) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q & 3], 0, 0, 0);
            }
        }
    }
    if (MODE == 1 || MODE == 2) {
        if (!mf) {
            for (int it = 0; it < n_iter; ++it) {
#pragma unroll
                for (int q = 0; q < 16 * 7; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 0.999f, 0.001f);
            }
        }
    }
    if (MODE == 3 || MODE == 4) {
        if (mf || MODE == 4) {
            const int iters = MODE == 4 ? n_iter / 2 : n_iter;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q & 3], 0, 0, 0);
#pragma unroll
                    for (int f = 0; f < F; ++f) v[(q * F + f) & 7] = __builtin_fmaf(v[(q * F + f) & 7], 0.999f, 0.001f);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int F>
static float run(float *out, int n_iter)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, F>), dim3(256), dim3(512), 0, 0, out, n_iter);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((probe<MODE, F>), dim3(256), dim3(512), 0, 0, out, n_iter);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1000.f;
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    const int n = 2000;   // x 16 MFMAs per wave
    printf("per wave: %d MFMAs (16x16x4 f32, 32 cycles each on the pipe) and/or %d dependent-chain VALU instructions\n", n * 16, n * 16 * 7);
    const float t0 = run<0, 0>(out, n), t1 = run<1, 0>(out, n), t2 = run<2, 0>(out, n);
    printf("mode 0 MFMA waves alone            %8.1f us  (%.1f cycles per MFMA at 2.4 GHz)\n", t0, t0 * 2400.f / (n * 16));
    printf("mode 1 VALU waves alone            %8.1f us  (%.2f cycles per VALU instruction)\n", t1, t1 * 2400.f / (n * 16 * 7));
    printf("mode 2 both, different waves       %8.1f us  (sum %.1f, max %.1f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
    printf("mode 3 one wave, F VALU per MFMA:  F=0 %.1f  F=2 %.1f  F=4 %.1f  F=6 %.1f  F=7 %.1f  F=8 %.1f  F=12 %.1f us\n", run<3, 0>(out, n), run<3, 2>(out, n), run<3, 4>(out, n),
           run<3, 6>(out, n), run<3, 7>(out, n), run<3, 8>(out, n), run<3, 12>(out, n));
    printf("mode 4 two waves per SIMD, mixed:  F=4 %.1f  F=7 %.1f  F=12 %.1f us (half the iterations each)\n", run<4, 4>(out, n), run<4, 7>(out, n), run<4, 12>(out, n));
    return 0;
}
