// Shared host/device helpers for librecad_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/recad_hip.h"

#define RK_EXPORT extern "C" __attribute__((visibility("default")))

extern thread_local char rk_err_buf[512];

#define RK_FAIL(code, ...)                                        \
    do {                                                          \
        snprintf(rk_err_buf, sizeof(rk_err_buf), __VA_ARGS__);    \
        return (code);                                            \
    } while (0)

#define RK_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) RK_FAIL(RK_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                        \
    } while (0)

#define RK_CHECK_LAUNCH() RK_HIP(hipGetLastError())

#include "host/layout.h"   // RK_TUNE_INT, plan / schedule layouts (pure C++, shared with the host-only builders)

static constexpr int kWave = 64;

// "Done once per DEVICE" flag for hipFuncSetAttribute (the attribute is applied to the current device's code object; a
// process that moves to a second GPU must set it there too).  The call is idempotent, so two host threads racing here at
// worst both make it.
struct RkPerDeviceOnce {
    unsigned long long mask[4] = {0ULL, 0ULL, 0ULL, 0ULL};   // up to 256 devices
    bool need(int *dev_out)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) { *dev_out = -1; return true; }
        *dev_out = dev;
        return !(__atomic_load_n(&mask[dev >> 6], __ATOMIC_ACQUIRE) & (1ULL << (dev & 63)));
    }
    void done(int dev) { if (dev >= 0) __atomic_fetch_or(&mask[dev >> 6], 1ULL << (dev & 63), __ATOMIC_RELEASE); }
};

// Row r of a logical [n_rows, d] matrix stored as two blocks (rows < split in lo, the rest in hi).
__device__ __forceinline__ const float *row2(const float *lo, const float *hi, int split, int r, int d)
{
    return r < split ? lo + (size_t)r * d : hi + (size_t)(r - split) * d;
}
__device__ __forceinline__ float *row2(float *lo, float *hi, int split, int r, int d)
{
    return r < split ? lo + (size_t)r * d : hi + (size_t)(r - split) * d;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Stream-ordered zero fill by a kernel.  Used instead of hipMemsetAsync where the NEXT launch on the stream
// accumulates into the buffer with atomics: as the trailing node of a replayed hipGraph a memset node was
// observed to overlap the next replay's first kernel on this stack (lightgcn.hip, launch_backward), so
// buffers that are zeroed and then added into are cleared by a kernel.
static __global__ void rk_zero_kernel(unsigned *p, long long n_words)
{
    const long long stride = (long long)gridDim.x * blockDim.x, t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
    const long long n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? n_words / 4 : 0;
    for (long long i = t; i < n4; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (long long i = n4 * 4 + t; i < n_words; i += stride) p[i] = 0u;
}
static inline hipError_t rk_zero_async(void *p, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    if (bytes & 3) return hipMemsetAsync(p, 0, bytes, s);
    const long long n = (long long)(bytes / 4);
    const int grid = (int)((n / 4 + 255) / 256 < 1 ? 1 : ((n / 4 + 255) / 256 > 2048 ? 2048 : (n / 4 + 255) / 256));
    hipLaunchKernelGGL(rk_zero_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<unsigned *>(p), n);
    return hipGetLastError();
}

// Adam coefficients for 1-based step t, torch.optim.Adam single-tensor math:
// step_size = lr / (1 - b1^t); bc2s = sqrt(1 - b2^t).  Computed in double like Python does.
struct AdamCoef {
    float step_size, bc2s;
};
__host__ __device__ inline AdamCoef adam_coef(int t, float lr, float b1, float b2)
{
    double bc1 = 1.0 - pow((double)b1, (double)t);
    double bc2 = 1.0 - pow((double)b2, (double)t);
    AdamCoef c;
    c.step_size = (float)((double)lr / bc1);
    c.bc2s = (float)sqrt(bc2);
    return c;
}

// One element of torch.optim.Adam (lerp / addcmul / addcdiv order).  Every operation is rounded on its own (no fused
// multiply-add), like the oracle's orc_adam (gcc -ffp-contract=off): left to the compiler, `v * b2 + w2 * g * g` was
// contracted one way in one kernel and another way in the next (the vector and the scalar loop of one kernel differed in the
// last bit), so the Adam paths -- SpMM epilogues, standalone kernels, captured and eager steps -- did not agree with each other
// or with the oracle to the bit on the same gradient.  (Division and square root are correctly rounded: hipcc's default.)
__device__ __forceinline__ void adam_elem(float &p, float &m, float &v, float g, float w1, float b2, float w2,
                                          float step_size, float bc2s, float eps)
{
#pragma clang fp contract(off)
    m = m + w1 * (g - m);
    v = v * b2 + w2 * g * g;
    float denom = sqrtf(v) / bc2s + eps;
    p = p - step_size * (m / denom);
}

// device state words (int32) shared by the kernels of a training epoch
enum { ST_STEP_BASE = 0, ST_ADAM_T = 1, ST_NTRIP_LO = 2, ST_NTRIP_HI = 3, ST_BATCH = 4, ST_DROP_LO = 5, ST_DROP_HI = 6,
       // 64-bit device pointers of the epoch's triplets and loss partials: the kernels of a train step read them from here, so a
       // captured hipGraph is independent of the caller's buffers
       ST_PTR_USERS = 8, ST_PTR_POS = 10, ST_PTR_NEG = 12, ST_PTR_LOSS = 14, ST_WORDS = 16 };
template <typename T>
__device__ __forceinline__ T *st_ptr(const int *state, int word)
{
    return reinterpret_cast<T *>(*reinterpret_cast<const unsigned long long *>(state + word));
}

// splitmix64 finaliser: the counter-based RNG of the samplers and of the graph dropout
__host__ __device__ __forceinline__ unsigned long long rk_mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// Graph dropout (lightgcn.py:62-71): stored entry `id` of the adjacency survives this step iff its 24-bit
// uniform is below keep_prob * 2^24.  seed_step = rk_drop_step_seed(base seed, step index).
__host__ __device__ __forceinline__ unsigned long long rk_drop_step_seed(unsigned long long base, unsigned long long step)
{
    return rk_mix64(base ^ rk_mix64(step));
}
__host__ __device__ __forceinline__ bool rk_drop_keep(unsigned long long seed_step, unsigned id, unsigned thresh24)
{
    return (unsigned)(rk_mix64(seed_step ^ ((unsigned long long)id * 0xD1342543DE82EF95ULL)) >> 40) < thresh24;
}
