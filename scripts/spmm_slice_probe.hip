// Review item 8, measured first (probe, not product): a row-gather SpMM whose source table is cut into column slices that fit
// an XCD's 4 MiB L2.  Y[:, s] = A . X[:, s] per slice; tables slice-major [d/D][N][D].  One launch walks the slices in order
// (blockIdx = slice * blocks + b, so the chip works on ONE slice table at a time and every XCD's L2 holds it), and inside a
// slice the user-row pieces come before the item-row pieces (one class's slice is live at a time).
// Mapping: G = D/4 lanes own one PIECE (a row, or <= `cap` consecutive nonzeros of a long row) and walk it in CSR order with
// UN * G 16-byte gathers in flight per lane; 64/G pieces per wave, pieces sorted by length so a wave's groups finish together.
// Pieces of split rows meet through float atomics on a zeroed row (probe only: the product would order them).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/spmm_slice_probe.hip -o recad_amd/lib/libspmm_slice_probe.so
#include <hip/hip_runtime.h>

struct SliceArgs {
    const int4 *pieces;   // {row, e_begin, e_end, split}
    int n_pieces, n_blocks, n_rows;
    const int *col;
    const float *val;
    const float *x;       // [n_slices][n_rows][D]
    float *y;             // same layout
};

template <int D, int UN>
__global__ __launch_bounds__(256) void spmm_slice_kernel(const SliceArgs a)
{
    constexpr int G = D / 4, NG = 64 / G;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = lane / G, sub = lane % G;
    const int slice = blockIdx.x / a.n_blocks, b = blockIdx.x % a.n_blocks;
    const int piece = (b * 4 + w) * NG + grp;
    int4 pd = make_int4(-1, 0, 0, 0);
    if (piece < a.n_pieces) pd = a.pieces[piece];
    const float *__restrict__ xs = a.x + (size_t)slice * a.n_rows * D;
    const int *__restrict__ col = a.col;
    const float *__restrict__ val = a.val;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = pd.y;
    int c[UN];
    float v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int i = e + u * G + sub;
        c[u] = 0; v[u] = 0.f;
        if (i < pd.z) { c[u] = col[i]; v[u] = val[i]; }
    }
    while (__any(e < pd.z)) {
        int cn[UN];
        float vn[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = e + (UN + u) * G + sub;
            cn[u] = 0; vn[u] = 0.f;
            if (i < pd.z) { cn[u] = col[i]; vn[u] = val[i]; }
        }
        float4 xv[UN * G];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int cc = __shfl(c[u], grp * G + k, 64);   // (entries past the end: column 0 with value 0)
                xv[u * G + k] = *reinterpret_cast<const float4 *>(xs + (size_t)(unsigned)cc * D + sub * 4);
            }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const float vv = __shfl(v[u], grp * G + k, 64);
                const float4 t = xv[u * G + k];
                acc.x += vv * t.x; acc.y += vv * t.y; acc.z += vv * t.z; acc.w += vv * t.w;
            }
        e += UN * G;
#pragma unroll
        for (int u = 0; u < UN; ++u) { c[u] = cn[u]; v[u] = vn[u]; }
    }
    if (pd.x >= 0) {
        float *yp = a.y + (size_t)slice * a.n_rows * D + (size_t)pd.x * D + sub * 4;
        if (pd.w) {
            atomicAdd(yp, acc.x); atomicAdd(yp + 1, acc.y); atomicAdd(yp + 2, acc.z); atomicAdd(yp + 3, acc.w);
        } else {
            *reinterpret_cast<float4 *>(yp) = acc;
        }
    }
}

__global__ void zero_rows_kernel(const int *rows, int n, int n_slices, int n_rows, int D, float *y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = n * D;
    if (i >= per * n_slices) return;
    const int s = i / per, j = i % per;
    y[(size_t)s * n_rows * D + (size_t)rows[j / D] * D + j % D] = 0.f;
}

extern "C" int slice_spmm(int D, int un, const int4 *pieces, int n_pieces, const int *split_rows, int n_split, const int *col, const float *val,
                          int n_rows, int n_slices, const float *x, float *y, hipStream_t s)
{
    const int NG = 64 / (D / 4);
    SliceArgs a{pieces, n_pieces, (n_pieces + 4 * NG - 1) / (4 * NG), n_rows, col, val, x, y};
    if (n_split > 0) {
        const int tot = n_split * D * n_slices;
        hipLaunchKernelGGL(zero_rows_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, split_rows, n_split, n_slices, n_rows, D, y);
    }
    const dim3 grid(a.n_blocks * n_slices), block(256);
    if (D == 16 && un == 1) hipLaunchKernelGGL((spmm_slice_kernel<16, 1>), grid, block, 0, s, a);
    else if (D == 16 && un == 2) hipLaunchKernelGGL((spmm_slice_kernel<16, 2>), grid, block, 0, s, a);
    else if (D == 16 && un == 4) hipLaunchKernelGGL((spmm_slice_kernel<16, 4>), grid, block, 0, s, a);
    else if (D == 8 && un == 2) hipLaunchKernelGGL((spmm_slice_kernel<8, 2>), grid, block, 0, s, a);
    else if (D == 8 && un == 4) hipLaunchKernelGGL((spmm_slice_kernel<8, 4>), grid, block, 0, s, a);
    else if (D == 32 && un == 1) hipLaunchKernelGGL((spmm_slice_kernel<32, 1>), grid, block, 0, s, a);
    else if (D == 32 && un == 2) hipLaunchKernelGGL((spmm_slice_kernel<32, 2>), grid, block, 0, s, a);
    else if (D == 64 && un == 1) hipLaunchKernelGGL((spmm_slice_kernel<64, 1>), grid, block, 0, s, a);
    else return -1;
    return (int)hipGetLastError();
}
