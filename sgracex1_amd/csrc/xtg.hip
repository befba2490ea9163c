// grad_W = X^T . G  for dense X -- the weight gradient of the layer's backward pass
// (FPYNQ.backward, MOL cell 16: `grad_weights = input.t() @ adj @ grad_output`, with G = adj @ grad_output
// already aggregated by the CSR kernel; SG.py:1094-1103 likewise).  The reference computes it in fp32
// torch on the ARM CPU (0.19-0.28 s per layer on MUTAG); "next" row f1 of the scope table.
//
// Shape: X [n_rows][M] (fp16 or fp32), G [n_rows][P] fp32, out [M][P] fp32: a tall-skinny TN product
// whose reduction runs over the millions of rows.  `v_mfma_f32_16x16x4_f32` (exact fp32 fma chain)
// fits it without any transpose: its A operand is ONE value per lane with lanes 0-15 along the OUTPUT
// row (here m) and lanes>>4 along K (here the graph row n), so a wavefront reads 4 graph rows x 16
// consecutive columns of X -- row-major as stored -- per instruction; B likewise from G.
// A wavefront reduces one slab of graph rows into a 64 x 64 tile of fp32 partials; the slabs are
// added in slab order by a second kernel (bitwise reproducible, no atomics).
#include "sgx_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 64;          // output tile edge per wavefront (4 x 4 MFMA tiles)
constexpr int kUnroll = 4;         // graph-row quads in flight per iteration

template <typename TX>
__device__ __forceinline__ float ld_x(const TX *p) { return (float)*p; }

template <typename TX>
__global__ __launch_bounds__(kBlock) void xtg_partial_kernel(
    int n_rows, int M, int P, const TX *__restrict__ X, int64_t ldx, const float *__restrict__ G, int64_t ldg,
    float *__restrict__ partial, int n_slabs, int rows_per_slab, int m_pad, int p_pad)
{
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int slab = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (slab >= n_slabs) return;
    const int m0 = blockIdx.y * kTile, p0 = blockIdx.z * kTile;
    const int64_t n_begin = (int64_t)slab * rows_per_slab;
    const int64_t n_end = n_begin + rows_per_slab < n_rows ? n_begin + rows_per_slab : n_rows;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

    for (int64_t n0 = n_begin; n0 < n_end; n0 += 4 * kUnroll) {
        float a[kUnroll][4], b[kUnroll][4];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t n = n0 + 4 * u + lq;
            const bool ok = n < n_end;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + t * 16 + l15, p = p0 + t * 16 + l15;
                a[u][t] = (ok && m < M) ? ld_x(X + n * ldx + m) : 0.0f;
                b[u][t] = (ok && p < P) ? G[n * ldg + p] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }
    // C/D layout: column = lane & 15 (p), row = 4 * (lane >> 4) + reg (m)
    float *out = partial + (size_t)slab * m_pad * p_pad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + 4 * lq + r, p = p0 + j * 16 + l15;
                out[(size_t)m * p_pad + p] = acc[i][j][r];
            }
}

__global__ __launch_bounds__(kBlock) void xtg_reduce_kernel(int M, int P, const float *__restrict__ partial, int n_slabs,
                                                            int m_pad, int p_pad, float *__restrict__ out, int64_t ldo)
{
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= M * P) return;
    const int m = gid / P, p = gid % P;
    float s = 0.0f;
    for (int k = 0; k < n_slabs; ++k) s += partial[((size_t)k * m_pad + m) * p_pad + p];
    out[(int64_t)m * ldo + p] = s;
}

struct XtgGeom { int n_slabs, rows_per_slab, m_pad, p_pad; };

XtgGeom geometry(int n_rows, int M, int P)
{
    XtgGeom g;
    g.m_pad = (M + kTile - 1) / kTile * kTile;
    g.p_pad = (P + kTile - 1) / kTile * kTile;
    const size_t tile_bytes = (size_t)g.m_pad * g.p_pad * sizeof(float);
    int64_t slabs = (n_rows + 255) / 256;                        // >= 256 graph rows per slab
    const int64_t by_mem = (int64_t)((64u << 20) / tile_bytes);  // partials capped at 64 MiB
    if (slabs > 2048) slabs = 2048;
    if (slabs > by_mem) slabs = by_mem;
    if (slabs < 1) slabs = 1;
    int64_t rps = (n_rows + slabs - 1) / slabs;
    rps = (rps + 3) / 4 * 4;
    g.rows_per_slab = (int)rps;
    g.n_slabs = (int)((n_rows + rps - 1) / rps);
    if (g.n_slabs < 1) g.n_slabs = 1;
    return g;
}

}  // namespace

extern "C" size_t sgx_xt_g_workspace_bytes(int n_rows, int M, int P)
{
    if (n_rows < 0 || M < 1 || P < 1) return 0;
    const XtgGeom g = geometry(n_rows, M, P);
    return sgx_align_up((size_t)g.n_slabs * g.m_pad * g.p_pad * sizeof(float), 256);
}

extern "C" int sgx_xt_g(int dtype_x, int n_rows, int M, int P, const void *X, int64_t ldx, const float *G, int64_t ldg,
                        float *out, int64_t ldo, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n_rows < 0 || M < 1 || P < 1 || ldx < M || ldg < P || ldo < P) return SGX_ERR_SHAPE;
    if (!out) return SGX_ERR_NULL;
    if (dtype_x != SGX_F16 && dtype_x != SGX_F32) return SGX_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (n_rows == 0) {
        for (int m = 0; m < M; ++m) SGX_HIP_CHECK(hipMemsetAsync(out + (int64_t)m * ldo, 0, sizeof(float) * P, s));
        return SGX_OK;
    }
    if (!X || !G) return SGX_ERR_NULL;
    const XtgGeom g = geometry(n_rows, M, P);
    if (!workspace || workspace_bytes < (size_t)g.n_slabs * g.m_pad * g.p_pad * sizeof(float)) return SGX_ERR_WORKSPACE;
    float *partial = (float *)workspace;
    dim3 grid((g.n_slabs + kBlock / 64 - 1) / (kBlock / 64), g.m_pad / kTile, g.p_pad / kTile);
    if (dtype_x == SGX_F16)
        hipLaunchKernelGGL(xtg_partial_kernel<f16>, grid, dim3(kBlock), 0, s, n_rows, M, P, (const f16 *)X, ldx, G, ldg,
                           partial, g.n_slabs, g.rows_per_slab, g.m_pad, g.p_pad);
    else
        hipLaunchKernelGGL(xtg_partial_kernel<float>, grid, dim3(kBlock), 0, s, n_rows, M, P, (const float *)X, ldx, G,
                           ldg, partial, g.n_slabs, g.rows_per_slab, g.m_pad, g.p_pad);
    SGX_LAUNCH_CHECK();
    hipLaunchKernelGGL(xtg_reduce_kernel, dim3((M * P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, M, P, partial, g.n_slabs,
                       g.m_pad, g.p_pad, out, ldo);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}
