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

#include <stdlib.h>

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

// The same partial products with ONE 16-byte load per lane and operand (8 bytes of halves) instead of four 4-byte ones:
// lane l15 takes columns 4 l15 .. 4 l15 + 3 of its tile and element t feeds MFMA tile t, i.e. MFMA row r of tile t stands
// for column 4 r + t -- a permutation of the 64 columns inside the wavefront's tile, undone in the stores (which become
// 16 bytes per lane too).  Loads through buffer resources: rows past the slab's end and columns past the table read as
// zero without a branch.  Every output still sums its graph rows in the same order: the same bits as the scalar kernel.
// (Reddit shape, 602 x 128: 0.67 -> see profiles; the scalar kernel stays for tables over 4 GiB or unaligned rows.)
typedef unsigned int xtg_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int xtg_u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kXtgOob = 0xFFFFFFF0u;

template <typename TX>
__global__ __launch_bounds__(kBlock) void xtg_partial_vec_kernel(
    int n_rows, int M, int P, const TX *__restrict__ X, unsigned ldx_bytes, const float *__restrict__ G, unsigned ldg_bytes,
    float *__restrict__ partial, int n_slabs, int rows_per_slab, int m_pad, int p_pad)
{
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int slab = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (slab >= n_slabs) return;
    const int m0 = blockIdx.y * kTile, p0 = blockIdx.z * kTile;
    const int64_t n_begin = (int64_t)slab * rows_per_slab;
    const int64_t n_end = n_begin + rows_per_slab < n_rows ? n_begin + rows_per_slab : n_rows;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<TX *>(X), 0, (unsigned)n_rows * ldx_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G), 0, (unsigned)n_rows * ldg_bytes, 0x00020000);
    const int ma = m0 + 4 * l15, pa = p0 + 4 * l15;                 // this lane's first column of X and of G
    const unsigned x_col = ma < M ? (unsigned)ma * (unsigned)sizeof(TX) : kXtgOob;
    const unsigned g_col = pa < P ? (unsigned)pa * 4u : kXtgOob;

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
            const unsigned xo = (ok && x_col != kXtgOob) ? (unsigned)n * ldx_bytes + x_col : kXtgOob;
            const unsigned go = (ok && g_col != kXtgOob) ? (unsigned)n * ldg_bytes + g_col : kXtgOob;
            if constexpr (sizeof(TX) == 2) {
                union { xtg_u32x2 v; f16 h[4]; } ux;
                ux.v = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, xo, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) a[u][t] = ma + t < M ? (float)ux.h[t] : 0.0f;
            } else {
                union { xtg_u32x4 v; float f[4]; } ux;
                ux.v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xo, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) a[u][t] = ma + t < M ? ux.f[t] : 0.0f;
            }
            union { xtg_u32x4 v; float f[4]; } ug;
            ug.v = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, go, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) b[u][t] = pa + t < P ? ug.f[t] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }
    // C/D layout: column = lane & 15 = B's row index (p = p0 + 4 (lane & 15) + j), row = 4 (lane >> 4) + reg = A's row
    // index (m = m0 + 4 row + i): a lane holds 4 consecutive p of one m per (i, reg)
    float *out = partial + (size_t)slab * m_pad * p_pad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * (4 * lq + r) + i;
            *reinterpret_cast<f32x4 *>(out + (size_t)m * p_pad + pa) = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
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
    // 16-byte loads (8 bytes of halves) where the rows are dword-aligned and the tables fit 32-bit offsets
    const size_t es = dtype_x == SGX_F16 ? 2 : 4;
    const bool vec = !getenv("SGX_XTG_SCALAR") && ((uintptr_t)X % 4 == 0) && ((ldx * es) % 4 == 0) && ((uintptr_t)G % 16 == 0) &&
                     ((ldg * 4) % 16 == 0) && (dtype_x == SGX_F16 ? (ldx * es) % 8 == 0 && (uintptr_t)X % 8 == 0 : (ldx * es) % 16 == 0 && (uintptr_t)X % 16 == 0) &&
                     (unsigned long long)n_rows * ldx * es < 0xFFF00000ull && (unsigned long long)n_rows * ldg * 4ull < 0xFFF00000ull;
    if (vec) {
        if (dtype_x == SGX_F16)
            hipLaunchKernelGGL(xtg_partial_vec_kernel<f16>, grid, dim3(kBlock), 0, s, n_rows, M, P, (const f16 *)X, (unsigned)(ldx * 2), G,
                               (unsigned)(ldg * 4), partial, g.n_slabs, g.rows_per_slab, g.m_pad, g.p_pad);
        else
            hipLaunchKernelGGL(xtg_partial_vec_kernel<float>, grid, dim3(kBlock), 0, s, n_rows, M, P, (const float *)X, (unsigned)(ldx * 4), G,
                               (unsigned)(ldg * 4), partial, g.n_slabs, g.rows_per_slab, g.m_pad, g.p_pad);
    } else if (dtype_x == SGX_F16)
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
