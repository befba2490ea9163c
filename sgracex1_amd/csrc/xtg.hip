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

// Workgroup-wide tile: 8 wavefronts as WM x WP sub-tiles of 64 x 64 share the operand rows through LDS, so a slab's rows
// of X and G are read from memory once per workgroup tile (64 WM columns of X, 64 WP of G) instead of once per 64 x 64
// wavefront tile -- at 602 x 128 G was read ten times out of L2, at 128 x 256 X four times.  16 graph rows per step, double
// buffered (the next step's global loads in flight during this step's 64 MFMAs per wavefront, one barrier per step);
// fp16 X is widened as it is staged.  Every output still sums its graph rows in ascending groups of four through the same
// MFMA: the same bits as the kernels above.
constexpr int kWgThreads = 512;
constexpr int kStepRows = 16;

template <typename TX, int WM, int WP>
__global__ __launch_bounds__(kWgThreads) void xtg_partial_wg_kernel(
    int n_rows, int M, int P, const TX *__restrict__ X, unsigned ldx_bytes, const float *__restrict__ G, unsigned ldg_bytes,
    float *__restrict__ partial, int rows_per_slab, int m_pad, int p_pad)
{
    static_assert(WM * WP == kWgThreads / 64, "one 64 x 64 sub-tile per wavefront");
    constexpr int XC = 64 * WM, GC = 64 * WP;                   // columns of X / G per workgroup tile
    constexpr int XCH = kStepRows * XC / 4, GCH = kStepRows * GC / 4;        // 16-byte chunks of a step
    constexpr int XPT = (XCH + kWgThreads - 1) / kWgThreads, GPT = (GCH + kWgThreads - 1) / kWgThreads;
    __shared__ __attribute__((aligned(16))) float sX[2][kStepRows][XC];
    __shared__ __attribute__((aligned(16))) float sG[2][kStepRows][GC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave / WP, wp = wave % WP;
    const int slab = blockIdx.x;
    const int m0 = blockIdx.y * XC, p0 = blockIdx.z * GC;
    const int64_t n_begin = (int64_t)slab * rows_per_slab;
    const int64_t n_end = n_begin + rows_per_slab < n_rows ? n_begin + rows_per_slab : n_rows;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<TX *>(X), 0, (unsigned)n_rows * ldx_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G), 0, (unsigned)n_rows * ldg_bytes, 0x00020000);

    f32x4 rx[XPT], rg[GPT];
    auto gload = [&](int64_t n0) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + kWgThreads * i, r = c / (XC / 4), col = m0 + 4 * (c % (XC / 4));
            const bool in = c < XCH && n0 + r < n_end && col < M;
            f32x4 v;
            if constexpr (sizeof(TX) == 2) {
                union { xtg_u32x2 v; f16 h[4]; } u;
                u.v = __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, in ? (unsigned)(n0 + r) * ldx_bytes + (unsigned)col * 2u : kXtgOob, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = col + t < M ? (float)u.h[t] : 0.0f;
            } else {
                union { xtg_u32x4 v; float f[4]; } u;
                u.v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, in ? (unsigned)(n0 + r) * ldx_bytes + (unsigned)col * 4u : kXtgOob, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = col + t < M ? u.f[t] : 0.0f;
            }
            rx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int c = tid + kWgThreads * i, r = c / (GC / 4), col = p0 + 4 * (c % (GC / 4));
            const bool in = c < GCH && n0 + r < n_end && col < P;
            union { xtg_u32x4 v; float f[4]; } u;
            u.v = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, in ? (unsigned)(n0 + r) * ldg_bytes + (unsigned)col * 4u : kXtgOob, 0, 0);
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = col + t < P ? u.f[t] : 0.0f;
            rg[i] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + kWgThreads * i;
            if (c < XCH) *reinterpret_cast<f32x4 *>(&sX[buf][c / (XC / 4)][4 * (c % (XC / 4))]) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int c = tid + kWgThreads * i;
            if (c < GCH) *reinterpret_cast<f32x4 *>(&sG[buf][c / (GC / 4)][4 * (c % (GC / 4))]) = rg[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

    const int64_t n_steps = (n_end - n_begin + kStepRows - 1) / kStepRows;
    if (n_steps > 0) {
        gload(n_begin);
        sstore(0);
    }
    __syncthreads();
    for (int64_t st = 0; st < n_steps; ++st) {
        const int buf = (int)(st & 1);
        if (st + 1 < n_steps) gload(n_begin + (st + 1) * kStepRows);
#pragma unroll
        for (int u = 0; u < kStepRows / 4; ++u) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&sX[buf][4 * u + lq][64 * wm + 4 * l15]);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(&sG[buf][4 * u + lq][64 * wp + 4 * l15]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (st + 1 < n_steps) sstore(buf ^ 1);          // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    // as in xtg_partial_vec_kernel: m = tile + 4 (4 lq + r) + i, p = tile + 4 l15 + j
    float *out = partial + (size_t)slab * m_pad * p_pad;
    const int mt = m0 + 64 * wm, pt = p0 + 64 * wp + 4 * l15;
    if (mt < m_pad && pt < p_pad) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt + 4 * (4 * lq + r) + i;
                *reinterpret_cast<f32x4 *>(out + (size_t)m * p_pad + pt) = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            }
    }
}

__global__ __launch_bounds__(kBlock) void xtg_reduce_kernel(int M, int P, const float *__restrict__ partial, int n_slabs,
                                                            int m_pad, int p_pad, float *__restrict__ out, int64_t ldo)
{
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= M * P) return;
    const int m = gid / P, p = gid % P;
    // eight slabs requested at a time, added in slab order (one load at a time left the 30 K - 80 K threads of this kernel
    // waiting on memory: 0.1 ms, as long as the partial products themselves)
    const size_t stride = (size_t)m_pad * p_pad;
    const float *src = partial + (size_t)m * p_pad + p;
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= n_slabs; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < n_slabs; ++k) s += src[(size_t)k * stride];
    out[(int64_t)m * ldo + p] = s;
}

struct XtgGeom { int n_slabs, rows_per_slab, m_pad, p_pad, wm; };

// Slabs of graph rows and the arrangement of a workgroup's 8 sub-tiles (wm x 8 / wm).  The slab count is chosen so that
// slabs x workgroup tiles fills the device once (two workgroups of 8 wavefronts per CU): with 612 workgroups on 512
// places the second round ran a fifth full and the Reddit shape took 0.63 ms instead of 0.35.
XtgGeom geometry(int n_rows, int M, int P)
{
    XtgGeom g;
    g.m_pad = (M + kTile - 1) / kTile * kTile;
    g.p_pad = (P + kTile - 1) / kTile * kTile;
    const int mt = g.m_pad / kTile, pt = g.p_pad / kTile;
    int64_t best = (int64_t)1 << 40;
    g.wm = 8;
    for (int wm = 8; wm >= 1; wm >>= 1) {
        const int wp = 8 / wm;
        const int64_t tiles = (int64_t)((mt + wm - 1) / wm) * ((pt + wp - 1) / wp);
        if (tiles < best) { best = tiles; g.wm = wm; }
    }
    const size_t tile_bytes = (size_t)g.m_pad * g.p_pad * sizeof(float);
    int64_t slabs = 512 / best;                                  // one round of workgroups on 256 CUs
    const int64_t by_rows = (n_rows + 63) / 64;                  // >= 64 graph rows per slab (a MUTAG batch of 3.4 K rows: 53 slabs
                                                                 // -- with 256 rows per slab 13 wavefronts did all the work: 29 us)
    const int64_t by_mem = (int64_t)((64u << 20) / tile_bytes);  // partials capped at 64 MiB
    if (slabs > by_rows) slabs = by_rows;
    if (slabs > by_mem) slabs = by_mem;
    if (slabs < 1) slabs = 1;
    int64_t rps = (n_rows + slabs - 1) / slabs;
    rps = (rps + 15) / 16 * 16;
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
    // (a 16- or 8-byte buffer load wants dword alignment, no more: rows of 602 floats qualify)
    const bool vec = !sgx_tune().xtg_scalar && ((uintptr_t)X % 4 == 0) && ((ldx * es) % 4 == 0) && ((uintptr_t)G % 4 == 0) &&
                     (unsigned long long)n_rows * ldx * es < 0xFFF00000ull && (unsigned long long)n_rows * ldg * 4ull < 0xFFF00000ull;
    if (vec && n_rows >= 16384 && !sgx_tune().xtg_wave_tiles) {
        // enough rows to fill the device with workgroup tiles: the arrangement of the 8 sub-tiles with the fewest tiles
        const int mt = g.m_pad / kTile, pt = g.p_pad / kTile;
        const int best_wm = g.wm;
#define SGX_XTG_WG(TX_, WM_, WP_)                                                                                              \
    hipLaunchKernelGGL((xtg_partial_wg_kernel<TX_, WM_, WP_>), dim3(g.n_slabs, (mt + WM_ - 1) / WM_, (pt + WP_ - 1) / WP_),        \
                       dim3(kWgThreads), 0, s, n_rows, M, P, (const TX_ *)X, (unsigned)(ldx * es), G, (unsigned)(ldg * 4), partial, \
                       g.rows_per_slab, g.m_pad, g.p_pad)
#define SGX_XTG_WG_T(TX_)                                                                                                      \
    switch (best_wm) {                                                                                                         \
    case 8: SGX_XTG_WG(TX_, 8, 1); break;                                                                                      \
    case 4: SGX_XTG_WG(TX_, 4, 2); break;                                                                                      \
    case 2: SGX_XTG_WG(TX_, 2, 4); break;                                                                                      \
    default: SGX_XTG_WG(TX_, 1, 8); break;                                                                                     \
    }
        if (dtype_x == SGX_F16) { SGX_XTG_WG_T(f16) } else { SGX_XTG_WG_T(float) }
#undef SGX_XTG_WG_T
#undef SGX_XTG_WG
    } else if (vec) {
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
