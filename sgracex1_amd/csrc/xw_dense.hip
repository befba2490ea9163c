// Dense feature x weight product  H = X . W  on the gfx950 matrix cores -- the X.W stage of the
// reference in gemm_mode 1 (loop_fea / compute1 / dsp_kernel_wrapper_fea with rnnz = M_fea,
// K.cpp:2932, :2605, :847-865, :985-1012).
//
// Shapes are skinny (M_fea 7..602, P 2..256, n_rows up to millions), so the kernel is bound by
// streaming X once from HBM and writing H once; W^T (= the reference's B buffer, [P][M_fea],
// K contiguous) stays in L2/L1.  Orientation: the MFMA computes the TRANSPOSED tile
//     H^T[n][m] = sum_k Wt[n][k] * X[m][k]
// with A := Wt rows and B := X rows, because (a) both operands are then 16 contiguous bytes per
// lane along K straight from their row-major storage -- no LDS staging, no transposes -- and
// (b) in the 16x16 C/D layout (col = lane&15 = m, row = 4*(lane>>4)+reg = n) a lane ends up
// holding 4 consecutive columns n of one row m of H: one 8-byte store per lane.
//
// fp16: v_mfma_f32_16x16x32_f16, fp32 accumulate, one rounding to fp16.
// fp32: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
#include "sgx_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kBlock = 256;

// Rows of X need not be 16-byte aligned (M_fea = 602 or 100 gives 4- / 8-byte aligned rows): gfx950
// global loads only need element alignment, so the fragment is still ONE dwordx4 load through an
// under-aligned vector type.
typedef f16x8 f16x8_u __attribute__((aligned(2)));

// K.cpp:2586-2590 on a rounded value: keep when (v > 0 || relu == 0), else +0
__device__ __forceinline__ f16 relu_f16(f16 v, int relu) { return (!relu || v > (f16)0) ? v : (f16)0; }
typedef f32x4 f32x4_u __attribute__((aligned(4)));

// 8 consecutive K elements of one row, zero beyond k_end
__device__ __forceinline__ f16x8 load_k8(const f16 *__restrict__ row, int k, int k_end, bool row_ok, bool aligned)
{
    (void)aligned;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!row_ok) return v;
    if (k + 8 <= k_end) return *reinterpret_cast<const f16x8_u *>(row + k);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (k + j < k_end) v[j] = row[k + j];
    return v;
}

__device__ __forceinline__ f32x4 load_k4(const float *__restrict__ row, int k, int k_end, bool row_ok, bool aligned)
{
    (void)aligned;
    f32x4 v = {0, 0, 0, 0};
    if (!row_ok) return v;
    if (k + 4 <= k_end) return *reinterpret_cast<const f32x4_u *>(row + k);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (k + j < k_end) v[j] = row[k + j];
    return v;
}

// One wavefront: MT tiles of 16 rows of X  x  NT tiles of 16 columns of W.
template <int NT, int MT>
__global__ __launch_bounds__(kBlock) void xw_dense_f16_kernel(
    int n_rows, int M, int P, int p_base, const f16 *__restrict__ X, int64_t ldx, const f16 *__restrict__ Wt, int64_t ldw,
    f16 *__restrict__ H, int64_t ldh, int x_aligned, int w_aligned, int h_aligned, int relu)
{
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t row0 = wave * (MT * 16);
    if (row0 >= n_rows) return;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};

    // the fragments of k-step s+1 are requested before the MFMAs of step s: without it every step waits out a full
    // L2 / HBM round trip (602 -> 128 on 233 K rows: 0.190 ms with the loads inside the step)
    f16x8 a[NT], b[MT], a_next[NT], b_next[MT];
    auto fetch = [&](int k0, f16x8 *fa, f16x8 *fb) {
        const int k = k0 + 8 * lq;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = p_base + nt * 16 + l15;
            fa[nt] = load_k8(Wt + (int64_t)n * ldw, k, M, n < P, w_aligned);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = row0 + mt * 16 + l15;
            fb[mt] = load_k8(X + m * ldx, k, M, m < n_rows, x_aligned);
        }
    };
    fetch(0, a_next, b_next);
    for (int k0 = 0; k0 < M; k0 += 32) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = a_next[nt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = b_next[mt];
        if (k0 + 32 < M) fetch(k0 + 32, a_next, b_next);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[nt], b[mt], acc[mt][nt], 0, 0, 0);
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = row0 + mt * 16 + l15;
        if (m >= n_rows) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = p_base + nt * 16 + 4 * lq;       // 4 consecutive columns of H
            f16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = relu_f16((f16)acc[mt][nt][j], relu);
            f16 *dst = H + m * ldh + n;
            if (h_aligned && n + 4 <= ldh) {
                *reinterpret_cast<f16x4 *>(dst) = o;        // columns P..ldh-1 receive exact zeros
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < ldh) dst[j] = o[j];
            }
        }
    }
}

// fp32: each lane loads 4 consecutive K of its row; MFMA step j consumes element j of both
// operands, so the four steps together cover k0..k0+15 (any consistent K order sums the same set).
template <int NT, int MT>
__global__ __launch_bounds__(kBlock) void xw_dense_f32_kernel(
    int n_rows, int M, int P, int p_base, const float *__restrict__ X, int64_t ldx, const float *__restrict__ Wt,
    int64_t ldw, float *__restrict__ H, int64_t ldh, int x_aligned, int w_aligned, int h_aligned, sgx_epilogue ep, int relu)
{
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t row0 = wave * (MT * 16);
    if (row0 >= n_rows) return;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};

    f32x4 a[NT], b[MT], a_next[NT], b_next[MT];
    auto fetch = [&](int k0, f32x4 *fa, f32x4 *fb) {
        const int k = k0 + 4 * lq;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = p_base + nt * 16 + l15;
            fa[nt] = load_k4(Wt + (int64_t)n * ldw, k, M, n < P, w_aligned);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = row0 + mt * 16 + l15;
            fb[mt] = load_k4(X + m * ldx, k, M, m < n_rows, x_aligned);
        }
    };
    fetch(0, a_next, b_next);
    for (int k0 = 0; k0 < M; k0 += 16) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = a_next[nt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = b_next[mt];
        if (k0 + 16 < M) fetch(k0 + 16, a_next, b_next);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt][j], b[mt][j], acc[mt][nt], 0, 0, 0);
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = row0 + mt * 16 + l15;
        if (m >= n_rows) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = p_base + nt * 16 + 4 * lq;
            float *dst = H + m * ldh + n;
            if (ep.rq_ten_pow != 0.0f) {                       // quantised layer: H is re-quantised as it is produced
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][nt][j] = sgx_requant_value(acc[mt][nt][j], ep);
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][nt][j] = acc[mt][nt][j] > 0.0f ? acc[mt][nt][j] : 0.0f;
            }
            if (h_aligned && n + 4 <= ldh) {
                *reinterpret_cast<f32x4 *>(dst) = acc[mt][nt];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < ldh) dst[j] = acc[mt][nt][j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Weights-stationary form (fp16, M_fea <= 128): the reference keeps its W tile on chip for the
// whole pass over X (B_accel, K.cpp:3038-3051); here a wavefront keeps the W fragments of its
// NTW column tiles for ALL of K in registers (KS k-steps x NTW tiles x 4 VGPRs <= 80) and walks
// row tiles of X persistently, so W is fetched once per wavefront and the loop only streams X in
// and H out -- no LDS, no barriers, next tile's X fragments requested before this tile's MFMAs.
// G = ceil(P/16 / NTW) column groups; consecutive wavefronts (one workgroup) take the G groups of
// the same row tile, so the G-fold re-read of X hits L1/L2.
// ---------------------------------------------------------------------------------------
// SC (GAT layer, heads of 32 columns, NTW even): beside H the kernel forms the attention scores s1 = H.a1, s2 = H.a2 per
// (row, head) from the ROUNDED values it stores -- a pair of tiles hands a lane quad the 32 columns of one head, 8 per
// lane: the same eight-term fma chain per lane and the same two-step tree over the four lanes as gat_scores_rows_kernel
// (gat.hip), hence the same bits -- which saves that kernel's pass over H (87 MB on the ogbn-arxiv shape: 35 us).
template <int KS, int NTW, int MT, int SC = 0>
__global__ __launch_bounds__(kBlock) void xw_dense_stationary_f16_kernel(
    int n_rows, int M, int P, int col_groups, const f16 *__restrict__ X, int64_t ldx, const f16 *__restrict__ Wt,
    int64_t ldw, f16 *__restrict__ H, int64_t ldh, int h_aligned, int relu, const f16 *__restrict__ att = nullptr,
    float *__restrict__ s1 = nullptr, float *__restrict__ s2 = nullptr, int n_heads = 0, int f_head = 32)
{
    // SC == 2: heads of whole 64-column groups -- the wavefront's 64 columns give one PARTIAL per row (its two quads' sums
    // added: the next level of the same tree), s1 / s2 are [n_rows x P / 64] and gat_scores_combine_kernel adds the groups
    // of a head in the tree's order.
    static_assert(!SC || NTW == 4, "a wavefront's columns = one 64-column group = two pairs of tiles");
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t total_waves = (int64_t)gridDim.x * (kBlock / 64);
    const int cg = (int)(gw % col_groups);
    const int64_t stream = gw / col_groups, n_streams = total_waves / col_groups;
    if (stream >= n_streams) return;                            // leftover wavefronts of the last workgroup
    const int n_base = cg * NTW * 16;

    // Which output column MFMA row i of column tile nt stands for.  With an even number of tiles the two
    // tiles of a pair interleave in groups of 4 -- tile 2q takes columns 8g..8g+3 and tile 2q+1 columns
    // 8g+4..8g+7 of the pair's 32 -- so that the 4 + 4 results a lane ends up with are 8 consecutive
    // columns of one row of H: one 16-byte store instead of two 8-byte ones.
    constexpr bool PAIRED = (NTW % 2 == 0);
    auto column_of = [&](int nt, int i) {
        return PAIRED ? n_base + (nt / 2) * 32 + 8 * (i / 4) + 4 * (nt % 2) + (i % 4) : n_base + nt * 16 + i;
    };
    f16x8 a[KS][NTW];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = column_of(nt, l15);
            a[ks][nt] = load_k8(Wt + (int64_t)n * ldw, ks * 32 + 8 * lq, M, n < P, true);
        }
    // SC: this lane's 8 columns of each of its heads (pair q of tiles = head n_base / 32 + q): the attention fragments
    float att1[SC ? NTW / 2 : 1][8], att2[SC ? NTW / 2 : 1][8];
    if constexpr (SC != 0) {
#pragma unroll
        for (int q = 0; q < NTW / 2; ++q) {
            const int c0 = n_base + 32 * q + 8 * lq;                 // this lane's first column of the pair
            const int head = c0 / f_head, j0 = c0 - head * f_head;
            const f16x8 v1 = load_k8(att + (int64_t)head * 2 * f_head, j0, f_head, head < n_heads, true);             // (one 16-byte load)
            const f16x8 v2 = load_k8(att + (int64_t)head * 2 * f_head + f_head, j0, f_head, head < n_heads, true);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                att1[q][j] = (float)v1[j];
                att2[q][j] = (float)v2[j];
            }
        }
    }

    const int64_t n_tiles = (n_rows + MT * 16 - 1) / (MT * 16);
    f16x8 b[KS][MT], b_next[KS][MT];
    int64_t tile = stream;
    if (tile < n_tiles) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int64_t m = tile * (MT * 16) + mt * 16 + l15;
                b_next[ks][mt] = load_k8(X + m * ldx, ks * 32 + 8 * lq, M, m < n_rows, true);
            }
    }
    for (; tile < n_tiles; tile += n_streams) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[ks][mt] = b_next[ks][mt];
        const int64_t next = tile + n_streams;
        if (next < n_tiles) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int64_t m = next * (MT * 16) + mt * 16 + l15;
                    b_next[ks][mt] = load_k8(X + m * ldx, ks * 32 + 8 * lq, M, m < n_rows, true);
                }
        }
        f32x4 acc[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][nt], b[ks][mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = tile * (MT * 16) + mt * 16 + l15;
            if constexpr (SC != 0) {
                // (before the row check: the tree runs over all four lanes of a row)
                float g1 = 0.0f, g2 = 0.0f;
#pragma unroll
                for (int q = 0; q < NTW / 2; ++q) {
                    float p1 = 0.0f, p2 = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = (float)relu_f16((f16)acc[mt][2 * q + j / 4][j % 4], relu);
                        p1 = __builtin_fmaf(x, att1[q][j], p1);
                        p2 = __builtin_fmaf(x, att2[q][j], p2);
                    }
                    p1 += __shfl_xor(p1, 16);
                    p2 += __shfl_xor(p2, 16);
                    p1 += __shfl_xor(p1, 32);
                    p2 += __shfl_xor(p2, 32);
                    if constexpr (SC == 1) {
                        const int head = n_base / 32 + q;
                        if (lq == 0 && m < n_rows && head < n_heads) {
                            s1[m * n_heads + head] = p1;
                            s2[m * n_heads + head] = p2;
                        }
                    } else {
                        g1 = q == 0 ? p1 : g1 + p1;
                        g2 = q == 0 ? p2 : g2 + p2;
                    }
                }
                if constexpr (SC == 2) {
                    const int n_groups = P / 64;
                    if (lq == 0 && m < n_rows && cg < n_groups) {
                        s1[m * n_groups + cg] = g1;
                        s2[m * n_groups + cg] = g2;
                    }
                }
            }
            if (m >= n_rows) continue;
            if constexpr (PAIRED) {
#pragma unroll
                for (int q = 0; q < NTW / 2; ++q) {
                    const int n = n_base + q * 32 + 8 * lq;          // column_of(2q, 4 lq) .. column_of(2q+1, 4 lq + 3)
                    f16x8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = relu_f16((f16)acc[mt][2 * q][j], relu);
                        o[4 + j] = relu_f16((f16)acc[mt][2 * q + 1][j], relu);
                    }
                    f16 *dst = H + m * ldh + n;
                    if (h_aligned && n + 8 <= ldh && (ldh % 8 == 0) && ((uintptr_t)H % 16 == 0)) {
                        *reinterpret_cast<f16x8 *>(dst) = o;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (n + j < ldh) dst[j] = o[j];
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int n = n_base + nt * 16 + 4 * lq;
                    f16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = relu_f16((f16)acc[mt][nt][j], relu);
                    f16 *dst = H + m * ldh + n;
                    if (h_aligned && n + 4 <= ldh) {
                        *reinterpret_cast<f16x4 *>(dst) = o;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (n + j < ldh) dst[j] = o[j];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same weights-stationary form in fp32 (the SGRACE library's own setting runs float32 buffers, SG.py:1645-1848):
// a wavefront keeps the fragments of NTW column tiles for all of K <= 128 in registers (KB blocks of 16 k x NTW tiles x
// 4 VGPRs <= 128), walks 16-row tiles of X persistently with the next tile's fragments requested before this tile's
// MFMAs, and sums in the order of xw_dense_f32_kernel (k blocks ascending, element j of a lane's four ascending), so the
// two kernels give the same bits.  The tile kernel re-reads its W fragments from L1 / L2 every k-step:
// 128 -> 256 on 169 K rows 0.184 ms against an MFMA floor of 0.07 ms (v_mfma_f32_16x16x4_f32 is the fp32 peak's 157 TF/s).
// ---------------------------------------------------------------------------------------
template <int KB, int NTW>
__global__ __launch_bounds__(kBlock) void xw_dense_stationary_f32_kernel(
    int n_rows, int M, int P, int col_groups, const float *__restrict__ X, int64_t ldx, const float *__restrict__ Wt,
    int64_t ldw, float *__restrict__ H, int64_t ldh, int x_aligned, int w_aligned, int h_aligned, sgx_epilogue ep, int relu)
{
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t total_waves = (int64_t)gridDim.x * (kBlock / 64);
    const int cg = (int)(gw % col_groups);
    const int64_t stream = gw / col_groups, n_streams = total_waves / col_groups;
    if (stream >= n_streams) return;                            // leftover wavefronts of the last workgroup
    const int n_base = cg * NTW * 16;

    f32x4 a[KB][NTW];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = n_base + nt * 16 + l15;
            a[kb][nt] = load_k4(Wt + (int64_t)n * ldw, kb * 16 + 4 * lq, M, n < P, w_aligned);
        }

    const int64_t n_tiles = ((int64_t)n_rows + 15) / 16;
    f32x4 b[KB], b_next[KB];
    int64_t tile = stream;
    if (tile < n_tiles) {
        const int64_t m = tile * 16 + l15;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b_next[kb] = load_k4(X + m * ldx, kb * 16 + 4 * lq, M, m < n_rows, x_aligned);
    }
    for (; tile < n_tiles; tile += n_streams) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b[kb] = b_next[kb];
        const int64_t next = tile + n_streams;
        if (next < n_tiles) {
            const int64_t m = next * 16 + l15;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) b_next[kb] = load_k4(X + m * ldx, kb * 16 + 4 * lq, M, m < n_rows, x_aligned);
        }
        f32x4 acc[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb][nt][j], b[kb][j], acc[nt], 0, 0, 0);
        const int64_t m = tile * 16 + l15;
        if (m >= n_rows) continue;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int n = n_base + nt * 16 + 4 * lq;
            float *dst = H + m * ldh + n;
            if (ep.rq_ten_pow != 0.0f) {                       // quantised layer: H is re-quantised as it is produced
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][j] = sgx_requant_value(acc[nt][j], ep);
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][j] = acc[nt][j] > 0.0f ? acc[nt][j] : 0.0f;
            }
            if (h_aligned && n + 4 <= ldh) {
                *reinterpret_cast<f32x4 *>(dst) = acc[nt];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < ldh) dst[j] = acc[nt][j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Long K (fp16, M_fea > 128) on many rows: a workgroup computes 128 rows x 128 columns, its four wavefronts 64 x 64
// each.  The k-steps' operand tiles (128 x 32 halves of X and of Wt) go through LDS, double buffered: the global
// loads of step s+1 are in flight (16 bytes per lane, whole 64-byte row pieces per 4 lanes) while the MFMAs of step s
// read their fragments from LDS, so W is fetched once per workgroup instead of once per wavefront and k-step, and
// three workgroups per CU keep ~48 KB of X in flight.  Rows of 40 halves in LDS: the 16-byte fragment reads of 8
// consecutive rows fall into 8 disjoint bank groups.
// ---------------------------------------------------------------------------------------
constexpr int kLdsBM = 128, kLdsBN = 128, kLdsBK = 32, kLdsPitch = kLdsBK + 8;

__global__ __launch_bounds__(kBlock) void xw_dense_lds_f16_kernel(
    int n_rows, int M, int P, const f16 *__restrict__ X, int64_t ldx, const f16 *__restrict__ Wt, int64_t ldw,
    f16 *__restrict__ H, int64_t ldh, int h_aligned, int relu)
{
    __shared__ __attribute__((aligned(16))) f16 sX[2][kLdsBM * kLdsPitch];
    __shared__ __attribute__((aligned(16))) f16 sW[2][kLdsBN * kLdsPitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t row_base = (int64_t)blockIdx.x * kLdsBM;
    const int col_base = blockIdx.y * kLdsBN;

    // staging: a tile is 128 rows x 4 chunks of 16 bytes = 512 chunks, two per thread
    f16x8 rx[2], rw[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + kBlock * i, r = c >> 2, k = k0 + 8 * (c & 3);
            rx[i] = load_k8(X + (row_base + r) * ldx, k, M, row_base + r < n_rows, true);
            rw[i] = load_k8(Wt + (int64_t)(col_base + r) * ldw, k, M, col_base + r < P, true);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + kBlock * i, r = c >> 2, part = c & 3;
            *reinterpret_cast<f16x8 *>(&sX[buf][r * kLdsPitch + 8 * part]) = rx[i];
            *reinterpret_cast<f16x8 *>(&sW[buf][r * kLdsPitch + 8 * part]) = rw[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};

    // (two k-steps in flight in registers, the loop unrolled by two, was measured and lost: 206 VGPRs leave two
    // workgroups per CU, 0.153 ms against 0.142 ms on 602 -> 128)
    const int n_steps = (M + kLdsBK - 1) / kLdsBK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int s = 0; s < n_steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_steps) gload((s + 1) * kLdsBK);
        f16x8 a[4], b[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            a[nt] = *reinterpret_cast<const f16x8 *>(&sW[buf][(64 * wc + nt * 16 + l15) * kLdsPitch + 8 * lq]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            b[mt] = *reinterpret_cast<const f16x8 *>(&sX[buf][(64 * wr + mt * 16 + l15) * kLdsPitch + 8 * lq]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[nt], b[mt], acc[mt][nt], 0, 0, 0);
        if (s + 1 < n_steps) sstore(buf ^ 1);       // the other buffer was last read before the previous barrier
        __syncthreads();
    }

#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t m = row_base + 64 * wr + mt * 16 + l15;
        if (m >= n_rows) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = col_base + 64 * wc + nt * 16 + 4 * lq;
            f16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = relu_f16((f16)acc[mt][nt][j], relu);
            f16 *dst = H + m * ldh + n;
            if (h_aligned && n + 4 <= ldh) {
                *reinterpret_cast<f16x4 *>(dst) = o;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < ldh) dst[j] = o[j];
            }
        }
    }
}

template <int KS, int NTW>
int launch_stationary(int n_rows, int M, int P, int nt_total, const void *X, int64_t ldx, const void *Wt, int64_t ldw,
                      void *H, int64_t ldh, int ha, hipStream_t s, int relu)
{
    constexpr int MT = 2;
    const int groups = (nt_total + NTW - 1) / NTW;
    const int64_t n_tiles = ((int64_t)n_rows + MT * 16 - 1) / (MT * 16);
    // persistent grid: 8 workgroups per CU, trimmed to the work there is, whole column-group sets only
    int64_t waves = (int64_t)256 * 8 * (kBlock / 64);
    if (waves > n_tiles * groups) waves = n_tiles * groups;
    waves = (waves + groups - 1) / groups * groups;
    const unsigned grid = (unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64));
    hipLaunchKernelGGL((xw_dense_stationary_f16_kernel<KS, NTW, MT>), dim3(grid), dim3(kBlock), 0, s, n_rows, M, P, groups,
                       (const f16 *)X, ldx, (const f16 *)Wt, ldw, (f16 *)H, ldh, ha, relu);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

// picks (KS, NTW) for the stationary kernel; SGX_ERR_UNSUPPORTED = use the tiled kernel
int try_stationary(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H,
                   int64_t ldh, int ha, hipStream_t s, int relu)
{
    // measured (tools/bench_configs.py): 100 -> 256 on 2.4 M rows 1.31 -> 0.66 ms; with K > 128 the W
    // fragments leave room for one column tile only and the re-reads of X cost more than they save
    // (602 -> 128: 0.23 -> 0.38 ms), so longer K stays on the tiled kernel
    if (M > 128 || n_rows < 8192) return SGX_ERR_UNSUPPORTED;
    const int nt_total = (int)((ldh + 15) / 16);               // pad columns P..ldh-1 are produced (as zeros) too
    const int ks = (M + 31) / 32;
#define SGX_ST(KS_, NTW_) return launch_stationary<KS_, NTW_>(n_rows, M, P, nt_total, X, ldx, Wt, ldw, H, ldh, ha, s, relu)
    if (ks <= 2) { if (nt_total >= 8) SGX_ST(2, 8); if (nt_total >= 4) SGX_ST(2, 4); if (nt_total >= 2) SGX_ST(2, 2); SGX_ST(2, 1); }
    if (ks <= 4) { if (nt_total >= 4) SGX_ST(4, 4); if (nt_total >= 2) SGX_ST(4, 2); SGX_ST(4, 1); }
    SGX_ST(4, 1);
#undef SGX_ST
}

template <int KB, int NTW>
int launch_stationary_f32(int n_rows, int M, int P, int nt_total, const void *X, int64_t ldx, const void *Wt, int64_t ldw,
                          void *H, int64_t ldh, int xa, int wa, int ha, hipStream_t s, sgx_epilogue ep, int relu)
{
    const int groups = (nt_total + NTW - 1) / NTW;
    const int64_t n_tiles = ((int64_t)n_rows + 15) / 16;
    // persistent grid: 2 workgroups per CU (some 200 VGPRs per lane), trimmed to the work there is, whole column-group sets only
    int64_t waves = (int64_t)256 * 2 * (kBlock / 64);
    if (waves > n_tiles * groups) waves = n_tiles * groups;
    waves = (waves + groups - 1) / groups * groups;
    const unsigned grid = (unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64));
    hipLaunchKernelGGL((xw_dense_stationary_f32_kernel<KB, NTW>), dim3(grid), dim3(kBlock), 0, s, n_rows, M, P, groups,
                       (const float *)X, ldx, (const float *)Wt, ldw, (float *)H, ldh, xa, wa, ha, ep, relu);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

// picks (KB, NTW) for the fp32 stationary kernel; SGX_ERR_UNSUPPORTED = use the tile kernel
int try_stationary_f32(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                       int xa, int wa, int ha, hipStream_t s, sgx_epilogue ep, int relu)
{
    if (M > 128 || n_rows < 8192 || sgx_tune().xw_no_stationary_f32) return SGX_ERR_UNSUPPORTED;      // (tuning / test override)
    // the workgroup's 4 wavefronts take the column groups of the same row tile when there are 4 of them (X from L1 then)
    const int nt_total = (int)((ldh + 15) / 16);               // pad columns P..ldh-1 are produced (as zeros) too
    const int kb = (M + 15) / 16;
#define SGX_ST32(KB_, NTW_) return launch_stationary_f32<KB_, NTW_>(n_rows, M, P, nt_total, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu)
    if (kb <= 2) { if (nt_total >= 4) SGX_ST32(2, 4); if (nt_total >= 2) SGX_ST32(2, 2); SGX_ST32(2, 1); }
    if (kb <= 4) { if (nt_total >= 4) SGX_ST32(4, 4); if (nt_total >= 2) SGX_ST32(4, 2); SGX_ST32(4, 1); }
    if (nt_total >= 4) SGX_ST32(8, 4);
    if (nt_total >= 2) SGX_ST32(8, 2);
    SGX_ST32(8, 1);
#undef SGX_ST32
}

template <int KS, int SC>
int launch_stationary_scores(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                             int ha, const void *att, int n_heads, float *s1, float *s2, hipStream_t s)
{
    constexpr int NTW = 4, MT = 2;
    const int nt_total = (int)((ldh + 15) / 16);
    const int groups = (nt_total + NTW - 1) / NTW;
    const int64_t n_tiles = ((int64_t)n_rows + MT * 16 - 1) / (MT * 16);
    int64_t waves = (int64_t)256 * 8 * (kBlock / 64);
    if (waves > n_tiles * groups) waves = n_tiles * groups;
    waves = (waves + groups - 1) / groups * groups;
    const unsigned grid = (unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64));
    hipLaunchKernelGGL((xw_dense_stationary_f16_kernel<KS, NTW, MT, SC>), dim3(grid), dim3(kBlock), 0, s, n_rows, M, P, groups,
                       (const f16 *)X, ldx, (const f16 *)Wt, ldw, (f16 *)H, ldh, ha, 0, (const f16 *)att, s1, s2, n_heads, P / n_heads);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

template <int NT, int MT>
int launch_tile(int dtype, int n_rows, int M, int P, int p_base, const void *X, int64_t ldx, const void *Wt,
                int64_t ldw, void *H, int64_t ldh, int xa, int wa, int ha, hipStream_t s, sgx_epilogue ep, int relu)
{
    const int64_t rows_per_block = (int64_t)MT * 16 * (kBlock / 64);
    const unsigned grid = (unsigned)((n_rows + rows_per_block - 1) / rows_per_block);
    if (dtype == SGX_F16)
        hipLaunchKernelGGL((xw_dense_f16_kernel<NT, MT>), dim3(grid), dim3(kBlock), 0, s, n_rows, M, P, p_base,
                           (const f16 *)X, ldx, (const f16 *)Wt, ldw, (f16 *)H, ldh, xa, wa, ha, relu);
    else
        hipLaunchKernelGGL((xw_dense_f32_kernel<NT, MT>), dim3(grid), dim3(kBlock), 0, s, n_rows, M, P, p_base,
                           (const float *)X, ldx, (const float *)Wt, ldw, (float *)H, ldh, xa, wa, ha, ep, relu);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

}  // namespace

extern "C" int sgx_xw_dense(int dtype, int acc_mode, int spmm_block, int n_rows, int M_fea, int P, const void *X, int64_t ldx,
                            const void *Wt, int64_t ldw, void *H, int64_t ldh, void *stream)
{
    return sgx_xw_dense_ep(dtype, acc_mode, spmm_block, n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, (hipStream_t)stream,
                           sgx_no_epilogue());
}

extern "C" int sgx_xw_dense_act(int dtype, int relu, int n_rows, int M_fea, int P, const void *X, int64_t ldx,
                                const void *Wt, int64_t ldw, void *H, int64_t ldh, void *stream)
{
    return sgx_xw_dense_ep(dtype, SGX_ACC_F32, 1, n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, (hipStream_t)stream,
                           sgx_no_epilogue(), relu ? 1 : 0);
}

int sgx_xw_dense_ep(int dtype, int acc_mode, int spmm_block, int n_rows, int M_fea, int P, const void *X, int64_t ldx,
                    const void *Wt, int64_t ldw, void *H, int64_t ldh, hipStream_t stream, sgx_epilogue ep, int relu)
{
    if (ep.rq_ten_pow != 0.0f && dtype != SGX_F32) return SGX_ERR_UNSUPPORTED;      // the quantised layer is fp32
    if (n_rows < 0 || M_fea < 1 || P < 1 || ldx < M_fea || ldw < M_fea || ldh < P) return SGX_ERR_SHAPE;
    if (n_rows == 0) return SGX_OK;
    if (!X || !Wt || !H) return SGX_ERR_NULL;
    if (dtype != SGX_F16 && dtype != SGX_F32) return SGX_ERR_UNSUPPORTED;
    hipStream_t s = stream;
    if (relu && acc_mode != SGX_ACC_F32) return SGX_ERR_UNSUPPORTED;                  // ReLU on the stores: fp32-accumulate kernels only
    if (acc_mode == SGX_ACC_REF_HALF) {
        // the reference's sequential half arithmetic (refhalf.hip)
        if (dtype != SGX_F16) return SGX_ERR_UNSUPPORTED;
        if (ldh > P) SGX_HIP_CHECK(hipMemsetAsync(H, 0, (size_t)n_rows * ldh * sizeof(f16), s));
        return sgx_refhalf_dense(spmm_block, 1, n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, s);
    }
    if (acc_mode != SGX_ACC_F32) return SGX_ERR_UNSUPPORTED;
    const size_t es = sgx_elem_size(dtype);
    const int xa = ((uintptr_t)X % 16 == 0) && ((ldx * es) % 16 == 0);
    const int wa = ((uintptr_t)Wt % 16 == 0) && ((ldw * es) % 16 == 0);
    const int ha = ((uintptr_t)H % (4 * es) == 0) && ((ldh * es) % (4 * es) == 0);
    // (the LDS kernel below loses to this one on short K: 100 -> 256 on 2.4 M rows 0.78 vs 0.52 ms, 64 -> 64 0.29 vs 0.22 ms)
    if (dtype == SGX_F16) {
        const int rc = try_stationary(n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, ha, s, relu);
        if (rc != SGX_ERR_UNSUPPORTED) return rc;
    } else {
        int rc = sgx_xw_dense_wlds_f32(n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, ha, ep, relu, s);
        if (rc != SGX_ERR_UNSUPPORTED) return rc;
        rc = try_stationary_f32(n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu);
        if (rc != SGX_ERR_UNSUPPORTED) return rc;
    }
    const bool short_tiles = sgx_tune().xw_short_tiles, no_lds = sgx_tune().xw_no_lds;      // tuning overrides
    const bool tall = n_rows >= 32768 && !short_tiles;
    if (dtype == SGX_F16 && tall && !no_lds && M_fea > 128) {
        // all of W^T in LDS, X streamed through a register ring (xw_dense_wlds.hip): 602 -> 128 on 233 K rows
        const int rc = sgx_xw_dense_wlds(n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, relu, s);
        if (rc != SGX_ERR_UNSUPPORTED) return rc;
        const dim3 grid((unsigned)((n_rows + kLdsBM - 1) / kLdsBM), (unsigned)((ldh + kLdsBN - 1) / kLdsBN));
        hipLaunchKernelGGL(xw_dense_lds_f16_kernel, grid, dim3(kBlock), 0, s, n_rows, M_fea, P, (const f16 *)X, ldx,
                           (const f16 *)Wt, ldw, (f16 *)H, ldh, ha, relu);
        SGX_LAUNCH_CHECK();
        return SGX_OK;
    }
    // columns are produced in blocks of up to 256 (16 tiles); the pad columns P..ldh-1 belong to the last block
    for (int p_base = 0; p_base < ldh; p_base += 256) {
        const int cols = (int)((ldh - p_base) < 256 ? (ldh - p_base) : 256);
        const int nt = (cols + 15) / 16;
        int rc;
        if (nt <= 1)       rc = launch_tile<1, 4>(dtype, n_rows, M_fea, P, p_base, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu);
        else if (nt <= 2)  rc = launch_tile<2, 4>(dtype, n_rows, M_fea, P, p_base, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu);
        else if (nt <= 4)  rc = launch_tile<4, 4>(dtype, n_rows, M_fea, P, p_base, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu);
        // every wavefront re-reads its W fragments from L1/L2 each k-step, so with rows to spare a taller tile halves
        // those reads per flop (602 -> 128 on 233 K rows: 0.234 -> 0.190 ms fp16, 0.62 -> 0.47 ms fp32)
        else if (nt <= 8)  rc = tall ? launch_tile<8, 4>(dtype, n_rows, M_fea, P, p_base, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu)
                                     : launch_tile<8, 2>(dtype, n_rows, M_fea, P, p_base, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu);
        else               rc = tall ? launch_tile<16, 2>(dtype, n_rows, M_fea, P, p_base, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu)
                                     : launch_tile<16, 1>(dtype, n_rows, M_fea, P, p_base, X, ldx, Wt, ldw, H, ldh, xa, wa, ha, s, ep, relu);
        if (rc != SGX_OK) return rc;
    }
    return SGX_OK;
}

// X.W of a GAT layer with the attention scores formed beside H (see the SC form of the stationary kernel): fp16, K <= 128,
// heads of 32 columns, P a multiple of 64, 8 K rows and more.
int sgx_xw_dense_scores(int n_rows, int M_fea, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                        const void *attention, int n_heads, float *s1, float *s2, hipStream_t stream)
{
    if (n_heads < 1) n_heads = 1;
    if (n_rows < 8192 || M_fea < 1 || M_fea > 128 || P < 64 || P % 64 != 0 || P % n_heads != 0 || ldh < P || ldx < M_fea || ldw < M_fea)
        return SGX_ERR_UNSUPPORTED;
    const int f_head = P / n_heads;
    if (f_head != 32 && f_head % 64 != 0) return SGX_ERR_UNSUPPORTED;
    if (!X || !Wt || !H || !attention || !s1 || !s2) return SGX_ERR_NULL;
    const int ha = ((uintptr_t)H % 8 == 0) && ((ldh * 2) % 8 == 0);
    const int ks = (M_fea + 31) / 32;
#define SGX_SC(KS_, SC_) return launch_stationary_scores<KS_, SC_>(n_rows, M_fea, P, X, ldx, Wt, ldw, H, ldh, ha, attention, n_heads, s1, s2, stream)
    if (f_head == 32) { if (ks <= 2) SGX_SC(2, 1); SGX_SC(4, 1); }
    if (ks <= 2) SGX_SC(2, 2);
    SGX_SC(4, 2);
#undef SGX_SC
}
