// Edge pass of the GAT layer's backward (FPYNQ_GAT.backward, SG.py:884-1126).  The reference forms
// dense N x N matrices on the CPU:
//     softmax_out = g @ Wh^T ;  dx = S * softmax_out ;  sg = dx - S * rowsum(dx)       (softmax backward)
//     sg = where(adj > 0, sg, 0) ;  sg *= (e > 0 ? 1 : alpha)                          (mask, LeakyReLU backward)
//     grad_attention = [ Wh^T . rowsum(sg) ; Wh^T . colsum(sg) ]
// Only the stored edges matter (S is zero elsewhere), so this is a sampled dense-dense product over
// the CSR pattern followed by a row-local softmax backward:
//     d_e = g[row(e)] . Wh[col(e)]                 one 16-byte gather per lane and edge, dot reduced over the lane group
//     sg_e as above,  g1[r] = sum over the row of sg_e
// The column sums (g2) are row sums over A^T and the two Wh^T products are sgx_xt_g -- existing entry points.
// All fp32, like the reference's backward.
#include "sgx_device.h"

namespace {

template <typename TV, int LPR>
__global__ __launch_bounds__(kBlock) void gat_bwd_edges_kernel(
    int n_rows, int n_feat, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const TV *__restrict__ val, const float *__restrict__ E, const float *__restrict__ S,
    const float *__restrict__ G, int64_t ldg, const float *__restrict__ Wh, unsigned w_bytes, unsigned ldw_bytes,
    float alpha, float *__restrict__ sg, float *__restrict__ g1)
{
    constexpr int VEC = 4;
    constexpr int RPW = 64 / LPR;
    constexpr int TILE = LPR * VEC;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int64_t r = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RPW + grp;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wh), 0, w_bytes, 0x00020000);
    const bool live = r < n_rows;
    int e0 = 0, e1 = 0;
    if (live) { e0 = rowptr[r]; e1 = rowptr[r + 1]; }

    // pass 1: d_e = g_r . Wh[col e], dx_e = S_e d_e (parked in sg), rs = sum of dx over the row
    float rs = 0.0f;
    for (int base = e0; base < e1; base += LPR) {
        const int idx = base + sub;
        const int c = idx < e1 ? col[idx] : 0;
        const int n = e1 - base < LPR ? e1 - base : LPR;
        float mine = 0.0f;
        for (int t = 0; t < n; ++t) {
            const int cc = __shfl(c, t, LPR);
            float part = 0.0f;
            for (int c0 = 0; c0 < n_feat; c0 += TILE) {
                const int col0 = c0 + sub * VEC;
                float w[VEC] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (col0 < n_feat)
                    Fma<float, 4>::run(w, 1.0f, __builtin_amdgcn_raw_buffer_load_b128(
                                                    rsrc, (unsigned)cc * ldw_bytes + (unsigned)col0 * 4u, 0, 0));
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (col0 + i < n_feat) part = __builtin_fmaf(G[r * ldg + col0 + i], w[i], part);
            }
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) part += __shfl_xor(part, off);
            if (t == sub) mine = part;
        }
        if (idx < e1) {
            const float dx = S[idx] * mine;
            sg[idx] = dx;
            rs += dx;
        }
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) rs += __shfl_xor(rs, off);

    // pass 2: softmax backward, mask, LeakyReLU backward; each lane revisits the edges it parked
    float acc = 0.0f;
    for (int idx = e0 + sub; idx < e1; idx += LPR) {
        float v = sg[idx] - S[idx] * rs;
        if (!(Elem<TV>::to_f32(val[idx]) > 0.0f)) v = 0.0f;
        if (!(E[idx] > 0.0f)) v *= alpha;
        sg[idx] = v;
        acc += v;
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) acc += __shfl_xor(acc, off);
    if (live && sub == 0) g1[r] = acc;
}

struct BwdArgs {
    int n_rows, n_feat;
    const int32_t *rowptr, *col;
    const void *val;
    const float *E, *S, *G, *Wh;
    int64_t ldg;
    unsigned w_bytes, ldw_bytes;
    float alpha;
    float *sg, *g1;
    hipStream_t stream;
};

template <typename TV, int LPR>
int bwd_launch(const BwdArgs &a)
{
    const int rows_per_block = (64 / LPR) * (kBlock / 64);
    const unsigned grid = (unsigned)((a.n_rows + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((gat_bwd_edges_kernel<TV, LPR>), dim3(grid), dim3(kBlock), 0, a.stream, a.n_rows, a.n_feat, a.rowptr,
                       a.col, (const TV *)a.val, a.E, a.S, a.G, a.ldg, a.Wh, a.w_bytes, a.ldw_bytes, a.alpha, a.sg, a.g1);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

template <typename TV>
int bwd_dispatch(const BwdArgs &a, int lpr)
{
    switch (lpr) {
    case 1: return bwd_launch<TV, 1>(a);
    case 2: return bwd_launch<TV, 2>(a);
    case 4: return bwd_launch<TV, 4>(a);
    case 8: return bwd_launch<TV, 8>(a);
    case 16: return bwd_launch<TV, 16>(a);
    case 32: return bwd_launch<TV, 32>(a);
    default: return bwd_launch<TV, 64>(a);
    }
}

}  // namespace

extern "C" int sgx_gat_backward_edges(int dtype_values, int n_rows, int n_cols, int n_feat, float alpha,
                                      const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                                      const float *E, const float *S, const float *G, int64_t ldg, const float *Wh,
                                      int64_t ldw, float *sg, float *g1, void *stream)
{
    if (n_rows < 0 || n_cols < 0 || n_feat < 1 || ldg < n_feat || ldw < n_feat) return SGX_ERR_SHAPE;
    if (n_rows == 0) return SGX_OK;
    if (!rowPtr || !columnIndex || !values || !E || !S || !G || !Wh || !sg || !g1) return SGX_ERR_NULL;
    if (dtype_values != SGX_F16 && dtype_values != SGX_F32) return SGX_ERR_UNSUPPORTED;
    if ((uintptr_t)Wh % 16 != 0 || (ldw * 4) % 16 != 0) return SGX_ERR_ALIGN;          // 16-byte gathers of fp32 rows
    const unsigned long long table_bytes = (unsigned long long)n_cols * (unsigned long long)ldw * 4ull;
    if (table_bytes >= 0xFFFFFFF0ull) return SGX_ERR_UNSUPPORTED;
    BwdArgs a;
    a.n_rows = n_rows; a.n_feat = n_feat; a.rowptr = rowPtr; a.col = columnIndex; a.val = values;
    a.E = E; a.S = S; a.G = G; a.Wh = Wh; a.ldg = ldg;
    a.w_bytes = (unsigned)table_bytes; a.ldw_bytes = (unsigned)(ldw * 4);
    a.alpha = alpha; a.sg = sg; a.g1 = g1; a.stream = (hipStream_t)stream;
    int lpr = sgx_next_pow2((n_feat + 3) / 4);
    if (lpr > 64) lpr = 64;
    return dtype_values == SGX_F16 ? bwd_dispatch<f16>(a, lpr) : bwd_dispatch<float>(a, lpr);
}
