// Edge pass of the GAT layer's backward (FPYNQ_GAT.backward, SG.py:884-1126).  The reference forms
// dense N x N matrices on the CPU:
//     softmax_out = g @ Wh^T ;  dx = S * softmax_out ;  sg = dx - S * rowsum(dx)       (softmax backward)
//     sg = where(adj > 0, sg, 0) ;  sg *= (e > 0 ? 1 : alpha)                          (mask, LeakyReLU backward)
//     grad_attention = [ Wh^T . rowsum(sg) ; Wh^T . colsum(sg) ]
// Only the stored edges matter (S is zero elsewhere), so this is a sampled dense-dense product over
// the CSR pattern followed by a row-local softmax backward:
//     d_e = g[row(e)] . Wh[col(e)]                 16-byte gathers of both rows, the dot reduced over a lane group
//     sg_e as above,  g1[r] = sum over the row of sg_e
// The column sums (g2) are row sums over A^T and the two Wh^T products are sgx_xt_g -- existing entry points.
// All fp32, like the reference's backward.
//
// Two launches (round 2; one row-per-wavefront kernel before: a row's edges one after the other, each a gather and a
// 6-step shuffle reduction with nothing else in flight -- 0.79 ms on the ogbn-arxiv shape and 10 ms on its R-MAT
// stand-in, whose hub rows were walked by one wavefront each):
//   dots   EDGE-parallel and therefore balanced whatever the degrees: a workgroup takes 256 consecutive stored
//          entries, every thread finds its entry's row by bisection of rowPtr (the probes of neighbouring entries
//          coincide: L1), then each wavefront walks its 64 entries 64 / LPR at a time, LPR lanes per entry, and
//          parks dx_e = S_e d_e in sg.  The entry count comes from rowPtr[n_rows] on the device (persistent grid):
//          the host never reads it.
//   rows   8 lanes per row: the row sum of dx, then sg_e and g1; rows over 256 entries are left to a third launch in
//          which a workgroup of 1024 takes each of them (a hub row of 20 K entries by one wavefront was 0.4 ms).
#include "sgx_device.h"

namespace {

constexpr int kDotBlock = 256;           // stored entries per workgroup pass of the dots kernel
constexpr int kRowLanes = 8;             // lanes per row in the rows kernel
constexpr int kRowLong = 256;            // rows over this many entries: the whole wavefront

// dx_e = S_e (G[row e] . Wh[col e]) for every stored entry, into sg
template <int LPR>
__global__ __launch_bounds__(kDotBlock) void gat_bwd_dots_kernel(
    int n_rows, int n_feat, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ S, const float *__restrict__ G, unsigned g_bytes, unsigned ldg_bytes,
    const float *__restrict__ Wh, unsigned w_bytes, unsigned ldw_bytes, float *__restrict__ sg, int g_vec)
{
    constexpr int GROUPS = 64 / LPR;         // entries a wavefront works on at a time
    constexpr int TILE = LPR * 4;            // columns one pass of a lane group covers
    const int lane = threadIdx.x & 63, sub = lane % LPR, grp = lane / LPR;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wh), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(G), 0, g_bytes, 0x00020000);
    const int64_t nnz = rowptr[n_rows];
    for (int64_t base = (int64_t)blockIdx.x * kDotBlock; base < nnz; base += (int64_t)gridDim.x * kDotBlock) {
        const int64_t e = base + threadIdx.x;                    // lane l of a wavefront owns the l-th of its 64 entries
        const bool have = e < nnz;
        // the row of entry e: the last r with rowptr[r] <= e (rows without entries are stepped over)
        int lo = 0, hi = n_rows;
        while (hi - lo > 1) {
            const int mid = (int)(((int64_t)lo + hi) >> 1);
            if ((int64_t)rowptr[mid] <= e) lo = mid;
            else hi = mid;
        }
        const unsigned g_off = have ? (unsigned)lo * ldg_bytes : kOOB;
        const unsigned w_off = have ? (unsigned)col[have ? e : 0] * ldw_bytes : kOOB;
        const float s_e = have ? S[have ? e : 0] : 0.0f;
        const int64_t wave_base = e - lane;
#pragma unroll 2
        for (int j = 0; j < LPR; ++j) {
            const int q = j * GROUPS + grp;                      // this lane group's entry, by its owner's lane
            const unsigned go = (unsigned)__shfl((int)g_off, q), wo = (unsigned)__shfl((int)w_off, q);
            const float sq = __shfl(s_e, q);
            float part = 0.0f;
            for (int c0 = 0; c0 < n_feat; c0 += TILE) {
                const int col0 = c0 + sub * 4;
                // out of range for lanes past the last column and for slots past the last entry: zeros, no access
                const bool in = col0 < n_feat && go != kOOB;
                const u32x4 wv = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, in ? wo + (unsigned)col0 * 4u : kOOB, 0, 0);
                const unsigned g_at = in ? go + (unsigned)col0 * 4u : kOOB;
                u32x4 gv;
                if (g_vec) {
                    gv = __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, g_at, 0, 0);
                } else {                                 // rows of G that are not 16-byte aligned (kOOB + 12 stays out of range)
#pragma unroll
                    for (int i = 0; i < 4; ++i) gv[i] = __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, g_at + 4u * i, 0, 0);
                }
                // (through a union: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index)
                union { u32x4 v; float f[4]; } ug, uw;
                ug.v = gv;
                uw.v = wv;
#pragma unroll
                for (int i = 0; i < 4; ++i)          // (a row's pad columns hold anything: cut here)
                    if (col0 + i < n_feat) part = __builtin_fmaf(ug.f[i], uw.f[i], part);
            }
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) part += __shfl_xor(part, off);
            if (sub == 0 && wave_base + q < nnz) sg[wave_base + q] = sq * part;
        }
    }
}

// softmax backward, mask, LeakyReLU backward on the parked dx; g1 = row sums of the result
template <typename TV>
__global__ __launch_bounds__(kBlock) void gat_bwd_rows_kernel(
    int n_rows, const int32_t *__restrict__ rowptr, const TV *__restrict__ val, const float *__restrict__ E,
    const float *__restrict__ S, float alpha, float *__restrict__ sg, float *__restrict__ g1)
{
    constexpr int RPW = 64 / kRowLanes;
    const int lane = threadIdx.x & 63, sub = lane % kRowLanes, grp = lane / kRowLanes;
    const int64_t r = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RPW + grp;
    const bool live = r < n_rows;
    int e0 = 0, e1 = 0;
    if (live) { e0 = rowptr[r]; e1 = rowptr[r + 1]; }
    const bool is_long = e1 - e0 > kRowLong;
    auto finish = [&](int idx, float rs) -> float {
        float v = sg[idx] - S[idx] * rs;
        if (!(Elem<TV>::to_f32(val[idx]) > 0.0f)) v = 0.0f;
        if (!(E[idx] > 0.0f)) v *= alpha;
        sg[idx] = v;
        return v;
    };
    if (!is_long) {
        float rs = 0.0f;
        for (int idx = e0 + sub; idx < e1; idx += kRowLanes) rs += sg[idx];
#pragma unroll
        for (int off = 1; off < kRowLanes; off <<= 1) rs += __shfl_xor(rs, off);
        float acc = 0.0f;
        for (int idx = e0 + sub; idx < e1; idx += kRowLanes) acc += finish(idx, rs);
#pragma unroll
        for (int off = 1; off < kRowLanes; off <<= 1) acc += __shfl_xor(acc, off);
        if (live && sub == 0) g1[r] = acc;
    }
}

// rows over kRowLong entries: a workgroup of 1024 looks at 64 consecutive rows and takes the long ones among them one
// after the other with all its threads (most workgroups find none and leave after reading 65 row pointers)
constexpr int kLongThreads = 1024;
template <typename TV>
__global__ __launch_bounds__(kLongThreads) void gat_bwd_long_rows_kernel(
    int n_rows, const int32_t *__restrict__ rowptr, const TV *__restrict__ val, const float *__restrict__ E,
    const float *__restrict__ S, float alpha, float *__restrict__ sg, float *__restrict__ g1)
{
    __shared__ float part[kLongThreads / 64];
    __shared__ float total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r_first = (int64_t)blockIdx.x * 64;
    auto block_sum = [&](float v) -> float {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
        __syncthreads();                              // `total` of the previous sum has been read by everyone
        if (lane == 0) part[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int i = 0; i < kLongThreads / 64; ++i) t += part[i];
            total = t;
        }
        __syncthreads();
        return total;
    };
    // which of the 64 rows are long: one look by the first wavefront
    __shared__ unsigned long long long_mask;
    if (wave == 0) {
        const int64_t r = r_first + lane;
        const int deg = r < n_rows ? rowptr[r + 1] - rowptr[r] : 0;
        const unsigned long long m = __ballot(deg > kRowLong);
        if (lane == 0) long_mask = m;
    }
    __syncthreads();
    unsigned long long todo = long_mask;
    while (todo) {                                    // (uniform over the workgroup)
        const int64_t r = r_first + (__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        const int e0 = rowptr[r], e1 = rowptr[r + 1];
        float rs = 0.0f;
        for (int idx = e0 + (int)threadIdx.x; idx < e1; idx += kLongThreads) rs += sg[idx];
        rs = block_sum(rs);
        float acc = 0.0f;
        for (int idx = e0 + (int)threadIdx.x; idx < e1; idx += kLongThreads) {
            float v = sg[idx] - S[idx] * rs;
            if (!(Elem<TV>::to_f32(val[idx]) > 0.0f)) v = 0.0f;
            if (!(E[idx] > 0.0f)) v *= alpha;
            sg[idx] = v;
            acc += v;
        }
        acc = block_sum(acc);
        if (threadIdx.x == 0) g1[r] = acc;
    }
}

struct BwdArgs {
    int n_rows, n_feat;
    const int32_t *rowptr, *col;
    const float *S, *G, *Wh;
    unsigned g_bytes, ldg_bytes, w_bytes, ldw_bytes;
    float *sg;
    hipStream_t stream;
    unsigned grid;
    int g_vec;
};

template <int LPR>
int dots_launch(const BwdArgs &a)
{
    hipLaunchKernelGGL((gat_bwd_dots_kernel<LPR>), dim3(a.grid), dim3(kDotBlock), 0, a.stream, a.n_rows, a.n_feat, a.rowptr, a.col,
                       a.S, a.G, a.g_bytes, a.ldg_bytes, a.Wh, a.w_bytes, a.ldw_bytes, a.sg, a.g_vec);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
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
    const unsigned long long w_bytes = (unsigned long long)n_cols * (unsigned long long)ldw * 4ull;
    const unsigned long long g_bytes = (unsigned long long)n_rows * (unsigned long long)ldg * 4ull;
    if (w_bytes >= 0xFFFFFFF0ull || g_bytes >= 0xFFFFFFF0ull) return SGX_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    BwdArgs a;
    a.n_rows = n_rows; a.n_feat = n_feat; a.rowptr = rowPtr; a.col = columnIndex;
    a.S = S; a.G = G; a.Wh = Wh;
    a.g_bytes = (unsigned)g_bytes; a.ldg_bytes = (unsigned)(ldg * 4);
    a.w_bytes = (unsigned)w_bytes; a.ldw_bytes = (unsigned)(ldw * 4);
    a.sg = sg; a.stream = s;
    a.g_vec = ((uintptr_t)G % 16 == 0 && (ldg * 4) % 16 == 0) ? 1 : 0;     // 16-byte loads of G's rows too, where they are aligned
    // persistent over the stored entries (their count is read on the device): 8 workgroups per CU
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    a.grid = (unsigned)cus * 8u;
    // lanes per entry: a quarter of the row's 16-byte chunks (four gathers per lane and table), at least 1, at most 64
    int lpr = sgx_next_pow2((n_feat + 15) / 16);
    if (lpr > 64) lpr = 64;
    int rc;
    switch (lpr) {
    case 1: rc = dots_launch<1>(a); break;
    case 2: rc = dots_launch<2>(a); break;
    case 4: rc = dots_launch<4>(a); break;
    case 8: rc = dots_launch<8>(a); break;
    case 16: rc = dots_launch<16>(a); break;
    case 32: rc = dots_launch<32>(a); break;
    default: rc = dots_launch<64>(a); break;
    }
    if (rc != SGX_OK) return rc;
    const int rows_per_block = (64 / kRowLanes) * (kBlock / 64);
    const unsigned grid = (unsigned)((n_rows + rows_per_block - 1) / rows_per_block);
    if (dtype_values == SGX_F16)
        hipLaunchKernelGGL(gat_bwd_rows_kernel<f16>, dim3(grid), dim3(kBlock), 0, s, n_rows, rowPtr, (const f16 *)values, E, S, alpha, sg, g1);
    else
        hipLaunchKernelGGL(gat_bwd_rows_kernel<float>, dim3(grid), dim3(kBlock), 0, s, n_rows, rowPtr, (const float *)values, E, S, alpha, sg, g1);
    SGX_LAUNCH_CHECK();
    const unsigned grid_long = (unsigned)((n_rows + 63) / 64);
    if (dtype_values == SGX_F16)
        hipLaunchKernelGGL(gat_bwd_long_rows_kernel<f16>, dim3(grid_long), dim3(kLongThreads), 0, s, n_rows, rowPtr, (const f16 *)values, E, S, alpha, sg, g1);
    else
        hipLaunchKernelGGL(gat_bwd_long_rows_kernel<float>, dim3(grid_long), dim3(kLongThreads), 0, s, n_rows, rowPtr, (const float *)values, E, S, alpha, sg, g1);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}
