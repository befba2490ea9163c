// SGX_ACC_REF_HALF: the arithmetic of the reference's HALF build, bit for bit, on the GPU.
//
// In the HLS kernel every product and every add is rounded to binary16, and the MAC loop keeps
// FADD_LATENCY = 4 partial sums: element k of an sblock (SPMM_BLOCK consecutive rows streamed
// as one sequence) is added into partial k mod 4, and after the block the partials are folded
// ((p0+p1)+p2)+p3 (dsp_kernel_wrapper_fea / _adj_1, K.cpp:2009-2061, :1829-1884; MM.h:137-138).
// Seen from one output element (row r, column j) that is: walk the row's entries in order,
// entry i goes to partial (phase + i) mod 4 with phase = (entries of the block before row r)
// mod 4, fold.  Rows are therefore independent and one thread can own one (row, column) pair;
// this mode exists for exact regression against the reference, not for speed.
//
// Rounding: half x half is exact in fp32 and is rounded once by the conversion; fp32 addition of
// two halves followed by the conversion equals the correctly rounded half sum (24 >= 2*11+2).
// hipcc contracts a*b+c into one fma by default (the compiler first narrows the fp32 round trips to
// half operations, then fuses them: one rounding instead of two) -- contraction is switched off
// for this file.
#include "sgx_internal.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;
constexpr int kLat = 4;            // FTYPE_LATENCY_FEA = FTYPE_LATENCY_ADJ = 4 (MM.h:137-138)

__device__ __forceinline__ f16 hmul(f16 a, f16 b) { return (f16)((float)a * (float)b); }
__device__ __forceinline__ f16 hadd(f16 a, f16 b) { return (f16)((float)a + (float)b); }

// First row of the sblock that holds row r.  The reference gives each of its FEA_THREADS /
// ADJ_THREADS a contiguous row block -- n_rows / threads each, the remainder to the last
// (K.cpp:3159-3164, :3517-3523) -- and the sblock grouping restarts at the block's first row.
__device__ __forceinline__ int sblock_first_row(int r, int n_rows, int spmm_block, int threads)
{
    const int blk = n_rows / threads;
    int t = blk > 0 ? r / blk : threads - 1;
    if (t > threads - 1) t = threads - 1;
    const int first = t * blk;
    return first + (r - first) / spmm_block * spmm_block;
}

__device__ __forceinline__ f16 fold(const f16 *part)
{
    f16 a = part[0];
#pragma unroll
    for (int l = 1; l < kLat; ++l) a = hadd(a, part[l]);      // ACC_PART3, K.cpp:1879-1882
    return a;
}

// out[r][j] = sum over the CSR row r of  val[e] * table[col[e]][j]   (sparse X.W and A.H)
__global__ __launch_bounds__(kBlock) void refhalf_csr_kernel(
    int n_rows, int n_feat, int spmm_block, int threads, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const f16 *__restrict__ val, const f16 *__restrict__ table, int64_t ldt,
    f16 *__restrict__ out, int64_t ldo, int relu)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (int64_t)n_rows * n_feat) return;
    const int r = (int)(gid / n_feat), j = (int)(gid % n_feat);
    const int block_first = sblock_first_row(r, n_rows, spmm_block, threads);
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    const int phase = (e0 - rowptr[block_first]) % kLat;
    f16 part[kLat] = {(f16)0, (f16)0, (f16)0, (f16)0};
    for (int e = e0; e < e1; ++e) {
        const f16 p = hmul(val[e], table[(int64_t)col[e] * ldt + j]);
        const int l = (phase + (e - e0)) % kLat;
#pragma unroll
        for (int q = 0; q < kLat; ++q)                         // static indices keep `part` in registers
            if (q == l) part[q] = hadd(part[q], p);
    }
    f16 v = fold(part);
    if (relu && !(v > (f16)0)) v = (f16)0;                     // K.cpp:2586-2590
    out[(int64_t)r * ldo + j] = v;
}

// out[r][j] = sum_k X[r][k] * Wt[j][k]   with the dense stream's lane rule (K.cpp:849-863, :985-1012):
// every row contributes M entries (zeros included), column = position in the row
__global__ __launch_bounds__(kBlock) void refhalf_dense_kernel(
    int n_rows, int M, int n_feat, int spmm_block, int threads, const f16 *__restrict__ X, int64_t ldx,
    const f16 *__restrict__ Wt, int64_t ldw, f16 *__restrict__ out, int64_t ldo)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (int64_t)n_rows * n_feat) return;
    const int r = (int)(gid / n_feat), j = (int)(gid % n_feat);
    const int phase = (int)(((int64_t)(r - sblock_first_row(r, n_rows, spmm_block, threads)) * M) % kLat);
    f16 part[kLat] = {(f16)0, (f16)0, (f16)0, (f16)0};
    const f16 *x = X + (int64_t)r * ldx;
    const f16 *w = Wt + (int64_t)j * ldw;
    for (int k = 0; k < M; ++k) {
        const f16 p = hmul(x[k], w[k]);
        const int l = (phase + k) % kLat;
#pragma unroll
        for (int q = 0; q < kLat; ++q)
            if (q == l) part[q] = hadd(part[q], p);
    }
    out[(int64_t)r * ldo + j] = fold(part);
}

}  // namespace

int sgx_refhalf_csr(int spmm_block, int threads, int relu, int n_rows, int n_feat, const int32_t *rowPtr, const int32_t *columnIndex,
                    const void *values, const void *table, int64_t ldt, void *out, int64_t ldo, hipStream_t s)
{
    if (n_rows == 0) return SGX_OK;
    if (spmm_block < 1) spmm_block = 1;
    if (threads < 1) threads = 1;
    const int64_t total = (int64_t)n_rows * n_feat;
    hipLaunchKernelGGL(refhalf_csr_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows,
                       n_feat, spmm_block, threads, rowPtr, columnIndex, (const f16 *)values, (const f16 *)table, ldt,
                       (f16 *)out, ldo, relu);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

int sgx_refhalf_dense(int spmm_block, int threads, int n_rows, int M, int n_feat, const void *X, int64_t ldx, const void *Wt,
                      int64_t ldw, void *out, int64_t ldo, hipStream_t s)
{
    if (n_rows == 0) return SGX_OK;
    if (spmm_block < 1) spmm_block = 1;
    if (threads < 1) threads = 1;
    const int64_t total = (int64_t)n_rows * n_feat;
    hipLaunchKernelGGL(refhalf_dense_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows,
                       M, n_feat, spmm_block, threads, (const f16 *)X, ldx, (const f16 *)Wt, ldw, (f16 *)out, ldo);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}
