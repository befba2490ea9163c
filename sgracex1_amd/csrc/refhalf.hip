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
#include "sgx_device.h"

#pragma clang fp contract(off)

namespace {

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

// The same arithmetic in the lane-group layout of the fast kernels (rows of at least 25 halves, 16-byte
// aligned table): a group of LPR lanes owns a row, a lane 8 columns of it with its 4 x 8 partial sums in
// packed half registers; the row's entries are broadcast in order and every lane does the 8 rounded
// products and the 8 rounded adds of its columns (v_pk_mul_f16 / v_pk_add_f16: one rounding each, what
// the reference's half type does).  The partial an entry goes to is (phase + i) mod 4; the walk starts
// `phase` positions early with zero entries -- adding +0 changes no partial -- so that the partial index is
// the static position inside an unrolled group of 4.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int LPR>
__global__ __launch_bounds__(kBlock) void refhalf_csr_rows_kernel(
    int n_rows, int n_feat, int spmm_block, int threads, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const f16 *__restrict__ val, const f16 *__restrict__ table, unsigned t_bytes,
    unsigned ld_bytes, f16 *__restrict__ out, int64_t ldo, int relu)
{
    static_assert(LPR % 4 == 0, "pieces must be multiples of the 4 partial sums");
    constexpr int VEC = 8, RPW = 64 / LPR, TILE = LPR * VEC;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int64_t r = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RPW + grp;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16 *>(table), 0, t_bytes, 0x00020000);
    const bool live = r < n_rows;
    int e0 = 0, e1 = 0, phase = 0;
    if (live) {
        e0 = rowptr[r];
        e1 = rowptr[r + 1];
        phase = (e0 - rowptr[sblock_first_row((int)r, n_rows, spmm_block, threads)]) % kLat;
    }
    const int total = live ? phase + (e1 - e0) : 0;              // positions of the walk, the first `phase` empty

    for (int c0 = 0; c0 < n_feat; c0 += TILE) {
        const int col0 = c0 + sub * VEC;
        const unsigned col_off = col0 < n_feat ? (unsigned)col0 * 2u : kOOB;
        h2 part[kLat][4];
#pragma unroll
        for (int l = 0; l < kLat; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[l][j] = (h2){(f16)0, (f16)0};
        for (int kb = 0; kb < total; kb += LPR) {
            const int k = kb + sub;
            unsigned roff = kOOB;
            int vbits = 0;
            if (k >= phase && k < total) {
                const int e = e0 + k - phase;
                roff = (unsigned)col[e] * ld_bytes;
                union { f16 h; unsigned short s; } u;
                u.h = val[e];
                vbits = u.s;
            }
            const int n = total - kb;
#pragma unroll 1
            for (int t0 = 0; t0 < LPR; t0 += kLat) {
                if (t0 >= n) break;
#pragma unroll
                for (int u4 = 0; u4 < kLat; ++u4) {                // position kb + t0 + u4 -> partial u4 (kb, t0 multiples of 4)
                    const unsigned ro = (unsigned)__shfl((int)roff, t0 + u4, LPR);
                    union { unsigned short s; f16 h; } v;
                    v.s = (unsigned short)__shfl(vbits, t0 + u4, LPR);
                    const h2 vv = (h2){v.h, v.h};
                    union { u32x4 q; h2 h[4]; } raw;
                    raw.q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (ro != kOOB && col_off != kOOB) ? ro + col_off : kOOB,
                                                                  0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) part[u4][j] = part[u4][j] + vv * raw.h[j];   // two roundings, contraction is off
                }
            }
        }
        if (live && col0 < n_feat) {
            union { u32x4 q; h2 h[4]; f16 e[8]; } o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o.h[j] = ((part[0][j] + part[1][j]) + part[2][j]) + part[3][j];   // K.cpp:1879-1882
            if (relu) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (!(o.e[i] > (f16)0)) o.e[i] = (f16)0;     // K.cpp:2586-2590
            }
            f16 *dst = out + r * ldo + col0;
            if (col0 + VEC <= n_feat) {
                *reinterpret_cast<Elem<f16>::vec16_u *>(dst) = o.q;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (col0 + i < n_feat) dst[i] = o.e[i];
            }
        }
    }
}

template <int LPR>
int launch_rows(int spmm_block, int threads, int relu, int n_rows, int n_feat, const int32_t *rowPtr, const int32_t *columnIndex,
                const void *values, const void *table, unsigned t_bytes, unsigned ld_bytes, void *out, int64_t ldo, hipStream_t s)
{
    const int rows_per_block = (64 / LPR) * (kBlock / 64);
    hipLaunchKernelGGL((refhalf_csr_rows_kernel<LPR>), dim3((unsigned)((n_rows + rows_per_block - 1) / rows_per_block)),
                       dim3(kBlock), 0, s, n_rows, n_feat, spmm_block, threads, rowPtr, columnIndex, (const f16 *)values,
                       (const f16 *)table, t_bytes, ld_bytes, (f16 *)out, ldo, relu);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
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

// The dense stage in the lane-group layout: a group of LPR lanes owns a row, a lane 8 output columns.  The
// weight rows W[k][0:P] of a block of k sit in LDS (written transposed from B = W^T); per k a lane reads
// its 16 bytes of W[k], multiplies by x[r][k] and adds into partial k mod 4 -- static inside the unrolled
// loop.  The reference's partial index is (phase + k) mod 4 with a per-row phase; since the four chains are
// independent, the partials are renamed by `phase` once, before the fold.
template <int LPR>
__global__ __launch_bounds__(kBlock) void refhalf_dense_rows_kernel(
    int n_rows, int M, int n_feat, int spmm_block, int threads, int kc_rows, const f16 *__restrict__ X, int64_t ldx,
    const f16 *__restrict__ Wt, int64_t ldw, f16 *__restrict__ out, int64_t ldo)
{
    constexpr int VEC = 8, RPW = 64 / LPR, PL = LPR * VEC;        // PL: padded row width of the LDS tile
    constexpr int kRowsPerBlock = RPW * (kBlock / 64);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f16 *Wl = reinterpret_cast<f16 *>(smem_raw);                  // [kc_rows][PL]
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int col0 = sub * VEC;
    const bool one_block_of_k = M <= kc_rows;                     // then W is staged once and the workgroup walks row groups
    const int64_t n_groups = ((int64_t)n_rows + kRowsPerBlock - 1) / kRowsPerBlock;

    auto stage = [&](int kc, int kn) {
        __syncthreads();
        for (int i = threadIdx.x; i < kc_rows * PL; i += kBlock) {
            const int kk = i / PL, j = i - kk * PL;
            Wl[i] = (kk < kn && j < n_feat) ? Wt[(int64_t)j * ldw + kc + kk] : (f16)0;
        }
        __syncthreads();
    };
    if (one_block_of_k) stage(0, M);

    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t r = g * kRowsPerBlock + (threadIdx.x >> 6) * RPW + grp;
        const bool live = r < n_rows;
        const f16 *x = X + (live ? r : 0) * ldx;
        h2 q[kLat][4];
#pragma unroll
        for (int l = 0; l < kLat; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) q[l][j] = (h2){(f16)0, (f16)0};
        for (int kc = 0; kc < M; kc += kc_rows) {                 // kc_rows is a multiple of 8
            const int kn = M - kc < kc_rows ? M - kc : kc_rows;
            if (!one_block_of_k) stage(kc, kn);
            for (int k8 = 0; k8 < kn; k8 += 8) {
                union { u32x4 v; f16 e[8]; } xv;
                if (live && k8 + 8 <= kn) {
                    xv.v = *reinterpret_cast<const Elem<f16>::vec16_u *>(x + kc + k8);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) xv.e[i] = (live && k8 + i < kn) ? x[kc + k8 + i] : (f16)0;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {                      // k = kc + k8 + i: k mod 4 == i mod 4
                    union { u32x4 v; h2 h[4]; } w;
                    w.v = *reinterpret_cast<const u32x4 *>(Wl + (k8 + i) * PL + col0);
                    const h2 xx = (h2){xv.e[i], xv.e[i]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[i & 3][j] = q[i & 3][j] + xx * w.h[j];
                }
            }
        }
        if (live && col0 < n_feat) {
            const int phase = (int)(((int64_t)(r - sblock_first_row((int)r, n_rows, spmm_block, threads)) * M) % kLat);
            union { u32x4 v; h2 h[4]; f16 e[8]; } o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h2 p[kLat];                                        // partial l of the reference = chain (l - phase) mod 4
#pragma unroll
                for (int l = 0; l < kLat; ++l) {
                    const int src = (l - phase) & 3;
                    p[l] = src == 0 ? q[0][j] : src == 1 ? q[1][j] : src == 2 ? q[2][j] : q[3][j];
                }
                o.h[j] = ((p[0] + p[1]) + p[2]) + p[3];
            }
            f16 *dst = out + r * ldo + col0;
            if (col0 + VEC <= n_feat) {
                *reinterpret_cast<Elem<f16>::vec16_u *>(dst) = o.v;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (col0 + i < n_feat) dst[i] = o.e[i];
            }
        }
    }
}

template <int LPR>
int launch_dense_rows(int spmm_block, int threads, int n_rows, int M, int n_feat, const void *X, int64_t ldx, const void *Wt,
                      int64_t ldw, void *out, int64_t ldo, hipStream_t s)
{
    const int PL = LPR * 8;
    int kc_rows = (int)((64 * 1024) / (PL * sizeof(f16)));       // <= 64 KB of LDS for the W block
    if (kc_rows > 128) kc_rows = 128;
    kc_rows = kc_rows / 8 * 8;
    const size_t lds = (size_t)kc_rows * PL * sizeof(f16);
    const int rows_per_block = (64 / LPR) * (kBlock / 64);
    if (lds > 48 * 1024)
        SGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&refhalf_dense_rows_kernel<LPR>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t blocks = ((int64_t)n_rows + rows_per_block - 1) / rows_per_block;
    if (M <= kc_rows && blocks > 256 * 8) blocks = 256 * 8;       // W staged once per workgroup, row groups walked in a stride
    hipLaunchKernelGGL((refhalf_dense_rows_kernel<LPR>), dim3((unsigned)blocks), dim3(kBlock), lds, s, n_rows, M, n_feat,
                       spmm_block, threads, kc_rows, (const f16 *)X, ldx, (const f16 *)Wt, ldw, (f16 *)out, ldo);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

}  // namespace

int sgx_refhalf_csr(int spmm_block, int threads, int relu, int n_rows, int n_cols, int n_feat, const int32_t *rowPtr,
                    const int32_t *columnIndex, const void *values, const void *table, int64_t ldt, void *out, int64_t ldo,
                    hipStream_t s)
{
    if (n_rows == 0) return SGX_OK;
    if (spmm_block < 1) spmm_block = 1;
    if (threads < 1) threads = 1;
    // lane-group form when a row is at least 4 lanes wide and the table can be gathered 16 bytes at a time
    const unsigned long long t_bytes = (unsigned long long)n_cols * (unsigned long long)ldt * 2ull;
    int lpr = sgx_next_pow2((n_feat + 7) / 8);
    if (lpr > 64) lpr = 64;
    if (lpr < 4) lpr = 4;                 // a piece must span the 4 partial sums; narrow rows leave lanes of the group idle
    if ((uintptr_t)table % 16 == 0 && (ldt * 2) % 16 == 0 && t_bytes <= kOOBRow && n_cols > 0) {
        const unsigned tb = (unsigned)t_bytes, lb = (unsigned)(ldt * 2);
        switch (lpr) {
        case 4: return launch_rows<4>(spmm_block, threads, relu, n_rows, n_feat, rowPtr, columnIndex, values, table, tb, lb, out, ldo, s);
        case 8: return launch_rows<8>(spmm_block, threads, relu, n_rows, n_feat, rowPtr, columnIndex, values, table, tb, lb, out, ldo, s);
        case 16: return launch_rows<16>(spmm_block, threads, relu, n_rows, n_feat, rowPtr, columnIndex, values, table, tb, lb, out, ldo, s);
        case 32: return launch_rows<32>(spmm_block, threads, relu, n_rows, n_feat, rowPtr, columnIndex, values, table, tb, lb, out, ldo, s);
        default: return launch_rows<64>(spmm_block, threads, relu, n_rows, n_feat, rowPtr, columnIndex, values, table, tb, lb, out, ldo, s);
        }
    }
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
    if (n_feat <= 512 && n_rows >= 1024) {                      // lane-group form; tiny inputs keep the one-thread-per-output kernel
        int lpr = sgx_next_pow2((n_feat + 7) / 8);
        switch (lpr) {
        case 1: return launch_dense_rows<1>(spmm_block, threads, n_rows, M, n_feat, X, ldx, Wt, ldw, out, ldo, s);
        case 2: return launch_dense_rows<2>(spmm_block, threads, n_rows, M, n_feat, X, ldx, Wt, ldw, out, ldo, s);
        case 4: return launch_dense_rows<4>(spmm_block, threads, n_rows, M, n_feat, X, ldx, Wt, ldw, out, ldo, s);
        case 8: return launch_dense_rows<8>(spmm_block, threads, n_rows, M, n_feat, X, ldx, Wt, ldw, out, ldo, s);
        case 16: return launch_dense_rows<16>(spmm_block, threads, n_rows, M, n_feat, X, ldx, Wt, ldw, out, ldo, s);
        case 32: return launch_dense_rows<32>(spmm_block, threads, n_rows, M, n_feat, X, ldx, Wt, ldw, out, ldo, s);
        default: return launch_dense_rows<64>(spmm_block, threads, n_rows, M, n_feat, X, ldx, Wt, ldw, out, ldo, s);
        }
    }
    const int64_t total = (int64_t)n_rows * n_feat;
    hipLaunchKernelGGL(refhalf_dense_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n_rows,
                       M, n_feat, spmm_block, threads, (const f16 *)X, ldx, (const f16 *)Wt, ldw, (f16 *)out, ldo);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}
