// Small kernels on either side of the hot path: weight-tile transpose (K.cpp:3038-3051),
// CSR validation (new: the reference validates nothing), COO -> CSR row pointer for the
// COO staging of the GAT host code (SG.py:1222, :1245), the ReLU mask of RPYNQ.backward
// (MOL cell 16), and the row schedule (sgx_plan).
#include "sgx_internal.h"

#include <stdlib.h>

namespace {

constexpr int kBlock = 256;

template <typename T>
__global__ __launch_bounds__(kBlock) void transpose_kernel(int rows, int cols, const T *__restrict__ in, int64_t ldi,
                                                           T *__restrict__ out, int64_t ldo)
{
    // out[c][r] = in[r][c]; 32x32 tile through LDS so that both sides are coalesced
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(int64_t)r * ldi + c] : (T)0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;                           // out row c, out column r
        if (c < cols && r < ldo) out[(int64_t)c * ldo + r] = (r < rows) ? tile[tx][i] : (T)0;
    }
}

__global__ __launch_bounds__(kBlock) void csr_validate_kernel(const int32_t *__restrict__ rowptr,
                                                              const int32_t *__restrict__ col, int n_rows, int n_cols,
                                                              int64_t nnz, int *__restrict__ bad)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if (gid == 0 && (rowptr[0] != 0 || (int64_t)rowptr[n_rows] != nnz)) atomicOr(bad, 1);
    for (int64_t r = gid; r < n_rows; r += stride)
        if (rowptr[r + 1] < rowptr[r]) atomicOr(bad, 2);
    if (col)
        for (int64_t e = gid; e < nnz; e += stride)
            if (col[e] < 0 || col[e] >= n_cols) atomicOr(bad, 4);
}

// rowPtr[r] = number of edges with row index < r  (rowIndex sorted ascending)
__global__ __launch_bounds__(kBlock) void coo_to_csr_kernel(const int32_t *__restrict__ row, int64_t nnz, int n_rows,
                                                            int32_t *__restrict__ rowptr)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = gid; e <= nnz; e += stride) {
        const int prev = e == 0 ? -1 : row[e - 1];
        const int cur = e == nnz ? n_rows : row[e];
        for (int r = prev + 1; r <= cur; ++r) rowptr[r] = (int32_t)e;   // every row in (prev, cur] starts at e
    }
}

template <typename TO, typename TG>
__global__ __launch_bounds__(kBlock) void relu_mask_kernel(const TO *__restrict__ out, TG *__restrict__ grad, int64_t n)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = gid; i < n; i += stride)
        if (out[i] == (TO)0) grad[i] = (TG)0;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// The plain streaming copy the attainable HBM rate is quoted on: 16 bytes per lane, non-temporal both ways, 4 loads in
// flight per lane, every workgroup on a contiguous chunk of its own.  The best of the forms swept on these boxes
// (tools/micro/copy_sweep.hip, profiles/r02_copy_sweep.txt: 5.5-5.6 TB/s read + write at every grid size; the
// grid-stride form this replaces 4.3-5.1 TB/s depending on the grid, hipMemcpy device-to-device 5.1).
__global__ __launch_bounds__(kBlock) void stream_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, int64_t n16)
{
    const int64_t per_block = (n16 + gridDim.x - 1) / gridDim.x;
    int64_t i = (int64_t)blockIdx.x * per_block + threadIdx.x;
    const int64_t end = (int64_t)(blockIdx.x + 1) * per_block < n16 ? (int64_t)(blockIdx.x + 1) * per_block : n16;
    for (; i + 3 * kBlock < end; i += 4 * kBlock) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + kBlock);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * kBlock), d = __builtin_nontemporal_load(src + i + 3 * kBlock);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + kBlock);
        __builtin_nontemporal_store(c, dst + i + 2 * kBlock);
        __builtin_nontemporal_store(d, dst + i + 3 * kBlock);
    }
    for (; i < end; i += kBlock) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// rows of row_bytes bytes from pitch src_pitch to pitch dst_pitch (a multiple of 16, dst 16-byte aligned): one lane per
// 16-byte chunk of a destination row; the source needs element alignment only (under-aligned vector load), the bytes
// of a row's last chunk beyond row_bytes are written as zero
typedef u32x4 u32x4_u __attribute__((aligned(2)));
__global__ __launch_bounds__(kBlock) void repitch_rows_kernel(const char *__restrict__ src, int64_t src_pitch,
                                                              char *__restrict__ dst, int64_t dst_pitch, int row_bytes,
                                                              int64_t n_rows)
{
    const int chunks = (row_bytes + 15) / 16;
    const int64_t total = n_rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / chunks;
        const int k = (int)(i - r * chunks);
        const char *from = src + r * src_pitch + 16 * k;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (16 * k + 16 <= row_bytes) {
            v = *reinterpret_cast<const u32x4_u *>(from);
        } else {
            union { u32x4 v; unsigned short h[8]; } u;
            u.v = v;
            for (int b = 0; 16 * k + 2 * b < row_bytes; ++b) u.h[b] = *reinterpret_cast<const unsigned short *>(from + 2 * b);
            v = u.v;
        }
        *reinterpret_cast<u32x4 *>(dst + r * dst_pitch + 16 * k) = v;
    }
}

// dst[i][:] = src[row_index[i]][:]: the pack step of the halo exchange (rows of H a peer asked for, gathered into the
// send buffer).  One lane per 16-byte chunk of a packed row: the reads are whole rows at random places, the writes
// stream.  A row's last chunk is cut to row_bytes on both sides.
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(const char *__restrict__ src, int64_t src_pitch,
                                                           const int32_t *__restrict__ row_index, char *__restrict__ dst,
                                                           int64_t dst_pitch, int row_bytes, int64_t n_rows)
{
    const int chunks = (row_bytes + 15) / 16;
    const int64_t total = n_rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / chunks;
        const int k = (int)(i - r * chunks);
        const char *from = src + (int64_t)row_index[r] * src_pitch + 16 * k;
        char *to = dst + r * dst_pitch + 16 * k;
        if (16 * k + 16 <= row_bytes) {
            *reinterpret_cast<u32x4_u *>(to) = *reinterpret_cast<const u32x4_u *>(from);
        } else {
            for (int b = 0; 16 * k + 2 * b < row_bytes; ++b)
                *reinterpret_cast<unsigned short *>(to + 2 * b) = *reinterpret_cast<const unsigned short *>(from + 2 * b);
        }
    }
}

int grid_1d(int64_t n)
{
    int64_t b = (n + kBlock - 1) / kBlock;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int sgx_transpose(int dtype, int rows, int cols, const void *in, int64_t ldi, void *out, int64_t ldo,
                             void *stream)
{
    if (rows < 0 || cols < 0 || ldi < cols || ldo < rows) return SGX_ERR_SHAPE;
    if (rows == 0 || cols == 0) return SGX_OK;
    if (!in || !out) return SGX_ERR_NULL;
    // grid.y covers ldo so that the pad columns rows..ldo-1 are zeroed too
    dim3 grid((cols + 31) / 32, (unsigned)((ldo + 31) / 32));
    if (dtype == SGX_F16)
        hipLaunchKernelGGL(transpose_kernel<f16>, grid, dim3(kBlock), 0, (hipStream_t)stream, rows, cols,
                           (const f16 *)in, ldi, (f16 *)out, ldo);
    else if (dtype == SGX_F32)
        hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(kBlock), 0, (hipStream_t)stream, rows, cols,
                           (const float *)in, ldi, (float *)out, ldo);
    else
        return SGX_ERR_UNSUPPORTED;
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

extern "C" int sgx_csr_validate(const int32_t *rowPtr, const int32_t *columnIndex, int n_rows, int n_cols, int64_t nnz,
                                void *stream)
{
    if (!rowPtr) return SGX_ERR_NULL;
    if (n_rows < 0 || n_cols < 0 || nnz < 0) return SGX_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    int *bad = nullptr;
    SGX_HIP_CHECK(hipMalloc(&bad, sizeof(int)));
    int host = 0;
    int rc = SGX_OK;
    if (hipMemsetAsync(bad, 0, sizeof(int), s) != hipSuccess) rc = SGX_ERR_HIP;
    if (rc == SGX_OK) {
        hipLaunchKernelGGL(csr_validate_kernel, dim3(grid_1d(nnz > n_rows ? nnz : n_rows)), dim3(kBlock), 0, s, rowPtr,
                           columnIndex, n_rows, n_cols, nnz, bad);
        if (hipGetLastError() != hipSuccess) rc = SGX_ERR_HIP;
    }
    if (rc == SGX_OK && hipMemcpyAsync(&host, bad, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) rc = SGX_ERR_HIP;
    if (rc == SGX_OK && hipStreamSynchronize(s) != hipSuccess) rc = SGX_ERR_HIP;
    (void)hipFree(bad);
    if (rc != SGX_OK) return rc;
    return host ? SGX_ERR_CSR : SGX_OK;
}

extern "C" int sgx_coo_to_csr(const int32_t *rowIndex, int64_t nnz, int n_rows, int32_t *rowPtr, void *stream)
{
    if (!rowPtr || (nnz > 0 && !rowIndex)) return SGX_ERR_NULL;
    if (nnz < 0 || n_rows < 0) return SGX_ERR_SHAPE;
    hipLaunchKernelGGL(coo_to_csr_kernel, dim3(grid_1d(nnz + 1)), dim3(kBlock), 0, (hipStream_t)stream, rowIndex, nnz,
                       n_rows, rowPtr);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

int sgx_repitch_rows(const void *src, int64_t src_pitch, void *dst, int64_t dst_pitch, int row_bytes, int64_t n_rows,
                     hipStream_t stream)
{
    if (n_rows <= 0 || row_bytes <= 0) return SGX_OK;
    if (row_bytes % 2 != 0 || dst_pitch % 16 != 0 || (uintptr_t)dst % 16 != 0 || dst_pitch < (row_bytes + 15) / 16 * 16)
        return SGX_ERR_ALIGN;
    const int64_t total = n_rows * ((row_bytes + 15) / 16);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(repitch_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, (const char *)src, src_pitch,
                       (char *)dst, dst_pitch, row_bytes, n_rows);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

extern "C" int sgx_pack_rows(int dtype, int64_t n_rows, int n_feat, const void *src, int64_t ld_src, const int32_t *row_index,
                             void *dst, int64_t ld_dst, void *stream)
{
    if (n_rows < 0 || n_feat < 1 || ld_src < n_feat || ld_dst < n_feat) return SGX_ERR_SHAPE;
    if (n_rows == 0) return SGX_OK;
    if (!src || !dst || !row_index) return SGX_ERR_NULL;
    if (dtype != SGX_F16 && dtype != SGX_F32) return SGX_ERR_UNSUPPORTED;
    const int64_t es = (int64_t)sgx_elem_size(dtype);
    const int64_t total = n_rows * ((n_feat * es + 15) / 16);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, (const char *)src,
                       ld_src * es, row_index, (char *)dst, ld_dst * es, (int)(n_feat * es), n_rows);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

extern "C" int sgx_stream_copy(void *dst, const void *src, int64_t bytes, void *stream)
{
    if (bytes < 0 || bytes % 16 != 0) return SGX_ERR_SHAPE;
    if (bytes == 0) return SGX_OK;
    if (!dst || !src) return SGX_ERR_NULL;
    if ((uintptr_t)dst % 16 != 0 || (uintptr_t)src % 16 != 0) return SGX_ERR_ALIGN;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(kBlock), 0, (hipStream_t)stream, (const u32x4 *)src,
                       (u32x4 *)dst, bytes / 16);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

extern "C" int sgx_relu_mask_backward(int dtype_out, const void *out, int dtype_grad, void *grad, int64_t n, void *stream)
{
    if (n < 0) return SGX_ERR_SHAPE;
    if (n == 0) return SGX_OK;
    if (!out || !grad) return SGX_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(grid_1d(n)), block(kBlock);
    if (dtype_out == SGX_F16 && dtype_grad == SGX_F16)
        hipLaunchKernelGGL((relu_mask_kernel<f16, f16>), grid, block, 0, s, (const f16 *)out, (f16 *)grad, n);
    else if (dtype_out == SGX_F16 && dtype_grad == SGX_F32)
        hipLaunchKernelGGL((relu_mask_kernel<f16, float>), grid, block, 0, s, (const f16 *)out, (float *)grad, n);
    else if (dtype_out == SGX_F32 && dtype_grad == SGX_F32)
        hipLaunchKernelGGL((relu_mask_kernel<float, float>), grid, block, 0, s, (const float *)out, (float *)grad, n);
    else if (dtype_out == SGX_F32 && dtype_grad == SGX_F16)
        hipLaunchKernelGGL((relu_mask_kernel<float, f16>), grid, block, 0, s, (const float *)out, (f16 *)grad, n);
    else
        return SGX_ERR_UNSUPPORTED;
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

// ---- accessors of the row schedule (built by plan_build.hip) ----
extern "C" float sgx_plan_natural_utilization(const sgx_plan *plan) { return plan ? plan->natural_utilization : 1.0f; }
extern "C" int sgx_plan_reordered(const sgx_plan *plan) { return plan && plan->row_order ? 1 : 0; }

extern "C" int sgx_plan_long_rows(const sgx_plan *plan) { return plan ? plan->n_long : 0; }
extern "C" int sgx_plan_long_threshold(const sgx_plan *plan) { return plan ? plan->long_threshold : 0; }

// ---- tuning overrides: the environment, read once (sgx_internal.h) -----------------------------------------------------
namespace {
sgx_tuning g_tuning;
bool g_tuning_loaded = false;
void load_tuning()
{
    sgx_tuning t{};
    auto flag = [](const char *name) { return getenv(name) != nullptr; };
    auto num = [](const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; };
    t.gat_one_pass = flag("SGX_GAT_ONE_PASS");
    t.gat_no_fused_scores = flag("SGX_GAT_NO_FUSED_SCORES");
    { const char *v = getenv("SGX_GAT_SCAN"); t.gat_scan = v ? atoi(v) : 1; }
    { const char *v = getenv("SGX_GAT_FUSED"); t.gat_fused = v ? atoi(v) : 1; }
    t.xw_no_wlds = flag("SGX_XW_NO_WLDS");
    t.xw_no_stationary_f32 = flag("SGX_XW_NO_STATIONARY_F32");
    t.xw_sparse_no_lds = flag("SGX_XW_SPARSE_NO_LDS");
    t.xw_short_tiles = flag("SGX_XW_SHORT_TILES");
    t.xw_no_lds = flag("SGX_XW_NO_LDS");
    t.xtg_scalar = flag("SGX_XTG_SCALAR");
    t.xtg_wave_tiles = flag("SGX_XTG_WAVE_TILES");
    t.spmm_no_short_tail = flag("SGX_SPMM_NO_SHORT_TAIL");
    t.spmm_cpl = num("SGX_SPMM_CPL");
    t.xw_sparse_lpr = num("SGX_XW_SPARSE_LPR");
    t.plan_long_threshold = num("SGX_PLAN_LONG_THRESHOLD");
    t.plan_chunk = num("SGX_PLAN_CHUNK");
    const char *rb = getenv("SGX_PLAN_REORDER_BELOW");
    t.plan_reorder_below = rb ? (float)atof(rb) : -1.0f;
    g_tuning = t;
    g_tuning_loaded = true;
}
}  // namespace

const sgx_tuning &sgx_tune()
{
    if (!g_tuning_loaded) load_tuning();
    return g_tuning;
}

extern "C" void sgx_reload_env(void) { load_tuning(); }
