// Internal declarations shared by the HIP translation units of libsgx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "sgx.h"

#define SGX_HIP_CHECK(expr)                     \
    do {                                        \
        hipError_t _e = (expr);                 \
        if (_e != hipSuccess) return SGX_ERR_HIP; \
    } while (0)

#define SGX_LAUNCH_CHECK()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return SGX_ERR_HIP; \
    } while (0)

typedef _Float16 f16;

static inline size_t sgx_elem_size(int dtype) { return dtype == SGX_F16 ? 2 : 4; }
static inline size_t sgx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int sgx_next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// Row schedule (see sgx.h).  Rows with more than `long_threshold` edges are cut into tasks
// of at most `chunk` edges; each task is summed by one wavefront into an fp32 partial row,
// and the partial rows of one long row are added in task order (bitwise reproducible).
struct sgx_plan {
    int n_rows;
    int64_t nnz;
    int long_threshold;
    int chunk;
    int n_long;        // number of long rows
    int n_tasks;       // total edge chunks over all long rows
    // device arrays
    int32_t *long_row;     // [n_long]   row id
    int32_t *long_first;   // [n_long+1] first task of each long row
    int32_t *task_row;     // [n_tasks]
    int32_t *task_e0;      // [n_tasks]
    int32_t *task_e1;      // [n_tasks]
    // Degree-ordered schedule of the rows that are not long (NULL = natural order).  Built only
    // when packing consecutive rows would leave most lane groups idle (power-law graphs): rows are
    // bucketed by the number of 8-edge steps they need, longest first, ascending inside a bucket.
    int32_t *row_order;    // [n_ordered]
    int n_ordered;
    float natural_utilization;   // share of lane-group steps doing work when rows are packed in natural order
};

// leading dimension (elements) the library uses for its own H = X.W scratch: rows are padded
// to a multiple of 16 bytes so that every gather is one aligned 16-byte load per lane
static inline int64_t sgx_ldh(int dtype, int P) {
    int per16 = dtype == SGX_F16 ? 8 : 4;
    return (int64_t)((P + per16 - 1) / per16) * per16;
}

// SGX_ACC_REF_HALF stages (refhalf.hip), fp16 only
int sgx_refhalf_csr(int spmm_block, int threads, int relu, int n_rows, int n_cols, int n_feat, const int32_t *rowPtr, const int32_t *columnIndex,
                    const void *values, const void *table, int64_t ldt, void *out, int64_t ldo, hipStream_t s);
int sgx_refhalf_dense(int spmm_block, int threads, int n_rows, int M, int n_feat, const void *X, int64_t ldx, const void *Wt,
                      int64_t ldw, void *out, int64_t ldo, hipStream_t s);

int sgx_spmm_launch(int dtype, int acc_mode, int spmm_block, int relu, int n_rows, int n_cols, int n_feat,
                    const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                    const void *H, int64_t ldh, void *D, int64_t ldd,
                    const sgx_plan *plan, void *scratch, size_t scratch_bytes, hipStream_t stream,
                    const float *acc_in = nullptr, float *acc_out = nullptr, int64_t ld_acc = 0,
                    bool fea_stage = false, int ref_threads = 1);

// D *= factor, fp32 (the deq_o step of the quantised layer, quant.hip)
int sgx_scale_f32(int64_t n, float *D, float factor, hipStream_t s);
