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

// Tuning / test overrides: the SGX_* environment variables, read ONCE per process -- not per call: a getenv
// walks the whole environment, and half a dozen of them per layer were a measurable share of a 20-microsecond layer
// (Cora, MUTAG).  Sizing a scratch and launching on it therefore see the same settings.  A process that changes a
// variable afterwards (tests and probes that compare two forms in one process) calls sgx_reload_env().
struct sgx_tuning {
    bool gat_one_pass, gat_no_fused_scores;
    int gat_fused;             // SGX_GAT_FUSED: without E / S outputs the aggregate as one walk (gat_fused.hip): 0 = never, 1 = by shape (default), 2 = wherever it applies
    int gat_scan;              // SGX_GAT_SCAN: 0 = the GAT aggregate's short rows never in entry order, 1 = by shape (default), 2 = always
    bool xw_no_wlds, xw_no_stationary_f32, xw_sparse_no_lds, xw_short_tiles, xw_no_lds;
    bool xtg_scalar, xtg_wave_tiles;
    bool spmm_no_short_tail;      // the one-step tail of a degree order through the sblock path (as in round 2)
    int spmm_cpl;                 // 0 = unset
    int xw_sparse_lpr;            // 0 = unset: lanes per row (slice width / 16 bytes) of the sparse X.W stage's LDS form, at most
    int plan_long_threshold;      // 0 = unset
    int plan_chunk;               // 0 = unset
    float plan_reorder_below;     // < 0 = unset
};
const sgx_tuning &sgx_tune();     // util_kernels.hip

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
    int n_multi;           // the order's rows of two steps and more: row_order[0, n_multi); the rest hold at most 8 edges (-1: no order)
    // The rows of every window of 64 consecutive rows by length, longest first (ties in row order): win_order[64 w + k] =
    // the row (0..63, inside window w) of rank k.  What the LDS form of the sparse X.W stage deals to its sub-tiles
    // (xw_sparse_lds.hip; round 2 sorted every window inside the kernel, 7 % of its instructions); built for the
    // matrices that form can take (2^20 entries and more, no long rows), NULL otherwise.
    uint8_t *win_order;    // [ceil(n_rows / 64) * 64]
    // Where the windows of kScanGranule stored entries meet among the rows, per boundary g = 0 .. n_scan_win (entry
    // g * kScanGranule): scan_win[4 g] = the first row that starts at or behind it, [4 g + 1] = that row's first entry --
    // where the window behind the boundary begins; [4 g + 2], [4 g + 3] = the same pair, or the row before and ITS first
    // entry when that row is a long one (it belongs to the tasks) -- where the window before the boundary ends.  The
    // entry-order form of the GAT aggregate's softmax weights (gat_scan.hip) takes its row-aligned ranges from it; built
    // for plans asked to cut at 256 entries (Csr.gat_plan) that do, or hold no longer row; NULL otherwise.
    int32_t *scan_win;     // [4 (n_scan_win + 1)]
    int64_t n_scan_win;
    float natural_utilization;   // share of lane-group steps doing work when rows are packed in natural order
    int max_degree;              // the longest row (lets a caller skip launches that only serve rows above some length)
};

// leading dimension (elements) the library uses for its own H = X.W scratch.  Rows are padded to a multiple of
// 16 bytes so that every gather is one aligned 16-byte load per lane, and further to whole 128-byte lines where
// that costs at most a third more bytes: a row that straddles a line costs the gather a second line
// (tools/pitch_probe.py, 126 M edges: 47 columns fp16 at pitch 48 / 64: 3.65 / 2.75 ms; 100 columns at pitch
// 104 / 128: 5.58 / 4.59 ms; 72 columns at 72 / 128: 4.51 / 4.57 ms).  Rows under 128 bytes go to the next power
// of two, which never straddles.
static inline int64_t sgx_ldh(int dtype, int P) {
    const int64_t es = dtype == SGX_F16 ? 2 : 4;
    const int64_t row = ((int64_t)P * es + 15) / 16 * 16;
    int64_t pitch = row;
    if (row < 128) {
        pitch = 16;
        while (pitch < row) pitch *= 2;
    } else {
        const int64_t lines = (row + 127) / 128 * 128;
        if (3 * lines <= 4 * row) pitch = lines;
    }
    return pitch / es;
}

// Epilogue of the quantised layer, applied where a stage stores fp32 results (fp16 instantiations ignore it):
//   X.W stage : H = round_decimals(clip(H * rq_shift, +-rq_bound), d) with rq_ten_pow = 10^d   (0 = off; SG.py:607-616)
//   A.H stage : D = act(sum) * out_scale                                                      (0 = off; SG.py:666-667)
// One fp32 operation per rounding point of the reference, as in quant.hip.
struct sgx_epilogue {
    float out_scale, rq_shift, rq_bound, rq_ten_pow;
};
static inline sgx_epilogue sgx_no_epilogue() { return sgx_epilogue{0.0f, 0.0f, 0.0f, 0.0f}; }
sgx_epilogue sgx_requant_epilogue(int scale_fea, int internal_bits);          // quant.hip
// the re-quantisation of one fp32 value of H; contraction must stay off around it
__device__ __forceinline__ float sgx_requant_value(float v, const sgx_epilogue &ep)
{
#pragma clang fp contract(off)
    v = v * ep.rq_shift;
    v = v < -ep.rq_bound ? -ep.rq_bound : v;
    v = v > ep.rq_bound ? ep.rq_bound : v;
    return rintf(v * ep.rq_ten_pow) / ep.rq_ten_pow;
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
                    bool fea_stage = false, int ref_threads = 1, sgx_epilogue ep = sgx_no_epilogue());

// X.W for a CSR X with the weight slice resident in LDS (xw_sparse_lds.hip); the caller checks _applicable first
bool sgx_xw_sparse_lds_applicable(int dtype, int n_rows, int m_fea, int n_feat, int64_t ldh, const sgx_plan *plan);
int sgx_xw_sparse_lds(int dtype, int n_rows, int m_fea, int n_feat, const int32_t *rowPtr, const int32_t *columnIndex,
                      const void *values, const void *W, int64_t ldw, void *H, int64_t ldh, const sgx_plan *plan,
                      sgx_epilogue ep, hipStream_t stream);

// sgx_xw_dense / sgx_gat_aggregate with the quantised layer's epilogue (the public entry points pass none)
// rows copied from one pitch to another (util_kernels.hip); dst 16-byte aligned with a pitch that is a multiple of 16
int sgx_repitch_rows(const void *src, int64_t src_pitch, void *dst, int64_t dst_pitch, int row_bytes, int64_t n_rows,
                     hipStream_t stream);
// X.W, fp16, long K, the weight tile resident in LDS (xw_dense_wlds.hip); SGX_ERR_UNSUPPORTED = not its shape
int sgx_xw_dense_wlds(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                      int relu, hipStream_t stream);
// the fp32 counterpart for K <= 128 and wide outputs (W^T for all columns in LDS, X read once)
int sgx_xw_dense_wlds_f32(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                          int h_aligned, sgx_epilogue ep, int relu, hipStream_t stream);
int sgx_xw_dense_ep(int dtype, int acc_mode, int spmm_block, int n_rows, int M_fea, int P, const void *X, int64_t ldx,
                    const void *Wt, int64_t ldw, void *H, int64_t ldh, hipStream_t stream, sgx_epilogue ep, int relu = 0);
int sgx_gat_aggregate_ep(int dtype, int relu, int fill_dead_rows, int n_rows, int n_cols, int n_feat, int n_heads, float alpha,
                         const int32_t *rowPtr, const int32_t *columnIndex, const void *values, const void *Wh, int64_t ldh,
                         const void *attention, void *D, int64_t ldd, float *E, float *S, const sgx_plan *plan,
                         float *s_scratch, hipStream_t stream, float out_scale, const float *ext_fill = nullptr, int ext_n = 0,
                         int scores_ready = 0);
// gat_scan.hip: the softmax weights of the stored entries of every row of up to kScanMaxRow entries (stage A of the
// two-stage aggregate) as a segmented scan in entry order; longer rows are the plan's tasks
constexpr int kScanGranule = 64, kScanMaxRow = 256;
bool sgx_gat_scan_applicable(const sgx_plan *plan);
int sgx_gat_alpha_scan(int dtype, int n_rows, int n_heads, const sgx_plan *plan, const int32_t *rowptr, const int32_t *col,
                       const void *val, const float *s1, const float *s2, float alpha, float *W, float *E, unsigned char *dead,
                       hipStream_t stream);

// gat_fused.hip: the GAT aggregate in one walk, the neighbours' scores formed from the rows it gathers (no E / S outputs)
struct sgx_gat_fused_args {
    int dtype, lpr, relu, n_work, n_feat, n_heads, long_threshold, vec_store, n_tasks, ldp, n_long, n_multi;
    float alpha, out_scale;
    const int32_t *rowptr, *col, *row_order, *task_row, *task_e0, *task_e1, *long_row, *long_first;
    const void *val, *Wh, *att;
    unsigned h_bytes, ld_bytes;
    const float *s1, *fill;
    void *D;
    int64_t ldd;
    float *pacc, *pm, *pl;
    hipStream_t stream;
};
bool sgx_gat_fused_applicable(int dtype, int n_feat, int n_heads, int lpr);
int sgx_gat_fused(const sgx_gat_fused_args &a);

// GAT layer: the attention scores formed by the X.W kernel's epilogue (fp16, heads of 32 columns, two-stage aggregate)
bool sgx_gat_scores_fusable(int dtype, int n_feat, int n_heads, const sgx_plan *plan);
float *sgx_gat_score_partials(float *s_scratch, int n_cols, int n_feat, int n_heads, int fill_dead_rows);
int sgx_gat_scores_combine(float *s_scratch, int n_cols, int n_feat, int n_heads, int fill_dead_rows, hipStream_t stream);
// sgx_xw_dense_ep for that case: s1 / s2 [n_rows x n_heads] = H.a1 / H.a2 per head beside H; SGX_ERR_UNSUPPORTED when
// the shape is not the stationary kernel's (the caller then runs the plain product and lets the aggregate form the scores)
// (heads of 32 columns: s1 / s2 are the scores themselves; heads of 64 columns and more: one partial per (row, 64-column
// group), [n_rows x P / 64] each, for sgx_gat_scores_combine)
int sgx_xw_dense_scores(int n_rows, int M_fea, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                        const void *attention, int n_heads, float *s1, float *s2, hipStream_t stream);
