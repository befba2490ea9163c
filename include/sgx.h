/*
 * sgx.h -- C ABI of libsgx.so, the MI355X (gfx950) replacement for the one device
 * kernel of hadimsnj/SGRACEx1: the fused GNN layer  D = act( A . (X . W) ).
 *
 * Every entry point below is what the reference's host code binds for this path.
 * Citations are relative to the reference checkout:
 *   K.cpp  = gnn-rfsoc-mt-all-2022/src/kernelMatrixmult_all.cpp
 *   KH     = gnn-rfsoc-mt-all-2022/src/kernelMatrixmult.h
 *   MM.h   = gnn-rfsoc-mt-all-2022/src/matrix_mult.h
 *   MOL    = jupyter/molecule_gcn/Graph_Classification.ipynb   (cell numbers, 0-based)
 *   MMN    = jupyter/test/mmult-master.ipynb
 *   SG.py  = demo/sgrace_lib/sgrace.py
 *
 * Conventions (same as the reference, K.cpp:3762-3774):
 *   - all matrix pointers are DEVICE pointers (HBM), caller-owned; the library only
 *     reads inputs and writes D / E / S / the workspace;
 *   - indices are int32, zero based; CSR = rowPtr[N+1], columnIndex[nnz], values[nnz]; the number of
 *     stored entries is read from rowPtr on the device, so columnIndex / values must be valid
 *     pointers even for a matrix without entries (nothing is read through them then);
 *   - B holds the weights TRANSPOSED, [P_w][M_fea] row-major (K.cpp:3043, MOL cell 16);
 *   - D is [N_adj][P_w] row-major; the feature matrix has M_adj rows (K.cpp:3734);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream and re-entrant per stream; they never synchronise,
 *     allocate or free, so they can be captured in a hipGraph;
 *   - return value: SGX_OK (0) or a negative sgx_status.  The reference validates
 *     nothing and returns nothing (K.cpp:3762); the argument checks are new.
 */
#ifndef SGX_H
#define SGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGX_VERSION 107

typedef enum sgx_status {
    SGX_OK = 0,
    SGX_ERR_NULL = -1,         /* a required pointer is NULL                         */
    SGX_ERR_SHAPE = -2,        /* a dimension is negative / zero where it may not be */
    SGX_ERR_UNSUPPORTED = -3,  /* dtype / mode not built                             */
    SGX_ERR_WORKSPACE = -4,    /* workspace missing or too small                     */
    SGX_ERR_HIP = -5,          /* a HIP call or launch failed                        */
    SGX_ERR_CSR = -6,          /* sgx_csr_validate: rowPtr not monotone / index out of range */
    SGX_ERR_ALIGN = -7         /* pointer or leading dimension not aligned as required */
} sgx_status;

/* Element type of B, D, values_fea, values_adj (MM.h:76-148 selects ONE type for all:
 * HALF in the live build, FLOAT optional; SG.py:1545 allocates float32 buffers). */
typedef enum sgx_dtype { SGX_F16 = 0, SGX_F32 = 1 } sgx_dtype;

/* How sums are formed.
 *   SGX_ACC_F32       products and sums in fp32, one rounding to the storage type at the
 *                     end (default; compared with the exact oracle at a stated tolerance).
 *   SGX_ACC_REF_HALF  fp16 only: every product and every add rounded to binary16, element
 *                     k of an sblock accumulated in partial-sum lane k mod 4, lanes folded
 *                     ((p0+p1)+p2)+p3 -- the arithmetic of the reference's HALF build
 *                     (K.cpp:1829-1884, :2009-2061; MM.h:137-138), bit for bit.       */
typedef enum sgx_acc_mode { SGX_ACC_F32 = 0, SGX_ACC_REF_HALF = 1 } sgx_acc_mode;

/* ---- row schedule of a CSR matrix ------------------------------------------------
 * The reference splits rows over ADJ_THREADS / FEA_THREADS by row count and groups
 * SPMM_BLOCK rows per pipelined loop (K.cpp:3517-3523, :826-845; MM.h:166-191).  On the
 * GPU the same two decisions are made once per matrix: which rows are packed several to
 * a wavefront (the sblock path) and which are long enough to be cut into edge chunks
 * handled by separate wavefronts.  A plan is optional: without one every row takes the
 * sblock path (correct, slower on power-law graphs). */
typedef struct sgx_plan sgx_plan;

/* Builds the plan for rowPtr (device), on the device (csrc/plan_build.hip): the host reads back the entry count and
 * then the number of long rows / tasks it has to size arrays for -- 28 bytes, never rowPtr -- so `stream` is
 * synchronised twice (the fill kernels are still in flight on `stream` when this returns; the builder's scratch comes
 * from the stream-ordered pool); not capturable.
 * n_feat_hint is unused (kept for callers of the first version). */
int sgx_plan_create(sgx_plan **plan, const int32_t *rowPtr, int n_rows, int n_feat_hint,
                    void *stream);
/* The same with the cut chosen by the caller: rows over `long_threshold` edges are split into tasks of `chunk`
 * edges (0 = the default: both sqrt(nnz) / 2 rounded down to a power of two, 64 .. 4096 -- the measured optimum of the A.H
 * aggregation moves with the size of the graph; the first stage of the GAT aggregate runs best with 256 / 256).
 * Matrices under 2^20 entries always use 64 / 64; a cut above 65536 is taken as 65536. */
int sgx_plan_create_ex(sgx_plan **plan, const int32_t *rowPtr, int n_rows, int long_threshold, int chunk,
                       void *stream);
void sgx_plan_destroy(sgx_plan *plan);
/* number of rows that take the split path, and the edge count above which a row does (for reports /
 * tests): sqrt(nnz) / 2 as a power of two (at most 4096) by default, 64 for matrices under 2^20 stored entries, whose
 * run time is the longest row's chain of dependent steps */
int sgx_plan_long_rows(const sgx_plan *plan);
int sgx_plan_long_threshold(const sgx_plan *plan);
/* share of lane-group steps that do work when 8 consecutive rows are packed per wavefront, and
 * whether the plan therefore schedules the short rows in degree order instead (1) or not (0) */
float sgx_plan_natural_utilization(const sgx_plan *plan);
int sgx_plan_reordered(const sgx_plan *plan);
/* One of the plan's device arrays copied to dst (device, int32, `capacity` entries) for inspection and tests:
 * which = 0 long_row, 1 long_first, 2 task_row, 3 task_e0, 4 task_e1, 5 row_order, 6 win_order (the rows of every 64-row
 * window by length, one byte per row, four to an int32; built for matrices of 2^20 entries and more without long rows),
 * 7 scan_win (per boundary g = 0 .. ceil(nnz / 64) between windows of 64 stored entries: the first row starting at or
 * behind entry 64 g and its first entry, then the same pair or -- when the row before is a long one -- that row and its
 * first entry: the row-aligned entry ranges of the GAT aggregate's scan; built for plans created with a cut of 256 entries that are cut there or hold no
 * longer row).  Returns the array's length
 * (dst NULL: the length only) or a negative sgx error. */
int64_t sgx_plan_export(const sgx_plan *plan, int which, int32_t *dst, int64_t capacity, void *stream);

/* ---- quantised layer of the SGRACE bitstream (SG.py:53-265, :570-667, :1645-1848) -------
 * The reference quantises inside the device kernel from scale registers and states the arithmetic
 * in its emulation branch (SG.py:570-667): features and adjacency values go to an unsigned w_qbits
 * grid, weights and the attention vector to a signed one, each put back on a fractional grid
 * (x_q / 2^(w_qbits-1)); H = X.W is divided by 2^scale_fea, clipped to +-(2^ib - 1)/2^ib and rounded
 * to ib - 1 decimals (ib = internal_quantization); after aggregation and ReLU the result is
 * multiplied by deq_o.  All of it in fp32 (dtype must be SGX_F32).  The fields are the registers of
 * the GAT bitstream (SG.py:335-365, :476; demo/zcu104/gat_all_unsigned.hwh). */
#define SGX_QUANT_ADJ_DONE 1   /* flags: values_adj already hold quantised values (graph cached by the host) */
#define SGX_QUANT_INT8 2       /* flags: gemm_mode 1, qbits <= 8, P_w <= 256: X and W go to the int8 matrix cores as the
                                  integer codes of their grids (sgx_quantize_codes_i8 + sgx_xw_dense_i8) -- X.W summed
                                  exactly in int32 instead of in fp32, X read as 1 byte per element.  Equal to the fp32
                                  form whenever that form's sums are exact (|sum of code products| < 2^24), to fp32
                                  rounding otherwise.  Ignored (fp32 form) where it does not apply.              */
#define SGX_QUANT_INT8_AUTO 4  /* flags: the integer operands where they are the faster form -- M_fea > 128, i.e. where
                                  the fp32 product no longer has its weights-stationary kernels (those hold W for
                                  K <= 128: at K = 128 the two forms tie, at Reddit's 602 the integer form takes half
                                  the time, X being read as bytes).  What `config.hardware_quantize = 1` selects in
                                  the SGRACE library's layers (SG.py:570-616: the bitstream's own quantiser).       */
typedef struct sgx_quant {
    int32_t qbits;              /* config.w_qbits: 8, 4, 2 or 1                                       */
    int32_t scale_fea;          /* register scale_fea                                                 */
    int32_t internal_bits;      /* register quantized_multiplier = internal_quantization (SG.py:476)  */
    int32_t flags;
    float   inv_scale_fea;      /* register quantization_scale_fea = 1 / f_s                          */
    float   zero_fea;           /* f_z                                                                */
    float   inv_scale_w;        /* register quantization_scale_w = 1 / w_s (also used for `attention`) */
    float   zero_w;             /* w_z                                                                */
    float   inv_scale_adj;      /* register quantization_scale_adj = 1 / a_s                          */
    float   zero_adj;           /* a_z                                                                */
    float   deq_factor;         /* register deq_factor = deq_o                                        */
    float   reserved;
    int64_t nnz_adj;            /* registers nnz_adj1..4 (SG.py:1205-1260): stored entries of A       */
    int64_t nnz_fea;            /* registers nnz_fea1..4: stored entries of X (gemm_mode 0)           */
} sgx_quant;

/* The two rounding steps on their own (fp32, device pointers; out may alias x):
 *   out = clip(round(inv_scale * x + zero), lo, hi) / 2^(qbits-1)   unsigned: lo = 0, hi = 2^qbits - 1
 *                                                                   signed:   -+(2^(qbits-1) - 1)
 *   qbits = 1: signed -> -0.5 / +0.5 by sign (SG.py:177-182), unsigned -> clip(round, 0, 1) / 2 (SG.py:184-189)
 *   H = round_decimals(clip(H / 2^scale_fea, +-(2^ib - 1) / 2^ib), ib - 1), in place (SG.py:607-616) */
int sgx_fake_quantize(int is_signed, int qbits, float inv_scale, float zero, int64_t n, const float *x,
                      float *out, void *stream);
int sgx_requantize(int n_rows, int n_feat, int64_t ld, float *H, int scale_fea, int internal_bits, void *stream);

/* Integer operands (what the EIGHTBIT / quantised builds do in hardware, MM.h:85-118): the codes of the w_qbits grids
 * as bytes.  codes[r][c] = clip(round(x / s + z)) - sgx_code_bias(is_signed, qbits), columns n_cols..ldc-1 = 0
 * (unsigned 8-bit codes 0..255 are stored minus 128; every other grid fits a signed byte as it is); value = code /
 * 2^(qbits-1) (1 bit: code / 2).  ldc a multiple of 16, codes 16-byte aligned for sgx_xw_dense_i8. */
int sgx_code_bias(int is_signed, int qbits);
int sgx_quantize_codes_i8(int is_signed, int qbits, float inv_scale, float zero, int n_rows, int n_cols, const float *x,
                          int64_t ldx, int8_t *codes, int64_t ldc, void *stream);
/* H[r][p] = requant( (sum_k Xc[r][k] Wc[p][k] + bias terms) / 2^(2(qbits-1)) ) on v_mfma_i32_16x16x64_i8: Xc unsigned
 * feature codes [n_rows][ldx], Wc signed weight codes in the layout of B, [P][ldw]; the epilogue is the fp32 form's
 * (shift by scale_fea, clip, decimal rounding; internal_bits = 0: none).  P <= 256.  workspace:
 * sgx_xw_dense_i8_workspace_bytes(P) bytes. */
size_t sgx_xw_dense_i8_workspace_bytes(int P);
int sgx_xw_dense_i8(int qbits, int n_rows, int M_fea, int P, const int8_t *Xc, int64_t ldx, const int8_t *Wc, int64_t ldw,
                    int scale_fea, int internal_bits, float *H, int64_t ldh, void *workspace, void *stream);

/* ---- the layer: replaces mmult_top / kernelmult1 (K.cpp:3762, :3969; KH:13-58) ------ */
typedef enum sgx_layer_order {
    SGX_ORDER_REFERENCE = 0,        /* D = act(A.(X.W))                                    */
    SGX_ORDER_AGGREGATE_FIRST = 1   /* D = act((A.X).W)                                    */
} sgx_layer_order;

typedef struct sgx_layer_desc {
    /* AXI-Lite scalars of the reference, same names (K.cpp:3777-3790, MMN cell 13) */
    int32_t gemm_mode;   /* 0: X is CSR (rowPtr_fea, columnIndex_fea, values_fea)
                            1: X is dense row-major [M_adj][M_fea] in values_fea; rowPtr_fea /
                               columnIndex_fea ignored (K.cpp:847-865, :985-1012)            */
    int32_t relu;        /* 1: D = max(D, 0) fused (K.cpp:2586-2590, :801-804)               */
    int32_t gat_mode;    /* 0: GCN aggregate  A.H ; 1: edge-softmax aggregate (SG.py:649-657) */
    int32_t N_adj;       /* rows of A and of D                                               */
    int32_t M_adj;       /* columns of A = rows of X                                         */
    int32_t M_fea;       /* columns of X = rows of W                                         */
    int32_t P_w;         /* columns of W and of D; any value >= 1 (no B_WIDTH_BLOCK tail rule) */
    int32_t bias_count;  /* must be 0; > 0 makes the reference preload and RETURN WITHOUT
                            COMPUTING (K.cpp:3876-3889) -- reproduced: D is left untouched    */
    int32_t dtype;       /* sgx_dtype                                                        */
    int32_t acc_mode;    /* sgx_acc_mode                                                     */
    int32_t spmm_block;  /* SPMM_BLOCK of the reference (MM.h:188); only observable in
                            SGX_ACC_REF_HALF (it fixes the partial-sum lane of each element);
                            0 means 1                                                        */
    int32_t gat_fill_dead_rows; /* gat_mode: what a row of A without a positive entry receives.
                            1: the mean of all rows of Wh -- the reference's dense emulation (its masked
                               row is constant, the softmax uniform over all N nodes, SG.py:638-641);
                            0: zero.  Equal whenever every row has a positive entry (self loops)   */

    /* buffers (device) -- the m_axi ports of K.cpp:3792-3828; the reference's four
     * aliases per port (rowPtr_fea1..4 etc., main_float.cpp:880-887) collapse to one */
    const void    *B;                 /* W^T  [P_w][M_fea]                          */
    void          *D;                 /* out  [N_adj][P_w]                          */
    const int32_t *rowPtr_fea;        /* [M_adj+1]            (gemm_mode 0)          */
    const int32_t *columnIndex_fea;   /* [nnz_fea]            (gemm_mode 0)          */
    const void    *values_fea;        /* [nnz_fea] or dense [M_adj*M_fea]            */
    const int32_t *rowPtr_adj;        /* [N_adj+1]                                   */
    const int32_t *columnIndex_adj;   /* [nnz_adj]                                   */
    const void    *values_adj;        /* [nnz_adj]                                   */

    /* GAT (gat_mode = 1): single head as in the reference (SG.py:1176-1178) */
    const void    *attention;         /* a [2*P_w]  (a1 = a[:P_w], a2 = a[P_w:]), same dtype */
    void          *E;                 /* optional out [nnz_adj] fp32: LeakyReLU(e_ij)        */
    void          *S;                 /* optional out [nnz_adj] fp32: softmax alpha_ij       */
    float          alpha;             /* LeakyReLU slope (SG.py:1172, default 0.2)           */
    /* FEA_THREADS / ADJ_THREADS of the reference (MM.h:166-167; 1, 2 or 4 there): each stage's rows
     * are cut into that many contiguous blocks -- rows/threads each, the remainder to the last
     * (K.cpp:3159-3164, :3517-3523) -- and the SPMM_BLOCK grouping restarts at every block.  Like
     * spmm_block only observable in SGX_ACC_REF_HALF; 0 means 1. */
    int32_t        fea_threads;
    int32_t        adj_threads;
    int32_t        gat_heads;         /* gat_mode: number of heads (see sgx_gat_aggregate); 0 means 1; the
                                         attention buffer then holds gat_heads vectors of 2*P_w/gat_heads */

    /* scratch in HBM for H = X.W (the reference's on-chip C tile, K.cpp:27) and split-row
     * partial sums; at least sgx_layer_workspace_bytes(desc) bytes, 256-byte aligned */
    void          *workspace;
    size_t         workspace_bytes;

    /* optional row schedules (see sgx_plan); NULL = none */
    const sgx_plan *plan_adj;
    const sgx_plan *plan_fea;

    /* optional profiling taps, the counterpart of the reference's profiling[] port
     * (K.cpp:3948-3962): hipEvent_t handles (as void*) recorded on `stream` right before and
     * right after the aggregation stage (A.H or GAT).  NULL = not recorded. */
    void *ev_agg_begin;
    void *ev_agg_end;

    /* optional: run the layer with the quantised arithmetic above (NULL = plain fp16/fp32 layer) */
    const sgx_quant *quant;

    /* sgx_layer_order.  The reference always forms H = X.W first (loop_fea feeds loop_adj, K.cpp:3629-3752).
     * With a dense X narrower than the output (M_fea < P_w: ogbn-products' 100 -> 256) aggregating first,
     * D = act((A.X).W), gathers M_fea instead of P_w columns per edge: the same sums in another association,
     * one rounding to the storage type in between as in the reference's order (Z = A.X in place of H).
     * Only gemm_mode 1, gat_mode 0, SGX_ACC_F32, no quant block; SGX_ERR_UNSUPPORTED otherwise. */
    int32_t order;
} sgx_layer_desc;

size_t sgx_layer_workspace_bytes(const sgx_layer_desc *desc);
int    sgx_layer_forward(const sgx_layer_desc *desc, void *stream);

/* ---- the stages, individually (the dataflow processes of K.cpp:3629-3752) ---------- */

/* A.H aggregation = loop_adj / compute2 / writec (K.cpp:3339, :2483, :713):
 *   D[r][0:n_feat] = act( sum_e values[e] * H[columnIndex[e]][0:n_feat] ),  r in [0,n_rows)
 * H is [n_cols][ldh], D is [n_rows][ldd] (leading dimensions in elements).  Rows of H that start on
 * a dword (H 4-byte aligned, ldh*sizeof(elem) a multiple of 4) are gathered 16 bytes per lane; rows on
 * odd halves -- and, for tables of 4 GiB and more, rows that are not 16-byte aligned -- one element per
 * lane.  Fastest when no row straddles a 128-byte line (ldh*sizeof(elem) a multiple of 128, or a power
 * of two below it).  scratch/scratch_bytes: needed only when `plan` has long rows
 * (sgx_spmm_scratch_bytes). */
int sgx_spmm_csr(int dtype, int acc_mode, int spmm_block, int relu,
                 int n_rows, int n_cols, int n_feat,
                 const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                 const void *H, int64_t ldh, void *D, int64_t ldd,
                 const sgx_plan *plan, void *scratch, size_t scratch_bytes, void *stream);
size_t sgx_spmm_scratch_bytes(const sgx_plan *plan, int n_feat);

/* The same aggregation in two passes over disjoint edge sets, for the multi-GPU path (the edges
 * whose column lives in the rank's own partition while the halo rows travel over xGMI, then the
 * halo edges): pass 1 writes the fp32 sums to acc_out (D = NULL), pass 2 starts from acc_in and
 * writes D = act(acc_in + A.H).  acc_* are [n_rows][ld_acc] fp32; either may be NULL. */
int sgx_spmm_csr_acc(int dtype, int relu, int n_rows, int n_cols, int n_feat,
                     const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                     const void *H, int64_t ldh, void *D, int64_t ldd,
                     const float *acc_in, float *acc_out, int64_t ld_acc,
                     const sgx_plan *plan, void *scratch, size_t scratch_bytes, void *stream);

/* X.W with dense X = loop_fea / compute1 in gemm_mode 1 (K.cpp:2932, :2605, :847-865),
 * on the matrix cores:  H[r][0:P] = sum_k X[r][k] * Wt[p][k].
 * X [n_rows][ldx], Wt [P][ldw] (= B), H [n_rows][ldh]; columns P..ldh-1 of H are zeroed. */
int sgx_xw_dense(int dtype, int acc_mode, int spmm_block, int n_rows, int M_fea, int P,
                 const void *X, int64_t ldx, const void *Wt, int64_t ldw,
                 void *H, int64_t ldh, void *stream);

/* The same product with the activation on its stores, D = act(X.Wt^T): the second stage of
 * SGX_ORDER_AGGREGATE_FIRST (X := A.X) for callers that run the stages themselves (the multi-GPU exchange of
 * sgracex1_amd/dist.py).  fp32-accumulate arithmetic only; pad columns P..ldh-1 are zeroed. */
int sgx_xw_dense_act(int dtype, int relu, int n_rows, int M_fea, int P,
                     const void *X, int64_t ldx, const void *Wt, int64_t ldw,
                     void *H, int64_t ldh, void *stream);

/* X.W with CSR X = loop_fea / compute1 in gemm_mode 0 (K.cpp:1960-2078).
 * W_rowmajor is [M_fea][ldw] (use sgx_transpose to get it from B). */
int sgx_xw_sparse(int dtype, int acc_mode, int spmm_block, int n_rows, int M_fea, int P,
                  const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                  const void *W_rowmajor, int64_t ldw, void *H, int64_t ldh,
                  const sgx_plan *plan, void *scratch, size_t scratch_bytes, void *stream);

/* out[c][r] = in[r][c]; in [rows][ldi], out [cols][ldo]; pads out columns rows..ldo-1 with 0.
 * The weight-tile load B_accel[i][j] = B[i + j*M_fea] (K.cpp:3038-3051). */
int sgx_transpose(int dtype, int rows, int cols, const void *in, int64_t ldi,
                  void *out, int64_t ldo, void *stream);

/* GAT aggregation on an already computed Wh (SG.py:309-314, :634-661), single head:
 *   e_ij = LeakyReLU_alpha(Wh_i.a1 + Wh_j.a2) for stored edges with values[e] > 0,
 *   alpha_ij = softmax_j(e_ij),  D_i = act(sum_j alpha_ij Wh_j).
 * Wh has n_cols rows; row r of the adjacency is node r of Wh (n_rows <= n_cols: the reference's
 * square case is n_rows == n_cols, a rank of the partitioned graph passes its own rows first and
 * the halo rows behind them).
 * fill_dead_rows: see sgx_layer_desc.gat_fill_dead_rows.  s_scratch: sgx_gat_scratch_bytes() bytes
 * (the per-node scores Wh.a1, Wh.a2 and the column-mean partials).  E/S optional [nnz] fp32.
 * n_heads > 1 (BASELINE config 5; the reference itself has one head, SG.py:1176-1178): the formula
 * above on each slice of n_feat / n_heads columns with its own vector attention[h][0 : 2*F_head],
 * outputs concatenated -- what n_heads single-head calls on the slices give; E/S are [nnz][n_heads].
 * plan (optional): rows it marks long are cut into edge chunks with per-chunk softmax states that are
 * merged in a fixed order.  With a plan (which tells the stored-entry count) the aggregate runs in two stages -- the
 * softmax weights on the edges, then the A.H aggregation with those weights; sgx_gat_scratch_bytes then includes
 * nnz * n_heads floats for the weights (S takes them when given). */
size_t sgx_gat_scratch_bytes(int n_cols, int n_feat, int n_heads, int fill_dead_rows, const sgx_plan *plan);
int sgx_gat_aggregate(int dtype, int relu, int fill_dead_rows, int n_rows, int n_cols, int n_feat, int n_heads,
                      float alpha,
                      const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                      const void *Wh, int64_t ldh, const void *attention,
                      void *D, int64_t ldd, float *E, float *S, const sgx_plan *plan, float *s_scratch,
                      void *stream);

/* The same aggregate for one rank of a node-partitioned graph: a row without a live edge receives `fill`
 * (fp32 [n_feat], device) -- the mean of the rows of Wh of ALL n_nodes nodes, which the caller reduces across ranks
 * (sgx_col_sums per rank, one all-reduce of n_feat floats) -- and S = 1/n_nodes on its stored edges: the reference's
 * uniform softmax over every node (SG.py:638-641), which one rank's table (own rows + halo rows) cannot give.
 * fill = NULL: such rows give 0.  Scratch: sgx_gat_scratch_bytes(n_cols, n_feat, n_heads, 0, plan). */
int sgx_gat_aggregate_fill(int dtype, int relu, int n_rows, int n_cols, int n_feat, int n_heads, float alpha,
                           const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                           const void *Wh, int64_t ldh, const void *attention, void *D, int64_t ldd, float *E, float *S,
                           const float *fill, int64_t n_nodes, const sgx_plan *plan, float *s_scratch, void *stream);
/* out[j] = sum over rows of X[r][j], fp32, slab sums added in a fixed order (bitwise reproducible).
 * scratch: sgx_col_sums_scratch_bytes(n_feat) bytes. */
size_t sgx_col_sums_scratch_bytes(int n_feat);
int sgx_col_sums(int dtype, int n_rows, int n_feat, const void *X, int64_t ldx, float *out, float *scratch, void *stream);

/* ---- helpers on either side of the path (SURVEY 8f "next" rows) -------------------- */

/* Checks rowPtr[0]==0, monotone, rowPtr[n_rows]==nnz and 0 <= columnIndex < n_cols on the
 * device.  Synchronises the stream.  Returns SGX_OK or SGX_ERR_CSR. */
int sgx_csr_validate(const int32_t *rowPtr, const int32_t *columnIndex, int n_rows, int n_cols,
                     int64_t nnz, void *stream);

/* COO (row index per edge, sorted by row) -> CSR row pointer; the GAT bitstream is fed COO
 * (SG.py:1222, :1245).  rowPtr [n_rows+1]. */
int sgx_coo_to_csr(const int32_t *rowIndex, int64_t nnz, int n_rows, int32_t *rowPtr, void *stream);

/* Weight gradient of the layer's backward pass, grad_W = X^T . G with G = adj @ grad_output already
 * aggregated by sgx_spmm_csr (FPYNQ.backward, MOL cell 16; the reference runs it in torch on the CPU).
 * X [n_rows][ldx] fp16|fp32 dense, G [n_rows][ldg] fp32, out [M][ldo] fp32 (exact fp32 fma chains,
 * slab sums added in a fixed order).  The other two products of the backward pass are existing
 * entry points: adj @ g = sgx_spmm_csr, grad_x = G . W^T = sgx_xw_dense(G, Wt := W [M][P]); for a CSR
 * X, X^T . G = sgx_spmm_csr over the CSR of X^T. */
size_t sgx_xt_g_workspace_bytes(int n_rows, int M, int P);
int sgx_xt_g(int dtype_x, int n_rows, int M, int P, const void *X, int64_t ldx, const float *G, int64_t ldg,
             float *out, int64_t ldo, void *workspace, size_t workspace_bytes, void *stream);

/* Edge pass of the GAT layer's backward (FPYNQ_GAT.backward, SG.py:884-1126, on the stored edges instead
 * of dense N x N matrices): with E, S the forward's per-edge outputs, G = grad_output [n_rows][ldg] and
 * Wh [n_cols][ldw], all fp32,
 *   d_e = G[row e] . Wh[col e];  dx_e = S_e d_e;  sg_e = dx_e - S_e sum_row(dx);
 *   sg_e = 0 where values[e] <= 0;  sg_e *= (E_e > 0 ? 1 : alpha)
 * writes sg [nnz] and g1[r] = sum_row(sg).  The attention gradient is then [Wh^T g1 ; Wh^T g2] with
 * g2 = the column sums of sg (row sums over A^T: sgx_spmm_csr) through sgx_xt_g. */
int sgx_gat_backward_edges(int dtype_values, int n_rows, int n_cols, int n_feat, float alpha,
                           const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                           const float *E, const float *S, const float *G, int64_t ldg, const float *Wh, int64_t ldw,
                           float *sg, float *g1, void *stream);

/* Readout + classifier head of the graph-classification model (MOL cell 18 tail) in one launch:
 * pooled[g][:] = mean of X rows [graph_ptr[g], graph_ptr[g+1])  (global_mean_pool over a sorted
 * `batch` vector), logits[g][c] = bias[c] + W[c][:] . pooled[g][:]  (torch Linear, W [C][F] fp32).
 * pooled or logits may be NULL (then W, bias are not read); bias may be NULL. */
int sgx_readout_mean_linear(int dtype, int n_graphs, int F, int C, const void *X, int64_t ldx,
                            const int32_t *graph_ptr, const float *W, const float *bias, float *pooled,
                            float *logits, void *stream);

/* Backward of that pooling for the training step (the autograd of global_mean_pool in MOL cell 18's forward, cell 20's
 * loop): grad_X[r][:] = grad_pooled[g][:] / (graph_ptr[g+1] - graph_ptr[g]) for the rows r of graph g, written in
 * `dtype` (the element type of the layer output the pooling read); rows outside every graph are not written. */
int sgx_readout_mean_backward(int dtype, int n_graphs, int F, const float *grad_pooled, const int32_t *graph_ptr,
                              void *grad_X, int64_t ldg, void *stream);

/* ReLU backward of RPYNQ (MOL cell 16): grad[i] = (out[i] == 0) ? 0 : grad[i], in place. */
int sgx_relu_mask_backward(int dtype_out, const void *out, int dtype_grad, void *grad, int64_t n,
                           void *stream);

/* dst[i][0:n_feat] = src[row_index[i]][0:n_feat], i in [0, n_rows): the pack step of the halo exchange between the
 * GPUs of a node -- the rows of H a peer's edges reference, gathered into the send buffer of the all-to-all (the
 * block select of dsp_kernel_float_adj_4, K.cpp:217-264, done on the sending side).  ld_* in elements. */
int sgx_pack_rows(int dtype, int64_t n_rows, int n_feat, const void *src, int64_t ld_src, const int32_t *row_index,
                  void *dst, int64_t ld_dst, void *stream);

/* A plain streaming copy (16 bytes per lane, non-temporal, each workgroup on a contiguous chunk), the kernel the attainable HBM rate of a device is
 * measured with next to the nominal 8 TB/s (bench.py reports it as roofline.stream_copy_GBps_this_device).
 * bytes must be a multiple of 16, both pointers 16-byte aligned. */
int sgx_stream_copy(void *dst, const void *src, int64_t bytes, void *stream);

/* hipEvent_t helpers for the profiling taps of sgx_layer_desc (handles travel as void*), so that
 * a host that does not link the HIP runtime itself can time launches on the library's runtime.
 * sgx_event_elapsed_ms waits for `end` and returns the milliseconds between the two events. */
int sgx_event_create(void **event);
int sgx_event_destroy(void *event);
int sgx_event_record(void *event, void *stream);
int sgx_event_elapsed_ms(void *begin, void *end, float *ms);

int         sgx_version(void);
const char *sgx_status_string(int status);

/* The library reads its SGX_* tuning overrides from the environment once, at its first use.  A process that changes one
 * of them later (a test comparing two kernel forms) calls this to have them read again.  Not for concurrent use with
 * other calls into the library. */
void sgx_reload_env(void);

#ifdef __cplusplus
}
#endif
#endif /* SGX_H */
