/*
 * sgx_oracle.c -- CPU oracle for the fused GNN layer  D = act( A . (X . W) ).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it.  Nothing under sgracex1_amd/ links or imports it,
 * and the product path never falls back to it.
 *
 * It restates, in plain C, the algorithm of the reference's single kernel
 * (reference = /root/reference, paths relative to it,
 *  K.cpp = gnn-rfsoc-mt-all-2022/src/kernelMatrixmult_all.cpp,
 *  MM.h  = gnn-rfsoc-mt-all-2022/src/matrix_mult.h,
 *  SG.py = demo/sgrace_lib/sgrace.py):
 *
 *   orc_layer_f64      exact-math statement  D = relu?(A @ (X @ W)), double
 *                      accumulation; the reference's own CPU check
 *                      `csr(adj) @ (csr(fea) @ w)` (jupyter/test/mmult-master.ipynb
 *                      cells 51-53) and `adj @ input @ weight` (molecule_gcn
 *                      notebook cell 17, acc==0 branch).
 *   orc_layer_refhalf  bit-accurate model of the HLS kernel as built with
 *                      `#define HALF` (MM.h:80,129-139): every product and every
 *                      add rounded to IEEE binary16, element k of an sblock
 *                      accumulated in partial-sum lane k mod FADD_LATENCY,
 *                      lanes folded ((p0+p1)+p2)+p3, sblock interval routing.
 *                      Follows K.cpp:815-867 (readptr_fea), :869-895
 *                      (readptr_adj), :952-1015 (readval_fea), :1778-1898
 *                      (dsp_kernel_wrapper_adj_1), :1960-2078
 *                      (dsp_kernel_wrapper_fea), :2483-2603 (compute2_1),
 *                      :2605-2712 (compute1_1), :713-812 (writec).
 *   orc_gat_f64        single-head GAT forward of SG.py:309-314, :634-661
 *                      restated on the edge list (softmax over the stored
 *                      neighbours whose adjacency value is > 0).
 *
 * The Xilinx `half` type is not in the reference checkout and the HLS headers
 * are absent from this image, so the reference C++ is NOT buildable here; the
 * model is pinned instead against the reference's own recorded outputs
 * (csim log, README, notebook cells) -- see tests/test_oracle_pinned.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------- */
/* IEEE binary16 <-> binary32, round-to-nearest-even, subnormals kept.        */
/* ------------------------------------------------------------------------- */
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

uint16_t orc_f32_to_f16(float f)
{
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t abs = x & 0x7fffffffu;
    if (abs >= 0x7f800000u) {                      /* inf / nan */
        if (abs > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((abs >> 13) & 0x1ffu));
        return (uint16_t)(sign | 0x7c00u);
    }
    if (abs >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);   /* >= 65520 -> inf */
    if (abs < 0x33000001u) return (uint16_t)sign;                /* <= 2^-25 -> 0 (tie to even) */
    int32_t exp = (int32_t)(abs >> 23) - 127;
    uint32_t man = (abs & 0x7fffffu) | 0x800000u;
    uint32_t shift, half;
    if (exp < -14) {                                /* subnormal half */
        shift = (uint32_t)(13 + (-14 - exp));       /* 14..24 */
        half = 0;
    } else {
        shift = 13;
        half = (uint32_t)(exp + 15) << 10;
        man &= 0x7fffffu;
    }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    half += q;
    if (rem > halfway || (rem == halfway && (q & 1u))) half += 1u;   /* carries into exponent correctly */
    return (uint16_t)(sign | half);
}

float orc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        float v = (float)man * (1.0f / 16777216.0f);            /* man * 2^-24, exact */
        return (sign ? -v : v);
    }
    if (exp == 31) return bits_f32(sign | 0x7f800000u | (man << 13));
    return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

static inline float rhalf(float x) { return orc_f16_to_f32(orc_f32_to_f16(x)); }

void orc_f32_to_f16_array(const float *in, uint16_t *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = orc_f32_to_f16(in[i]);
}
void orc_f16_to_f32_array(const uint16_t *in, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = orc_f16_to_f32(in[i]);
}

/* ------------------------------------------------------------------------- */
/* Exact-math oracle.                                                          */
/*   A   : CSR (rowptr_a[N+1], col_a, val_a)   N x N  (any rectangular N x M_adj) */
/*   X   : gemm_mode 0 -> CSR (rowptr_x[M_adj+1], col_x, val_x), M_adj x M_fea  */
/*         gemm_mode 1 -> dense row-major val_x[M_adj * M_fea]                  */
/*   Wt  : weights TRANSPOSED, [P][M_fea] row-major, exactly what the reference */
/*         passes in B (K.cpp:3043 `B[i + j*M_fea + ...]`, molecule notebook    */
/*         cell 16 `torch.transpose(weights,0,1)`).                             */
/*   D   : [N][P] row-major.                                                    */
/*   h_round: 0 keep H = X.W in double; 1 round H to float; 2 round H to half   */
/*         (the device keeps H in the storage dtype, like the reference C tile).*/
/*   H_out (optional, may be NULL): receives H as float [M_adj][P].             */
/* ------------------------------------------------------------------------- */
static void xw_f64(int gemm_mode, int M_adj, int M_fea, int P,
                   const int32_t *rowptr_x, const int32_t *col_x, const float *val_x,
                   const float *Wt, int h_round, double *H)
{
    for (int r = 0; r < M_adj; r++) {
        double *hr = H + (size_t)r * P;
        for (int j = 0; j < P; j++) hr[j] = 0.0;
        if (gemm_mode == 0) {
            for (int32_t k = rowptr_x[r]; k < rowptr_x[r + 1]; k++) {
                double v = val_x[k];
                int c = col_x[k];
                for (int j = 0; j < P; j++) hr[j] += v * (double)Wt[(size_t)j * M_fea + c];
            }
        } else {
            const float *xr = val_x + (size_t)r * M_fea;
            for (int j = 0; j < P; j++) {
                const float *w = Wt + (size_t)j * M_fea;
                double s = 0.0;
                for (int c = 0; c < M_fea; c++) s += (double)xr[c] * (double)w[c];
                hr[j] = s;
            }
        }
        if (h_round == 1) for (int j = 0; j < P; j++) hr[j] = (double)(float)hr[j];
        if (h_round == 2) for (int j = 0; j < P; j++) hr[j] = (double)rhalf((float)hr[j]);
    }
}

int orc_layer_f64(int gemm_mode, int relu, int N, int M_adj, int M_fea, int P,
                  const int32_t *rowptr_a, const int32_t *col_a, const float *val_a,
                  const int32_t *rowptr_x, const int32_t *col_x, const float *val_x,
                  const float *Wt, int h_round, float *D, float *H_out)
{
    double *H = (double *)malloc(sizeof(double) * (size_t)M_adj * P + 8);
    double *acc = (double *)malloc(sizeof(double) * (size_t)P + 8);
    if (!H || !acc) { free(H); free(acc); return -1; }
    xw_f64(gemm_mode, M_adj, M_fea, P, rowptr_x, col_x, val_x, Wt, h_round, H);
    if (H_out) for (size_t i = 0; i < (size_t)M_adj * P; i++) H_out[i] = (float)H[i];
    for (int r = 0; r < N; r++) {
        for (int j = 0; j < P; j++) acc[j] = 0.0;
        for (int32_t e = rowptr_a[r]; e < rowptr_a[r + 1]; e++) {
            double a = val_a[e];
            const double *h = H + (size_t)col_a[e] * P;
            for (int j = 0; j < P; j++) acc[j] += a * h[j];
        }
        float *d = D + (size_t)r * P;
        for (int j = 0; j < P; j++) {
            float v = (float)acc[j];
            /* K.cpp:2586-2590 / :801-804: keep when (v > 0 || relu == 0), else +0 */
            d[j] = (v > 0.0f || !relu) ? v : 0.0f;
        }
    }
    free(H); free(acc);
    return 0;
}

/* Plain CSR SpMM  D = relu?(A . H), H dense [M][ldh] float, double accumulate.
 * Used as the CPU baseline kernel and for SpMM-only parity.  nthreads > 1 splits
 * rows in contiguous blocks (the reference's ADJ_THREADS split, K.cpp:3517-3523). */
int orc_spmm_f32(int relu, int64_t row_begin, int64_t row_end, int P, int64_t ldh, int64_t ldd,
                 const int32_t *rowptr_a, const int32_t *col_a, const float *val_a,
                 const float *H, float *D)
{
    float acc[1024];
    if (P > 1024) return -2;
    for (int64_t r = row_begin; r < row_end; r++) {
        for (int j = 0; j < P; j++) acc[j] = 0.0f;
        for (int32_t e = rowptr_a[r]; e < rowptr_a[r + 1]; e++) {
            float a = val_a[e];
            const float *h = H + (size_t)col_a[e] * ldh;
            for (int j = 0; j < P; j++) acc[j] += a * h[j];
        }
        float *d = D + (size_t)r * ldd;
        for (int j = 0; j < P; j++) d[j] = (acc[j] > 0.0f || !relu) ? acc[j] : 0.0f;
    }
    return 0;
}

/* Dense X.W for rows [row_begin,row_end): H[r][j] = sum_k X[r][k] * W[k][j], W row-major
 * [M][P] -- the `support = torch.mm(input, weight)` half of the reference's CPU path
 * (GNN_arc.pdf Listing 1.3; MOL cell 17 acc==0 branch).  CPU-baseline kernel.          */
int orc_xw_dense_f32(int64_t row_begin, int64_t row_end, int M, int P, int64_t ldx, int64_t ldh,
                     const float *X, const float *W, float *H)
{
    float acc[1024];
    if (P > 1024) return -2;
    for (int64_t r = row_begin; r < row_end; r++) {
        for (int j = 0; j < P; j++) acc[j] = 0.0f;
        const float *x = X + (size_t)r * ldx;
        for (int k = 0; k < M; k++) {
            float xv = x[k];
            const float *w = W + (size_t)k * P;
            for (int j = 0; j < P; j++) acc[j] += xv * w[j];
        }
        float *h = H + (size_t)r * ldh;
        for (int j = 0; j < P; j++) h[j] = acc[j];
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Reference-half model (bit-accurate restatement of the HALF build).          */
/* ------------------------------------------------------------------------- */
typedef struct {
    /* stream cursor over (value, column) pairs -- the A_fifo / col_indices_fifo
     * pair of K.cpp:897-908 / :952-1015 */
    const uint16_t *val;
    const int32_t *col;
    int64_t pos;
    int dense_cols;     /* >0: dense mode, column synthesised as pos mod dense_cols (K.cpp:985-1012) */
} stream_t;

/* One sblock of the pipelined MAC loop (K.cpp:1829-1884 == :2009-2061).
 *   M[z]   running nnz after row z of the block (readptr_*: K.cpp:826-845)
 *   table  half matrix [rows][ld] gathered by column index (B_accel / C_buf)
 *   out    acc2[z] for column j of the table                                  */
static void sblock_mac(stream_t *s, const int *M, int S, int L,
                       const uint16_t *table, int64_t ld, int j, float *out /*[S]*/)
{
    float part[8][16];                      /* acc_part[l][z]  (L <= 8, S <= 16) */
    for (int l = 0; l < L; l++) for (int z = 0; z < S; z++) part[l][z] = 0.0f;
    int BM = M[S - 1];
    for (int k = 0; k < BM; k++) {
        float v = orc_f16_to_f32(s->val[s->pos]);
        int64_t ci = s->dense_cols > 0 ? (s->pos % s->dense_cols) : s->col[s->pos];
        s->pos++;
        float b = orc_f16_to_f32(table[(size_t)ci * ld + j]);
        float prod = rhalf(v * b);          /* (ITYPE)a_val*(ITYPE)b_val, K.cpp:98 / :299 */
        int lo = 0;
        for (int z = 0; z < S; z++) {       /* interval routing, K.cpp:1859 / :2040 */
            if (k >= lo && k < M[z]) part[k % L][z] = rhalf(part[k % L][z] + prod);
            lo = M[z];
        }
    }
    for (int z = 0; z < S; z++) {           /* ACC_PART, K.cpp:1874-1884 / :2050-2061 */
        float a = part[0][z];
        for (int l = 1; l < L; l++) a = rhalf(a + part[l][z]);
        out[z] = a;
    }
}

/* Runs one "thread" (contiguous row block) of a stage over all P columns.
 * The value/column stream is re-walked per column: columns are independent in the
 * reference (one compute unit per column of the W tile, K.cpp:92-107), so the
 * result does not depend on B_WIDTH_BLOCK.                                      */
static void stage_refhalf(int first_row, int row_count, int S, int L,
                          const int32_t *rowptr, const int32_t *col, const uint16_t *val,
                          int dense_cols,
                          const uint16_t *table, int64_t ld, int P,
                          int relu, uint16_t *out, int64_t ldo)
{
    for (int j = 0; j < P; j++) {
        stream_t s;
        s.val = val; s.col = col; s.dense_cols = dense_cols;
        /* reada1/reada2 rebase the pointers to the thread's first row
         * (K.cpp:1316-1326, :1378-1382) */
        s.pos = dense_cols > 0 ? (int64_t)first_row * dense_cols : rowptr[first_row];
        for (int a = 0; a < row_count; a += S) {
            int M[16];
            int brnnz = 0;
            for (int b = 0; b < S; b++) {
                if (a + b < row_count) {
                    int r = first_row + a + b;
                    brnnz += dense_cols > 0 ? dense_cols : (rowptr[r + 1] - rowptr[r]);
                }
                M[b] = brnnz;               /* rows past the end repeat the last value */
            }
            float acc2[16];
            sblock_mac(&s, M, S, L, table, ld, j, acc2);
            for (int b = 0; b < S && a + b < row_count; b++) {
                float v = acc2[b];
                if (!(v > 0.0f || !relu)) v = 0.0f;     /* K.cpp:2586-2590 */
                out[(size_t)(first_row + a + b) * ldo + j] = orc_f32_to_f16(v);
            }
        }
    }
}

/*
 * orc_layer_refhalf: all tensors are binary16 bit patterns.
 *   spmm_block   SPMM_BLOCK (MM.h:188)          1..16
 *   lat_fea/adj  FTYPE_LATENCY_FEA/ADJ (MM.h:137-138), 4 in the HALF build
 *   fea_threads / adj_threads  FEA_THREADS / ADJ_THREADS (MM.h:166-167): rows split
 *                N/threads each, remainder to the last (K.cpp:3159-3164, :3517-3523);
 *                the sblock grouping restarts at each thread's first row.
 *   H_out (optional) receives the intermediate C tile [M_adj][P].
 */
int orc_layer_refhalf(int gemm_mode, int relu, int N, int M_adj, int M_fea, int P,
                      const int32_t *rowptr_a, const int32_t *col_a, const uint16_t *val_a,
                      const int32_t *rowptr_x, const int32_t *col_x, const uint16_t *val_x,
                      const uint16_t *Wt /* [P][M_fea] */,
                      int spmm_block, int lat_fea, int lat_adj, int fea_threads, int adj_threads,
                      uint16_t *D, uint16_t *H_out)
{
    if (spmm_block < 1 || spmm_block > 16 || lat_fea < 1 || lat_fea > 8 || lat_adj < 1 || lat_adj > 8)
        return -2;
    if (fea_threads < 1 || adj_threads < 1) return -2;
    /* W row-major [M_fea][P] so that the stage code gathers rows of a table,
     * mirroring B_accel[i][j] = B[i + j*M_fea] (K.cpp:3038-3051) */
    uint16_t *W = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)M_fea * P + 8);
    uint16_t *H = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)M_adj * P + 8);
    if (!W || !H) { free(W); free(H); return -1; }
    for (int i = 0; i < M_fea; i++)
        for (int j = 0; j < P; j++) W[(size_t)i * P + j] = Wt[(size_t)j * M_fea + i];

    /* loop_fea is called with N_fea := M_adj (K.cpp:3734) */
    int blk = M_adj / fea_threads;
    for (int t = 0; t < fea_threads; t++) {
        int first = t * blk;
        int count = (t == fea_threads - 1) ? (M_adj - first) : blk;
        stage_refhalf(first, count, spmm_block, lat_fea, rowptr_x, col_x, val_x,
                      gemm_mode ? M_fea : 0, W, P, P, /*relu*/0, H, P);
    }
    blk = N / adj_threads;
    for (int t = 0; t < adj_threads; t++) {
        int first = t * blk;
        int count = (t == adj_threads - 1) ? (N - first) : blk;
        stage_refhalf(first, count, spmm_block, lat_adj, rowptr_a, col_a, val_a,
                      0, H, P, P, relu, D, P);
    }
    if (H_out) memcpy(H_out, H, sizeof(uint16_t) * (size_t)M_adj * P);
    free(W); free(H);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* GAT forward, single head (SG.py:309-314, :634-661), on the stored edges.    */
/*   Wh = X @ W                                    (SG.py:601)                  */
/*   e_ij = LeakyReLU_alpha( Wh_i . a[:F] + Wh_j . a[F:] )   (SG.py:309-314,635)*/
/*   softmax over j with adj_ij > 0                (SG.py:638-641)              */
/*   out_i = sum_j alpha_ij Wh_j ; ReLU if relu    (SG.py:649-661)              */
/* Rows without any positive edge: the dense emulation softmaxes a row of      */
/* -9e15 into 1/N everywhere; the call path always adds self loops             */
/* (sym_norm2, SG.py:42), so that case never occurs there.  Here such a row    */
/* yields 0 and the tests never feed one.                                      */
/*   E_out/S_out (optional): per stored edge, e_ij after LeakyReLU and          */
/*   alpha_ij (0 for edges with adj <= 0)           (SG.py:500-502)             */
/* ------------------------------------------------------------------------- */
int orc_gat_f64(int relu, int N, int F, float alpha,
                const int32_t *rowptr_a, const int32_t *col_a, const float *val_a,
                const float *Wh /* [N][F] */, const float *att /* [2F] */,
                float *D, float *E_out, float *S_out)
{
    double *s1 = (double *)malloc(sizeof(double) * (size_t)N + 8);
    double *s2 = (double *)malloc(sizeof(double) * (size_t)N + 8);
    double *acc = (double *)malloc(sizeof(double) * (size_t)F + 8);
    if (!s1 || !s2 || !acc) { free(s1); free(s2); free(acc); return -1; }
    for (int i = 0; i < N; i++) {
        double a = 0, b = 0;
        for (int j = 0; j < F; j++) {
            a += (double)Wh[(size_t)i * F + j] * att[j];
            b += (double)Wh[(size_t)i * F + j] * att[F + j];
        }
        s1[i] = a; s2[i] = b;
    }
    for (int i = 0; i < N; i++) {
        double mx = -INFINITY;
        for (int32_t e = rowptr_a[i]; e < rowptr_a[i + 1]; e++) {
            double x = s1[i] + s2[col_a[e]];
            x = x > 0 ? x : x * (double)alpha;
            if (E_out) E_out[e] = (float)x;
            if (val_a[e] > 0.0f && x > mx) mx = x;
        }
        double den = 0;
        for (int32_t e = rowptr_a[i]; e < rowptr_a[i + 1]; e++) {
            if (!(val_a[e] > 0.0f)) continue;
            double x = s1[i] + s2[col_a[e]];
            x = x > 0 ? x : x * (double)alpha;
            den += exp(x - mx);
        }
        for (int j = 0; j < F; j++) acc[j] = 0;
        for (int32_t e = rowptr_a[i]; e < rowptr_a[i + 1]; e++) {
            double p = 0;
            if (val_a[e] > 0.0f) {
                double x = s1[i] + s2[col_a[e]];
                x = x > 0 ? x : x * (double)alpha;
                p = exp(x - mx) / den;
                const float *h = Wh + (size_t)col_a[e] * F;
                for (int j = 0; j < F; j++) acc[j] += p * h[j];
            }
            if (S_out) S_out[e] = (float)p;
        }
        for (int j = 0; j < F; j++) {
            float v = (float)acc[j];
            D[(size_t)i * F + j] = (v > 0.0f || !relu) ? v : 0.0f;
        }
    }
    free(s1); free(s2); free(acc);
    return 0;
}
