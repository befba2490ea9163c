// sgx_layer_forward: the drop-in for mmult_top / kernelmult1 (K.cpp:3762, :3969).
//
// The reference overlaps X.W of output tile b+1 with A.H of tile b through a ping-pong PIPO
// (mmult_wrapper, K.cpp:3629-3752) and re-streams both CSR matrices once per B_WIDTH_BLOCK
// columns.  Here all P_w columns of a row are produced at once, so each CSR is read exactly
// once and the two stages are two launches on one stream with H = X.W kept in HBM scratch
// (in the storage dtype, as the reference keeps its C tile in `half`).
#include "sgx_internal.h"

namespace {

struct Carve {
    size_t h_off, h_bytes;        // H  [M_adj][ldh]
    size_t w_off, w_bytes;        // W  [M_fea][ldh]  (row-major copy of B, gemm_mode 0 only)
    size_t s_off, s_bytes;        // split-row partial sums (max of both stages)
    size_t g_off, g_bytes;        // GAT per-node scores, 2*N floats
    size_t q_off, q_bytes;        // quantised layer: copies of B, X (values), A (values), attention
    size_t qb, qx, qa, qt;        //   their offsets inside the q block
    size_t qw;                    //   int8 form: column sums of the weight codes
    int64_t ldc;                  //   int8 form: row pitch of the code matrices (bytes, multiple of 16); 0 = fp32 form
    size_t total;
};

// Aggregate-first order: X is the gathered table and comes with the caller's pitch (M_fea).  When that pitch lets rows
// straddle 128-byte lines (100 halves: 5.70 ms against 4.59 ms at pitch 128 on the products shape, tools/pitch_probe.py)
// and the graph is large enough for the gathers to outweigh one streaming copy of X, the rows are first copied onto
// whole lines.
bool repitch_x(const sgx_layer_desc *d)
{
    return sgx_ldh(d->dtype, d->M_fea) != d->M_fea && (int64_t)d->M_adj * d->M_fea >= (int64_t)1 << 22;
}

// the int8 operand form applies to a dense X with codes of at most 8 bits and an output of at most 256 columns
bool int8_form(const sgx_layer_desc *d)
{
    if (!d->quant || d->gemm_mode != 1 || d->quant->qbits > 8 || d->P_w > 256) return false;
    if (d->quant->flags & SGX_QUANT_INT8) return true;
    // by shape: the fp32 form keeps W in registers / LDS for K <= 128 and is as fast there; beyond it runs the fp32 tile
    // kernel at the fp32 matrix rate while the integer form reads X as bytes (profiles/r03_configs_c5_int8.jsonl)
    return (d->quant->flags & SGX_QUANT_INT8_AUTO) && d->M_fea > 128;
}

Carve carve(const sgx_layer_desc *d)
{
    Carve c;
    c.qw = 0;
    c.ldc = 0;
    const size_t es = sgx_elem_size(d->dtype);
    const size_t ldh = (size_t)sgx_ldh(d->dtype, d->P_w);
    size_t off = 0;
    if (d->order == SGX_ORDER_AGGREGATE_FIRST) {
        // Z = A.X  [N_adj][ldz] takes the place of H; split-row partials are M_fea wide
        const size_t ldz = (size_t)sgx_ldh(d->dtype, d->M_fea);
        c.h_off = off; c.h_bytes = sgx_align_up((size_t)d->N_adj * ldz * es, 256); off += c.h_bytes;
        // a copy of X with rows on whole lines, when X's own pitch (M_fea) makes its rows straddle them (w block)
        c.w_off = off; c.w_bytes = repitch_x(d) ? sgx_align_up((size_t)d->M_adj * ldz * es, 256) : 0; off += c.w_bytes;
        c.s_off = off; c.s_bytes = sgx_spmm_scratch_bytes(d->plan_adj, d->M_fea); off += c.s_bytes;
        c.g_off = off; c.g_bytes = 0;
        c.q_off = off; c.q_bytes = 0; c.qb = c.qx = c.qa = c.qt = 0;
        c.total = off;
        return c;
    }
    c.h_off = off; c.h_bytes = sgx_align_up((size_t)d->M_adj * ldh * es, 256); off += c.h_bytes;
    c.w_off = off; c.w_bytes = d->gemm_mode == 0 ? sgx_align_up((size_t)d->M_fea * ldh * es, 256) : 0; off += c.w_bytes;
    size_t s1 = sgx_spmm_scratch_bytes(d->plan_adj, d->P_w);
    size_t s2 = d->gemm_mode == 0 ? sgx_spmm_scratch_bytes(d->plan_fea, d->P_w) : 0;
    c.s_off = off; c.s_bytes = s1 > s2 ? s1 : s2; off += c.s_bytes;
    c.g_off = off; c.g_bytes = d->gat_mode ? sgx_gat_scratch_bytes(d->M_adj, d->P_w, d->gat_heads, d->gat_fill_dead_rows, d->plan_adj) : 0; off += c.g_bytes;
    c.q_off = off; c.q_bytes = 0; c.qb = c.qx = c.qa = c.qt = 0;
    if (d->quant) {
        const sgx_quant *q = d->quant;
        size_t o = 0;
        if (int8_form(d)) {
            c.ldc = (int64_t)sgx_align_up((size_t)d->M_fea, 16);
            c.qb = o; o += sgx_align_up((size_t)d->P_w * c.ldc, 256);
            c.qx = o; o += sgx_align_up((size_t)d->M_adj * c.ldc, 256);
            c.qw = o; o += sgx_xw_dense_i8_workspace_bytes(d->P_w);
        } else {
            c.qb = o; o += sgx_align_up((size_t)d->P_w * d->M_fea * 4, 256);
            c.qx = o; o += sgx_align_up((d->gemm_mode == 0 ? (size_t)q->nnz_fea : (size_t)d->M_adj * d->M_fea) * 4, 256);
        }
        c.qa = o; o += (q->flags & SGX_QUANT_ADJ_DONE) ? 0 : sgx_align_up((size_t)q->nnz_adj * 4, 256);
        c.qt = o; o += d->gat_mode ? sgx_align_up((size_t)2 * d->P_w * 4, 256) : 0;
        c.q_bytes = o; off += o;
    }
    c.total = off;
    return c;
}

int check_desc(const sgx_layer_desc *d)
{
    if (!d) return SGX_ERR_NULL;
    if (d->N_adj < 0 || d->M_adj < 0 || d->M_fea < 1 || d->P_w < 1) return SGX_ERR_SHAPE;
    if (d->dtype != SGX_F16 && d->dtype != SGX_F32) return SGX_ERR_UNSUPPORTED;
    if (d->gemm_mode != 0 && d->gemm_mode != 1) return SGX_ERR_UNSUPPORTED;   // 2 = backward offload, not in the public HLS
    if (d->order != SGX_ORDER_REFERENCE && d->order != SGX_ORDER_AGGREGATE_FIRST) return SGX_ERR_UNSUPPORTED;
    if (d->order == SGX_ORDER_AGGREGATE_FIRST &&
        (d->gemm_mode != 1 || d->gat_mode || d->quant || d->acc_mode != SGX_ACC_F32))
        return SGX_ERR_UNSUPPORTED;
    if (d->quant) {
        const sgx_quant *q = d->quant;
        if (d->dtype != SGX_F32 || d->acc_mode != SGX_ACC_F32) return SGX_ERR_UNSUPPORTED;   // SG.py:1545: float32 buffers
        if (q->qbits < 1 || q->qbits > 16 || q->scale_fea < 0 || q->scale_fea > 30 || q->internal_bits < 1 ||
            q->internal_bits > 30)
            return SGX_ERR_UNSUPPORTED;
        if (q->nnz_adj < 0 || q->nnz_fea < 0) return SGX_ERR_SHAPE;
        // entries that are not stored must stay zero after quantisation (a_min = f_min = 0 in every
        // table of SG.py:1298-1537): a non-zero zero point would turn the CSR operands dense
        if (q->zero_adj != 0.0f || (d->gemm_mode == 0 && q->zero_fea != 0.0f)) return SGX_ERR_UNSUPPORTED;
    }
    return SGX_OK;
}

}  // namespace

extern "C" size_t sgx_layer_workspace_bytes(const sgx_layer_desc *desc)
{
    if (check_desc(desc) != SGX_OK) return 0;
    return carve(desc).total;
}

extern "C" int sgx_xw_sparse(int dtype, int acc_mode, int spmm_block, int n_rows, int M_fea, int P,
                             const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                             const void *W_rowmajor, int64_t ldw, void *H, int64_t ldh,
                             const sgx_plan *plan, void *scratch, size_t scratch_bytes, void *stream)
{
    // C[r][:] = sum_k Xval[k] * W[Xcol[k]][:]  (K.cpp:2009-2061).  From 2^20 entries with a plan that cuts no row: the
    // weight slice resident in LDS (xw_sparse_lds.hip, the reference's B_accel); otherwise the aggregation kernel with
    // the weight matrix as the gathered table (it fits L2: M_fea x P elements).  The same sums in the same order.
    return sgx_spmm_launch(dtype, acc_mode, spmm_block, /*relu*/0, n_rows, M_fea, P, rowPtr, columnIndex, values,
                           W_rowmajor, ldw, H, ldh, plan, scratch, scratch_bytes, (hipStream_t)stream, nullptr, nullptr, 0,
                           /*fea_stage*/true);
}

namespace {
int thread_count(int32_t t) { return t < 1 ? 1 : t; }
}

extern "C" int sgx_layer_forward(const sgx_layer_desc *d, void *stream)
{
    int rc = check_desc(d);
    if (rc != SGX_OK) return rc;
    // K.cpp:3876-3889: with bias_count > 0 the kernel only preloads bias/shift/multiplier and returns
    if (d->bias_count > 0) return SGX_OK;
    if (d->N_adj == 0) return SGX_OK;
    if (!d->B || !d->D || !d->rowPtr_adj || !d->values_fea) return SGX_ERR_NULL;
    if (d->gemm_mode == 0 && (!d->rowPtr_fea)) return SGX_ERR_NULL;
    const Carve c = carve(d);
    if (!d->workspace || d->workspace_bytes < c.total) return SGX_ERR_WORKSPACE;
    if ((uintptr_t)d->workspace % 256 != 0) return SGX_ERR_ALIGN;

    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d->workspace;
    void *H = ws + c.h_off;
    void *W = ws + c.w_off;
    void *scratch = c.s_bytes ? ws + c.s_off : nullptr;
    const int64_t ldh = sgx_ldh(d->dtype, d->P_w);

    if (d->order == SGX_ORDER_AGGREGATE_FIRST) {
        // Z = A.X (M_fea columns gathered per edge), then D = act(Z.W) on the matrix cores with the ReLU on its stores
        if (!d->values_adj && d->M_adj > 0) return SGX_ERR_NULL;
        const int64_t ldz = sgx_ldh(d->dtype, d->M_fea);
        const void *table = d->values_fea;
        int64_t ld_table = d->M_fea;
        if (c.w_bytes) {
            const size_t es = sgx_elem_size(d->dtype);
            rc = sgx_repitch_rows(d->values_fea, (int64_t)d->M_fea * es, W, ldz * (int64_t)es, (int)(d->M_fea * es), d->M_adj, s);
            if (rc != SGX_OK) return rc;
            table = W;
            ld_table = ldz;
        }
        if (d->ev_agg_begin) SGX_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_agg_begin, s));
        rc = sgx_spmm_launch(d->dtype, d->acc_mode, 1, /*relu*/0, d->N_adj, d->M_adj, d->M_fea, d->rowPtr_adj,
                             d->columnIndex_adj, d->values_adj, table, ld_table, H, ldz, d->plan_adj, scratch,
                             c.s_bytes, s, nullptr, nullptr, 0);
        if (rc != SGX_OK) return rc;
        if (d->ev_agg_end) SGX_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_agg_end, s));
        return sgx_xw_dense_ep(d->dtype, d->acc_mode, 1, d->N_adj, d->M_fea, d->P_w, H, ldz, d->B, d->M_fea, d->D, d->P_w, s,
                               sgx_no_epilogue(), d->relu);
    }

    // quantised layer: the operands are replaced by their quantised copies (SG.py:574, :593, :624-626)
    const void *B = d->B, *values_fea = d->values_fea, *values_adj = d->values_adj, *attention = d->attention;
    const sgx_quant *q = d->quant;
    if (q) {
        if (!d->values_adj) return SGX_ERR_NULL;
        float *qbase = (float *)(ws + c.q_off);
        float *Bq = (float *)((char *)qbase + c.qb), *Xq = (float *)((char *)qbase + c.qx);
        if (c.ldc) {
            // integer codes for the int8 matrix cores: weights signed, features unsigned (8-bit codes stored minus 128)
            rc = sgx_quantize_codes_i8(1, q->qbits, q->inv_scale_w, q->zero_w, d->P_w, d->M_fea, (const float *)d->B, d->M_fea,
                                       (int8_t *)Bq, c.ldc, s);
            if (rc != SGX_OK) return rc;
            rc = sgx_quantize_codes_i8(0, q->qbits, q->inv_scale_fea, q->zero_fea, d->M_adj, d->M_fea, (const float *)d->values_fea,
                                       d->M_fea, (int8_t *)Xq, c.ldc, s);
            if (rc != SGX_OK) return rc;
        } else {
            rc = sgx_fake_quantize(1, q->qbits, q->inv_scale_w, q->zero_w, (int64_t)d->P_w * d->M_fea, (const float *)d->B, Bq, s);
            if (rc != SGX_OK) return rc;
            const int64_t nx = d->gemm_mode == 0 ? q->nnz_fea : (int64_t)d->M_adj * d->M_fea;
            rc = sgx_fake_quantize(0, q->qbits, q->inv_scale_fea, q->zero_fea, nx, (const float *)d->values_fea, Xq, s);
            if (rc != SGX_OK) return rc;
        }
        B = Bq; values_fea = Xq;
        if (!(q->flags & SGX_QUANT_ADJ_DONE)) {
            float *Aq = (float *)((char *)qbase + c.qa);
            rc = sgx_fake_quantize(0, q->qbits, q->inv_scale_adj, q->zero_adj, q->nnz_adj, (const float *)d->values_adj, Aq, s);
            if (rc != SGX_OK) return rc;
            values_adj = Aq;
        }
        if (d->gat_mode) {
            if (!d->attention) return SGX_ERR_NULL;
            float *Tq = (float *)((char *)qbase + c.qt);
            rc = sgx_fake_quantize(1, q->qbits, q->inv_scale_w, q->zero_w, (int64_t)2 * d->P_w, (const float *)d->attention, Tq, s);
            if (rc != SGX_OK) return rc;
            attention = Tq;
        }
    }

    // the quantised layer's two rounding steps ride on the stores of the stages (SG.py:603-616, :666-667)
    sgx_epilogue ep_h = sgx_no_epilogue(), ep_d = sgx_no_epilogue();
    if (q) {
        ep_h = sgx_requant_epilogue(q->scale_fea, q->internal_bits);
        ep_d.out_scale = q->deq_factor;
    }

    // stage 1: H = X . W          (loop_fea, K.cpp:2932)
    int scores_ready = 0;
    if (d->gat_mode && !q && d->gemm_mode == 1 && d->acc_mode == SGX_ACC_F32 && attention &&
        sgx_gat_scores_fusable(d->dtype, d->P_w, d->gat_heads, d->plan_adj)) {
        // the product's epilogue forms the attention scores beside H (one pass over H less)
        // heads of 32 columns: the scores themselves; wider heads: a partial per 64-column group, added up by a small kernel
        const int heads = d->gat_heads < 1 ? 1 : d->gat_heads;
        float *gs = (float *)(ws + c.g_off);
        const bool direct = d->P_w / heads == 32;
        float *s1 = direct ? gs : sgx_gat_score_partials(gs, d->M_adj, d->P_w, heads, d->gat_fill_dead_rows);
        float *s2 = s1 ? s1 + (size_t)d->M_adj * (direct ? heads : d->P_w / 64) : nullptr;
        rc = s1 ? sgx_xw_dense_scores(d->M_adj, d->M_fea, d->P_w, values_fea, d->M_fea, B, d->M_fea, H, ldh, attention, heads, s1, s2, s)
                : SGX_ERR_UNSUPPORTED;
        if (rc == SGX_OK && !direct) rc = sgx_gat_scores_combine(gs, d->M_adj, d->P_w, heads, d->gat_fill_dead_rows, s);
        if (rc == SGX_OK) scores_ready = 1;
        else if (rc != SGX_ERR_UNSUPPORTED) return rc;
    }
    if (scores_ready) {
        rc = SGX_OK;
    } else if (c.ldc) {
        rc = sgx_xw_dense_i8(q->qbits, d->M_adj, d->M_fea, d->P_w, (const int8_t *)values_fea, c.ldc, (const int8_t *)B, c.ldc,
                             q->scale_fea, q->internal_bits, (float *)H, ldh, ws + c.q_off + c.qw, s);
    } else if (d->gemm_mode == 0) {
        rc = sgx_transpose(d->dtype, d->P_w, d->M_fea, B, d->M_fea, W, ldh, s);         // B [P][M] -> W [M][ldh]
        if (rc != SGX_OK) return rc;
        rc = sgx_spmm_launch(d->dtype, d->acc_mode, d->spmm_block, /*relu*/0, d->M_adj, d->M_fea, d->P_w, d->rowPtr_fea,
                             d->columnIndex_fea, values_fea, W, ldh, H, ldh, d->plan_fea, scratch, c.s_bytes, s, nullptr,
                             nullptr, 0, /*fea_stage*/true, thread_count(d->fea_threads), ep_h);
    } else if (d->acc_mode == SGX_ACC_REF_HALF && thread_count(d->fea_threads) > 1) {
        if (d->dtype != SGX_F16) return SGX_ERR_UNSUPPORTED;
        if (ldh > d->P_w) SGX_HIP_CHECK(hipMemsetAsync(H, 0, (size_t)d->M_adj * ldh * sizeof(f16), s));
        rc = sgx_refhalf_dense(d->spmm_block, d->fea_threads, d->M_adj, d->M_fea, d->P_w, values_fea, d->M_fea, B, d->M_fea,
                               H, ldh, s);
    } else {
        rc = sgx_xw_dense_ep(d->dtype, d->acc_mode, d->spmm_block, d->M_adj, d->M_fea, d->P_w, values_fea, d->M_fea, B, d->M_fea,
                             H, ldh, s, ep_h);
    }
    if (rc != SGX_OK) return rc;

    // stage 2: D = act(A . H)     (loop_adj, K.cpp:3339) or the edge-softmax aggregate (SG.py:634-661)
    if (d->ev_agg_begin) SGX_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_agg_begin, s));
    if (d->gat_mode) {
        if (!attention) return SGX_ERR_NULL;
        rc = sgx_gat_aggregate_ep(d->dtype, d->relu, d->gat_fill_dead_rows, d->N_adj, d->M_adj, d->P_w, d->gat_heads, d->alpha,
                                  d->rowPtr_adj, d->columnIndex_adj,
                                  values_adj, H, ldh, attention, d->D, d->P_w, (float *)d->E, (float *)d->S, d->plan_adj,
                                  (float *)(ws + c.g_off), s, ep_d.out_scale, nullptr, 0, scores_ready);
    } else {
        rc = sgx_spmm_launch(d->dtype, d->acc_mode, d->spmm_block, d->relu, d->N_adj, d->M_adj, d->P_w,
                             d->rowPtr_adj, d->columnIndex_adj, values_adj, H, ldh, d->D, d->P_w, d->plan_adj,
                             scratch, c.s_bytes, s, nullptr, nullptr, 0, /*fea_stage*/false, thread_count(d->adj_threads), ep_d);
    }
    if (rc != SGX_OK) return rc;

    if (d->ev_agg_end) SGX_HIP_CHECK(hipEventRecord((hipEvent_t)d->ev_agg_end, s));
    return SGX_OK;
}

extern "C" int sgx_event_create(void **event)
{
    if (!event) return SGX_ERR_NULL;
    hipEvent_t e;
    SGX_HIP_CHECK(hipEventCreate(&e));
    *event = (void *)e;
    return SGX_OK;
}

extern "C" int sgx_event_destroy(void *event)
{
    if (!event) return SGX_OK;
    SGX_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
    return SGX_OK;
}

extern "C" int sgx_event_record(void *event, void *stream)
{
    if (!event) return SGX_ERR_NULL;
    SGX_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return SGX_OK;
}

extern "C" int sgx_event_elapsed_ms(void *begin, void *end, float *ms)
{
    if (!begin || !end || !ms) return SGX_ERR_NULL;
    SGX_HIP_CHECK(hipEventSynchronize((hipEvent_t)end));
    SGX_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end));
    return SGX_OK;
}

extern "C" int sgx_version(void) { return SGX_VERSION; }

extern "C" const char *sgx_status_string(int status)
{
    switch (status) {
    case SGX_OK: return "ok";
    case SGX_ERR_NULL: return "required pointer is NULL";
    case SGX_ERR_SHAPE: return "bad dimension";
    case SGX_ERR_UNSUPPORTED: return "dtype or mode not supported";
    case SGX_ERR_WORKSPACE: return "workspace missing or too small";
    case SGX_ERR_HIP: return "HIP call or kernel launch failed";
    case SGX_ERR_CSR: return "CSR structure invalid";
    case SGX_ERR_ALIGN: return "pointer or leading dimension misaligned";
    default: return "unknown status";
    }
}
