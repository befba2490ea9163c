// Single-head GAT aggregation for gfx950 (the GAT bitstream has no public HLS source; the
// arithmetic is the reference's CPU emulation, SG.py:309-314 and :634-661):
//     s1_i = Wh_i . a[:F]      s2_j = Wh_j . a[F:]
//     e_ij = LeakyReLU_alpha(s1_i + s2_j)            for stored edges with values[e] > 0
//     alpha_ij = softmax_j(e_ij)                     (rows of the masked dense matrix)
//     D_i = act( sum_j alpha_ij Wh_j )
// The emulation builds dense N x N matrices; here the softmax runs over the CSR row: one group
// of LPR lanes per row (the same sblock layout as spmm_csr.hip) walks the edges once with a running
// (max, sum, weighted row) state, rescaled when the maximum moves; the row is normalised at the end.
// The hardware's per-edge side outputs E (pre-softmax) and S (softmax) (SG.py:500-502) are optional
// (S costs a second, gather-free walk over the row once its max and sum are known).
// Rows with no positive edge: the emulation's masked dense row is constant (-9e15 everywhere,
// SG.py:638-641), its softmax uniform over all N nodes, so the row receives the mean of all rows
// of Wh.  sym_norm2's self loops (SG.py:42) keep the plain path away from this case, the quantised
// adjacency does not (small values round to 0).  `fill_dead_rows` selects that result (one more
// pass over Wh for the column means); without it such rows produce 0.
#include "sgx_device.h"

#include <math.h>

namespace {

template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_scores_kernel(int n_rows, int n_feat, const T *__restrict__ Wh, int64_t ldh,
                                                           const T *__restrict__ att, float *__restrict__ s1,
                                                           float *__restrict__ s2, int vec_ok)
{
    constexpr int RPW = 64 / LPR;
    constexpr int TILE = LPR * VEC;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int64_t r = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RPW + grp;
    float p1 = 0.0f, p2 = 0.0f;
    if (r < n_rows) {
        for (int c0 = sub * VEC; c0 < n_feat; c0 += TILE) {
            T h[VEC];
            if (VEC > 1 && vec_ok && c0 + VEC <= n_feat) {
                *reinterpret_cast<u32x4 *>(h) = *reinterpret_cast<const u32x4 *>(Wh + r * ldh + c0);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) h[i] = (c0 + i < n_feat) ? Wh[r * ldh + c0 + i] : (T)0;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if (c0 + i < n_feat) {
                    p1 = __builtin_fmaf(Elem<T>::to_f32(h[i]), Elem<T>::to_f32(att[c0 + i]), p1);
                    p2 = __builtin_fmaf(Elem<T>::to_f32(h[i]), Elem<T>::to_f32(att[n_feat + c0 + i]), p2);
                }
            }
        }
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) {
        p1 += __shfl_xor(p1, off);
        p2 += __shfl_xor(p2, off);
    }
    if (r < n_rows && sub == 0) { s1[r] = p1; s2[r] = p2; }
}

__device__ __forceinline__ float leaky(float x, float alpha) { return x > 0.0f ? x : x * alpha; }

// ReLU (SG.py:660-661), then the quantised layer's deq_o factor on fp32 outputs (SG.py:666-667; 0 = off)
template <typename T>
__device__ __forceinline__ T gat_finish(float sum, int relu, float out_scale)
{
    T v = Elem<T>::from_f32(sum);
    v = (!relu || v > (T)0) ? v : (T)0;
    if constexpr (sizeof(T) == 4) {
        if (out_scale != 0.0f) v = v * out_scale;
    }
    return v;
}

// merge two online-softmax states (m, l); (-inf, 0) is the empty state
__device__ __forceinline__ void softmax_merge(float &m, float &l, float m2, float l2)
{
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; l = 0.0f; return; }
    l = l * expf(m - mn) + l2 * expf(m2 - mn);
    m = mn;
}

__device__ __forceinline__ float rescale_factor(float m_old, float m_new)
{
    return m_old == -INFINITY ? 0.0f : expf(m_old - m_new);        // (-inf) - (-inf) never reaches expf
}

template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_aggregate_kernel(
    int n_rows, int n_cols, int n_feat, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const T *__restrict__ val, const T *__restrict__ Wh, unsigned h_bytes, unsigned ld_bytes,
    const float *__restrict__ s1, const float *__restrict__ s2, float alpha,
    T *__restrict__ D, int64_t ldd, int relu, float *__restrict__ E, float *__restrict__ S, int vec_store,
    const float *__restrict__ fill, int long_threshold, float out_scale)
{
    constexpr int RPW = 64 / LPR;
    constexpr int TILE = LPR * VEC;
    constexpr int UNR = LPR < 8 ? LPR : 8;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int64_t r = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RPW + grp;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Wh), 0, h_bytes, 0x00020000);
    bool live = r < n_rows;
    int e0 = 0, e1 = 0;
    float si = 0.0f;
    if (live) { e0 = rowptr[r]; e1 = rowptr[r + 1]; si = s1[r]; }
    if (live && long_threshold > 0 && e1 - e0 > long_threshold) { live = false; e1 = e0; }   // the split path owns it

    const float uniform = 1.0f / (float)n_cols;

    // One pass over the row's edges with a running softmax state (max m, sum l, weighted row acc): a piece
    // of LPR edges is scored by its lanes (one edge each), the piece maximum is reduced over the group, the
    // state is rescaled when the maximum moves, then the piece's rows are gathered with weights exp(x - m).
    // Rows of up to LPR edges -- most rows of a citation graph at F = 256 -- never rescale.
    for (int c0 = 0; c0 < n_feat; c0 += TILE) {
        const int col0 = c0 + sub * VEC;
        const unsigned col_off = col0 < n_feat ? (unsigned)col0 * (unsigned)sizeof(T) : kOOB;
        float m = -INFINITY, l = 0.0f;              // l: this lane's share of the sum
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        for (int base = e0; base < e1; base += LPR) {
            const int idx = base + sub;
            int c = 0;
            float x = -INFINITY;
            if (idx < e1) {
                c = col[idx];
                const float xe = leaky(si + s2[c], alpha);
                if (E && c0 == 0) E[idx] = xe;
                if (Elem<T>::to_f32(val[idx]) > 0.0f) x = xe;
            }
            float pmax = x;
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, off));
            if (pmax == -INFINITY) continue;        // no live edge in this piece (uniform across the group)
            const float m_new = fmaxf(m, pmax);
            const float scale = rescale_factor(m, m_new);
            const float p = x == -INFINITY ? 0.0f : expf(x - m_new);
            m = m_new;
            l = l * scale + p;
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] *= scale;
            const int n = e1 - base;
#pragma unroll 1
            for (int t0 = 0; t0 < LPR; t0 += UNR) {
                if (t0 >= n) break;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int t = t0 + u;
                    const int cc = __shfl(c, t, LPR);
                    const float pp = __shfl(p, t, LPR);
                    const unsigned off = (t < n && col_off != kOOB) ? (unsigned)cc * ld_bytes + col_off : kOOB;
                    Gather<T, VEC>::run(acc, pp, rsrc, off);
                }
            }
        }
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) l += __shfl_xor(l, off);
        const float inv_l = l > 0.0f ? 1.0f / l : 0.0f;
        const bool dead = live && !(l > 0.0f) && fill != nullptr;
        if (S && c0 == 0) {                         // the softmax values, now that the row's (m, l) are known
            for (int idx = e0 + sub; idx < e1; idx += LPR) {
                float p = 0.0f;
                if (dead) p = uniform;
                else if (Elem<T>::to_f32(val[idx]) > 0.0f) p = expf(leaky(si + s2[col[idx]], alpha) - m) * inv_l;
                S[idx] = p;
            }
        }
        if (live && col0 < n_feat) {
            T out[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = dead ? ((col0 + i < n_feat) ? fill[col0 + i] : 0.0f) : acc[i] * inv_l;
#pragma unroll
            for (int i = 0; i < VEC; ++i) out[i] = gat_finish<T>(acc[i], relu, out_scale);
            T *drow = D + r * ldd;
            if (VEC > 1 && vec_store && col0 + VEC <= n_feat) {
                *reinterpret_cast<u32x4 *>(drow + col0) = *reinterpret_cast<const u32x4 *>(out);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (col0 + i < n_feat) drow[col0 + i] = out[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Long rows (sgx_plan): a hub row of a power-law graph would keep one lane group busy for
// thousands of dependent steps.  Its edges are cut into the plan's 512-edge tasks; one wavefront
// per task keeps a running (max, sum, weighted row sum) per lane group -- rescaled once per piece
// of LPR edges -- and merges its groups; the tasks of a row are then merged in task order
// (m = max m_t, l = sum l_t e^(m_t - m), row = sum acc_t e^(m_t - m) / l): the same softmax, and
// the same bits from run to run.
// ---------------------------------------------------------------------------------------
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_split_kernel(
    int n_tasks, int n_feat, const int32_t *__restrict__ task_row, const int32_t *__restrict__ task_e0,
    const int32_t *__restrict__ task_e1, const int32_t *__restrict__ col, const T *__restrict__ val,
    const T *__restrict__ Wh, unsigned h_bytes, unsigned ld_bytes, const float *__restrict__ s1,
    const float *__restrict__ s2, float alpha, float *__restrict__ E, float *__restrict__ pacc, int ldp,
    float *__restrict__ pm, float *__restrict__ pl)
{
    constexpr int RPW = 64 / LPR;
    constexpr int TILE = LPR * VEC;
    const int task = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Wh), 0, h_bytes, 0x00020000);
    const int te0 = task_e0[task], te1 = task_e1[task];
    const float si = s1[task_row[task]];

    for (int c0 = 0; c0 < n_feat; c0 += TILE) {
        const int col0 = c0 + sub * VEC;
        const unsigned col_off = col0 < n_feat ? (unsigned)col0 * (unsigned)sizeof(T) : kOOB;
        float m = -INFINITY, l = 0.0f;             // l: this lane's share of the group's sum
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        for (int base = te0 + grp * LPR; base < te1; base += RPW * LPR) {
            const int idx = base + sub;
            int c = 0;
            float x = -INFINITY;
            if (idx < te1) {
                c = col[idx];
                const float xe = leaky(si + s2[c], alpha);
                if (E && c0 == 0) E[idx] = xe;
                if (Elem<T>::to_f32(val[idx]) > 0.0f) x = xe;
            }
            float pmax = x;
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, off));
            if (pmax == -INFINITY) continue;        // no live edge in this piece (uniform across the group)
            const float m_new = fmaxf(m, pmax);
            const float scale = rescale_factor(m, m_new);
            const float p = x == -INFINITY ? 0.0f : expf(x - m_new);
            m = m_new;
            l = l * scale + p;
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] *= scale;
            const int n = te1 - base;
            constexpr int UNR = LPR < 8 ? LPR : 8;
#pragma unroll 1
            for (int t0 = 0; t0 < LPR; t0 += UNR) {
                if (t0 >= n) break;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int t = t0 + u;
                    const int cc = __shfl(c, t, LPR);
                    const float pp = __shfl(p, t, LPR);
                    Gather<T, VEC>::run(acc, pp, rsrc, (t < n && col_off != kOOB) ? (unsigned)cc * ld_bytes + col_off : kOOB);
                }
            }
        }
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) l += __shfl_xor(l, off);        // the group's sum
        // merge the lane groups of the wavefront (fixed tree order)
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
            const float mn = fmaxf(m, m2);
            const float a = rescale_factor(m, mn), b = rescale_factor(m2, mn);
            l = l * a + l2 * b;
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = acc[i] * a + __shfl_xor(acc[i], off) * b;
            m = mn;
        }
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (col0 + i < n_feat) pacc[(int64_t)task * ldp + col0 + i] = acc[i];
            if (sub == 0 && c0 == 0) { pm[task] = m; pl[task] = l; }
        }
    }
}

// (pm, pl are [task][head], row_m / row_l [long row][head]; one head: plain [task] / [long row])
template <typename T>
__global__ __launch_bounds__(kBlock) void gat_split_finalize_kernel(
    int n_long, int n_feat, int n_heads, int f_head, const int32_t *__restrict__ long_row,
    const int32_t *__restrict__ long_first, const float *__restrict__ pacc, int ldp, const float *__restrict__ pm,
    const float *__restrict__ pl, T *__restrict__ D, int64_t ldd, int relu, const float *__restrict__ fill,
    float *__restrict__ row_m, float *__restrict__ row_l, float out_scale)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (int64_t)n_long * n_feat) return;
    const int i = (int)(gid / n_feat), j = (int)(gid % n_feat);
    const int h = j / f_head;
    const int t0 = long_first[i], t1 = long_first[i + 1];
    float m = -INFINITY;
    for (int t = t0; t < t1; ++t) m = fmaxf(m, pm[(int64_t)t * n_heads + h]);
    float l = 0.0f, a = 0.0f;
    for (int t = t0; t < t1; ++t) {
        const float w = rescale_factor(pm[(int64_t)t * n_heads + h], m);
        l += pl[(int64_t)t * n_heads + h] * w;
        a += pacc[(int64_t)t * ldp + j] * w;
    }
    float out = l > 0.0f ? a / l : (fill ? fill[j] : 0.0f);
    D[(int64_t)long_row[i] * ldd + j] = gat_finish<T>(out, relu, out_scale);
    if (j % f_head == 0) { row_m[(int64_t)i * n_heads + h] = m; row_l[(int64_t)i * n_heads + h] = l; }
}

// softmax values of the long rows' edges, once the rows' (max, sum) are known: workgroup (i, y) walks
// every gridDim.y-th 256-edge piece of long row i
template <typename T>
__global__ __launch_bounds__(kBlock) void gat_split_softmax_kernel(
    int n_cols, int n_heads, const int32_t *__restrict__ long_row, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const T *__restrict__ val, const float *__restrict__ s1,
    const float *__restrict__ s2, float alpha, const float *__restrict__ row_m, const float *__restrict__ row_l,
    int filled, float *__restrict__ S)
{
    const int i = blockIdx.x;
    const int row = long_row[i];
    const int e1 = rowptr[row + 1];
    for (int idx = rowptr[row] + blockIdx.y * kBlock + threadIdx.x; idx < e1; idx += gridDim.y * kBlock) {
        const bool pos = Elem<T>::to_f32(val[idx]) > 0.0f;
        const int c = col[idx];
        for (int h = 0; h < n_heads; ++h) {
            const float m = row_m[(int64_t)i * n_heads + h], l = row_l[(int64_t)i * n_heads + h];
            float p = 0.0f;
            if (l > 0.0f) {
                if (pos) p = expf(leaky(s1[(int64_t)row * n_heads + h] + s2[(int64_t)c * n_heads + h], alpha) - m) / l;
            } else if (filled) {
                p = 1.0f / (float)n_cols;
            }
            S[(int64_t)idx * n_heads + h] = p;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Several heads (BASELINE config 5: 8 heads on ogbn-arxiv).  The reference has one head -- its
// `nheads` only widens W (SG.py:1176-1178) -- so this is that single-head formula applied to each
// slice of F_head = n_feat / n_heads columns with its own attention vector
// a_h = attention[h][0 : 2*F_head], outputs concatenated: what n_heads single-head calls on the
// column slices give, in one pass over the edges.  A lane owns VEC columns of one head; it walks
// all edges of its row for that head (scores are 4-byte reads of the per-node, per-head table), so
// no reduction across lanes is needed and each neighbour row is still gathered once.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void gat_scores_heads_kernel(int n_cols, int n_heads, int f_head,
                                                                 const T *__restrict__ Wh, int64_t ldh,
                                                                 const T *__restrict__ att, float *__restrict__ s1,
                                                                 float *__restrict__ s2)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (int64_t)n_cols * n_heads) return;
    const int64_t r = gid / n_heads;
    const int h = (int)(gid - r * n_heads);
    const T *w = Wh + r * ldh + (int64_t)h * f_head;
    const T *a = att + (int64_t)h * 2 * f_head;
    float p1 = 0.0f, p2 = 0.0f;
    for (int i = 0; i < f_head; ++i) {
        const float v = Elem<T>::to_f32(w[i]);
        p1 = __builtin_fmaf(v, Elem<T>::to_f32(a[i]), p1);
        p2 = __builtin_fmaf(v, Elem<T>::to_f32(a[f_head + i]), p2);
    }
    s1[gid] = p1;
    s2[gid] = p2;
}

// TASKS = false: work item = a row (rows over long_threshold edges are left to the tasks).
// TASKS = true:  work item = a task of the plan (an edge chunk of a long row): the lane group leaves the
//                chunk's state -- per head (max, sum) in pm / pl, the unnormalised weighted row in pacc --
//                for gat_split_finalize_kernel; n_rows is then the number of tasks.
template <typename T, int VEC, int LPR, bool TASKS>
__global__ __launch_bounds__(kBlock) void gat_aggregate_heads_kernel(
    int n_rows, int n_cols, int n_feat, int n_heads, int f_head, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const T *__restrict__ val, const T *__restrict__ Wh, unsigned h_bytes,
    unsigned ld_bytes, const float *__restrict__ s1, const float *__restrict__ s2, float alpha,
    T *__restrict__ D, int64_t ldd, int relu, float *__restrict__ E, float *__restrict__ S, int vec_store,
    const float *__restrict__ fill, int share, int long_threshold, const int32_t *__restrict__ task_row,
    const int32_t *__restrict__ task_e0, const int32_t *__restrict__ task_e1, float *__restrict__ pacc, int ldp,
    float *__restrict__ pm, float *__restrict__ pl, float out_scale)
{
    constexpr int RPW = 64 / LPR;
    constexpr int TILE = LPR * VEC;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int64_t w = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RPW + grp;      // work item
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Wh), 0, h_bytes, 0x00020000);
    bool live = w < n_rows;
    int e0 = 0, e1 = 0;
    int64_t r = w;
    if (live) {
        if (TASKS) { r = task_row[w]; e0 = task_e0[w]; e1 = task_e1[w]; }
        else { e0 = rowptr[r]; e1 = rowptr[r + 1]; }
    }
    if (!TASKS && live && long_threshold > 0 && e1 - e0 > long_threshold) { live = false; e1 = e0; }
    const float uniform = 1.0f / (float)n_cols;

    for (int c0 = 0; c0 < n_feat; c0 += TILE) {
        const int col0 = c0 + sub * VEC;
        const bool mine = col0 < n_feat;
        const int h = mine ? col0 / f_head : 0;
        const unsigned col_off = mine ? (unsigned)col0 * (unsigned)sizeof(T) : kOOB;
        const float si = (live && mine) ? s1[r * n_heads + h] : 0.0f;
        const bool writer = mine && (col0 % f_head == 0);          // one lane per (row, head) writes E / S

        // pass 1: the softmax state of this lane's head over all edges of the row.  The `share` lanes that
        // hold one head (a power of two, adjacent) take every share-th edge each and merge their states.
        float m = -INFINITY, l = 0.0f;
        for (int base = e0; base < e1; base += LPR) {
            const int idx = base + sub;
            int c = 0, pos = 0;
            if (idx < e1) { c = col[idx]; pos = Elem<T>::to_f32(val[idx]) > 0.0f; }
            const int n = e1 - base < LPR ? e1 - base : LPR;
            for (int t0 = 0; t0 < n; t0 += share) {
                const int t = t0 + (sub & (share - 1));
                const int cc = __shfl(c, t, LPR);
                const int pp = __shfl(pos, t, LPR);
                if (t < n && pp && mine) softmax_merge(m, l, leaky(si + s2[(int64_t)cc * n_heads + h], alpha), 1.0f);
            }
        }
        for (int off = 1; off < share; off <<= 1) {
            const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
            softmax_merge(m, l, m2, l2);
        }
        const float inv_l = TASKS ? 1.0f : (l > 0.0f ? 1.0f / l : 0.0f);      // a task stays unnormalised
        const bool dead = !TASKS && live && mine && !(l > 0.0f) && fill != nullptr;

        // pass 2: weighted gather
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        for (int base = e0; base < e1; base += LPR) {
            const int idx = base + sub;
            int c = 0, pos = 0;
            if (idx < e1) { c = col[idx]; pos = Elem<T>::to_f32(val[idx]) > 0.0f; }
            const int n = e1 - base < LPR ? e1 - base : LPR;
            for (int t = 0; t < n; ++t) {
                const int cc = __shfl(c, t, LPR);
                const int pp = __shfl(pos, t, LPR);
                float x = 0.0f, p = 0.0f;
                if (mine) {
                    x = leaky(si + s2[(int64_t)cc * n_heads + h], alpha);
                    if (pp) p = expf(x - m) * inv_l;
                }
                if (writer) {
                    const int64_t o = (int64_t)(base + t) * n_heads + h;
                    if (E) E[o] = x;
                    if (!TASKS && S) S[o] = dead ? uniform : p;
                }
                Gather<T, VEC>::run(acc, p, rsrc, mine ? (unsigned)cc * ld_bytes + col_off : kOOB);
            }
        }
        if (TASKS) {
            if (live && mine) {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (col0 + i < n_feat) pacc[w * ldp + col0 + i] = acc[i];
                if (writer) { pm[w * n_heads + h] = m; pl[w * n_heads + h] = l; }
            }
            continue;
        }
        if (live && mine) {
            T out[VEC];
            if (dead) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = (col0 + i < n_feat) ? fill[col0 + i] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) out[i] = gat_finish<T>(acc[i], relu, out_scale);
            T *drow = D + r * ldd;
            if (VEC > 1 && vec_store && col0 + VEC <= n_feat) {
                *reinterpret_cast<u32x4 *>(drow + col0) = *reinterpret_cast<const u32x4 *>(out);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (col0 + i < n_feat) drow[col0 + i] = out[i];
            }
        }
    }
}

// Column means of Wh in two fixed-order stages: slab sums, then the slabs added in order.
constexpr int kMeanSlabs = 512;

template <typename T>
__global__ __launch_bounds__(kBlock) void col_sum_slab_kernel(int n_rows, int n_feat, const T *__restrict__ Wh, int64_t ldh,
                                                             float *__restrict__ partial)
{
    const int rows_per = (n_rows + kMeanSlabs - 1) / kMeanSlabs;
    const int r0 = blockIdx.x * rows_per;
    const int r1 = r0 + rows_per < n_rows ? r0 + rows_per : n_rows;
    for (int j = threadIdx.x; j < n_feat; j += kBlock) {
        float s = 0.0f;
        for (int r = r0; r < r1; ++r) s += Elem<T>::to_f32(Wh[(int64_t)r * ldh + j]);
        partial[(int64_t)blockIdx.x * n_feat + j] = s;
    }
}

__global__ __launch_bounds__(kBlock) void col_mean_finish_kernel(int n_rows, int n_feat, const float *__restrict__ partial,
                                                                float *__restrict__ mean)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_feat) return;
    float s = 0.0f;
    for (int b = 0; b < kMeanSlabs; ++b) s += partial[(int64_t)b * n_feat + j];
    mean[j] = s / (float)n_rows;
}

struct GatArgs {
    int relu, n_rows, n_cols, n_feat, n_heads;
    float alpha;
    const int32_t *rowptr, *col;
    const void *val, *Wh, *att;
    int64_t ldh, ldd;
    unsigned h_bytes, ld_bytes;
    void *D;
    float *E, *S, *s;
    const float *fill;
    int uniform_n;             // the N of the uniform softmax a dead row gets (S = 1/N): n_cols, or all nodes of a partitioned graph
    float out_scale;           // deq_o of the quantised layer on fp32 outputs (0 = off)
    const sgx_plan *plan;      // long rows -> split path
    float *split;              // scratch of the split path, behind the scores / column means
    int vec_ok, vec_store;
    hipStream_t stream;
};

template <typename T, int VEC, int LPR>
int gat_launch_one(const GatArgs &a)
{
    const int rows_per_block = (64 / LPR) * (kBlock / 64);
    const unsigned grid = (unsigned)((a.n_rows + rows_per_block - 1) / rows_per_block);
    const unsigned grid_s = (unsigned)((a.n_cols + rows_per_block - 1) / rows_per_block);
    if (a.n_heads > 1) {
        const int f_head = a.n_feat / a.n_heads;
        // lanes per head; they share the softmax pass when that is a power of two that divides the lane
        // group and no head straddles a column tile (otherwise every lane walks all edges itself)
        const int lanes_per_head = f_head / VEC;
        const bool pow2 = lanes_per_head > 0 && (lanes_per_head & (lanes_per_head - 1)) == 0;
        const int share = (f_head % VEC == 0 && pow2 && lanes_per_head <= LPR) ? lanes_per_head : 1;
        float *h1 = a.s, *h2 = a.s + (size_t)a.n_cols * a.n_heads;
        const int64_t pairs = (int64_t)a.n_cols * a.n_heads;
        hipLaunchKernelGGL((gat_scores_heads_kernel<T>), dim3((unsigned)((pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           a.stream, a.n_cols, a.n_heads, f_head, (const T *)a.Wh, a.ldh, (const T *)a.att, h1, h2);
        SGX_LAUNCH_CHECK();
        const sgx_plan *hp = a.plan;
        const int thr = (hp && hp->n_long > 0) ? hp->long_threshold : 0;
        const int ldp = (int)sgx_align_up((size_t)a.n_feat, 4);
        float *pacc = a.split, *pm = nullptr, *pl = nullptr, *row_m = nullptr, *row_l = nullptr;
        if (thr > 0) {
            pm = pacc + (size_t)hp->n_tasks * ldp;
            pl = pm + (size_t)hp->n_tasks * a.n_heads;
            row_m = pl + (size_t)hp->n_tasks * a.n_heads;
            row_l = row_m + (size_t)hp->n_long * a.n_heads;
        }
        hipLaunchKernelGGL((gat_aggregate_heads_kernel<T, VEC, LPR, false>), dim3(grid), dim3(kBlock), 0, a.stream, a.n_rows,
                           a.uniform_n, a.n_feat, a.n_heads, f_head, a.rowptr, a.col, (const T *)a.val, (const T *)a.Wh,
                           a.h_bytes, a.ld_bytes, h1, h2, a.alpha, (T *)a.D, a.ldd, a.relu, a.E, a.S, a.vec_store, a.fill,
                           share, thr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, a.out_scale);
        SGX_LAUNCH_CHECK();
        if (thr > 0) {
            const unsigned tgrid = (unsigned)((hp->n_tasks + rows_per_block - 1) / rows_per_block);
            hipLaunchKernelGGL((gat_aggregate_heads_kernel<T, VEC, LPR, true>), dim3(tgrid), dim3(kBlock), 0, a.stream,
                               hp->n_tasks, a.uniform_n, a.n_feat, a.n_heads, f_head, a.rowptr, a.col, (const T *)a.val,
                               (const T *)a.Wh, a.h_bytes, a.ld_bytes, h1, h2, a.alpha, (T *)a.D, a.ldd, a.relu, a.E, nullptr,
                               a.vec_store, nullptr, share, 0, hp->task_row, hp->task_e0, hp->task_e1, pacc, ldp, pm, pl,
                               a.out_scale);
            SGX_LAUNCH_CHECK();
            const int64_t total = (int64_t)hp->n_long * a.n_feat;
            hipLaunchKernelGGL((gat_split_finalize_kernel<T>), dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock),
                               0, a.stream, hp->n_long, a.n_feat, a.n_heads, f_head, hp->long_row, hp->long_first, pacc, ldp,
                               pm, pl, (T *)a.D, a.ldd, a.relu, a.fill, row_m, row_l, a.out_scale);
            SGX_LAUNCH_CHECK();
            if (a.S) {
                hipLaunchKernelGGL((gat_split_softmax_kernel<T>), dim3(hp->n_long, 16), dim3(kBlock), 0, a.stream, a.uniform_n,
                                   a.n_heads, hp->long_row, a.rowptr, a.col, (const T *)a.val, h1, h2, a.alpha, row_m, row_l,
                                   a.fill != nullptr, a.S);
                SGX_LAUNCH_CHECK();
            }
        }
        return SGX_OK;
    }
    float *s1 = a.s, *s2 = a.s + a.n_cols;                    // scores of every row of the table
    hipLaunchKernelGGL((gat_scores_kernel<T, VEC, LPR>), dim3(grid_s), dim3(kBlock), 0, a.stream, a.n_cols, a.n_feat,
                       (const T *)a.Wh, a.ldh, (const T *)a.att, s1, s2, a.vec_ok);
    SGX_LAUNCH_CHECK();
    const sgx_plan *p = a.plan;
    const int long_thr = (p && p->n_long > 0) ? p->long_threshold : 0;
    if (long_thr > 0) {
        const int ldp = (int)sgx_align_up((size_t)a.n_feat, 4);
        float *pacc = a.split, *pm = pacc + (size_t)p->n_tasks * ldp, *pl = pm + p->n_tasks;
        float *row_m = pl + p->n_tasks, *row_l = row_m + p->n_long;
        hipLaunchKernelGGL((gat_split_kernel<T, VEC, LPR>), dim3((p->n_tasks + kBlock / 64 - 1) / (kBlock / 64)),
                           dim3(kBlock), 0, a.stream, p->n_tasks, a.n_feat, p->task_row, p->task_e0, p->task_e1, a.col,
                           (const T *)a.val, (const T *)a.Wh, a.h_bytes, a.ld_bytes, s1, s2, a.alpha, a.E, pacc, ldp, pm, pl);
        SGX_LAUNCH_CHECK();
        const int64_t total = (int64_t)p->n_long * a.n_feat;
        hipLaunchKernelGGL((gat_split_finalize_kernel<T>), dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           a.stream, p->n_long, a.n_feat, 1, a.n_feat, p->long_row, p->long_first, pacc, ldp, pm, pl, (T *)a.D,
                           a.ldd, a.relu, a.fill, row_m, row_l, a.out_scale);
        SGX_LAUNCH_CHECK();
        if (a.S) {
            hipLaunchKernelGGL((gat_split_softmax_kernel<T>), dim3(p->n_long, 16), dim3(kBlock), 0, a.stream, a.uniform_n, 1,
                               p->long_row, a.rowptr, a.col, (const T *)a.val, s1, s2, a.alpha, row_m, row_l,
                               a.fill != nullptr, a.S);
            SGX_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL((gat_aggregate_kernel<T, VEC, LPR>), dim3(grid), dim3(kBlock), 0, a.stream, a.n_rows, a.uniform_n, a.n_feat,
                       a.rowptr, a.col, (const T *)a.val, (const T *)a.Wh, a.h_bytes, a.ld_bytes, s1, s2, a.alpha,
                       (T *)a.D, a.ldd, a.relu, a.E, a.S, a.vec_store, a.fill, long_thr, a.out_scale);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

template <typename T, int VEC>
int gat_launch_lpr(const GatArgs &a, int lpr)
{
    switch (lpr) {
    case 1: return gat_launch_one<T, VEC, 1>(a);
    case 2: return gat_launch_one<T, VEC, 2>(a);
    case 4: return gat_launch_one<T, VEC, 4>(a);
    case 8: return gat_launch_one<T, VEC, 8>(a);
    case 16: return gat_launch_one<T, VEC, 16>(a);
    case 32: return gat_launch_one<T, VEC, 32>(a);
    default: return gat_launch_one<T, VEC, 64>(a);
    }
}

}  // namespace

namespace {
size_t base_scratch_floats(int n_cols, int n_feat, int n_heads, int fill_dead_rows)
{
    size_t floats = (size_t)2 * n_cols * n_heads;
    if (fill_dead_rows) floats += (size_t)(kMeanSlabs + 1) * n_feat;
    return sgx_align_up(floats, 64);
}
bool uses_split(const sgx_plan *plan) { return plan && plan->n_long > 0; }
}  // namespace

extern "C" size_t sgx_gat_scratch_bytes(int n_cols, int n_feat, int n_heads, int fill_dead_rows, const sgx_plan *plan)
{
    if (n_cols < 0 || n_feat < 1) return 0;
    if (n_heads < 1) n_heads = 1;
    size_t floats = base_scratch_floats(n_cols, n_feat, n_heads, fill_dead_rows);
    if (uses_split(plan))        // per task: fp32 partial row + (max, sum) per head; per long row: (max, sum) per head
        floats += (size_t)plan->n_tasks * (sgx_align_up((size_t)n_feat, 4) + 2 * (size_t)n_heads) +
                  (size_t)2 * plan->n_long * n_heads;
    return sgx_align_up(floats * sizeof(float), 256);
}

extern "C" int sgx_gat_aggregate(int dtype, int relu, int fill_dead_rows, int n_rows, int n_cols, int n_feat, int n_heads,
                                 float alpha,
                                 const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                                 const void *Wh, int64_t ldh, const void *attention,
                                 void *D, int64_t ldd, float *E, float *S, const sgx_plan *plan, float *s_scratch,
                                 void *stream)
{
    return sgx_gat_aggregate_ep(dtype, relu, fill_dead_rows, n_rows, n_cols, n_feat, n_heads, alpha, rowPtr, columnIndex, values,
                                Wh, ldh, attention, D, ldd, E, S, plan, s_scratch, (hipStream_t)stream, 0.0f);
}

extern "C" int sgx_gat_aggregate_fill(int dtype, int relu, int n_rows, int n_cols, int n_feat, int n_heads, float alpha,
                                      const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                                      const void *Wh, int64_t ldh, const void *attention, void *D, int64_t ldd, float *E, float *S,
                                      const float *fill, int64_t n_nodes, const sgx_plan *plan, float *s_scratch, void *stream)
{
    if (fill && (n_nodes < 1 || n_nodes > 0x7FFFFFFF)) return SGX_ERR_SHAPE;
    return sgx_gat_aggregate_ep(dtype, relu, 0, n_rows, n_cols, n_feat, n_heads, alpha, rowPtr, columnIndex, values, Wh, ldh,
                                attention, D, ldd, E, S, plan, s_scratch, (hipStream_t)stream, 0.0f, fill, (int)n_nodes);
}

namespace {
__global__ __launch_bounds__(kBlock) void col_sum_finish_kernel(int n_feat, const float *__restrict__ partial, float *__restrict__ out)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n_feat) return;
    float s = 0.0f;
    for (int b = 0; b < kMeanSlabs; ++b) s += partial[(int64_t)b * n_feat + j];
    out[j] = s;
}
}  // namespace

extern "C" size_t sgx_col_sums_scratch_bytes(int n_feat) { return n_feat < 1 ? 0 : (size_t)kMeanSlabs * n_feat * sizeof(float); }

extern "C" int sgx_col_sums(int dtype, int n_rows, int n_feat, const void *X, int64_t ldx, float *out, float *scratch, void *stream)
{
    if (n_rows < 0 || n_feat < 1 || ldx < n_feat) return SGX_ERR_SHAPE;
    if (!out || !scratch || (n_rows > 0 && !X)) return SGX_ERR_NULL;
    if (dtype != SGX_F16 && dtype != SGX_F32) return SGX_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SGX_F16)
        hipLaunchKernelGGL(col_sum_slab_kernel<f16>, dim3(kMeanSlabs), dim3(kBlock), 0, s, n_rows, n_feat, (const f16 *)X, ldx, scratch);
    else
        hipLaunchKernelGGL(col_sum_slab_kernel<float>, dim3(kMeanSlabs), dim3(kBlock), 0, s, n_rows, n_feat, (const float *)X, ldx, scratch);
    SGX_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_sum_finish_kernel, dim3((n_feat + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n_feat, scratch, out);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

int sgx_gat_aggregate_ep(int dtype, int relu, int fill_dead_rows, int n_rows, int n_cols, int n_feat, int n_heads, float alpha,
                         const int32_t *rowPtr, const int32_t *columnIndex, const void *values, const void *Wh, int64_t ldh,
                         const void *attention, void *D, int64_t ldd, float *E, float *S, const sgx_plan *plan,
                         float *s_scratch, hipStream_t stream, float out_scale, const float *ext_fill, int ext_n)
{
    if (n_heads < 1) n_heads = 1;
    if (plan && plan->n_rows != n_rows) return SGX_ERR_SHAPE;
    if (n_rows < 0 || n_cols < n_rows || n_feat < 1 || ldh < n_feat || ldd < n_feat) return SGX_ERR_SHAPE;
    if (n_feat % n_heads != 0) return SGX_ERR_SHAPE;
    if (n_rows == 0) return SGX_OK;
    if (!rowPtr || !columnIndex || !values || !Wh || !attention || !D) return SGX_ERR_NULL;
    if (!s_scratch) return SGX_ERR_WORKSPACE;
    if (dtype != SGX_F16 && dtype != SGX_F32) return SGX_ERR_UNSUPPORTED;
    const size_t es = sgx_elem_size(dtype);
    const unsigned long long table_bytes = (unsigned long long)n_cols * (unsigned long long)ldh * es;
    if (table_bytes >= 0xFFFFFFF0ull) return SGX_ERR_UNSUPPORTED;
    GatArgs a;
    a.relu = relu; a.n_rows = n_rows; a.n_cols = n_cols; a.n_feat = n_feat; a.n_heads = n_heads; a.alpha = alpha;
    a.rowptr = rowPtr; a.col = columnIndex; a.val = values; a.Wh = Wh; a.att = attention;
    a.ldh = ldh; a.ldd = ldd; a.h_bytes = (unsigned)table_bytes; a.ld_bytes = (unsigned)(ldh * es);
    a.D = D; a.E = E; a.S = S; a.s = s_scratch; a.stream = stream; a.out_scale = out_scale;
    a.fill = ext_fill;                       // a caller-provided row for dead rows (partitioned graph), or the means below
    a.uniform_n = ext_fill ? ext_n : n_cols;
    a.plan = uses_split(plan) ? plan : nullptr;
    a.split = s_scratch + base_scratch_floats(n_cols, n_feat, n_heads, fill_dead_rows);
    if (fill_dead_rows) {
        float *partial = s_scratch + (size_t)2 * n_cols * n_heads, *mean = partial + (size_t)kMeanSlabs * n_feat;
        if (dtype == SGX_F16)
            hipLaunchKernelGGL(col_sum_slab_kernel<f16>, dim3(kMeanSlabs), dim3(kBlock), 0, a.stream, n_cols, n_feat,
                               (const f16 *)Wh, ldh, partial);
        else
            hipLaunchKernelGGL(col_sum_slab_kernel<float>, dim3(kMeanSlabs), dim3(kBlock), 0, a.stream, n_cols, n_feat,
                               (const float *)Wh, ldh, partial);
        SGX_LAUNCH_CHECK();
        hipLaunchKernelGGL(col_mean_finish_kernel, dim3((n_feat + kBlock - 1) / kBlock), dim3(kBlock), 0, a.stream, n_cols,
                           n_feat, partial, mean);
        SGX_LAUNCH_CHECK();
        a.fill = mean;
    }
    a.vec_ok = ((uintptr_t)Wh % 16 == 0) && ((ldh * es) % 16 == 0);
    a.vec_store = ((uintptr_t)D % 16 == 0) && ((ldd * es) % 16 == 0);
    const int per16 = (int)(16 / es);
    if (n_heads > 1 && (n_feat / n_heads) % per16 != 0) a.vec_ok = 0;      // a lane's 16 bytes must stay inside one head
    if (a.vec_ok) {
        int lpr = sgx_next_pow2((n_feat + per16 - 1) / per16);
        if (lpr > 64) lpr = 64;
        return dtype == SGX_F16 ? gat_launch_lpr<f16, 8>(a, lpr) : gat_launch_lpr<float, 4>(a, lpr);
    }
    int lpr = sgx_next_pow2(n_feat);
    if (lpr > 64) lpr = 64;
    return dtype == SGX_F16 ? gat_launch_lpr<f16, 1>(a, lpr) : gat_launch_lpr<float, 1>(a, lpr);
}
