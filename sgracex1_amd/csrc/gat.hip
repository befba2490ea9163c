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
#include "gat_device.h"

#include <stdlib.h>

#include <type_traits>

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
    // (eight slabs requested at a time, added in slab order: one load in flight per thread made this 512 round trips)
    static_assert(kMeanSlabs % 8 == 0, "");
    for (int b0 = 0; b0 < kMeanSlabs; b0 += 8) {
        float v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) v[b] = partial[(int64_t)(b0 + b) * n_feat + j];
#pragma unroll
        for (int b = 0; b < 8; ++b) s += v[b];
    }
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
    const sgx_plan *plan_any;  // the caller's plan, long rows or not (two-stage form: it tells the stored-entry count)
    float *two_stage;          // scratch of the two-stage form: weights [nnz * heads], then dead-row flags [n_rows bytes]
    const sgx_plan *plan;      // long rows -> split path
    float *split;              // scratch of the split path, behind the scores / column means
    int vec_ok, vec_store;
    int scores_ready;          // s already holds Wh.a1 / Wh.a2 (formed in the epilogue of the X.W kernel that produced Wh)
    hipStream_t stream;
};

template <typename T, int VEC, int LPR>
int gat_two_stage(const GatArgs &a);

template <typename T, int VEC, int LPR>
int gat_launch_one(const GatArgs &a)
{
    if (a.two_stage) return gat_two_stage<T, VEC, LPR>(a);
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

// =======================================================================================
// Two-stage form (used whenever a plan tells the stored-entry count): the softmax weights first, then a plain
// weighted aggregation.
//   stage A (edge work only: 4-byte score gathers, no rows of Wh): per row the maximum and the sum of its live
//     edges' scores, then alpha_e = exp(x_e - m) / l for every stored edge -- the reference's `attention` matrix on
//     the stored entries (SG.py:649-653), which is also the S output.  Rows over the plan's cut go through its
//     tasks (per-task states merged in task order).
//   stage B: D = act(sum_e alpha_e Wh[col_e]) -- the A.H aggregation with fp32 edge weights: the same gather loop,
//     long-row tasks and fixed-order finalize as spmm_csr.hip.  With several heads a lane reads the weight of ITS
//     head for each edge (8 weights per edge lie in one 32-byte piece); each neighbour row is still gathered once.
// Why: the one-pass kernels above chain three dependent memory latencies per piece (column -> score -> rows) and carry a
// softmax state through every step; on the ogbn-arxiv shape they take 0.27 ms (8 heads 0.41) against 0.18 ms for the
// plain aggregate of the same rows, and hub rows multiply that (R-MAT arxiv shape: 8 heads 1.08 ms).  Stage A moves
// ~14 bytes per edge, stage B is the plain aggregate.
// =======================================================================================
constexpr int kAlphaLanes = 8;             // lanes per row in stage A (8 rows per wavefront)

__device__ __forceinline__ void online_add(float &m, float &l, float x)
{
    if (x > m) { l = l * rescale_factor(m, x) + 1.0f; m = x; }
    else l += expf(x - m);
}

// short rows: (max, sum) per head, then the weights; E optional; dead[r] = 1 when the row has no live edge.
// 8 lanes per row split its edges; a row over kCoopEdges8 edges (up to the plan's cut) is taken by the whole wavefront.
constexpr int kCoopEdges8 = 64;

template <typename T, int HB>
__global__ __launch_bounds__(kBlock) void gat_alpha_rows_kernel(
    int n_rows, int n_heads, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const T *__restrict__ val,
    const float *__restrict__ s1, const float *__restrict__ s2, float alpha, int long_threshold,
    float *__restrict__ W, float *__restrict__ E, unsigned char *__restrict__ dead)
{
    constexpr int GL = kAlphaLanes;
    const int lane = threadIdx.x & 63, sub = lane % GL, grp = lane / GL;
    const int64_t r_first = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * (64 / GL);
    const int64_t r = r_first + grp;
    int e0 = 0, e1 = 0;
    bool live_row = r < n_rows;
    if (live_row) { e0 = rowptr[r]; e1 = rowptr[r + 1]; }
    if (live_row && long_threshold > 0 && e1 - e0 > long_threshold) { live_row = false; e1 = e0; }   // the tasks own it
    const int coop_deg = e1 - e0 > kCoopEdges8 ? e1 - e0 : 0;
    const int ce0 = e0;
    if (coop_deg) { live_row = false; e1 = e0; }                                 // taken by the whole wavefront below
    for (int hb0 = 0; hb0 < n_heads; hb0 += HB) {
        float si[HB], m[HB], l[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) { si[h] = live_row ? s1[r * n_heads + hb0 + h] : 0.0f; m[h] = -INFINITY; l[h] = 0.0f; }
        for (int idx = e0 + sub; idx < e1; idx += GL) {
            const int c = col[idx];
            const bool pos = Elem<T>::to_f32(val[idx]) > 0.0f;
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                const float x = leaky(si[h] + s2[(int64_t)c * n_heads + hb0 + h], alpha);
                if (E) E[(int64_t)idx * n_heads + hb0 + h] = x;
                if (pos) online_add(m[h], l[h], x);
            }
        }
#pragma unroll
        for (int off = 1; off < GL; off <<= 1) {
#pragma unroll
            for (int h = 0; h < HB; ++h) softmax_merge(m[h], l[h], __shfl_xor(m[h], off), __shfl_xor(l[h], off));
        }
        for (int idx = e0 + sub; idx < e1; idx += GL) {
            const int c = col[idx];
            const bool pos = Elem<T>::to_f32(val[idx]) > 0.0f;
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                float w = 0.0f;
                if (pos && l[h] > 0.0f) w = expf(leaky(si[h] + s2[(int64_t)c * n_heads + hb0 + h], alpha) - m[h]) / l[h];
                W[(int64_t)idx * n_heads + hb0 + h] = w;
            }
        }
        if (hb0 == 0 && live_row && sub == 0) dead[r] = l[0] > 0.0f ? 0 : 1;      // the mask does not depend on the head
    }
    for (int g = 0; g < 64 / GL; ++g) {
        const int dg = __shfl(coop_deg, g * GL);
        if (dg == 0) continue;                                                     // wave-uniform
        const int ge0 = __shfl(ce0, g * GL), ge1 = ge0 + dg;
        const int64_t gr = r_first + g;
        for (int hb0 = 0; hb0 < n_heads; hb0 += HB) {
            float si[HB], m[HB], l[HB];
#pragma unroll
            for (int h = 0; h < HB; ++h) { si[h] = s1[gr * n_heads + hb0 + h]; m[h] = -INFINITY; l[h] = 0.0f; }
            for (int idx = ge0 + lane; idx < ge1; idx += 64) {
                const int c = col[idx];
                const bool pos = Elem<T>::to_f32(val[idx]) > 0.0f;
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    const float x = leaky(si[h] + s2[(int64_t)c * n_heads + hb0 + h], alpha);
                    if (E) E[(int64_t)idx * n_heads + hb0 + h] = x;
                    if (pos) online_add(m[h], l[h], x);
                }
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
                for (int h = 0; h < HB; ++h) softmax_merge(m[h], l[h], __shfl_xor(m[h], off), __shfl_xor(l[h], off));
            }
            for (int idx = ge0 + lane; idx < ge1; idx += 64) {
                const int c = col[idx];
                const bool pos = Elem<T>::to_f32(val[idx]) > 0.0f;
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    float w = 0.0f;
                    if (pos && l[h] > 0.0f) w = expf(leaky(si[h] + s2[(int64_t)c * n_heads + hb0 + h], alpha) - m[h]) / l[h];
                    W[(int64_t)idx * n_heads + hb0 + h] = w;
                }
            }
            if (hb0 == 0 && lane == 0) dead[gr] = l[0] > 0.0f ? 0 : 1;
        }
    }
}

// One head, rows up to 512 edges (every row when the plan cuts at 256): the row's entries live in registers -- 8 per
// lane -- so a row costs two memory round trips (columns and values, then the scores of those columns) whatever its
// length: 8 lanes per row for rows of up to 64 edges (8 rows per wavefront together), the whole wavefront for one row
// of 65..512 edges at a time.  Out-of-range buffer offsets stand in for branches.  Longer rows (a caller's plan with a
// larger cut) take two walks over memory.
template <typename T, int STRIDE>
__device__ __forceinline__ void alpha_row_in_registers(
    bool active, int e0, int deg, int first, int kmax, float si, float alpha, const __amdgpu_buffer_rsrc_t &col_rsrc,
    const __amdgpu_buffer_rsrc_t &val_rsrc, const __amdgpu_buffer_rsrc_t &s2_rsrc, const __amdgpu_buffer_rsrc_t &w_rsrc,
    const __amdgpu_buffer_rsrc_t &e_rsrc, bool want_e, float &l_out)
{
    float x[8];
    unsigned pos = 0u;
    unsigned c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= kmax) break;                                                      // wave-uniform
        const bool ok = active && first + k * STRIDE < deg;
        const unsigned off = ok ? (unsigned)(e0 + first + k * STRIDE) * 4u : kOOB;
        c[k] = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, off, 0, 0);
        float v;
        if constexpr (sizeof(T) == 2) v = (float)__builtin_bit_cast(T, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(val_rsrc, off >> 1, 0, 0));
        else v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, off, 0, 0));
        pos |= (ok && v > 0.0f) ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= kmax) break;
        const bool ok = active && first + k * STRIDE < deg;
        x[k] = leaky(si + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s2_rsrc, ok ? c[k] * 4u : kOOB, 0, 0)), alpha);
        if (want_e) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[k]), e_rsrc,
                                                          ok ? (unsigned)(e0 + first + k * STRIDE) * 4u : kOOB, 0, 0);
    }
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= kmax) break;
        m = (pos >> k) & 1u ? fmaxf(m, x[k]) : m;
    }
#pragma unroll
    for (int off = 1; off < STRIDE; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
    float l = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= kmax) break;
        x[k] = (pos >> k) & 1u ? exp_weight(x[k] - m) : 0.0f;
        l += x[k];
    }
#pragma unroll
    for (int off = 1; off < STRIDE; off <<= 1) l += __shfl_xor(l, off);
    const float inv_l = l > 0.0f ? 1.0f / l : 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= kmax) break;
        const bool ok = active && first + k * STRIDE < deg;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[k] * inv_l), w_rsrc,
                                              ok ? (unsigned)(e0 + first + k * STRIDE) * 4u : kOOB, 0, 0);
    }
    l_out = l;
}

// (Round 3 tried the several-heads kernel's split here too -- the rows of up to 64 edges in one launch, the longer ones dealt
// out cyclically in a second -- and measured nothing: 1.026 against 1.009 ms on a 29 M-edge R-MAT graph; one launch stays.)
template <typename T>
__global__ __launch_bounds__(kBlock) void gat_alpha_rows_1head_kernel(
    int n_rows, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const T *__restrict__ val,
    unsigned nnz_bytes_col, const float *__restrict__ s1, const float *__restrict__ s2, unsigned s_bytes, float alpha,
    int long_threshold, float *__restrict__ W, float *__restrict__ E, unsigned char *__restrict__ dead)
{
    constexpr int GL = 8;
    const int lane = threadIdx.x & 63, sub = lane % GL, grp = lane / GL;
    const int64_t r_first = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * (64 / GL);
    const int64_t r = r_first + grp;
    auto row_of = [&](int g) -> int64_t { return r_first + g; };
    int e0 = 0, e1 = 0;
    if (r < n_rows) { e0 = rowptr[r]; e1 = rowptr[r + 1]; }
    const bool tasked = long_threshold > 0 && e1 - e0 > long_threshold;           // the tasks own it
    const int deg = tasked ? 0 : e1 - e0;
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(col), 0, nnz_bytes_col, 0x00020000);
    const __amdgpu_buffer_rsrc_t val_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(val), 0, (unsigned)(nnz_bytes_col / 4 * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t s2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s2), 0, s_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(W, 0, nnz_bytes_col, 0x00020000);
    const __amdgpu_buffer_rsrc_t e_rsrc = __builtin_amdgcn_make_buffer_rsrc(E ? E : W, 0, nnz_bytes_col, 0x00020000);

    // rows of up to 64 edges: 8 lanes each, all 8 rows of the wavefront together
    const bool small = r < n_rows && !tasked && deg <= 64;
    int nm = small ? deg : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nm = max(nm, __shfl_xor(nm, off));
    nm = __builtin_amdgcn_readfirstlane(nm);
    {
        float l = 0.0f;
        const float si = small ? s1[r] : 0.0f;
        alpha_row_in_registers<T, GL>(small, e0, deg, sub, (nm + GL - 1) / GL, si, alpha, col_rsrc, val_rsrc, s2_rsrc, w_rsrc, e_rsrc,
                                      E != nullptr, l);
        if (small && sub == 0) dead[r] = l > 0.0f ? 0 : 1;
    }
    // rows of 65..512 edges: the whole wavefront, one row at a time
    const int mid_deg = (r < n_rows && !tasked && deg > 64 && deg <= 512) ? deg : 0;
    const int big_deg = (r < n_rows && !tasked && deg > 512) ? deg : 0;
    for (int g = 0; g < 64 / GL; ++g) {
        const int dg = __shfl(mid_deg, g * GL);
        if (dg == 0) continue;                                                     // wave-uniform
        const int ge0 = __shfl(e0, g * GL);
        float l = 0.0f;
        alpha_row_in_registers<T, 64>(true, ge0, dg, lane, (dg + 63) / 64, s1[row_of(g)], alpha, col_rsrc, val_rsrc, s2_rsrc, w_rsrc,
                                      e_rsrc, E != nullptr, l);
        if (lane == 0) dead[row_of(g)] = l > 0.0f ? 0 : 1;
    }
    // rows over 512 edges that the plan did not cut: two walks over memory, whole wavefront
    for (int g = 0; g < 64 / GL; ++g) {
        const int dg = __shfl(big_deg, g * GL);
        if (dg == 0) continue;
        const int ge0 = __shfl(e0, g * GL), ge1 = ge0 + dg;
        const int64_t gr = row_of(g);
        const float si = s1[gr];
        float m = -INFINITY, l = 0.0f;
        for (int idx = ge0 + lane; idx < ge1; idx += 64) {
            const float xk = leaky(si + s2[col[idx]], alpha);
            if (E) E[idx] = xk;
            if (Elem<T>::to_f32(val[idx]) > 0.0f) m = fmaxf(m, xk);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
        for (int idx = ge0 + lane; idx < ge1; idx += 64)
            if (Elem<T>::to_f32(val[idx]) > 0.0f) l += expf(leaky(si + s2[col[idx]], alpha) - m);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) l += __shfl_xor(l, off);
        for (int idx = ge0 + lane; idx < ge1; idx += 64) {
            float w = 0.0f;
            if (Elem<T>::to_f32(val[idx]) > 0.0f && l > 0.0f) w = expf(leaky(si + s2[col[idx]], alpha) - m) / l;
            W[idx] = w;
        }
        if (lane == 0) dead[gr] = l > 0.0f ? 0 : 1;
    }
}

// short rows, several heads: one lane per (row, head), LH = heads rounded up to a power of two lanes per row.  The
// lanes of a row read the same column indices and one contiguous piece of the score / weight rows (LH x 4 bytes).
// A row of up to kAloneEdges edges is taken in ONE pass with everything in registers: its column indices, then its
// scores, are requested together (out-of-range offsets past the row's end: no branches, no access), so a row costs
// two memory round trips whatever its length; maximum, sum and weights follow from the registers.  Longer rows (up to
// the plan's cut) are taken by the whole wavefront one at a time -- a lane per edge, 8 heads in its registers -- with
// the maximum and the sum folded across lanes separately (a max / an add per shuffle instead of a softmax merge).
constexpr int kAloneEdges = 32;

// PART: 0 = everything in one launch; 1 = only the rows of up to kAloneEdges edges (the register pass: a launch of its own
// needs far fewer registers than the two forms together -- more wavefronts in flight for a kernel that is all latency);
// 2 = only the longer rows (the cooperative passes).
template <typename T, int LH, int PART>
__global__ __launch_bounds__(kBlock) void gat_alpha_rows_heads_kernel(
    int n_rows, int n_heads, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const T *__restrict__ val,
    unsigned nnz_bytes_col, const float *__restrict__ s1, const float *__restrict__ s2, unsigned s_bytes, float alpha,
    int long_threshold, float *__restrict__ W, float *__restrict__ E, unsigned char *__restrict__ dead)
{
    constexpr int RPW = 64 / LH;
    constexpr int KB = kAloneEdges;
    const int lane = threadIdx.x & 63, h = lane % LH, grp = lane / LH;
    // PART 2 deals the rows out cyclically (slot g of wavefront w takes row g W + w, W = all wavefronts): the longer rows of
    // a power-law graph sit next to each other, and taken 8 to a wavefront they would queue up behind one another
    const int64_t gwave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
    const int64_t r = PART == 2 ? (int64_t)grp * n_waves + gwave : gwave * RPW + grp;
    const bool head_ok = h < n_heads;
    int e0 = 0, e1 = 0;
    if (r < n_rows) { e0 = rowptr[r]; e1 = rowptr[r + 1]; }
    const bool tasked = long_threshold > 0 && e1 - e0 > long_threshold;           // the tasks own it
    if (tasked) e1 = e0;
    const int deg = e1 - e0;
    const bool alone = r < n_rows && !tasked && deg <= KB;
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(col), 0, nnz_bytes_col, 0x00020000);
    const __amdgpu_buffer_rsrc_t val_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(val), 0, (unsigned)(nnz_bytes_col / 4 * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t s2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s2), 0, s_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(W, 0, nnz_bytes_col * (unsigned)n_heads, 0x00020000);
    const __amdgpu_buffer_rsrc_t e_rsrc = __builtin_amdgcn_make_buffer_rsrc(E ? E : W, 0, nnz_bytes_col * (unsigned)n_heads, 0x00020000);

    int nm = alone ? deg : 0;                                                    // the longest such row of the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nm = max(nm, __shfl_xor(nm, off));
    nm = __builtin_amdgcn_readfirstlane(nm);
    if (PART != 2 && nm > 0) {
        const float si = (alone && head_ok) ? s1[r * n_heads + h] : 0.0f;
        float x[KB];
        unsigned pos = 0u;
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 8) {
            if (k0 >= nm) break;
            unsigned c[8];
            if constexpr (LH >= 8) {
                // lane j of a row requests entry k0 + j -- one column and one value instruction per 8 entries, 32 contiguous
                // bytes per row, instead of one per entry with the row's lanes all on the same address (every such
                // instruction is 8 rows' lines to look up; the kernel is bound by those look-ups) -- and the row's lanes
                // take the columns from one another; the live flags of the row's 8 entries come out of one ballot
                const bool mine = alone && h < 8 && k0 + h < deg;
                const unsigned off = mine ? (unsigned)(e0 + k0 + h) * 4u : kOOB;
                const unsigned cm = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, off, 0, 0);
                float v;
                if constexpr (sizeof(T) == 2) {
                    const unsigned short hb = __builtin_amdgcn_raw_buffer_load_b16(val_rsrc, off >> 1, 0, 0);
                    v = (float)__builtin_bit_cast(T, hb);
                } else {
                    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, off, 0, 0));
                }
                const unsigned long long live = __ballot(mine && v > 0.0f);
                pos |= ((unsigned)(live >> (grp * LH)) & 0xFFu) << k0;
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = (unsigned)__builtin_amdgcn_ds_bpermute((grp * LH + k) * 4, (int)cm);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool ok = alone && head_ok && k0 + k < deg;
                    const unsigned off = ok ? (unsigned)(e0 + k0 + k) * 4u : kOOB;
                    c[k] = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, off, 0, 0);
                    float v;
                    if constexpr (sizeof(T) == 2) {
                        const unsigned short hb = __builtin_amdgcn_raw_buffer_load_b16(val_rsrc, off >> 1, 0, 0);
                        v = (float)__builtin_bit_cast(T, hb);
                    } else {
                        v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, off, 0, 0));
                    }
                    pos |= (ok && v > 0.0f) ? (1u << (k0 + k)) : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool ok = alone && head_ok && k0 + k < deg;
                const float sj = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                     s2_rsrc, ok ? (c[k] * (unsigned)n_heads + (unsigned)h) * 4u : kOOB, 0, 0));
                x[k0 + k] = leaky(si + sj, alpha);
                if (E) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[k0 + k]), e_rsrc,
                                                             ok ? ((unsigned)(e0 + k0 + k) * (unsigned)n_heads + (unsigned)h) * 4u : kOOB, 0, 0);
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 8) {
            if (k0 >= nm) break;
#pragma unroll
            for (int k = 0; k < 8; ++k) m = (pos >> (k0 + k)) & 1u ? fmaxf(m, x[k0 + k]) : m;
        }
        float l = 0.0f;
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 8) {
            if (k0 >= nm) break;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float p = (pos >> (k0 + k)) & 1u ? exp_weight(x[k0 + k] - m) : 0.0f;
                x[k0 + k] = p;
                l += p;
            }
        }
        const float inv_l = l > 0.0f ? 1.0f / l : 0.0f;
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 8) {
            if (k0 >= nm) break;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool ok = alone && head_ok && k0 + k < deg;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[k0 + k] * inv_l), w_rsrc,
                                                      ok ? ((unsigned)(e0 + k0 + k) * (unsigned)n_heads + (unsigned)h) * 4u : kOOB, 0, 0);
            }
        }
        if (alone && h == 0) dead[r] = l > 0.0f ? 0 : 1;
    } else if (PART != 2 && alone && h == 0) {
        dead[r] = 1;             // a wavefront whose rows are all empty: they are rows without a live edge all the same
    }
    if (PART == 1) return;

    // the longer rows of this wavefront, one at a time with every lane: a lane per edge, the heads (8 at a time) in its
    // registers; maximum first, then the sum of exp(x - max), then the weights
    const int coop_deg = (!tasked && deg > KB) ? deg : 0;
    const bool vec8 = n_heads % 8 == 0 && (reinterpret_cast<uintptr_t>(s2) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(E)) % 16 == 0;
    for (int g = 0; g < RPW; ++g) {
        const int dg = __shfl(coop_deg, g * LH);
        if (dg == 0) continue;                                                     // wave-uniform
        const int ge0 = __shfl(e0, g * LH), ge1 = ge0 + dg;
        const int64_t gr = PART == 2 ? (int64_t)g * n_waves + gwave : gwave * RPW + g;
        if (dg <= 256) {
            // up to 4 edges per lane: the row's scores (8 heads at a time) stay in registers -- columns and values
            // requested together, then the score rows, then maximum, sum and weights without another read
            const int kmax = (dg + 63) / 64;
            unsigned c[4];
            unsigned pv = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= kmax) break;
                const bool ok = lane + 64 * k < dg;
                const unsigned off = ok ? (unsigned)(ge0 + lane + 64 * k) * 4u : kOOB;
                c[k] = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, off, 0, 0);
                float v;
                if constexpr (sizeof(T) == 2) v = (float)__builtin_bit_cast(T, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(val_rsrc, off >> 1, 0, 0));
                else v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, off, 0, 0));
                pv |= (ok && v > 0.0f) ? (1u << k) : 0u;
            }
            for (int hb0 = 0; hb0 < n_heads; hb0 += 8) {
                float si[8], m[8], l[8], x[4][8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { si[q] = hb0 + q < n_heads ? s1[gr * n_heads + hb0 + q] : 0.0f; m[q] = -INFINITY; l[q] = 0.0f; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= kmax) break;
                    const bool ok = lane + 64 * k < dg;
                    load_scores8(s2, ok ? (int64_t)c[k] : 0, n_heads, hb0, vec8, x[k]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[k][q] = leaky(si[q] + x[k][q], alpha);
                    if (E && ok) store8(E, ge0 + lane + 64 * k, n_heads, hb0, vec8, x[k]);
                    if ((pv >> k) & 1u) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], x[k][q]);
                    }
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], __shfl_xor(m[q], off));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= kmax) break;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        x[k][q] = (pv >> k) & 1u ? exp_weight(x[k][q] - m[q]) : 0.0f;
                        l[q] += x[k][q];
                    }
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) l[q] += __shfl_xor(l[q], off);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= kmax) break;
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[k][q] = l[q] > 0.0f ? x[k][q] / l[q] : 0.0f;
                    if (lane + 64 * k < dg) store8(W, ge0 + lane + 64 * k, n_heads, hb0, vec8, x[k]);
                }
                if (hb0 == 0 && lane == 0) dead[gr] = l[0] > 0.0f ? 0 : 1;
            }
            continue;
        }
        for (int hb0 = 0; hb0 < n_heads; hb0 += 8) {
            float si[8], m[8], l[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { si[k] = hb0 + k < n_heads ? s1[gr * n_heads + hb0 + k] : 0.0f; m[k] = -INFINITY; l[k] = 0.0f; }
            for (int idx = ge0 + lane; idx < ge1; idx += 64) {
                const int c = col[idx];
                const bool pv = Elem<T>::to_f32(val[idx]) > 0.0f;
                float sc[8];
                load_scores8(s2, c, n_heads, hb0, vec8, sc);
#pragma unroll
                for (int k = 0; k < 8; ++k) sc[k] = leaky(si[k] + sc[k], alpha);
                if (E) store8(E, idx, n_heads, hb0, vec8, sc);
                if (pv) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], sc[k]);
                }
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], __shfl_xor(m[k], off));
            }
            for (int idx = ge0 + lane; idx < ge1; idx += 64) {
                const int c = col[idx];
                if (Elem<T>::to_f32(val[idx]) > 0.0f) {
                    float sc[8];
                    load_scores8(s2, c, n_heads, hb0, vec8, sc);
#pragma unroll
                    for (int k = 0; k < 8; ++k) l[k] += expf(leaky(si[k] + sc[k], alpha) - m[k]);
                }
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) l[k] += __shfl_xor(l[k], off);
            }
            for (int idx = ge0 + lane; idx < ge1; idx += 64) {
                const int c = col[idx];
                const bool pv = Elem<T>::to_f32(val[idx]) > 0.0f;
                float sc[8];
                load_scores8(s2, c, n_heads, hb0, vec8, sc);
#pragma unroll
                for (int k = 0; k < 8; ++k) sc[k] = (pv && l[k] > 0.0f) ? expf(leaky(si[k] + sc[k], alpha) - m[k]) / l[k] : 0.0f;
                store8(W, idx, n_heads, hb0, vec8, sc);
            }
            if (hb0 == 0 && lane == 0) dead[gr] = l[0] > 0.0f ? 0 : 1;
        }
    }
}

// Scores Wh.a1, Wh.a2 per (node, head) with the rows read 16 bytes per lane: a lane keeps the attention fragments of
// its columns in registers and walks rows grid-stride; the lanes of a head (F_head / VEC of them, a power of two) fold
// their partial dot products with shuffles.  One tile of LPR x VEC columns covers the row.
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(kBlock) void gat_scores_rows_kernel(int n_rows, int n_feat, int n_heads, int f_head,
                                                                const T *__restrict__ Wh, int64_t ldh,
                                                                const T *__restrict__ att, float *__restrict__ s1,
                                                                float *__restrict__ s2)
{
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR, grp = lane / LPR;
    const int col0 = sub * VEC;
    const bool mine = col0 < n_feat;
    const int h = mine ? col0 / f_head : 0, j0 = mine ? col0 - h * f_head : 0;
    const int lanes_per_head = f_head / VEC;
    // the lane's fragments of the attention vectors: one 16-byte load each where they are aligned (element loads were
    // 2 VEC memory instructions per wavefront ahead of its 4 row loads -- with one wavefront per 8 rows, most of the kernel)
    float a1[VEC], a2[VEC];
    const T *p1 = att + (int64_t)h * 2 * f_head + j0, *p2 = p1 + f_head;
    if (((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) % 16) == 0 && VEC * sizeof(T) == 16) {
        union { u32x4 v; T e[VEC]; } u1, u2;
        u1.v = mine ? *reinterpret_cast<const u32x4 *>(p1) : u32x4{0u, 0u, 0u, 0u};
        u2.v = mine ? *reinterpret_cast<const u32x4 *>(p2) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            a1[i] = mine ? Elem<T>::to_f32(u1.e[i]) : 0.0f;
            a2[i] = mine ? Elem<T>::to_f32(u2.e[i]) : 0.0f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            a1[i] = mine ? Elem<T>::to_f32(p1[i]) : 0.0f;
            a2[i] = mine ? Elem<T>::to_f32(p2[i]) : 0.0f;
        }
    }
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
    // kU row groups per pass, their loads requested before the first is reduced (one at a time was a round trip to
    // memory per 512 bytes: 35 us for the 87 MB of the ogbn-arxiv shape)
    constexpr int kU = 4;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Wh), 0, (unsigned)(((int64_t)(n_rows - 1) * ldh + n_feat) * (int64_t)sizeof(T)), 0x00020000);
    for (int64_t r0 = wave * (RPW * kU); r0 < n_rows; r0 += n_waves * (RPW * kU)) {
        u32x4 raw[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = r0 + u * RPW + grp;
            raw[u] = __builtin_amdgcn_raw_buffer_load_b128(
                rsrc, (r < n_rows && mine) ? (unsigned)((r * ldh + col0) * (int64_t)sizeof(T)) : kOOB, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int64_t r = r0 + u * RPW + grp;
            union { u32x4 v; T e[VEC]; } x;
            x.v = raw[u];
            float p1 = 0.0f, p2 = 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float xv = Elem<T>::to_f32(x.e[i]);
                p1 = __builtin_fmaf(xv, a1[i], p1);
                p2 = __builtin_fmaf(xv, a2[i], p2);
            }
            for (int off = 1; off < lanes_per_head; off <<= 1) {
                p1 += __shfl_xor(p1, off);
                p2 += __shfl_xor(p2, off);
            }
            if (r < n_rows && mine && (sub % lanes_per_head) == 0) {
                s1[r * n_heads + h] = p1;
                s2[r * n_heads + h] = p2;
            }
        }
    }
}

// long rows, step 1: one wavefront per task -- its (max, sum) per head, E of its entries, and the scores themselves left
// in W (-inf for a masked entry), so that step 3 streams them back instead of gathering a second time.  256 entries a
// pass: columns and values requested together, then their score rows (one entry per lane and pass was a chain of two
// memory round trips per 64 entries: 100 us for the 18 M long-row entries of a 29 M-entry R-MAT graph).
template <typename T, int HB>
__global__ __launch_bounds__(kBlock) void gat_alpha_task_stats_kernel(
    int n_tasks, int n_heads, const int32_t *__restrict__ task_row, const int32_t *__restrict__ task_e0,
    const int32_t *__restrict__ task_e1, const int32_t *__restrict__ col, const T *__restrict__ val,
    const float *__restrict__ s1, const float *__restrict__ s2, float alpha, float *__restrict__ E, float *__restrict__ W,
    float *__restrict__ pm, float *__restrict__ pl)
{
    constexpr int U = 4;
    const int task = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    const int lane = threadIdx.x & 63;
    const int64_t r = task_row[task];
    const int te0 = task_e0[task], te1 = task_e1[task];
    if (te1 <= te0) return;
    const bool vec = HB >= 4 && (reinterpret_cast<uintptr_t>(s1) | reinterpret_cast<uintptr_t>(s2) | reinterpret_cast<uintptr_t>(E) |
                                 reinterpret_cast<uintptr_t>(W)) % 16 == 0;
    for (int hb0 = 0; hb0 < n_heads; hb0 += HB) {
        float si[HB], m[HB], l[HB];
        load_scores<HB>(s1, r, n_heads, hb0, vec, si);
#pragma unroll
        for (int h = 0; h < HB; ++h) { m[h] = -INFINITY; l[h] = 0.0f; }
        for (int i0 = te0; i0 < te1; i0 += 64 * U) {
            int c[U];
            unsigned live = 0u;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + 64 * u + lane, at = min(idx, te1 - 1);
                c[u] = col[at];
                live |= (idx < te1 && Elem<T>::to_f32(val[at]) > 0.0f) ? (1u << u) : 0u;
            }
            float x[U][HB];
#pragma unroll
            for (int u = 0; u < U; ++u) load_scores<HB>(s2, (int64_t)c[u], n_heads, hb0, vec, x[u]);
            float mk[HB];
#pragma unroll
            for (int h = 0; h < HB; ++h) mk[h] = m[h];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = i0 + 64 * u + lane;
#pragma unroll
                for (int h = 0; h < HB; ++h) x[u][h] = leaky(si[h] + x[u][h], alpha);
                if (E && idx < te1) store_heads<HB>(E, idx, n_heads, hb0, vec, x[u]);
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    x[u][h] = (live >> u) & 1u ? x[u][h] : -INFINITY;
                    mk[h] = fmaxf(mk[h], x[u][h]);
                }
                if (idx < te1) store_heads<HB>(W, idx, n_heads, hb0, vec, x[u]);
            }
#pragma unroll
            for (int h = 0; h < HB; ++h) {
                if (mk[h] == -INFINITY) continue;
                float sum = l[h] * rescale_factor(m[h], mk[h]);
#pragma unroll
                for (int u = 0; u < U; ++u) sum += exp_weight(x[u][h] - mk[h]);           // (a masked entry: exp(-inf) = 0)
                l[h] = sum;
                m[h] = mk[h];
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int h = 0; h < HB; ++h) softmax_merge(m[h], l[h], __shfl_xor(m[h], off), __shfl_xor(l[h], off));
        }
        if (lane == 0) {
#pragma unroll
            for (int h = 0; h < HB; ++h) { pm[(int64_t)task * n_heads + hb0 + h] = m[h]; pl[(int64_t)task * n_heads + hb0 + h] = l[h]; }
        }
    }
}

// long rows, step 2: one wavefront per (long row, head) -- its tasks' states merged, 64 at a time in a fixed lane order,
// and the row's state written back over every one of them, so that step 3 can run per TASK and read pm / pl at its own
// index (a thread per row and head walking up to hundreds of tasks one after the other took 38 us)
__global__ __launch_bounds__(kBlock) void gat_alpha_long_merge_kernel(
    int n_long, int n_heads, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_first,
    float *__restrict__ pm, float *__restrict__ pl, float *__restrict__ row_m, float *__restrict__ row_l,
    unsigned char *__restrict__ dead)
{
    const int64_t pair = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (pair >= (int64_t)n_long * n_heads) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(pair / n_heads), h = (int)(pair % n_heads);
    const int t0 = long_first[i], t_end = long_first[i + 1];
    float m = -INFINITY, l = 0.0f;
    for (int t = t0 + lane; t < t_end; t += 64) softmax_merge(m, l, pm[(int64_t)t * n_heads + h], pl[(int64_t)t * n_heads + h]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) softmax_merge(m, l, __shfl_xor(m, off), __shfl_xor(l, off));
    m = __shfl(m, 0);                           // (one lane's result for all: the merge is not symmetric in its last bits)
    l = __shfl(l, 0);
    for (int t = t0 + lane; t < t_end; t += 64) { pm[(int64_t)t * n_heads + h] = m; pl[(int64_t)t * n_heads + h] = l; }
    if (lane == 0) {
        row_m[pair] = m;
        row_l[pair] = l;
        if (h == 0) dead[long_row[i]] = l > 0.0f ? 0 : 1;
    }
}

// long rows, step 3: the weights of their entries from the scores step 1 left in W, one wavefront per TASK (the row's
// merged state lies at the task's own index after step 2): a streaming pass, no gathers
__global__ __launch_bounds__(kBlock) void gat_alpha_long_write_kernel(
    int n_tasks, int n_heads, const int32_t *__restrict__ task_e0, const int32_t *__restrict__ task_e1,
    const float *__restrict__ pm, const float *__restrict__ pl, float *__restrict__ W)
{
    constexpr int U = 4;
    const int task = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    const int lane = threadIdx.x & 63;
    const int64_t f0 = (int64_t)task_e0[task] * n_heads, f1 = (int64_t)task_e1[task] * n_heads;
    if (f1 <= f0) return;
    const float *tm = pm + (int64_t)task * n_heads, *tl = pl + (int64_t)task * n_heads;
    float *Wt = W + f0;
    const int n = (int)(f1 - f0);                                  // (a task's scores: entries x heads, well under 2^31)
    const bool pow2 = (n_heads & (n_heads - 1)) == 0;
    for (int j0 = 0; j0 < n; j0 += 64 * U) {
        float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = Wt[min(j0 + 64 * u + lane, n - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + 64 * u + lane;
            const int h = pow2 ? (j & (n_heads - 1)) : j % n_heads;        // (the task begins at head 0 of an entry)
            const float m = tm[h], l = tl[h];
            if (j < n) Wt[j] = l > 0.0f ? exp_weight(x[u] - m) * (1.0f / l) : 0.0f;     // (the expression of gat_weighted_kernel's from_scores)
        }
    }
}

// Stage B: D[r][:] = act(sum_e W[e][head of the column] * Wh[col[e]][:]).  Workgroups [0, split_blocks) sum the plan's
// tasks (all lane groups of a wavefront on one task, fp32 partial rows), the others one row per lane group.  HEADS = 0:
// one weight per edge, loaded with the column by the edge's lane and shuffled; HEADS = 1: every lane loads the weight
// of its own head for each edge through a buffer resource (out of range past the row's end: 0, no access).
// SHORT: the degree order's tail of one-step rows (at most 8 edges) 64 rows per wavefront, as spmm_short_rows does for the
// plain aggregation (spmm_csr.hip: every link of row id -> row pointers -> (column, weight) -> gather is one round trip
// for 64 rows; the same fma chain per output element, hence the same bits); workgroups from short_first on.
template <typename T, int VEC, int LPR, int HEADS, bool SHORT>
__global__ __launch_bounds__(kBlock) void gat_weighted_kernel(
    int n_work, int n_feat, int n_heads, int f_head, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ W, unsigned w_bytes, const T *__restrict__ Wh, unsigned h_bytes, unsigned ld_bytes,
    T *__restrict__ D, int64_t ldd, int relu, float out_scale, int long_threshold, int vec_store,
    const int32_t *__restrict__ row_order, int split_blocks, int n_tasks, const int32_t *__restrict__ task_e0,
    const int32_t *__restrict__ task_e1, float *__restrict__ partial, int ldp, int n_multi, int short_first,
    const float *__restrict__ task_m, const float *__restrict__ task_l)
{
    constexpr int RPW = 64 / LPR;
    constexpr int TILE = LPR * VEC;
    constexpr int UNR = LPR < 8 ? LPR : 8;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Wh), 0, h_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, w_bytes, 0x00020000);

    // the sums of edges [e0, e1) taken `stride` apart in pieces of LPR, for the lane's VEC columns at col0
    // from_scores (a task of a long row whose W still holds stage A's scores, -inf for a masked entry): the weight is
    // exp(score - m) * (1 / l) with the row's merged state, formed here instead of by a pass of its own over W
    auto accumulate = [&](auto from_scores, float *acc, int e0, int e1, int stride, int col0, const float *state_m, const float *state_l) {
        constexpr bool XF = decltype(from_scores)::value;
        const unsigned col_off = col0 < n_feat ? (unsigned)col0 * (unsigned)sizeof(T) : kOOB;
        const unsigned my_head = HEADS ? (unsigned)((col0 < n_feat ? col0 : 0) / f_head) : 0u;
        float xm = 0.0f, xinv = 0.0f;
        if constexpr (XF) {
            const float l = state_l[my_head];
            xm = state_m[my_head];
            xinv = l > 0.0f ? 1.0f / l : 0.0f;
        }
        unsigned c_next = 0;
        float a_next = 0.0f;
        auto fetch = [&](int idx, unsigned &c, float &a) {
            c = 0u;
            a = 0.0f;
            if (idx < e1) {
                c = (unsigned)__builtin_nontemporal_load(col + idx);
                if (!HEADS) {
                    a = __builtin_nontemporal_load(W + idx);
                    if constexpr (XF) a = xinv > 0.0f ? exp_weight(a - xm) * xinv : 0.0f;
                }
            }
        };
        fetch(e0 + sub, c_next, a_next);
        for (int base = e0; base < e1; base += stride) {
            const unsigned c = c_next;
            const float a = a_next;
            fetch(base + stride + sub, c_next, a_next);
            const int n = e1 - base;
#pragma unroll 1
            for (int t0 = 0; t0 < LPR; t0 += UNR) {
                if (t0 >= n) break;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int t = t0 + u;
                    const unsigned cc = (unsigned)__shfl((int)c, t, LPR);
                    float aa;
                    if (HEADS) {
                        aa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                 wsrc, t < n ? ((unsigned)(base + t) * (unsigned)n_heads + my_head) * 4u : kOOB, 0, 0));
                        if constexpr (XF) aa = (t < n && xinv > 0.0f) ? exp_weight(aa - xm) * xinv : 0.0f;
                    } else {
                        aa = __shfl(a, t, LPR);
                    }
                    Gather<T, VEC>::run(acc, aa, rsrc, (t < n && col_off != kOOB) ? cc * ld_bytes + col_off : kOOB);
                }
            }
        }
    };

    if ((int)blockIdx.x < split_blocks) {
        const int task = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        if (task >= n_tasks) return;
        const int e0 = task_e0[task], e1 = task_e1[task];
        for (int c0 = 0; c0 < n_feat; c0 += TILE) {
            const int col0 = c0 + sub * VEC;
            float acc[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
            if (task_m) accumulate(std::true_type{}, acc, e0 + grp * LPR, e1, 64, col0, task_m + (int64_t)task * n_heads, task_l + (int64_t)task * n_heads);
            else accumulate(std::false_type{}, acc, e0 + grp * LPR, e1, 64, col0, nullptr, nullptr);
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off);
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    if (col0 + i < n_feat) partial[(int64_t)task * ldp + col0 + i] = acc[i];
            }
        }
        return;
    }
    if constexpr (SHORT && LPR >= 8) {
        if ((int)blockIdx.x >= short_first) {
            constexpr int ITER = LPR;
            const int64_t i0 = (int64_t)n_multi + ((int64_t)((int)blockIdx.x - short_first) * (kBlock / 64) + (threadIdx.x >> 6)) * 64;
            if (i0 >= n_work) return;
            const int64_t idx = i0 + lane;
            const bool valid = idx < n_work;
            const int rid = row_order[valid ? idx : (int64_t)n_work - 1];
            const int re0 = rowptr[rid];
            const int rdeg = valid ? rowptr[rid + 1] - re0 : 0;              // at most 8 (the order's last buckets)
            constexpr int CH = 8;                        // iterations per batch of (column, weight) requests
            const int col0 = sub * VEC;
            const unsigned col_off = col0 < n_feat ? (unsigned)col0 * (unsigned)sizeof(T) : kOOB;
            const unsigned my_head = HEADS ? (unsigned)((col0 < n_feat ? col0 : 0) / f_head) : 0u;
            for (int it0 = 0; it0 < ITER; it0 += CH) {
            unsigned c[CH];
            float a[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int s = (it0 + i) * RPW + grp;
                const int se0 = __shfl(re0, s), sdeg = __shfl(rdeg, s);
                const int e = sub < sdeg ? se0 + sub : 0;                    // (unconditional loads: slots past the row read entry 0, masked at use)
                c[i] = (unsigned)__builtin_nontemporal_load(col + e);
                a[i] = HEADS ? 0.0f : __builtin_nontemporal_load(W + e);
            }
#pragma unroll
            for (int it = 0; it < CH; ++it) {
                const int s = (it0 + it) * RPW + grp;
                const int se0 = __shfl(re0, s), sdeg = __shfl(rdeg, s);
                const int64_t rr = __shfl(rid, s);
                const bool live = __shfl((int)valid, s) != 0;
                float acc[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned cc = (unsigned)__shfl((int)c[it], t, LPR);
                    float aa;
                    if (HEADS)
                        aa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                 wsrc, t < sdeg ? ((unsigned)(se0 + t) * (unsigned)n_heads + my_head) * 4u : kOOB, 0, 0));
                    else
                        aa = __shfl(a[it], t, LPR);
                    Gather<T, VEC>::run(acc, aa, rsrc, (t < sdeg && col_off != kOOB) ? cc * ld_bytes + col_off : kOOB);
                }
                if (live && col0 < n_feat) {
                    T out[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) out[i] = gat_finish<T>(acc[i], relu, out_scale);
                    T *drow = D + rr * ldd;
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
            return;
        }
        n_work = n_multi;                    // the walk below takes the rows of two steps and more
    }
    const int row_grid = (SHORT ? short_first : (int)gridDim.x) - split_blocks;
    const int64_t wave = (int64_t)(blockIdx.x - split_blocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)row_grid * (kBlock / 64);
    for (int64_t r0 = wave * RPW; r0 < n_work; r0 += n_waves * RPW) {
        int64_t r = r0 + grp;
        int e0 = 0, e1 = 0;
        bool live = r < n_work;
        if (live) {
            if (row_order) r = row_order[r];
            e0 = rowptr[r];
            e1 = rowptr[r + 1];
            if (long_threshold > 0 && e1 - e0 > long_threshold) live = false;
        }
        if (!live) e1 = e0;
        for (int c0 = 0; c0 < n_feat; c0 += TILE) {
            const int col0 = c0 + sub * VEC;
            float acc[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
            accumulate(std::false_type{}, acc, e0, e1, LPR, col0, nullptr, nullptr);
            if (live && col0 < n_feat) {
                T out[VEC];
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
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gat_weighted_finalize_kernel(
    int n_long, int n_feat, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_first,
    const float *__restrict__ partial, int ldp, T *__restrict__ D, int64_t ldd, int relu, float out_scale)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (int64_t)n_long * n_feat) return;
    const int l = (int)(gid / n_feat), j = (int)(gid % n_feat);
    float s = 0.0f;
    // task order, eight loads in flight at a time (the GAT plan cuts at 256 edges: a hub row of a power-law graph has
    // hundreds of tasks, and one dependent load after the other made this kernel 89 us on a 29 M-edge graph)
    const int t_end = long_first[l + 1];
    int t = long_first[l];
    for (; t + 8 <= t_end; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(t + u) * ldp + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; t < t_end; ++t) s += partial[(int64_t)t * ldp + j];
    D[(int64_t)long_row[l] * ldd + j] = gat_finish<T>(s, relu, out_scale);
}

// rows without a live edge: the row `fill` (the mean of all rows of Wh, SG.py:638-641) and S = 1/N on their edges
template <typename T>
__global__ __launch_bounds__(kBlock) void gat_dead_fill_kernel(
    int n_rows, int n_feat, int n_heads, const unsigned char *__restrict__ dead, const int32_t *__restrict__ rowptr,
    const float *__restrict__ fill, float uniform, T *__restrict__ D, int64_t ldd, int relu, float out_scale,
    float *__restrict__ S)
{
    const int64_t r = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (r >= n_rows || !dead[r]) return;
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < n_feat; j += 64) D[r * ldd + j] = gat_finish<T>(fill[j], relu, out_scale);
    if (S)
        for (int64_t i = (int64_t)rowptr[r] * n_heads + lane; i < (int64_t)rowptr[r + 1] * n_heads; i += 64) S[i] = uniform;
}

// Stage A's short rows in entry order (gat_scan.hip) instead of 8 rows per wavefront?
// Measured (tools/gat_probe.py, round 3): on a power-law graph one head's short rows take 146 us in entry order against
// 240 us as 8-row wavefronts (29 M-entry R-MAT, 11 M entries in rows up to 256); with 8 heads the log-step scans cost
// six times the vector instructions of the per-row lanes' running maximum and sum and lose on every shape (ogbn-arxiv
// shape 113 against 69 us), and on a uniform graph the 8-row wavefronts have no idle lanes to win back.  So: one or two
// heads, on a plan whose short rows are scheduled in degree order -- the plan's own sign of rows of very unequal length.
bool gat_scan_wanted(const sgx_plan *p, int n_heads)
{
    const int mode = sgx_tune().gat_scan;          // tuning override: 0 = never, 1 = by shape, 2 = wherever the plan allows
    if (mode == 0 || !sgx_gat_scan_applicable(p)) return false;
    (void)n_heads;
    return mode == 2 || p->row_order != nullptr;
}

template <typename T, int HB>
int gat_alpha_stage(const GatArgs &a, const float *s1, const float *s2, float *W, unsigned char *dead, float *pm, float *pl,
                    float *row_m, float *row_l)
{
    const sgx_plan *p = a.plan_any;
    const int thr = p->n_long > 0 ? p->long_threshold : 0;
    const int rows_per_block = (64 / kAlphaLanes) * (kBlock / 64);
    if (gat_scan_wanted(p, a.n_heads)) {     // the rows up to the cut in entry order (gat_scan.hip); the longer ones below, as ever
        const int rc = sgx_gat_alpha_scan(sizeof(T) == 2 ? SGX_F16 : SGX_F32, a.n_rows, a.n_heads, p, a.rowptr, a.col, a.val, s1, s2,
                                          a.alpha, W, a.E, a.fill ? dead : nullptr, a.stream);
        if (rc != SGX_OK) return rc;
    } else if (a.n_heads == 1) {
        const dim3 grid1((unsigned)((a.n_rows + rows_per_block - 1) / rows_per_block));
        hipLaunchKernelGGL((gat_alpha_rows_1head_kernel<T>), grid1, dim3(kBlock), 0, a.stream, a.n_rows, a.rowptr, a.col,
                           (const T *)a.val, (unsigned)(p->nnz * 4), s1, s2, (unsigned)((size_t)a.n_cols * 4), a.alpha, thr, W, a.E, dead);
    } else if (a.n_heads <= 64) {
        int lh = sgx_next_pow2(a.n_heads);
        const int rpb = (64 / lh) * (kBlock / 64);
        const dim3 grid((unsigned)((a.n_rows + rpb - 1) / rpb));
#define SGX_GAT_LH(L)                                                                                                     \
    case L:                                                                                                               \
        hipLaunchKernelGGL((gat_alpha_rows_heads_kernel<T, L, 1>), grid, dim3(kBlock), 0, a.stream, a.n_rows, a.n_heads,      \
                           a.rowptr, a.col, (const T *)a.val, (unsigned)(p->nnz * 4), s1, s2,                                \
                           (unsigned)((size_t)a.n_cols * a.n_heads * 4), a.alpha, thr, W, a.E, dead);                        \
        if (p->max_degree > kAloneEdges) /* (the second launch serves only rows above that: none on e.g. a uniform graph) */ \
            hipLaunchKernelGGL((gat_alpha_rows_heads_kernel<T, L, 2>), grid, dim3(kBlock), 0, a.stream, a.n_rows, a.n_heads,  \
                               a.rowptr, a.col, (const T *)a.val, (unsigned)(p->nnz * 4), s1, s2,                            \
                               (unsigned)((size_t)a.n_cols * a.n_heads * 4), a.alpha, thr, W, a.E, dead);                    \
        break;
        switch (lh) {
            SGX_GAT_LH(2) SGX_GAT_LH(4) SGX_GAT_LH(8) SGX_GAT_LH(16) SGX_GAT_LH(32) SGX_GAT_LH(64)
        }
#undef SGX_GAT_LH
    } else {
        hipLaunchKernelGGL((gat_alpha_rows_kernel<T, HB>), dim3((unsigned)((a.n_rows + rows_per_block - 1) / rows_per_block)),
                           dim3(kBlock), 0, a.stream, a.n_rows, a.n_heads, a.rowptr, a.col, (const T *)a.val, s1, s2, a.alpha, thr,
                           W, a.E, dead);
    }
    SGX_LAUNCH_CHECK();
    if (thr > 0) {
        hipLaunchKernelGGL((gat_alpha_task_stats_kernel<T, HB>), dim3((p->n_tasks + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock),
                           0, a.stream, p->n_tasks, a.n_heads, p->task_row, p->task_e0, p->task_e1, a.col, (const T *)a.val, s1,
                           s2, a.alpha, a.E, W, pm, pl);
        SGX_LAUNCH_CHECK();
        const int64_t pairs = (int64_t)p->n_long * a.n_heads;
        hipLaunchKernelGGL(gat_alpha_long_merge_kernel, dim3((unsigned)((pairs + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                           a.stream, p->n_long, a.n_heads, p->long_row, p->long_first, pm, pl, row_m, row_l, dead);
        SGX_LAUNCH_CHECK();
        if (a.S) {                             // the caller wants the weights themselves; otherwise stage B forms them from the scores
            hipLaunchKernelGGL(gat_alpha_long_write_kernel, dim3((p->n_tasks + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0,
                               a.stream, p->n_tasks, a.n_heads, p->task_e0, p->task_e1, pm, pl, W);
            SGX_LAUNCH_CHECK();
        }
    }
    return SGX_OK;
}

template <typename T, int VEC, int LPR>
int gat_two_stage(const GatArgs &a)
{
    const sgx_plan *p = a.plan_any;
    const int rows_per_block = (64 / LPR) * (kBlock / 64);
    const unsigned grid_s = (unsigned)((a.n_cols + rows_per_block - 1) / rows_per_block);
    const int f_head = a.n_feat / a.n_heads;
    float *s1 = a.s, *s2 = a.s + (size_t)a.n_cols * a.n_heads;
    const int lanes_per_head = VEC > 1 ? f_head / VEC : 0;
    if (a.scores_ready) {
        // (nothing to launch)
    } else if (VEC > 1 && a.vec_ok && a.n_feat <= LPR * VEC && f_head % VEC == 0 && lanes_per_head >= 1 && lanes_per_head <= LPR &&
        (lanes_per_head & (lanes_per_head - 1)) == 0 &&
        (unsigned long long)a.n_cols * (unsigned long long)a.ldh * sizeof(T) < 0xFFF00000ull) {       // (32-bit buffer offsets)
        int64_t blocks = ((int64_t)a.n_cols + rows_per_block - 1) / rows_per_block;
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL((gat_scores_rows_kernel<T, VEC, LPR>), dim3((unsigned)blocks), dim3(kBlock), 0, a.stream, a.n_cols,
                           a.n_feat, a.n_heads, f_head, (const T *)a.Wh, a.ldh, (const T *)a.att, s1, s2);
    } else if (a.n_heads > 1) {
        const int64_t pairs = (int64_t)a.n_cols * a.n_heads;
        hipLaunchKernelGGL((gat_scores_heads_kernel<T>), dim3((unsigned)((pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           a.stream, a.n_cols, a.n_heads, f_head, (const T *)a.Wh, a.ldh, (const T *)a.att, s1, s2);
    } else {
        hipLaunchKernelGGL((gat_scores_kernel<T, VEC, LPR>), dim3(grid_s), dim3(kBlock), 0, a.stream, a.n_cols, a.n_feat,
                           (const T *)a.Wh, a.ldh, (const T *)a.att, s1, s2, a.vec_ok);
    }
    SGX_LAUNCH_CHECK();
    // scratch behind the split area: the weights (unless the caller's S takes them) and the dead-row flags
    const int thr = p->n_long > 0 ? p->long_threshold : 0;
    const int ldp = (int)sgx_align_up((size_t)a.n_feat, 4);
    float *pacc = a.split, *pm = pacc + (size_t)p->n_tasks * ldp, *pl = pm + (size_t)p->n_tasks * a.n_heads;
    float *row_m = pl + (size_t)p->n_tasks * a.n_heads, *row_l = row_m + (size_t)p->n_long * a.n_heads;
    const int dtype_code = sizeof(T) == 2 ? SGX_F16 : SGX_F32;
    // One walk pays while a head spans few lanes -- every lane of a head forms the piece's 8 scores and exponentials itself, and
    // a head wider than a DPP row half sums its dots through LDS permutes (measured, tools/gat_probe.py: 8 heads x 32 columns
    // 0.286 -> 0.234 ms on the arxiv shape, one head of 64 columns 0.80 -> 0.70 ms and 8 heads 1.94 -> 0.74 ms on a 29 M-entry
    // R-MAT graph, 2 heads x 128 columns = 16 lanes 0.252 -> 0.230 ms; one head of 256 columns = 32 lanes 0.235 -> 0.254 ms: the
    // two stages stay).  SGX_GAT_FUSED = 0 / 2: never / wherever it applies.
    const int lanes_of_a_head = (a.n_feat / a.n_heads) / (VEC > 0 ? VEC : 1);
    const bool fused_pays = sgx_tune().gat_fused == 2 || (sgx_tune().gat_fused == 1 && lanes_of_a_head <= 16);
    if (!a.E && !a.S && fused_pays && VEC == Elem<T>::kVec && a.vec_ok && sgx_gat_fused_applicable(dtype_code, a.n_feat, a.n_heads, LPR)) {
        // no side outputs wanted: one walk over the rows, the neighbours' scores formed from the rows it gathers (gat_fused.hip)
        sgx_gat_fused_args f{};
        f.dtype = dtype_code; f.lpr = LPR; f.relu = a.relu; f.n_feat = a.n_feat; f.n_heads = a.n_heads;
        f.n_work = p->row_order ? p->n_ordered : a.n_rows;
        f.long_threshold = thr; f.vec_store = a.vec_store; f.n_tasks = thr > 0 ? p->n_tasks : 0; f.ldp = ldp;
        f.alpha = a.alpha; f.out_scale = a.out_scale;
        f.rowptr = a.rowptr; f.col = a.col; f.row_order = p->row_order;
        f.task_row = p->task_row; f.task_e0 = p->task_e0; f.task_e1 = p->task_e1;
        f.long_row = p->long_row; f.long_first = p->long_first; f.n_long = p->n_long; f.n_multi = p->n_multi;
        f.val = a.val; f.Wh = a.Wh; f.att = a.att; f.h_bytes = a.h_bytes; f.ld_bytes = a.ld_bytes;
        f.s1 = s1; f.fill = a.fill; f.D = a.D; f.ldd = a.ldd; f.pacc = pacc; f.pm = pm; f.pl = pl; f.stream = a.stream;
        return sgx_gat_fused(f);
    }
    float *W = a.S ? a.S : a.two_stage;
    unsigned char *dead = reinterpret_cast<unsigned char *>(a.two_stage + (size_t)p->nnz * a.n_heads);
    int rc;
    if (a.n_heads % 8 == 0) rc = gat_alpha_stage<T, 8>(a, s1, s2, W, dead, pm, pl, row_m, row_l);
    else if (a.n_heads % 4 == 0) rc = gat_alpha_stage<T, 4>(a, s1, s2, W, dead, pm, pl, row_m, row_l);
    else if (a.n_heads % 2 == 0) rc = gat_alpha_stage<T, 2>(a, s1, s2, W, dead, pm, pl, row_m, row_l);
    else rc = gat_alpha_stage<T, 1>(a, s1, s2, W, dead, pm, pl, row_m, row_l);
    if (rc != SGX_OK) return rc;

    const int32_t *order = p->row_order;
    const int n_work = order ? p->n_ordered : a.n_rows;
    const int n_tasks = thr > 0 ? p->n_tasks : 0;
    const int split_blocks = (n_tasks + kBlock / 64 - 1) / (kBlock / 64);
    // the one-step tail of a degree order 64 rows per wavefront (as the plain aggregation does, spmm_csr.hip)
    const bool short_tail = LPR >= 8 && order && !sgx_tune().spmm_no_short_tail && p->n_multi >= 0 && p->n_multi < n_work &&
                            a.n_feat <= LPR * VEC && n_work - p->n_multi >= 4096;
    const int n_multi = short_tail ? p->n_multi : n_work;
    const int64_t short_blocks = short_tail ? ((int64_t)(n_work - n_multi) + 64 * (kBlock / 64) - 1) / (64 * (kBlock / 64)) : 0;
    int64_t row_blocks = ((int64_t)n_multi + rows_per_block - 1) / rows_per_block;
    if (row_blocks > 256 * 512) row_blocks = 256 * 512;
    const unsigned w_bytes = (unsigned)((size_t)p->nnz * a.n_heads * sizeof(float));
    const dim3 grid_b((unsigned)(split_blocks + row_blocks + short_blocks));
    const int short_first = (int)(split_blocks + row_blocks);
#define SGX_GAT_WEIGHTED(HEADS_, SHORT_)                                                                                          \
    hipLaunchKernelGGL((gat_weighted_kernel<T, VEC, LPR, HEADS_, SHORT_>), grid_b, dim3(kBlock), 0, a.stream, n_work, a.n_feat,     \
                       a.n_heads, f_head, a.rowptr, a.col, W, w_bytes, (const T *)a.Wh, a.h_bytes, a.ld_bytes, (T *)a.D, a.ldd,      \
                       a.relu, a.out_scale, thr, a.vec_store, order, split_blocks, n_tasks, n_tasks ? p->task_e0 : nullptr,         \
                       n_tasks ? p->task_e1 : nullptr, pacc, ldp, n_multi, short_first,                                            \
                       (n_tasks && !a.S) ? pm : nullptr, (n_tasks && !a.S) ? pl : nullptr)
    if (a.n_heads > 1) {
        if (short_tail) SGX_GAT_WEIGHTED(1, true);
        else SGX_GAT_WEIGHTED(1, false);
    } else {
        if (short_tail) SGX_GAT_WEIGHTED(0, true);
        else SGX_GAT_WEIGHTED(0, false);
    }
#undef SGX_GAT_WEIGHTED
    SGX_LAUNCH_CHECK();
    if (n_tasks > 0) {
        const int64_t total = (int64_t)p->n_long * a.n_feat;
        hipLaunchKernelGGL((gat_weighted_finalize_kernel<T>), dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           a.stream, p->n_long, a.n_feat, p->long_row, p->long_first, pacc, ldp, (T *)a.D, a.ldd, a.relu,
                           a.out_scale);
        SGX_LAUNCH_CHECK();
    }
    if (a.fill) {
        hipLaunchKernelGGL((gat_dead_fill_kernel<T>), dim3((unsigned)((a.n_rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                           a.stream, a.n_rows, a.n_feat, a.n_heads, dead, a.rowptr, a.fill, 1.0f / (float)a.uniform_n, (T *)a.D,
                           a.ldd, a.relu, a.out_scale, a.S);
        SGX_LAUNCH_CHECK();
    }
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
// heads of a whole number of 64-column groups: the X.W kernel of a layer may leave the scores as one partial per
// (row, group) for gat_scores_combine_kernel to add up (sgx_gat_score_partials)
bool score_partials_possible(int n_feat, int n_heads) { return n_heads >= 1 && n_feat % n_heads == 0 && (n_feat / n_heads) % 64 == 0; }
size_t score_partial_floats(int n_cols, int n_feat, int n_heads)
{
    return score_partials_possible(n_feat, n_heads) ? (size_t)2 * n_cols * (n_feat / 64) : 0;
}
size_t base_scratch_floats(int n_cols, int n_feat, int n_heads, int fill_dead_rows)
{
    size_t floats = (size_t)2 * n_cols * n_heads;
    if (fill_dead_rows) floats += (size_t)(kMeanSlabs + 1) * n_feat;
    floats += score_partial_floats(n_cols, n_feat, n_heads);
    return sgx_align_up(floats, 64);
}
bool uses_split(const sgx_plan *plan) { return plan && plan->n_long > 0; }
size_t split_floats(const sgx_plan *plan, int n_feat, int n_heads)
{
    if (!uses_split(plan)) return 0;     // per task: fp32 partial row + (max, sum) per head; per long row: (max, sum) per head
    return sgx_align_up((size_t)plan->n_tasks * (sgx_align_up((size_t)n_feat, 4) + 2 * (size_t)n_heads) +
                        (size_t)2 * plan->n_long * n_heads, 64);
}
// the two-stage form needs the stored-entry count on the host (a plan carries it) and 32-bit offsets into the weights
bool two_stage_ok(const sgx_plan *plan, int n_heads)
{
    if (sgx_tune().gat_one_pass) return false;           // tuning override: the one-pass kernels
    return plan && plan->nnz > 0 && (unsigned long long)plan->nnz * (unsigned long long)n_heads < (1ull << 30) &&
           (unsigned long long)plan->n_rows * (unsigned long long)n_heads < (1ull << 30);
}
size_t two_stage_floats(const sgx_plan *plan, int n_heads)
{
    return two_stage_ok(plan, n_heads) ? (size_t)plan->nnz * n_heads + ((size_t)plan->n_rows + 3) / 4 + 16 : 0;
}
}  // namespace

// Whether a layer may have its X.W kernel form the attention scores in its epilogue (xw_dense.hip) and hand them to
// sgx_gat_aggregate_ep as scores_ready: the two-stage form must be the one that runs, and a head must be the 32 columns a
// lane quad of the MFMA tile holds.
bool sgx_gat_scores_fusable(int dtype, int n_feat, int n_heads, const sgx_plan *plan)
{
    if (n_heads < 1) n_heads = 1;
    if (dtype != SGX_F16 || n_feat % n_heads != 0 || n_feat % 64 != 0 || !two_stage_ok(plan, n_heads) || sgx_tune().gat_no_fused_scores)
        return false;
    const int f_head = n_feat / n_heads;
    return f_head == 32 || f_head % 64 == 0;         // a head = a lane quad's pair of tiles, or whole 64-column groups of a wavefront
}

// where the per-group partial scores of that form go: behind the scores and the column means of the scratch
float *sgx_gat_score_partials(float *s_scratch, int n_cols, int n_feat, int n_heads, int fill_dead_rows)
{
    if (n_heads < 1) n_heads = 1;
    if (!score_partials_possible(n_feat, n_heads)) return nullptr;
    return s_scratch + (size_t)2 * n_cols * n_heads + (fill_dead_rows ? (size_t)(kMeanSlabs + 1) * n_feat : 0);
}

namespace {
// s[r][h] = the groups of head h added in the order of the scores kernel's lane tree: (g0 + g1) + (g2 + g3) ...
__global__ __launch_bounds__(kBlock) void gat_scores_combine_kernel(int64_t n_pairs, int n_heads, int groups_per_head, int n_groups,
                                                                    const float *__restrict__ sp1, const float *__restrict__ sp2,
                                                                    float *__restrict__ s1, float *__restrict__ s2)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_pairs) return;
    const int64_t r = i / n_heads;
    const int h = (int)(i % n_heads);
    const float *p1 = sp1 + r * n_groups + (int64_t)h * groups_per_head, *p2 = sp2 + r * n_groups + (int64_t)h * groups_per_head;
    float a1[8], a2[8];                              // (up to 512 columns per head)
    for (int g = 0; g < groups_per_head; ++g) { a1[g] = p1[g]; a2[g] = p2[g]; }
    for (int width = 1; width < groups_per_head; width <<= 1)
        for (int g = 0; g + width < groups_per_head; g += 2 * width) { a1[g] += a1[g + width]; a2[g] += a2[g + width]; }
    s1[i] = a1[0];
    s2[i] = a2[0];
}
}  // namespace

// the partial scores of sgx_xw_dense_scores (heads of 64 columns and more) added up into s_scratch's s1 / s2
int sgx_gat_scores_combine(float *s_scratch, int n_cols, int n_feat, int n_heads, int fill_dead_rows, hipStream_t stream)
{
    if (n_heads < 1) n_heads = 1;
    float *sp1 = sgx_gat_score_partials(s_scratch, n_cols, n_feat, n_heads, fill_dead_rows);
    const int n_groups = n_feat / 64, gph = n_groups / n_heads;
    if (!sp1 || gph < 1 || gph > 8 || (gph & (gph - 1))) return SGX_ERR_UNSUPPORTED;
    const int64_t pairs = (int64_t)n_cols * n_heads;
    hipLaunchKernelGGL(gat_scores_combine_kernel, dim3((unsigned)((pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, pairs, n_heads,
                       gph, n_groups, sp1, sp1 + (size_t)n_cols * n_groups, s_scratch, s_scratch + (size_t)n_cols * n_heads);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

extern "C" size_t sgx_gat_scratch_bytes(int n_cols, int n_feat, int n_heads, int fill_dead_rows, const sgx_plan *plan)
{
    if (n_cols < 0 || n_feat < 1) return 0;
    if (n_heads < 1) n_heads = 1;
    size_t floats = base_scratch_floats(n_cols, n_feat, n_heads, fill_dead_rows);
    floats += split_floats(plan, n_feat, n_heads) + two_stage_floats(plan, n_heads);
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
    // (eight slabs requested at a time, added in slab order: one load in flight per thread made this 512 round trips)
    static_assert(kMeanSlabs % 8 == 0, "");
    for (int b0 = 0; b0 < kMeanSlabs; b0 += 8) {
        float v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) v[b] = partial[(int64_t)(b0 + b) * n_feat + j];
#pragma unroll
        for (int b = 0; b < 8; ++b) s += v[b];
    }
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
                         float *s_scratch, hipStream_t stream, float out_scale, const float *ext_fill, int ext_n, int scores_ready)
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
    a.scores_ready = scores_ready && two_stage_ok(plan, n_heads);      // (only the two-stage form takes them; see sgx_gat_scores_fusable)
    if (scores_ready && !a.scores_ready) return SGX_ERR_UNSUPPORTED;
    a.fill = ext_fill;                       // a caller-provided row for dead rows (partitioned graph), or the means below
    a.uniform_n = ext_fill ? ext_n : n_cols;
    a.plan = uses_split(plan) ? plan : nullptr;
    a.split = s_scratch + base_scratch_floats(n_cols, n_feat, n_heads, fill_dead_rows);
    a.plan_any = plan;
    a.two_stage = two_stage_ok(plan, n_heads) ? a.split + split_floats(plan, n_feat, n_heads) : nullptr;
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
