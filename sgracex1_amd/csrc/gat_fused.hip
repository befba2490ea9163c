// The GAT aggregate in ONE walk over the rows, with no per-edge score gather (round 3; used when the caller wants no E / S):
//     D_i = act( sum_j softmax_j(LeakyReLU(s1_i + Wh_j . a2)) Wh_j )                       (SG.py:634-661)
// The two-stage form (gat.hip) spends its first stage gathering 4 bytes of s2 per stored entry -- one L2 line request
// each, the rate that bounds it -- to weight rows that its second stage gathers anyway.  Here a lane group gathers a
// piece of 8 neighbour rows (16 bytes per lane, the spmm_csr.hip layout), forms their second-half scores FROM the
// gathered rows (8 fmas per lane and row against the attention fragment of the lane's columns, summed over the lanes
// of the head by DPP butterflies: quad_perm, row_half_mirror, row_mirror -- every lane of a head ends with the same
// bits), and folds the piece into a running (max, sum, weighted row) state: the rows are used twice while they sit in
// registers, and the softmax costs no memory traffic at all.  exp is the hardware's v_exp_f32 on (x - max) log2(e) --
// relative error about (2 + |x|) 2^-24 on a weight that only ever reaches the caller inside the 16- or 32-bit sums of D.
// Rows over the plan's cut: a wavefront per task with a state per lane group, merged over the groups, the tasks of a row
// merged in task order by gat_fused_finalize_kernel.  Same schedule as gat_weighted_kernel otherwise: degree order where
// the plan has one (its one-piece tail 64 rows per wavefront), long rows through tasks.
// Rows without a live entry give `fill` (the dense emulation's mean row, SG.py:638-641) or 0.
#include "gat_device.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_swap(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// sum over the HL lanes of a head (HL a power of two, the head's lanes aligned to HL): the same bits in every lane
template <int HL>
__device__ __forceinline__ float head_sum(float v, int lane)
{
    // v comes out of Dot's inline assembly, and a DPP read wants two wait states behind the VALU write of its source: the
    // compiler keeps that distance for instructions it knows, not for ones inside an asm statement
    if constexpr (HL >= 2) asm volatile("s_nop 1" : "+v"(v));
    if constexpr (HL >= 2) v += dpp_swap<0xB1>(v);           // quad_perm [1,0,3,2]
    if constexpr (HL >= 4) v += dpp_swap<0x4E>(v);           // quad_perm [2,3,0,1]
    if constexpr (HL >= 8) v += dpp_swap<0x141>(v);          // row_half_mirror: the other quad of the 8
    if constexpr (HL >= 16) v += dpp_swap<0x140>(v);         // row_mirror: the other half of the 16
    if constexpr (HL >= 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 16) * 4, __builtin_bit_cast(int, v)));
    if constexpr (HL >= 64) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) * 4, __builtin_bit_cast(int, v)));
    return v;
}

__device__ __forceinline__ float exp2_of(float y) { return __builtin_amdgcn_exp2f(y * 1.44269504088896340736f); }

// sum_i frag[i] * (element i of the 16 gathered bytes)
template <typename T, int VEC> struct Dot;
template <> struct Dot<f16, 8> {
    static __device__ __forceinline__ float run(const float *frag, u32x4 raw)
    {
        float d = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned pair = raw[i];
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(pair), "v"(frag[2 * i]));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(pair), "v"(frag[2 * i + 1]));
        }
        return d;
    }
};
template <> struct Dot<float, 4> {
    static __device__ __forceinline__ float run(const float *frag, u32x4 raw)
    {
        union { u32x4 v; float f[4]; } u; u.v = raw;
        float d = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) d = __builtin_fmaf(frag[i], u.f[i], d);
        return d;
    }
};

#ifndef SGX_GAT_FUSED_WAVES
#define SGX_GAT_FUSED_WAVES 1      // (a launch bound of 5 wavefronts per SIMD = 96 registers: no gain measured)
#endif

template <int VEC> struct SoftState {
    float m, l, acc[VEC];
    __device__ __forceinline__ void clear()
    {
        m = -INFINITY;
        l = 0.0f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    }
};

// UNR entries of a piece -- entry T0 + u held by lane T0 + u of the lane group (column C, bit T0 + u of LIVE), N entries in
// the piece -- folded into the state ST of the lane's head: gather, score from the gathered row, rescale, accumulate.
// (Text, not a function: as an inlined function or lambda the same statements cost the row walk 7 more registers -- 102
// instead of 95, a wavefront per SIMD -- and 8 % on a uniform graph.)
// The s_nop: v_exp_f32 is a transcendental, its result needs a wait state before another VALU instruction reads it.  hipcc
// inserts that for its own instructions and does not look inside the inline assembly of Fma<f16>, whose first
// v_fma_mix_f32 then read the weight before it was there (the lane's first column, in the lanes the quarter-rate unit
// serves first: wrong sums, NaNs on partial pieces).
#define SGX_GAT_FOLD(ST, SI, C, LIVE, N, T0)                                                                                       \
    do {                                                                                                                             \
        u32x4 raw_[UNR];                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                                            \
            const int t_ = (T0) + u;                                                                                                 \
            const unsigned cc_ = (unsigned)__shfl((int)(C), t_, LPR);                                                                \
            raw_[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (t_ < (N) && col_off != kOOB) ? cc_ * ld_bytes + col_off : kOOB, 0, 0); \
        }                                                                                                                            \
        float x_[UNR];                                                                                                               \
        float mp_ = (ST).m;                                                                                                          \
        _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                                            \
            const float d_ = head_sum<HL>(Dot<T, VEC>::run(a2, raw_[u]), lane);                                                      \
            const bool lv_ = (T0) + u < (N) && (((LIVE) >> ((T0) + u)) & 1ull);                                                      \
            x_[u] = lv_ ? leaky((SI) + d_, alpha) : -INFINITY;                                                                       \
            mp_ = fmaxf(mp_, x_[u]);                                                                                                 \
        }                                                                                                                            \
        const float ref_ = mp_ == -INFINITY ? 0.0f : mp_; /* (no live entry so far: every exponential below is 0) */                 \
        const float scale_ = exp2_of((ST).m - ref_);                                                                                 \
        float sum_ = (ST).l * scale_;                                                                                                \
        _Pragma("unroll") for (int i = 0; i < VEC; ++i)(ST).acc[i] *= scale_;                                                        \
        _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                                            \
            float p_ = exp2_of(x_[u] - ref_);                                                                                        \
            asm volatile("s_nop 1" : "+v"(p_));                                                                                      \
            sum_ += p_;                                                                                                              \
            Fma<T, VEC>::run((ST).acc, p_, raw_[u]);                                                                                 \
        }                                                                                                                            \
        (ST).l = sum_;                                                                                                               \
        (ST).m = mp_;                                                                                                                \
    } while (0)

template <typename T, int VEC, int LPR, int HL, bool SHORT>
__global__ __launch_bounds__(kBlock, SGX_GAT_FUSED_WAVES) void gat_fused_kernel(
    int n_work, int n_feat, int n_heads, int f_head, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const T *__restrict__ val, const T *__restrict__ Wh, unsigned h_bytes, unsigned ld_bytes, const T *__restrict__ att,
    const float *__restrict__ s1, float alpha, T *__restrict__ D, int64_t ldd, int relu, float out_scale, int long_threshold,
    int vec_store, const int32_t *__restrict__ row_order, int split_blocks, int n_tasks, const int32_t *__restrict__ task_row,
    const int32_t *__restrict__ task_e0, const int32_t *__restrict__ task_e1, float *__restrict__ pacc, int ldp,
    float *__restrict__ pm, float *__restrict__ pl, const float *__restrict__ fill, int n_multi, int short_first)
{
    constexpr int RPW = 64 / LPR;
    constexpr int UNR = LPR < 8 ? LPR : 8;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int col0 = sub * VEC;
    const bool mine = col0 < n_feat;
    const unsigned col_off = mine ? (unsigned)col0 * (unsigned)sizeof(T) : kOOB;
    const int my_head = mine ? col0 / f_head : 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Wh), 0, h_bytes, 0x00020000);
    // the lane's fragment of its head's second attention half (att: per head [a1 (f_head), a2 (f_head)])
    float a2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i)
        a2[i] = (mine && col0 + i < n_feat) ? Elem<T>::to_f32(att[(int64_t)my_head * 2 * f_head + f_head + (col0 - my_head * f_head) + i]) : 0.0f;

    // entries [e0, e1) taken `stride` apart in pieces of LPR
    auto walk = [&](SoftState<VEC> &st, float si, int e0, int e1, int stride) {
        unsigned c_next = 0;
        bool v_next = false;
        auto fetch = [&](int idx, unsigned &c, bool &v) {
            c = 0u;
            v = false;
            if (idx < e1) {
                c = (unsigned)__builtin_nontemporal_load(col + idx);
                v = Elem<T>::to_f32(__builtin_nontemporal_load(val + idx)) > 0.0f;
            }
        };
        fetch(e0 + sub, c_next, v_next);
        for (int base = e0; base < e1; base += stride) {
            const unsigned c = c_next;
            const bool v = v_next;
            fetch(base + stride + sub, c_next, v_next);
            const int n = e1 - base;
            const unsigned long long live = __ballot(v) >> (grp * LPR);             // bit t: entry t of this group's piece is live
#pragma unroll 1
            for (int t0 = 0; t0 < LPR; t0 += UNR) {
                if (t0 >= n) break;
                SGX_GAT_FOLD(st, si, c, live, n, t0);
            }
        }
    };

    auto finish_row = [&](const SoftState<VEC> &st, int64_t r) {
        if (!mine) return;
        T out[VEC];
        const float inv = st.l > 0.0f ? 1.0f / st.l : 0.0f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float o = st.l > 0.0f ? st.acc[i] * inv : ((fill && col0 + i < n_feat) ? fill[col0 + i] : 0.0f);
            out[i] = gat_finish<T>(o, relu, out_scale);
        }
        T *drow = D + r * ldd;
        if (VEC > 1 && vec_store && col0 + VEC <= n_feat) {
            *reinterpret_cast<u32x4 *>(drow + col0) = *reinterpret_cast<const u32x4 *>(out);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (col0 + i < n_feat) drow[col0 + i] = out[i];
        }
    };

    if ((int)blockIdx.x < split_blocks) {
        // a task of a long row: every lane group of the wavefront on its own pieces, then the groups merged
        const int task = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        if (task >= n_tasks) return;
        const int e0 = task_e0[task], e1 = task_e1[task];
        const float si = s1[(int64_t)task_row[task] * n_heads + my_head];
        SoftState<VEC> st;
        st.clear();
        walk(st, si, e0 + grp * LPR, e1, 64);
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(st.m, off), l2 = __shfl_xor(st.l, off);
            const float mn = fmaxf(st.m, m2);
            const float ref = mn == -INFINITY ? 0.0f : mn;
            const float a = exp2_of(st.m - ref), b = exp2_of(m2 - ref);
            st.l = st.l * a + l2 * b;
#pragma unroll
            for (int i = 0; i < VEC; ++i) st.acc[i] = st.acc[i] * a + __shfl_xor(st.acc[i], off) * b;
            st.m = mn;
        }
        if (grp == 0 && mine) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (col0 + i < n_feat) pacc[(int64_t)task * ldp + col0 + i] = st.acc[i];
            if (sub % HL == 0) { pm[(int64_t)task * n_heads + my_head] = st.m; pl[(int64_t)task * n_heads + my_head] = st.l; }
        }
        return;
    }
    if constexpr (SHORT && LPR >= 8) {
        // the degree order's tail of one-piece rows (at most 8 entries), 64 rows per wavefront: row ids, row pointers and the
        // (column, value) pieces of 64 rows are one round trip each instead of one per 64 / LPR rows (spmm_csr.hip's
        // spmm_short_rows, gat_weighted_kernel's SHORT); a row is one fold
        if ((int)blockIdx.x >= short_first) {
            const int64_t i0 = (int64_t)n_multi + ((int64_t)((int)blockIdx.x - short_first) * (kBlock / 64) + (threadIdx.x >> 6)) * 64;
            if (i0 >= n_work) return;
            const int64_t idx = i0 + lane;
            const bool valid = idx < n_work;
            const int rid = row_order[valid ? idx : (int64_t)n_work - 1];
            const int re0 = rowptr[rid];
            const int rdeg = valid ? rowptr[rid + 1] - re0 : 0;                      // at most 8 (the order's last buckets)
            constexpr int CH = 4;                        // rows per lane group and batch of requests
            for (int it0 = 0; it0 < LPR; it0 += CH) {
                unsigned c[CH];
                unsigned live[CH];                       // (a row of the tail has at most 8 entries: 8 bits)
                float si[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int s = (it0 + i) * RPW + grp;
                    const int se0 = __shfl(re0, s), sdeg = __shfl(rdeg, s);
                    const int e = sub < sdeg ? se0 + sub : 0;                        // (unconditional loads: slots past the row read entry 0, masked at use)
                    c[i] = (unsigned)__builtin_nontemporal_load(col + e);
                    const bool lv = sub < sdeg && Elem<T>::to_f32(__builtin_nontemporal_load(val + e)) > 0.0f;
                    live[i] = (unsigned)(__ballot(lv) >> (grp * LPR)) & 0xFFu;
                    si[i] = s1[(int64_t)__shfl(rid, s) * n_heads + my_head];
                }
#pragma unroll
                for (int it = 0; it < CH; ++it) {
                    const int s = (it0 + it) * RPW + grp;
                    const int sdeg = __shfl(rdeg, s);
                    const int64_t rr = __shfl(rid, s);                               // (every lane takes part: a shuffle inside the
                    const bool row_ok = __shfl((int)valid, s) != 0;                  //  branch below would read lanes that are not in it)
                    SoftState<VEC> st;
                    st.clear();
                    SGX_GAT_FOLD(st, si[it], c[it], (unsigned long long)live[it], sdeg, 0);
                    if (row_ok) finish_row(st, rr);
                }
            }
            return;
        }
        n_work = n_multi;                    // the walk below takes the rows of two pieces and more
    }
    const int row_grid = (SHORT ? short_first : (int)gridDim.x) - split_blocks;
    const int64_t wave = (int64_t)((int)blockIdx.x - split_blocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)row_grid * (kBlock / 64);
    for (int64_t r0 = wave * RPW; r0 < n_work; r0 += n_waves * RPW) {
        int64_t r = r0 + grp;
        int e0 = 0, e1 = 0;
        bool live_row = r < n_work;
        if (live_row) {
            if (row_order) r = row_order[r];
            e0 = rowptr[r];
            e1 = rowptr[r + 1];
            if (long_threshold > 0 && e1 - e0 > long_threshold) live_row = false;       // the tasks own it
        }
        if (!live_row) e1 = e0;
        const float si = live_row ? s1[r * n_heads + my_head] : 0.0f;
        SoftState<VEC> st;
        st.clear();
        walk(st, si, e0, e1, LPR);
        if (live_row) finish_row(st, r);
    }
}

// the tasks of a long row merged in task order: m = max m_t, l = sum l_t 2^((m_t - m) log2 e), row = the same sum over the
// partial rows, / l -- a thread per (long row, column), eight tasks' states requested at a time (a hub row of a plan cut
// at 256 entries has hundreds of tasks: one dependent load after the other took 117 us on a 29 M-entry R-MAT graph)
template <typename T>
__global__ __launch_bounds__(kBlock) void gat_fused_finalize_kernel(
    int n_long, int n_feat, int n_heads, int f_head, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_first,
    const float *__restrict__ pacc, int ldp, const float *__restrict__ pm, const float *__restrict__ pl, T *__restrict__ D,
    int64_t ldd, int relu, const float *__restrict__ fill, float out_scale)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (int64_t)n_long * n_feat) return;
    const int i = (int)(gid / n_feat), j = (int)(gid % n_feat);
    const int h = j / f_head;
    const int t0 = long_first[i], t1 = long_first[i + 1];
    float m = -INFINITY;
    int t = t0;
    for (; t + 8 <= t1; t += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pm[(int64_t)(t + u) * n_heads + h];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
    }
    for (; t < t1; ++t) m = fmaxf(m, pm[(int64_t)t * n_heads + h]);
    const float ref = m == -INFINITY ? 0.0f : m;
    float l = 0.0f, a = 0.0f;
    for (t = t0; t + 8 <= t1; t += 8) {
        float vm[8], vl[8], va[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            vm[u] = pm[(int64_t)(t + u) * n_heads + h];
            vl[u] = pl[(int64_t)(t + u) * n_heads + h];
            va[u] = pacc[(int64_t)(t + u) * ldp + j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = exp2_of(vm[u] - ref);
            l += vl[u] * w;
            a += va[u] * w;
        }
    }
    for (; t < t1; ++t) {
        const float w = exp2_of(pm[(int64_t)t * n_heads + h] - ref);
        l += pl[(int64_t)t * n_heads + h] * w;
        a += pacc[(int64_t)t * ldp + j] * w;
    }
    const float out = l > 0.0f ? a / l : (fill ? fill[j] : 0.0f);
    D[(int64_t)long_row[i] * ldd + j] = gat_finish<T>(out, relu, out_scale);
}

template <typename T, int VEC, int LPR, int HL>
int launch_fused(const sgx_gat_fused_args &a)
{
    const int rows_per_block = (64 / LPR) * (kBlock / 64);
    const int split_blocks = (a.n_tasks + kBlock / 64 - 1) / (kBlock / 64);
    // the one-piece tail of a degree order 64 rows per wavefront (as the plain aggregation does, spmm_csr.hip)
    const bool short_tail = LPR >= 8 && a.row_order && !sgx_tune().spmm_no_short_tail && a.n_multi >= 0 && a.n_multi < a.n_work &&
                            a.n_work - a.n_multi >= 4096;
    const int n_multi = short_tail ? a.n_multi : a.n_work;
    const int64_t short_blocks = short_tail ? ((int64_t)(a.n_work - n_multi) + 64 * (kBlock / 64) - 1) / (64 * (kBlock / 64)) : 0;
    int64_t row_blocks = ((int64_t)n_multi + rows_per_block - 1) / rows_per_block;
    if (row_blocks > 256 * 512) row_blocks = 256 * 512;
    const dim3 grid((unsigned)(split_blocks + row_blocks + short_blocks));
    const int short_first = (int)(split_blocks + row_blocks);
#define SGX_GAT_FUSED_LAUNCH(SHORT_)                                                                                             \
    hipLaunchKernelGGL((gat_fused_kernel<T, VEC, LPR, HL, SHORT_>), grid, dim3(kBlock), 0, a.stream, a.n_work, a.n_feat, a.n_heads,  \
                       a.n_feat / a.n_heads, a.rowptr, a.col, (const T *)a.val, (const T *)a.Wh, a.h_bytes, a.ld_bytes,             \
                       (const T *)a.att, a.s1, a.alpha, (T *)a.D, a.ldd, a.relu, a.out_scale, a.long_threshold, a.vec_store,        \
                       a.row_order, split_blocks, a.n_tasks, a.task_row, a.task_e0, a.task_e1, a.pacc, a.ldp, a.pm, a.pl, a.fill,   \
                       n_multi, short_first)
    if (short_tail) SGX_GAT_FUSED_LAUNCH(true);
    else SGX_GAT_FUSED_LAUNCH(false);
#undef SGX_GAT_FUSED_LAUNCH
    SGX_LAUNCH_CHECK();
    if (a.n_tasks > 0) {
        const int64_t total = (int64_t)a.n_long * a.n_feat;
        hipLaunchKernelGGL((gat_fused_finalize_kernel<T>), dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, a.stream,
                           a.n_long, a.n_feat, a.n_heads, a.n_feat / a.n_heads, a.long_row, a.long_first, a.pacc, a.ldp, a.pm, a.pl,
                           (T *)a.D, a.ldd, a.relu, a.fill, a.out_scale);
        SGX_LAUNCH_CHECK();
    }
    return SGX_OK;
}

template <typename T, int VEC, int LPR>
int launch_fused_hl(const sgx_gat_fused_args &a, int hl)
{
    switch (hl) {
    case 1: return launch_fused<T, VEC, LPR, 1>(a);
    case 2: if constexpr (LPR >= 2) return launch_fused<T, VEC, LPR, 2>(a); break;
    case 4: if constexpr (LPR >= 4) return launch_fused<T, VEC, LPR, 4>(a); break;
    case 8: if constexpr (LPR >= 8) return launch_fused<T, VEC, LPR, 8>(a); break;
    case 16: if constexpr (LPR >= 16) return launch_fused<T, VEC, LPR, 16>(a); break;
    case 32: if constexpr (LPR >= 32) return launch_fused<T, VEC, LPR, 32>(a); break;
    case 64: if constexpr (LPR >= 64) return launch_fused<T, VEC, LPR, 64>(a); break;
    }
    return SGX_ERR_UNSUPPORTED;
}

template <typename T, int VEC>
int launch_fused_lpr(const sgx_gat_fused_args &a, int hl)
{
    switch (a.lpr) {
    case 1: return launch_fused_hl<T, VEC, 1>(a, hl);
    case 2: return launch_fused_hl<T, VEC, 2>(a, hl);
    case 4: return launch_fused_hl<T, VEC, 4>(a, hl);
    case 8: return launch_fused_hl<T, VEC, 8>(a, hl);
    case 16: return launch_fused_hl<T, VEC, 16>(a, hl);
    case 32: return launch_fused_hl<T, VEC, 32>(a, hl);
    case 64: return launch_fused_hl<T, VEC, 64>(a, hl);
    }
    return SGX_ERR_UNSUPPORTED;
}

}  // namespace

// one 16-byte tile of lanes covers the row, a head is a power-of-two number of whole lanes
bool sgx_gat_fused_applicable(int dtype, int n_feat, int n_heads, int lpr)
{
    const int vec = dtype == SGX_F16 ? 8 : 4;
    if (n_heads < 1 || n_feat % n_heads != 0 || n_feat > lpr * vec) return false;
    const int f_head = n_feat / n_heads;
    if (f_head % vec != 0) return false;
    const int hl = f_head / vec;
    return (hl & (hl - 1)) == 0 && hl <= lpr;
}

int sgx_gat_fused(const sgx_gat_fused_args &a)
{
    if (!sgx_gat_fused_applicable(a.dtype, a.n_feat, a.n_heads, a.lpr)) return SGX_ERR_UNSUPPORTED;
    const int vec = a.dtype == SGX_F16 ? 8 : 4;
    const int hl = a.n_feat / a.n_heads / vec;
    if (a.dtype == SGX_F16) return launch_fused_lpr<f16, 8>(a, hl);
    return launch_fused_lpr<float, 4>(a, hl);
}
