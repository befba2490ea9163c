// CSR aggregation  D = act(A . H)  for gfx950 -- the A.H stage of the reference
// (loop_adj / compute2 / dsp_kernel_wrapper_adj / writec: K.cpp:3339, :2483, :1778, :713)
// and, with H := W, the sparse-feature X.W stage (compute1 in gemm_mode 0, K.cpp:1960-2078).
//
// Mapping to the hardware (HBM-bound: ~2*F flops per 6+2F bytes):
//   * sblock path.  A wavefront is cut into groups of LPR lanes; LPR lanes x 16 bytes cover
//     one row of H (F=64 fp16 -> 8 lanes, one 128-byte line).  Each group owns ONE row of A,
//     so a wavefront works on 64/LPR rows at once -- the reference's SPMM_BLOCK row grouping
//     (K.cpp:826-845) with the running-nnz interval test replaced by lane ownership.  Per
//     step a group reads LPR (column, value) pairs with one coalesced load each, broadcasts
//     them with lane shuffles, and issues LPR independent 16-byte gathers (buffer loads: an
//     edge past the end of the row gets an out-of-range offset, which returns 0 without
//     touching memory), i.e. 64 gathers of 16 bytes in flight per wavefront.
//   * split path.  Rows longer than plan->long_threshold are cut into chunks; one wavefront
//     sums one chunk with all 64/LPR groups on the same row, the groups are folded with
//     lane shuffles, the fp32 partial rows are added in chunk order by a small second kernel.
//   * products and sums in fp32, one rounding to the storage type; ReLU fused into the store.
#include "sgx_device.h"

// Tuning knobs of the sblock kernel (defaults = the measured best, tools/sweep_spmm.py):
#ifndef SGX_SPMM_MINWAVES
#define SGX_SPMM_MINWAVES 1          // 2nd __launch_bounds__ argument: min wavefronts per SIMD
#endif
#ifndef SGX_SPMM_NT_STORE
#define SGX_SPMM_NT_STORE 0          // non-temporal stores of D
#endif
#include <stdlib.h>

#ifndef SGX_SPMM_ADJ_STYLE
#define SGX_SPMM_ADJ_STYLE 0   // A.H: gather + accumulate edge by edge (rolling window of loads)
#endif
#ifndef SGX_SPMM_FEA_STYLE
#define SGX_SPMM_FEA_STYLE 1   // X.W: all gathers of a step, then the arithmetic
#endif
#ifndef SGX_SPMM_PIECES
#define SGX_SPMM_PIECES 1            // pieces of LPR edges per loop iteration (gathers in flight x PIECES)
#endif
#ifndef SGX_SPMM_BLOCKS_PER_CU
#define SGX_SPMM_BLOCKS_PER_CU 512   // grid cap (2.18 vs 2.21 ms at 64 on S-100M); rows beyond it are grid-strided
#endif

namespace {

constexpr double kShortRowDegree = 5.0;      // choose_cpl: mean degree below which lane groups are halved

// Sum of  values[e] * H[columnIndex[e]][this lane's columns]  over e in [e0, e1) stepping `stride`
// edges between this group's LPR-edge pieces.  A lane owns CPL chunks of VEC consecutive columns
// (CPL x 16 bytes of the row); chunk_off[j] = byte offset of chunk j inside a row of H, or kOOB when
// that chunk lies beyond n_feat.
//   LPR x CPL x 16 bytes = one row of H.  CPL = 1 spends all lanes of a group on the row (8 edges per
//   step at F = 64 fp16); CPL = 2 / 4 halves / quarters the group, so a wavefront keeps 2x / 4x as
//   many rows in flight and wastes fewer gather slots on short rows -- chosen from the matrix's mean
//   degree (sgx_plan).  Gathers in flight per lane stay at 8 either way.
// BIG = the table is 4 GiB or larger: buffer offsets are 32 bit, so the gathers become 64-bit
// global loads under a per-edge predicate (the all-gathered H of an 8-GPU run crosses this size).
template <typename T, int VEC, int LPR, int CPL, bool BIG, int STYLE>
__device__ __forceinline__ void accumulate_edges(float *acc, int e0, int e1, int stride, int sub,
                                                 const int32_t *__restrict__ col, const T *__restrict__ val,
                                                 __amdgpu_buffer_rsrc_t rsrc, const T *__restrict__ table,
                                                 unsigned ld_bytes, const unsigned *chunk_off)
{
    // PIECES pieces of LPR edges are handled per iteration: their (column, value) pairs arrive with
    // one load each (requested one iteration ahead) and all their gathers are issued before the first
    // is consumed -- SGX_SPMM_PIECES * min(LPR, 8/CPL) * CPL gathers in flight per lane.
    constexpr int PIECES = SGX_SPMM_PIECES;
    // What travels from the lane that loaded an edge to the lanes that gather for it: the byte offset
    // of the neighbour's row (column index x row pitch, multiplied once by the loading lane instead
    // of once per receiving lane) -- kOOBRow for an edge past the end of the row, which makes every
    // gather of that slot an out-of-range buffer access (returns 0, no memory traffic) without any
    // per-gather predicate.  Tables of 4 GiB and more (BIG) keep the column index and a predicate.
    // (the prefetched column index stays raw until the iteration that uses it: multiplying at fetch time
    // would put the wait for the load right behind the load)
    unsigned c_next[PIECES];
    T a_next[PIECES];
    constexpr unsigned kNoEdge = 0xFFFFFFFFu;
    auto fetch = [&](int idx, unsigned &c, T &a) {
        // a slot past the end of the row: the sentinel wherever a predicate reads it (STYLE 1, and every style on the
        // 64-bit pointer path, which has no range check: column 0 would be loaded and 0 x inf = NaN)
        c = (STYLE == 0 && !BIG) ? 0u : kNoEdge;
        a = (T)0;
        if (idx < e1) {
            c = (unsigned)__builtin_nontemporal_load(col + idx);                 // streamed once: keep L2 for H
            a = __builtin_nontemporal_load(val + idx);
        }
    };
#pragma unroll
    for (int p = 0; p < PIECES; ++p) fetch(e0 + p * stride + sub, c_next[p], a_next[p]);
    for (int base = e0; base < e1; base += PIECES * stride) {
        unsigned r[PIECES];
        float a[PIECES];
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            r[p] = (BIG || STYLE == 0) ? c_next[p] : (c_next[p] == kNoEdge ? kOOBRow : c_next[p] * ld_bytes);
            a[p] = Elem<T>::to_f32(a_next[p]);
            // the next iteration's (column, value) pairs are requested before this iteration's gathers
            fetch(base + (PIECES + p) * stride + sub, c_next[p], a_next[p]);
        }
#ifndef SGX_SPMM_ADJ_INFLIGHT
#define SGX_SPMM_ADJ_INFLIGHT 8
#endif
#ifndef SGX_SPMM_FEA_INFLIGHT
#define SGX_SPMM_FEA_INFLIGHT 4
#endif
        // edges of one piece whose gathers go together (the loop leaves a piece after the last group that
        // holds a valid edge, so a smaller group wastes fewer slots on short rows)
        constexpr int kInFlight = (STYLE == 1 ? SGX_SPMM_FEA_INFLIGHT : SGX_SPMM_ADJ_INFLIGHT) / CPL;
        static_assert(kInFlight >= 1, "SGX_SPMM_*_INFLIGHT must be at least the largest CPL (4): a group of 0 edges never advances");
        constexpr int UNR = LPR < kInFlight ? LPR : kInFlight;
#pragma unroll 1
        for (int t0 = 0; t0 < LPR; t0 += UNR) {
            if (t0 >= e1 - base) break;
            if constexpr (!BIG && STYLE == 0) {
                // gather and accumulate edge by edge; the compiler's schedule keeps a rolling window of
                // loads in flight across steps (best for the HBM-bound A.H stage)
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    const int n = e1 - base - p * stride;            // valid edges in this piece (may be <= 0)
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        // here the column index itself travels and every receiving lane forms its offset
                        // (measured: the A.H stage loses 1 % with the pre-multiplied row offset of STYLE 1)
                        const int t = t0 + u;
                        const unsigned cc = (unsigned)__shfl((int)r[p], t, LPR);
                        const float aa = __shfl(a[p], t, LPR);
#pragma unroll
                        for (int j = 0; j < CPL; ++j)
                            Gather<T, VEC>::run(acc + j * VEC, aa, rsrc,
                                                (t < n && chunk_off[j] != kOOB) ? cc * ld_bytes + chunk_off[j] : kOOB);
                    }
                }
            } else if constexpr (!BIG) {
                // all gathers of the step first, then the arithmetic (fewer registers, more wavefronts: best
                // for X.W, whose table sits in L2 and whose cost is instruction issue)
                typename GatherRaw<T, VEC>::raw_t raw[PIECES][UNR][CPL];
                float aa[PIECES][UNR];
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const unsigned rr = (unsigned)__shfl((int)r[p], t0 + u, LPR);
                        aa[p][u] = __shfl(a[p], t0 + u, LPR);
#pragma unroll
                        for (int j = 0; j < CPL; ++j)       // a lane whose chunk lies beyond n_feat stays out of range
                            raw[p][u][j] = GatherRaw<T, VEC>::load(rsrc, chunk_off[j] != kOOB ? rr + chunk_off[j] : kOOB);
                    }
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
#pragma unroll
                        for (int j = 0; j < CPL; ++j) GatherRaw<T, VEC>::fma(acc + j * VEC, aa[p][u], raw[p][u][j]);
            } else {
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const unsigned rr = (unsigned)__shfl((int)r[p], t0 + u, LPR);
                        const float aa = __shfl(a[p], t0 + u, LPR);
#pragma unroll
                        for (int j = 0; j < CPL; ++j)
                            if (rr != 0xFFFFFFFFu && chunk_off[j] != kOOB)
                                GatherPtr<T, VEC>::run(acc + j * VEC, aa, reinterpret_cast<const char *>(table) +
                                                                              (size_t)rr * ld_bytes + chunk_off[j]);
                    }
            }
        }
    }
}

template <typename T, int VEC>
__device__ __forceinline__ void store_row(T *__restrict__ drow, int col0, int n_feat, const float *acc, int relu,
                                          bool vec_store, const sgx_epilogue &ep)
{
    T out[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = finish_value<T>(acc[i], relu, ep);
    (void)vec_store;
    if (VEC > 1 && col0 + VEC <= n_feat) {
        // one 16-byte store even when the row of D is only element-aligned (P_w = 41, 47 ...):
        // gfx950 global stores need element alignment only
#if SGX_SPMM_NT_STORE
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(out),
                                    reinterpret_cast<typename Elem<T>::vec16_u *>(drow + col0));
#else
        *reinterpret_cast<typename Elem<T>::vec16_u *>(drow + col0) = *reinterpret_cast<const u32x4 *>(out);
#endif
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (col0 + i < n_feat) drow[col0 + i] = out[i];
    }
}

// ---------------------------------------------------------------------------------------
// One-step rows, 64 per wavefront.  In a power-law graph most rows hold a handful of edges (R-MAT S-100M: 47 % of the
// rows are their self loop, 80 % hold at most 8 edges) and their cost is not bytes but the chain row id -> row pointers
// -> (column, value) -> gather -> store, a memory round trip per link, paid by a wavefront for 64 / LPR rows at a
// time.  Here every link of the chain is ONE round trip for 64 rows: lane l reads the row id and the row pointers of
// row l, the (column, value) pairs of all LPR iterations are requested back to back, then the lane groups take
// 64 / LPR rows per iteration as in the sblock path -- the same fma chain per output element, hence the same bits.
// The plan's degree order ends with these rows (row_order[n_multi, n_rows): at most 8 edges each).
// ---------------------------------------------------------------------------------------
template <typename T, int VEC, int LPR>
__device__ __forceinline__ void spmm_short_rows(
    int64_t i0, int n_items, int n_feat, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const T *__restrict__ val, __amdgpu_buffer_rsrc_t rsrc, unsigned ld_bytes, T *__restrict__ D, int64_t ldd, int relu,
    int vec_store, const int32_t *__restrict__ row_order, const float *__restrict__ acc_in, float *__restrict__ acc_out,
    int64_t ld_acc, const sgx_epilogue &ep)
{
    static_assert(LPR >= 8, "a row's 8 edge slots are loaded by 8 lanes of its group");
    constexpr int RPW = 64 / LPR, ITER = LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, grp = lane / LPR;
    const int64_t idx = i0 + lane;
    const bool valid = idx < n_items;
    const int r = row_order[valid ? idx : (int64_t)n_items - 1];
    const int e0 = rowptr[r];
    const int deg = valid ? rowptr[r + 1] - e0 : 0;               // at most 8 (the plan's last buckets)
    // the (column, value) pairs of all iterations; unconditional loads (a conditional load makes hipcc wait for it at
    // the join): slots past the end of a row read entry 0 and are masked when they are used
    // (the iterations are taken 8 at a time -- the pairs of one batch are 16 registers -- whatever the lane split)
    constexpr int CH = 8;
    const int col0 = sub * VEC;
    const unsigned chunk_off = col0 < n_feat ? (unsigned)col0 * (unsigned)sizeof(T) : kOOB;
    for (int it0 = 0; it0 < ITER; it0 += CH) {
    unsigned c[CH];
    T a[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int s = (it0 + i) * RPW + grp;
        const int se0 = __shfl(e0, s), sdeg = __shfl(deg, s);
        const int e = sub < sdeg ? se0 + sub : 0;
        c[i] = (unsigned)__builtin_nontemporal_load(col + e);
        a[i] = __builtin_nontemporal_load(val + e);
    }
#pragma unroll
    for (int it = 0; it < CH; ++it) {
        const int s = (it0 + it) * RPW + grp;
        const int sdeg = __shfl(deg, s);
        const int64_t rr = __shfl(r, s);
        const bool live = __shfl((int)valid, s) != 0;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        if (acc_in && live) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (col0 + i < n_feat) acc[i] = acc_in[rr * ld_acc + col0 + i];
        }
        const float af = Elem<T>::to_f32(a[it]);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned cc = (unsigned)__shfl((int)c[it], t, LPR);
            const float aa = __shfl(af, t, LPR);
            Gather<T, VEC>::run(acc, aa, rsrc, (t < sdeg && chunk_off != kOOB) ? cc * ld_bytes + chunk_off : kOOB);
        }
        if (live && acc_out) {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                if (col0 + i < n_feat) acc_out[rr * ld_acc + col0 + i] = acc[i];
        } else if (live && col0 < n_feat) {
            store_row<T, VEC>(D + rr * ldd, col0, n_feat, acc, relu, vec_store != 0, ep);
        }
    }
    }
}

// ---------------------------------------------------------------------------------------
// One launch, two kinds of workgroup.  Workgroups [0, split_blocks) run the split path: one
// wavefront per (long row, 512-edge chunk), all 64/LPR lane groups on the same row, fp32 partial
// rows.  The others run the sblock path: one group of LPR lanes per row, 64/LPR rows per
// wavefront.  Putting both in one grid lets the heavy chunks start first and the short rows fill
// in around them (on R-MAT the two paths as separate launches took 0.98 + 1.24 ms back to back).
// ---------------------------------------------------------------------------------------
template <typename T, int VEC, int LPR, int CPL, bool BIG, int STYLE, bool SHORT>
__device__ __forceinline__ void spmm_body(
    int n_rows, int n_feat, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const T *__restrict__ val, const T *__restrict__ H, unsigned h_bytes, unsigned ld_bytes,
    T *__restrict__ D, int64_t ldd, int relu, int long_threshold, int vec_store,
    const int32_t *__restrict__ row_order,
    int split_blocks, int n_tasks, const int32_t *__restrict__ task_e0, const int32_t *__restrict__ task_e1,
    float *__restrict__ partial, int ldp,
    const float *__restrict__ acc_in, float *__restrict__ acc_out, int64_t ld_acc, sgx_epilogue ep, int n_multi, int short_first)
{
    constexpr int RPW = 64 / LPR;                 // rows per wavefront
    constexpr int LANE_COLS = CPL * VEC;          // columns per lane
    constexpr int TILE = LPR * LANE_COLS;         // columns covered per pass
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(H), 0, h_bytes, 0x00020000);

    if ((int)blockIdx.x < split_blocks) {
        // ---- split path ----
        const int task = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        if (task >= n_tasks) return;
        const int e0 = task_e0[task], e1 = task_e1[task];
        for (int c0 = 0; c0 < n_feat; c0 += TILE) {
            const int col0 = c0 + sub * LANE_COLS;
            unsigned chunk_off[CPL];
            float acc[LANE_COLS];
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                chunk_off[j] = col0 + j * VEC < n_feat ? (unsigned)(col0 + j * VEC) * (unsigned)sizeof(T) : kOOB;
#pragma unroll
            for (int i = 0; i < LANE_COLS; ++i) acc[i] = 0.0f;
            accumulate_edges<T, VEC, LPR, CPL, BIG, STYLE>(acc, e0 + grp * LPR, e1, 64, sub, col, val, rsrc, H, ld_bytes,
                                                    chunk_off);
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int i = 0; i < LANE_COLS; ++i) acc[i] += __shfl_xor(acc[i], off);
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < LANE_COLS; ++i)
                    if (col0 + i < n_feat) partial[(int64_t)task * ldp + col0 + i] = acc[i];
            }
        }
        return;
    }

    if constexpr (SHORT && LPR >= 8 && CPL == 1 && !BIG) {
        // ---- one-step rows: the tail of the degree order, 64 rows per wavefront (workgroups from short_first on) ----
        if ((int)blockIdx.x >= short_first) {
            const int64_t i0 = (int64_t)n_multi + ((int64_t)((int)blockIdx.x - short_first) * (kBlock / 64) + (threadIdx.x >> 6)) * 64;
            if (i0 < n_rows)
                spmm_short_rows<T, VEC, LPR>(i0, n_rows, n_feat, rowptr, col, val, rsrc, ld_bytes, D, ldd, relu, vec_store, row_order,
                                             acc_in, acc_out, ld_acc, ep);
            return;
        }
        n_rows = n_multi;                    // the sblock path below walks the rows of two steps and more
    }
    const int row_grid = (SHORT ? short_first : (int)gridDim.x) - split_blocks;
    // ---- sblock path ----
    const int64_t wave = (int64_t)(blockIdx.x - split_blocks) * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)row_grid * (kBlock / 64);
    // n_rows = number of work items; row_order (from the plan) lists the rows in degree order so
    // that the 64/LPR rows a wavefront owns need about the same number of steps
    for (int64_t r0 = wave * RPW; r0 < n_rows; r0 += n_waves * RPW) {
        int64_t r = r0 + grp;
        int e0 = 0, e1 = 0;
        bool live = r < n_rows;
        if (live) {
            if (row_order) r = row_order[r];
            e0 = rowptr[r];
            e1 = rowptr[r + 1];
            if (long_threshold > 0 && e1 - e0 > long_threshold) live = false;   // split path owns it
        }
        if (!live) e1 = e0;
        for (int c0 = 0; c0 < n_feat; c0 += TILE) {
            const int col0 = c0 + sub * LANE_COLS;
            unsigned chunk_off[CPL];
            float acc[LANE_COLS];
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                chunk_off[j] = col0 + j * VEC < n_feat ? (unsigned)(col0 + j * VEC) * (unsigned)sizeof(T) : kOOB;
#pragma unroll
            for (int i = 0; i < LANE_COLS; ++i) acc[i] = 0.0f;
            // two-pass aggregation (own-partition edges while the halo rows travel, then the halo
            // edges): the second pass starts from the first pass's fp32 sums
            if (acc_in && live) {
#pragma unroll
                for (int i = 0; i < LANE_COLS; ++i)
                    if (col0 + i < n_feat) acc[i] = acc_in[r * ld_acc + col0 + i];
            }
            accumulate_edges<T, VEC, LPR, CPL, BIG, STYLE>(acc, e0, e1, LPR, sub, col, val, rsrc, H, ld_bytes, chunk_off);
            if (live && acc_out) {
#pragma unroll
                for (int i = 0; i < LANE_COLS; ++i)
                    if (col0 + i < n_feat) acc_out[r * ld_acc + col0 + i] = acc[i];
            } else if (live) {
#pragma unroll
                for (int j = 0; j < CPL; ++j)
                    if (col0 + j * VEC < n_feat)
                        store_row<T, VEC>(D + r * ldd, col0 + j * VEC, n_feat, acc + j * VEC, relu, vec_store != 0, ep);
            }
        }
    }
}

#define SGX_SPMM_PARAMS                                                                                           \
    int n_rows, int n_feat, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,                \
        const T *__restrict__ val, const T *__restrict__ H, unsigned h_bytes, unsigned ld_bytes, T *__restrict__ D, \
        int64_t ldd, int relu, int long_threshold, int vec_store, const int32_t *__restrict__ row_order,          \
        int split_blocks, int n_tasks, const int32_t *__restrict__ task_e0, const int32_t *__restrict__ task_e1,  \
        float *__restrict__ partial, int ldp, const float *__restrict__ acc_in, float *__restrict__ acc_out,       \
        int64_t ld_acc, sgx_epilogue ep, int n_multi, int short_first
#define SGX_SPMM_ARGS                                                                                            \
    n_rows, n_feat, rowptr, col, val, H, h_bytes, ld_bytes, D, ldd, relu, long_threshold, vec_store, row_order,  \
        split_blocks, n_tasks, task_e0, task_e1, partial, ldp, acc_in, acc_out, ld_acc, ep, n_multi, short_first

// Two entry points over the one body, so that a profile tells the stages apart:
// spmm_kernel = the A.H aggregation (loop_adj), xw_sparse_kernel = X.W with a CSR X (loop_fea, the
// weight matrix as the gathered table).
template <typename T, int VEC, int LPR, int CPL, bool BIG>
__global__ __launch_bounds__(kBlock, SGX_SPMM_MINWAVES) void spmm_kernel(SGX_SPMM_PARAMS)
{
    spmm_body<T, VEC, LPR, CPL, BIG, SGX_SPMM_ADJ_STYLE, false>(SGX_SPMM_ARGS);
}

// the A.H aggregation under a degree-ordered plan whose order ends in one-step rows: those 64 to a wavefront
// (spmm_short_rows); its own kernel so that the plain one -- the uniform graph's -- keeps its code as measured
template <typename T, int VEC, int LPR, int CPL, bool BIG>
__global__ __launch_bounds__(kBlock, SGX_SPMM_MINWAVES) void spmm_short_tail_kernel(SGX_SPMM_PARAMS)
{
    spmm_body<T, VEC, LPR, CPL, BIG, SGX_SPMM_ADJ_STYLE, true>(SGX_SPMM_ARGS);
}

template <typename T, int VEC, int LPR, int CPL, bool BIG>
__global__ __launch_bounds__(kBlock, SGX_SPMM_MINWAVES) void xw_sparse_kernel(SGX_SPMM_PARAMS)
{
    spmm_body<T, VEC, LPR, CPL, BIG, SGX_SPMM_FEA_STYLE, false>(SGX_SPMM_ARGS);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void spmm_split_finalize_kernel(
    int n_long, int n_feat, const int32_t *__restrict__ long_row, const int32_t *__restrict__ long_first,
    const float *__restrict__ partial, int ldp, T *__restrict__ D, int64_t ldd, int relu,
    const float *__restrict__ acc_in, float *__restrict__ acc_out, int64_t ld_acc, sgx_epilogue ep)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t total = (int64_t)n_long * n_feat;
    if (gid >= total) return;
    const int l = (int)(gid / n_feat), j = (int)(gid % n_feat);
    const int64_t row = long_row[l];
    float s = acc_in ? acc_in[row * ld_acc + j] : 0.0f;
    // the partial rows of a long row are added in task order; eight loads are in flight at a time (a hub row has tens to
    // hundreds of tasks: one dependent load after the other was a round trip to memory each)
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
    if (acc_out) { acc_out[row * ld_acc + j] = s; return; }
    D[row * ldd + j] = finish_value<T>(s, relu, ep);
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
struct LaunchArgs {
    int relu, n_rows, n_feat;
    const int32_t *rowptr, *col;
    const void *val, *H;
    unsigned h_bytes, ld_bytes;
    void *D;
    int64_t ldd;
    const sgx_plan *plan;
    float *partial;
    int ldp;
    int vec_store;
    bool big;
    const float *acc_in;
    float *acc_out;
    int64_t ld_acc;
    bool fea_stage;           // launched for X.W (sgx_xw_sparse): same body under its own kernel name
    sgx_epilogue ep;
    hipStream_t stream;
};

int grid_for_rows(int64_t n_rows, int rows_per_wave)
{
    const int64_t rows_per_block = (int64_t)rows_per_wave * (kBlock / 64);
    int64_t blocks = (n_rows + rows_per_block - 1) / rows_per_block;
    const int64_t cap = 256 * SGX_SPMM_BLOCKS_PER_CU;      // grid-stride beyond
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <typename T, int VEC, int LPR, int CPL, bool BIG>
int launch_one_impl(const LaunchArgs &a)
{
    const sgx_plan *p = a.plan;
    const int long_thr = (p && p->n_long > 0) ? p->long_threshold : 0;
    const int32_t *order = p ? p->row_order : nullptr;
    const int n_work = order ? p->n_ordered : a.n_rows;
    const int n_tasks = long_thr > 0 ? p->n_tasks : 0;
    const int split_blocks = (n_tasks + kBlock / 64 - 1) / (kBlock / 64);
    // the one-step tail of a degree order: 64 rows per wavefront (spmm_short_rows); the rows ahead of it keep the sblock path
    int n_multi = n_work, short_blocks = 0;
    bool short_tail = false;
    if constexpr (LPR >= 8 && CPL == 1 && !BIG) {
        short_tail = order && !a.fea_stage && !sgx_tune().spmm_no_short_tail && p->n_multi >= 0 && p->n_multi < n_work &&
                     a.n_feat <= LPR * VEC && n_work - p->n_multi >= 4096;
        if (short_tail) {
            n_multi = p->n_multi;
            short_blocks = (int)(((int64_t)(n_work - n_multi) + 64 * (kBlock / 64) - 1) / (64 * (kBlock / 64)));
        }
    }
    const int row_blocks = n_multi > 0 ? grid_for_rows(n_multi, 64 / LPR) : 0;
    if (split_blocks + row_blocks + short_blocks > 0) {
        auto kernel = a.fea_stage ? xw_sparse_kernel<T, VEC, LPR, CPL, BIG>
                                  : (short_tail ? spmm_short_tail_kernel<T, VEC, LPR, CPL, BIG> : spmm_kernel<T, VEC, LPR, CPL, BIG>);
        hipLaunchKernelGGL(kernel, dim3(split_blocks + row_blocks + short_blocks), dim3(kBlock), 0,
                           a.stream, n_work, a.n_feat, a.rowptr, a.col, (const T *)a.val, (const T *)a.H, a.h_bytes,
                           a.ld_bytes, (T *)a.D, a.ldd, a.relu, long_thr, a.vec_store, order, split_blocks, n_tasks,
                           n_tasks ? p->task_e0 : nullptr, n_tasks ? p->task_e1 : nullptr, a.partial, a.ldp, a.acc_in,
                           a.acc_out, a.ld_acc, a.ep, n_multi, split_blocks + row_blocks);
        SGX_LAUNCH_CHECK();
    }
    if (n_tasks > 0) {
        const int64_t total = (int64_t)p->n_long * a.n_feat;
        hipLaunchKernelGGL((spmm_split_finalize_kernel<T>), dim3((unsigned)((total + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, a.stream, p->n_long, a.n_feat, p->long_row, p->long_first, a.partial,
                           a.ldp, (T *)a.D, a.ldd, a.relu, a.acc_in, a.acc_out, a.ld_acc, a.ep);
        SGX_LAUNCH_CHECK();
    }
    return SGX_OK;
}

// slots = LPR * CPL = 16-byte chunks per row (power of two); cpl in {1, 2, 4}, cpl <= slots.
// Tables of 4 GiB and more always run with CPL = 1.
template <typename T, int VEC, int SLOTS>
int launch_slots(const LaunchArgs &a, int cpl)
{
    if (a.big) return launch_one_impl<T, VEC, SLOTS, 1, true>(a);
    if constexpr (VEC > 1 && SLOTS >= 4) {
        if (cpl == 4) return launch_one_impl<T, VEC, SLOTS / 4, 4, false>(a);
    }
    if constexpr (VEC > 1 && SLOTS >= 2) {
        if (cpl >= 2) return launch_one_impl<T, VEC, SLOTS / 2, 2, false>(a);
    }
    return launch_one_impl<T, VEC, SLOTS, 1, false>(a);
}

template <typename T, int VEC>
int launch_lpr(const LaunchArgs &a, int slots, int cpl)
{
    switch (slots) {
    case 1: return launch_slots<T, VEC, 1>(a, cpl);
    case 2: return launch_slots<T, VEC, 2>(a, cpl);
    case 4: return launch_slots<T, VEC, 4>(a, cpl);
    case 8: return launch_slots<T, VEC, 8>(a, cpl);
    case 16: return launch_slots<T, VEC, 16>(a, cpl);
    case 32: return launch_slots<T, VEC, 32>(a, cpl);
    default: return launch_slots<T, VEC, 64>(a, cpl);
    }
}

// Lanes per row from the mean degree.  Measured on 4 M-row uniform graphs at F = 64 fp16
// (tools/cpl_probe.py): halving the lane group (CPL = 2) wins only below ~5 edges per row
// (0.34 vs 0.47 ms at 2 edges/row, 0.48 vs 0.55 ms at 4) and loses above (1.23 vs 1.13 ms at 13,
// 2.52 vs 2.18 ms at 25; sparse X.W 0.98 vs 0.86 ms); CPL = 4 never wins.
int choose_cpl(const sgx_plan *plan, int slots)
{
    if (sgx_tune().spmm_cpl) return sgx_tune().spmm_cpl;                // tuning override (tools/cpl_probe.py)
    if (!plan || plan->n_rows <= 0 || slots < 2) return 1;
    const double avg = (double)plan->nnz / (double)plan->n_rows;
    return avg < kShortRowDegree ? 2 : 1;
}

}  // namespace

size_t sgx_spmm_scratch_bytes(const sgx_plan *plan, int n_feat)
{
    if (!plan || plan->n_tasks == 0 || n_feat <= 0) return 0;
    return sgx_align_up((size_t)plan->n_tasks * (size_t)sgx_align_up((size_t)n_feat, 4) * sizeof(float), 256);
}

int sgx_spmm_launch(int dtype, int acc_mode, int spmm_block, int relu, int n_rows, int n_cols, int n_feat,
                    const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                    const void *H, int64_t ldh, void *D, int64_t ldd,
                    const sgx_plan *plan, void *scratch, size_t scratch_bytes, hipStream_t stream,
                    const float *acc_in, float *acc_out, int64_t ld_acc, bool fea_stage, int ref_threads, sgx_epilogue ep)
{
    (void)spmm_block;
    if (n_rows < 0 || n_cols < 0 || n_feat < 1 || ldh < n_feat) return SGX_ERR_SHAPE;
    if (n_rows == 0) return SGX_OK;
    if (!rowPtr || (!D && !acc_out)) return SGX_ERR_NULL;
    if (D && ldd < n_feat) return SGX_ERR_SHAPE;
    if ((acc_in || acc_out) && (ld_acc < n_feat || acc_mode != SGX_ACC_F32)) return SGX_ERR_SHAPE;
    if (dtype != SGX_F16 && dtype != SGX_F32) return SGX_ERR_UNSUPPORTED;
    if (acc_mode == SGX_ACC_REF_HALF) {
        if (dtype != SGX_F16) return SGX_ERR_UNSUPPORTED;
        if (n_cols > 0 && (!H || !columnIndex || !values)) return SGX_ERR_NULL;
        return sgx_refhalf_csr(spmm_block, ref_threads, relu, n_rows, n_cols, n_feat, rowPtr, columnIndex, values, H, ldh, D, ldd, stream);
    }
    if (acc_mode != SGX_ACC_F32) return SGX_ERR_UNSUPPORTED;
    if (plan && plan->n_rows != n_rows) return SGX_ERR_SHAPE;
    const size_t es = sgx_elem_size(dtype);
    const unsigned long long table_bytes = (unsigned long long)n_cols * (unsigned long long)ldh * es;
    // 32-bit buffer offsets need the whole table below kOOBRow and a row below 1 MiB (sgx_device.h)
    const bool big = table_bytes > kOOBRow || (unsigned long long)n_feat * es > kMaxRowBytes;
    if ((unsigned long long)ldh * es >= 0xFFFFFFF0ull) return SGX_ERR_UNSUPPORTED;
    if (n_cols > 0 && (!H || !columnIndex || !values)) return SGX_ERR_NULL;
    // X.W over a large CSR X: the weight slice resident in LDS instead of gathered through L2 (same sums, same order)
    if (fea_stage && !acc_in && !acc_out && !big && sgx_xw_sparse_lds_applicable(dtype, n_rows, n_cols, n_feat, ldd, plan))
        return sgx_xw_sparse_lds(dtype, n_rows, n_cols, n_feat, rowPtr, columnIndex, values, H, ldh, D, ldd, plan, ep, stream);

    LaunchArgs a;
    a.relu = relu; a.n_rows = n_rows; a.n_feat = n_feat;
    a.rowptr = rowPtr; a.col = columnIndex; a.val = values; a.H = H;
    a.h_bytes = big ? 0u : (unsigned)table_bytes; a.ld_bytes = (unsigned)(ldh * es); a.big = big;
    a.D = D; a.ldd = ldd; a.plan = plan; a.stream = stream;
    a.acc_in = acc_in; a.acc_out = acc_out; a.ld_acc = ld_acc; a.fea_stage = fea_stage; a.ep = ep;
    a.partial = (float *)scratch;
    a.ldp = (int)sgx_align_up((size_t)n_feat, 4);
    if (plan && plan->n_tasks > 0) {
        if (!scratch || scratch_bytes < sgx_spmm_scratch_bytes(plan, n_feat)) return SGX_ERR_WORKSPACE;
    }
    const int per16 = (int)(16 / es);
    // 16-byte gathers need rows that start on a dword: buffer loads take dword-aligned addresses (gfx950 runs in
    // unaligned-access mode) and are range-checked dword by dword, so a chunk that runs past the end of a row reads
    // the neighbouring row's first elements into lanes that are never stored, and past the end of the table zeros.
    // Rows on odd halves (P_w = 41 unpadded) would pair their last element with a dword outside the table, and the
    // 64-bit pointer path has no range check: those keep one element per lane unless rows are 16-byte aligned.
    const bool rows16 = ((uintptr_t)H % 16 == 0) && ((ldh * es) % 16 == 0);
    const bool rows4 = !big && ((uintptr_t)H % 4 == 0) && ((ldh * es) % 4 == 0);
    const bool vec_gather = rows16 || rows4;
    a.vec_store = ((uintptr_t)D % 16 == 0) && ((ldd * es) % 16 == 0);
    if (vec_gather) {
        int slots = sgx_next_pow2((n_feat + per16 - 1) / per16);
        if (slots > 64) slots = 64;
        const int cpl = choose_cpl(plan, slots);
        return dtype == SGX_F16 ? launch_lpr<f16, 8>(a, slots, cpl) : launch_lpr<float, 4>(a, slots, cpl);
    }
    int slots = sgx_next_pow2(n_feat);
    if (slots > 64) slots = 64;
    return dtype == SGX_F16 ? launch_lpr<f16, 1>(a, slots, 1) : launch_lpr<float, 1>(a, slots, 1);
}

extern "C" int sgx_spmm_csr(int dtype, int acc_mode, int spmm_block, int relu, int n_rows, int n_cols, int n_feat,
                            const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                            const void *H, int64_t ldh, void *D, int64_t ldd,
                            const sgx_plan *plan, void *scratch, size_t scratch_bytes, void *stream)
{
    return sgx_spmm_launch(dtype, acc_mode, spmm_block, relu, n_rows, n_cols, n_feat, rowPtr, columnIndex, values,
                           H, ldh, D, ldd, plan, scratch, scratch_bytes, (hipStream_t)stream, nullptr, nullptr, 0);
}

extern "C" int sgx_spmm_csr_acc(int dtype, int relu, int n_rows, int n_cols, int n_feat,
                                const int32_t *rowPtr, const int32_t *columnIndex, const void *values,
                                const void *H, int64_t ldh, void *D, int64_t ldd,
                                const float *acc_in, float *acc_out, int64_t ld_acc,
                                const sgx_plan *plan, void *scratch, size_t scratch_bytes, void *stream)
{
    if (D && acc_out) return SGX_ERR_SHAPE;           // one destination: the final rows or the fp32 partial
    return sgx_spmm_launch(dtype, SGX_ACC_F32, 1, relu, n_rows, n_cols, n_feat, rowPtr, columnIndex, values, H, ldh, D,
                           ldd, plan, scratch, scratch_bytes, (hipStream_t)stream, acc_in, acc_out, ld_acc);
}
