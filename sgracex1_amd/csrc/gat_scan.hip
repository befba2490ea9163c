// Stage A of the two-stage GAT aggregate (gat.hip) for rows of up to kScanMaxRow entries, as a segmented scan in entry order:
//     alpha_e = exp(x_e - m_row) / l_row,   x_e = LeakyReLU(s1[row] + s2[col_e]),   live edges only (values[e] > 0)
// -- the reference's `attention` matrix on the stored entries (SG.py:634-657; the hardware's E / S side outputs,
// SG.py:500-502).  A wavefront owns the rows that START in one window of WIN stored entries (the plan's scan_win tells the
// first row and entry of every window: the reference's routing of rows to compute units by running entry count,
// K.cpp:826-845, made once per matrix): at most WIN + 255 entries, a lane per entry and chunk of 64, the heads (HB at a
// time) in its registers.  Columns, values and score rows are read in entry order -- whole lines, every lane busy
// whatever the rows' lengths; a wavefront of 8-lane row groups idles behind its longest row on a power-law graph.
// The rows of the range: the next row starts, 64 at a time, each marked at its first entry's slot in LDS; a maximum scan
// of the marks gives every entry its row.  Per head: the running maximum within rows (segmented scan over the lanes with
// row_shr / row_bcast DPP steps, carried from chunk to chunk), the row total handed back to every entry of the row,
// exp(x - m), the same for the sum, and the weights.  Rows over kScanMaxRow entries stay with the plan's tasks (gat.hip).
// Numerics: the maximum is exact; the sum is added in scan order instead of the register pass's lane order, so a weight
// may differ from that pass's in the last bits (tests/test_gpu_gat_scan.py states the bound against an fp64 softmax).
#include "gat_device.h"

namespace {

constexpr int kScanWaves = kBlock / 64;
#ifndef SGX_GAT_SCAN_WIN
#define SGX_GAT_SCAN_WIN 128        // stored entries per window with one or two heads (64 / 128 / 256; measured, see DESIGN)
#endif
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118, kRowBcast15 = 0x142, kRowBcast31 = 0x143;

template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_f32(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, false));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWS, 0xf, false); }

struct MaxOp {
    static __device__ __forceinline__ float ident() { return -INFINITY; }
    // (fmaxf quiets a signalling NaN first: one more v_max_f32 per operand the compiler cannot see through, a third of the
    // scan's instructions; the instruction itself does the same to its inputs)
    static __device__ __forceinline__ float op(float a, float b)
    {
        float r;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
};
struct AddOp {
    static __device__ __forceinline__ float ident() { return 0.0f; }
    static __device__ __forceinline__ float op(float a, float b) { return a + b; }
};

// Inclusive scan over the lanes within segments.  `dist` = lane - first lane of the lane's segment in this chunk (the
// segment may have begun earlier: then 0 counts as its first lane).  Lanes without a source take the identity; a step
// applies where its source lane lies in the same segment.
template <typename Op>
__device__ __forceinline__ float seg_scan(float v, int dist, int lane)
{
    // (the six comparisons below do not depend on the head: hoisted out of the head loop, 6 x chunks lane masks per
    // wavefront live in scalar registers and spill; recomputed here they cost one v_cmp each, straight into the select)
    asm volatile("" : "+v"(dist));
    float t;
    t = dpp_f32<kRowShr1, 0xf>(Op::ident(), v); v = dist >= 1 ? Op::op(v, t) : v;
    t = dpp_f32<kRowShr2, 0xf>(Op::ident(), v); v = dist >= 2 ? Op::op(v, t) : v;
    t = dpp_f32<kRowShr4, 0xf>(Op::ident(), v); v = dist >= 4 ? Op::op(v, t) : v;
    t = dpp_f32<kRowShr8, 0xf>(Op::ident(), v); v = dist >= 8 ? Op::op(v, t) : v;
    t = dpp_f32<kRowBcast15, 0xa>(Op::ident(), v); v = dist > (lane & 15) ? Op::op(v, t) : v;     // lane 15 -> row 1, lane 47 -> row 3
    t = dpp_f32<kRowBcast31, 0xc>(Op::ident(), v); v = dist >= lane - 31 ? Op::op(v, t) : v;       // lane 31 -> rows 2 and 3
    return v;
}

// inclusive maximum over lanes [0, lane] of non-negative marks
__device__ __forceinline__ int wave_scan_max(int v)
{
    v = max(v, dpp_i32<kRowShr1, 0xf>(0, v));
    v = max(v, dpp_i32<kRowShr2, 0xf>(0, v));
    v = max(v, dpp_i32<kRowShr4, 0xf>(0, v));
    v = max(v, dpp_i32<kRowShr8, 0xf>(0, v));
    v = max(v, dpp_i32<kRowBcast15, 0xa>(0, v));
    v = max(v, dpp_i32<kRowBcast31, 0xc>(0, v));
    return v;
}

__device__ __forceinline__ float lane_value(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane * 4, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float read_lane(float v, int src_lane)     // src_lane a constant
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// the row holding stored entry e: the last r with rowptr[r] <= e (rowptr[0] == 0 <= e < rowptr[n_rows]); 64 probes a round
__device__ __forceinline__ int row_of_entry(const int32_t *__restrict__ rowptr, int n_rows, int e, int lane)
{
    int lo = 0, hi = n_rows;
    while (hi - lo > 1) {
        const int stride = (hi - lo + 63) / 64;
        const int p = min(lo + lane * stride, hi);
        const unsigned long long b = __ballot(rowptr[p] <= e);                     // a prefix of ones
        const int cnt = max(1, (int)__popcll(b));
        const int nhi = min(lo + cnt * stride, hi);
        lo = lo + (cnt - 1) * stride;
        hi = nhi;
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

// One range: NCH chunks of 64 entries from `base`, the rows [R0, R1) that start in it, entries up to `end`.
template <typename T, int HB, int NCH>
__device__ __forceinline__ void scan_range(
    int lane, int n_rows, int n_heads, int R0, int R1, int base, int end, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const __amdgpu_buffer_rsrc_t &col_rsrc, const __amdgpu_buffer_rsrc_t &val_rsrc,
    const float *__restrict__ s1, const float *__restrict__ s2, float alpha, float *__restrict__ W, float *__restrict__ E,
    unsigned char *__restrict__ dead, int *marks, bool vec)
{
    float x[NCH][HB];                         // the score rows of the columns; then scores, exponentials, weights
    unsigned pos = 0u, valid = 0u;            // per chunk: a live entry / an entry of the range
    {
        unsigned c[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int e = base + 64 * k + lane;
            const unsigned off = e < end ? (unsigned)e * 4u : kOOB;
            c[k] = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, off, 0, 0);        // (out of range: node 0, never stored)
            float v;
            if constexpr (sizeof(T) == 2) v = (float)__builtin_bit_cast(T, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(val_rsrc, off >> 1, 0, 0));
            else v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, off, 0, 0));
            pos |= v > 0.0f ? (1u << k) : 0u;
            valid |= e < end ? (1u << k) : 0u;
            marks[64 * k + lane] = 0;
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) load_scores<HB>(s2, (int64_t)c[k], n_heads, 0, vec, x[k]);
    }
    // the rows of the range: marks[slot of a row's first entry] = row + 1
    for (int r0 = R0;;) {
        const int rj = r0 + lane;
        const int st = rowptr[min(rj, n_rows)], en = rowptr[min(rj + 1, n_rows)];
        if (rj < R1 && en > st && st - base < 64 * NCH) marks[st - base] = rj + 1;
        if (r0 + 64 >= R1) break;
        const int st0 = __builtin_amdgcn_readfirstlane(st), en_last = __builtin_amdgcn_readlane(en, 63);
        if (en_last >= end) break;
        r0 = en_last == st0 ? max(r0 + 64, row_of_entry(rowptr, n_rows, en_last, lane)) : r0 + 64;   // (64 rows without entries: on to the next entry's row)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // every entry's row (left in the marks' place), its distance from the row's first lane in the chunk and the row's last lane
    int where[NCH];                           // dist | last lane << 8
    unsigned cont = 0u, reach = 0u, starts = 0u;     // per chunk: no row start at or before the lane / the segment runs into the next chunk / a row's first entry
    {
        int carry = R0 + 1;
        unsigned long long heads[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int mk = marks[64 * k + lane];
            heads[k] = __ballot(mk != 0);
            starts |= mk != 0 ? (1u << k) : 0u;
            const int v = max(wave_scan_max(mk), carry);
            marks[64 * k + lane] = min(v - 1, n_rows - 1);
            carry = __builtin_amdgcn_readlane(v, 63);
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
            const unsigned long long mine = heads[k] & le;
            const int s_eff = mine ? 63 - __builtin_clzll(mine) : 0;
            cont |= mine ? 0u : (1u << k);
            const unsigned long long above = lane == 63 ? 0ull : (heads[k] >> (lane + 1));
            where[k] = (lane - s_eff) | ((above ? lane + __builtin_ctzll(above) : 63) << 8);
            const bool next_goes_on = k + 1 < NCH && !(heads[k + 1 < NCH ? k + 1 : k] & 1ull);
            reach |= (!above && next_goes_on) ? (1u << k) : 0u;
        }
    }
    pos &= valid;

    for (int hb0 = 0; hb0 < n_heads; hb0 += HB) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (hb0) {
                const int e = base + 64 * k + lane;
                load_scores<HB>(s2, (int64_t)(e < end ? col[e] : 0), n_heads, hb0, vec, x[k]);
            }
            float si[HB];
            load_scores<HB>(s1, (int64_t)marks[64 * k + lane], n_heads, hb0, vec, si);
#pragma unroll
            for (int h = 0; h < HB; ++h) x[k][h] = leaky(si[h] + x[k][h], alpha);
            if (E && ((valid >> k) & 1u)) store_heads<HB>(E, base + 64 * k + lane, n_heads, hb0, vec, x[k]);
        }
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            float run[NCH], tot[NCH];
            {   // maximum of the row's live entries
                float carry = -INFINITY;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    float v = seg_scan<MaxOp>((pos >> k) & 1u ? x[k][h] : -INFINITY, where[k] & 255, lane);
                    v = (cont >> k) & 1u ? MaxOp::op(v, carry) : v;
                    carry = read_lane(v, 63);
                    run[k] = v;
                }
                float back = -INFINITY;
#pragma unroll
                for (int k = NCH - 1; k >= 0; --k) {
                    const float t = lane_value(run[k], where[k] >> 8);
                    tot[k] = (reach >> k) & 1u ? back : t;
                    back = read_lane(tot[k], 0);
                }
            }
            {   // sum of exp(x - maximum)
                float carry = 0.0f;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const float p = (pos >> k) & 1u ? exp_weight(x[k][h] - tot[k]) : 0.0f;
                    x[k][h] = p;
                    float v = seg_scan<AddOp>(p, where[k] & 255, lane);
                    v = (cont >> k) & 1u ? v + carry : v;
                    carry = read_lane(v, 63);
                    run[k] = v;
                }
                float back = 0.0f;
#pragma unroll
                for (int k = NCH - 1; k >= 0; --k) {
                    const float t = lane_value(run[k], where[k] >> 8);
                    tot[k] = (reach >> k) & 1u ? back : t;
                    back = read_lane(tot[k], 0);
                }
            }
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                x[k][h] = tot[k] > 0.0f ? x[k][h] * __builtin_amdgcn_rcpf(tot[k]) : 0.0f;     // (v_rcp_f32: 1 ulp; a true division is ten instructions)
                if (dead && hb0 == 0 && h == 0 && ((starts >> k) & 1u)) dead[marks[64 * k + lane]] = tot[k] > 0.0f ? 0 : 1;    // (the mask does not depend on the head)
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if ((valid >> k) & 1u) store_heads<HB>(W, base + 64 * k + lane, n_heads, hb0, vec, x[k]);
    }
}

template <typename T, int HB, int WIN>
__global__ __launch_bounds__(kBlock) void gat_alpha_scan_kernel(
    int n_rows, int n_heads, int n_ranges, int n_granules, unsigned nnz, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ col, const T *__restrict__ val, const float *__restrict__ s1, const float *__restrict__ s2,
    float alpha, float *__restrict__ W, float *__restrict__ E, unsigned char *__restrict__ dead,
    const int32_t *__restrict__ scan_win, int vec)
{
    constexpr int STEP = WIN / kScanGranule, CH = (WIN + kScanMaxRow) / 64, HALF = (CH + 1) / 2;
    __shared__ int marks_lds[kScanWaves][64 * CH];
    const int lane = threadIdx.x & 63, wslot = threadIdx.x >> 6;
    const int wave = blockIdx.x * kScanWaves + wslot;
    if (wave >= n_ranges) return;                                                  // wave-uniform
    const int g0 = wave * STEP, g1 = min(g0 + STEP, n_granules);
    const int R0 = __builtin_amdgcn_readfirstlane(scan_win[4 * g0]), base = __builtin_amdgcn_readfirstlane(scan_win[4 * g0 + 1]);
    const int R1 = __builtin_amdgcn_readfirstlane(scan_win[4 * g1 + 2]), end = __builtin_amdgcn_readfirstlane(scan_win[4 * g1 + 3]);
    if (R1 <= R0 || end <= base) return;      // no row of the scan's starts here (inside a long row), or only rows without entries
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(col), 0, nnz * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(val), 0, nnz * (unsigned)sizeof(T), 0x00020000);
    if (end - base <= 64 * HALF)
        scan_range<T, HB, HALF>(lane, n_rows, n_heads, R0, R1, base, end, rowptr, col, col_rsrc, val_rsrc, s1, s2, alpha, W, E, dead,
                                marks_lds[wslot], vec);
    else
        scan_range<T, HB, CH>(lane, n_rows, n_heads, R0, R1, base, min(end, base + 64 * CH), rowptr, col, col_rsrc, val_rsrc, s1, s2, alpha,
                              W, E, dead, marks_lds[wslot], vec);
}

template <typename T, int HB, int WIN>
int launch_scan(int n_rows, int n_heads, const sgx_plan *p, const int32_t *rowptr, const int32_t *col, const void *val, const float *s1,
                const float *s2, float alpha, float *W, float *E, unsigned char *dead, hipStream_t stream)
{
    constexpr int STEP = WIN / kScanGranule;
    const int n_granules = (int)p->n_scan_win, n_ranges = (n_granules + STEP - 1) / STEP;
    const int vec = (HB >= 4 && n_heads % HB == 0 &&
                     (reinterpret_cast<uintptr_t>(s1) | reinterpret_cast<uintptr_t>(s2) | reinterpret_cast<uintptr_t>(W) |
                      reinterpret_cast<uintptr_t>(E)) % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL((gat_alpha_scan_kernel<T, HB, WIN>), dim3((unsigned)((n_ranges + kScanWaves - 1) / kScanWaves)), dim3(kBlock), 0,
                       stream, n_rows, n_heads, n_ranges, n_granules, (unsigned)p->nnz, rowptr, col, (const T *)val, s1, s2, alpha, W, E,
                       dead, p->scan_win, vec);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

template <typename T>
int launch_scan_heads(int n_rows, int n_heads, const sgx_plan *p, const int32_t *rowptr, const int32_t *col, const void *val,
                      const float *s1, const float *s2, float alpha, float *W, float *E, unsigned char *dead, hipStream_t stream)
{
    // (8 heads at a time: 149 registers, three wavefronts per SIMD, 657 us where two passes of 4 take 607)
    if (n_heads % 4 == 0) return launch_scan<T, 4, 128>(n_rows, n_heads, p, rowptr, col, val, s1, s2, alpha, W, E, dead, stream);
    if (n_heads % 2 == 0) return launch_scan<T, 2, SGX_GAT_SCAN_WIN>(n_rows, n_heads, p, rowptr, col, val, s1, s2, alpha, W, E, dead, stream);
    return launch_scan<T, 1, SGX_GAT_SCAN_WIN>(n_rows, n_heads, p, rowptr, col, val, s1, s2, alpha, W, E, dead, stream);
}

}  // namespace

// the plan carries the entry windows (its longer rows are cut at the scan's row limit, or it has none)
bool sgx_gat_scan_applicable(const sgx_plan *plan)
{
    return plan && plan->scan_win && plan->n_scan_win > 0 &&
           (plan->n_long > 0 ? plan->long_threshold == kScanMaxRow : plan->max_degree <= kScanMaxRow);
}

// W[e][h] (and E[e][h] = the scores before the softmax, if asked for) for the stored entries of every row of up to
// kScanMaxRow entries; dead[r] = 1 for such a row without a live entry and for rows without entries (nullptr: not wanted;
// the longer rows' flags are their tasks').  s1 / s2: [n_rows x n_heads] / [n_cols x n_heads].
int sgx_gat_alpha_scan(int dtype, int n_rows, int n_heads, const sgx_plan *plan, const int32_t *rowptr, const int32_t *col,
                       const void *val, const float *s1, const float *s2, float alpha, float *W, float *E, unsigned char *dead,
                       hipStream_t stream)
{
    if (!sgx_gat_scan_applicable(plan)) return SGX_ERR_UNSUPPORTED;
    if (dead && n_rows > 0 && hipMemsetAsync(dead, 1, (size_t)n_rows, stream) != hipSuccess) return SGX_ERR_HIP;   // rows without entries are met by no range
    if (dtype == SGX_F16) return launch_scan_heads<f16>(n_rows, n_heads, plan, rowptr, col, val, s1, s2, alpha, W, E, dead, stream);
    return launch_scan_heads<float>(n_rows, n_heads, plan, rowptr, col, val, s1, s2, alpha, W, E, dead, stream);
}
