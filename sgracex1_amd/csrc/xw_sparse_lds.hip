// X.W with a CSR X and the weight tile resident in LDS -- loop_fea / compute1 in gemm_mode 0 with the
// reference's B_accel tile (K.cpp:1960-2078, :3038-3051) as what it is there: an on-chip copy of W that
// every feature entry indexes.
//
// Why: with W gathered through L2 the stage moves 128 bytes of W per stored entry of X from L2 to L1
// (9.8 GB per S-100M launch at 13 TB/s: 0.70 ms), while HBM only has to deliver the CSR of X once and
// take H (1.0 GB).  From LDS the same rows come at 256 bytes per clock and CU.
//
// Layout.  W [M_fea][P] does not fit one CU's 160 KB at the Cora shape (1433 x 64 fp16 = 183 KB), so the
// columns are cut into S slices of CP = LPR x 16 bytes; a workgroup keeps ONE slice for all of K in LDS
// ((M_fea + 1) rows of LPR x 16 bytes; the extra row is zero and is what an empty entry slot reads) and
// walks row tiles persistently.  The S workgroups that hold the slices of the same rows are given
// block ids b, b + 8, ... -- the same XCD under round-robin placement -- and walk the same tiles in the
// same order, so the CSR of X comes from HBM once and from that XCD's L2 for the others (speed only;
// any placement gives the same results).
//
// A wavefront owns windows of 64 consecutive rows of X.  It is cut into groups of LPR lanes, one row per
// group, min(LPR, 4) entries per step: lane i of a quad loads entry i (column, value), quad_perm DPP
// broadcasts it, every lane reads its 16 bytes of that W row from LDS and adds value x W in fp32 in CSR
// order -- the same fma chain per output element as the gather kernel (spmm_csr.hip), hence the same bits.
// The groups of a wavefront step together, so a sub-tile of 64 / LPR rows costs its LONGEST row's steps:
// the 64 rows of a window are therefore sorted by step count inside the wavefront (bitonic network over
// the lanes) and dealt to the sub-tiles in that order -- the sblock idea of the reference (rows grouped
// per pipelined loop, K.cpp:826-845) with the grouping chosen by length.  The measured kernel is bound by
// vector-instruction issue, not by LDS or HBM, which is why the wasted steps matter.
// 16 wavefronts per CU is all a 92 KB table admits, so the latency of the (column, value) stream is
// covered in software: the entries of the NEXT sub-tile are requested before the current one is summed,
// the row pointers one window ahead; a row longer than D steps refills its ring as it goes.
#include "sgx_device.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kWaves = 16;                 // wavefronts per workgroup (1024 threads: one workgroup per CU)
constexpr int kThreads = kWaves * 64;
#ifndef SGX_XW_LDS_DEPTH
#define SGX_XW_LDS_DEPTH 8
#endif
constexpr int kDepth = SGX_XW_LDS_DEPTH;                  // steps of a row requested ahead (8 x 4 = 32 entries: S-100M rows hold 18, the longest of 64 about 29)
constexpr size_t kLdsBudget = 160 * 1024;

// lane TT of every quad (EPS = 4) or pair (EPS = 2) to all its lanes; v_mov_b32_dpp with bound_ctrl, so that no
// `old` value has to be materialised (every lane has a valid source)
template <int EPS, int TT> __device__ __forceinline__ unsigned step_bcast(unsigned x)
{
    if constexpr (EPS == 1) return x;
    else if constexpr (EPS == 4)
        return (unsigned)__builtin_amdgcn_mov_dpp((int)x, TT | (TT << 2) | (TT << 4) | (TT << 6), 0xF, 0xF, true);
    else
        return (unsigned)__builtin_amdgcn_mov_dpp((int)x, TT | (TT << 2) | ((2 + TT) << 4) | ((2 + TT) << 6), 0xF, 0xF, true);
}

// FULL: every lane's 16 bytes of a row of H exist and are 16-byte aligned (P a multiple of the slice width, aligned H):
// one 16-byte store per row and lane; otherwise element stores.
template <typename T, int VEC, int LPR, bool FULL>
__global__ __launch_bounds__(kThreads) void xw_sparse_lds_kernel(
    int n_rows, int n_feat, int m_fea, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const T *__restrict__ val, const T *__restrict__ W, int64_t ldw, int w_vec, T *__restrict__ H, int64_t ldh,
    int n_split, unsigned nnz_bytes_col, sgx_epilogue ep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int CP = LPR * VEC;              // columns per slice
    constexpr unsigned ROWB = LPR * 16;        // bytes of one W row in LDS
    constexpr int RPW = 64 / LPR;              // rows of X per sub-tile
    constexpr int SUBS = LPR;                  // sub-tiles per 64-row window
    constexpr int EPS = LPR >= 4 ? 4 : LPR;    // entries per step and group
    constexpr int D = kDepth;

    // workgroup -> (slice, stream of row tiles); the S slices of a stream sit on block ids 8 apart
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int streams_per_xcd = (int)(gridDim.x >> 3) / n_split;
    if (q >= streams_per_xcd * n_split) return;
    const int split = q % n_split;
    const int stream = (q / n_split) * 8 + xcd;
    const int n_streams = streams_per_xcd * 8;
    const int c_base = split * CP;

    // ---- the slice of W into LDS (rows [0, m_fea), then one zero row) ----
    if (w_vec) {
        // kFill chunks per thread are requested before the first is stored (one at a time is a round trip to L2 per
        // chunk, six in a row at the Cora shape); out-of-range offsets where there is nothing to load (the zero row,
        // chunks past the last row of the slice): zeros, and the same number of requests on every path
        constexpr int kFill = 6;
        const int total = (m_fea + 1) * LPR;
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T *>(W), 0, (unsigned)(((int64_t)(m_fea - 1) * ldw + ldw) * (int64_t)sizeof(T)), 0x00020000);
        for (int i0 = threadIdx.x; i0 < total; i0 += kThreads * kFill) {
            u32x4 v[kFill];
#pragma unroll
            for (int f = 0; f < kFill; ++f) {
                const int i = i0 + kThreads * f;
                const int r = i / LPR, c0 = c_base + (i % LPR) * VEC;
                const bool in = i < total && r < m_fea && c0 + VEC <= ldw;
                v[f] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, in ? (unsigned)(((int64_t)r * ldw + c0) * (int64_t)sizeof(T)) : kOOB, 0, 0);
            }
#pragma unroll
            for (int f = 0; f < kFill; ++f) {
                const int i = i0 + kThreads * f;
                if (i < total) *reinterpret_cast<u32x4 *>(lds + (size_t)i * 16) = v[f];
            }
        }
    } else {
        for (int i = threadIdx.x; i < (m_fea + 1) * LPR; i += kThreads) {
            const int r = i / LPR, c0 = c_base + (i % LPR) * VEC;
            union { u32x4 v; T e[VEC]; } u;
            u.v = u32x4{0u, 0u, 0u, 0u};
            if (r < m_fea) {
                const T *src = W + (int64_t)r * ldw + c0;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (c0 + j < n_feat) u.e[j] = src[j];
            }
            *reinterpret_cast<u32x4 *>(lds + (size_t)i * 16) = u.v;
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // provably wave-uniform: scalar control flow below
    const int sub = lane % LPR, grp = lane / LPR;
    const int esub = sub % EPS;                          // the entry of a step this lane loads
    const unsigned my_off = (unsigned)sub * 16u;
    const int col0 = c_base + sub * VEC;
    const unsigned zero_col = (unsigned)m_fea;
    const int64_t n_windows = ((int64_t)n_rows + 63) / 64;
    const int64_t win_step = (int64_t)n_streams * kWaves;        // a tile = kWaves consecutive windows, one per wavefront

    struct Window {                     // per lane: the row of sorted rank `lane` in this window
        int64_t base;                   // first row of the window
        int e0, e1, src, steps;
    };
    struct Meta {                       // one row per lane group (no padding: the struct is copied member by member)
        int64_t r;
        int live;
        int e0, e1;
        int pad_;
    };
    struct Entries {                    // ring of D steps: the (column, value) this lane loaded for each
        unsigned c[D];
        T a[D];
    };

    // Loads without divergent branches (a conditional load makes hipcc wait for it at the join, which would
    // serialise the D requests of a row): row pointers through a clamped index, entries through buffer loads whose
    // offset is out of range past the end of the row (returns 0, no memory access); the select to the zero row of
    // the LDS tile happens when the value is used.
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(col), 0, nnz_bytes_col, 0x00020000);
    const __amdgpu_buffer_rsrc_t val_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(val), 0, (unsigned)(nnz_bytes_col / 4 * sizeof(T)), 0x00020000);
    const unsigned h_pitch_bytes = (unsigned)ldh * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t h_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(H, 0, (unsigned)((int64_t)(n_rows - 1) * ldh * (int64_t)sizeof(T)) + (unsigned)n_feat * (unsigned)sizeof(T), 0x00020000);

    // the row pointers of window w: lane l takes row 64 w + l (requested one window ahead)
    auto load_window = [&](int64_t w, int &e0, int &e1) {
        const int64_t row = w * 64 + lane;
        const int64_t rc = row < n_rows ? row : (int64_t)n_rows - 1;
        const int a = rowptr[rc], b = rowptr[rc + 1];
        e0 = a;
        e1 = row < n_rows ? b : a;
    };
    // sorts the 64 rows of a window by their step count, longest first (ties in row order): a bitonic network on
    // keys (0xFFFFF - steps) : lane
    auto sort_window = [&](int64_t w, int e0, int e1, Window &win) {
        int steps = (e1 - e0 + EPS - 1) / EPS;
        steps = steps > 0xFFFFF ? 0xFFFFF : steps;
        unsigned key = ((unsigned)(0xFFFFF - steps) << 6) | (unsigned)lane;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned other = (unsigned)__shfl_xor((int)key, j);
                const bool up = (lane & k) == 0 || k == 64;
                const bool lower = (lane & j) == 0;
                const unsigned lo = key < other ? key : other, hi = key < other ? other : key;
                key = (lower == up) ? lo : hi;
            }
        }
        win.base = w * 64;
        win.src = (int)(key & 63u);
        win.steps = 0xFFFFF - (int)(key >> 6);
        win.e0 = __shfl(e0, win.src);
        win.e1 = __shfl(e1, win.src);
    };
    // sub-tile t of a window: group g takes the row of rank t * RPW + g; nsteps = the sub-tile's longest row
    auto sub_meta = [&](const Window &win, int t, Meta &m, int &nsteps) {
        const int p = t * RPW + grp;
        m.e0 = __shfl(win.e0, p);
        m.e1 = __shfl(win.e1, p);
        m.r = win.base + __shfl(win.src, p);
        m.live = m.r < n_rows;
        nsteps = __builtin_amdgcn_readlane(win.steps, t * RPW);
    };
    // One (column, value) request.  `wanted` is wave-uniform: when false the address arithmetic is skipped and the
    // loads go out of range -- they are still ISSUED, so that the number of vector-memory operations between a
    // request and its use is the same on every path (hipcc's s_waitcnt counts stay exact: with a varying count it
    // falls back to waiting for everything, the newest requests included, which is the prefetch undone).
    // select_now = false leaves the raw column in c (0 past the end of the row) for the caller to redirect when it
    // uses the slot: inside the step loop an immediate select would wait for the load it belongs to.
    auto fetch = [&](bool wanted, int idx, int e1, unsigned &c, T &a, bool select_now = true) {
        unsigned off = kOOB;
        bool ok = false;
        if (wanted) {
            ok = idx < e1;
            off = ok ? (unsigned)idx * 4u : kOOB;
        }
        const unsigned cc = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, off, 0, 0);
        if constexpr (sizeof(T) == 2) {
            // kOOB / 2 is out of range for the value buffer too (nnz < 2^30)
            const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(val_rsrc, off >> 1, 0, 0);
            a = __builtin_bit_cast(T, h);                 // 0 past the end of the row
        } else {
            a = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, off, 0, 0));
        }
        c = (ok || !select_now) ? cc : zero_col;
    };
    // A UNIT of work = D consecutive steps of one sub-tile: chunk k covers steps [k D, k D + D).  Almost every
    // sub-tile is one unit; rows over D steps continue in further units of the same sub-tile with the sums carried
    // in registers.  Every unit runs the same straight-line code with the same number of memory operations.
    auto issue_entries = [&](const Meta &m, int k, int nsteps, Entries &en) {
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(k * D + d < nsteps, m.e0 + (k * D + d) * EPS + esub, m.e1, en.c[d], en.a[d]);
    };

    float acc[VEC];
    // the sums of one unit; after a sub-tile's last unit its rows are stored
    auto compute = [&](const Meta &cur, int k, int nsteps, Entries &en) {
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (k * D + d >= nsteps) break;
            const unsigned rr = en.c[d] * ROWB;
            const unsigned a = __builtin_bit_cast(unsigned, Elem<T>::to_f32(en.a[d]));
            u32x4 raw[EPS];
            float aa[EPS];
            if constexpr (EPS == 4) {
                raw[0] = *reinterpret_cast<const u32x4 *>(lds + step_bcast<4, 0>(rr) + my_off);
                raw[1] = *reinterpret_cast<const u32x4 *>(lds + step_bcast<4, 1>(rr) + my_off);
                raw[2] = *reinterpret_cast<const u32x4 *>(lds + step_bcast<4, 2>(rr) + my_off);
                raw[3] = *reinterpret_cast<const u32x4 *>(lds + step_bcast<4, 3>(rr) + my_off);
                aa[0] = __builtin_bit_cast(float, step_bcast<4, 0>(a));
                aa[1] = __builtin_bit_cast(float, step_bcast<4, 1>(a));
                aa[2] = __builtin_bit_cast(float, step_bcast<4, 2>(a));
                aa[3] = __builtin_bit_cast(float, step_bcast<4, 3>(a));
            } else {
                raw[0] = *reinterpret_cast<const u32x4 *>(lds + step_bcast<2, 0>(rr) + my_off);
                raw[1] = *reinterpret_cast<const u32x4 *>(lds + step_bcast<2, 1>(rr) + my_off);
                aa[0] = __builtin_bit_cast(float, step_bcast<2, 0>(a));
                aa[1] = __builtin_bit_cast(float, step_bcast<2, 1>(a));
            }
#pragma unroll
            for (int t = 0; t < EPS; ++t) Fma<T, VEC>::run(acc, aa[t], raw[t]);
        }
        // The stores go through a buffer resource over H with an out-of-range offset for lanes that have nothing to
        // store (rows past the end; units that are not their sub-tile's last): the same number of store instructions
        // on every path (see fetch).
        const bool last = (k + 1) * D >= nsteps;
        T out[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = finish_value<T>(acc[i], 0, ep);
        const unsigned row_off = (unsigned)cur.r * h_pitch_bytes + (unsigned)col0 * (unsigned)sizeof(T);
        if constexpr (FULL) {
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(out), h_rsrc, (last && cur.live) ? row_off : kOOB, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const unsigned off = (last && cur.live && col0 + i < n_feat) ? row_off + (unsigned)i * (unsigned)sizeof(T) : kOOB;
                if constexpr (sizeof(T) == 2)
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, out[i]), h_rsrc, off, 0, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, out[i]), h_rsrc, off, 0, 0);
            }
        }
    };

    int64_t w = (int64_t)stream * kWaves + wave;
    if (w >= n_windows) return;
    int e0_raw, e1_raw;                   // the next window's row pointers, in flight
    Window win;
    load_window(w, e0_raw, e1_raw);
    sort_window(w, e0_raw, e1_raw, win);
    load_window(w + win_step, e0_raw, e1_raw);
    Meta m_cur, m_nxt;
    int n_cur, n_nxt, k_cur = 0, k_nxt, t = 0;
    Entries en0, en1;
    sub_meta(win, 0, m_cur, n_cur);
    issue_entries(m_cur, 0, n_cur, en0);
    // One unit per call: the next unit's entries are requested (the same sub-tile's next chunk, the window's next
    // sub-tile, or the first sub-tile of the next window, whose row pointers arrived earlier and are sorted now), then
    // the current unit is summed.  The two entry rings alternate.  Returns false after the stream's last unit.
    auto advance = [&](Entries &en_cur, Entries &en_nxt) -> bool {
        bool more = true;
        if ((k_cur + 1) * D < n_cur) {
            m_nxt = m_cur;
            n_nxt = n_cur;
            k_nxt = k_cur + 1;
        } else if (t + 1 < SUBS) {
            ++t;
            k_nxt = 0;
            sub_meta(win, t, m_nxt, n_nxt);
        } else {
            t = 0;
            k_nxt = 0;
            w += win_step;
            more = w < n_windows;
            // (a window past the end sorts clamped row pointers: every row empty, nothing requested, nothing stored)
            sort_window(w, e0_raw, e1_raw, win);
            sub_meta(win, 0, m_nxt, n_nxt);
        }
        // requested on every call (mostly re-reads of cached lines): a fixed number of loads per call
        load_window(w + win_step, e0_raw, e1_raw);
        issue_entries(m_nxt, k_nxt, n_nxt, en_nxt);
        compute(m_cur, k_cur, n_cur, en_cur);
        m_cur = m_nxt;
        n_cur = n_nxt;
        k_cur = k_nxt;
        return more;
    };
    while (true) {
        if (!advance(en0, en1)) break;
        if (!advance(en1, en0)) break;
    }
}

int device_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
            n = 256;
        cus = n;
    }
    return cus;
}

template <typename T, int VEC, int LPR, bool FULL>
int launch_lds_impl(int n_work, int n_feat, int m_fea, const int32_t *rowptr, const int32_t *col, const void *val, const void *W,
               int64_t ldw, void *H, int64_t ldh, int64_t nnz, sgx_epilogue ep, hipStream_t stream)
{
    auto kernel = xw_sparse_lds_kernel<T, VEC, LPR, FULL>;
    const size_t lds_bytes = (size_t)(m_fea + 1) * LPR * 16;
    static bool attr_set = false;                     // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kLdsBudget) != hipSuccess)
            return SGX_ERR_HIP;
        attr_set = true;
    }
    const int n_split = (n_feat + LPR * VEC - 1) / (LPR * VEC);
    const int wgs_per_cu = lds_bytes * 2 <= kLdsBudget ? 2 : 1;       // 32 wavefronts per CU at most
    int grid = device_cus() / 8 * 8 * wgs_per_cu;
    // no more streams than row tiles
    const int64_t tiles = ((int64_t)n_work + 64 * kWaves - 1) / (64 * kWaves);
    const int64_t want = (tiles + 7) / 8 * 8 * n_split;
    if (want < grid) grid = (int)want;
    if (grid < 8 * n_split) grid = 8 * n_split;
    const int w_vec = ((uintptr_t)W % 16 == 0) && ((ldw * (int64_t)sizeof(T)) % 16 == 0);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds_bytes, stream, n_work, n_feat, m_fea, rowptr, col,
                       (const T *)val, (const T *)W, ldw, w_vec, (T *)H, ldh, n_split, (unsigned)(nnz * 4), ep);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

template <typename T, int VEC, int LPR>
int launch_lds(int n_work, int n_feat, int m_fea, const int32_t *rowptr, const int32_t *col, const void *val, const void *W,
               int64_t ldw, void *H, int64_t ldh, int64_t nnz, sgx_epilogue ep, hipStream_t stream)
{
    const bool full = n_feat % (LPR * VEC) == 0 && ((uintptr_t)H % 16 == 0) && ((ldh * (int64_t)sizeof(T)) % 16 == 0);
    return full ? launch_lds_impl<T, VEC, LPR, true>(n_work, n_feat, m_fea, rowptr, col, val, W, ldw, H, ldh, nnz, ep, stream)
                : launch_lds_impl<T, VEC, LPR, false>(n_work, n_feat, m_fea, rowptr, col, val, W, ldw, H, ldh, nnz, ep, stream);
}

// lanes per row of the LDS tile: the widest slice (power-of-two lanes x 16 bytes) whose (M_fea + 1) rows fit
int choose_lpr(int dtype, int m_fea, int n_feat)
{
    const int per16 = dtype == SGX_F16 ? 8 : 4;
    int need = sgx_next_pow2((n_feat + per16 - 1) / per16);        // lanes that cover all of P
    if (need > 16) need = 16;
    int lpr = need;
    while (lpr >= 1 && (size_t)(m_fea + 1) * lpr * 16 > kLdsBudget) lpr >>= 1;
    return lpr >= 2 ? lpr : 0;                                      // 0: not even a 32-byte slice fits (one-lane rows stay with the gather kernel)
}

}  // namespace

// Whether the LDS form applies, and its launch.  Kept to matrices large enough to pay for every workgroup's copy
// of its slice (nnz from 2^20), to rows the plan does not cut (the split path of the gather kernel keeps those),
// and to at most 4 slices (the CSR of X is read once per slice, the re-reads out of L2).
bool sgx_xw_sparse_lds_applicable(int dtype, int n_rows, int m_fea, int n_feat, int64_t ldh, const sgx_plan *plan)
{
    if (sgx_tune().xw_sparse_no_lds) return false;         // tuning override (tools/xw_sparse_probe.py)
    if (!plan || plan->n_tasks > 0 || plan->nnz < ((int64_t)1 << 20) || plan->nnz >= ((int64_t)1 << 30) || n_rows < 4096)
        return false;                                 // (32-bit buffer offsets into columnIndex: nnz x 4 bytes below 4 GiB)
    if ((unsigned long long)n_rows * (unsigned long long)ldh * (dtype == SGX_F16 ? 2ull : 4ull) >= 0xFFF00000ull)
        return false;                                 // H is stored through 32-bit buffer offsets too
    const int lpr = choose_lpr(dtype, m_fea, n_feat);
    if (lpr < 1) return false;
    const int per16 = dtype == SGX_F16 ? 8 : 4;
    const int n_split = (n_feat + lpr * per16 - 1) / (lpr * per16);
    return n_split <= 4;
}

int sgx_xw_sparse_lds(int dtype, int n_rows, int m_fea, int n_feat, const int32_t *rowPtr, const int32_t *columnIndex,
                      const void *values, const void *W, int64_t ldw, void *H, int64_t ldh, const sgx_plan *plan,
                      sgx_epilogue ep, hipStream_t stream)
{
    const int n_work = n_rows;            // the kernel orders rows itself (inside windows of 64): the plan's order is not used
    const int lpr = choose_lpr(dtype, m_fea, n_feat);
#define SGX_LDS_CASE(L)                                                                                                     \
    case L:                                                                                                                 \
        return dtype == SGX_F16 ? launch_lds<f16, 8, L>(n_work, n_feat, m_fea, rowPtr, columnIndex, values, W, ldw, H, ldh, \
                                                        plan->nnz, ep, stream)                                                \
                                : launch_lds<float, 4, L>(n_work, n_feat, m_fea, rowPtr, columnIndex, values, W, ldw, H,    \
                                                          ldh, plan->nnz, ep, stream);
    switch (lpr) {
        SGX_LDS_CASE(2)
        SGX_LDS_CASE(4)
        SGX_LDS_CASE(8)
        SGX_LDS_CASE(16)
    default: return SGX_ERR_UNSUPPORTED;
    }
#undef SGX_LDS_CASE
}
