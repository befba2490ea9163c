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
// the 64 rows of a window are therefore dealt to the sub-tiles by length, longest first -- the sblock idea of
// the reference (rows grouped per pipelined loop, K.cpp:826-845) with the grouping chosen by length.  The
// order comes from the plan (sgx_plan::win_order, one byte per row, built once per matrix; round 2 sorted
// every window inside this kernel, a bitonic network that was 7 % of its vector instructions).  The measured
// kernel is bound by vector-instruction issue, not by LDS or HBM, which is why the wasted steps and every
// instruction beside the fmas matter.
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
#ifndef SGX_XW_LDS_SKIP_UNUSED
#define SGX_XW_LDS_SKIP_UNUSED 1
#endif
#ifndef SGX_XW_LDS_PAIRS
#define SGX_XW_LDS_PAIRS 0          // 1: entries requested two per lane and instruction (half the vector-memory instructions;
                                    // measured: the same time, 4 more registers -- tools/sweep_xw_sparse.py)
#endif
// an offset that stays out of range after a step's immediate offset (< 4 KiB) is added to it, for the column indices
// (nnz x 4 bytes) and, halved, for fp16 values (nnz x 2 bytes): the LDS form takes matrices below 2^30 - 2^16 entries
constexpr unsigned kFarOOB = 0xFFFF0000u;
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
    int n_split, unsigned nnz_bytes_col, const uint8_t *__restrict__ win_order, sgx_epilogue ep)
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
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_byte *)lds;        // LDS byte address of the tile (what a ds_read takes)
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
    // PAIRS: a lane requests TWO consecutive entries per instruction (8 bytes of column indices; 8 bytes around its two
    // values) -- lane i of a quad the entries 2 i, 2 i + 1 of every group of 8, i.e. of two steps -- instead of one entry
    // per step: half the vector-memory instructions for the same bytes.  The texture-address unit looks every lane of
    // such an instruction up in the cache, 16 different rows' lines per instruction, and was 67 % busy with one entry
    // per lane; that, not the vector ALU, is what the stage waits for.
    constexpr bool PAIRS = SGX_XW_LDS_PAIRS && EPS == 4 && D % 2 == 0;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    struct Entries {                    // ring of D steps: the (column, value) this lane loaded for each
        unsigned c[PAIRS ? 1 : D];
        T a[PAIRS ? 1 : D];
        u32x2 c2[PAIRS ? D / 2 : 1];    // PAIRS: the two column indices; the two values (fp16: the 8 aligned bytes around them)
        u32x2 a2[PAIRS ? D / 2 : 1];
    };

    // Loads without divergent branches (a conditional load makes hipcc wait for it at the join, which would
    // serialise the D requests of a row): row pointers through a clamped index, entries through buffer loads that
    // run on past the end of the row (into the next rows' entries; past the end of the arrays the range check
    // returns 0); the select to the zero row of the LDS tile and to a zero value happens when a slot is USED.
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(col), 0, nnz_bytes_col, 0x00020000);
    // (fp16 values are requested as whole dwords, and the range check drops a dword that straddles the end: the
    // records are rounded up to whole dwords -- the half behind an odd count lies in the same dword, hence the same page)
    const __amdgpu_buffer_rsrc_t val_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(val), 0, ((unsigned)(nnz_bytes_col / 4 * sizeof(T)) + 3u) & ~3u, 0x00020000);
    const unsigned h_pitch_bytes = (unsigned)ldh * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t h_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(H, 0, (unsigned)((int64_t)(n_rows - 1) * ldh * (int64_t)sizeof(T)) + (unsigned)n_feat * (unsigned)sizeof(T), 0x00020000);

    // The rows of window w in the plan's order (rank `lane` -> row 64 w + src) and their row pointers: two dependent
    // requests, spread over two windows -- the order of window w + 2 steps is requested while the row pointers of
    // window w + 1 step are, whose order arrived a window earlier.
    auto load_order = [&](int64_t w) -> int {
        const int64_t wc = w < n_windows ? w : n_windows - 1;
        return (int)win_order[wc * 64 + lane];
    };
    auto load_rows = [&](int64_t w, int src, int &e0, int &e1) {
        const int64_t row = w * 64 + src;
        const int64_t rc = row < n_rows ? row : (int64_t)n_rows - 1;       // (a window past the end: clamped, then emptied)
        const int a = rowptr[rc], b = rowptr[rc + 1];
        e0 = a;
        e1 = row < n_rows ? b : a;
    };
    auto make_window = [&](int64_t w, int src, int e0, int e1, Window &win) {
        win.base = w * 64;
        win.src = src;
        win.e0 = e0;
        win.e1 = e1;
        win.steps = (e1 - e0 + EPS - 1) / EPS;
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
    // A UNIT of work = D consecutive steps of one sub-tile: chunk k covers steps [k D, k D + D).  Almost every
    // sub-tile is one unit; rows over D steps continue in further units of the same sub-tile with the sums carried
    // in registers.  Every unit runs the same straight-line code with the same number of memory operations
    // (hipcc's s_waitcnt counts stay exact: with a varying count it falls back to waiting for everything, the
    // newest requests included, which is the prefetch undone): all D (column, value) requests of a unit are
    // issued, from ONE base offset with the step as the instruction's immediate offset -- no per-step address or
    // range arithmetic; slots past the end of the row are dealt with when they are used.
    auto issue_entries = [&](const Meta &m, int k, int nsteps, Entries &en) {
        if constexpr (PAIRS) {
            const int e = m.e0 + k * (D * EPS) + 2 * esub;            // this lane's first entry of the unit
            const unsigned cbase = (unsigned)e * 4u;
            // fp16 values: the 8 aligned bytes that hold halves e and e + 1 (a 4-byte load wants dword alignment; the
            // pair is shifted into place when it is used); fp32 values lie like the column indices
            const unsigned abase = sizeof(T) == 2 ? (unsigned)(e >> 1) * 4u : cbase;
#pragma unroll
            for (int j = 0; j < D / 2; ++j) {
                // (the lanes whose pair belongs to a step behind the sub-tile's last one go out of range: issued, not looked up)
                const bool wanted = !SGX_XW_LDS_SKIP_UNUSED || k * D + 2 * j + (esub >> 1) < nsteps;
                en.c2[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(col_rsrc, (wanted ? cbase : kFarOOB) + (unsigned)(j * 32), 0, 0));
                en.a2[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(val_rsrc, (wanted ? abase : kFarOOB) + (unsigned)(j * (sizeof(T) == 2 ? 16 : 32)), 0, 0));
            }
            return;
        }
        const unsigned base = (unsigned)(m.e0 + k * (D * EPS) + esub) * 4u;
#pragma unroll
        for (int d = 0; d < D; ++d) {
#if SGX_XW_LDS_SKIP_UNUSED
            // steps behind the sub-tile's last one (a wave-uniform condition: one select) go out of range: the request is
            // still ISSUED, but the texture-address unit does not look its 64 lanes up in the cache -- with those lookups
            // the unit (67 % busy before) became the limit, which cost more than the per-step address arithmetic saved
            const unsigned off = (k * D + d < nsteps) ? base : kFarOOB;
#else
            const unsigned off = base;
#endif
            en.c[d] = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, off + (unsigned)(d * EPS * 4), 0, 0);
            if constexpr (sizeof(T) == 2) {
                const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(val_rsrc, (off >> 1) + (unsigned)(d * EPS * 2), 0, 0);
                en.a[d] = __builtin_bit_cast(T, h);
            } else {
                en.a[d] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, off + (unsigned)(d * EPS * 4), 0, 0));
            }
        }
    };

    float acc[VEC];
    // the sums of one unit; after a sub-tile's last unit its rows are stored
    auto compute = [&](const Meta &cur, int k, int nsteps, Entries &en) {
        // this lane's slot of step d of the unit lies inside its row while rem > d EPS
        const int rem = cur.e1 - cur.e0 - k * (D * EPS) - esub;
        // One step = the broadcasts and the EPS reads of W rows from LDS (prep), then EPS x VEC fmas.  The reads of step
        // d + 1 are issued BEFORE the fmas of step d.  Every step ends in a wave-uniform branch (the sub-tile's step
        // count), and hipcc sinks loads that are not used before such a branch into the block behind it -- with the reads
        // there each step waited out its own LDS round trip (SQ_WAIT_INST_ANY was 1.9 x the cycles that issued vector
        // instructions).  So the reads are inline assembly (never moved), and so is the wait before a step's fmas: it
        // names the step's registers as read-write operands, which keeps every consumer behind it, and counts what may
        // stay outstanding -- the EPS reads of the NEXT step (LDS operations return in order; operations the compiler
        // issues in between only make the count stricter; there is no scalar load inside the loop).  Two named register
        // sets, indexed by the unrolled step's parity.
        auto lds_read = [&](u32x4 &dst, unsigned byte_off) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_base + byte_off) : "memory");
        };
        // PAIRS: what this lane's two entries of the current step pair become (LDS row offsets, fp32 values)
        unsigned pr_rr[2] = {0u, 0u}, pr_a[2] = {0u, 0u};
        const int remp = cur.e1 - cur.e0 - k * (D * EPS) - 2 * esub;
        const unsigned half_shift = (unsigned)((cur.e0 + k * (D * EPS) + 2 * esub) & 1) * 16u;
        auto prep = [&](int d, u32x4 (&raw)[EPS], float (&aa)[EPS]) {
            if constexpr (PAIRS) {
                const int j = d >> 1;
                if ((d & 1) == 0) {
                    const bool ok0 = remp > 8 * j, ok1 = remp > 8 * j + 1;
                    pr_rr[0] = (ok0 ? en.c2[j][0] : zero_col) * ROWB;
                    pr_rr[1] = (ok1 ? en.c2[j][1] : zero_col) * ROWB;
                    if constexpr (sizeof(T) == 2) {
                        const unsigned w = __builtin_amdgcn_alignbit(en.a2[j][1], en.a2[j][0], half_shift);    // halves e, e + 1
                        const f16 h0 = __builtin_bit_cast(f16, (unsigned short)(w & 0xFFFFu)), h1 = __builtin_bit_cast(f16, (unsigned short)(w >> 16));
                        pr_a[0] = __builtin_bit_cast(unsigned, ok0 ? (float)h0 : 0.0f);
                        pr_a[1] = __builtin_bit_cast(unsigned, ok1 ? (float)h1 : 0.0f);
                    } else {
                        pr_a[0] = ok0 ? en.a2[j][0] : 0u;
                        pr_a[1] = ok1 ? en.a2[j][1] : 0u;
                    }
                }
                // step d sums entries 4 d .. 4 d + 3 of the unit: the pairs of lanes 2 (d & 1) and 2 (d & 1) + 1 of the quad
                if ((d & 1) == 0) {
                    lds_read(raw[0], step_bcast<4, 0>(pr_rr[0]) + my_off);
                    lds_read(raw[1], step_bcast<4, 0>(pr_rr[1]) + my_off);
                    lds_read(raw[2], step_bcast<4, 1>(pr_rr[0]) + my_off);
                    lds_read(raw[3], step_bcast<4, 1>(pr_rr[1]) + my_off);
                    aa[0] = __builtin_bit_cast(float, step_bcast<4, 0>(pr_a[0]));
                    aa[1] = __builtin_bit_cast(float, step_bcast<4, 0>(pr_a[1]));
                    aa[2] = __builtin_bit_cast(float, step_bcast<4, 1>(pr_a[0]));
                    aa[3] = __builtin_bit_cast(float, step_bcast<4, 1>(pr_a[1]));
                } else {
                    lds_read(raw[0], step_bcast<4, 2>(pr_rr[0]) + my_off);
                    lds_read(raw[1], step_bcast<4, 2>(pr_rr[1]) + my_off);
                    lds_read(raw[2], step_bcast<4, 3>(pr_rr[0]) + my_off);
                    lds_read(raw[3], step_bcast<4, 3>(pr_rr[1]) + my_off);
                    aa[0] = __builtin_bit_cast(float, step_bcast<4, 2>(pr_a[0]));
                    aa[1] = __builtin_bit_cast(float, step_bcast<4, 2>(pr_a[1]));
                    aa[2] = __builtin_bit_cast(float, step_bcast<4, 3>(pr_a[0]));
                    aa[3] = __builtin_bit_cast(float, step_bcast<4, 3>(pr_a[1]));
                }
                return;
            }
            // a slot past the end of the row holds another row's entry (or 0 past the arrays): the zero row of the
            // tile and a zero value, so that a non-finite neighbour never leaks
            const bool ok = rem > d * EPS;
            const unsigned rr = (ok ? en.c[d] : zero_col) * ROWB;
            const unsigned a = __builtin_bit_cast(unsigned, Elem<T>::to_f32(ok ? en.a[d] : (T)0));
            if constexpr (EPS == 4) {
                lds_read(raw[0], step_bcast<4, 0>(rr) + my_off);
                lds_read(raw[1], step_bcast<4, 1>(rr) + my_off);
                lds_read(raw[2], step_bcast<4, 2>(rr) + my_off);
                lds_read(raw[3], step_bcast<4, 3>(rr) + my_off);
                aa[0] = __builtin_bit_cast(float, step_bcast<4, 0>(a));
                aa[1] = __builtin_bit_cast(float, step_bcast<4, 1>(a));
                aa[2] = __builtin_bit_cast(float, step_bcast<4, 2>(a));
                aa[3] = __builtin_bit_cast(float, step_bcast<4, 3>(a));
            } else {
                lds_read(raw[0], step_bcast<2, 0>(rr) + my_off);
                lds_read(raw[1], step_bcast<2, 1>(rr) + my_off);
                aa[0] = __builtin_bit_cast(float, step_bcast<2, 0>(a));
                aa[1] = __builtin_bit_cast(float, step_bcast<2, 1>(a));
            }
        };
        // waits until at most `pending` LDS operations are outstanding; the registers of the step about to be summed
        // pass through it
        auto arrive = [&](u32x4 (&raw)[EPS], bool next_in_flight) {
            if constexpr (EPS == 4) {
                if (next_in_flight) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]) :: "memory");
            } else {
                if (next_in_flight) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(raw[0]), "+v"(raw[1]) :: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]) :: "memory");
            }
        };
        u32x4 raw_even[EPS], raw_odd[EPS];
        float aa_even[EPS], aa_odd[EPS];
        prep(0, raw_even, aa_even);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (k * D + d >= nsteps) break;
            u32x4 (&raw)[EPS] = (d & 1) ? raw_odd : raw_even;
            float (&aa)[EPS] = (d & 1) ? aa_odd : aa_even;
            // (requested for the step behind the sub-tile's last one too: it reads the zero row and is never summed)
            if (d + 1 < D) prep(d + 1, (d & 1) ? raw_even : raw_odd, (d & 1) ? aa_even : aa_odd);
            arrive(raw, d + 1 < D);
            // the first entry of a row STARTS the sums (fma onto a literal +0: the same value as an fma onto a zeroed
            // register, without the eight moves that zero it)
            if (d == 0 && k == 0) FmaInit<T, VEC>::run(acc, aa[0], raw[0]);
            else Fma<T, VEC>::run(acc, aa[0], raw[0]);
#pragma unroll
            for (int t = 1; t < EPS; ++t) Fma<T, VEC>::run(acc, aa[t], raw[t]);
        }
        // reads still in flight when the unit ends early (the step behind the last one): they must land before their
        // registers are reused
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // a sub-tile without a step (64 empty rows): zeros
        if (nsteps == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
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
    Window win;
    int src_nxt;                          // the order of window w + 1 step (arrived)
    {
        const int src0 = load_order(w);
        src_nxt = load_order(w + win_step);
        int a0, b0;
        load_rows(w, src0, a0, b0);
        make_window(w, src0, a0, b0, win);
    }
    int src_raw = load_order(w + 2 * win_step);      // in flight: the order of window w + 2 steps ...
    int e0_raw, e1_raw;                              // ... and the row pointers of window w + 1 step
    load_rows(w + win_step, src_nxt, e0_raw, e1_raw);
    Meta m_cur, m_nxt;
    int n_cur, n_nxt, k_cur = 0, k_nxt, t = 0;
    Entries en0, en1;
    sub_meta(win, 0, m_cur, n_cur);
    issue_entries(m_cur, 0, n_cur, en0);
    // One unit per call: the next unit's entries are requested (the same sub-tile's next chunk, the window's next
    // sub-tile, or the first sub-tile of the next window, whose row pointers arrived earlier), then the current unit
    // is summed.  The two entry rings alternate.  Returns false after the stream's last unit.
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
            // (a window past the end has clamped row pointers: every row empty, nothing stored)
            make_window(w, src_nxt, e0_raw, e1_raw, win);
            src_nxt = src_raw;
            sub_meta(win, 0, m_nxt, n_nxt);
        }
        // requested on every call (mostly re-reads of cached lines): a fixed number of loads per call
        src_raw = load_order(w + 2 * win_step);
        load_rows(w + win_step, src_nxt, e0_raw, e1_raw);
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
               int64_t ldw, void *H, int64_t ldh, int64_t nnz, const uint8_t *win_order, sgx_epilogue ep, hipStream_t stream)
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
                       (const T *)val, (const T *)W, ldw, w_vec, (T *)H, ldh, n_split, (unsigned)(nnz * 4), win_order, ep);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

template <typename T, int VEC, int LPR>
int launch_lds(int n_work, int n_feat, int m_fea, const int32_t *rowptr, const int32_t *col, const void *val, const void *W,
               int64_t ldw, void *H, int64_t ldh, int64_t nnz, const uint8_t *win_order, sgx_epilogue ep, hipStream_t stream)
{
    const bool full = n_feat % (LPR * VEC) == 0 && ((uintptr_t)H % 16 == 0) && ((ldh * (int64_t)sizeof(T)) % 16 == 0);
    return full ? launch_lds_impl<T, VEC, LPR, true>(n_work, n_feat, m_fea, rowptr, col, val, W, ldw, H, ldh, nnz, win_order, ep, stream)
                : launch_lds_impl<T, VEC, LPR, false>(n_work, n_feat, m_fea, rowptr, col, val, W, ldw, H, ldh, nnz, win_order, ep, stream);
}

// lanes per row of the LDS tile: the widest slice (power-of-two lanes x 16 bytes) whose (M_fea + 1) rows fit
int choose_lpr(int dtype, int m_fea, int n_feat)
{
    const int per16 = dtype == SGX_F16 ? 8 : 4;
    int need = sgx_next_pow2((n_feat + per16 - 1) / per16);        // lanes that cover all of P
    if (need > 16) need = 16;
    int lpr = need;
    while (lpr >= 1 && (size_t)(m_fea + 1) * lpr * 16 > kLdsBudget) lpr >>= 1;
    if (sgx_tune().xw_sparse_lpr > 0 && sgx_tune().xw_sparse_lpr < lpr) lpr = sgx_tune().xw_sparse_lpr;      // tuning override: narrower slices
    return lpr >= 2 ? lpr : 0;                                      // 0: not even a 32-byte slice fits (one-lane rows stay with the gather kernel)
}

}  // namespace

// Whether the LDS form applies, and its launch.  Kept to matrices large enough to pay for every workgroup's copy
// of its slice (nnz from 2^20), to rows the plan does not cut (the split path of the gather kernel keeps those),
// and to at most 4 slices (the CSR of X is read once per slice, the re-reads out of L2).
bool sgx_xw_sparse_lds_applicable(int dtype, int n_rows, int m_fea, int n_feat, int64_t ldh, const sgx_plan *plan)
{
    if (sgx_tune().xw_sparse_no_lds) return false;         // tuning override (tools/xw_sparse_probe.py)
    if (!plan || !plan->win_order || plan->n_tasks > 0 || plan->nnz < ((int64_t)1 << 20) || plan->nnz >= ((int64_t)1 << 30) - 65536 || n_rows < 4096)
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
    const int n_work = n_rows;            // rows are dealt by length inside windows of 64 (plan->win_order), not in the plan's degree order
    const int lpr = choose_lpr(dtype, m_fea, n_feat);
#define SGX_LDS_CASE(L)                                                                                                     \
    case L:                                                                                                                 \
        return dtype == SGX_F16 ? launch_lds<f16, 8, L>(n_work, n_feat, m_fea, rowPtr, columnIndex, values, W, ldw, H, ldh, \
                                                        plan->nnz, plan->win_order, ep, stream)                               \
                                : launch_lds<float, 4, L>(n_work, n_feat, m_fea, rowPtr, columnIndex, values, W, ldw, H,    \
                                                          ldh, plan->nnz, plan->win_order, ep, stream);
    switch (lpr) {
        SGX_LDS_CASE(2)
        SGX_LDS_CASE(4)
        SGX_LDS_CASE(8)
        SGX_LDS_CASE(16)
    default: return SGX_ERR_UNSUPPORTED;
    }
#undef SGX_LDS_CASE
}
