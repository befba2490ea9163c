// Dense X . W for a long K (fp16, M_fea > 128) with the WHOLE weight tile resident in LDS -- loop_fea / compute1 in
// gemm_mode 1 with B_accel as the reference has it: on chip for the entire pass over X (K.cpp:3038-3051, :2605).
//
// Why (round 2): the 128 x 128 tile kernel (xw_dense.hip) stages X AND W per k-step through LDS with one step of
// global loads in flight per workgroup: 3 workgroups x 8 KB of X per CU = 6 MB on the chip, where 8 TB/s x ~2 us of
// latency wants 16 MB -- Reddit's 602 -> 128 ran at 2.4 TB/s (0.30 of the HBM roofline).  Here a workgroup of 8
// wavefronts copies its column block of W^T (all of K: 128 x 602 halves = 154 KB) into LDS once and then only streams
// X: every wavefront owns 32-row tiles, loads the MFMA B fragments of X straight from HBM into a ring of kRing PAIRS of
// k-steps (16 bytes per lane and load, 128 bytes of a row per pair: 16 KB in flight per wavefront, 128 KB per CU), reads
// the A fragments of W from LDS, and never meets a barrier after the fill.  X is read once, H written once -- its tiles
// leave as 16-byte pieces spread over the k-steps (see the step lambda) so that the stores never queue up behind the
// ring's loads.
//
// Orientation as in xw_dense.hip: H^T[n][m] = sum_k Wt[n][k] X[m][k], A := rows of W^T (LDS), B := rows of X, so a
// lane ends with 4 consecutive columns of one row of H; k-steps of 32 in ascending order into v_mfma_f32_16x16x32_f16
// -- the same sums in the same order as the tile kernel, hence the same bits.
//
// LDS rows are K_pad + 8 halves: (K_pad + 8) / 8 is odd, so the 16-byte fragment reads of 16 consecutive rows fall
// into 16 different 16-byte bank groups.  k >= M_fea is zero in LDS; the X fragment of a row's last k-step reaches
// into the next row, so its lanes past M_fea are masked (0 x NaN would not be 0).  X goes through a buffer resource:
// rows past the end read as zero and cost no memory access, and the ring issues the same number of loads on every path.
#include "sgx_device.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kWldsThreads = 512;          // 8 wavefronts: two per SIMD, 256 VGPRs each
constexpr int kWldsWaves = kWldsThreads / 64;
#ifndef SGX_WLDS_RING
#define SGX_WLDS_RING 4
#endif
constexpr int kRing = SGX_WLDS_RING;       // pairs of k-steps of X in flight per wavefront (16 loads, 4 KB)
constexpr int kMT = 2;                     // 16-row tiles of X per wavefront tile
constexpr size_t kWldsLds = 160 * 1024;

typedef f16x8 f16x8_u2 __attribute__((aligned(2)));
typedef f32x4 f32x4_u4 __attribute__((aligned(4)));

__device__ __forceinline__ f16 relu_half(f16 v, int relu) { return (!relu || v > (f16)0) ? v : (f16)0; }

template <int NT>
__global__ __launch_bounds__(kWldsThreads) void xw_dense_wlds_f16_kernel(
    int n_rows, int M, int P, const f16 *__restrict__ X, int64_t ldx, const f16 *__restrict__ Wt, int64_t ldw,
    f16 *__restrict__ H, int64_t ldh, int relu, int n_cb, int LP, int KS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    f16 *sW = reinterpret_cast<f16 *>(lds_raw);                 // [16 NT][LP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;

    // workgroup -> (column block, stream of row tiles); the column blocks of a stream sit on block ids 8 apart (one XCD:
    // the second one's X comes out of that L2)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int cb = q % n_cb;
    const int stream = (q / n_cb) * 8 + xcd;
    const int n_streams = (int)(gridDim.x >> 3) / n_cb * 8;
    const int col_base = cb * 16 * NT;

    const int64_t n_tiles = ((int64_t)n_rows + 16 * kMT - 1) / (16 * kMT);
    const int64_t tile_step = (int64_t)n_streams * kWldsWaves;

    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16 *>(X), 0, (unsigned)((int64_t)n_rows * ldx * 2), 0x00020000);
    const unsigned x_pitch = (unsigned)ldx * 2u;
    const unsigned lane_off = (unsigned)l15 * x_pitch + (unsigned)lq * 16u;

    // The unit of the ring is a PAIR of k-steps -- 128 bytes of each of the 16 rows, requested by two loads back to
    // back, so that a row's 128-byte line is fetched once (with one k-step per unit the second half of every line was
    // asked for a step later, by when 3 MB of requests per XCD had gone through its 4 MB L2: 4.1 TB/s of X at best, and
    // 2.6 with loads that bypass L2).  The request cursor runs kRing pairs ahead of the compute cursor over the same
    // sequence (tile, pair); an odd KS leaves the last pair's second half out of range (zeros, and its MFMAs skipped).
    const int KS2 = (KS + 1) / 2;
    int64_t it = (int64_t)stream * kWldsWaves + wave;
    int is = 0;
    f16x8 xr[kRing][2][kMT];
    auto issue = [&](f16x8 (&x)[2][kMT]) {
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt) {
            const int64_t row = it * (16 * kMT) + 16 * mt;          // of lane 0
            const bool ok = row + l15 < n_rows;
            const unsigned off = (unsigned)row * x_pitch + lane_off + (unsigned)is * 128u;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                x[h][mt] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         x_rsrc, (ok && 2 * is + h < KS) ? off + 64u * h : kOOB, 0, 0));
        }
        if (++is == KS2) {
            is = 0;
            it += tile_step;
        }
    };

    // ---- the column block of W^T into LDS: rows [col_base, col_base + 16 NT), k in [0, LP), zero outside W ----
    // kFill chunks of 16 bytes per thread are requested before the first is stored (one request at a time would cost a
    // round trip to L2 each: 20 of them for 128 x 602).  Rows past P and chunks past M are out of range for the buffer
    // (zeros); the chunk that straddles M is cut by a mask.
    {
        constexpr int kFill = 8;
        const __amdgpu_buffer_rsrc_t w_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<f16 *>(Wt), 0, (unsigned)(((int64_t)(P - 1) * ldw + M) * 2), 0x00020000);
        const int cpr = LP / 8, total = 16 * NT * cpr;              // 16-byte chunks per LDS row, and in all
        for (int c0 = tid; c0 < total; c0 += kWldsThreads * kFill) {
            u32x4 v[kFill];
            int kk[kFill];
#pragma unroll
            for (int u = 0; u < kFill; ++u) {
                const int c = c0 + kWldsThreads * u;
                const int nl = c / cpr, k = 8 * (c - nl * cpr);
                // LDS row nl = 16 nt + 4 q + i is column 32 (nt / 2) + 8 q + 4 (nt % 2) + i of the block: the MFMA hands
                // lane quad q rows 4 q + i of each tile, so two neighbouring tiles give it 8 consecutive columns of H
                const int n = ((nl >> 5) << 5) | (((nl >> 2) & 3) << 3) | (((nl >> 4) & 1) << 2) | (nl & 3);
                kk[u] = k;
                const bool in = c < total && col_base + n < P && k < M;
                const unsigned off = in ? (unsigned)(((int64_t)(col_base + n) * ldw + k) * 2) : kOOB;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, off, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < kFill; ++u) {
                const int c = c0 + kWldsThreads * u;
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    v[u][d] &= (kk[u] + 2 * d < M ? 0x0000FFFFu : 0u) | (kk[u] + 2 * d + 1 < M ? 0xFFFF0000u : 0u);
                if (c < total) *reinterpret_cast<u32x4 *>(lds_raw + (size_t)c * 16) = v[u];
            }
        }
    }
    __syncthreads();

    // lanes of a row's last k-step that lie past M (they hold the next row's first elements)
    u32x4 tail_mask;
    {
        const int k_last = 32 * (KS - 1) + 8 * lq;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            tail_mask[d] = (k_last + 2 * d < M ? 0x0000FFFFu : 0u) | (k_last + 2 * d + 1 < M ? 0xFFFF0000u : 0u);
    }

    constexpr int NP = kMT * NT / 2;            // 16-byte pieces of a finished tile per lane
    f32x4 acc[kMT][NT];
#pragma unroll
    for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};

    int64_t ct = (int64_t)stream * kWldsWaves + wave;
    int cs = 0;
    const f16 *w_lane = sW + (size_t)l15 * LP + 8 * lq;
    // Tiles this wavefront owns x k-steps (+ the steps the last tile's stores are spread over), rounded up to whole
    // turns of the ring: the loop below is straight-line code with one back edge, kRing steps per turn.  The steps past
    // the end multiply zeros (their loads are out of range) and the tiles they finish lie past the last row.
    const int64_t first = (int64_t)stream * kWldsWaves + wave;
    const int64_t my_tiles = first < n_tiles ? (n_tiles - first + tile_step - 1) / tile_step : 0;
    const int64_t turns = (my_tiles * KS2 + (NP + 3) / 4 + kRing - 1) / kRing;

    // A finished tile is NP pieces of 16 bytes per lane (8 consecutive columns of one row).  They are not stored in a
    // burst: on gfx9 stores count in vmcnt like loads, so a burst of stores behind the ring's loads makes the next
    // step's wait drain the ring.  Four pieces leave per step instead (out of range when none is pending): every step
    // issues the same 4 loads and 4 stores, the waits stay exact, and two neighbouring pieces complete a 128-byte line.
    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(H, 0, (unsigned)((int64_t)n_rows * ldh * 2), 0x00020000);
    const unsigned h_pitch = (unsigned)ldh * 2u;
    u32x4 pend[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) pend[i] = u32x4{0u, 0u, 0u, 0u};
    unsigned pend_base = 0;                      // byte offset of (the pending tile's first row + l15, this lane's 8 columns of pair 0)
    int pend_i = NP;                             // next piece to store; NP = nothing pending
    // column pairs of this block that exist for this lane (the last block may reach past ldh)
    const int t_lim = (int)((ldh - col_base - 8 * lq) / 32) + (((ldh - col_base - 8 * lq) % 32) >= 8 ? 1 : 0);
    constexpr int kOut = NP < 4 ? NP : 4;        // pieces stored per step

#pragma unroll
    for (int d = 0; d < kRing; ++d) issue(xr[d]);
    auto half_step = [&](f16x8 (&x)[kMT], int ks) {
        f16x8 a[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const f16x8 *>(w_lane + (size_t)(16 * nt) * LP + 32 * ks);
        if (ks == KS - 1) {
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt) x[mt] = __builtin_bit_cast(f16x8, __builtin_bit_cast(u32x4, x[mt]) & tail_mask);
        }
#pragma unroll
        for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[nt], x[mt], acc[mt][nt], 0, 0, 0);
    };
    auto step = [&](f16x8 (&x)[2][kMT]) {
        // pending pieces out
#pragma unroll
        for (int j = 0; j < kOut; ++j) {
            const int mt = pend_i / (NT / 2), t = pend_i % (NT / 2);
            const bool ok = pend_i < NP && t < t_lim;
            const unsigned off = ok ? pend_base + (unsigned)mt * 16u * h_pitch + (unsigned)t * 64u : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(pend[j], h_rsrc, off, 0, 0);
            pend_i = pend_i < NP ? pend_i + 1 : NP;
        }
#pragma unroll
        for (int i = 0; i + kOut < NP; ++i) pend[i] = pend[i + kOut];
        half_step(x[0], 2 * cs);
        if (2 * cs + 1 < KS) half_step(x[1], 2 * cs + 1);        // (no memory operation inside: the counts stay the same)
        if (cs == KS2 - 1) {
            // (the previous tile's pieces left during this tile's first NP / 4 steps: KS2 >= 3)
#pragma unroll
            for (int mt = 0; mt < kMT; ++mt)
#pragma unroll
                for (int t = 0; t < NT / 2; ++t) {
                    f16x8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = relu_half((f16)acc[mt][2 * t][j], relu);
                        o[4 + j] = relu_half((f16)acc[mt][2 * t + 1][j], relu);
                    }
                    pend[mt * (NT / 2) + t] = __builtin_bit_cast(u32x4, o);
                    acc[mt][2 * t] = (f32x4){0, 0, 0, 0};
                    acc[mt][2 * t + 1] = (f32x4){0, 0, 0, 0};
                }
            // rows past n_rows of a real tile fall outside the buffer; a tile past the end is not stored at all
            pend_i = ct < n_tiles ? 0 : NP;
            pend_base = (unsigned)(ct * (16 * kMT) + l15) * h_pitch + (unsigned)(col_base * 2 + 16 * lq);
            cs = 0;
            ct += tile_step;
        } else {
            ++cs;
        }
        issue(x);                    // this ring slot takes the pair kRing ahead
    };
    // The first turn is its own copy of the code: a wait count is a property of the instruction, and behind the first
    // turn's requests lie only the loads that filled the ring, not the steady state's 4 loads + 4 stores per step --
    // in a shared copy hipcc sets every count to what the first turn allows and all later turns wait for requests
    // newer than the one they need.
    if (turns > 0) {
#pragma unroll
        for (int d = 0; d < kRing; ++d) step(xr[d]);
        for (int64_t turn = 1; turn < turns; ++turn) {          // (inside the branch: entered from the first turn only)
#pragma unroll
            for (int d = 0; d < kRing; ++d) step(xr[d]);
        }
    }
}

template <int NT>
int launch_wlds(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                int relu, int n_cb, int LP, int KS, hipStream_t s)
{
    auto kernel = xw_dense_wlds_f16_kernel<NT>;
    const size_t lds_bytes = (size_t)16 * NT * LP * 2;
    static bool attr_set = false;                     // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWldsLds) !=
            hipSuccess)
            return SGX_ERR_HIP;
        attr_set = true;
    }
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
        cus = 256;
    const int wgs_per_cu = lds_bytes * 2 <= kWldsLds ? 2 : 1;
    int grid = cus / 8 * 8 * wgs_per_cu;
    grid = grid / (8 * n_cb) * (8 * n_cb);
    const int64_t tiles = ((int64_t)n_rows + 16 * kMT - 1) / (16 * kMT);
    const int64_t want = ((tiles + kWldsWaves - 1) / kWldsWaves + 7) / 8 * 8 * n_cb;       // no more streams than sets of 8 tiles
    if (want < grid) grid = (int)want;
    if (grid < 8 * n_cb) grid = 8 * n_cb;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kWldsThreads), lds_bytes, s, n_rows, M, P, (const f16 *)X, ldx, (const f16 *)Wt, ldw,
                       (f16 *)H, ldh, relu, n_cb, LP, KS);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

}  // namespace

// SGX_ERR_UNSUPPORTED: the shape is not this kernel's (the caller goes on to the tile kernels)
int sgx_xw_dense_wlds(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                      int relu, hipStream_t stream)
{
    if (sgx_tune().xw_no_wlds) return SGX_ERR_UNSUPPORTED;             // tuning override (tools/xw_dense_long_k_probe.py)
    if (M <= 128 || n_rows < 32768) return SGX_ERR_UNSUPPORTED;
    // X through 32-bit buffer offsets, 4-byte aligned rows (a 16-byte buffer load wants dword alignment)
    if ((uint64_t)n_rows * (uint64_t)ldx * 2ull >= 0xFFF00000ull || (ldx & 1) || ((uintptr_t)X & 3)) return SGX_ERR_UNSUPPORTED;
    // H in 16-byte pieces through 32-bit buffer offsets
    if ((uint64_t)n_rows * (uint64_t)ldh * 2ull >= 0xFFF00000ull || (ldh & 7) || ((uintptr_t)H & 15)) return SGX_ERR_UNSUPPORTED;
    if ((uint64_t)P * (uint64_t)ldw * 2ull >= 0xFFF00000ull || (ldw & 1) || ((uintptr_t)Wt & 3)) return SGX_ERR_UNSUPPORTED;
    const int KS = (M + 31) / 32, LP = 32 * KS + 8;
    const int cols = (int)ldh;                                          // the pad columns P .. ldh - 1 are produced (as zeros) too
    int nt = 0, n_cb = 0;
    for (int cand = 2; cand <= 8; cand *= 2) {                          // the narrowest column block that covers all columns ...
        if ((size_t)16 * cand * LP * 2 > kWldsLds) break;
        nt = cand;
        n_cb = (cols + 16 * cand - 1) / (16 * cand);
        if (n_cb == 1) break;
    }                                                                   // ... or the widest that fits, X then read once per block
    if (nt == 0 || n_cb > 2) return SGX_ERR_UNSUPPORTED;
    switch (nt) {
    case 2: return launch_wlds<2>(n_rows, M, P, X, ldx, Wt, ldw, H, ldh, relu, n_cb, LP, KS, stream);
    case 4: return launch_wlds<4>(n_rows, M, P, X, ldx, Wt, ldw, H, ldh, relu, n_cb, LP, KS, stream);
    default: return launch_wlds<8>(n_rows, M, P, X, ldx, Wt, ldw, H, ldh, relu, n_cb, LP, KS, stream);
    }
}

// ---------------------------------------------------------------------------------------
// fp32, K <= 128, wide outputs: all of W^T in LDS (256 x 128 floats = 128 KB, rows of K_pad + 4 floats so that the
// 16-byte fragment reads of 16 consecutive rows fall into 16 different bank groups), every wavefront computes ALL
// column tiles of its 16-row tiles -- X is read once, not once per column group as in the register-stationary kernel
// (xw_dense.hip), and a tile is 128 LDS reads + 512 MFMAs (16 K cycles of v_mfma_f32_16x16x4_f32) behind which the
// next tile's 8 KB of X arrive.  fp32 X.W is MFMA-bound (157 TF/s): 128 -> 256 on 169 K rows has a floor of 0.07 ms.
// Sums in the order of xw_dense_f32_kernel (k blocks ascending, element j of a lane's four ascending): the same bits.
// ---------------------------------------------------------------------------------------
namespace {

constexpr int kW32Threads = 512;

template <int KB, int NT>
__global__ __launch_bounds__(kW32Threads) void xw_dense_wlds_f32_kernel(
    int n_rows, int M, int P, const float *__restrict__ X, int64_t ldx, const float *__restrict__ Wt, int64_t ldw,
    float *__restrict__ H, int64_t ldh, int n_out, int h_aligned, sgx_epilogue ep, int relu)
{
    // (n_out: the columns this launch produces -- its block of a wider output; ldh stays the pitch of H)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw32[];
    float *sW = reinterpret_cast<float *>(lds_raw32);            // [16 NT][LP]
    constexpr int LP = 16 * KB + 4;                              // floats per LDS row: (LP / 4) odd
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, lq = lane >> 4;

    // W^T into LDS, 8 chunks of 16 bytes per thread requested at a time; zero outside W
    {
        constexpr int kFill = 8;
        constexpr int cpr = LP / 4, total = 16 * NT * cpr;
        const __amdgpu_buffer_rsrc_t w_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wt), 0, P > 0 ? (unsigned)(((int64_t)(P - 1) * ldw + M) * 4) : 0u, 0x00020000);
        for (int c0 = tid; c0 < total; c0 += kW32Threads * kFill) {
            u32x4 v[kFill];
            int kk[kFill];
#pragma unroll
            for (int u = 0; u < kFill; ++u) {
                const int c = c0 + kW32Threads * u;
                const int n = c / cpr, k = 4 * (c - n * cpr);
                kk[u] = k;
                const bool in = c < total && n < P && k < M;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, in ? (unsigned)(((int64_t)n * ldw + k) * 4) : kOOB, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < kFill; ++u) {
                const int c = c0 + kW32Threads * u;
#pragma unroll
                for (int d = 0; d < 4; ++d) v[u][d] = kk[u] + d < M ? v[u][d] : 0u;
                if (c < total) *reinterpret_cast<u32x4 *>(lds_raw32 + (size_t)c * 16) = v[u];
            }
        }
    }
    __syncthreads();

    const int64_t n_tiles = ((int64_t)n_rows + 15) / 16;
    const int64_t gw = (int64_t)blockIdx.x * (kW32Threads / 64) + (tid >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kW32Threads / 64);
    auto load_x = [&](int64_t tile, f32x4 (&b)[KB]) {
        const int64_t m = tile * 16 + l15;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int k = kb * 16 + 4 * lq;
            f32x4 v = {0, 0, 0, 0};
            if (m < n_rows) {
                const float *row = X + m * ldx;
                if (k + 4 <= M) {
                    v = *reinterpret_cast<const f32x4_u4 *>(row + k);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < M) v[j] = row[k + j];
                }
            }
            b[kb] = v;
        }
    };
    f32x4 b[KB], b_next[KB];
    int64_t tile = gw;
    if (tile < n_tiles) load_x(tile, b_next);
    const float *w_lane = sW + (size_t)l15 * LP + 4 * lq;
    for (; tile < n_tiles; tile += n_waves) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) b[kb] = b_next[kb];
        if (tile + n_waves < n_tiles) load_x(tile + n_waves, b_next);
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0, 0, 0, 0};
        // (column tiles 8 at a time: the fragments of all 16 at once do not fit beside the sums and two tiles of X)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int n0 = 0; n0 < NT; n0 += 8) {
                f32x4 a[8];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) a[nt] = *reinterpret_cast<const f32x4 *>(w_lane + (size_t)(16 * (n0 + nt)) * LP + 16 * kb);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt)
                        acc[n0 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt][j], b[kb][j], acc[n0 + nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);       // (or hipcc hoists the LDS reads of all 16 blocks: 512 VGPRs of fragments)
            }
        }
        const int64_t m = tile * 16 + l15;
        if (m >= n_rows) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + 4 * lq;
            float *dst = H + m * ldh + n;
            if (ep.rq_ten_pow != 0.0f) {                       // quantised layer: H is re-quantised as it is produced
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][j] = sgx_requant_value(acc[nt][j], ep);
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][j] = acc[nt][j] > 0.0f ? acc[nt][j] : 0.0f;
            }
            if (h_aligned && n + 4 <= n_out) {
                *reinterpret_cast<f32x4 *>(dst) = acc[nt];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < n_out) dst[j] = acc[nt][j];
            }
        }
    }
}

template <int KB, int NT>
int launch_wlds_f32(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh, int n_out,
                    int ha, sgx_epilogue ep, int relu, hipStream_t s)
{
    auto kernel = xw_dense_wlds_f32_kernel<KB, NT>;
    const size_t lds_bytes = (size_t)16 * NT * (16 * KB + 4) * 4;
    static bool attr_set = false;                     // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWldsLds) !=
            hipSuccess)
            return SGX_ERR_HIP;
        attr_set = true;
    }
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    const int64_t tiles = ((int64_t)n_rows + 15) / 16;
    int64_t grid = cus;                               // one workgroup per CU (the table takes most of its LDS)
    const int64_t want = (tiles + kW32Threads / 64 - 1) / (kW32Threads / 64);
    if (want < grid) grid = want;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kW32Threads), lds_bytes, s, n_rows, M, P, (const float *)X, ldx,
                       (const float *)Wt, ldw, (float *)H, ldh, n_out, ha, ep, relu);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

}  // namespace

// SGX_ERR_UNSUPPORTED: not this kernel's shape (the caller goes on to the register-stationary and tile kernels)
int sgx_xw_dense_wlds_f32(int n_rows, int M, int P, const void *X, int64_t ldx, const void *Wt, int64_t ldw, void *H, int64_t ldh,
                          int h_aligned, sgx_epilogue ep, int relu, hipStream_t stream)
{
    if (sgx_tune().xw_no_wlds) return SGX_ERR_UNSUPPORTED;             // tuning / test override
    const int cols = (int)ldh;                                          // pad columns are produced (as zeros) too
    // wide outputs only: with few column tiles the register-stationary kernel reads X once as well; more than 256 columns
    // go in blocks of 256 (X re-read per block: 602 columns of the backward's g . W^T = 3 passes instead of the 10 of the
    // register-stationary kernel's column groups)
    if (M > 128 || M <= 32 || cols <= 64 || cols > 1024 || n_rows < 32768) return SGX_ERR_UNSUPPORTED;
    if ((uint64_t)P * (uint64_t)ldw * 4ull >= 0xFFF00000ull || (ldw & 3) || ((uintptr_t)Wt & 15)) return SGX_ERR_UNSUPPORTED;
    const int kb = (M + 15) / 16 <= 4 ? 4 : 8;
    for (int c0 = 0; c0 < cols; c0 += 256) {
        const int n_out = cols - c0 < 256 ? cols - c0 : 256;
        const int p_blk = P - c0 < 0 ? 0 : (P - c0 < 256 ? P - c0 : 256);
        const float *w_blk = (const float *)Wt + (int64_t)c0 * ldw;
        float *h_blk = (float *)H + c0;
        const int ha = h_aligned && ((uintptr_t)h_blk % 16 == 0);
        int rc;
        if (p_blk == 0) w_blk = (const float *)Wt;                      // (only pad columns left: every row of the block reads as zero)
#define SGX_W32(KB_, NT_) rc = launch_wlds_f32<KB_, NT_>(n_rows, M, p_blk, X, ldx, w_blk, ldw, h_blk, ldh, n_out, ha, ep, relu, stream)
        if (kb == 4) { if (n_out <= 128) SGX_W32(4, 8); else SGX_W32(4, 16); }
        else { if (n_out <= 128) SGX_W32(8, 8); else SGX_W32(8, 16); }
#undef SGX_W32
        if (rc != SGX_OK) return rc;
    }
    return SGX_OK;
}
