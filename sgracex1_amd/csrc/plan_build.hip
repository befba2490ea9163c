// Row schedule of a CSR matrix (sgx_plan), built on the device.
//
// The reference decides on the host, once per matrix, how rows are split over ADJ_THREADS / FEA_THREADS and grouped
// per pipelined loop (K.cpp:3517-3523, :826-845; MM.h:166-191).  Here the same two decisions -- which rows are cut
// into edge tasks, and whether the others are walked in degree order -- are made from rowPtr where it lies, in HBM:
// the host reads back 40 bytes ONCE (the entry count, the cut the device picked from it, how many long rows and tasks
// there are, to size the arrays, and the lane-group utilisation of the natural order), never rowPtr itself.  A
// sampled mini-batch (the demo's NeighborLoader call pattern) pays one stream synchronisation of a few microseconds
// per adjacency (a second one at the end only when kernels follow the read-back: long rows or the degree order)
// instead of a copy of rowPtr and three passes over it on one host core.
//
//   count      per row: long? how many tasks? how many 8-edge steps?  per 8 consecutive rows: the longest's steps
//              (what a wavefront that packs them spends).  Totals by atomics, long rows / tasks per row block.
//   long rows  exclusive scan over the row blocks, then every block numbers its long rows in row order:
//              long_row[], long_first[]; one thread per task fills task_row / task_e0 / task_e1.
//   order      (only when the natural order would leave lane groups idle) a stable counting sort of the short rows
//              by step count, longest first, ascending row id inside a bucket: histogram per row block, scan per
//              bucket over the blocks, scan over the buckets, scatter with the rows of a block taken in order.
// The arrays are the ones the host-side builder of round 1 produced, entry for entry (tests/test_gpu_plan.py
// restates the rules in numpy and compares).
#include "sgx_device.h"

#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;          // row blocks: scanned by one workgroup in one pass

constexpr int kLongThreshold = 4096;
constexpr int kMaxCut = 1 << 16;            // the largest cut a caller may ask for
// Rows with more edges than the threshold take the split path, in tasks of `chunk` edges.  Measured on the
// R-MAT S-100M aggregation (threshold = chunk): 512: 2.20 ms, 1024-2048: 2.10, 3072: 2.00, 4096: 1.94,
// 6144: 2.04, 8192: 2.32 -- a lane group walks a 4096-edge row in 512 steps while the degree-ordered schedule
// keeps its wavefront full, and every task costs a partial row and a finalize read.  (The GAT aggregate, with
// its softmax state per step, prefers 256: sgx_plan_create_ex.)
// The best cut moves with the size of the graph -- a launch of a smaller graph is over before a 4096-edge row's 512
// dependent steps are (tools/plan_cut_probe.py, R-MAT, cut / ms of the plain aggregate: 2.4 M edges 512 / 0.136 against
// 4096 / 0.400; 7.5 M edges 1024 / 0.179 against 0.241; 29 M edges 2048 / 0.448 against 0.786; 104 M edges 4096): the
// optimum follows sqrt(nnz) / 2 rounded down to a power of two, which is what default_cut returns.
__host__ __device__ inline int default_cut(int64_t nnz)
{
    int cut = 64;
    while (cut < kLongThreshold && (int64_t)(2 * cut) * (2 * cut) * 4 <= nnz) cut *= 2;      // 2 cut <= sqrt(nnz) / 2
    return cut;
}
// Small matrices finish in microseconds and their time IS the longest row's chain of dependent
// steps (Cora: 168 edges = 21 steps on one lane group), so there rows are cut much earlier: a
// 64-edge task is one step for every lane group of its wavefront.
constexpr int64_t kSmallNnz = 1 << 20;
constexpr int kSmallThreshold = 64, kSmallChunk = 64;
const float kReorderBelow = 0.7f; // natural-order lane-group utilisation below which rows are degree-ordered

struct PlanCounts {
    unsigned long long useful;     // lane-group steps that do work, rows packed 8 to a wavefront in natural order
    unsigned long long spent;      // lane-group steps such wavefronts run for
    unsigned n_long, n_tasks;
    int nnz, long_threshold, chunk, max_degree;      // the entry count the device read, the cut it picked from it (below), the longest row
};

// The cut depends on the entry count, which lives in HBM (rowPtr[n_rows]): the counting kernel reads it there and picks
// the cut itself, so the host needs ONE read-back per plan (counts + entry count + cut), not one for the entry count
// and a second for the counts.  caller_* = sgx_plan_create_ex's arguments, env_* = the tuning overrides (0 = unset).
__host__ __device__ inline void resolve_cut(int64_t nnz, int caller_threshold, int caller_chunk, int env_threshold, int env_chunk,
                                            int &long_threshold, int &chunk)
{
    const bool small = nnz < kSmallNnz;
    long_threshold = small ? kSmallThreshold : default_cut(nnz);
    chunk = small ? kSmallChunk : long_threshold;
    if (!small && caller_threshold >= 8) {                            // the caller's cut (large matrices only)
        long_threshold = caller_threshold / 8 * 8;
        chunk = caller_chunk >= 8 ? caller_chunk / 8 * 8 : long_threshold;
    }
    if (env_threshold >= 8) long_threshold = chunk = env_threshold / 8 * 8;
    if (env_chunk >= 8) chunk = env_chunk / 8 * 8;
    if (long_threshold > kMaxCut) long_threshold = kMaxCut;           // (one LDS counter per step count in the degree order)
    if (chunk > kMaxCut) chunk = kMaxCut;
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// inclusive scan of one value per thread over the workgroup, threads in order; returns the workgroup's total too
__device__ __forceinline__ int block_scan_inclusive(int v, int *wave_tot, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    __syncthreads();                               // wave_tot may still be read from the previous call
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    int before = 0;
    total = 0;
    for (int i = 0; i < n_waves; ++i) {
        const int t = wave_tot[i];
        if (i < wave) before += t;
        total += t;
    }
    return v + before;
}

__global__ __launch_bounds__(kThreads) void plan_count_kernel(const int32_t *__restrict__ rowptr, int n_rows, int rows_per_block,
                                                              int caller_threshold, int caller_chunk, int env_threshold, int env_chunk,
                                                              PlanCounts *__restrict__ totals,
                                                              int32_t *__restrict__ block_long, int32_t *__restrict__ block_tasks)
{
    __shared__ unsigned long long s_useful, s_spent;
    __shared__ int s_long, s_tasks;
    int long_threshold, chunk;
    const int nnz = rowptr[n_rows];
    resolve_cut(nnz, caller_threshold, caller_chunk, env_threshold, env_chunk, long_threshold, chunk);
    if (threadIdx.x == 0) {
        s_useful = s_spent = 0ull;
        s_long = s_tasks = 0;
        if (blockIdx.x == 0) {
            totals->nnz = nnz;
            totals->long_threshold = long_threshold;
            totals->chunk = chunk;
        }
    }
    __syncthreads();
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = r_begin + rows_per_block < n_rows ? r_begin + rows_per_block : (int64_t)n_rows;
    int useful = 0, spent = 0, n_long = 0, n_tasks = 0, max_deg = 0;
    for (int64_t base = r_begin; base < r_end; base += kThreads) {
        const int64_t r = base + threadIdx.x;
        const bool valid = r < r_end;
        const int deg = valid ? rowptr[r + 1] - rowptr[r] : 0;
        const bool is_long = deg > long_threshold;
        max_deg = max(max_deg, deg);
        const int steps = is_long ? 0 : (deg + 7) / 8;
        int mx = steps;                            // the longest of the 8 rows a wavefront would pack with this one
        mx = max(mx, __shfl_xor(mx, 1));
        mx = max(mx, __shfl_xor(mx, 2));
        mx = max(mx, __shfl_xor(mx, 4));
        useful += steps;
        if (valid && (threadIdx.x & 7) == 0) spent += mx * 8;
        if (is_long) {
            ++n_long;
            n_tasks += (deg + chunk - 1) / chunk;
        }
    }
    // per-thread sums stay below 2^31: a thread sees rows_per_block / 256 rows of at most long_threshold / 8 + 1 steps
    // each; the wave sums go to 64 bits
    unsigned long long u = (unsigned long long)useful, s = (unsigned long long)spent;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u += __shfl_xor(u, o);
        s += __shfl_xor(s, o);
    }
    n_long = wave_sum(n_long);
    n_tasks = wave_sum(n_tasks);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_deg = max(max_deg, __shfl_xor(max_deg, o));
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&totals->max_degree, max_deg);
        atomicAdd(&s_useful, u);
        atomicAdd(&s_spent, s);
        atomicAdd(&s_long, n_long);
        atomicAdd(&s_tasks, n_tasks);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_long[blockIdx.x] = s_long;
        block_tasks[blockIdx.x] = s_tasks;
        atomicAdd(&totals->useful, s_useful);
        atomicAdd(&totals->spent, s_spent);
        atomicAdd(&totals->n_long, (unsigned)s_long);
        atomicAdd(&totals->n_tasks, (unsigned)s_tasks);
    }
}

// exclusive scans, in place, of the `n` entries of each of gridDim.x arrays (array a at data + a * pitch): one workgroup
// per array, 1024 entries per pass with the running sum carried; totals[a] (optional) takes the array's sum
__global__ __launch_bounds__(kMaxBlocks) void plan_scan_kernel(int32_t *__restrict__ data, int n, int64_t pitch, int32_t *__restrict__ totals)
{
    __shared__ int wave_tot[kMaxBlocks / 64];
    int32_t *a = data + (int64_t)blockIdx.x * pitch;
    int carry = 0;
    for (int base = 0; base < n; base += kMaxBlocks) {
        const int i = base + (int)threadIdx.x;
        const int v = i < n ? a[i] : 0;
        int total;
        const int incl = block_scan_inclusive(v, wave_tot, total);
        if (i < n) a[i] = carry + incl - v;
        carry += total;
    }
    if (totals && threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kThreads) void plan_fill_long_kernel(const int32_t *__restrict__ rowptr, int n_rows, int rows_per_block,
                                                                  int long_threshold, int chunk,
                                                                  const int32_t *__restrict__ block_long, const int32_t *__restrict__ block_tasks,
                                                                  int32_t *__restrict__ long_row, int32_t *__restrict__ long_first, int n_long,
                                                                  int n_tasks)
{
    __shared__ int wave_tot[kThreads / 64];
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = r_begin + rows_per_block < n_rows ? r_begin + rows_per_block : (int64_t)n_rows;
    int long_at = block_long[blockIdx.x], task_at = block_tasks[blockIdx.x];      // exclusive prefixes over the row blocks
    if (blockIdx.x == 0 && threadIdx.x == 0) long_first[n_long] = n_tasks;
    for (int64_t base = r_begin; base < r_end; base += kThreads) {
        const int64_t r = base + threadIdx.x;
        const int deg = r < r_end ? rowptr[r + 1] - rowptr[r] : 0;
        const int is_long = deg > long_threshold ? 1 : 0;
        const int tasks = is_long ? (deg + chunk - 1) / chunk : 0;
        int tot_long, tot_tasks;
        const int pos_long = block_scan_inclusive(is_long, wave_tot, tot_long) - is_long;
        if (tot_long == 0) continue;                         // (uniform over the workgroup)
        const int pos_task = block_scan_inclusive(tasks, wave_tot, tot_tasks) - tasks;
        if (is_long) {
            long_row[long_at + pos_long] = (int32_t)r;
            long_first[long_at + pos_long] = task_at + pos_task;
        }
        long_at += tot_long;
        task_at += tot_tasks;
    }
}

__global__ void plan_fill_tasks_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ long_row,
                                       const int32_t *__restrict__ long_first, int n_long, int n_tasks, int chunk,
                                       int32_t *__restrict__ task_row, int32_t *__restrict__ task_e0, int32_t *__restrict__ task_e1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tasks) return;
    int lo = 0, hi = n_long - 1;                             // the last long row whose first task is <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (long_first[mid] <= i) lo = mid;
        else hi = mid - 1;
    }
    const int r = long_row[lo];
    const int e_end = rowptr[r + 1];
    const int e0 = rowptr[r] + (i - long_first[lo]) * chunk;
    task_row[i] = r;
    task_e0[i] = e0;
    task_e1[i] = (e_end - e0 > chunk) ? e0 + chunk : e_end;
}

// short rows of every row block counted per bucket; counts[k * n_blocks + block], k = steps_max - steps (longest first)
__global__ __launch_bounds__(kThreads) void plan_hist_kernel(const int32_t *__restrict__ rowptr, int n_rows, int rows_per_block,
                                                             int long_threshold, int steps_max, int32_t *__restrict__ counts)
{
    extern __shared__ int hist[];
    for (int i = threadIdx.x; i <= steps_max; i += kThreads) hist[i] = 0;
    __syncthreads();
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = r_begin + rows_per_block < n_rows ? r_begin + rows_per_block : (int64_t)n_rows;
    for (int64_t r = r_begin + threadIdx.x; r < r_end; r += kThreads) {
        const int deg = rowptr[r + 1] - rowptr[r];
        if (deg <= long_threshold) atomicAdd(&hist[steps_max - (deg + 7) / 8], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= steps_max; i += kThreads) counts[(int64_t)i * gridDim.x + blockIdx.x] = hist[i];
}

// The rows of a block are taken in order: 256 at a time, the wavefronts of those one after the other, and inside a
// wavefront the lanes of the same bucket are numbered by lane id -- so every bucket receives its rows ascending.
__global__ __launch_bounds__(kThreads) void plan_scatter_kernel(const int32_t *__restrict__ rowptr, int n_rows, int rows_per_block,
                                                                int long_threshold, int steps_max,
                                                                const int32_t *__restrict__ counts_excl, const int32_t *__restrict__ bucket_base,
                                                                int32_t *__restrict__ order)
{
    extern __shared__ int cursor[];
    for (int i = threadIdx.x; i <= steps_max; i += kThreads)
        cursor[i] = bucket_base[i] + counts_excl[(int64_t)i * gridDim.x + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = r_begin + rows_per_block < n_rows ? r_begin + rows_per_block : (int64_t)n_rows;
    for (int64_t base = r_begin; base < r_end; base += kThreads) {
        const int64_t r = base + threadIdx.x;
        const int deg = r < r_end ? rowptr[r + 1] - rowptr[r] : 0;
        const bool mine = r < r_end && deg <= long_threshold;
        const int k = steps_max - (deg + 7) / 8;
        for (int turn = 0; turn < kThreads / 64; ++turn) {
            if (turn == wave) {
                unsigned long long todo = __ballot(mine);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int kb = __shfl(k, leader);
                    const unsigned long long same = __ballot(mine && k == kb);
                    const int at = cursor[kb];                               // read by every lane before the leader moves it
                    if (mine && k == kb) order[at + __popcll(same & below)] = (int32_t)r;
                    if (lane == leader) cursor[kb] = at + __popcll(same);
                    todo &= ~same;
                }
            }
            __syncthreads();
        }
    }
}

// win_order (sgx_internal.h): one wavefront per window of 64 rows, a bitonic network over the lanes on keys
// (0xFFFFF - min(length, 0xFFFFF)) : lane -- ascending keys = longest first, ties in row order.
__global__ __launch_bounds__(kThreads) void plan_window_order_kernel(const int32_t *__restrict__ rowptr, int n_rows, int64_t n_windows,
                                                                     uint8_t *__restrict__ win_order)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (w >= n_windows) return;
    const int64_t row = w * 64 + lane;
    int len = row < n_rows ? rowptr[row + 1] - rowptr[row] : 0;
    len = len > 0xFFFFF ? 0xFFFFF : len;
    unsigned key = ((unsigned)(0xFFFFF - len) << 6) | (unsigned)lane;
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
    win_order[w * 64 + lane] = (uint8_t)(key & 63u);
}

// scan_win (sgx_internal.h): a thread per window of kScanGranule entries, lower bound of the window's first entry among
// the row starts
__global__ __launch_bounds__(kThreads) void plan_scan_windows_kernel(const int32_t *__restrict__ rowptr, int n_rows, int64_t nnz,
                                                                     int64_t n_win, int long_threshold, int32_t *__restrict__ scan_win)
{
    const int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (g > n_win) return;
    const int64_t e64 = g * kScanGranule;
    const int e = (int)(e64 < nnz ? e64 : nnz);
    int lo = 0, hi = n_rows;                       // the first r in [0, n_rows] with rowptr[r] >= e (rowptr[n_rows] = nnz >= e)
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (rowptr[mid] < e) lo = mid + 1; else hi = mid;
    }
    const int first = rowptr[lo];
    const bool long_before = lo > 0 && first - rowptr[lo - 1] > long_threshold;
    scan_win[4 * g] = lo;
    scan_win[4 * g + 1] = first;
    scan_win[4 * g + 2] = long_before ? lo - 1 : lo;
    scan_win[4 * g + 3] = long_before ? rowptr[lo - 1] : first;
}

// The builder's own scratch comes from the stream-ordered pool and goes back to it on the same stream (on every way
// out): no synchronisation is needed to free it after the kernels that read it, and a builder called per mini-batch
// does not pay hipMalloc / hipFree each time.
struct Scratch {
    void *p = nullptr;
    hipStream_t s = nullptr;
    ~Scratch() { if (p) (void)hipFreeAsync(p, s); }
};

}  // namespace

extern "C" int sgx_plan_create_ex(sgx_plan **out, const int32_t *rowPtr, int n_rows, int long_threshold_arg, int chunk_arg,
                                  void *stream)
{
    if (!out || !rowPtr) return SGX_ERR_NULL;
    if (long_threshold_arg < 0 || chunk_arg < 0) return SGX_ERR_SHAPE;
    if (n_rows < 0) return SGX_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const sgx_tuning &tune = sgx_tune();

    // row blocks: whole multiples of the workgroup, at most kMaxBlocks of them
    int64_t rows_per_block64 = 4096;
    while ((n_rows + rows_per_block64 - 1) / rows_per_block64 > kMaxBlocks) rows_per_block64 *= 2;
    const int rows_per_block = (int)rows_per_block64;
    const int n_blocks = n_rows > 0 ? (int)((n_rows + rows_per_block64 - 1) / rows_per_block64) : 0;

    sgx_plan *p = new sgx_plan();
    p->n_rows = n_rows;
    p->nnz = 0;
    p->n_long = p->n_tasks = 0;
    p->long_row = p->long_first = p->task_row = p->task_e0 = p->task_e1 = nullptr;
    p->row_order = nullptr;
    p->win_order = nullptr;
    p->scan_win = nullptr;
    p->n_scan_win = 0;
    p->n_ordered = 0;
    p->n_multi = -1;
    p->max_degree = 0;
    p->natural_utilization = 1.0f;
    resolve_cut(0, long_threshold_arg, chunk_arg, tune.plan_long_threshold, tune.plan_chunk, p->long_threshold, p->chunk);
    if (n_blocks == 0) {
        *out = p;
        return SGX_OK;
    }
    struct Guard {                // the plan is destroyed on every error path
        sgx_plan *p;
        ~Guard() { if (p) sgx_plan_destroy(p); }
    } guard{p};

    Scratch counts_mem;
    counts_mem.s = s;
    const size_t counts_bytes = sizeof(PlanCounts) + sizeof(int32_t) * 2 * (size_t)n_blocks;
    SGX_HIP_CHECK(hipMallocAsync(&counts_mem.p, counts_bytes, s));
    PlanCounts *totals = (PlanCounts *)counts_mem.p;
    int32_t *block_long = (int32_t *)(totals + 1), *block_tasks = block_long + n_blocks;
    SGX_HIP_CHECK(hipMemsetAsync(totals, 0, sizeof(PlanCounts), s));
    hipLaunchKernelGGL(plan_count_kernel, dim3(n_blocks), dim3(kThreads), 0, s, rowPtr, n_rows, rows_per_block, long_threshold_arg,
                       chunk_arg, tune.plan_long_threshold, tune.plan_chunk, totals, block_long, block_tasks);
    SGX_LAUNCH_CHECK();
    PlanCounts host{};
    SGX_HIP_CHECK(hipMemcpyAsync(&host, totals, sizeof(PlanCounts), hipMemcpyDeviceToHost, s));
    SGX_HIP_CHECK(hipStreamSynchronize(s));                          // the ONE read-back of a plan build (40 bytes)
    if (host.n_tasks > 0x7FFFFFFFu) return SGX_ERR_SHAPE;
    p->nnz = host.nnz;
    const int long_threshold = p->long_threshold = host.long_threshold;
    const int chunk = p->chunk = host.chunk;
    const int steps_max = long_threshold / 8 + 1, n_buckets = steps_max + 1;
    p->n_long = (int)host.n_long;
    p->n_tasks = (int)host.n_tasks;
    p->natural_utilization = host.spent > 0 ? (float)((double)host.useful / (double)host.spent) : 1.0f;
    p->max_degree = host.max_degree;
    bool launched_after_readback = false;

    if (p->n_long > 0) {
        const size_t nl = (size_t)p->n_long, nt = (size_t)p->n_tasks;
        int32_t *blob = nullptr;
        SGX_HIP_CHECK(hipMalloc(&blob, sizeof(int32_t) * (nl + nl + 1 + 3 * nt)));
        p->long_row = blob;
        p->long_first = blob + nl;
        p->task_row = p->long_first + nl + 1;
        p->task_e0 = p->task_row + nt;
        p->task_e1 = p->task_e0 + nt;
        launched_after_readback = true;
        hipLaunchKernelGGL(plan_scan_kernel, dim3(2), dim3(kMaxBlocks), 0, s, block_long, n_blocks, (int64_t)n_blocks, (int32_t *)nullptr);
        SGX_LAUNCH_CHECK();
        hipLaunchKernelGGL(plan_fill_long_kernel, dim3(n_blocks), dim3(kThreads), 0, s, rowPtr, n_rows, rows_per_block, long_threshold,
                           chunk, block_long, block_tasks, p->long_row, p->long_first, p->n_long, p->n_tasks);
        SGX_LAUNCH_CHECK();
        hipLaunchKernelGGL(plan_fill_tasks_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, rowPtr, p->long_row,
                           p->long_first, p->n_long, p->n_tasks, chunk, p->task_row, p->task_e0, p->task_e1);
        SGX_LAUNCH_CHECK();
    }

    // Would packing 8 consecutive rows per wavefront keep the lane groups busy?  A group needs ceil(deg / 8) steps, the
    // wavefront runs for the largest of its 8 rows: natural_utilization.  Below kReorderBelow the short rows are
    // scheduled in degree order instead.
    const float reorder_below = tune.plan_reorder_below >= 0.0f ? tune.plan_reorder_below : kReorderBelow;      // (tuning override)
    Scratch order_mem;
    order_mem.s = s;
    if (p->natural_utilization < reorder_below && n_rows - p->n_long > 0) {
        const size_t n_counts = (size_t)n_buckets * (size_t)n_blocks;
        SGX_HIP_CHECK(hipMallocAsync(&order_mem.p, sizeof(int32_t) * (n_counts + (size_t)n_buckets), s));
        int32_t *counts = (int32_t *)order_mem.p, *bucket_base = counts + n_counts;
        SGX_HIP_CHECK(hipMalloc(&p->row_order, sizeof(int32_t) * (size_t)(n_rows - p->n_long)));
        p->n_ordered = n_rows - p->n_long;
        launched_after_readback = true;
        const size_t lds = sizeof(int) * (size_t)n_buckets;
        hipLaunchKernelGGL(plan_hist_kernel, dim3(n_blocks), dim3(kThreads), lds, s, rowPtr, n_rows, rows_per_block, long_threshold,
                           steps_max, counts);
        SGX_LAUNCH_CHECK();
        // per bucket: exclusive scan over the row blocks, the bucket's total aside; then the totals themselves
        hipLaunchKernelGGL(plan_scan_kernel, dim3(n_buckets), dim3(kMaxBlocks), 0, s, counts, n_blocks, (int64_t)n_blocks, bucket_base);
        SGX_LAUNCH_CHECK();
        hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(kMaxBlocks), 0, s, bucket_base, n_buckets, (int64_t)0, (int32_t *)nullptr);
        SGX_LAUNCH_CHECK();
        hipLaunchKernelGGL(plan_scatter_kernel, dim3(n_blocks), dim3(kThreads), lds, s, rowPtr, n_rows, rows_per_block, long_threshold,
                           steps_max, counts, bucket_base, p->row_order);
        SGX_LAUNCH_CHECK();
        // where the one-step rows begin in the order (bucket k = steps_max - steps, scanned: the start of bucket steps_max - 1);
        // read with the synchronisation that ends the build anyway
        int32_t start_one = 0;
        SGX_HIP_CHECK(hipMemcpyAsync(&start_one, bucket_base + (steps_max - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SGX_HIP_CHECK(hipStreamSynchronize(s));
        p->n_multi = start_one;
    }
    // the window order of the sparse X.W stage's LDS form, for the matrices that form takes (xw_sparse_lds.hip)
    if (p->n_tasks == 0 && p->nnz >= ((int64_t)1 << 20) && n_rows >= 4096) {
        const int64_t n_windows = ((int64_t)n_rows + 63) / 64;
        SGX_HIP_CHECK(hipMalloc(&p->win_order, (size_t)n_windows * 64));
        launched_after_readback = true;
        hipLaunchKernelGGL(plan_window_order_kernel, dim3((unsigned)((n_windows + kThreads / 64 - 1) / (kThreads / 64))), dim3(kThreads), 0, s,
                           rowPtr, n_rows, n_windows, p->win_order);
        SGX_LAUNCH_CHECK();
    }
    // the row-aligned entry windows of the GAT aggregate's scan (gat_scan.hip): plans asked to cut at its row limit
    // (Csr.gat_plan) and cut there -- or without a longer row; the aggregation's own plans never pay for them
    if (long_threshold_arg == kScanMaxRow && p->nnz > 0 && p->nnz < ((int64_t)1 << 30) &&
        (p->n_long > 0 ? long_threshold == kScanMaxRow : p->max_degree <= kScanMaxRow)) {
        p->n_scan_win = (p->nnz + kScanGranule - 1) / kScanGranule;
        SGX_HIP_CHECK(hipMalloc(&p->scan_win, sizeof(int32_t) * 4 * (size_t)(p->n_scan_win + 1)));
        launched_after_readback = true;
        hipLaunchKernelGGL(plan_scan_windows_kernel, dim3((unsigned)((p->n_scan_win + 1 + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                           rowPtr, n_rows, p->nnz, p->n_scan_win, p->n_long > 0 ? long_threshold : 0x7FFFFFFF, p->scan_win);
        SGX_LAUNCH_CHECK();
    }
    // A plan is used from any stream (the partitioned layer launches on side streams): its arrays must be complete when
    // this returns, not merely ordered on `stream`.  Only a plan with long rows or a degree order has kernels behind the
    // read-back; the common plan (uniform graph, mini-batch) returned complete at the read-back's synchronisation.
    if (launched_after_readback) SGX_HIP_CHECK(hipStreamSynchronize(s));
    guard.p = nullptr;
    *out = p;
    return SGX_OK;
}

extern "C" int sgx_plan_create(sgx_plan **out, const int32_t *rowPtr, int n_rows, int n_feat_hint, void *stream)
{
    (void)n_feat_hint;
    return sgx_plan_create_ex(out, rowPtr, n_rows, 0, 0, stream);
}

extern "C" void sgx_plan_destroy(sgx_plan *plan)
{
    if (!plan) return;
    if (plan->long_row) (void)hipFree(plan->long_row);     // one blob, long_row is its base
    if (plan->row_order) (void)hipFree(plan->row_order);
    if (plan->win_order) (void)hipFree(plan->win_order);
    if (plan->scan_win) (void)hipFree(plan->scan_win);
    delete plan;
}

// One of the plan's arrays copied (device to device) for inspection: 0 long_row, 1 long_first, 2 task_row, 3 task_e0,
// 4 task_e1, 5 row_order, 6 win_order (its bytes, four to an int32), 7 scan_win (four per window boundary).  Returns the array's length (dst == NULL: the length only) or a negative sgx error.
extern "C" int64_t sgx_plan_export(const sgx_plan *plan, int which, int32_t *dst, int64_t capacity, void *stream)
{
    if (!plan) return SGX_ERR_NULL;
    const int32_t *src = nullptr;
    int64_t n = 0;
    switch (which) {
    case 0: src = plan->long_row; n = plan->n_long; break;
    case 1: src = plan->long_first; n = plan->n_long > 0 ? plan->n_long + 1 : 0; break;
    case 2: src = plan->task_row; n = plan->n_tasks; break;
    case 3: src = plan->task_e0; n = plan->n_tasks; break;
    case 4: src = plan->task_e1; n = plan->n_tasks; break;
    case 5: src = plan->row_order; n = plan->row_order ? plan->n_ordered : 0; break;
    case 6:                      // win_order: bytes, four to an int32 (64 per window of 64 rows)
        src = reinterpret_cast<const int32_t *>(plan->win_order);
        n = plan->win_order ? ((int64_t)plan->n_rows + 63) / 64 * 16 : 0;
        break;
    case 7: src = plan->scan_win; n = plan->scan_win ? 4 * (plan->n_scan_win + 1) : 0; break;
    default: return SGX_ERR_UNSUPPORTED;
    }
    if (!dst || n == 0) return n;
    if (capacity < n) return SGX_ERR_SHAPE;
    SGX_HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return n;
}
