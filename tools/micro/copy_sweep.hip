// Streaming-copy sweep on gfx950: which form of a plain 16-byte-per-lane copy reaches the highest read + write rate
// (the "attainable roof" bench.py quotes beside the nominal 8 TB/s).
// Build: hipcc -w --offload-arch=gfx950 -O3 tools/micro/copy_sweep.hip -o tools/micro/copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT_LD, bool NT_ST, int BLOCK, bool CONTIG>
__global__ __launch_bounds__(BLOCK) void copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, long n16)
{
    // CONTIG: a workgroup walks its own contiguous chunk (UNROLL x BLOCK x 16 bytes per step); else grid-stride
    long stride = CONTIG ? BLOCK : (long)gridDim.x * BLOCK;
    long per_block = (n16 + gridDim.x - 1) / gridDim.x;
    long i = CONTIG ? (long)blockIdx.x * per_block + threadIdx.x : (long)blockIdx.x * BLOCK + threadIdx.x;
    long end = CONTIG ? ((long)(blockIdx.x + 1) * per_block < n16 ? (long)(blockIdx.x + 1) * per_block : n16) : n16;
    for (; i + (UNROLL - 1) * stride < end; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT_ST) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < end; i += stride) dst[i] = src[i];
}

template <int UNROLL, bool NT_LD, bool NT_ST, int BLOCK, bool CONTIG>
void run(const char *name, const u32x4 *src, u32x4 *dst, long n16, int blocks_per_cu)
{
    int grid = 256 * blocks_per_cu;
    hipEvent_t b, e;
    hipEventCreate(&b); hipEventCreate(&e);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((copy_kernel<UNROLL, NT_LD, NT_ST, BLOCK, CONTIG>), dim3(grid), dim3(BLOCK), 0, 0, src, dst, n16);
    hipEventRecord(b);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((copy_kernel<UNROLL, NT_LD, NT_ST, BLOCK, CONTIG>), dim3(grid), dim3(BLOCK), 0, 0, src, dst, n16);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, b, e);
    printf("%-40s block %4d blocks/CU %3d  %7.1f GB/s (read + write)\n", name, BLOCK, blocks_per_cu, reps * 2.0 * n16 * 16 / (ms * 1e-3) / 1e9);
}

int main()
{
    const long bytes = 1L << 30;
    const long n16 = bytes / 16;
    u32x4 *src, *dst;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    hipEvent_t b, e; hipEventCreate(&b); hipEventCreate(&e);
    hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b);
    for (int r = 0; r < 5; ++r) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, b, e);
    printf("%-40s %7.1f GB/s (read + write)\n", "hipMemcpyAsync D2D", 5 * 2.0 * bytes / (ms * 1e-3) / 1e9);
    for (int bpc : {2, 4, 8, 16, 32}) {
        run<4, true, true, 256, false>("nt ld, nt st, unroll 4, grid-stride", src, dst, n16, bpc);
        run<4, false, false, 256, false>("plain, unroll 4, grid-stride", src, dst, n16, bpc);
        run<4, false, true, 256, false>("plain ld, nt st, unroll 4, grid-stride", src, dst, n16, bpc);
        run<4, true, false, 256, false>("nt ld, plain st, unroll 4, grid-stride", src, dst, n16, bpc);
        run<8, false, false, 256, false>("plain, unroll 8, grid-stride", src, dst, n16, bpc);
        run<2, false, false, 256, false>("plain, unroll 2, grid-stride", src, dst, n16, bpc);
        run<1, false, false, 256, false>("plain, unroll 1, grid-stride", src, dst, n16, bpc);
        run<4, false, false, 256, true>("plain, unroll 4, contiguous chunk", src, dst, n16, bpc);
        run<4, true, true, 256, true>("nt, unroll 4, contiguous chunk", src, dst, n16, bpc);
    }
    for (int bpc : {1, 2, 4}) {
        run<4, false, false, 1024, false>("plain, unroll 4, grid-stride", src, dst, n16, bpc);
        run<4, true, true, 1024, false>("nt, unroll 4, grid-stride", src, dst, n16, bpc);
        run<8, false, false, 512, false>("plain, unroll 8, grid-stride", src, dst, n16, bpc);
    }
    // one launch per element: no loop at all
    return 0;
}
