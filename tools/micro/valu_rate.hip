// Issue rate of vector instructions on gfx950: cycles per wave-instruction per SIMD for independent chains,
// at 1, 2 and 4 wavefronts per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int KIND>
__global__ __launch_bounds__(1024) void rate(float *out, int iters)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    float x = 1.0001f + threadIdx.x * 1e-6f;
    unsigned p = 0x3c003c00u + threadIdx.x;      // two halves
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(x));
                if (KIND == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(p), "v"(x));
                if (KIND == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
                if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(x));
                if (KIND == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(p));
                if (KIND == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(p));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.0f;
    if (threadIdx.x == 0) ((long long *)out)[0] = 0;      // keep
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long *>(out + 4096)[0] = t1 - t0;
}

template <int KIND> void run(const char *name, float *d)
{
    const int iters = 2000;
    for (int threads : {256, 512, 1024}) {          // 1, 2, 4 wavefronts per SIMD (one workgroup per CU)
        hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipEvent_t b, e;
        hipEventCreate(&b); hipEventCreate(&e);
        hipEventRecord(b);
        hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, b, e);
        long long cyc;
        hipMemcpy(&cyc, d + 4096, 8, hipMemcpyDeviceToHost);
        const double instr_per_simd = (double)iters * REP * (threads / 256);
        printf("%-14s waves/SIMD %d  memtime-cycles/instr/SIMD %.2f   (kernel %.3f ms, %.2f G wave-instr/s/SIMD)\n", name, threads / 256,
               (double)cyc / instr_per_simd, ms, instr_per_simd / ms * 1e-6);
    }
}

int main()
{
    float *d;
    hipMalloc(&d, 1 << 22);
    run<0>("v_fma_f32", d);
    run<1>("v_fma_mix_f32", d);
    run<2>("v_add_f32", d);
    run<3>("v_mov_dpp", d);
    run<4>("v_add_u32", d);
    run<5>("v_cvt_f32_f16", d);
    return 0;
}
