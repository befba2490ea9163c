// Device-side helpers shared by the aggregation kernels (spmm_csr.hip, gat.hip).
#pragma once
#include "sgx_internal.h"

constexpr int kBlock = 256;              // 4 wavefronts
constexpr unsigned kOOB = 0xFFFFFFF0u;   // buffer offset that is out of range for any table
// Row offset that stays out of range after a column offset below 1 MiB is added to it; tables up to
// kOOBRow bytes with rows under 1 MiB use 32-bit buffer offsets, larger ones 64-bit pointers.
constexpr unsigned kOOBRow = 0xFFF00000u;
constexpr unsigned kMaxRowBytes = 0x000FFFF0u;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


template <typename T> struct Elem;
template <> struct Elem<f16> {
    static constexpr int kVec = 8;       // elements per 16-byte gather
    typedef u32x4 vec16_u __attribute__((aligned(2)));   // 16 bytes at element alignment
    static __device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
    static __device__ __forceinline__ f16 from_f32(float v) { return (f16)v; }   // v_cvt_f16_f32: RNE
};
template <> struct Elem<float> {
    static constexpr int kVec = 4;
    typedef u32x4 vec16_u __attribute__((aligned(4)));
    static __device__ __forceinline__ float to_f32(float v) { return v; }
    static __device__ __forceinline__ float from_f32(float v) { return v; }
};

// acc[0:VEC] += a * (VEC elements of type T held in `raw`)
template <typename T, int VEC> struct Fma;
// the same in plain C: the compiler turns it into 8 converts + 4 packed fp32 fmas and is free to
// interleave them with the loads of later edges
struct FmaPlainF16 {
    static __device__ __forceinline__ void run(float *acc, float a, u32x4 raw) {
        union { u32x4 v; f16 h[8]; } u; u.v = raw;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(a, (float)u.h[i], acc[i]);
    }
};
template <> struct Fma<f16, 8> {
    // v_fma_mix_f32 reads one half of a packed register as an fp32 operand: acc += a * float(h), one
    // instruction per element and the same single rounding as convert + fma (the compiler's own choice,
    // 8 converts + 4 packed fp32 fmas, is 12 instructions per 16 bytes)
    static __device__ __forceinline__ void run(float *acc, float a, u32x4 raw) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned pair = raw[i];
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * i]) : "v"(pair), "v"(a));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * i + 1]) : "v"(pair), "v"(a));
        }
    }
};
// acc[0:VEC] = a * raw + (+0): the first term of a sum -- the same value as Fma onto a register holding +0
template <typename T, int VEC> struct FmaInit;
template <> struct FmaInit<f16, 8> {
    static __device__ __forceinline__ void run(float *acc, float a, u32x4 raw) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned pair = raw[i];
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(acc[2 * i]) : "v"(pair), "v"(a));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc[2 * i + 1]) : "v"(pair), "v"(a));
        }
    }
};
template <> struct FmaInit<float, 4> {
    static __device__ __forceinline__ void run(float *acc, float a, u32x4 raw) {
        union { u32x4 v; float f[4]; } u; u.v = raw;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(a, u.f[i], 0.0f);
    }
};
template <> struct Fma<float, 4> {
    static __device__ __forceinline__ void run(float *acc, float a, u32x4 raw) {
        union { u32x4 v; float f[4]; } u; u.v = raw;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_fmaf(a, u.f[i], acc[i]);
    }
};

#ifndef SGX_GATHER_AUX
#define SGX_GATHER_AUX 0     // cache-policy bits of the 16-byte gathers (experiment knob: 1 = sc0, 2 = nt, 16 = sc1)
#endif
// One gather of VEC elements at byte offset `off` of the table behind `rsrc`.  load() and fma() are
// separate so that a caller can put all loads of a step ahead of the arithmetic in program order.
template <typename T, int VEC> struct GatherRaw;
template <> struct GatherRaw<f16, 8> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, SGX_GATHER_AUX);
    }
    static __device__ __forceinline__ void fma(float *acc, float a, raw_t raw) { Fma<f16, 8>::run(acc, a, raw); }
};
template <> struct GatherRaw<float, 4> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, SGX_GATHER_AUX);
    }
    static __device__ __forceinline__ void fma(float *acc, float a, raw_t raw) { Fma<float, 4>::run(acc, a, raw); }
};
template <> struct GatherRaw<f16, 1> {
    typedef unsigned short raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b16(rsrc, off, 0, 0);
    }
    static __device__ __forceinline__ void fma(float *acc, float a, raw_t raw) {
        union { unsigned short s; f16 h; } u; u.s = raw;
        acc[0] = __builtin_fmaf(a, (float)u.h, acc[0]);
    }
};
template <> struct GatherRaw<float, 1> {
    typedef unsigned raw_t;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
    }
    static __device__ __forceinline__ void fma(float *acc, float a, raw_t raw) {
        acc[0] = __builtin_fmaf(a, __builtin_bit_cast(float, raw), acc[0]);
    }
};

template <typename T, int VEC> struct Gather;
template <> struct Gather<f16, 8> {
    static __device__ __forceinline__ void run(float *acc, float a, __amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        FmaPlainF16::run(acc, a, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, SGX_GATHER_AUX));
    }
};
template <> struct Gather<float, 4> {
    static __device__ __forceinline__ void run(float *acc, float a, __amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        Fma<float, 4>::run(acc, a, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, SGX_GATHER_AUX));
    }
};
template <> struct Gather<f16, 1> {
    static __device__ __forceinline__ void run(float *acc, float a, __amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        unsigned short raw = __builtin_amdgcn_raw_buffer_load_b16(rsrc, off, 0, 0);
        union { unsigned short s; f16 h; } u; u.s = raw;
        acc[0] = __builtin_fmaf(a, (float)u.h, acc[0]);
    }
};
template <> struct Gather<float, 1> {
    static __device__ __forceinline__ void run(float *acc, float a, __amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        acc[0] = __builtin_fmaf(a, __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0)), acc[0]);
    }
};


// The same gathers through a 64-bit pointer (tables of 4 GiB and more).
template <typename T, int VEC> struct GatherPtr;
template <> struct GatherPtr<f16, 8> {
    static __device__ __forceinline__ void run(float *acc, float a, const char *p) {
        Fma<f16, 8>::run(acc, a, *reinterpret_cast<const u32x4 *>(p));
    }
};
template <> struct GatherPtr<float, 4> {
    static __device__ __forceinline__ void run(float *acc, float a, const char *p) {
        Fma<float, 4>::run(acc, a, *reinterpret_cast<const u32x4 *>(p));
    }
};
template <> struct GatherPtr<f16, 1> {
    static __device__ __forceinline__ void run(float *acc, float a, const char *p) {
        acc[0] = __builtin_fmaf(a, (float)*reinterpret_cast<const f16 *>(p), acc[0]);
    }
};
template <> struct GatherPtr<float, 1> {
    static __device__ __forceinline__ void run(float *acc, float a, const char *p) {
        acc[0] = __builtin_fmaf(a, *reinterpret_cast<const float *>(p), acc[0]);
    }
};

// Broadcast of lane t of every LPR-lane group.  Groups of 2 or 4 lanes sit inside a quad, where DPP
// quad_perm moves data inside the VALU (no LDS crossbar trip, no address register); wider groups
// use ds_bpermute through __shfl.
template <int LPR, int TT> struct GroupBcast {
    static __device__ __forceinline__ int run(int x) { return __shfl(x, TT, LPR); }
};
template <int TT> struct GroupBcast<1, TT> {
    static __device__ __forceinline__ int run(int x) { return x; }
};
template <int TT> struct GroupBcast<4, TT> {
    static __device__ __forceinline__ int run(int x) {
        return __builtin_amdgcn_update_dpp(0, x, TT | (TT << 2) | (TT << 4) | (TT << 6), 0xF, 0xF, false);
    }
};
template <int TT> struct GroupBcast<2, TT> {
    static __device__ __forceinline__ int run(int x) {
        return __builtin_amdgcn_update_dpp(0, x, TT | (TT << 2) | ((2 + TT) << 4) | ((2 + TT) << 6), 0xF, 0xF, false);
    }
};
template <int LPR, int TT> __device__ __forceinline__ int group_bcast(int x) { return GroupBcast<LPR, TT>::run(x); }
template <int LPR, int TT> __device__ __forceinline__ float group_bcast(float x)
{
    return __builtin_bit_cast(float, GroupBcast<LPR, TT>::run(__builtin_bit_cast(int, x)));
}

// what one fp32 sum becomes in D: the stage's epilogue (fp32 outputs of the quantised layer only), then ReLU
template <typename T>
__device__ __forceinline__ T finish_value(float sum, int relu, const sgx_epilogue &ep)
{
    if constexpr (sizeof(T) == 4) {
        if (ep.rq_ten_pow != 0.0f) sum = sgx_requant_value(sum, ep);
    }
    T v = Elem<T>::from_f32(sum);
    v = (!relu || v > (T)0) ? v : (T)0;                    // K.cpp:2586-2590: keep when (v > 0 || relu == 0), else +0
    if constexpr (sizeof(T) == 4) {
        if (ep.out_scale != 0.0f) v = v * ep.out_scale;
    }
    return v;
}
