// Device-side pieces shared by the GAT aggregate's kernels (gat.hip, gat_scan.hip).
#pragma once
#include "sgx_device.h"

#include <math.h>

namespace {

__device__ __forceinline__ float leaky(float x, float alpha) { return x > 0.0f ? x : x * alpha; }

// exp(x) for the softmax weights of the two-stage aggregate, x = score - row maximum <= 0.  -D SGX_GAT_FAST_EXP: the
// hardware's v_exp_f32 on x log2(e) -- 2 instructions instead of expf's dozen, relative error about (2 + |x|) 2^-24
// instead of 2^-24; measured 3 % on the 8-head aggregates (arxiv shape 0.287 -> 0.276 ms, 29 M-entry R-MAT 1.93 -> 1.86),
// 1 % with one head, tests/test_gpu_gat_scan.py green with it.  Off: the weights are the library's only fp32 output
// besides E, and 3 % does not buy the last digits of the small ones.
#ifdef SGX_GAT_FAST_EXP
__device__ __forceinline__ float exp_weight(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
#else
__device__ __forceinline__ float exp_weight(float x) { return expf(x); }
#endif

// ReLU (SG.py:660-661), then the quantised layer's deq_o factor on fp32 outputs (SG.py:666-667; 0 = off)
template <typename T>
__device__ __forceinline__ T gat_finish(float sum, int relu, float out_scale)
{
    T v = Elem<T>::from_f32(sum);
    v = (!relu || v > (T)0) ? v : (T)0;
    if constexpr (sizeof(T) == 4) {
        if (out_scale != 0.0f) v = v * out_scale;
    }
    return v;
}

// merge two online-softmax states (m, l); (-inf, 0) is the empty state
__device__ __forceinline__ void softmax_merge(float &m, float &l, float m2, float l2)
{
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; l = 0.0f; return; }
    l = l * expf(m - mn) + l2 * expf(m2 - mn);
    m = mn;
}

__device__ __forceinline__ float rescale_factor(float m_old, float m_new)
{
    return m_old == -INFINITY ? 0.0f : expf(m_old - m_new);        // (-inf) - (-inf) never reaches expf
}

// the scores of heads [hb0, hb0 + 8) of node c: two 16-byte loads when the row of 8 floats is aligned (n_heads a
// multiple of 8 and an aligned table), else element loads; entries past n_heads are 0
__device__ __forceinline__ void load_scores8(const float *__restrict__ s2, int64_t c, int n_heads, int hb0, bool vec, float *out)
{
    if (vec) {
        const float4 a = *reinterpret_cast<const float4 *>(s2 + c * n_heads + hb0);
        const float4 b = *reinterpret_cast<const float4 *>(s2 + c * n_heads + hb0 + 4);
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = hb0 + k < n_heads ? s2[c * n_heads + hb0 + k] : 0.0f;
    }
}
__device__ __forceinline__ void store8(float *__restrict__ dst, int64_t idx, int n_heads, int hb0, bool vec, const float *v)
{
    if (vec) {
        *reinterpret_cast<float4 *>(dst + idx * n_heads + hb0) = float4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<float4 *>(dst + idx * n_heads + hb0 + 4) = float4{v[4], v[5], v[6], v[7]};
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (hb0 + k < n_heads) dst[idx * n_heads + hb0 + k] = v[k];
    }
}

// HB heads [hb0, hb0 + HB) of one node / one entry (HB a divisor of n_heads; vec: 16-byte accesses are aligned)
template <int HB>
__device__ __forceinline__ void load_scores(const float *__restrict__ s, int64_t node, int n_heads, int hb0, bool vec, float *out)
{
    if constexpr (HB == 8) {
        load_scores8(s, node, n_heads, hb0, vec, out);
    } else if constexpr (HB == 4) {
        if (vec) {
            const float4 a = *reinterpret_cast<const float4 *>(s + node * n_heads + hb0);
            out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = s[node * n_heads + hb0 + k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < HB; ++k) out[k] = s[node * n_heads + hb0 + k];
    }
}
template <int HB>
__device__ __forceinline__ void store_heads(float *__restrict__ dst, int64_t idx, int n_heads, int hb0, bool vec, const float *v)
{
    if constexpr (HB == 8) {
        store8(dst, idx, n_heads, hb0, vec, v);
    } else if constexpr (HB == 4) {
        if (vec) *reinterpret_cast<float4 *>(dst + idx * n_heads + hb0) = float4{v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[idx * n_heads + hb0 + k] = v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < HB; ++k) dst[idx * n_heads + hb0 + k] = v[k];
    }
}

}  // namespace
