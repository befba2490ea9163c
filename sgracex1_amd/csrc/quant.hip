// The quantised SGRACE layer, as the reference states it in its emulation of the quantised
// bitstream (SG.py:570-667 with the helpers of SG.py:177-265): operands are rounded to w_qbits
// integers and put back on a fractional grid, H = X.W is shifted, clipped and rounded to the
// width of the internal pipeline, the aggregate is rescaled by deq_o.  Everything stays fp32, as
// in the reference; these kernels only reproduce its rounding points, one fp32 operation each.
// Inside sgx_layer_forward the re-quantisation of H and the deq_o factor ride on the stores of the two
// stages (sgx_epilogue, sgx_internal.h); the kernels here quantise the operands and serve the standalone
// entry points.
//
// fp contraction is off for this file: `1 / s * x + z` is a rounded product followed by a
// rounded sum in the reference (two torch ops), not one fma.
#include "sgx_internal.h"
#include "sgx_device.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float clipf(float v, float lo, float hi)
{
    // torch.clip: min(max(v, lo), hi); NaN propagates
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

// kind 0: quantization_ufbits (SG.py:253-265): unsigned grid 0 .. 2^q - 1
// kind 1: quantization_fbits  (SG.py:238-251): signed grid -(2^(q-1) - 1) .. 2^(q-1) - 1
__global__ __launch_bounds__(kBlock) void fake_quantize_kernel(int kind, int qbits, float inv_scale, float zero,
                                                               int64_t n, const float *__restrict__ x,
                                                               float *__restrict__ out)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const float lo = kind ? -(float)((1 << (qbits - 1)) - 1) : 0.0f;
    const float hi = kind ? (float)((1 << (qbits - 1)) - 1) : (float)((1 << qbits) - 1);
    const float back = 1.0f / (float)(1 << (qbits - 1));          // x_q / 2^(w_qbits - 1), SG.py:220
    for (int64_t i = gid; i < n; i += stride) {
        const float t = inv_scale * x[i] + zero;                   // 1 / s * x + z
        float q;
        if (qbits == 1 && kind == 1)
            q = t < 0.0f ? -0.5f : 0.5f;                           // fake_quantization_b, SG.py:177-182
        else if (qbits == 1)
            q = clipf(rintf(t), 0.0f, 1.0f) * 0.5f;                // fake_quantization_b2, SG.py:184-189
        else
            q = clipf(rintf(t), lo, hi) * back;                    // fake_quantization, SG.py:191-235
        out[i] = q;
    }
}

// SG.py:607-616: Wh / 2^scale_fea, clip to +-(2^iq - 1) / 2^iq, torch.round(decimals = iq - 1)
// (= nearbyint(v * 10^d) / 10^d in fp32).  Pad columns n_feat..ld-1 are left as they are (zero).
__global__ __launch_bounds__(kBlock) void requantize_kernel(int64_t n_rows, int n_feat, int64_t ld, float *__restrict__ H,
                                                            float shift, float bound, float ten_pow)
{
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t n = n_rows * n_feat;
    for (int64_t i = gid; i < n; i += stride) {
        const int64_t r = i / n_feat;
        const int c = (int)(i - r * n_feat);
        float v = H[r * ld + c] * shift;
        v = clipf(v, -bound, bound);
        H[r * ld + c] = rintf(v * ten_pow) / ten_pow;
    }
}

int grid_for(int64_t n)
{
    int64_t b = (n + kBlock - 1) / kBlock;
    if (b > 4096) b = 4096;
    return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" int sgx_fake_quantize(int is_signed, int qbits, float inv_scale, float zero, int64_t n, const float *x,
                                 float *out, void *stream)
{
    if (n < 0) return SGX_ERR_SHAPE;
    if (qbits < 1 || qbits > 16 || (is_signed != 0 && is_signed != 1)) return SGX_ERR_UNSUPPORTED;
    if (n == 0) return SGX_OK;
    if (!x || !out) return SGX_ERR_NULL;
    hipLaunchKernelGGL(fake_quantize_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, is_signed, qbits,
                       inv_scale, zero, n, x, out);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

sgx_epilogue sgx_requant_epilogue(int scale_fea, int internal_bits)
{
    sgx_epilogue ep = sgx_no_epilogue();
    ep.rq_shift = 1.0f / (float)(1 << scale_fea);                            // exact
    const double full = (double)(1u << internal_bits);
    ep.rq_bound = (float)((full - 1.0) / full);                              // a_max of SG.py:609, as torch casts it
    double p = 1.0;                                                          // static_cast<float>(std::pow(10, d))
    for (int i = 0; i < internal_bits - 1; ++i) p *= 10.0;
    ep.rq_ten_pow = (float)p;
    return ep;
}

extern "C" int sgx_requantize(int n_rows, int n_feat, int64_t ld, float *H, int scale_fea, int internal_bits, void *stream)
{
    if (n_rows < 0 || n_feat < 1 || ld < n_feat) return SGX_ERR_SHAPE;
    if (scale_fea < 0 || scale_fea > 30 || internal_bits < 1 || internal_bits > 30) return SGX_ERR_UNSUPPORTED;
    if (n_rows == 0) return SGX_OK;
    if (!H) return SGX_ERR_NULL;
    const sgx_epilogue ep = sgx_requant_epilogue(scale_fea, internal_bits);
    hipLaunchKernelGGL(requantize_kernel, dim3(grid_for((int64_t)n_rows * n_feat)), dim3(kBlock), 0, (hipStream_t)stream,
                       (int64_t)n_rows, n_feat, ld, H, ep.rq_shift, ep.rq_bound, ep.rq_ten_pow);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

