// Readout + head of the graph-classification model in one launch ("next" row f3; the pooling's backward for the
// training step at the end of the file): the tail of
// GCN_PYNQ.forward (MOL cell 18) after the second layer,
//     x = global_mean_pool(x.float(), batch);  x = lin(x)            (dropout is the identity in eval)
// i.e. logits[g][c] = bias[c] + sum_j W[c][j] * mean_{i in graph g} x[i][j].
// Nodes of one graph are contiguous (PyG batching, `batch` sorted), so a graph is a row segment
// [ptr[g], ptr[g+1]).  One workgroup per graph: the segment mean is accumulated in fp32 with one
// column per lane (coalesced row reads), then the C output classes are C wave reductions.
#include "sgx_internal.h"

namespace {

constexpr int kBlock = 256;

template <typename T>
__global__ __launch_bounds__(kBlock) void readout_mean_linear_kernel(
    int n_graphs, int F, int C, const T *__restrict__ X, int64_t ldx, const int32_t *__restrict__ ptr,
    const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ pooled, float *__restrict__ logits)
{
    extern __shared__ float mean[];                       // [F]
    const int g = blockIdx.x;
    if (g >= n_graphs) return;                            // (the grid is one workgroup per graph)
    const int r0 = ptr[g], r1 = ptr[g + 1];
    const float inv = r1 > r0 ? 1.0f / (float)(r1 - r0) : 0.0f;
    for (int j = threadIdx.x; j < F; j += kBlock) {
        float s = 0.0f;
        for (int r = r0; r < r1; ++r) s += (float)X[(int64_t)r * ldx + j];
        s *= inv;
        mean[j] = s;
        if (pooled) pooled[(int64_t)g * F + j] = s;
    }
    __syncthreads();
    if (!logits) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = wave; c < C; c += kBlock / 64) {
        float s = 0.0f;
        for (int j = lane; j < F; j += 64) s = __builtin_fmaf(W[(int64_t)c * F + j], mean[j], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) logits[(int64_t)g * C + c] = s + (bias ? bias[c] : 0.0f);
    }
}

// the pooling's backward: one workgroup per graph, a column per thread, the graph's rows one after the other
template <typename T>
__global__ __launch_bounds__(kBlock) void readout_mean_backward_kernel(int n_graphs, int F, const float *__restrict__ grad_pooled,
                                                                       const int32_t *__restrict__ ptr, T *__restrict__ grad_x,
                                                                       int64_t ldg)
{
    const int g = blockIdx.x;
    if (g >= n_graphs) return;
    const int r0 = ptr[g], r1 = ptr[g + 1];
    if (r1 <= r0) return;
    const float inv = 1.0f / (float)(r1 - r0);
    for (int j = threadIdx.x; j < F; j += kBlock) {
        const T v = (T)(grad_pooled[(int64_t)g * F + j] * inv);
        for (int r = r0; r < r1; ++r) grad_x[(int64_t)r * ldg + j] = v;
    }
}

}  // namespace

extern "C" int sgx_readout_mean_backward(int dtype, int n_graphs, int F, const float *grad_pooled, const int32_t *graph_ptr,
                                         void *grad_X, int64_t ldg, void *stream)
{
    if (n_graphs < 0 || F < 1 || ldg < F) return SGX_ERR_SHAPE;
    if (n_graphs == 0) return SGX_OK;
    if (!grad_pooled || !graph_ptr || !grad_X) return SGX_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SGX_F16)
        hipLaunchKernelGGL(readout_mean_backward_kernel<f16>, dim3(n_graphs), dim3(kBlock), 0, s, n_graphs, F, grad_pooled, graph_ptr,
                           (f16 *)grad_X, ldg);
    else if (dtype == SGX_F32)
        hipLaunchKernelGGL(readout_mean_backward_kernel<float>, dim3(n_graphs), dim3(kBlock), 0, s, n_graphs, F, grad_pooled, graph_ptr,
                           (float *)grad_X, ldg);
    else
        return SGX_ERR_UNSUPPORTED;
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

extern "C" int sgx_readout_mean_linear(int dtype, int n_graphs, int F, int C, const void *X, int64_t ldx,
                                       const int32_t *graph_ptr, const float *W, const float *bias, float *pooled,
                                       float *logits, void *stream)
{
    if (n_graphs < 0 || F < 1 || C < 0 || ldx < F) return SGX_ERR_SHAPE;
    if (n_graphs == 0) return SGX_OK;
    if (!X || !graph_ptr || (!pooled && !logits) || (logits && !W)) return SGX_ERR_NULL;
    if (F * sizeof(float) > 64 * 1024) return SGX_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)F * sizeof(float);
    if (dtype == SGX_F16)
        hipLaunchKernelGGL(readout_mean_linear_kernel<f16>, dim3(n_graphs), dim3(kBlock), lds, s, n_graphs, F, C,
                           (const f16 *)X, ldx, graph_ptr, W, bias, pooled, logits);
    else if (dtype == SGX_F32)
        hipLaunchKernelGGL(readout_mean_linear_kernel<float>, dim3(n_graphs), dim3(kBlock), lds, s, n_graphs, F, C,
                           (const float *)X, ldx, graph_ptr, W, bias, pooled, logits);
    else
        return SGX_ERR_UNSUPPORTED;
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}
