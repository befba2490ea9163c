// Integer operands for the quantised layer's X.W (dense X): the codes of the w_qbits grids stored as bytes and
// multiplied on the int8 matrix cores (v_mfma_i32_16x16x64_i8), instead of fp32 values on a w_qbits grid multiplied
// in fp32 (quant.hip).  What the reference's EIGHTBIT / quantised bitstreams do in hardware (MM.h:85-118: 8/16-bit
// integer types for A, B, D; SG.py:570-616 states the arithmetic): X is read as 1 byte per element instead of 4.
//
//   x_code = clip(round(x / f_s + f_z), 0, 2^b - 1)            unsigned (features)         value = x_code / 2^(b-1)
//   w_code = clip(round(w / w_s + w_z), -(2^(b-1) - 1), ..)     signed   (weights)          value = w_code / 2^(b-1)
//   H[r][p] = requant( (sum_k x_code w_code) * 2^-(2(b-1)) )   the same fp32 shift / clip / decimal rounding as the
//                                                              fp32 form (sgx_requant_value), on an EXACT sum
// An unsigned 8-bit code does not fit a signed byte: it is stored minus 128 and the product is repaired with the column
// sums of W (sum_k (x - 128) w = sum_k x w - 128 sum_k w).  The integer sum is exact; the fp32 emulation of the
// reference rounds once its partial sums pass 2^24 units, so the two agree bit for bit exactly when the emulation's own
// sums are exact (|sum| < 2^24: M_fea up to 518 at 8 bits, any M_fea in practice at 4 bits and below) and to fp32
// rounding otherwise -- with the integer form the more exact of the two.  Parity unpinned, like the rest of the
// quantised layer (the reference records no output of it).
//
// Kernel: the transposed tile H^T = W^T . X^T, as xw_dense.hip: both operands are 16 contiguous bytes per lane along K
// straight from row-major storage (W^T [P][ldw] is how the reference stores B), and a lane ends with 4 consecutive
// columns of one row of H (one 16-byte store of fp32).  A wavefront owns 16 rows of X and all P columns; X is read
// once, W from L1 / L2.
#include "sgx_device.h"

#pragma clang fp contract(off)

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float clipf8(float v, float lo, float hi)
{
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

// codes[r][c] = the integer code of x[r][c] (minus `bias`), columns n_cols..ldc-1 zero.  One lane per 4 codes (ldc is
// a multiple of 16): consecutive lanes take consecutive groups of 4 columns -- 16 bytes in, one dword of codes out.
__global__ __launch_bounds__(kBlock) void quantize_codes_kernel(int kind, int qbits, float inv_scale, float zero, int bias,
                                                                int64_t n_rows, int n_cols, const float *__restrict__ x,
                                                                int64_t ldx, signed char *__restrict__ codes, int64_t ldc)
{
    const float lo = kind ? -(float)((1 << (qbits - 1)) - 1) : 0.0f;
    const float hi = kind ? (float)((1 << (qbits - 1)) - 1) : (float)((1 << qbits) - 1);
    const int64_t chunks = ldc / 4, total = n_rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / chunks;
        const int c0 = (int)(i - r * chunks) * 4;
        union { unsigned v; signed char b[4]; } u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int q = 0;
            if (c0 + j < n_cols) {
                const float t = inv_scale * x[r * ldx + c0 + j] + zero;          // 1 / s * x + z, as quant.hip
                float f;
                if (qbits == 1 && kind == 1) f = t < 0.0f ? -1.0f : 1.0f;        // fake_quantization_b: -+0.5 = -+1 / 2
                else if (qbits == 1) f = clipf8(rintf(t), 0.0f, 1.0f);
                else f = clipf8(rintf(t), lo, hi);
                q = (int)f - bias;
            }
            u.b[j] = (signed char)q;
        }
        *reinterpret_cast<unsigned *>(codes + r * ldc + c0) = u.v;
    }
}

// wsum[p] = sum_k W[p][k]
__global__ __launch_bounds__(64) void code_row_sums_kernel(int P, int M, const signed char *__restrict__ Wc, int64_t ldw,
                                                          int *__restrict__ wsum)
{
    const int p = blockIdx.x;
    if (p >= P) return;
    int s = 0;
    for (int k = threadIdx.x; k < M; k += 64) s += Wc[(int64_t)p * ldw + k];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) wsum[p] = s;
}

// 16 rows of X x (16 NT) columns per wavefront; K in steps of 64 (16 bytes per lane)
template <int NT>
__global__ __launch_bounds__(kBlock) void xw_i8_kernel(int n_rows, int M, int P, const signed char *__restrict__ Xc, int64_t ldx,
                                                       const signed char *__restrict__ Wc, int64_t ldw, const int *__restrict__ wsum,
                                                       int x_bias, float code_scale, sgx_epilogue ep, float *__restrict__ H,
                                                       int64_t ldh)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
    const int rl = lane & 15, kq = lane >> 4;                      // operand row inside the tile, 16-byte quarter of the k-step
    for (int64_t r0 = wave * 16; r0 < n_rows; r0 += n_waves * 16) {
        i32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = i32x4{0, 0, 0, 0};
        const int64_t xr = r0 + rl < n_rows ? r0 + rl : n_rows - 1;               // rows past the end: computed, never stored
        for (int k0 = 0; k0 < M; k0 += 64) {
            const int k = k0 + 16 * kq;
            i32x4 xb = i32x4{0, 0, 0, 0};
            if (k < M) xb = *reinterpret_cast<const i32x4 *>(Xc + xr * ldx + k);     // pad columns up to ldx are zero codes
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int p = t * 16 + rl;
                i32x4 wb = i32x4{0, 0, 0, 0};
                if (k < M && p < P) wb = *reinterpret_cast<const i32x4 *>(Wc + (int64_t)p * ldw + k);
                // D[i = p][j = row] += sum_k W[p][k] X[row][k]
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wb, xb, acc[t], 0, 0, 0);
            }
        }
        // lane: column (row of X) = lane & 15, rows (output columns p) = 4 (lane >> 4) + i
        const int64_t r = r0 + (lane & 15);
        if (r < n_rows) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int p0 = t * 16 + 4 * (lane >> 4);
                float out[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int p = p0 + i;
                    const int total = acc[t][i] + (p < P ? x_bias * wsum[p] : 0);
                    float v = (float)total * code_scale;                           // exact below 2^24
                    if (ep.rq_ten_pow != 0.0f) v = sgx_requant_value(v, ep);
                    out[i] = v;
                }
                if (p0 + 4 <= P && ((ldh & 3) == 0)) {
                    *reinterpret_cast<float4 *>(H + r * ldh + p0) = float4{out[0], out[1], out[2], out[3]};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (p0 + i < P) H[r * ldh + p0 + i] = out[i];
                }
            }
        }
    }
}

int frac_bits(int qbits) { return qbits == 1 ? 1 : qbits - 1; }      // value = code / 2^frac_bits (1 bit: -+1/2, 0 | 1/2)

}  // namespace

extern "C" int sgx_code_bias(int is_signed, int qbits) { return (!is_signed && qbits == 8) ? 128 : 0; }

extern "C" int sgx_quantize_codes_i8(int is_signed, int qbits, float inv_scale, float zero, int n_rows, int n_cols, const float *x,
                                     int64_t ldx, int8_t *codes, int64_t ldc, void *stream)
{
    if (n_rows < 0 || n_cols < 1 || ldx < n_cols || ldc < n_cols) return SGX_ERR_SHAPE;
    if (ldc % 16 != 0 || (uintptr_t)codes % 16 != 0) return SGX_ERR_ALIGN;
    if (qbits < 1 || qbits > 8 || (is_signed != 0 && is_signed != 1)) return SGX_ERR_UNSUPPORTED;
    if (n_rows == 0) return SGX_OK;
    if (!x || !codes) return SGX_ERR_NULL;
    const int64_t total = (int64_t)n_rows * (ldc / 4);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(quantize_codes_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, is_signed, qbits,
                       inv_scale, zero, sgx_code_bias(is_signed, qbits), (int64_t)n_rows, n_cols, x, ldx, (signed char *)codes, ldc);
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}

extern "C" size_t sgx_xw_dense_i8_workspace_bytes(int P) { return P < 1 ? 0 : sgx_align_up((size_t)P * sizeof(int), 256); }

extern "C" int sgx_xw_dense_i8(int qbits, int n_rows, int M_fea, int P, const int8_t *Xc, int64_t ldx, const int8_t *Wc, int64_t ldw,
                               int scale_fea, int internal_bits, float *H, int64_t ldh, void *workspace, void *stream)
{
    if (n_rows < 0 || M_fea < 1 || P < 1 || ldx < M_fea || ldw < M_fea || ldh < P) return SGX_ERR_SHAPE;
    if (qbits < 1 || qbits > 8 || scale_fea < 0 || scale_fea > 30 || internal_bits < 0 || internal_bits > 30) return SGX_ERR_UNSUPPORTED;
    if (P > 256) return SGX_ERR_UNSUPPORTED;                       // 16 column tiles of accumulators per wavefront
    if (n_rows == 0) return SGX_OK;
    if (!Xc || !Wc || !H) return SGX_ERR_NULL;
    if (!workspace) return SGX_ERR_WORKSPACE;
    // 16-byte operand loads: rows on 16 bytes, pad columns (zero codes) up to a multiple of 16
    if ((uintptr_t)Xc % 16 || (uintptr_t)Wc % 16 || ldx % 16 || ldw % 16 || (uintptr_t)H % 16) return SGX_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    int *wsum = (int *)workspace;
    hipLaunchKernelGGL(code_row_sums_kernel, dim3(P), dim3(64), 0, s, P, M_fea, (const signed char *)Wc, ldw, wsum);
    SGX_LAUNCH_CHECK();
    sgx_epilogue ep = sgx_no_epilogue();
    if (internal_bits > 0) ep = sgx_requant_epilogue(scale_fea, internal_bits);
    const float code_scale = 1.0f / (float)(1 << (2 * frac_bits(qbits)));
    const int x_bias = sgx_code_bias(0, qbits);
    int64_t blocks = ((int64_t)n_rows + 16 * (kBlock / 64) - 1) / (16 * (kBlock / 64));
    if (blocks > 256 * 8) blocks = 256 * 8;
    const int nt = (P + 15) / 16;
#define SGX_I8_CASE(N)                                                                                                          \
    hipLaunchKernelGGL((xw_i8_kernel<N>), dim3((unsigned)blocks), dim3(kBlock), 0, s, n_rows, M_fea, P, (const signed char *)Xc, ldx, \
                       (const signed char *)Wc, ldw, wsum, x_bias, code_scale, ep, H, ldh)
    if (nt <= 1) SGX_I8_CASE(1);
    else if (nt <= 2) SGX_I8_CASE(2);
    else if (nt <= 4) SGX_I8_CASE(4);
    else if (nt <= 8) SGX_I8_CASE(8);
    else SGX_I8_CASE(16);
#undef SGX_I8_CASE
    SGX_LAUNCH_CHECK();
    return SGX_OK;
}
