/* bgk_reduce.hip -- column sums of a tall row-major matrix: the bias gradient of a Linear layer,
 * grad_bias = grad_out.sum(0) (autograd of nn/dense.py:47-48 in the KL / NLL training step).
 * PyTorch's generic column reduction reads a [2^18, 425] operand at ~170 GB/s on MI355X (2.7 ms per layer, a third
 * of a training step); this is a plain coalesced two-stage reduction at HBM speed, deterministic (fixed
 * partition, fixed order).  Roofline: HBM, 4 B per element read once.
 */
#include "bgk_common.h"

namespace {

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ldx, int64_t B, int P,
                                                             int64_t rows_per_block, float* __restrict__ partial) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block) < B ? (r0 + rows_per_block) : B;
    for (int col = threadIdx.x; col < P; col += 256) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            a0 += x[(r + 0) * ldx + col];
            a1 += x[(r + 1) * ldx + col];
            a2 += x[(r + 2) * ldx + col];
            a3 += x[(r + 3) * ldx + col];
        }
        for (; r < r1; ++r) a0 += x[r * ldx + col];
        partial[(int64_t)blockIdx.x * P + col] = (a0 + a1) + (a2 + a3);
    }
}

/* 64 columns x 4 row segments per block; each thread sums its segment with 4 independent accumulators, the 4
 * segment sums are combined in fixed order through LDS */
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int P, float* __restrict__ out) {
    __shared__ float s[4][64];
    const int c = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int per = (nblk + 3) / 4;
    const int b0 = seg * per, b1 = (b0 + per) < nblk ? (b0 + per) : nblk;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (col < P) {
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            a0 += partial[(int64_t)(b + 0) * P + col];
            a1 += partial[(int64_t)(b + 1) * P + col];
            a2 += partial[(int64_t)(b + 2) * P + col];
            a3 += partial[(int64_t)(b + 3) * P + col];
        }
        for (; b < b1; ++b) a0 += partial[(int64_t)b * P + col];
    }
    s[seg][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (seg == 0 && col < P) out[col] = (s[0][c] + s[1][c]) + (s[2][c] + s[3][c]);
}

/* largest magnitude of a row-major [B, P] matrix, raised into out[0] (non-negative floats order like their bit patterns) */
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t ldx, int64_t B, int P, float* out) {
    float m = 0.0f;
    const int64_t n = B * (int64_t)P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / P;
        m = __builtin_fmaxf(m, __builtin_fabsf(x[r * ldx + (i - r * P)]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off));
    const unsigned mb = __builtin_bit_cast(unsigned, m);
    if ((threadIdx.x & 63) == 0 && mb > *reinterpret_cast<volatile unsigned*>(out)) atomicMax(reinterpret_cast<unsigned*>(out), mb);
}

}  // namespace

extern "C" int bgk_absmax(const float* x, int64_t ldx, int64_t B, int32_t P, float* out, void* stream) {
    if (B == 0) return 0;
    BGK_CHECK_ARG(x && out, "bgk_absmax: null pointer");
    BGK_CHECK_ARG(B >= 0 && P > 0 && ldx >= P, "bgk_absmax: bad sizes");
    const int64_t n = B * (int64_t)P;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, B, (int)P, out);
    return bgk_launch_status("bgk_absmax");
}

extern "C" int bgk_column_sum(const float* x, int64_t ldx, int64_t B, int32_t P, float* partial, int32_t nblk,
                              float* out, void* stream) {
    if (B == 0) {               /* the sum over an empty batch is zero (x has no storage: null pointer) */
        if (out && P > 0 && hipMemsetAsync(out, 0, sizeof(float) * (size_t)P, (hipStream_t)stream) != hipSuccess) return BGK_EINVAL;
        return 0;
    }
    BGK_CHECK_ARG(x && partial && out, "bgk_column_sum: null pointer");
    BGK_CHECK_ARG(B >= 0 && P > 0 && nblk > 0 && ldx >= P, "bgk_column_sum: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rpb = (B + nblk - 1) / nblk;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, x, ldx, B, (int)P, rpb > 0 ? rpb : 1, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((P + 63) / 64), dim3(256), 0, st, partial, (int)nblk, (int)P, out);
    return bgk_launch_status("bgk_column_sum");
}
