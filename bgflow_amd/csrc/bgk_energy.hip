/* bgk_energy.hip -- energy of the isotropic (optionally shifted) normal distribution, the synthetic target of the cfg 3 / cfg 5
 * generators and the prior of cfg 1 / cfg 2 (bgflow/distribution/normal.py:61-72, `NormalDistribution._energy` without `cov`):
 *   u(x) = 0.5 sum_j ((x_j - mean_j) / sqrt(T))^2 + log_z          (log_z = d / 2 log(2 pi T), computed by the caller)
 * one launch instead of the sub / div / pow / sum / add chain (5 elementwise launches + a row reduction), and one launch for the
 * VJP g_x = g_u (x - mean) / T instead of five.  This is the "target energy" end of the KL integrand u(F(z)) - log|det J|
 * (bgflow/bg.py:13-17).
 * Roofline: HBM, 4 (d + 1) B per sample forward, 4 (2 d + 1) B backward.  Rows of the tile are staged coalesced through LDS
 * (odd stride), one lane per row adds its d terms in ascending order (deterministic). */
#include "bgk_common.h"

namespace {

constexpr int NE_THREADS = 256;
constexpr int NE_ROWS = 128;          /* rows per tile */

struct NormalEnergyArgs {
    const float* x; int64_t ldx; const float* mean; int d; int64_t B;
    float inv_t, log_z; float* u; uint32_t magic_d;
};

__global__ __launch_bounds__(NE_THREADS) void normal_energy_kernel(NormalEnergyArgs a) {
    extern __shared__ float s_x[];
    const int d = a.d, sd = d | 1, tid = threadIdx.x;
    const int64_t n_tiles = (a.B + NE_ROWS - 1) / NE_ROWS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * NE_ROWS;
        const int rows = (int)((a.B - b0) < NE_ROWS ? (a.B - b0) : NE_ROWS);
        for (int i = tid; i < rows * d; i += NE_THREADS) {
            const int r = (int)__umulhi((unsigned)i, a.magic_d), c = i - r * d;
            const float v = a.x[(b0 + r) * a.ldx + c] - (a.mean ? a.mean[c] : 0.0f);
            s_x[r * sd + c] = v * v;
        }
        __syncthreads();
        if (tid < rows) {
            float acc = 0.0f;
            for (int c = 0; c < d; ++c) acc += s_x[tid * sd + c];
            a.u[b0 + tid] = 0.5f * acc * a.inv_t + a.log_z;
        }
        __syncthreads();
    }
}

struct NormalEnergyBwdArgs {
    const float* x; int64_t ldx; const float* mean; int d; int64_t B;
    float inv_t; const float* g_u; float* g_x; int64_t ldg; uint32_t magic_d;
};

__global__ __launch_bounds__(NE_THREADS) void normal_energy_bwd_kernel(NormalEnergyBwdArgs a) {
    const int d = a.d;
    const int64_t total = a.B * d;
    for (int64_t i0 = (int64_t)blockIdx.x * NE_THREADS * 4; i0 < total; i0 += (int64_t)gridDim.x * NE_THREADS * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + (int64_t)u * NE_THREADS + threadIdx.x;
            if (i < total) {
                const int64_t r = i / d;           /* 64-bit: B d can exceed 2^32 */
                const int c = (int)(i - r * d);
                a.g_x[r * a.ldg + c] = a.g_u[r] * (a.x[r * a.ldx + c] - (a.mean ? a.mean[c] : 0.0f)) * a.inv_t;
            }
        }
    }
}

}  // namespace

extern "C" int bgk_normal_energy(const float* x, int64_t ldx, const float* mean, int32_t d, int64_t B,
                                 double temperature, double log_z, float* u, void* stream) {
    BGK_CHECK_ARG(x && u, "bgk_normal_energy: null pointer");
    BGK_CHECK_ARG(B >= 0 && d > 0 && ldx >= d && temperature > 0.0, "bgk_normal_energy: bad sizes");
    BGK_CHECK_ARG((size_t)NE_ROWS * (size_t)(d | 1) * sizeof(float) <= 160 * 1024, "bgk_normal_energy: %d dims do not fit the LDS tile", d);
    if (B == 0) return 0;
    NormalEnergyArgs a{x, ldx, mean, d, B, (float)(1.0 / temperature), (float)log_z, u,
                       (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d)};
    const size_t shmem = (size_t)NE_ROWS * (size_t)(d | 1) * sizeof(float);
    const int64_t n_tiles = (B + NE_ROWS - 1) / NE_ROWS;
    const int grid = (int)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(normal_energy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(normal_energy_kernel, dim3(grid), dim3(NE_THREADS), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_normal_energy");
}

extern "C" int bgk_normal_energy_backward(const float* x, int64_t ldx, const float* mean, int32_t d, int64_t B,
                                          double temperature, const float* g_u, float* g_x, int64_t ldg, void* stream) {
    BGK_CHECK_ARG(x && g_u && g_x, "bgk_normal_energy_backward: null pointer");
    BGK_CHECK_ARG(B >= 0 && d > 0 && ldx >= d && ldg >= d && temperature > 0.0, "bgk_normal_energy_backward: bad sizes");
    if (B == 0) return 0;
    NormalEnergyBwdArgs a{x, ldx, mean, d, B, (float)(1.0 / temperature), g_u, g_x, ldg, 0u};
    const int64_t blocks = (B * d + NE_THREADS * 4 - 1) / (NE_THREADS * 4);
    const int grid = (int)(blocks < 256 * 16 ? blocks : 256 * 16);
    hipLaunchKernelGGL(normal_energy_bwd_kernel, dim3(grid), dim3(NE_THREADS), 0, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_normal_energy_backward");
}
