/* bgk_optim.hip -- optimizer step of the training loop on the flat parameter / gradient bucket:
 *   bgk_grad_nan_flag   flag[0] = 1 if any gradient is NaN (the reference KLTrainer skips the optimizer step in that case,
 *                       nn/training/trainers.py:198-201 -- there with a host round trip per parameter tensor)
 *   bgk_adam_step       torch.optim.Adam's update (bias-corrected first / second moments, eps outside the square root, optional
 *                       L2 weight decay) over ONE contiguous bucket, skipped on the device when flag[0] != 0; counts skipped steps
 * One launch each instead of torch's per-tensor-list multi-tensor kernels + the NaN scans; HBM-bound: 4 x 4 B read + 3 x 4 B
 * written per parameter (1.07 M parameters for cfg 3: microseconds). */
#include "bgk_common.h"

namespace {

__global__ __launch_bounds__(256) void nan_flag_kernel(const float* g, int64_t n, int32_t* flag) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= (g[i] != g[i]);
    if (__builtin_amdgcn_ballot_w64(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; int64_t n;
    float lr, beta1, beta2, eps, weight_decay;
    int64_t step;                                                  /* number of bgk_adam_step calls so far, this one included */
    const int32_t* flag; int32_t* skipped;
};

/* Adam's time step is `step - skipped[0]`: the skip count is read on the device, so an update skipped for a NaN gradient does not
 * advance the bias corrections 1 - beta^t (the reference does not call optim.step() at all in that case, trainers.py:198-201)
 * and the host never has to know whether a step was skipped.  Races: the skip path only writes `skipped`, the update path
 * only reads it. */
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    if (a.flag && a.flag[0] != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.skipped) atomicAdd(a.skipped, 1);
        return;
    }
    __shared__ float s_bc[2];
    if (threadIdx.x == 0) {
        const double t = (double)(a.step - (a.skipped ? (int64_t)a.skipped[0] : 0));
        s_bc[0] = (float)(1.0 - pow((double)a.beta1, t));               /* bc1 = 1 - beta1^t */
        s_bc[1] = (float)sqrt(1.0 - pow((double)a.beta2, t)); /* sqrt(1 - beta2^t) */
    }
    __syncthreads();
    const float step_size = a.lr / s_bc[0], bc2_sqrt = s_bc[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        float g = a.g[i];
        const float p = a.p[i];
        if (a.weight_decay != 0.0f) g = g + a.weight_decay * p;
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * (g * g);
        a.m[i] = m; a.v[i] = v;
        const float denom = __builtin_sqrtf(v) / bc2_sqrt + a.eps;
        a.p[i] = p - step_size * (m / denom);
    }
}

}  // namespace

extern "C" int bgk_grad_nan_flag(const float* g, int64_t n, int32_t* flag, void* stream) {
    BGK_CHECK_ARG(g && flag && n >= 0, "bgk_grad_nan_flag: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int32_t), st);
    if (e != hipSuccess) { bgk_set_error("bgk_grad_nan_flag: %s", hipGetErrorString(e)); return (int)e; }
    if (n == 0) return 0;
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(nan_flag_kernel, dim3((unsigned)(nb < 2048 ? nb : 2048)), dim3(256), 0, st, g, n, flag);
    return bgk_launch_status("bgk_grad_nan_flag");
}

extern "C" int bgk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, const int32_t* skip_flag, int32_t* skipped_count, void* stream) {
    BGK_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "bgk_adam_step: bad arguments");
    if (n == 0) return 0;
    AdamArgs a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, skip_flag, skipped_count};
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_adam_step");
}
