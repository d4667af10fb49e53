/* bgk_cdf.hip -- domain-mapping layers CDFTransform._forward/_inverse (nn/flow/cdf.py:28-46) over the
 * marginals the builder installs (factory/icmarginals.py:41-77): TruncatedNormalDistribution
 * (distribution/normal.py:215-227), torch.distributions.Normal and (Sloppy)Uniform.
 * One launch replaces ~10 elementwise aten kernels + a row reduction: 4*(2d + 2) algorithmic bytes
 * per sample, HBM-bound.  erf / erfinv are the OCML functions torch's ROCm kernels call as well.
 *
 * per-column descriptor (6 floats): kind, p0..p4
 *   kind 0 uniform : p0 = low, p1 = high, p2 = tol
 *   kind 1 normal  : p0 = loc, p1 = scale
 *   kind 2 truncated normal : p0 = mu, p1 = sigma, p2 = cdf_lower_bound, p3 = Z = cdf_upper - cdf_lower
 */
#include "bgk_common.h"

namespace {

constexpr int CDF_THREADS = 256;
#define SQRT2_F 1.41421356237309504880f
#define LOG_SQRT_2PI_F 0.91893853320467274178f

struct CdfArgs {
    const float* x; int64_t ldx;
    const float* desc;      /* [d][6] */
    int64_t B; int d; int inverse; int use_eps; float eps;
    float* out; int64_t ldo;
    float* dlogp; int accumulate;
    int TS;
};

__device__ __forceinline__ float std_logp(float z) { return -(z * z) / 2.0f - LOG_SQRT_2PI_F; }

__global__ __launch_bounds__(CDF_THREADS) void cdf_kernel(CdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = a.TS, d = a.d, tid = threadIdx.x;
    float* s_ld = smem;   /* [TS][d] per-element log-dets */
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        for (int i = tid; i < rows * d; i += CDF_THREADS) {
            const int r = i / d, j = i - r * d;
            const float* ds = a.desc + 6 * j;
            const int kind = (int)ds[0];
            float v = a.x[(b0 + r) * a.ldx + j];
            float y, ld;
            if (a.inverse) {                 /* x in [0,1] -> icdf; logdet = -log_prob(y) */
                if (a.use_eps) v = v < a.eps ? a.eps : (v > 1.0f - a.eps ? 1.0f - a.eps : v);
                if (kind == 0) {
                    y = ds[1] + v * (ds[2] - ds[1]);
                    ld = logf(ds[2] - ds[1]);
                } else if (kind == 1) {
                    y = ds[1] + ds[2] * erfinvf(2.0f * v - 1.0f) * SQRT2_F;
                    const float dv = y - ds[1];
                    ld = -(-(dv * dv) / (2.0f * (ds[2] * ds[2])) - logf(ds[2]) - LOG_SQRT_2PI_F);
                } else {
                    const float r0 = ds[4] * v + ds[3];
                    y = (erfinvf(2.0f * r0 - 1.0f) * SQRT2_F) * ds[2] + ds[1];
                    ld = -(std_logp((y - ds[1]) / ds[2]) - logf(ds[4] * ds[2]));
                }
            } else {                         /* x -> cdf(x); logdet = log_prob(x) */
                if (kind == 0) {
                    y = (v - ds[1]) / (ds[2] - ds[1]);
                    y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
                    const bool inside = (v >= ds[1] - ds[3]) & (v <= ds[2] + ds[3]);
                    ld = inside ? -logf(ds[2] - ds[1]) : -__builtin_inff();
                } else if (kind == 1) {
                    y = 0.5f * (1.0f + erff((v - ds[1]) * (1.0f / ds[2]) / SQRT2_F));
                    const float dv = v - ds[1];
                    ld = -(dv * dv) / (2.0f * (ds[2] * ds[2])) - logf(ds[2]) - LOG_SQRT_2PI_F;
                } else {
                    const float z = (v - ds[1]) / ds[2];
                    y = (0.5f * (1.0f + erff(z / SQRT2_F)) - ds[3]) / ds[4];
                    ld = std_logp(z) - logf(ds[4] * ds[2]);
                }
                if (a.use_eps) y = y < a.eps ? a.eps : (y > 1.0f - a.eps ? 1.0f - a.eps : y);
            }
            if (a.use_eps) ld = ld < -1.0f / a.eps ? -1.0f / a.eps : ld;
            a.out[(b0 + r) * a.ldo + j] = y;
            s_ld[i] = ld;
        }
        __syncthreads();
        for (int r = tid; r < rows; r += CDF_THREADS) {
            float acc = 0.0f;
            for (int j = 0; j < d; ++j) acc += s_ld[r * d + j];
            if (a.accumulate) a.dlogp[b0 + r] += acc; else a.dlogp[b0 + r] = acc;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int bgk_cdf_transform(const float* x, int64_t ldx, const float* desc, int64_t B, int32_t d,
                                 int32_t inverse, int32_t use_eps, float eps, float* out, int64_t ldo,
                                 float* dlogp, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && d > 0 && d <= 8192 && x && desc && out && dlogp, "bgk_cdf_transform: bad arguments");
    if (B == 0) return 0;
    CdfArgs a{x, ldx, desc, B, d, inverse, use_eps, eps, out, ldo, dlogp, accumulate, 0};
    int TS = 4096 / d;
    TS = TS < 1 ? 1 : (TS > 256 ? 256 : TS);
    a.TS = TS;
    size_t shmem = sizeof(float) * (size_t)TS * d;
    int64_t n_tiles = (B + TS - 1) / TS;
    int grid = (int)(n_tiles < 256 * 16 ? n_tiles : 256 * 16);
    hipLaunchKernelGGL(cdf_kernel, dim3(grid), dim3(CDF_THREADS), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_cdf_transform");
}

/* ---- backward (VJP) of cdf_kernel: g_x = g_y dy/dx + g_dlogp d logdet/dx, elementwise --------------------------------
 * With y saved from the forward pass no erf / erfinv is needed: dy/dx = exp(logdet element) in both directions (the
 * density resp. its reciprocal), d logdet/dx = -z/sigma (cdf direction) or (z/sigma) dy/dx (icdf direction); clamped
 * values (eps) pass no gradient, like torch.clamp in the reference (nn/flow/cdf.py:31-45). */
namespace {
struct CdfBwdArgs {
    const float* x; int64_t ldx; const float* y; int64_t ldy; const float* desc;
    int64_t B; int d; int inverse; int use_eps; float eps;
    const float* g_y; int64_t ldgy; const float* g_dlogp; float* g_x; int64_t ldgx;
};

__global__ __launch_bounds__(CDF_THREADS) void cdf_bwd_kernel(CdfBwdArgs a) {
    const int64_t total = a.B * a.d;
    for (int64_t i = (int64_t)blockIdx.x * CDF_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * CDF_THREADS) {
        const int64_t r = i / a.d;
        const int j = (int)(i - r * a.d);
        const float* ds = a.desc + 6 * j;
        const int kind = (int)ds[0];
        const float x = a.x[r * a.ldx + j], y = a.y[r * a.ldy + j];
        const float gy = a.g_y[r * a.ldgy + j], gl = a.g_dlogp[r];
        float dy, dld, ld;
        bool pass;
        if (a.inverse) {                     /* x = u in [0,1], y = icdf(u), logdet = -log_prob(y) */
            pass = !a.use_eps || (x >= a.eps && x <= 1.0f - a.eps);
            if (kind == 0) { dy = ds[2] - ds[1]; dld = 0.0f; ld = 0.0f; }
            else {
                const float sig = kind == 1 ? ds[2] : ds[2], mu = ds[1];
                const float z = (y - mu) / sig;
                ld = 0.5f * z * z + LOG_SQRT_2PI_F + (kind == 1 ? logf(sig) : logf(ds[4] * sig));
                dy = expf(ld);
                dld = (z / sig) * dy;
            }
        } else {                             /* y = cdf(x), logdet = log_prob(x) */
            pass = !a.use_eps || (y > a.eps && y < 1.0f - a.eps);
            if (kind == 0) {
                const float u = (x - ds[1]) / (ds[2] - ds[1]);
                dy = (u >= 0.0f && u <= 1.0f) ? 1.0f / (ds[2] - ds[1]) : 0.0f;
                dld = 0.0f; ld = 0.0f;
            } else {
                const float sig = ds[2], mu = ds[1];
                const float z = (x - mu) / sig;
                ld = -0.5f * z * z - LOG_SQRT_2PI_F - (kind == 1 ? logf(sig) : logf(ds[4] * sig));
                dy = expf(ld);
                dld = -z / sig;
            }
        }
        if (a.use_eps && ld < -1.0f / a.eps) dld = 0.0f;
        const float g = (pass ? gy * dy : 0.0f) + gl * dld * ((a.inverse && !pass) ? 0.0f : 1.0f);
        a.g_x[r * a.ldgx + j] = g;
    }
}
}  // namespace

extern "C" int bgk_cdf_backward(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* desc, int64_t B, int32_t d,
                                int32_t inverse, int32_t use_eps, float eps, const float* g_y, int64_t ldgy,
                                const float* g_dlogp, float* g_x, int64_t ldgx, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && d > 0 && x && y && desc && g_y && g_dlogp && g_x, "bgk_cdf_backward: bad arguments");
    if (B == 0) return 0;
    CdfBwdArgs a{x, ldx, y, ldy, desc, B, d, inverse, use_eps, eps, g_y, ldgy, g_dlogp, g_x, ldgx};
    const int64_t nb = (B * d + CDF_THREADS - 1) / CDF_THREADS;
    const int grid = (int)(nb < 256 * 32 ? nb : 256 * 32);
    hipLaunchKernelGGL(cdf_bwd_kernel, dim3(grid), dim3(CDF_THREADS), 0, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_cdf_backward");
}
