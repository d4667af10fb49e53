/* bgk_cdf.hip -- domain-mapping layers CDFTransform._forward/_inverse (nn/flow/cdf.py:28-46) over the
 * marginals the builder installs (factory/icmarginals.py:41-77): TruncatedNormalDistribution
 * (distribution/normal.py:215-227), torch.distributions.Normal and (Sloppy)Uniform.
 * One launch replaces ~10 elementwise aten kernels + a row reduction: 4*(2d + 2) algorithmic bytes
 * per sample, HBM-bound.  erf / erfinv: the polynomial forms of bgk_erf.h (Juffa / Giles, < 1 / < 4 ulp; the OCML functions
 * made this a VALU-bound kernel: 0.30 ms for the 66-wide slot of cfg 5 at 2^20 samples); the per-column logarithms and
 * reciprocals are formed once per workgroup in LDS; row index by multiply-high instead of an integer division.
 *
 * per-column descriptor (6 floats): kind, p0..p4
 *   kind 0 uniform : p0 = low, p1 = high, p2 = tol
 *   kind 1 normal  : p0 = loc, p1 = scale
 *   kind 2 truncated normal : p0 = mu, p1 = sigma, p2 = cdf_lower_bound, p3 = Z = cdf_upper - cdf_lower
 */
#include "bgk_common.h"
#include "bgk_erf.h"

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
    int TS, col_lds;
};

__device__ __forceinline__ float std_logp(float z) { return -(z * z) / 2.0f - LOG_SQRT_2PI_F; }

__global__ __launch_bounds__(CDF_THREADS) void cdf_kernel(CdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = a.TS, d = a.d, tid = threadIdx.x;
    float4* s_col = reinterpret_cast<float4*>(smem);     /* [d][2]: {kind, p0, p1, p2}, {p3, log normaliser, 1 / width, -} */
    float* s_ld = smem + (a.col_lds ? 8 * d : 0);        /* [TS][d] per-element log-dets */
    for (int j = tid; a.col_lds && j < d; j += CDF_THREADS) {
        const float* ds = a.desc + 6 * j;
        const int kind = (int)ds[0];
        const float w = kind == 0 ? ds[2] - ds[1] : ds[2];     /* high - low | scale | sigma */
        s_col[2 * j] = make_float4(ds[0], ds[1], ds[2], ds[3]);
        s_col[2 * j + 1] = make_float4(ds[4], logf(kind == 2 ? ds[4] * ds[2] : w), 1.0f / w, 0.0f);
    }
    __syncthreads();
    const uint32_t magic = (uint32_t)(((1ull << 32) + (uint32_t)d - 1) / (uint32_t)d);      /* i / d for i < TS d <= 4096 + d (d > 1) */
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        for (int i = tid; i < rows * d; i += CDF_THREADS) {
            const int r = d == 1 ? i : (int)__umulhi((unsigned)i, magic), j = i - r * d;
            float4 c0, c1;
            if (a.col_lds) { c0 = s_col[2 * j]; c1 = s_col[2 * j + 1]; }
            else {                           /* very wide maps (d > 1800): the column constants do not fit LDS beside a tile */
                const float* ds = a.desc + 6 * j;
                const float w = (int)ds[0] == 0 ? ds[2] - ds[1] : ds[2];
                c0 = make_float4(ds[0], ds[1], ds[2], ds[3]);
                c1 = make_float4(ds[4], logf((int)ds[0] == 2 ? ds[4] * ds[2] : w), 1.0f / w, 0.0f);
            }
            const int kind = (int)c0.x;
            const float p0 = c0.y, p1 = c0.z, p2 = c0.w, p3 = c1.x, lg = c1.y, iv = c1.z;
            float v = a.x[(b0 + r) * a.ldx + j];
            float y, ld;
            if (a.inverse) {                 /* x in [0,1] -> icdf; logdet = -log_prob(y) */
                if (a.use_eps) v = v < a.eps ? a.eps : (v > 1.0f - a.eps ? 1.0f - a.eps : v);
                if (kind == 0) {
                    y = p0 + v * (p1 - p0);
                    ld = lg;
                } else if (kind == 1) {
                    const float z = erfinv_fast(2.0f * v - 1.0f) * SQRT2_F;
                    y = p0 + p1 * z;
                    const float zz = (y - p0) * iv;
                    ld = (zz * zz) * 0.5f + lg + LOG_SQRT_2PI_F;
                } else {
                    const float r0 = p3 * v + p2;
                    y = (erfinv_fast(2.0f * r0 - 1.0f) * SQRT2_F) * p1 + p0;
                    ld = -(std_logp((y - p0) * iv) - lg);
                }
            } else {                         /* x -> cdf(x); logdet = log_prob(x) */
                if (kind == 0) {
                    y = (v - p0) * iv;
                    y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
                    const bool inside = (v >= p0 - p2) & (v <= p1 + p2);
                    ld = inside ? -lg : -__builtin_inff();
                } else if (kind == 1) {
                    const float z = (v - p0) * iv;
                    y = 0.5f * (1.0f + erf_fast(z * (1.0f / SQRT2_F)));
                    ld = -(z * z) * 0.5f - lg - LOG_SQRT_2PI_F;
                } else {
                    const float z = (v - p0) * iv;
                    y = (0.5f * (1.0f + erf_fast(z * (1.0f / SQRT2_F))) - p2) / p3;
                    ld = std_logp(z) - lg;
                }
                if (a.use_eps) y = y < a.eps ? a.eps : (y > 1.0f - a.eps ? 1.0f - a.eps : y);
            }
            if (a.use_eps) ld = ld < -1.0f / a.eps ? -1.0f / a.eps : ld;
            a.out[(b0 + r) * a.ldo + j] = y;
            s_ld[i] = ld;
        }
        __syncthreads();
        /* row sums of the log-dets: four lanes per row (columns q, q + 4, ... in ascending order each), combined in a fixed order */
        for (int t0 = 0; t0 < 4 * rows; t0 += CDF_THREADS) {
            const int t = t0 + tid, r = t >> 2, q = t & 3;
            float acc = 0.0f;
            if (r < rows)
                for (int j = q; j < d; j += 4) acc += s_ld[r * d + j];
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (r < rows && q == 0) {
                if (a.accumulate) a.dlogp[b0 + r] += acc; else a.dlogp[b0 + r] = acc;
            }
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
    CdfArgs a{x, ldx, desc, B, d, inverse, use_eps, eps, out, ldo, dlogp, accumulate, 0, 0};
    int TS = 4096 / d;
    TS = TS < 1 ? 1 : (TS > 256 ? 256 : TS);
    a.TS = TS;
    a.col_lds = ((size_t)TS + 8) * d * sizeof(float) <= 60 * 1024;       /* the per-column constants beside the tile's log-dets */
    size_t shmem = sizeof(float) * ((size_t)TS * d + (a.col_lds ? 8 * (size_t)d : 0));
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
