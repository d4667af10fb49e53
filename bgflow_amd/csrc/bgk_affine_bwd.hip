/* bgk_affine_bwd.hip -- analytic backward (VJP) of the affine transformer tail
 * (nn/flow/transformer/affine.py:41-70 differentiated; same math as oracle bgo_affine_backward).
 * HBM-bound: reads y, mu, s_raw, g_out (+ g_dlogp), writes g_y, g_mu, g_s: 4*(7d + 1) B / sample.
 * g_log_alpha (a scalar parameter) is reduced wave -> block -> one atomicAdd per workgroup.
 */
#include "bgk_common.h"

namespace {

constexpr int AB_THREADS = 256;

struct AffBwdArgs {
    const float* y; int64_t ldy;
    const float* mu; int64_t ldmu;
    const float* s_raw; int64_t lds;
    const float* log_alpha;
    int preserve_volume, inverse;
    int64_t B; int d;
    const float* g_out; int64_t ldgo;
    const float* g_dlogp;
    float* g_y; int64_t ldgy;
    float* g_mu; int64_t ldgmu;
    float* g_s; int64_t ldgs;
    float* g_log_alpha;
    int TS;
    float* mu_absmax; float* s_absmax;   /* NULL, or [1] each: raised to max |g_mu| / max |g_s| (the scale source of the backward GEMMs that consume them) */
};

__global__ __launch_bounds__(AB_THREADS) void affine_bwd_kernel(AffBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = a.TS, d = a.d, tid = threadIdx.x;
    float* s_ls = smem;                 /* [TS][d] log_sigma, then g_ls */
    float* s_row = smem + TS * d;       /* [TS] per-row mean */
    __shared__ float s_red[AB_THREADS / 64];
    const float alpha = a.s_raw ? bgk_expf(a.log_alpha[0]) : 0.0f;
    const bool pv = a.preserve_volume && a.s_raw;
    float g_alpha = 0.0f, max_mu = 0.0f, max_s = 0.0f;
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        const int n = rows * d;
        if (pv) {
            for (int i = tid; i < n; i += AB_THREADS) {
                int r = i / d, j = i - r * d;
                s_ls[i] = bgk_tanhf(a.s_raw[(b0 + r) * a.lds + j]) * alpha;
            }
            __syncthreads();
            for (int r = tid; r < rows; r += AB_THREADS) {
                float s = 0.0f;
                for (int j = 0; j < d; ++j) s += s_ls[r * d + j];
                s_row[r] = s / (float)d;
            }
            __syncthreads();
        }
        /* g_ls per element (+ g_y, g_mu) */
        for (int i = tid; i < n; i += AB_THREADS) {
            int r = i / d, j = i - r * d;
            float th = a.s_raw ? bgk_tanhf(a.s_raw[(b0 + r) * a.lds + j]) : 0.0f;
            float ls = a.s_raw ? th * alpha - (pv ? s_row[r] : 0.0f) : 0.0f;
            float m = a.mu ? a.mu[(b0 + r) * a.ldmu + j] : 0.0f;
            float v = a.y[(b0 + r) * a.ldy + j], go = a.g_out[(b0 + r) * a.ldgo + j], gl = a.g_dlogp[b0 + r];
            float gy, gm, gls;
            if (!a.inverse) { float e = bgk_expf(ls); gy = go * e; gm = go; gls = go * e * v + gl; }
            else { float e = bgk_expf(-ls); gy = go * e; gm = -go * e; gls = -go * e * (v - m) - gl; }
            a.g_y[(b0 + r) * a.ldgy + j] = gy;
            if (a.g_mu) a.g_mu[(b0 + r) * a.ldgmu + j] = gm;
            max_mu = __builtin_fmaxf(max_mu, __builtin_fabsf(gm));
            s_ls[i] = gls;
        }
        __syncthreads();
        if (pv) {
            for (int r = tid; r < rows; r += AB_THREADS) {
                float s = 0.0f;
                for (int j = 0; j < d; ++j) s += s_ls[r * d + j];
                s_row[r] = s / (float)d;
            }
            __syncthreads();
        }
        if (a.s_raw) {
            for (int i = tid; i < n; i += AB_THREADS) {
                int r = i / d, j = i - r * d;
                float th = bgk_tanhf(a.s_raw[(b0 + r) * a.lds + j]);
                float g = s_ls[i] - (pv ? s_row[r] : 0.0f);
                const float gs = g * alpha * (1.0f - th * th);
                if (a.g_s) a.g_s[(b0 + r) * a.ldgs + j] = gs;
                max_s = __builtin_fmaxf(max_s, __builtin_fabsf(gs));
                g_alpha += g * th;
            }
        }
        __syncthreads();
    }
    if (a.mu_absmax || a.s_absmax) {        /* (non-negative floats order like their bit patterns; NaNs do not take part) */
        for (int off = 32; off > 0; off >>= 1) { max_mu = __builtin_fmaxf(max_mu, __shfl_xor(max_mu, off)); max_s = __builtin_fmaxf(max_s, __shfl_xor(max_s, off)); }
        if ((tid & 63) == 0) {
            const unsigned bm = __builtin_bit_cast(unsigned, max_mu), bs = __builtin_bit_cast(unsigned, max_s);
            if (a.mu_absmax && a.g_mu && bm > *reinterpret_cast<volatile unsigned*>(a.mu_absmax)) atomicMax(reinterpret_cast<unsigned*>(a.mu_absmax), bm);
            if (a.s_absmax && a.g_s && bs > *reinterpret_cast<volatile unsigned*>(a.s_absmax)) atomicMax(reinterpret_cast<unsigned*>(a.s_absmax), bs);
        }
    }
    if (a.g_log_alpha && a.s_raw) {
        for (int off = 32; off > 0; off >>= 1) g_alpha += __shfl_xor(g_alpha, off);
        if ((tid & 63) == 0) s_red[tid >> 6] = g_alpha;
        __syncthreads();
        if (tid == 0) {
            float s = 0.0f;
            for (int w = 0; w < AB_THREADS / 64; ++w) s += s_red[w];
            atomicAdd(a.g_log_alpha, s * alpha);
        }
    }
}

}  // namespace

extern "C" int bgk_affine_backward(const float* y, int64_t ldy, const float* mu, int64_t ldmu,
                                   const float* s_raw, int64_t lds, const float* log_alpha,
                                   int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                                   int64_t B, int32_t d, const float* g_out, int64_t ldgo,
                                   const float* g_dlogp, float* g_y, int64_t ldgy, float* g_mu,
                                   int64_t ldgmu, float* g_s, int64_t ldgs, float* g_log_alpha,
                                   float* g_mu_absmax, float* g_s_absmax, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    (void)is_circular;   /* d(o mod 1)/do = 1 */
    BGK_CHECK_ARG(B >= 0 && d > 0 && d <= 8192, "bgk_affine_backward: bad sizes B=%lld d=%d", (long long)B, d);
    BGK_CHECK_ARG(y && g_out && g_dlogp && g_y, "bgk_affine_backward: null pointer");
    BGK_CHECK_ARG(!(s_raw && !log_alpha), "bgk_affine_backward: s_raw given without log_alpha");
    if (B == 0) return 0;
    AffBwdArgs a{y, ldy, mu, ldmu, s_raw, lds, log_alpha, preserve_volume, inverse, B, d, g_out, ldgo, g_dlogp,
                 g_y, ldgy, g_mu, ldgmu, g_s, ldgs, g_log_alpha, 0, g_mu_absmax, g_s_absmax};
    int TS = 4096 / d;
    TS = TS < 1 ? 1 : (TS > 256 ? 256 : TS);
    a.TS = TS;
    size_t shmem = sizeof(float) * ((size_t)TS * d + TS);
    int64_t n_tiles = (B + TS - 1) / TS;
    int grid = (int)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
    hipLaunchKernelGGL(affine_bwd_kernel, dim3(grid), dim3(AB_THREADS), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_affine_backward");
}
