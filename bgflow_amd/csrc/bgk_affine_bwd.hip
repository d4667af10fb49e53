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

/* Without volume preservation an element's gradients need nothing of its row but g_dlogp: ONE pass, no LDS, no second tanh -- a thread
 * walks float2 pieces (rows of even length at 8-byte aligned addresses) with running (row, column) counters instead of a division per
 * element.  Same arithmetic per element as the kernel above (bit-identical gradients); 0.31 -> 0.2x ms per layer of cfg 2 at 2^20. */
template <bool HAS_MU, bool HAS_S>
__global__ __launch_bounds__(AB_THREADS) void affine_bwd_pairs_kernel(AffBwdArgs a) {
    __shared__ float s_red[AB_THREADS / 64];
    const int tid = threadIdx.x, hd = a.d >> 1;
    const float alpha = HAS_S ? bgk_expf(a.log_alpha[0]) : 0.0f;
    float g_alpha = 0.0f, max_mu = 0.0f, max_s = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * AB_THREADS, i0 = (int64_t)blockIdx.x * AB_THREADS + tid;
    int64_t r = i0 / hd;
    int c = (int)(i0 - r * hd);
    const int64_t dr = stride / hd;
    const int dc = (int)(stride - dr * hd);
    for (; r < a.B;) {
        const float2 v = *reinterpret_cast<const float2*>(a.y + r * a.ldy + 2 * c);
        const float2 go = *reinterpret_cast<const float2*>(a.g_out + r * a.ldgo + 2 * c);
        const float gl = a.g_dlogp[r];
        float2 m = make_float2(0.0f, 0.0f), sr = make_float2(0.0f, 0.0f);
        if (HAS_MU) m = *reinterpret_cast<const float2*>(a.mu + r * a.ldmu + 2 * c);
        if (HAS_S) sr = *reinterpret_cast<const float2*>(a.s_raw + r * a.lds + 2 * c);
        float gy[2], gm[2], gs[2];
        const float vv[2] = {v.x, v.y}, gg[2] = {go.x, go.y}, mm[2] = {m.x, m.y}, ss[2] = {sr.x, sr.y};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float th = HAS_S ? bgk_tanhf(ss[e]) : 0.0f;
            const float ls = th * alpha;
            float gls;
            if (!a.inverse) { const float ex = bgk_expf(ls); gy[e] = gg[e] * ex; gm[e] = gg[e]; gls = gg[e] * ex * vv[e] + gl; }
            else { const float ex = bgk_expf(-ls); gy[e] = gg[e] * ex; gm[e] = -gg[e] * ex; gls = -gg[e] * ex * (vv[e] - mm[e]) - gl; }
            gs[e] = gls * alpha * (1.0f - th * th);
            g_alpha += gls * th;
            max_mu = __builtin_fmaxf(max_mu, __builtin_fabsf(gm[e]));
            max_s = __builtin_fmaxf(max_s, __builtin_fabsf(gs[e]));
        }
        *reinterpret_cast<float2*>(a.g_y + r * a.ldgy + 2 * c) = make_float2(gy[0], gy[1]);
        if (HAS_MU && a.g_mu) *reinterpret_cast<float2*>(a.g_mu + r * a.ldgmu + 2 * c) = make_float2(gm[0], gm[1]);
        if (HAS_S && a.g_s) *reinterpret_cast<float2*>(a.g_s + r * a.ldgs + 2 * c) = make_float2(gs[0], gs[1]);
        c += dc; r += dr;
        if (c >= hd) { c -= hd; ++r; }
    }
    if (a.mu_absmax || a.s_absmax) {
        for (int off = 32; off > 0; off >>= 1) { max_mu = __builtin_fmaxf(max_mu, __shfl_xor(max_mu, off)); max_s = __builtin_fmaxf(max_s, __shfl_xor(max_s, off)); }
        if ((tid & 63) == 0) {
            const unsigned bm = __builtin_bit_cast(unsigned, max_mu), bs = __builtin_bit_cast(unsigned, max_s);
            if (HAS_MU && a.mu_absmax && a.g_mu && bm > *reinterpret_cast<volatile unsigned*>(a.mu_absmax)) atomicMax(reinterpret_cast<unsigned*>(a.mu_absmax), bm);
            if (HAS_S && a.s_absmax && a.g_s && bs > *reinterpret_cast<volatile unsigned*>(a.s_absmax)) atomicMax(reinterpret_cast<unsigned*>(a.s_absmax), bs);
        }
    }
    if (HAS_S && a.g_log_alpha) {
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
    /* rows of even length, every row an 8-byte aligned address, no volume preservation: the one-pass float2 kernel */
    const auto even = [](const void* p, int64_t ld) { return p == nullptr || (ld % 2 == 0 && ((uintptr_t)p & 7) == 0); };
    if (!(preserve_volume && s_raw) && d % 2 == 0 && even(y, ldy) && even(mu, ldmu) && even(s_raw, lds) && even(g_out, ldgo) && even(g_y, ldgy)
        && even(g_mu, ldgmu) && even(g_s, ldgs)) {
        const int64_t pairs = B * (d / 2);
        int64_t want = (pairs + AB_THREADS - 1) / AB_THREADS;
        const int grid = (int)(want < 256 * 16 ? want : 256 * 16);
        hipStream_t st = (hipStream_t)stream;
        if (mu && s_raw) hipLaunchKernelGGL((affine_bwd_pairs_kernel<true, true>), dim3(grid), dim3(AB_THREADS), 0, st, a);
        else if (mu) hipLaunchKernelGGL((affine_bwd_pairs_kernel<true, false>), dim3(grid), dim3(AB_THREADS), 0, st, a);
        else if (s_raw) hipLaunchKernelGGL((affine_bwd_pairs_kernel<false, true>), dim3(grid), dim3(AB_THREADS), 0, st, a);
        else hipLaunchKernelGGL((affine_bwd_pairs_kernel<false, false>), dim3(grid), dim3(AB_THREADS), 0, st, a);
        return bgk_launch_status("bgk_affine_backward");
    }
    int TS = 4096 / d;
    TS = TS < 1 ? 1 : (TS > 256 ? 256 : TS);
    a.TS = TS;
    size_t shmem = sizeof(float) * ((size_t)TS * d + TS);
    int64_t n_tiles = (B + TS - 1) / TS;
    int grid = (int)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
    hipLaunchKernelGGL(affine_bwd_kernel, dim3(grid), dim3(AB_THREADS), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_affine_backward");
}
