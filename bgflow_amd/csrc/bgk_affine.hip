/* bgk_affine.hip -- affine (RealNVP / NICE) transformer tail, conditioner outputs read from HBM.
 *
 * HBM-bound: 4*(4d + 2) algorithmic bytes per sample (mu, s_raw, y in; y out; dlogp r/w).
 * A 256-thread workgroup owns TS consecutive rows; since y / mu / s_raw / out are row-contiguous
 * the tile is one flat coalesced stream.  log_sigma goes through LDS so that the per-row sums
 * (mean for preserve_volume, dlogp) are taken in ascending-dim order by one lane per row:
 * deterministic and bit-identical with oracle/bgo_impl.h::bgo_affine.
 */
#include "bgk_common.h"

namespace {

constexpr int AFF_THREADS = 256;

struct AffArgs {
    const float* y; int64_t ldy;
    const float* mu; int64_t ldmu;
    const float* s_raw; int64_t lds;
    const float* log_alpha;
    int preserve_volume, is_circular, inverse;
    int64_t B; int d;
    float* out; int64_t ldo;
    float* dlogp; int accumulate;
    int TS;
};

__global__ __launch_bounds__(AFF_THREADS) void affine_kernel(AffArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = a.TS, d = a.d, tid = threadIdx.x;
    float* s_ls = smem;            /* [TS][d] log_sigma */
    float* s_mean = smem + TS * d; /* [TS] */
    const float alpha = a.s_raw ? bgk_expf(a.log_alpha[0]) : 0.0f;
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        const int n = rows * d;
        for (int i = tid; i < n; i += AFF_THREADS) {
            int r = i / d, j = i - r * d;
            float ls = 0.0f;
            if (a.s_raw) ls = bgk_tanhf(a.s_raw[(b0 + r) * a.lds + j]) * alpha;
            s_ls[i] = ls;
        }
        __syncthreads();
        if (a.preserve_volume && a.s_raw) {
            for (int r = tid; r < rows; r += AFF_THREADS) {
                float s = 0.0f;
                for (int j = 0; j < d; ++j) s += s_ls[r * d + j];
                s_mean[r] = s / (float)d;
            }
            __syncthreads();
        }
        for (int i = tid; i < n; i += AFF_THREADS) {
            int r = i / d, j = i - r * d;
            float ls = s_ls[i];
            if (a.preserve_volume && a.s_raw) { ls = ls - s_mean[r]; s_ls[i] = ls; }
            float m = a.mu ? a.mu[(b0 + r) * a.ldmu + j] : 0.0f;
            float v = a.y[(b0 + r) * a.ldy + j];
            float o = a.inverse ? bgk_expf(-ls) * (v - m) : bgk_expf(ls) * v + m;
            if (a.is_circular) { o = o - __builtin_truncf(o); if (o < 0.0f) o = o + 1.0f; }
            a.out[(b0 + r) * a.ldo + j] = o;
        }
        __syncthreads();
        for (int r = tid; r < rows; r += AFF_THREADS) {
            float acc = 0.0f;
            if (a.inverse) { for (int j = 0; j < d; ++j) acc += -s_ls[r * d + j]; }
            else { for (int j = 0; j < d; ++j) acc += s_ls[r * d + j]; }
            if (a.accumulate) a.dlogp[b0 + r] += acc; else a.dlogp[b0 + r] = acc;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int bgk_affine_transform(const float* y, int64_t ldy, const float* mu, int64_t ldmu,
                                    const float* s_raw, int64_t lds, const float* log_alpha,
                                    int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                                    int64_t B, int32_t d, float* out, int64_t ldo, float* dlogp,
                                    int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && d > 0, "bgk_affine_transform: bad sizes B=%lld d=%d", (long long)B, d);
    BGK_CHECK_ARG(y && out && dlogp, "bgk_affine_transform: null pointer");
    BGK_CHECK_ARG(!(s_raw && !log_alpha), "bgk_affine_transform: s_raw given without log_alpha");
    BGK_CHECK_ARG(!(s_raw && is_circular), "Scaling is not compatible with periodicity.");
    BGK_CHECK_ARG(d <= 8192, "bgk_affine_transform: d=%d too large for the LDS tile", d);
    if (B == 0) return 0;
    AffArgs a{y, ldy, mu, ldmu, s_raw, lds, log_alpha, preserve_volume, is_circular, inverse,
              B, d, out, ldo, dlogp, accumulate, 0};
    int TS = 4096 / d;               /* ~16 elements per thread per tile */
    TS = TS < 1 ? 1 : (TS > 256 ? 256 : TS);
    a.TS = TS;
    size_t shmem = sizeof(float) * ((size_t)TS * d + TS);
    int64_t n_tiles = (B + TS - 1) / TS;
    int grid = (int)(n_tiles < 256 * 16 ? n_tiles : 256 * 16);
    hipLaunchKernelGGL(affine_kernel, dim3(grid), dim3(AFF_THREADS), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_affine_transform");
}
