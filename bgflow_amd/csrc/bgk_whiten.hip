/* bgk_whiten.hip -- static PCA whitening / blackening of a coordinate block on its own:
 *   WhitenFlow._whiten:  z = (x - mean) Twhiten      WhitenFlow._blacken:  x = z Tblacken + mean      (nn/flow/pca.py:74-93)
 * i.e. out = (x - pre) T + post with a small constant matrix T [n_in, n_out] (<= 128 x 128: Cartesian blocks of a few atoms; inside
 * MixedCoordinateTransformation the same product is fused into the IC kernels).  The log-det is the constant -+ sum log std, formed by
 * the host.  Also the VJP: g_x = g_out T^T is the same kernel on the transposed matrix.
 * Roofline: HBM, 4 (n_in + n_out) B per sample.  A wave owns 64 samples: the [64][n_in] tile arrives as a linear copy (rows are
 * contiguous), T sits in LDS once per workgroup; lane = sample walks its row (LDS row stride n_in | 1: conflict free) against
 * broadcast reads of T, four output columns at a time; the finished rows leave through the tile's LDS image as 16-byte pieces.
 */
#include "bgk_common.h"

namespace {

constexpr int WH_W = 4;          /* waves per workgroup (fewer when the tiles of four waves do not fit beside the matrix) */

struct WhArgs {
    const float* x; int64_t ldx; const float* T; const float* pre; const float* post;
    float* out; int64_t ldo; int64_t B; int n_in, n_out; int xs, os;   /* xs / os: LDS row strides of the input / output tile */
};

__global__ __launch_bounds__(WH_W * 64) void whiten_kernel(WhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_T = smem;                                        /* [n_in][n_out] */
    float* s_pre = s_T + a.n_in * a.n_out;                    /* [n_in] */
    float* s_post = s_pre + a.n_in;                           /* [n_out] */
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = (int)blockDim.x >> 6, nthr = (int)blockDim.x;
    float* s_x = s_post + a.n_out + wave * (64 * a.xs + 64 * a.os);
    float* s_o = s_x + 64 * a.xs;
    for (int i = tid; i < a.n_in * a.n_out; i += nthr) s_T[i] = a.T[i];
    for (int i = tid; i < a.n_in; i += nthr) s_pre[i] = a.pre ? a.pre[i] : 0.0f;
    for (int i = tid; i < a.n_out; i += nthr) s_post[i] = a.post ? a.post[i] : 0.0f;
    __syncthreads();
    const int64_t n_tiles = (a.B + 63) >> 6;
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        const int64_t b0 = tile << 6;
        const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
        /* coalesced along the rows: element i of the [rows][n_in] tile */
        for (int i = lane; i < rows * a.n_in; i += 64) {
            const int r = i / a.n_in, c = i - r * a.n_in;
            s_x[r * a.xs + c] = a.x[(b0 + r) * a.ldx + c] - s_pre[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < rows) {
            const float* xr = s_x + lane * a.xs;
            float* orow = s_o + lane * a.os;
            for (int j0 = 0; j0 < a.n_out; j0 += 4) {
                float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const int nj = a.n_out - j0 < 4 ? a.n_out - j0 : 4;
                for (int k = 0; k < a.n_in; ++k) {
                    const float xv = xr[k];
                    const float* t = s_T + k * a.n_out + j0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (u < nj) acc[u] = __builtin_fmaf(xv, t[u], acc[u]);
                }
                for (int u = 0; u < nj; ++u) orow[j0 + u] = acc[u] + s_post[j0 + u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < rows * a.n_out; i += 64) {
            const int r = i / a.n_out, c = i - r * a.n_out;
            a.out[(b0 + r) * a.ldo + c] = s_o[r * a.os + c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int bgk_whiten(const float* x, int64_t ldx, const float* T, const float* pre, const float* post,
                          int32_t n_in, int32_t n_out, int64_t B, float* out, int64_t ldo, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(x && T && out && B > 0 && n_in > 0 && n_out > 0 && ldx >= n_in && ldo >= n_out, "bgk_whiten: bad arguments");
    WhArgs a{x, ldx, T, pre, post, out, ldo, B, n_in, n_out, n_in | 1, n_out | 1};
    if (n_in > 128 || n_out > 128) {
        bgk_set_error("bgk_whiten: %d x %d is beyond the kernel's LDS tiles (<= 128 x 128)", n_in, n_out);
        return BGK_EUNSUPPORTED;
    }
    const size_t fixed = sizeof(float) * ((size_t)n_in * n_out + n_in + n_out), per_wave = sizeof(float) * 64 * (size_t)(a.xs + a.os);
    int nw = WH_W;
    while (nw > 1 && fixed + nw * per_wave > 160 * 1024) --nw;
    const size_t shmem = fixed + nw * per_wave;       /* 128 x 128: 64 KB + one wave's 66 KB */
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(whiten_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int64_t n_wg = (((B + 63) >> 6) + nw - 1) / nw;
    const int grid = (int)(n_wg < 256 * 4 ? n_wg : 256 * 4);
    hipLaunchKernelGGL(whiten_kernel, dim3(grid), dim3(nw * 64), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_whiten");
}
