/* bgk_rqs_bwd.hip -- analytic backward (VJP) of the rational-quadratic spline transformer for
 * first-order losses (KL / NLL): replaces torch autograd through the reference's op chain
 * (nn/flow/transformer/spline.py:109-188 + nflows).  Same math as oracle/bgo_impl.h::bgo_rqs_backward.
 *
 * HBM-bound: per sample it reads P + 2d + 1 floats (params, y, g_out, g_dlogp) and writes P + d
 * floats (g_params, g_y): 4*(2P + 3d + 1) algorithmic bytes.  Every (sample, dim) element owns its 3K(+1)
 * parameter slots exclusively, so an element's gradients are written straight from registers.
 */
#include "bgk_rqs_vjp.h"

namespace {

constexpr int BWD_THREADS = 256;

struct RqsBwdArgs {
    const float* y; int64_t ldy;
    const float* params; int64_t ldp;
    const int32_t* nc_slot;
    int64_t B; int d; int K; int P; int inverse;
    const float* g_out; int64_t ldgo;
    const float* g_dlogp;
    float* g_y; int64_t ldgy;
    float* g_params; int64_t ldgp;
    float* g_absmax;             /* [1] raised to the largest |g_params| written (NULL: not wanted): the power-of-two scale of the
                                  * backward GEMMs that consume g_params (bgk_dense_backward_dx, bgk_dense_weight_grad) */
    int TS, Pp;
    uint32_t magicP;
    BgkRqsCfg cfg;
};

/* Streaming form (same scheme as rqs_stream_kernel, bgk_rqs.hip): elements enumerated dim-fastest, the 24 parameters
 * of an element loaded with six 16-byte loads into registers, the 24 (+1) parameter gradients written back the same
 * way; no LDS, no barrier -- nothing in the backward pass crosses lanes. */
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int BWD_TS = 128;

/* PACKED: the parameters arrive element-major, [B][d][3 K + 1] (what the fused training forward writes since round 5: an element's
 * widths | heights | slopes | slot in one 100-byte run); the gradients leave in the reference's column order either way */
template <int KT, bool PACKED = false>
__global__ __launch_bounds__(BWD_THREADS) void rqs_bwd_kernel(RqsBwdArgs a) {
    constexpr int K = KT;
    static_assert(KT % 4 == 0, "KT / 4 16-byte loads per parameter group");
    const int d = a.d, tid = threadIdx.x;
    const BgkRqsCfg& c = a.cfg;
    const uint64_t magicd = (0x100000000ull + (uint64_t)d - 1) / (uint64_t)d;
    const int64_t n_tiles = (a.B + BWD_TS - 1) / BWD_TS;
    float gmax = 0.0f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * BWD_TS;
        const int rows = (int)((a.B - b0) < BWD_TS ? (a.B - b0) : BWD_TS);
        for (int e = tid; e < rows * d; e += BWD_THREADS) {
            const int s = (int)(((uint64_t)(uint32_t)e * magicd) >> 32), j = e - s * d;
            const float* row = a.params + (b0 + s) * a.ldp;
            float* grow = a.g_params + (b0 + s) * a.ldgp;
            const float* gw = PACKED ? row + j * (3 * K + 1) : row + j * K;
            const float* gh = PACKED ? gw + K : gw + d * K;
            const float* gs = PACKED ? gh + K : gh + d * K;
            float rw[K], rh[K], rs[K], ow[K], oh[K], os[K];
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                const f4u w4 = *reinterpret_cast<const f4u*>(gw + 4 * q), h4 = *reinterpret_cast<const f4u*>(gh + 4 * q);
                const f4u t4 = *reinterpret_cast<const f4u*>(gs + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) { rw[4 * q + k] = w4[k]; rh[4 * q + k] = h4[k]; rs[4 * q + k] = t4[k]; }
            }
            const int slot = a.nc_slot[j];
            const float s_K = slot >= 0 ? (PACKED ? gw[3 * K] : row[3 * d * K + slot]) : rs[0];   /* slope at knot K */
            const float x = a.y[(b0 + s) * a.ldy + j];
            const float gy = a.g_out[(b0 + s) * a.ldgo + j], gl = a.g_dlogp[b0 + s];
            float g_slot, gx;
            bgk_rqs_vjp_element<K>(c, a.inverse, rw, rh, rs, s_K, slot >= 0, x, gy, gl, ow, oh, os, g_slot, gx);
            a.g_y[(b0 + s) * a.ldgy + j] = gx;
            if (slot >= 0) grow[3 * d * K + slot] = g_slot;
            gmax = __builtin_fmaxf(gmax, __builtin_fabsf(g_slot));
#pragma unroll
            for (int k = 0; k < K; ++k) gmax = __builtin_fmaxf(gmax, __builtin_fmaxf(__builtin_fabsf(ow[k]), __builtin_fmaxf(__builtin_fabsf(oh[k]), __builtin_fabsf(os[k]))));
            {
                float* qw = grow + j * K;
                float* qh = qw + d * K;
                float* qs = qh + d * K;
#pragma unroll
                for (int q = 0; q < K / 4; ++q) {
                    *reinterpret_cast<f4u*>(qw + 4 * q) = (f4u){ow[4 * q], ow[4 * q + 1], ow[4 * q + 2], ow[4 * q + 3]};
                    *reinterpret_cast<f4u*>(qh + 4 * q) = (f4u){oh[4 * q], oh[4 * q + 1], oh[4 * q + 2], oh[4 * q + 3]};
                    *reinterpret_cast<f4u*>(qs + 4 * q) = (f4u){os[4 * q], os[4 * q + 1], os[4 * q + 2], os[4 * q + 3]};
                }
            }
        }
    }
    bgk_publish_absmax(a.g_absmax, gmax);
}

/* ---- any bin count: every lane walks the 3 K (+1) parameters of ITS element in memory (the backward twin of rqs_direct_kernel,
 * bgk_rqs.hip) and writes the 3 K (+1) gradients the same way; elements dim-fastest, so the K-float runs of 64 lanes are
 * consecutive in a row.  Four walks per set (maximum, sum of exps, knots, gradients): the lines stay in L1 / L2 meanwhile.  Keeps
 * the backward of bin counts without a register-resident instance off device torch ops; not a roofline kernel. ---- */
__global__ __launch_bounds__(BWD_THREADS) void rqs_bwd_direct_kernel(RqsBwdArgs a) {
    const int d = a.d, K = a.K, tid = threadIdx.x;
    const BgkRqsCfg& c = a.cfg;
    const bool comp = K >= BGK_RQS_COMP_FROM;
    const uint64_t magicd = (0x100000000ull + (uint64_t)d - 1) / (uint64_t)d;
    const int64_t n_tiles = (a.B + BWD_TS - 1) / BWD_TS;
    float gmax = 0.0f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * BWD_TS;
        const int rows = (int)((a.B - b0) < BWD_TS ? (a.B - b0) : BWD_TS);
        for (int e = tid; e < rows * d; e += BWD_THREADS) {
            const int s = (int)(((uint64_t)(uint32_t)e * magicd) >> 32), j = e - s * d;
            const float* row = a.params + (b0 + s) * a.ldp;
            float* grow = a.g_params + (b0 + s) * a.ldgp;
            const float* pw = row + (int64_t)j * K;
            const float* ph = pw + (int64_t)d * K;
            const float* ps = ph + (int64_t)d * K;
            const int slot = a.nc_slot[j];
            float x = a.y[(b0 + s) * a.ldy + j];
            const float gy = a.g_out[(b0 + s) * a.ldgo + j], gl = a.g_dlogp[b0 + s];
            const bool clamped = (x < c.left) | (x > c.right);
            x = x < c.left ? c.left : (x > c.right ? c.right : x);
            const BgkSoftmaxSet qw = bgk_softmax_set(pw, K, comp), qh = bgk_softmax_set(ph, K, comp);
            int idx;
            BgkKnotPair kw, kh;
            if (a.inverse) {
                kw = bgk_walk_search(qw, K, comp, c.min_w, c.w_scale, c.xspan, c.left, c.right, x, idx);
                kh = bgk_walk_to(qh, K, comp, c.min_h, c.h_scale, c.yspan, c.bottom, c.top, idx);
            } else {
                kh = bgk_walk_search(qh, K, comp, c.min_h, c.h_scale, c.yspan, c.bottom, c.top, x, idx);
                kw = bgk_walk_to(qw, K, comp, c.min_w, c.w_scale, c.xspan, c.left, c.right, idx);
            }
            const bool hi_last = idx + 1 == K, has_slot = slot >= 0;
            const float s_lo = ps[idx];
            const float s_hi = !hi_last ? ps[idx + 1] : (has_slot ? row[(int64_t)3 * d * K + slot] : ps[0]);
            const BgkVjpBin b = bgk_rqs_vjp_bin(c, a.inverse, x, kw.k_i, kw.k_n, kh.k_i, kh.k_n, s_lo, s_hi, gy, gl);
            a.g_y[(b0 + s) * a.ldgy + j] = clamped ? 0.0f : b.gx;
            float* qgw = grow + (int64_t)j * K;
            float* qgh = qgw + (int64_t)d * K;
            float* qgs = qgh + (int64_t)d * K;
            gmax = __builtin_fmaxf(gmax, bgk_walk_grad(qw, K, c.w_scale, c.xspan, idx, kw, b.G_cw, b.G_W, qgw));
            gmax = __builtin_fmaxf(gmax, bgk_walk_grad(qh, K, c.h_scale, c.yspan, idx, kh, b.G_ch, b.G_H, qgh));
            gmax = __builtin_fmaxf(gmax, __builtin_fmaxf(__builtin_fabsf(b.g0), __builtin_fabsf(b.g1)));
            for (int k = 0; k < K; ++k) {
                float g = 0.0f;
                g += (k == idx) ? b.g0 : 0.0f;
                g += (!hi_last && k == idx + 1) ? b.g1 : 0.0f;
                g += (hi_last && !has_slot && k == 0) ? b.g1 : 0.0f;
                qgs[k] = g;
            }
            if (has_slot) grow[(int64_t)3 * d * K + slot] = hi_last ? b.g1 : 0.0f;
        }
    }
    bgk_publish_absmax(a.g_absmax, gmax);
}

}  // namespace

extern "C" int bgk_rqs_backward(const float* y, int64_t ldy, const float* params, int64_t ldp, int32_t P,
                                const int32_t* nc_slot, int64_t B, int32_t d, int32_t K, int32_t inverse,
                                double left, double right, double bottom, double top,
                                double min_bin_width, double min_bin_height, double min_derivative,
                                int32_t identity_init, const float* g_out, int64_t ldgo,
                                const float* g_dlogp, float* g_y, int64_t ldgy, float* g_params,
                                int64_t ldgp, float* g_absmax, int32_t params_layout, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && d > 0 && K > 0, "bgk_rqs_backward: bad sizes");
    BGK_CHECK_ARG(y && params && nc_slot && g_out && g_dlogp && g_y && g_params, "bgk_rqs_backward: null pointer");
    BGK_CHECK_ARG(params_layout == 0 || (params_layout == 1 && K == 8), "bgk_rqs_backward: params_layout %d (1 = element-major: 8 bins only)", params_layout);
    BGK_CHECK_ARG(P >= 3 * K * d && P <= 3 * K * d + d && ldp >= (params_layout == 1 ? (3 * K + 1) * d : P) && ldgp >= P, "bgk_rqs_backward: bad params width %d", P);
    BGK_CHECK_ARG(min_bin_width * K <= 1.0 && min_bin_height * K <= 1.0, "Minimal bin width/height too large for the number of bins");
    RqsBwdArgs a;
    a.y = y; a.ldy = ldy; a.params = params; a.ldp = ldp; a.nc_slot = nc_slot; a.B = B; a.d = d; a.K = K; a.P = P;
    a.inverse = inverse; a.g_out = g_out; a.ldgo = ldgo; a.g_dlogp = g_dlogp; a.g_y = g_y; a.ldgy = ldgy;
    a.g_params = g_params; a.ldgp = ldgp; a.g_absmax = g_absmax;
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
    a.Pp = P | 1; a.TS = BWD_TS; a.magicP = 0;
    int64_t n_tiles = (B + BWD_TS - 1) / BWD_TS;
    int grid = (int)(n_tiles < 256 * 16 ? n_tiles : 256 * 16);
    const size_t shmem = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
        case 4: hipLaunchKernelGGL(rqs_bwd_kernel<4>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        case 8: if (params_layout == 1) hipLaunchKernelGGL((rqs_bwd_kernel<8, true>), dim3(grid), dim3(BWD_THREADS), shmem, st, a);
                else hipLaunchKernelGGL(rqs_bwd_kernel<8>, dim3(grid), dim3(BWD_THREADS), shmem, st, a);
                break;
        case 12: hipLaunchKernelGGL(rqs_bwd_kernel<12>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        case 16: hipLaunchKernelGGL(rqs_bwd_kernel<16>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        case 32: hipLaunchKernelGGL(rqs_bwd_kernel<32>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        /* any other bin count: the parameters stay in memory and are walked */
        default: hipLaunchKernelGGL(rqs_bwd_direct_kernel, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
    }
    return bgk_launch_status("bgk_rqs_backward");
}
