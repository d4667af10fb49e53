/* bgk_rqs_bwd.hip -- analytic backward (VJP) of the rational-quadratic spline transformer for
 * first-order losses (KL / NLL): replaces torch autograd through the reference's op chain
 * (nn/flow/transformer/spline.py:109-188 + nflows).  Same math as oracle/bgo_impl.h::bgo_rqs_backward.
 *
 * HBM-bound: per sample it reads P + 2d + 1 floats (params, y, g_out, g_dlogp) and writes P + d
 * floats (g_params, g_y): 4*(2P + 3d + 1) algorithmic bytes.  Every (sample, dim) element owns its 3K(+1)
 * parameter slots exclusively, so an element's gradients are written straight from registers.
 */
#include "bgk_common.h"

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
    int TS, Pp;
    uint32_t magicP;
    BgkRqsCfg cfg;
};

/* softmax probabilities p[k] and the K+1 knots of one parameter set */
template <int KT>
__device__ __forceinline__ void softmax_knots(const float* u, float mn, float sc, float span, float low, float high,
                                              float (&p)[KT], float (&kn)[KT + 1]) {
    float m = u[0];
#pragma unroll
    for (int k = 1; k < KT; ++k) m = u[k] > m ? u[k] : m;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) { p[k] = bgk_expf(u[k] - m); s += p[k]; }
    float c = 0.0f;
    kn[0] = low;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        p[k] = p[k] / s;
        c += mn + sc * p[k];
        kn[k + 1] = span * c + low;
    }
    kn[KT] = high;
}

template <int KT>
__device__ __forceinline__ float pick(const float (&a)[KT], int i) {
    float v = a[0];
#pragma unroll
    for (int k = 1; k < KT; ++k) v = (i == k) ? a[k] : v;
    return v;
}

/* Streaming form (same scheme as rqs_stream_kernel, bgk_rqs.hip): elements enumerated dim-fastest, the 24 parameters
 * of an element loaded with six 16-byte loads into registers, the 24 (+1) parameter gradients written back the same
 * way; no LDS, no barrier -- nothing in the backward pass crosses lanes. */
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int BWD_TS = 128;

template <int KT>
__global__ __launch_bounds__(BWD_THREADS) void rqs_bwd_kernel(RqsBwdArgs a) {
    constexpr int K = KT;
    static_assert(KT % 4 == 0, "KT / 4 16-byte loads per parameter group");
    const int d = a.d, tid = threadIdx.x;
    const BgkRqsCfg& c = a.cfg;
    const uint64_t magicd = (0x100000000ull + (uint64_t)d - 1) / (uint64_t)d;
    const int64_t n_tiles = (a.B + BWD_TS - 1) / BWD_TS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * BWD_TS;
        const int rows = (int)((a.B - b0) < BWD_TS ? (a.B - b0) : BWD_TS);
        for (int e = tid; e < rows * d; e += BWD_THREADS) {
            const int s = (int)(((uint64_t)(uint32_t)e * magicd) >> 32), j = e - s * d;
            const float* row = a.params + (b0 + s) * a.ldp;
            float* grow = a.g_params + (b0 + s) * a.ldgp;
            const float* gw = row + j * K;
            const float* gh = gw + d * K;
            const float* gs = gh + d * K;
            float rw[K], rh[K], rs[K], ow[K], oh[K], os[K];
#pragma unroll
            for (int q = 0; q < K / 4; ++q) {
                const f4u w4 = *reinterpret_cast<const f4u*>(gw + 4 * q), h4 = *reinterpret_cast<const f4u*>(gh + 4 * q);
                const f4u t4 = *reinterpret_cast<const f4u*>(gs + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) { rw[4 * q + k] = w4[k]; rh[4 * q + k] = h4[k]; rs[4 * q + k] = t4[k]; }
            }
            const int slot = a.nc_slot[j];
            const float s_K = slot >= 0 ? row[3 * d * K + slot] : rs[0];   /* slope at knot K */
            float pw[K], ph[K], cw[K + 1], ch[K + 1];
            softmax_knots<K>(rw, c.min_w, c.w_scale, c.xspan, c.left, c.right, pw, cw);
            softmax_knots<K>(rh, c.min_h, c.h_scale, c.yspan, c.bottom, c.top, ph, ch);
            float x = a.y[(b0 + s) * a.ldy + j];
            const bool clamped = (x < c.left) | (x > c.right);
            x = x < c.left ? c.left : (x > c.right ? c.right : x);
            int idx = -1;
#pragma unroll
            for (int k = 0; k <= K; ++k) {
                float kn = a.inverse ? cw[k] : ch[k];
                if (k == K) kn = kn + 1e-6f;
                idx += (x >= kn) ? 1 : 0;
            }
            idx = idx < 0 ? 0 : (idx > K - 1 ? K - 1 : idx);
            const bool hi_last = (idx + 1 == K);
            float cw_i = cw[0], cw_n = cw[1], ch_i = ch[0], ch_n = ch[1];
#pragma unroll
            for (int k = 1; k < K; ++k) {
                cw_i = (idx == k) ? cw[k] : cw_i; cw_n = (idx == k) ? cw[k + 1] : cw_n;
                ch_i = (idx == k) ? ch[k] : ch_i; ch_n = (idx == k) ? ch[k + 1] : ch_n;
            }
            const float s_lo = pick<K>(rs, idx);
            float s_hi = s_K;
#pragma unroll
            for (int k = 1; k < K; ++k) s_hi = (idx + 1 == k) ? rs[k] : s_hi;
            const float d0 = c.min_d + bgk_softplusf(s_lo, c.beta), d1 = c.min_d + bgk_softplusf(s_hi, c.beta);
            const float W_i = cw_n - cw_i, H_i = ch_n - ch_i;
            const float delta = H_i / W_i, S = d0 + d1 - 2.0f * delta;
            float theta;
            if (!a.inverse) {
                float dx = x - ch_i;
                float qa = dx * S + H_i * (delta - d0), qb = H_i * d0 - dx * S, qc = -delta * dx;
                theta = (2.0f * qc) / (-qb - __builtin_sqrtf(qb * qb - 4.0f * qa * qc));
            } else {
                theta = (x - cw_i) / W_i;
            }
            const float t = theta * (1.0f - theta), tp = 1.0f - 2.0f * theta, omt = 1.0f - theta;
            const float N = delta * theta * theta + d0 * t, den = delta + S * t;
            const float iden2 = 1.0f / (den * den);
            const float Q = N / den;
            const float N_th = 2.0f * delta * theta + d0 * tp, den_th = S * tp;
            const float Q_th = (N_th * den - N * den_th) * iden2;
            const float Q_de = (theta * theta * den - N * (1.0f - 2.0f * t)) * iden2;
            const float Q_d0 = (t * den - N * t) * iden2;
            const float Q_d1 = (-N * t) * iden2;
            const float M = d1 * theta * theta + 2.0f * delta * t + d0 * omt * omt;
            const float lf_th = (2.0f * d1 * theta + 2.0f * delta * tp - 2.0f * d0 * omt) / M - 2.0f * den_th / den;
            const float lf_de = 2.0f / delta + 2.0f * t / M - 2.0f * (1.0f - 2.0f * t) / den;
            const float lf_d0 = omt * omt / M - 2.0f * t / den;
            const float lf_d1 = theta * theta / M - 2.0f * t / den;
            const float gy = a.g_out[(b0 + s) * a.ldgo + j], gl = a.g_dlogp[b0 + s];
            float G_de, G_d0, G_d1, G_H, G_W, G_cw, G_ch, gx;
            if (a.inverse) {
                const float G_th = gy * H_i * Q_th + gl * lf_th;
                G_de = gy * H_i * Q_de + gl * lf_de;
                G_d0 = gy * H_i * Q_d0 + gl * lf_d0;
                G_d1 = gy * H_i * Q_d1 + gl * lf_d1;
                G_H = gy * Q + G_de / W_i;
                G_W = -G_de * delta / W_i - G_th * theta / W_i;
                G_ch = gy;
                G_cw = -G_th / W_i;
                gx = G_th / W_i;
            } else {
                const float A_th = gy * W_i - gl * lf_th;
                const float inv = 1.0f / (H_i * Q_th);
                G_de = -gl * lf_de - A_th * Q_de / Q_th;
                G_d0 = -gl * lf_d0 - A_th * Q_d0 / Q_th;
                G_d1 = -gl * lf_d1 - A_th * Q_d1 / Q_th;
                G_H = G_de / W_i - A_th * Q * inv;
                G_W = -G_de * delta / W_i + gy * theta;
                G_cw = gy;
                G_ch = -A_th * inv;
                gx = A_th * inv;
            }
            const bool dead = (gy == 0.0f) & (gl == 0.0f);   /* masked-out sample: exact zeros, never 0 * inf */
            if (dead) { G_de = G_d0 = G_d1 = G_H = G_W = G_cw = G_ch = gx = 0.0f; }
            a.g_y[(b0 + s) * a.ldgy + j] = clamped ? 0.0f : gx;
            {
                const float gA = (idx >= 1) ? (G_cw - G_W) : 0.0f, gB = (idx + 1 <= K - 1) ? G_W : 0.0f;
                float gp[K], dot = 0.0f;
#pragma unroll
                for (int m = 0; m < K; ++m) { gp[m] = c.w_scale * c.xspan * ((m < idx ? gA : 0.0f) + (m <= idx ? gB : 0.0f)); dot += pw[m] * gp[m]; }
#pragma unroll
                for (int m = 0; m < K; ++m) ow[m] = pw[m] * (gp[m] - dot);
            }
            {
                const float gA = (idx >= 1) ? (G_ch - G_H) : 0.0f, gB = (idx + 1 <= K - 1) ? G_H : 0.0f;
                float gp[K], dot = 0.0f;
#pragma unroll
                for (int m = 0; m < K; ++m) { gp[m] = c.h_scale * c.yspan * ((m < idx ? gA : 0.0f) + (m <= idx ? gB : 0.0f)); dot += ph[m] * gp[m]; }
#pragma unroll
                for (int m = 0; m < K; ++m) oh[m] = ph[m] * (gp[m] - dot);
            }
            {
                const float z0 = s_lo * c.beta, z1 = s_hi * c.beta;
                const float sg0 = z0 > 20.0f ? 1.0f : 1.0f / (1.0f + bgk_expf(-z0));
                const float sg1 = z1 > 20.0f ? 1.0f : 1.0f / (1.0f + bgk_expf(-z1));
                const float g0 = G_d0 * sg0, g1 = G_d1 * sg1;
                if (slot >= 0) grow[3 * d * K + slot] = hi_last ? g1 : 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float g = 0.0f;
                    g += (k == idx) ? g0 : 0.0f;
                    g += (!hi_last && k == idx + 1) ? g1 : 0.0f;
                    g += (hi_last && slot < 0 && k == 0) ? g1 : 0.0f;
                    os[k] = g;
                }
            }
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
}

}  // namespace

extern "C" int bgk_rqs_backward(const float* y, int64_t ldy, const float* params, int64_t ldp, int32_t P,
                                const int32_t* nc_slot, int64_t B, int32_t d, int32_t K, int32_t inverse,
                                double left, double right, double bottom, double top,
                                double min_bin_width, double min_bin_height, double min_derivative,
                                int32_t identity_init, const float* g_out, int64_t ldgo,
                                const float* g_dlogp, float* g_y, int64_t ldgy, float* g_params,
                                int64_t ldgp, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && d > 0 && K > 0, "bgk_rqs_backward: bad sizes");
    BGK_CHECK_ARG(y && params && nc_slot && g_out && g_dlogp && g_y && g_params, "bgk_rqs_backward: null pointer");
    BGK_CHECK_ARG(P >= 3 * K * d && P <= 3 * K * d + d && ldp >= P && ldgp >= P, "bgk_rqs_backward: bad params width %d", P);
    if (K != 4 && K != 8 && K != 12 && K != 16 && K != 32) {
        bgk_set_error("bgk_rqs_backward: n_bins in {4, 8, 12, 16, 32} are implemented (got %d)", K);
        return BGK_EUNSUPPORTED;
    }
    if (B == 0) return 0;
    RqsBwdArgs a;
    a.y = y; a.ldy = ldy; a.params = params; a.ldp = ldp; a.nc_slot = nc_slot; a.B = B; a.d = d; a.K = K; a.P = P;
    a.inverse = inverse; a.g_out = g_out; a.ldgo = ldgo; a.g_dlogp = g_dlogp; a.g_y = g_y; a.ldgy = ldgy;
    a.g_params = g_params; a.ldgp = ldgp;
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
    a.Pp = P | 1; a.TS = BWD_TS; a.magicP = 0;
    int64_t n_tiles = (B + BWD_TS - 1) / BWD_TS;
    int grid = (int)(n_tiles < 256 * 16 ? n_tiles : 256 * 16);
    const size_t shmem = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (K) {
        case 4: hipLaunchKernelGGL(rqs_bwd_kernel<4>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        case 8: hipLaunchKernelGGL(rqs_bwd_kernel<8>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        case 12: hipLaunchKernelGGL(rqs_bwd_kernel<12>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        case 16: hipLaunchKernelGGL(rqs_bwd_kernel<16>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
        default: hipLaunchKernelGGL(rqs_bwd_kernel<32>, dim3(grid), dim3(BWD_THREADS), shmem, st, a); break;
    }
    return bgk_launch_status("bgk_rqs_backward");
}
