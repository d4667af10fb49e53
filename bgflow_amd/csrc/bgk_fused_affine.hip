/* bgk_fused_affine.hip -- one-launch affine (RealNVP / NICE) coupling layer:
 * CouplingFlow(AffineTransformer(shift = DenseNet, scale = DenseNet)) (nn/flow/coupling.py:162-182,
 * nn/flow/transformer/affine.py:41-70, nn/dense.py:47-48).  Both conditioner MLPs run on the f16 matrix cores in
 * split-f16 form (bgk_mfma_h2.h), their outputs mu / s_raw stay in accumulator registers and the affine tail
 *   log_sigma = tanh(s_raw) * exp(log_alpha) [- mean];  y' = exp(log_sigma) y + mu | exp(-log_sigma) (y - mu);
 *   dlogp = +- sum log_sigma;  y' % 1 if circular
 * is applied in place: per sample the kernel reads d_c + d floats and writes d (+1) floats.
 * Roofline: HBM, 4 (d_c + 2 d + 2) B per sample (cfg 2: 392 B vs 520 B + conditioner activations unfused).
 * A wave owns 32 samples; 4 independent waves per workgroup; no LDS beyond the staged conditioner input.
 */
#include "bgk_mfma_h2.h"

namespace {

constexpr int AW = 4;
constexpr int ASROW = 33;

struct AffNet {
    const uint4 *A0, *A1, *A2;
    float c0, c1, c2;
    int act;
};

struct FusedAffArgs {
    const float* cond; int64_t ldc; int d_c; int S0; int periodic;
    AffNet shift, scale; int has_shift, has_scale;
    const float* log_alpha; int preserve_volume, is_circular, inverse;
    const float* y; int64_t ldy; int64_t B; int d;
    float* out; int64_t ldo; float* dlogp; int accumulate;
    int lds_per_wave;
    int vec4;                    /* y / out rows 16-byte aligned */
};

template <int HT, int OT>
__device__ __forceinline__ void net_eval(h2_f32x16 (&res)[OT], const AffNet& n, const float* s_x, int S0, int lane) {
    h2_f32x16 h[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = 0.0f;
    h2_gemm_lds<HT>(h, s_x, ASROW, S0, n.A0, lane);
#pragma unroll
    for (int m = 0; m < HT; ++m) h2_act_tile(h[m], n.c0, n.act);
    H2B<HT> bf;
    h2_make_b<HT>(bf, h);
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = 0.0f;
    h2_gemm_hidden<HT, HT>(h, bf, n.A1, lane);
#pragma unroll
    for (int m = 0; m < HT; ++m) h2_act_tile(h[m], n.c1, n.act);
    h2_make_b<HT>(bf, h);
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) res[m][r] = 0.0f;
    h2_gemm_hidden<OT, HT>(res, bf, n.A2, lane);
}

template <int HT, int OT>
__global__ __launch_bounds__(AW * 64, 2) void coupling_affine_dense_kernel(FusedAffArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* s_x = smem + (size_t)wave * a.lds_per_wave;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * AW + wave;
    if (tile >= n_tiles) return;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
    const int d = a.d;

    /* conditioner input [feature][sample] + constant-1 row (bias column) + zero pad rows */
    const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
    for (int i = lane; i < 32 * a.d_c; i += 64) {
        const int r = i / a.d_c, c = i - r * a.d_c;
        const float v = r < rows ? a.cond[(b0 + r) * a.ldc + c] : 0.0f;
        if (a.periodic) {          /* WrapPeriodic featuriser (nn/periodic.py:30-37), all inputs circular on [0, 1] */
            float sv, cv;
            bgk_sincos2pif(v, &sv, &cv);
            s_x[c * ASROW + r] = cv;
            s_x[(a.d_c + c) * ASROW + r] = sv;
        } else {
            s_x[c * ASROW + r] = v;
        }
    }
    for (int i = lane; i < (16 * a.S0 - n_in) * 32; i += 64)
        s_x[(n_in + (i >> 5)) * ASROW + (i & 31)] = (i >> 5) == 0 ? 1.0f : 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    h2_f32x16 mu[OT], sr[OT];
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { mu[m][r] = 0.0f; sr[m][r] = 0.0f; }
    if (a.has_shift) net_eval<HT, OT>(mu, a.shift, s_x, a.S0, lane);
    if (a.has_scale) net_eval<HT, OT>(sr, a.scale, s_x, a.S0, lane);

    /* ---- affine tail on the accumulator layout: this lane holds dims h2_row(m, r, hh) of sample j ---- */
    const float alpha = a.has_scale ? bgk_expf(a.log_alpha[0]) : 0.0f;
    float lsum = 0.0f;
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const bgk_f2 th = bgk_tanhf2((bgk_f2){sr[m][r] * a.scale.c2, sr[m][r + 1] * a.scale.c2});
            const float l0 = (a.has_scale && h2_row(m, r, hh) < d) ? th.x * alpha : 0.0f;
            const float l1 = (a.has_scale && h2_row(m, r + 1, hh) < d) ? th.y * alpha : 0.0f;
            sr[m][r] = l0; sr[m][r + 1] = l1;
            lsum += l0;
            lsum += l1;
        }
    float total = lsum + __shfl_xor(lsum, 32);
    if (a.preserve_volume && a.has_scale) {
        const float mean = total / (float)d;
        lsum = 0.0f;
#pragma unroll
        for (int m = 0; m < OT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool valid = h2_row(m, r, hh) < d;
                const float ls = valid ? sr[m][r] - mean : 0.0f;
                sr[m][r] = ls;
                lsum += ls;
            }
        total = lsum + __shfl_xor(lsum, 32);
    }
    if (j < rows) {
        const float* yr = a.y + (b0 + j) * a.ldy;
        float* orow = a.out + (b0 + j) * a.ldo;
        /* registers 4q..4q+3 of a tile hold 4 consecutive dims: one 16-byte access per group when aligned */
#pragma unroll
        for (int m = 0; m < OT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int dim0 = h2_row(m, 4 * q, hh);
                if (dim0 >= d) continue;
                const bool full = a.vec4 && dim0 + 4 <= d;
                float v[4];
                if (full) {
                    const float4 t4 = *reinterpret_cast<const float4*>(yr + dim0);
                    v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = dim0 + u < d ? yr[dim0 + u] : 0.0f;
                }
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * q + u;
                    const float mm = a.has_shift ? mu[m][r] * a.shift.c2 : 0.0f;
                    const float ls = sr[m][r];
                    float t = a.inverse ? bgk_expf(-ls) * (v[u] - mm) : bgk_expf(ls) * v[u] + mm;
                    if (a.is_circular) { t = t - __builtin_truncf(t); if (t < 0.0f) t = t + 1.0f; }
                    o[u] = t;
                }
                if (full) {
                    *reinterpret_cast<float4*>(orow + dim0) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (dim0 + u < d) orow[dim0 + u] = o[u];
                }
            }
        if (hh == 0) {
            const float dl = a.inverse ? -total : total;
            if (a.accumulate) a.dlogp[b0 + j] += dl; else a.dlogp[b0 + j] = dl;
        }
    }
}

}  // namespace

extern "C" int bgk_coupling_affine_dense_h2(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                            const void* sA0, const void* sA1, const void* sA2,
                                            float sc0, float sc1, float sc2, int32_t s_act,
                                            const void* tA0, const void* tA1, const void* tA2,
                                            float tc0, float tc1, float tc2, int32_t t_act,
                                            int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                            int32_t is_circular, int32_t inverse,
                                            const float* y, int64_t ldy, int64_t B, int32_t d,
                                            float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream) {
    BGK_CHECK_ARG(cond && y && out && dlogp, "bgk_coupling_affine_dense_h2: null pointer");
    BGK_CHECK_ARG(B >= 0 && d > 0 && d_c > 0, "bgk_coupling_affine_dense_h2: bad sizes");
    const int has_shift = sA0 != nullptr, has_scale = tA0 != nullptr;
    BGK_CHECK_ARG(has_shift || has_scale, "bgk_coupling_affine_dense_h2: no conditioner network");
    BGK_CHECK_ARG(!has_shift || (sA1 && sA2), "bgk_coupling_affine_dense_h2: incomplete shift network");
    BGK_CHECK_ARG(!has_scale || (tA1 && tA2 && log_alpha), "bgk_coupling_affine_dense_h2: incomplete scale network");
    BGK_CHECK_ARG(!(has_scale && is_circular), "Scaling is not compatible with periodicity.");
    const int n_in = periodic ? 2 * d_c : d_c;
    if ((hidden != 64 && hidden != 128) || d > 96 || n_in > 127 || s_act < 0 || s_act > 3 || t_act < 0 || t_act > 3) {
        bgk_set_error("bgk_coupling_affine_dense_h2: only hidden = (64,64) | (128,128), d <= 96, <= 127 input features are fused "
                      "(got hidden=%d d=%d n_in=%d)", hidden, d, n_in);
        return BGK_EUNSUPPORTED;
    }
    if (B == 0) return 0;
    FusedAffArgs a;
    a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic; a.S0 = (n_in + 1 + 15) / 16;
    a.shift = AffNet{(const uint4*)sA0, (const uint4*)sA1, (const uint4*)sA2, sc0, sc1, sc2, s_act};
    a.scale = AffNet{(const uint4*)tA0, (const uint4*)tA1, (const uint4*)tA2, tc0, tc1, tc2, t_act};
    a.has_shift = has_shift; a.has_scale = has_scale;
    a.log_alpha = log_alpha; a.preserve_volume = preserve_volume; a.is_circular = is_circular; a.inverse = inverse;
    a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.lds_per_wave = 16 * a.S0 * ASROW;
    a.vec4 = (ldy % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)y | (uintptr_t)out) % 16 == 0);
    const size_t shmem = sizeof(float) * (size_t)AW * a.lds_per_wave;
    const int64_t n_wg = ((B + 31) / 32 + AW - 1) / AW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_coupling_affine_dense_h2: batch too large for one launch");
    const int OT = (d + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH(H, O) hipLaunchKernelGGL((coupling_affine_dense_kernel<H, O>), dim3((int)n_wg), dim3(AW * 64), shmem, st, a)
    if (hidden == 64) { if (OT == 1) BGK_LAUNCH(2, 1); else if (OT == 2) BGK_LAUNCH(2, 2); else BGK_LAUNCH(2, 3); }
    else { if (OT == 1) BGK_LAUNCH(4, 1); else if (OT == 2) BGK_LAUNCH(4, 2); else BGK_LAUNCH(4, 3); }
#undef BGK_LAUNCH
    return bgk_launch_status("bgk_coupling_affine_dense_h2");
}
