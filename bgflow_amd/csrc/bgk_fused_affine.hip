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
#include "bgk_fused2.h"

int bgk_affine_variant = 2;      /* 1: streaming kernel only, 2: weight-resident kernel where the operands fit LDS (hidden 64) and the
                                  * event-threaded kernel of bgk_fused2.hip for hidden (128,128) (bgk_set_option 2) */

namespace {

constexpr int AW = 4;
constexpr int ASROW = 33;

struct AffNet {
    const uint4 *A0, *A1, *A2;
    float c0, c1, c2;
    int act;
    const uint4* A1b;            /* second H x H layer of a three-hidden-layer network, else NULL */
    float c1b;
};

struct FusedAffArgs {
    const float* cond; int64_t ldc; int d_c; int S0; int periodic;
    AffNet shift, scale; int has_shift, has_scale;
    const float* log_alpha; int preserve_volume, is_circular, inverse;
    const float* y; int64_t ldy; int64_t B; int d;
    float* out; int64_t ldo; float* dlogp; int accumulate;
    int lds_per_wave;
    int vec4;                    /* y / out rows 16-byte aligned */
    int cvec4;                   /* cond rows 16-byte aligned */
};

/* ---- shared building blocks of both kernels (A fragments through a pointer: global memory in the streaming kernel, LDS in the
 * weight-resident one) ---- */
typedef unsigned int r_u32x4 __attribute__((ext_vector_type(4)));
#ifndef BGK_AFF_ABL
#define BGK_AFF_ABL 0          /* diagnostic builds: 1 no global loads, 2 no hidden activation math, 4 no output-layer math, 8 no MFMAs */
#endif
#ifndef BGK_AFF_NOPRIO
#define BGK_AFF_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define BGK_AFF_PRIO(p)
#endif

template <int NT>
struct RA { r_u32x4 v[NT][2]; };

template <int NT>
__device__ __forceinline__ void ra_load(RA<NT>& f, const r_u32x4* W, int s, int lane) {
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        f.v[m][0] = W[((s * NT + m) * 2 + 0) * 64 + lane];
        f.v[m][1] = W[((s * NT + m) * 2 + 1) * 64 + lane];
    }
}
/* 3 MFMAs of one k-step (lo*hi, hi*lo, hi*hi).  ZERO: the accumulators start from the inline constant 0 (saves the 16
 * v_mov per tile); NOLO: the B operand has no lo part (constant-1 feature of the bias column): 2 MFMAs. */
template <int NT, bool ZERO, bool NOLO>
__device__ __forceinline__ void ra_mfma3(h2_f32x16 (&out)[NT], const RA<NT>& a, const h2_h16x8& bhi, const h2_h16x8& blo) {
    const h2_f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if (BGK_AFF_ABL & 8)
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        if (ZERO) out[m] = z;
        out[m][0] += __builtin_bit_cast(float, a.v[m][0][0] ^ a.v[m][1][1]) + (float)bhi[0] + (float)blo[1];
    }
    return;
#endif
#pragma unroll
    for (int m = 0; m < NT; ++m)
        out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, a.v[m][1]), bhi, ZERO ? z : out[m], 0, 0, 0);
    if (!NOLO) {
#pragma unroll
        for (int m = 0; m < NT; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, a.v[m][0]), blo, out[m], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < NT; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, a.v[m][0]), bhi, out[m], 0, 0, 0);
}

template <int HT>
struct RB { r_u32x4 hi[2 * HT], lo[2 * HT]; };

/* out = W' * b + bias'  with W' resident in LDS (K = 32 HT); out need not be initialised */
template <int NT, int HT, bool PIN = false>
__device__ __forceinline__ void ra_gemm_hidden(h2_f32x16 (&out)[NT], const RB<HT>& b, const r_u32x4* W, int lane) {
    constexpr int S = 2 * HT;
    RA<NT> ring[2];
    ra_load<NT>(ring[0], W, 0, lane);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) ra_load<NT>(ring[(s + 1) & 1], W, s + 1, lane);
        else {
#pragma unroll
            for (int m = 0; m < NT; ++m) ring[(s + 1) & 1].v[m][0] = W[(S * NT * 2 + m) * 64 + lane];
        }
        if (PIN) __builtin_amdgcn_sched_barrier(0);       /* operands from L2: the next step's fragments stay requested ahead of this step's MFMAs
                                                           * (with operands in LDS the fence costs 10 %: the scheduler interleaves better) */
        if (s == 0) ra_mfma3<NT, true, false>(out, ring[0], __builtin_bit_cast(h2_h16x8, b.hi[0]), __builtin_bit_cast(h2_h16x8, b.lo[0]));
        else ra_mfma3<NT, false, false>(out, ring[s & 1], __builtin_bit_cast(h2_h16x8, b.hi[s]), __builtin_bit_cast(h2_h16x8, b.lo[s]));
    }
    const h2_h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < NT; ++m)
        out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, ring[S & 1].v[m][0]), one2, out[m], 0, 0, 0);
}

/* the same with the fragments of k-step 0 already requested by the caller (`first`): the resident kernels request them BEFORE the
 * activation code that produces the B operands, so the LDS round trip of the first k-step hides under that arithmetic */
template <int NT, int HT>
__device__ __forceinline__ void ra_gemm_hidden_pre(h2_f32x16 (&out)[NT], const RB<HT>& b, const r_u32x4* W, int lane, const RA<NT>& first) {
    constexpr int S = 2 * HT;
    RA<NT> ring[2];
    ring[0] = first;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) ra_load<NT>(ring[(s + 1) & 1], W, s + 1, lane);
        else {
#pragma unroll
            for (int m = 0; m < NT; ++m) ring[(s + 1) & 1].v[m][0] = W[(S * NT * 2 + m) * 64 + lane];
        }
#ifndef BGK_AFF_NOPIN
        /* keep the next step's LDS reads IN FRONT of this step's MFMAs: left alone, the scheduler sinks every ds_read_b128 to its use
         * (ds_read; s_waitcnt lgkmcnt(0); v_mfma -- the whole LDS latency exposed per k-step at two waves per SIMD) */
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (s == 0) ra_mfma3<NT, true, false>(out, ring[0], __builtin_bit_cast(h2_h16x8, b.hi[0]), __builtin_bit_cast(h2_h16x8, b.lo[0]));
        else ra_mfma3<NT, false, false>(out, ring[s & 1], __builtin_bit_cast(h2_h16x8, b.hi[s]), __builtin_bit_cast(h2_h16x8, b.lo[s]));
#ifndef BGK_AFF_NOPIN
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    const h2_h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < NT; ++m)
        out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, ring[S & 1].v[m][0]), one2, out[m], 0, 0, 0);
}

/* hidden activation of t * c with the hardware exp2 / rcp forms (bgk_detmath_pk.h explains why these are admissible for
 * HIDDEN activations), immediately split into f16 hi + lo: scalar (non-packed) VALU so that it overlaps other waves' MFMAs.
 * ACT 0 identity, 1 SiLU, 2 ReLU, 3 Tanh.  k = c (ACT 0, 2), c * log2(e) (1), 2 c log2(e) (3). */
template <int ACT>
__device__ __forceinline__ float r_act(float t, float c, float k) {
#if (BGK_AFF_ABL & 2)
    return t * c;
#endif
    if constexpr (ACT == 2) return __builtin_amdgcn_fmed3f(t * c, 0.0f, 65000.0f);
    else if constexpr (ACT == 3) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * k));
    else if constexpr (ACT == 1) {
        const float x = t * c;
        return __builtin_fminf(x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-(t * k))), 65000.0f);
    } else return __builtin_amdgcn_fmed3f(t * c, -65000.0f, 65000.0f);
}

/* f32 pair -> f16 hi pair + f16 lo pair in 3 instructions: v_cvt_pk_f16_f32 (RNE), then lo = f16(a - hi) with the mixed-precision
 * FMA reading hi as an f16 operand and writing one half of the destination (a - hi is exact in f32, so the single rounding equals
 * (_Float16)(a - (float)hi)). */
__device__ __forceinline__ void r_split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a0), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
}

template <int ACT, int HT>
__device__ __forceinline__ void r_act_split_t(RB<HT>& b, const h2_f32x16 (&in)[HT], float c) {
    const float k = ACT == 3 ? c * 2.88539008177792681f : (ACT == 1 ? c * 1.44269504088896341f : c);
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float a0 = r_act<ACT>(in[m][r], c, k), a1 = r_act<ACT>(in[m][r + 1], c, k);
            unsigned hi, lo;
            r_split_pair(a0, a1, hi, lo);
            const int s = 2 * m + (r >> 3), e = (r & 7) >> 1;
            b.hi[s][e] = hi; b.lo[s][e] = lo;
        }
}
template <int HT>
__device__ __forceinline__ void r_act_split(RB<HT>& b, const h2_f32x16 (&in)[HT], float c, int act) {
    if (act == 2) r_act_split_t<2, HT>(b, in, c);
    else if (act == 3) r_act_split_t<3, HT>(b, in, c);
    else if (act == 1) r_act_split_t<1, HT>(b, in, c);
    else r_act_split_t<0, HT>(b, in, c);
}

/* tanh for the OUTPUT layer (log sigma): hardware exp2 + Newton-refined rcp above 0.625 (abs error ~1e-7), odd polynomial below
 * (same coefficients as bgk_tanhf2) */
__device__ __forceinline__ float r_tanh_out(float x) {
#if (BGK_AFF_ABL & 4)
    return x;
#endif
    const float ax = __builtin_fabsf(x);
    const float d = 1.0f + __builtin_amdgcn_exp2f(ax * 2.88539008177792681f);
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(r, __builtin_fmaf(-d, r, 1.0f), r);
    const float big = __builtin_copysignf(__builtin_fmaf(-2.0f, r, 1.0f), x);
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    const float small = __builtin_fmaf(p * z, x, x);
    return ax >= 0.625f ? big : small;
}

template <int HT, int OT>
__device__ __forceinline__ void net_eval(h2_f32x16 (&res)[OT], const AffNet& n, const float* s_x, int S0, int lane) {
    h2_f32x16 h[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = 0.0f;
    h2_gemm_lds<HT>(h, s_x, ASROW, S0, n.A0, lane);
    /* activation + f16 hi / lo split in the lean forms of the resident kernel (hardware exp2 / rcp, 3-instruction split) */
    RB<HT> bf;
    r_act_split<HT>(bf, h, n.c0, n.act);
    ra_gemm_hidden<HT, HT, true>(h, bf, reinterpret_cast<const r_u32x4*>(n.A1), lane);
    r_act_split<HT>(bf, h, n.c1, n.act);
    if (n.A1b) {                 /* three hidden layers (e.g. the ala2 RealNVP conditioners [30, 128, 128, 128, 30]) */
        ra_gemm_hidden<HT, HT, true>(h, bf, reinterpret_cast<const r_u32x4*>(n.A1b), lane);
        r_act_split<HT>(bf, h, n.c1b, n.act);
    }
    ra_gemm_hidden<OT, HT, true>(res, bf, reinterpret_cast<const r_u32x4*>(n.A2), lane);
}

/* conditioner input of a 32-sample tile [feature][sample] + constant-1 row (bias column) + zero pad rows */
__device__ __forceinline__ void aff_stage_cond(const FusedAffArgs& a, float* s_x, int64_t b0, int rows, int lane) {
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

}

/* affine tail on the accumulator layout (mu, sr = the two networks' raw outputs): this lane holds dims h2_row(m, r, hh) of sample j */
template <int OT>
__device__ __forceinline__ void aff_tail(const FusedAffArgs& a, h2_f32x16 (&mu)[OT], h2_f32x16 (&sr)[OT], int64_t b0, int rows, int lane) {
    const int j = lane & 31, hh = lane >> 5;
    const int d = a.d;
    const float alpha = a.has_scale ? bgk_expf(a.log_alpha[0]) : 0.0f;
    float lsum = 0.0f;
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float l0 = (a.has_scale && h2_row(m, r, hh) < d) ? r_tanh_out(sr[m][r] * a.scale.c2) * alpha : 0.0f;
            const float l1 = (a.has_scale && h2_row(m, r + 1, hh) < d) ? r_tanh_out(sr[m][r + 1] * a.scale.c2) * alpha : 0.0f;
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
                    const float sg = __builtin_amdgcn_exp2f((a.inverse ? -ls : ls) * 1.44269504088896341f);     /* |ls| <= exp(log_alpha): 1 ulp */
                    float t = a.inverse ? sg * (v[u] - mm) : sg * v[u] + mm;
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

template <int HT, int OT>
__global__ __launch_bounds__(AW * 64, 2) void coupling_affine_dense_kernel(FusedAffArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_x = smem + (size_t)wave * a.lds_per_wave;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * AW + wave;
    if (tile >= n_tiles) return;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
    aff_stage_cond(a, s_x, b0, rows, lane);
    h2_f32x16 mu[OT], sr[OT];
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { mu[m][r] = 0.0f; sr[m][r] = 0.0f; }
    if (a.has_shift) net_eval<HT, OT>(mu, a.shift, s_x, a.S0, lane);
    if (a.has_scale) net_eval<HT, OT>(sr, a.scale, s_x, a.S0, lane);
    aff_tail<OT>(a, mu, sr, b0, rows, lane);
}

/* ---- conditioners with ANY number of hidden layers, n_hidden = 1 .. 8 (other than the 2 / 3 of the kernels above; README.md:72-79's
 * [1, 4, 1] networks have one): the streaming kernel with the hidden -> hidden GEMM + activation in a loop over the packed layers
 * (AffNet::A1 = their operands back to back, c1s[l] their unscale factors; both networks have the same depth) ---- */
constexpr int AFF_DEEP_MAX_HH = 7;
struct DeepAffArgs { FusedAffArgs a; int n_hh; float sc1s[AFF_DEEP_MAX_HH], tc1s[AFF_DEEP_MAX_HH]; };

template <int HT, int OT>
__device__ __forceinline__ void net_eval_deep(h2_f32x16 (&res)[OT], const AffNet& n, int n_hh, const float (&c1s)[AFF_DEEP_MAX_HH],
                                              const float* s_x, int S0, int lane) {
    h2_f32x16 h[HT];
#pragma unroll
    for (int m = 0; m < HT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = 0.0f;
    h2_gemm_lds<HT>(h, s_x, ASROW, S0, n.A0, lane);
    RB<HT> bf;
    r_act_split<HT>(bf, h, n.c0, n.act);
    constexpr int LAYER16 = (2 * HT * HT * 2 + HT) * 64;          /* 16-byte units of one packed hidden -> hidden layer */
    for (int l = 0; l < n_hh; ++l) {
        ra_gemm_hidden<HT, HT, true>(h, bf, reinterpret_cast<const r_u32x4*>(n.A1) + (size_t)l * LAYER16, lane);
        r_act_split<HT>(bf, h, c1s[l], n.act);
    }
    ra_gemm_hidden<OT, HT, true>(res, bf, reinterpret_cast<const r_u32x4*>(n.A2), lane);
}

template <int HT, int OT>
__global__ __launch_bounds__(AW * 64, 2) void coupling_affine_deep_kernel(DeepAffArgs da) {
    const FusedAffArgs& a = da.a;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_x = smem + (size_t)wave * a.lds_per_wave;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * AW + wave;
    if (tile >= n_tiles) return;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
    aff_stage_cond(a, s_x, b0, rows, lane);
    h2_f32x16 mu[OT], sr[OT];
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { mu[m][r] = 0.0f; sr[m][r] = 0.0f; }
    if (a.has_shift) net_eval_deep<HT, OT>(mu, a.shift, da.n_hh, da.sc1s, s_x, a.S0, lane);
    if (a.has_scale) net_eval_deep<HT, OT>(sr, a.scale, da.n_hh, da.tc1s, s_x, a.S0, lane);
    aff_tail<OT>(a, mu, sr, b0, rows, lane);
}


/* ---- weight-resident variant (hidden = 64): both conditioners' packed operands (78 KB for cfg 2) are staged ONCE per
 * workgroup in LDS and the workgroup's waves loop over 32-sample tiles.  The streaming kernel above re-reads every operand
 * block from L2 for every 32 samples (2.7 KB of L2 traffic per sample against 392 B of HBM traffic: L2-bandwidth bound,
 * 0.29 ms per cfg 2 layer); here the A fragments come from LDS (ds_read_b128, 624 LDS cycles per tile) and the conditioner
 * input is loaded from global memory directly in B-operand layout (lane = sample, 8 consecutive features), so the only
 * per-tile memory traffic is the algorithmic d_c + 2 d + 1 floats per sample. ---- */
struct ResOff { int a0, a1, a2; };          /* offsets (16-byte units) of a network's operands in the LDS image */

template <int HT, int OT>
__device__ __forceinline__ void res_net_tail(h2_f32x16 (&res)[OT], h2_f32x16 (&h)[HT], const AffNet& n, const r_u32x4* s_w, ResOff o, int lane) {
    RB<HT> bf;
#ifdef BGK_AFF_NOPRE
    r_act_split<HT>(bf, h, n.c0, n.act);
    BGK_AFF_PRIO(1);               /* matrix phases first: the other waves of the SIMD fill the gaps with their activation arithmetic */
    ra_gemm_hidden<HT, HT>(h, bf, s_w + o.a1, lane);
    BGK_AFF_PRIO(0);
    r_act_split<HT>(bf, h, n.c1, n.act);
    BGK_AFF_PRIO(1);
    ra_gemm_hidden<OT, HT>(res, bf, s_w + o.a2, lane);
    BGK_AFF_PRIO(0);
#else
    RA<HT> f1;
    ra_load<HT>(f1, s_w + o.a1, 0, lane);          /* the GEMM's first fragments travel while the activation code runs */
    r_act_split<HT>(bf, h, n.c0, n.act);
    BGK_AFF_PRIO(1);               /* matrix phases first: the other waves of the SIMD fill the gaps with their activation arithmetic */
    ra_gemm_hidden_pre<HT, HT>(h, bf, s_w + o.a1, lane, f1);
    BGK_AFF_PRIO(0);
    RA<OT> f2;
    ra_load<OT>(f2, s_w + o.a2, 0, lane);
    r_act_split<HT>(bf, h, n.c1, n.act);
    BGK_AFF_PRIO(1);
    ra_gemm_hidden_pre<OT, HT>(res, bf, s_w + o.a2, lane, f2);
    BGK_AFF_PRIO(0);
#endif
}

/* ---- both networks of a coupling layer as ONE software pipeline (round 5).  res_net_tail runs a network as activation -> GEMM ->
 * activation -> GEMM: pure-VALU and pure-MFMA phases that overlap only across the two waves of a SIMD (r04_cfg2_pmc.txt: MFMA busy
 * 115 M, VALU active 63 M, SQ_WAIT_ANY 56 M of 183 M wave cycles).  The shift and the scale network are independent, so a GEMM of one
 * can carry the activation (+ f16 split) of the other inside the same wave: per k-step of the GEMM (2 HT tiles x 3 MFMAs) a quarter of
 * the other network's 16 activation pairs, laid out MFMA / VALU / VALU ... by sched_group_barrier.
 *   stage A  act(hs)                       stage B  GEMM1_s  ||  act(ht)          stage C  GEMM1_t  ||  act(h1s)
 *   stage D  GEMM2_s || act(h1t)           stage E  GEMM2_t
 * BGK_AFF_PIPE2=0: the two networks one after the other (res_net_tail). */
#ifndef BGK_AFF_PIPE2
#define BGK_AFF_PIPE2 1
#endif
#ifndef BGK_AFF_PIPE2_VPM
#define BGK_AFF_PIPE2_VPM 7            /* VALU instructions the pattern asks for behind every MFMA */
#endif

/* activation + split of pairs [p0, p0 + np) of a 2-tile accumulator set (pair p: tile p >> 3, rows 2 (p & 7), + 1) */
template <int ACT, int HT>
__device__ __forceinline__ void r_act_split_pairs(RB<HT>& b, const h2_f32x16 (&in)[HT], float c, float k, int p0, int np) {
#pragma unroll
    for (int p = p0; p < p0 + np; ++p) {
        const int m = p >> 3, r = 2 * (p & 7);
        const float a0 = r_act<ACT>(in[m][r], c, k), a1 = r_act<ACT>(in[m][r + 1], c, k);
        unsigned hi, lo;
        r_split_pair(a0, a1, hi, lo);
        const int s = 2 * m + (r >> 3), e = (r & 7) >> 1;
        b.hi[s][e] = hi; b.lo[s][e] = lo;
    }
}

/* out = W' * b + bias' (ra_gemm_hidden_pre) while `other_in` is activated and split into `other_b`: S = 2 HT k-steps, each followed by
 * 8 HT / S activation pairs of the other network */
template <int ACT, int NT, int HT>
__device__ __forceinline__ void ra_gemm_with_act(h2_f32x16 (&out)[NT], const RB<HT>& b, const r_u32x4* W, int lane, const RA<NT>& first,
                                                 RB<HT>& other_b, const h2_f32x16 (&other_in)[HT], float oc) {
    constexpr int S = 2 * HT, PPS = (8 * HT) / S;
    const float ok = ACT == 3 ? oc * 2.88539008177792681f : (ACT == 1 ? oc * 1.44269504088896341f : oc);
    RA<NT> ring[2];
    ring[0] = first;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) ra_load<NT>(ring[(s + 1) & 1], W, s + 1, lane);
        else {
#pragma unroll
            for (int m = 0; m < NT; ++m) ring[(s + 1) & 1].v[m][0] = W[(S * NT * 2 + m) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);          /* the next step's LDS reads stay in front of this step's MFMAs (see ra_gemm_hidden_pre) */
        if (s == 0) ra_mfma3<NT, true, false>(out, ring[0], __builtin_bit_cast(h2_h16x8, b.hi[0]), __builtin_bit_cast(h2_h16x8, b.lo[0]));
        else ra_mfma3<NT, false, false>(out, ring[s & 1], __builtin_bit_cast(h2_h16x8, b.hi[s]), __builtin_bit_cast(h2_h16x8, b.lo[s]));
        r_act_split_pairs<ACT, HT>(other_b, other_in, oc, ok, PPS * s, PPS);
#pragma unroll
        for (int q = 0; q < 3 * NT; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       /* one MFMA */
            __builtin_amdgcn_sched_group_barrier(0x002, BGK_AFF_PIPE2_VPM, 0);       /* ... then VALU work of the other network */
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const h2_h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < NT; ++m)
        out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, ring[S & 1].v[m][0]), one2, out[m], 0, 0, 0);
}

/* the two networks (same activation ACT) behind their layers 0: hs / ht in, mu / sr (unscaled output-layer accumulators) out */
template <int AS, int AT, int HT, int OT>
__device__ __forceinline__ void res_two_nets_t(h2_f32x16 (&mu)[OT], h2_f32x16 (&sr)[OT], h2_f32x16 (&hs)[HT], h2_f32x16 (&ht)[HT],
                                               const AffNet& ns, const AffNet& nt, const r_u32x4* s_w, ResOff os, ResOff ot, int lane) {
    RB<HT> bs, bt;
    RA<HT> f1;
    ra_load<HT>(f1, s_w + os.a1, 0, lane);
    r_act_split_t<AS, HT>(bs, hs, ns.c0);                                                   /* A */
    BGK_AFF_PRIO(1);
    ra_gemm_with_act<AT, HT, HT>(hs, bs, s_w + os.a1, lane, f1, bt, ht, nt.c0);             /* B: hs <- layer-1 pre-activations of the shift net */
    ra_load<HT>(f1, s_w + ot.a1, 0, lane);
    ra_gemm_with_act<AS, HT, HT>(ht, bt, s_w + ot.a1, lane, f1, bs, hs, ns.c1);             /* C */
    RA<OT> f2;
    ra_load<OT>(f2, s_w + os.a2, 0, lane);
    ra_gemm_with_act<AT, OT, HT>(mu, bs, s_w + os.a2, lane, f2, bt, ht, nt.c1);             /* D */
    ra_load<OT>(f2, s_w + ot.a2, 0, lane);
    ra_gemm_hidden_pre<OT, HT>(sr, bt, s_w + ot.a2, lane, f2);                              /* E */
    BGK_AFF_PRIO(0);
}
constexpr int RES_HT = 2;

template <int OT, int RW>
__global__ __launch_bounds__(RW * 64, 1) void coupling_affine_resident_kernel(FusedAffArgs a, ResOff os, ResOff ot, int n16_s0, int n16_s1, int n16_s2,
                                                                              int n16_t0, int n16_t1, int n16_t2) {
    constexpr int HT = RES_HT;
    extern __shared__ __attribute__((aligned(16))) r_u32x4 s_w[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    {
        const r_u32x4* src[6] = {(const r_u32x4*)a.shift.A0, (const r_u32x4*)a.shift.A1, (const r_u32x4*)a.shift.A2,
                                 (const r_u32x4*)a.scale.A0, (const r_u32x4*)a.scale.A1, (const r_u32x4*)a.scale.A2};
        const int cnt[6] = {n16_s0, n16_s1, n16_s2, n16_t0, n16_t1, n16_t2};
        const int off[6] = {os.a0, os.a1, os.a2, ot.a0, ot.a1, ot.a2};
#pragma unroll
        for (int q = 0; q < 6; ++q)
            for (int i = tid; i < cnt[q]; i += RW * 64) s_w[off[q] + i] = src[q][i];
    }
    __syncthreads();
    const int d = a.d, d_c = a.d_c;
    const int n_in = a.periodic ? 2 * d_c : d_c;
    const float alpha = a.has_scale ? bgk_expf(a.log_alpha[0]) : 0.0f;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile_first = (int64_t)blockIdx.x * RW + wave, tile_step = (int64_t)gridDim.x * RW;
    for (int64_t tile = tile_first; tile < n_tiles; tile += tile_step) {
        const int64_t b0 = tile * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
        const int jr = j < rows ? j : rows - 1;                 /* rows past the batch end compute on a valid row, nothing is stored */
        const float* crow = a.cond + (b0 + jr) * a.ldc;

        h2_f32x16 hs[HT], ht[HT];
        for (int s = 0; s < a.S0; ++s) {
            const int f0 = 16 * s + 8 * hh;
            h2_h16x8 bhi, blo;
            const bool bias_only = 16 * s >= n_in;              /* wave-uniform: the k-step holds only the constant-1 feature */
            if (bias_only) {
                const _Float16 one = f0 == n_in ? (_Float16)1.0f : (_Float16)0.0f;
                bhi = h2_h16x8{one, 0, 0, 0, 0, 0, 0, 0};
                blo = h2_h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            } else {
                float v[8];
                if (!a.periodic && a.cvec4 && 16 * s + 16 <= d_c) {
#if (BGK_AFF_ABL & 1)
                    const float fl = (float)lane * 0.01f;
                    const float4 t0 = make_float4(fl, fl + 1.f, fl - 1.f, fl), t1 = make_float4(-fl, fl + .5f, fl - .5f, fl);
#else
                    const float4 t0 = *reinterpret_cast<const float4*>(crow + f0), t1 = *reinterpret_cast<const float4*>(crow + f0 + 4);
#endif
                    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int f = f0 + e;
                        float x = f == n_in ? 1.0f : 0.0f;          /* constant-1 feature: the bias column of the packed layer */
                        if (a.periodic) {                           /* WrapPeriodic featuriser (nn/periodic.py:30-37) */
                            if (f < n_in) {
                                float sv, cv;
                                bgk_sincos2pif(crow[f < d_c ? f : f - d_c], &sv, &cv);
                                x = f < d_c ? cv : sv;
                            }
                        } else if (f < d_c) x = crow[f];
                        v[e] = x;
                    }
                }
                r_u32x4 uh, ul;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float c0 = __builtin_amdgcn_fmed3f(v[e], -65000.0f, 65000.0f), c1 = __builtin_amdgcn_fmed3f(v[e + 1], -65000.0f, 65000.0f);
                    unsigned hi, lo;
                    r_split_pair(c0, c1, hi, lo);
                    uh[e >> 1] = hi; ul[e >> 1] = lo;
                }
                bhi = __builtin_bit_cast(h2_h16x8, uh); blo = __builtin_bit_cast(h2_h16x8, ul);
            }
            RA<HT> fr;
            if (a.has_shift) {
                ra_load<HT>(fr, s_w + os.a0, s, lane);
                if (s == 0) ra_mfma3<HT, true, false>(hs, fr, bhi, blo);
                else if (bias_only) ra_mfma3<HT, false, true>(hs, fr, bhi, blo);
                else ra_mfma3<HT, false, false>(hs, fr, bhi, blo);
            }
            if (a.has_scale) {
                ra_load<HT>(fr, s_w + ot.a0, s, lane);
                if (s == 0) ra_mfma3<HT, true, false>(ht, fr, bhi, blo);
                else if (bias_only) ra_mfma3<HT, false, true>(ht, fr, bhi, blo);
                else ra_mfma3<HT, false, false>(ht, fr, bhi, blo);
            }
        }
        h2_f32x16 mu[OT], sr[OT];
        if (a.has_shift) res_net_tail<HT, OT>(mu, hs, a.shift, s_w, os, lane);
        else {
#pragma unroll
            for (int m = 0; m < OT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) mu[m][r] = 0.0f;
        }
        if (a.has_scale) res_net_tail<HT, OT>(sr, ht, a.scale, s_w, ot, lane);
        else {
#pragma unroll
            for (int m = 0; m < OT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) sr[m][r] = 0.0f;
        }

        float lsum = 0.0f;
#pragma unroll
        for (int m = 0; m < OT; ++m)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float l0 = (a.has_scale && h2_row(m, r, hh) < d) ? r_tanh_out(sr[m][r] * a.scale.c2) * alpha : 0.0f;
                const float l1 = (a.has_scale && h2_row(m, r + 1, hh) < d) ? r_tanh_out(sr[m][r + 1] * a.scale.c2) * alpha : 0.0f;
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
#pragma unroll
            for (int m = 0; m < OT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int dim0 = h2_row(m, 4 * q, hh);
                    if (dim0 >= d) continue;
                    const bool full = a.vec4 && dim0 + 4 <= d;
                    float v[4];
                    if (full) {
#if (BGK_AFF_ABL & 1)
                        const float4 t4 = make_float4((float)lane, 1.f, 2.f, 3.f);
#else
                        const float4 t4 = *reinterpret_cast<const float4*>(yr + dim0);
#endif
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
        #if (BGK_AFF_ABL & 4)
                        const float sg = ls;
#else
                        const float sg = __builtin_amdgcn_exp2f((a.inverse ? -ls : ls) * 1.44269504088896341f);     /* |ls| <= exp(log_alpha): 1 ulp */
#endif
                        float t = a.inverse ? sg * (v[u] - mm) : sg * v[u] + mm;
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
}

/* ---- weight-resident variant with DMA-staged tiles (round 4): the layers of a `Split -> (Coupling, Swap)* -> Merge` stack read and
 * write column halves of one [B, D] buffer, i.e. 128-byte row pieces.  The kernel above fetches them with one 16-byte load per lane
 * and ROW (64 cache lines per wave instruction, the tile's latency exposed once per tile: SQ_WAIT_ANY 44 % of the wave cycles).  Here
 * the conditioner half and the y half of a 32-sample tile travel global -> LDS by the DMA path, whole lines per 8 lanes, and the
 * NEXT tile's halves are requested while the current tile computes: the conditioner half as soon as layer 0 has consumed the
 * current one, the y half as soon as the epilogue has read the current one back.  vmcnt bookkeeping (requests retire in order):
 *   top of a tile        : outstanding = [C_t | stores of tile t-1 | Y_t]          -> vmcnt(NY)   = C_t has landed
 *   before the epilogue  : outstanding = [Y_t | dlogp load | C_t+1]                -> vmcnt(NC)   = Y_t and the dlogp value are here
 * LDS image of a half: 16-byte granule (row j, column c) sits at granule j G + (c + rot(j)) mod G, rot(j) = j G / 16 -- the DMA
 * writes linearly (lane l of request i -> granule 64 i + l), so the rotation is applied to the global SOURCE address of each lane;
 * it makes the per-lane-row ds_read_b128 / ds_write_b128 of the B-operand build and of the epilogue bank-conflict free (row
 * stride 128 B = half the LDS width).  Envelope: d_c = d = 32 (G = 8), two hidden layers of 64, no periodic featuriser, 16-byte
 * aligned rows; everything else runs on the kernels above.  Measured (tools/r04_cfg2_ab.sh, same box, ms per 8-layer flow at 2^20): this
 * kernel 1.57, the kernel above 1.62; a three-waves-per-SIMD form of it (one tile buffer per wave, networks one after the other to
 * fit 168 VGPRs: 25 spilled registers, the next conditioner half requested behind the stores) 1.83 -- not shipped.  Per-phase wave
 * cycles (BGK_AFF_TS, tools/r04_cfg2_ts.py): of 21 k cycles per tile 0.6 k wait for the conditioner half, i.e. the memory latency is
 * hidden; what remains is the alternation of pure-VALU (activation, split) and pure-MFMA phases at two waves per SIMD. ---- */
typedef const __attribute__((address_space(1))) void* r_gvp_t;
typedef __attribute__((address_space(3))) void* r_lvp_t;

/* the NQ = 32 G / 64 requests of a half: request i, lane l -> LDS granule p = 64 i + l = (row p / G, slot p % G), which holds source column
 * (slot - rot(row)) mod G.  With 64 % G == 0: row = i (64 / G) + l / G and rot(row) = 4 i + rot(l / G), so a lane keeps two small
 * tile-independent values (its row within a request, its column for i = 0) and forms the offset of request i with two instructions */
template <int G>
struct ResLane { int rowl, c0; };
template <int G>
__device__ __forceinline__ ResLane<G> res_lane(int lane) {
    const int rowl = lane / G;
    return ResLane<G>{rowl, ((lane % G) - (rowl * G) / 16) & (G - 1)};
}
template <int G> __device__ __forceinline__ int res_row(const ResLane<G>& o, int i) { return i * (64 / G) + o.rowl; }
template <int G> __device__ __forceinline__ int res_col4(const ResLane<G>& o, int i) { return 4 * ((o.c0 - 4 * i) & (G - 1)); }
template <int G>
__device__ __forceinline__ void res_dma_half(float* dst, const float* src, int ld, const ResLane<G>& o, int rows) {
    const int lrow = o.rowl * ld;
#pragma unroll
    for (int i = 0; i < 32 * G / 64; ++i)
        if (res_row<G>(o, i) < rows)
            __builtin_amdgcn_global_load_lds((r_gvp_t)(src + i * (64 / G) * ld + lrow + res_col4<G>(o, i)), (r_lvp_t)(dst + 256 * i), 16, 0, 0);
}
template <int G>
__device__ __forceinline__ int res_gran(int jrow, int c) { return (jrow * G + ((c + (jrow * G) / 16) & (G - 1))) * 4; }   /* float offset */

/* v + (the value of the lane 32 away): v_permlane32_swap instead of a trip through the LDS crossbar (ds_bpermute).  Inline asm: hipcc 7.2's
 * __builtin_amdgcn_permlane32_swap returns the first result for both elements; s_nop 1 = the VALU-write -> permlane-read wait states */
__device__ __forceinline__ float res_half_sum(float v) {
    float l0 = v, l1 = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(l0), "+v"(l1));
    return l0 + l1;
}

#ifdef BGK_RES_SHFL
#define BGK_RES_SUM(v) ((v) + __shfl_xor((v), 32))
#else
#define BGK_RES_SUM(v) res_half_sum(v)
#endif

template <int N> __device__ __forceinline__ void res_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

#ifndef BGK_AFF_TS
#define BGK_AFF_TS 0           /* profiling build (tools/r04_cfg2_ts.py): lane 0 of every wave overwrites the first floats of its tile's first output row with
                                * s_memtime stamps at the phase boundaries */
#endif
#if BGK_AFF_TS
#define AFF_TS(k) ts[k] = (unsigned)__builtin_amdgcn_s_memtime()
#else
#define AFF_TS(k) do { } while (0)
#endif

/* AS, AT: the hidden activations of the shift / scale network (1 SiLU, 2 ReLU, 3 Tanh) when both exist and run as one software
 * pipeline (res_two_nets_t; one kernel instance per pair: dispatched inside the kernel the pipelines spill); AS = 0: the networks
 * one after the other, activations at run time */
template <int RW, int G, int AS, int AT>
__global__ __launch_bounds__(RW * 64, 1) void coupling_affine_resident_dma_kernel(FusedAffArgs a, ResOff os, ResOff ot, int n16_s0, int n16_s1, int n16_s2,
                                                                                  int n16_t0, int n16_t1, int n16_t2, int w16) {
    constexpr int HT = RES_HT, OT = 1, NQ = 32 * G / 64;          /* NQ: DMA requests (and store instructions) per half */
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");                 /* MODE.FP16_OVFL: f16 conversions saturate at +-65504 */
    extern __shared__ __attribute__((aligned(16))) r_u32x4 s_w[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* s_c = reinterpret_cast<float*>(s_w + w16) + wave * (2 * 32 * G * 4);       /* conditioner half, then the y half */
    float* s_y = s_c + 32 * G * 4;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile_first = (int64_t)blockIdx.x * RW + wave, tile_step = (int64_t)gridDim.x * RW;
    const ResLane<G> ol = res_lane<G>(lane);
    const int ldc = (int)a.ldc, ldy = (int)a.ldy, ldo = (int)a.ldo;
    if (tile_first < n_tiles) {                          /* the first tile's halves travel while the operands are staged */
        const int64_t b0 = tile_first * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
        res_dma_half<G>(s_c, a.cond + b0 * a.ldc, ldc, ol, rows);
        res_dma_half<G>(s_y, a.y + b0 * a.ldy, ldy, ol, rows);
    }
    {
        const r_u32x4* src[6] = {(const r_u32x4*)a.shift.A0, (const r_u32x4*)a.shift.A1, (const r_u32x4*)a.shift.A2,
                                 (const r_u32x4*)a.scale.A0, (const r_u32x4*)a.scale.A1, (const r_u32x4*)a.scale.A2};
        const int cnt[6] = {n16_s0, n16_s1, n16_s2, n16_t0, n16_t1, n16_t2};
        const int off[6] = {os.a0, os.a1, os.a2, ot.a0, ot.a1, ot.a2};
#pragma unroll
        for (int q = 0; q < 6; ++q)
            for (int i = tid; i < cnt[q]; i += RW * 64) s_w[off[q] + i] = src[q][i];
    }
    __syncthreads();
    const int d = a.d, n_in = a.d_c;
    const float alpha = a.has_scale ? bgk_expf(a.log_alpha[0]) : 0.0f;
    for (int64_t tile = tile_first; tile < n_tiles; tile += tile_step) {
        const int64_t b0 = tile * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
        const bool more = tile + tile_step < n_tiles;
        const int64_t b1 = b0 + tile_step * 32;
        const int rows1 = more ? (int)((a.B - b1) < 32 ? (a.B - b1) : 32) : 0;
#if BGK_AFF_TS
        unsigned ts[8];
#endif
        AFF_TS(0);
        res_wait_vm<NQ>();                               /* C_t has landed (Y_t may still be in flight) */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        AFF_TS(1);
        float dl_old = 0.0f;
        if (a.accumulate && hh == 0 && j < rows) dl_old = a.dlogp[b0 + j];

        /* ---- layer 0 of both networks: KR = G / 4 k-steps of real features + one bias-only k-step (the constant-1 feature is feature
         * 4 G = 16 KR: element 0 of the lower half-wave).  All B granules of the lane's row are read up front; the A fragments of the
         * next k-step are requested before the MFMAs of the current one. ---- */
        constexpr int KR = G / 4;
        h2_f32x16 hs[HT], ht[HT];
        float4 tb[KR][2];
#pragma unroll
        for (int s = 0; s < KR; ++s) {
            tb[s][0] = *reinterpret_cast<const float4*>(s_c + res_gran<G>(j, 4 * s + 2 * hh));
            tb[s][1] = *reinterpret_cast<const float4*>(s_c + res_gran<G>(j, 4 * s + 2 * hh + 1));
        }
        h2_h16x8 bh[KR + 1], bl[KR + 1];
#pragma unroll
        for (int s = 0; s < KR; ++s) {
            const float v[8] = {tb[s][0].x, tb[s][0].y, tb[s][0].z, tb[s][0].w, tb[s][1].x, tb[s][1].y, tb[s][1].z, tb[s][1].w};
            r_u32x4 uh, ul;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {               /* (MODE.FP16_OVFL: the conversions saturate, no clamp instructions) */
                unsigned hi, lo;
                r_split_pair(v[e], v[e + 1], hi, lo);
                uh[e >> 1] = hi; ul[e >> 1] = lo;
            }
            bh[s] = __builtin_bit_cast(h2_h16x8, uh); bl[s] = __builtin_bit_cast(h2_h16x8, ul);
        }
        bh[KR] = h2_h16x8{hh == 0 ? (_Float16)1.0f : (_Float16)0.0f, 0, 0, 0, 0, 0, 0, 0};
        bl[KR] = h2_h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        auto layer0 = [&](h2_f32x16 (&h)[HT], const r_u32x4* W) {
            RA<HT> fr[2];
            ra_load<HT>(fr[0], W, 0, lane);
#pragma unroll
            for (int s = 0; s <= KR; ++s) {
                if (s < KR) ra_load<HT>(fr[(s + 1) & 1], W, s + 1, lane);
#ifndef BGK_AFF_NOPIN
                __builtin_amdgcn_sched_barrier(0);       /* (see ra_gemm_hidden_pre) */
#endif
                if (s == 0) ra_mfma3<HT, true, false>(h, fr[s & 1], bh[s], bl[s]);
                else if (s == KR) ra_mfma3<HT, false, true>(h, fr[s & 1], bh[s], bl[s]);
                else ra_mfma3<HT, false, false>(h, fr[s & 1], bh[s], bl[s]);
#ifndef BGK_AFF_NOPIN
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        };
        if (a.has_shift) layer0(hs, s_w + os.a0);
        if (a.has_scale) layer0(ht, s_w + ot.a0);
        /* the conditioner half is consumed (its LDS reads have returned into the operand registers above): request the next tile's */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (more) res_dma_half<G>(s_c, a.cond + b1 * a.ldc, ldc, ol, rows1);
        AFF_TS(2);

        h2_f32x16 mu[OT], sr[OT];
        if constexpr (AS != 0) {
            res_two_nets_t<AS, AT, HT, OT>(mu, sr, hs, ht, a.shift, a.scale, s_w, os, ot, lane);
        } else {
            if (a.has_shift) res_net_tail<HT, OT>(mu, hs, a.shift, s_w, os, lane);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) mu[0][r] = 0.0f;
            }
            AFF_TS(3);
            if (a.has_scale) res_net_tail<HT, OT>(sr, ht, a.scale, s_w, ot, lane);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sr[0][r] = 0.0f;
            }
        }

        AFF_TS(4);
        float lsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float l0 = (a.has_scale && h2_row(0, r, hh) < d) ? r_tanh_out(sr[0][r] * a.scale.c2) * alpha : 0.0f;
            const float l1 = (a.has_scale && h2_row(0, r + 1, hh) < d) ? r_tanh_out(sr[0][r + 1] * a.scale.c2) * alpha : 0.0f;
            sr[0][r] = l0; sr[0][r + 1] = l1;
            lsum += l0;
            lsum += l1;
        }
        float total = BGK_RES_SUM(lsum);
        if (a.preserve_volume && a.has_scale) {
            const float mean = total / (float)d;
            lsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ls = h2_row(0, r, hh) < d ? sr[0][r] - mean : 0.0f;
                sr[0][r] = ls;
                lsum += ls;
            }
            total = BGK_RES_SUM(lsum);
        }
        /* ---- epilogue on the y half in LDS: lane (j, hh) owns dims 8 q + 4 hh .. + 3 of sample j = granule 2 q + hh ---- */
        AFF_TS(5);
        if (more) res_wait_vm<NQ>(); else res_wait_vm<0>();           /* Y_t and the dlogp value have arrived (C_t+1 may be in flight) */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (8 * q + 4 * hh >= d) continue;
            float* gp = s_y + res_gran<G>(j, 2 * q + hh);
            const float4 t4 = *reinterpret_cast<const float4*>(gp);
            const float v[4] = {t4.x, t4.y, t4.z, t4.w};
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = 4 * q + u;
                const float mm = a.has_shift ? mu[0][r] * a.shift.c2 : 0.0f;
                const float ls = sr[0][r];
                const float sg = __builtin_amdgcn_exp2f((a.inverse ? -ls : ls) * 1.44269504088896341f);     /* |ls| <= exp(log_alpha): 1 ulp */
                float t = a.inverse ? sg * (v[u] - mm) : sg * v[u] + mm;
                if (a.is_circular) { t = t - __builtin_truncf(t); if (t < 0.0f) t = t + 1.0f; }
                o[u] = t;
            }
            *reinterpret_cast<float4*>(gp) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        /* the finished half leaves as whole 128-byte row pieces: granule p = 64 i + lane of the LDS image -> its row and source column */
        float4 og[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) og[i] = *reinterpret_cast<const float4*>(s_y + 256 * i + 4 * lane);
        float* out_t = a.out + b0 * a.ldo;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int c4 = res_col4<G>(ol, i);
            if (res_row<G>(ol, i) < rows && c4 < d) *reinterpret_cast<float4*>(out_t + i * (64 / G) * ldo + ol.rowl * ldo + c4) = og[i];
        }
        if (hh == 0 && j < rows) {
            const float dl = a.inverse ? -total : total;
            a.dlogp[b0 + j] = dl_old + dl;
        }
        /* the y half is in registers / on its way out: request the next tile's (the LDS reads above have returned: og is live) */
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (more) res_dma_half<G>(s_y, a.y + b1 * a.ldy, ldy, ol, rows1);
#if BGK_AFF_TS
        AFF_TS(6);
        if (lane == 0) for (int k = 0; k < 7; ++k) reinterpret_cast<unsigned*>(a.out + b0 * a.ldo)[k] = ts[k];
#endif
    }
}

/* 16-byte blocks of one packed layer: S k-steps x NT tiles x {hi, lo} + NT bias blocks (bias: hidden / output layers only) */
inline int res_blocks16(int S, int NT, bool bias) { return (S * NT * 2 + (bias ? NT : 0)) * 64; }

}  // namespace

static int affine_dense_launch(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                               const void* sA0, const void* sA1, const void* sA1b, const void* sA2,
                               float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                               const void* tA0, const void* tA1, const void* tA1b, const void* tA2,
                               float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                               int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                               int32_t is_circular, int32_t inverse,
                               const float* y, int64_t ldy, int64_t B, int32_t d,
                               float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream, const BgkCondSegs* segs = nullptr) {
    const bool multi = segs && segs->n > 1;       /* several conditioning tensors: the event-threaded width-128 kernel only */
    if (segs && segs->n >= 1) { cond = segs->ptr[0]; ldc = segs->ld[0]; }
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
    a.shift = AffNet{(const uint4*)sA0, (const uint4*)sA1, (const uint4*)sA2, sc0, sc1, sc2, s_act, (const uint4*)sA1b, sc1b};
    a.scale = AffNet{(const uint4*)tA0, (const uint4*)tA1, (const uint4*)tA2, tc0, tc1, tc2, t_act, (const uint4*)tA1b, tc1b};
    a.has_shift = has_shift; a.has_scale = has_scale;
    a.log_alpha = log_alpha; a.preserve_volume = preserve_volume; a.is_circular = is_circular; a.inverse = inverse;
    a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.lds_per_wave = 16 * a.S0 * ASROW;
    a.vec4 = (ldy % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)y | (uintptr_t)out) % 16 == 0);
    const size_t shmem = sizeof(float) * (size_t)AW * a.lds_per_wave;
    const int64_t n_wg = ((B + 31) / 32 + AW - 1) / AW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_coupling_affine_dense_h2: batch too large for one launch");
    a.cvec4 = (ldc % 4 == 0) && ((uintptr_t)cond % 16 == 0);
    const int OT = (d + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (multi && !(hidden == 128 && bgk_affine_variant == 2)) return BGK_EUNSUPPORTED;
    if (hidden == 64 && bgk_affine_variant == 2 && !sA1b && !tA1b) {
        /* weight-resident kernel: operands of both networks in LDS */
        const int n0 = res_blocks16(a.S0, RES_HT, false), n1 = res_blocks16(2 * RES_HT, RES_HT, true), n2 = res_blocks16(2 * RES_HT, OT, true);
        ResOff os{0, 0, 0}, ot{0, 0, 0};
        int top = 0;
        if (has_shift) { os = ResOff{top, top + n0, top + n0 + n1}; top += n0 + n1 + n2; }
        if (has_scale) { ot = ResOff{top, top + n0, top + n0 + n1}; top += n0 + n1 + n2; }
        const size_t res_shmem = (size_t)top * 16;
        /* both halves of the stack's row as DMA-staged LDS tiles, next tile prefetched (cfg 2's shape) */
        constexpr int DRW = 8;
        const size_t dma_shmem = res_shmem + (size_t)DRW * 2 * 32 * 8 * 16;
        if (bgk_affine_variant == 2 && d_c == 32 && d == 32 && !periodic && a.cvec4 && a.vec4 && (ldc % 4 == 0) && dma_shmem <= 160 * 1024
            && ldc < (1 << 20) && ldy < (1 << 20) && ldo < (1 << 20) && a.S0 == 3
            && !getenv("BGK_AFFINE_NO_DMA")) {
            const int64_t n_tiles = (B + 31) / 32;
            int64_t grid = (n_tiles + DRW - 1) / DRW;
            if (grid > 256) grid = 256;
            const int c_s = has_shift, c_t = has_scale;
            int pact = 0;                                     /* 16 AS + AT of the pipelined instances: equal activations, or the RealNVP pair ReLU / Tanh */
#if BGK_AFF_PIPE2
            if (has_shift && has_scale && s_act >= 1 && s_act <= 3 && (s_act == t_act || (s_act == 2 && t_act == 3)) && !getenv("BGK_AFFINE_NO_PIPE2"))
                pact = 16 * s_act + t_act;
#endif
#define BGK_LAUNCH_DMA(SA, TA) do { auto K = coupling_affine_resident_dma_kernel<DRW, 8, SA, TA>; \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            hipLaunchKernelGGL(K, dim3((int)grid), dim3(DRW * 64), dma_shmem, st, a, os, ot, c_s * n0, c_s * n1, c_s * n2, c_t * n0, c_t * n1, c_t * n2, top); } while (0)
            if (pact == 0x11) BGK_LAUNCH_DMA(1, 1); else if (pact == 0x22) BGK_LAUNCH_DMA(2, 2); else if (pact == 0x33) BGK_LAUNCH_DMA(3, 3);
            else if (pact == 0x23) BGK_LAUNCH_DMA(2, 3); else BGK_LAUNCH_DMA(0, 0);
#undef BGK_LAUNCH_DMA
            return bgk_launch_status("bgk_coupling_affine_dense_h2");
        }
        if (res_shmem <= 150 * 1024) {
            const int RW = OT == 1 ? 12 : 8;        /* waves per workgroup: 3 per SIMD where the kernel fits 168 VGPRs, else 2 */
            const int64_t n_tiles = (B + 31) / 32;
            const int per_cu = (int)((160 * 1024) / res_shmem) > 2 ? 2 : (int)((160 * 1024) / res_shmem);
            int64_t grid = (n_tiles + RW - 1) / RW;
            if (grid > 256 * (per_cu < 1 ? 1 : per_cu)) grid = 256 * (per_cu < 1 ? 1 : per_cu);
            const int c_s = has_shift, c_t = has_scale;
#define BGK_LAUNCH_K(K) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                             hipLaunchKernelGGL(K, dim3((int)grid), dim3(RW * 64), res_shmem, st, a, os, ot, c_s * n0, c_s * n1, c_s * n2, c_t * n0, c_t * n1, c_t * n2); } while (0)
            if (OT == 1) BGK_LAUNCH_K((coupling_affine_resident_kernel<1, 12>));
            else if (OT == 2) BGK_LAUNCH_K((coupling_affine_resident_kernel<2, 8>));
            else BGK_LAUNCH_K((coupling_affine_resident_kernel<3, 8>));
#undef BGK_LAUNCH_K
#undef BGK_LAUNCH_R
            return bgk_launch_status("bgk_coupling_affine_dense_h2");
        }
    }
    const int v2_tile = 16 * a.S0 * ASROW > d * ASROW ? 16 * a.S0 * ASROW : d * ASROW;       /* floats per wave: conditioner / shift tile + y tile */
    if (hidden == 128 && bgk_affine_variant == 2 && (size_t)(v2_tile + d * ASROW) * 16 <= 80 * 1024
        && ldc < (1 << 24) && ldy < (1 << 24) && ldo < (1 << 24)) {
        /* width 128: MFMA events threaded through the activation code (bgk_fused2.hip); it declines activation pairs it has no instance for */
        const int st2 = bgk_launch_affine_dense_v2(cond, ldc, d_c, periodic, sA0, sA1, sA1b, sA2, sc0, sc1, sc1b, sc2, s_act,
                                                   tA0, tA1, tA1b, tA2, tc0, tc1, tc1b, tc2, t_act, log_alpha, preserve_volume, is_circular,
                                                   inverse, y, ldy, B, d, out, ldo, dlogp, accumulate, stream, segs);
        if (st2 != BGK_EUNSUPPORTED) return st2;
    }
    if (multi) return BGK_EUNSUPPORTED;
#define BGK_LAUNCH(H, O) hipLaunchKernelGGL((coupling_affine_dense_kernel<H, O>), dim3((int)n_wg), dim3(AW * 64), shmem, st, a)
    if (hidden == 64) { if (OT == 1) BGK_LAUNCH(2, 1); else if (OT == 2) BGK_LAUNCH(2, 2); else BGK_LAUNCH(2, 3); }
    else { if (OT == 1) BGK_LAUNCH(4, 1); else if (OT == 2) BGK_LAUNCH(4, 2); else BGK_LAUNCH(4, 3); }
#undef BGK_LAUNCH
    return bgk_launch_status("bgk_coupling_affine_dense_h2");
}

extern "C" int bgk_coupling_affine_dense_h2(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                            const void* sA0, const void* sA1, const void* sA2,
                                            float sc0, float sc1, float sc2, int32_t s_act,
                                            const void* tA0, const void* tA1, const void* tA2,
                                            float tc0, float tc1, float tc2, int32_t t_act,
                                            int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                            int32_t is_circular, int32_t inverse,
                                            const float* y, int64_t ldy, int64_t B, int32_t d,
                                            float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    return affine_dense_launch(cond, ldc, d_c, periodic, sA0, sA1, nullptr, sA2, sc0, sc1, 1.0f, sc2, s_act,
                               tA0, tA1, nullptr, tA2, tc0, tc1, 1.0f, tc2, t_act, hidden, log_alpha, preserve_volume, is_circular, inverse,
                               y, ldy, B, d, out, ldo, dlogp, accumulate, stream);
}

extern "C" int bgk_coupling_affine_dense_deep(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                              const void* sA0, const void* sA1, const void* sA2, float sc0, const float* sc1s, float sc2, int32_t s_act,
                                              const void* tA0, const void* tA1, const void* tA2, float tc0, const float* tc1s, float tc2, int32_t t_act,
                                              int32_t n_hidden, int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                              int32_t is_circular, int32_t inverse,
                                              const float* y, int64_t ldy, int64_t B, int32_t d,
                                              float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    const char* what = "bgk_coupling_affine_dense_deep";
    BGK_CHECK_ARG(cond && y && out && dlogp, "%s: null pointer", what);
    BGK_CHECK_ARG(B > 0 && d > 0 && d_c > 0, "%s: bad sizes", what);
    const int has_shift = sA0 != nullptr, has_scale = tA0 != nullptr;
    BGK_CHECK_ARG(has_shift || has_scale, "%s: no conditioner network", what);
    BGK_CHECK_ARG(!has_shift || (sA2 && (n_hidden == 1 || (sA1 && sc1s))), "%s: incomplete shift network", what);
    BGK_CHECK_ARG(!has_scale || (tA2 && log_alpha && (n_hidden == 1 || (tA1 && tc1s))), "%s: incomplete scale network", what);
    BGK_CHECK_ARG(!(has_scale && is_circular), "Scaling is not compatible with periodicity.");
    const int n_in = periodic ? 2 * d_c : d_c;
    if (n_hidden < 1 || n_hidden > AFF_DEEP_MAX_HH + 1 || (hidden != 64 && hidden != 128) || d > 96 || n_in > 127
        || s_act < 0 || s_act > 3 || t_act < 0 || t_act > 3) {
        bgk_set_error("%s: 1 .. %d hidden layers of width 64 | 128, d <= 96, <= 127 input features are fused (got n_hidden=%d hidden=%d d=%d n_in=%d)",
                      what, AFF_DEEP_MAX_HH + 1, n_hidden, hidden, d, n_in);
        return BGK_EUNSUPPORTED;
    }
    DeepAffArgs da;
    FusedAffArgs& a = da.a;
    a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic; a.S0 = (n_in + 1 + 15) / 16;
    a.shift = AffNet{(const uint4*)sA0, (const uint4*)sA1, (const uint4*)sA2, sc0, 1.0f, sc2, s_act, nullptr, 1.0f};
    a.scale = AffNet{(const uint4*)tA0, (const uint4*)tA1, (const uint4*)tA2, tc0, 1.0f, tc2, t_act, nullptr, 1.0f};
    a.has_shift = has_shift; a.has_scale = has_scale;
    a.log_alpha = log_alpha; a.preserve_volume = preserve_volume; a.is_circular = is_circular; a.inverse = inverse;
    a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.lds_per_wave = 16 * a.S0 * ASROW;
    a.vec4 = (ldy % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)y | (uintptr_t)out) % 16 == 0);
    a.cvec4 = (ldc % 4 == 0) && ((uintptr_t)cond % 16 == 0);
    da.n_hh = n_hidden - 1;
    for (int l = 0; l < AFF_DEEP_MAX_HH; ++l) {
        da.sc1s[l] = (has_shift && l < da.n_hh) ? sc1s[l] : 1.0f;
        da.tc1s[l] = (has_scale && l < da.n_hh) ? tc1s[l] : 1.0f;
    }
    const size_t shmem = sizeof(float) * (size_t)AW * a.lds_per_wave;
    const int64_t n_wg = ((B + 31) / 32 + AW - 1) / AW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "%s: batch too large for one launch", what);
    const int OT = (d + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH(H, O) hipLaunchKernelGGL((coupling_affine_deep_kernel<H, O>), dim3((int)n_wg), dim3(AW * 64), shmem, st, da)
    if (hidden == 64) { if (OT == 1) BGK_LAUNCH(2, 1); else if (OT == 2) BGK_LAUNCH(2, 2); else BGK_LAUNCH(2, 3); }
    else { if (OT == 1) BGK_LAUNCH(4, 1); else if (OT == 2) BGK_LAUNCH(4, 2); else BGK_LAUNCH(4, 3); }
#undef BGK_LAUNCH
    return bgk_launch_status(what);
}

extern "C" int bgk_coupling_affine_dense_h3(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                            const void* sA0, const void* sA1, const void* sA1b, const void* sA2,
                                            float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                                            const void* tA0, const void* tA1, const void* tA1b, const void* tA2,
                                            float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                                            int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                            int32_t is_circular, int32_t inverse,
                                            const float* y, int64_t ldy, int64_t B, int32_t d,
                                            float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG((sA0 == nullptr || sA1b) && (tA0 == nullptr || tA1b), "bgk_coupling_affine_dense_h3: missing third hidden layer");
    return affine_dense_launch(cond, ldc, d_c, periodic, sA0, sA1, sA1b, sA2, sc0, sc1, sc1b, sc2, s_act,
                               tA0, tA1, tA1b, tA2, tc0, tc1, tc1b, tc2, t_act, hidden, log_alpha, preserve_volume, is_circular, inverse,
                               y, ldy, B, d, out, ldo, dlogp, accumulate, stream);
}

/* the same layers with the conditioning input given as 1..BGK_MAX_COND tensors [B, width_i] that stand for their concatenation along
 * the feature axis (CouplingFlow's torch.cat over cond_indices, nn/flow/coupling.py:162-165, without the copy): cond / ldc / width
 * are HOST arrays of n_cond device pointers / row strides / widths */
static int affine_mc(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, BgkCondSegs& segs, int& d_c) {
    BGK_CHECK_ARG(cond && ldc && width && n_cond >= 1 && n_cond <= BGK_MAX_COND, "bgk_coupling_affine_dense_*_mc: 1..%d conditioning tensors", BGK_MAX_COND);
    d_c = 0;
    for (int i = 0; i < n_cond; ++i) { segs.ptr[i] = cond[i]; segs.ld[i] = ldc[i]; segs.w[i] = width[i]; d_c += width[i]; }
    segs.n = n_cond;
    return 0;
}

extern "C" int bgk_coupling_affine_dense_h2_mc(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                               const void* sA0, const void* sA1, const void* sA2,
                                               float sc0, float sc1, float sc2, int32_t s_act,
                                               const void* tA0, const void* tA1, const void* tA2,
                                               float tc0, float tc1, float tc2, int32_t t_act,
                                               int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                               int32_t is_circular, int32_t inverse,
                                               const float* y, int64_t ldy, int64_t B, int32_t d,
                                               float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BgkCondSegs segs{};
    int d_c = 0;
    const int st = affine_mc(cond, ldc, width, n_cond, segs, d_c);
    if (st) return st;
    return affine_dense_launch(cond[0], ldc[0], d_c, periodic, sA0, sA1, nullptr, sA2, sc0, sc1, 1.0f, sc2, s_act,
                               tA0, tA1, nullptr, tA2, tc0, tc1, 1.0f, tc2, t_act, hidden, log_alpha, preserve_volume, is_circular, inverse,
                               y, ldy, B, d, out, ldo, dlogp, accumulate, stream, &segs);
}

extern "C" int bgk_coupling_affine_dense_h3_mc(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                               const void* sA0, const void* sA1, const void* sA1b, const void* sA2,
                                               float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                                               const void* tA0, const void* tA1, const void* tA1b, const void* tA2,
                                               float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                                               int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                               int32_t is_circular, int32_t inverse,
                                               const float* y, int64_t ldy, int64_t B, int32_t d,
                                               float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG((sA0 == nullptr || sA1b) && (tA0 == nullptr || tA1b), "bgk_coupling_affine_dense_h3_mc: missing third hidden layer");
    BgkCondSegs segs{};
    int d_c = 0;
    const int st = affine_mc(cond, ldc, width, n_cond, segs, d_c);
    if (st) return st;
    return affine_dense_launch(cond[0], ldc[0], d_c, periodic, sA0, sA1, sA1b, sA2, sc0, sc1, sc1b, sc2, s_act,
                               tA0, tA1, tA1b, tA2, tc0, tc1, tc1b, tc2, t_act, hidden, log_alpha, preserve_volume, is_circular, inverse,
                               y, ldy, B, d, out, ldo, dlogp, accumulate, stream, &segs);
}
