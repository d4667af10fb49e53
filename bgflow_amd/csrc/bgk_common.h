/* bgk_common.h -- shared host/device helpers of libbgflow_amd (gfx950 only). */
#ifndef BGK_COMMON_H
#define BGK_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bgflow_amd.h"
#include "bgk_detmath.h"
#include "bgk_detmath_pk.h"

#define BGK_WAVE 64

/* thread-local last-error string (host) */
void bgk_set_error(const char* fmt, ...);

#define BGK_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            bgk_set_error(__VA_ARGS__);          \
            return BGK_EINVAL;                   \
        }                                        \
    } while (0)

static inline int bgk_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        bgk_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

/* scalar spline settings, converted from the python doubles exactly like torch applies python
 * scalars to f32 tensors (and like oracle/bgo_impl.h does) */
struct BgkRqsCfg {
    float left, right, bottom, top;
    float xspan, yspan;
    float min_w, min_h, min_d;
    float w_scale, h_scale;
    float beta;
};

static inline BgkRqsCfg bgk_make_rqs_cfg(double left, double right, double bottom, double top,
                                         double min_w, double min_h, double min_d,
                                         int identity_init, int K) {
    BgkRqsCfg c;
    c.left = (float)left; c.right = (float)right; c.bottom = (float)bottom; c.top = (float)top;
    c.xspan = (float)(right - left); c.yspan = (float)(top - bottom);
    c.min_w = (float)min_w; c.min_h = (float)min_h; c.min_d = (float)min_d;
    c.w_scale = (float)(1.0 - min_w * K);
    c.h_scale = (float)(1.0 - min_h * K);
    c.beta = (float)(identity_init ? (0.6931471805599453 / (1.0 - min_d)) : 1.0);
    return c;
}

/* compensated running sum: with more than 64 bins the plain f32 running sums of the knots (K terms of ~1/K) lose ~K/2 ulp of the
 * knot position, i.e. a relative error ~K^2 eps of a bin's size and of the log-det -- beyond what the torch ops (pairwise sums)
 * give.  The bin counts the C oracle pins bit for bit (K <= 64) keep the oracle's plain sums. */
struct BgkKahan {
    float s, c;
    __device__ __forceinline__ void add(float v) { const float y = v - c; const float t = s + y; c = (t - s) - y; s = t; }
};
constexpr int BGK_RQS_COMP_FROM = 65;     /* runtime-K elements with at least this many bins use compensated sums */

/* One rational-quadratic spline element (device).  `pw`, `ph`, `ps` point at the K unnormalised
 * widths / heights / slopes of this (sample, dim) with element stride `st`; s_last is the slope at
 * knot K (periodic copy of s[0] or the non-circular extra slope).  Same operation order as
 * oracle/bgo_impl.h::bgo_rqs (which follows nflows): every op is a
 * separately rounded f32 op, the TU is compiled with -ffp-contract=off.
 * Returns the transformed value; *lad = per-element log|det| contribution in bgflow's sign
 * convention; *bin = bin index; *oob = 1 if x had to be clamped.  */
/* REGS = true: pw / ph / ps are per-lane register arrays (st == 1): the two bin-indexed slope reads become select
 * chains instead of dynamically indexed loads (which would force the arrays into scratch memory). */
template <int KT, bool REGS = false>
__device__ __forceinline__ float bgk_rqs_element(float x, const float* pw, const float* ph,
                                                 const float* ps, int st, float s_last, int Krt,
                                                 int inverse, const BgkRqsCfg& c, float* lad,
                                                 int* bin, int* oob) {
    const int K = KT ? KT : Krt;
    const bool comp = (KT == 0) && K >= BGK_RQS_COMP_FROM;
    /* clamp (InputOutsideDomain path, spline.py:145-155) */
    int o = (x < c.left) | (x > c.right);
    x = x < c.left ? c.left : (x > c.right ? c.right : x);
    *oob = o;
    /* searched set A (heights for bgflow-forward, widths for bgflow-inverse), other set Bq */
    const float* pa = inverse ? pw : ph;
    const float* pb = inverse ? ph : pw;
    const float minA = inverse ? c.min_w : c.min_h, minB = inverse ? c.min_h : c.min_w;
    const float scA = inverse ? c.w_scale : c.h_scale, scB = inverse ? c.h_scale : c.w_scale;
    const float spanA = inverse ? c.xspan : c.yspan, spanB = inverse ? c.yspan : c.xspan;
    const float lowA = inverse ? c.left : c.bottom, lowB = inverse ? c.bottom : c.left;
    const float highA = inverse ? c.right : c.top, highB = inverse ? c.top : c.right;

    /* ---- searched set: softmax -> min + scale*p -> cumsum -> affine -> ends; count x >= knot ----
     * (KT > 0: exps two at a time on the packed-f32 VALU, the K divisions by the common sum share one refined
     *  reciprocal -- bit-identical to the scalar exp / IEEE divide of the oracle, see bgk_detmath_pk.h) */
    float mA = pa[0];
#pragma unroll
    for (int k = 1; k < K; ++k) { float v = pa[k * st]; mA = v > mA ? v : mA; }
    float eA[KT ? KT : 1];
    float sA = 0.0f;
    if (KT) {
#pragma unroll
        for (int k = 0; k + 1 < K; k += 2) {
            bgk_f2 ee = bgk_expf2((bgk_f2){pa[k * st] - mA, pa[(k + 1) * st] - mA});
            eA[KT ? k : 0] = ee.x; eA[KT ? k + 1 : 0] = ee.y;
        }
        if (K & 1) eA[KT ? K - 1 : 0] = bgk_expf(pa[(K - 1) * st] - mA);
#pragma unroll
        for (int k = 0; k < K; ++k) sA += eA[KT ? k : 0];
    } else if (comp) {
        BgkKahan acc = {0.0f, 0.0f};
        for (int k = 0; k < K; ++k) acc.add(bgk_expf(pa[k * st] - mA));
        sA = acc.s;
    } else {
        for (int k = 0; k < K; ++k) sA += bgk_expf(pa[k * st] - mA);
    }
    const float rA = bgk_rcp_refined(sA);
    int idx = -1 + (x >= lowA ? 1 : 0);
    float lo = lowA, hi = lowA;
    {
        float cum = 0.0f;
        BgkKahan kc = {0.0f, 0.0f};
        bool hi_set = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float p = bgk_div_r(KT ? eA[KT ? k : 0] : bgk_expf(pa[k * st] - mA), sA, rA);
            p = minA + scA * p;
            if (comp) { kc.add(p); cum = kc.s; } else cum += p;
            float kn = spanA * cum + lowA;
            if (k == K - 1) kn = highA;
            float ks = (k == K - 1) ? kn + 1e-6f : kn;   /* in-place eps of nflows' searchsorted */
            bool ge = x >= ks;
            idx += ge ? 1 : 0;
            if (ge) lo = kn;
            if (!ge && !hi_set) { hi = kn; hi_set = true; }
        }
    }
    idx = idx < 0 ? 0 : idx;
    *bin = idx;
    const float a_i = lo, A_i = hi - lo;   /* knot[idx], bin size in the searched direction */

    /* ---- other set: knot[idx], knot[idx+1] ---- */
    float mB = pb[0];
#pragma unroll
    for (int k = 1; k < K; ++k) { float v = pb[k * st]; mB = v > mB ? v : mB; }
    float eB[KT ? KT : 1];
    float sB = 0.0f;
    if (KT) {
#pragma unroll
        for (int k = 0; k + 1 < K; k += 2) {
            bgk_f2 ee = bgk_expf2((bgk_f2){pb[k * st] - mB, pb[(k + 1) * st] - mB});
            eB[KT ? k : 0] = ee.x; eB[KT ? k + 1 : 0] = ee.y;
        }
        if (K & 1) eB[KT ? K - 1 : 0] = bgk_expf(pb[(K - 1) * st] - mB);
#pragma unroll
        for (int k = 0; k < K; ++k) sB += eB[KT ? k : 0];
    } else if (comp) {
        BgkKahan acc = {0.0f, 0.0f};
        for (int k = 0; k < K; ++k) acc.add(bgk_expf(pb[k * st] - mB));
        sB = acc.s;
    } else {
        for (int k = 0; k < K; ++k) sB += bgk_expf(pb[k * st] - mB);
    }
    const float rB = bgk_rcp_refined(sB);
    float b_i = lowB, b_ip1 = lowB;
    {
        float cum = 0.0f;
        BgkKahan kc = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float p = bgk_div_r(KT ? eB[KT ? k : 0] : bgk_expf(pb[k * st] - mB), sB, rB);
            p = minB + scB * p;
            if (comp) { kc.add(p); cum = kc.s; } else cum += p;
            float kn = spanB * cum + lowB;
            if (k == K - 1) kn = highB;
            if (k + 1 == idx) b_i = kn;
            if (k == idx) b_ip1 = kn;
        }
    }
    const float B_i = b_ip1 - b_i;

    /* ---- the two derivatives that are gathered ---- */
    float s_lo, s_hi;
    if constexpr (REGS && KT > 0) {
        s_lo = ps[0]; s_hi = s_last;
#pragma unroll
        for (int k = 1; k < KT; ++k) {
            s_lo = (k == idx) ? ps[k] : s_lo;
            s_hi = (k == idx + 1) ? ps[k] : s_hi;
        }
    } else {
        s_lo = ps[idx * st];
        s_hi = (idx + 1 < K) ? ps[(idx + 1) * st] : s_last;
    }
    const bgk_f2 sp = bgk_softplusf2((bgk_f2){s_lo, s_hi}, c.beta);
    float d_i = c.min_d + sp.x;
    float d_ip1 = c.min_d + sp.y;

    float cw_i, W_i, ch_i, H_i;
    if (inverse) { cw_i = a_i; W_i = A_i; ch_i = b_i; H_i = B_i; }
    else { ch_i = a_i; H_i = A_i; cw_i = b_i; W_i = B_i; }
    float delta = bgk_div_safe(H_i, W_i);
    float S = d_i + d_ip1 - 2.0f * delta;
    float outv, l;
    if (!inverse) {
        float dx = x - ch_i;
        float a = dx * S + H_i * (delta - d_i);
        float b = H_i * d_i - dx * S;
        float cc = -delta * dx;
        float disc = b * b - 4.0f * a * cc;
        float root = bgk_div_safe(2.0f * cc, -b - __builtin_sqrtf(disc));
        outv = root * W_i + cw_i;
        float t1mt = root * (1.0f - root);
        float den = delta + S * t1mt;
        float omr = 1.0f - root;
        float num = (delta * delta) * (d_ip1 * (root * root) + 2.0f * delta * t1mt + d_i * (omr * omr));
        { const bgk_f2 lg = bgk_logf2((bgk_f2){num, den}); l = -(lg.x - 2.0f * lg.y); }
    } else {
        float theta = bgk_div_safe(x - cw_i, W_i);
        float t1mt = theta * (1.0f - theta);
        float numer = H_i * (delta * (theta * theta) + d_i * t1mt);
        float den = delta + S * t1mt;
        outv = ch_i + bgk_div_safe(numer, den);
        float omt = 1.0f - theta;
        float num = (delta * delta) * (d_ip1 * (theta * theta) + 2.0f * delta * t1mt + d_i * (omt * omt));
        { const bgk_f2 lg = bgk_logf2((bgk_f2){num, den}); l = lg.x - 2.0f * lg.y; }
    }
    *lad = l;
    return outv;
}

/* bgk_energy.hip: [sum, count] float partials -> f64 [2] in a fixed order (one small launch) */
int bgk_loss_partial_reduce(const float* partial, int n_partials, double* loss_sums, void* stream);

#endif /* BGK_COMMON_H */
