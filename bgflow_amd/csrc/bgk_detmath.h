/* bgk_detmath.h -- deterministic single-precision transcendental primitives.
 *
 * Every function here is a fixed sequence of IEEE-754 binary32 add / mul / div / fma / compare /
 * integer-bit operations.  It compiles to the same arithmetic under gcc (host C, used by the CPU
 * oracle in oracle/) and under hipcc for gfx950 (device code of the HIP kernels), PROVIDED both
 * translation units are built with floating-point contraction disabled (-ffp-contract=off), no
 * fast-math, IEEE division/sqrt on the device (-fhip-fp32-correctly-rounded-divide-sqrt, the hipcc
 * default) and f32 denormals kept (the gfx950 default).  That is what makes the spline knots --
 * and therefore the rational-quadratic-spline BIN INDICES -- bit-identical between the MI355X
 * kernels and the CPU restatement ("Bit-exact bin indices").
 *
 * Accuracy: expf/logf <= ~1.5 ulp (Cephes single-precision minimax polynomials, FMA-evaluated).
 *
 * No libm / OCML calls; usable from C99, C++ and HIP.
 */
#ifndef BGK_DETMATH_H
#define BGK_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define BGK_FN __host__ __device__ __forceinline__
#else
#define BGK_FN static inline __attribute__((always_inline))
#endif

BGK_FN uint32_t bgk_f2u(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
BGK_FN float bgk_u2f(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }

/* exp(x), x clamped to [-80, 80] (results and the reciprocal of 1 + result stay normal with headroom, which
 * is what bgk_div_safe needs; callers only use it on softmax-shifted, softplus-thresholded, tanh or SiLU
 * arguments where the clamp is below fp32 resolution of the result's consumer). */
BGK_FN float bgk_expf(float x) {
    x = x < -80.0f ? -80.0f : x;
    x = x > 80.0f ? 80.0f : x;
    /* n = round-half-even(x * log2(e)) through the 1.5*2^23 magic constant */
    const float magic = 12582912.0f;
    float t = __builtin_fmaf(x, 1.44269504088896341f, magic);
    float n = t - magic;
    int32_t ni = (int32_t)(bgk_f2u(t) - 0x4B400000u);
    /* r = x - n*ln2 in two pieces (ln2_hi has 9 trailing zero bits -> n*ln2_hi exact) */
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = __builtin_fmaf(p, r2, r);
    p = p + 1.0f;
    float scale = bgk_u2f((uint32_t)(ni + 127) << 23);
    return p * scale;
}

/* natural log of a non-negative finite float (denormals handled; log(0) = -inf; negative
 * arguments are never passed by the callers). */
BGK_FN float bgk_logf(float x) {
    if (x == 0.0f) return -__builtin_inff();
    int32_t eadj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; eadj = -23; }
    uint32_t bits = bgk_f2u(x);
    int32_t e = (int32_t)((bits >> 23) & 0xffu) - 126 + eadj;
    float m = bgk_u2f((bits & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, m, -1.1514610310e-1f);
    y = __builtin_fmaf(y, m, 1.1676998740e-1f);
    y = __builtin_fmaf(y, m, -1.2420140846e-1f);
    y = __builtin_fmaf(y, m, 1.4249322787e-1f);
    y = __builtin_fmaf(y, m, -1.6668057665e-1f);
    y = __builtin_fmaf(y, m, 2.0000714765e-1f);
    y = __builtin_fmaf(y, m, -2.4999993993e-1f);
    y = __builtin_fmaf(y, m, 3.3333331174e-1f);
    y = y * m;
    y = y * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

/* n / d for "safe-range" operands: d, n / d and 1 / d normal numbers far from the exponent limits (true at
 * every call site that uses it).  Host (oracle): IEEE division.  Device: v_rcp_f32 seed, one Newton step on the
 * reciprocal, two fma refinements of the quotient -- the core of hipcc's correctly-rounded sequence without
 * its v_div_scale / v_div_fmas / v_div_fixup range handling; returns the same correctly rounded quotient
 * (a zero quotient always as +0; both forms agree on that)
 * (checked against IEEE division on 3.3e7 operand pairs on MI355X, tools/ubench/pk_check.hip: 0 mismatches). */
BGK_FN float bgk_rcp_refined(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
#else
    return 1.0f / d;   /* unused by the host form of bgk_div_r */
#endif
}
BGK_FN float bgk_div_r(float n, float d, float r) {
#if defined(__HIP_DEVICE_COMPILE__)
    float q = n * r;
    float rem = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(rem, r, q);
#else
    (void)r;
    return n / d + 0.0f;   /* + 0: a zero quotient is +0 like the device sequence returns it */
#endif
}
BGK_FN float bgk_div_safe(float n, float d) { return bgk_div_r(n, d, bgk_rcp_refined(d)); }

/* log(1 + e) for e >= 0 (Kahan's correction keeps full relative accuracy for tiny e) */
BGK_FN float bgk_log1pf_pos(float e) {
    float u = 1.0f + e;
    if (u == 1.0f) return e;
    return bgk_logf(u) * bgk_div_safe(e, u - 1.0f);
}

/* torch.nn.functional.softplus(x, beta, threshold=20):  x*beta > 20 ? x : log1p(exp(x*beta))/beta */
BGK_FN float bgk_softplusf(float x, float beta) {
    float z = x * beta;
    if (z > 20.0f) return x;
    return bgk_div_safe(bgk_log1pf_pos(bgk_expf(z)), beta);
}

/* SiLU  x * sigmoid(x) = x / (1 + exp(-x)) */
BGK_FN float bgk_siluf(float x) {
    return bgk_div_safe(x, 1.0f + bgk_expf(-x));
}

/* tanh (Cephes tanhf split at 0.625) */
BGK_FN float bgk_tanhf(float x) {
    float ax = x < 0.0f ? -x : x;
    if (ax >= 0.625f) {
        float e = bgk_expf(ax + ax);
        float r = 1.0f - bgk_div_safe(2.0f, e + 1.0f);   /* e <= exp(80): safe range */
        return x < 0.0f ? -r : r;
    }
    float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    p = p * z;
    return __builtin_fmaf(p, x, x);
}

/* cos(2 pi x), sin(2 pi x) for the WrapPeriodic featuriser (|x| < 2^20): exact quadrant reduction
 * in units of 1/4 turn, then the Cephes sinf / cosf minimax polynomials on [-pi/4, pi/4]. */
BGK_FN void bgk_sincos2pif(float x, float* s_out, float* c_out) {
    const float magic = 12582912.0f;
    float t = __builtin_fmaf(x, 4.0f, magic);
    float kq = t - magic;                                   /* round-half-even(4x) */
    int32_t q = (int32_t)(bgk_f2u(t) - 0x4B400000u) & 3;
    float f = __builtin_fmaf(kq, -0.25f, x);                /* exact: x - kq/4 in [-1/8, 1/8] */
    float th = f * 6.28318530717958647692f;
    float z = th * th;
    float sp = -1.9515295891e-4f;
    sp = __builtin_fmaf(sp, z, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    sp = sp * z;
    float sn = __builtin_fmaf(sp, th, th);
    float cp = 2.443315711809948e-5f;
    cp = __builtin_fmaf(cp, z, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    cp = cp * z;
    cp = cp * z;
    float cs = __builtin_fmaf(z, -0.5f, cp) + 1.0f;
    float so, co;
    if (q == 0) { so = sn; co = cs; }
    else if (q == 1) { so = cs; co = -sn; }
    else if (q == 2) { so = -sn; co = -cs; }
    else { so = -cs; co = sn; }
    *s_out = so; *c_out = co;
}

#endif /* BGK_DETMATH_H */
