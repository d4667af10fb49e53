/* bgk_detmath_pk.h -- DEVICE-ONLY fast forms of the bgk_detmath.h primitives for gfx950.
 *
 * Same results, bit for bit, as the portable definitions (and therefore as the CPU oracle):
 *   - two values per instruction with the packed-f32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 /
 *     v_pk_add_f32): every lane-element still sees exactly the same IEEE fma / mul / add;
 *   - division n / d as  v_rcp_f32 seed + one Newton step on the reciprocal + two fma refinements of the
 *     quotient: the core of the correctly-rounded sequence hipcc emits for `/`, without the
 *     v_div_scale / v_div_fmas / v_div_fixup range handling.  It returns the correctly rounded quotient
 *     whenever d, n / d and the intermediate residuals are normal numbers well inside the exponent range
 *     (|d| in [2^-100, 2^100], result normal) -- guaranteed at every call site here: softmax terms
 *     e in [e^-80, 1] over sums in [1, K]; SiLU denominators 1 + e^-x <= 1 + e^80 (bgk_expf clamps at +-80).
 *     Verified on MI355X against IEEE division on 3.3e7 random operand pairs: 0 mismatches
 *     (tools/ubench/pk_check.hip), and continuously by the bit-exact parity tests.
 * Measured: SiLU chain 1.4x faster than the scalar IEEE form (tools/ubench/pk_test.hip).
 */
#ifndef BGK_DETMATH_PK_H
#define BGK_DETMATH_PK_H

#include "bgk_detmath.h"

typedef float bgk_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bgk_f2 bgk_fma2(bgk_f2 a, bgk_f2 b, bgk_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ bgk_f2 bgk_splat2(float v) { return (bgk_f2)(v); }

/* exp of two values; identical to bgk_expf on each */
__device__ __forceinline__ bgk_f2 bgk_expf2(bgk_f2 x) {
    x.x = __builtin_amdgcn_fmed3f(x.x, -80.0f, 80.0f);   /* one-instruction clamp */
    x.y = __builtin_amdgcn_fmed3f(x.y, -80.0f, 80.0f);
    const bgk_f2 magic = bgk_splat2(12582912.0f);
    bgk_f2 t = bgk_fma2(x, bgk_splat2(1.44269504088896341f), magic);
    bgk_f2 n = t - magic;
    const int32_t nx = (int32_t)(bgk_f2u(t.x) - 0x4B400000u), ny = (int32_t)(bgk_f2u(t.y) - 0x4B400000u);
    bgk_f2 r = bgk_fma2(n, bgk_splat2(-0.693359375f), x);
    r = bgk_fma2(n, bgk_splat2(2.12194440e-4f), r);
    bgk_f2 p = bgk_splat2(1.9875691500e-4f);
    p = bgk_fma2(p, r, bgk_splat2(1.3981999507e-3f));
    p = bgk_fma2(p, r, bgk_splat2(8.3334519073e-3f));
    p = bgk_fma2(p, r, bgk_splat2(4.1665795894e-2f));
    p = bgk_fma2(p, r, bgk_splat2(1.6666665459e-1f));
    p = bgk_fma2(p, r, bgk_splat2(5.0000001201e-1f));
    bgk_f2 r2 = r * r;
    p = bgk_fma2(p, r2, r);
    p = p + bgk_splat2(1.0f);
    bgk_f2 sc;
    sc.x = bgk_u2f((uint32_t)(nx + 127) << 23);
    sc.y = bgk_u2f((uint32_t)(ny + 127) << 23);
    return p * sc;
}

__device__ __forceinline__ bgk_f2 bgk_div_r2(bgk_f2 n, bgk_f2 d, bgk_f2 r) {
    bgk_f2 q = n * r;
    bgk_f2 rem = bgk_fma2(-d, q, n);
    q = bgk_fma2(rem, r, q);
    rem = bgk_fma2(-d, q, n);
    return bgk_fma2(rem, r, q);
}
__device__ __forceinline__ bgk_f2 bgk_div_fast2(bgk_f2 n, bgk_f2 d) {
    bgk_f2 r; r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    bgk_f2 e = bgk_fma2(-d, r, bgk_splat2(1.0f));
    r = bgk_fma2(e, r, r);
    return bgk_div_r2(n, d, r);
}

/* log of two non-negative finite values; identical to bgk_logf on each (v_frexp_mant/_exp give the same
 * [0.5, 1) mantissa / exponent split as the portable bit manipulation, denormals included) */
__device__ __forceinline__ bgk_f2 bgk_logf2(bgk_f2 x) {
    bgk_f2 m;
    m.x = __builtin_amdgcn_frexp_mantf(x.x); m.y = __builtin_amdgcn_frexp_mantf(x.y);
    int32_t ex = __builtin_amdgcn_frexp_expf(x.x), ey = __builtin_amdgcn_frexp_expf(x.y);
    const bool cx = m.x < 0.707106781186547524f, cy = m.y < 0.707106781186547524f;
    ex -= cx ? 1 : 0; ey -= cy ? 1 : 0;
    m.x = cx ? m.x + m.x : m.x; m.y = cy ? m.y + m.y : m.y;
    m = m - bgk_splat2(1.0f);
    bgk_f2 z = m * m;
    bgk_f2 y = bgk_splat2(7.0376836292e-2f);
    y = bgk_fma2(y, m, bgk_splat2(-1.1514610310e-1f));
    y = bgk_fma2(y, m, bgk_splat2(1.1676998740e-1f));
    y = bgk_fma2(y, m, bgk_splat2(-1.2420140846e-1f));
    y = bgk_fma2(y, m, bgk_splat2(1.4249322787e-1f));
    y = bgk_fma2(y, m, bgk_splat2(-1.6668057665e-1f));
    y = bgk_fma2(y, m, bgk_splat2(2.0000714765e-1f));
    y = bgk_fma2(y, m, bgk_splat2(-2.4999993993e-1f));
    y = bgk_fma2(y, m, bgk_splat2(3.3333331174e-1f));
    y = y * m;
    y = y * z;
    bgk_f2 fe = (bgk_f2){(float)ex, (float)ey};
    y = bgk_fma2(fe, bgk_splat2(-2.12194440e-4f), y);
    y = bgk_fma2(z, bgk_splat2(-0.5f), y);
    bgk_f2 r = m + y;
    r = bgk_fma2(fe, bgk_splat2(0.693359375f), r);
    r.x = x.x == 0.0f ? -__builtin_inff() : r.x;
    r.y = x.y == 0.0f ? -__builtin_inff() : r.y;
    return r;
}

/* softplus of two values with a common beta; identical to bgk_softplusf on each */
__device__ __forceinline__ bgk_f2 bgk_softplusf2(bgk_f2 x, float beta) {
    bgk_f2 z = x * bgk_splat2(beta);
    bgk_f2 e = bgk_expf2(z);
    bgk_f2 u = bgk_splat2(1.0f) + e;
    bgk_f2 l1p = bgk_logf2(u) * bgk_div_fast2(e, u - bgk_splat2(1.0f));   /* garbage where u == 1, replaced below */
    l1p.x = u.x == 1.0f ? e.x : l1p.x;
    l1p.y = u.y == 1.0f ? e.y : l1p.y;
    bgk_f2 r = bgk_div_r2(l1p, bgk_splat2(beta), bgk_splat2(bgk_rcp_refined(beta)));
    r.x = z.x > 20.0f ? x.x : r.x;
    r.y = z.y > 20.0f ? x.y : r.y;
    return r;
}

/* SiLU of two values: x / (1 + exp(-x)) */
__device__ __forceinline__ bgk_f2 bgk_siluf2(bgk_f2 x) {
    return bgk_div_fast2(x, bgk_splat2(1.0f) + bgk_expf2(-x));
}

/* tanh of two values; identical to bgk_tanhf on each (both branches evaluated, the scalar form's one selected) */
__device__ __forceinline__ bgk_f2 bgk_tanhf2(bgk_f2 x) {
    bgk_f2 ax;
    ax.x = __builtin_fabsf(x.x); ax.y = __builtin_fabsf(x.y);
    const bgk_f2 e = bgk_expf2(ax + ax);
    bgk_f2 r = bgk_splat2(1.0f) - bgk_div_fast2(bgk_splat2(2.0f), e + bgk_splat2(1.0f));
    const bgk_f2 z = x * x;
    bgk_f2 p = bgk_splat2(-5.70498872745e-3f);
    p = bgk_fma2(p, z, bgk_splat2(2.06390887954e-2f));
    p = bgk_fma2(p, z, bgk_splat2(-5.37397155531e-2f));
    p = bgk_fma2(p, z, bgk_splat2(1.33314422036e-1f));
    p = bgk_fma2(p, z, bgk_splat2(-3.33332819422e-1f));
    p = p * z;
    const bgk_f2 small = bgk_fma2(p, x, x);
    bgk_f2 o;
    o.x = ax.x >= 0.625f ? (x.x < 0.0f ? -r.x : r.x) : small.x;
    o.y = ax.y >= 0.625f ? (x.y < 0.0f ? -r.y : r.y) : small.y;
    return o;
}

/* ---- hardware-transcendental forms for HIDDEN activations of the split-f16 / bf16 kernels only ----------------------
 * v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the reproducible polynomial + exactly rounded division: relative error
 * <= ~|x| 2^-24 + 2^-22, i.e. below the 22-24 bit operand representation the activation is converted to right afterwards.
 * Not used for anything that reaches an output directly (spline arithmetic, log sigma) nor in the exact-f32 kernel.
 * Cost on gfx950 (packed-f32 ops issue at half the scalar rate, transcendentals at a quarter): ~80 cycles per pair vs
 * ~170 (SiLU) / ~220 (tanh) for the reproducible forms. */
__device__ __forceinline__ bgk_f2 bgk_siluf2_fast(bgk_f2 x) {
    const bgk_f2 y = x * bgk_splat2(-1.44269504088896341f);
    bgk_f2 d;
    d.x = __builtin_amdgcn_exp2f(y.x); d.y = __builtin_amdgcn_exp2f(y.y);
    d = d + bgk_splat2(1.0f);
    bgk_f2 r;
    r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    return x * r;
}
__device__ __forceinline__ bgk_f2 bgk_tanhf2_fast(bgk_f2 x) {
    const bgk_f2 y = x * bgk_splat2(2.88539008177792681f);
    bgk_f2 d;
    d.x = __builtin_amdgcn_exp2f(y.x); d.y = __builtin_amdgcn_exp2f(y.y);
    d = d + bgk_splat2(1.0f);
    bgk_f2 r;
    r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    return bgk_fma2(r, bgk_splat2(-2.0f), bgk_splat2(1.0f));
}

#endif /* BGK_DETMATH_PK_H */
