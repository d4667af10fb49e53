/* bgk_erf.h -- single-precision erf / erfinv as branch-free (wave-uniform branch at most) polynomial forms on the hardware log2 / exp2 /
 * sqrt: shared by the fused sampling-tail / inference-head kernels (bgk_tail.hip) and the stand-alone CDF maps (bgk_cdf.hip). */
#pragma once
#include "bgk_common.h"

#ifndef BGK_LN2_F
#define BGK_LN2_F 0.693147180559945309f
#endif

/* erfinv(x), |x| < 1: M. Giles, "Approximating the erfinv function" (GPU Computing Gems 2, 2010), single-precision version:
 * w = -ln(1 - x^2); central polynomial in w - 2.5 for w < 5, tail polynomial in sqrt(w) - 3 otherwise.  Max error 3.7 ulp
 * (mean 0.8) against the f64 function over (-1, 1) (tools/erfinv_check.py).  The tail branch is taken by ~0.3 % of uniform inputs:
 * it sits behind a wave-level ballot. */
__device__ __forceinline__ float erfinv_fast(float x) {
    const float t = __builtin_fmaf(-x, x, 1.0f);                        /* (1 - x)(1 + x) up to one rounding */
    float w = -BGK_LN2_F * __builtin_amdgcn_logf(t);
    const float u = w - 2.5f;
    float p = 2.81022636e-08f;
    p = __builtin_fmaf(p, u, 3.43273939e-07f);
    p = __builtin_fmaf(p, u, -3.5233877e-06f);
    p = __builtin_fmaf(p, u, -4.39150654e-06f);
    p = __builtin_fmaf(p, u, 0.00021858087f);
    p = __builtin_fmaf(p, u, -0.00125372503f);
    p = __builtin_fmaf(p, u, -0.00417768164f);
    p = __builtin_fmaf(p, u, 0.246640727f);
    p = __builtin_fmaf(p, u, 1.50140941f);
    const bool tail = w >= 5.0f;
    if (__builtin_amdgcn_ballot_w64(tail)) {
        const float s = __builtin_amdgcn_sqrtf(w) - 3.0f;
        float q = -0.000200214257f;
        q = __builtin_fmaf(q, s, 0.000100950558f);
        q = __builtin_fmaf(q, s, 0.00134934322f);
        q = __builtin_fmaf(q, s, -0.00367342844f);
        q = __builtin_fmaf(q, s, 0.00573950773f);
        q = __builtin_fmaf(q, s, -0.0076224613f);
        q = __builtin_fmaf(q, s, 0.00943887047f);
        q = __builtin_fmaf(q, s, 1.00167406f);
        q = __builtin_fmaf(q, s, 2.83297682f);
        p = tail ? q : p;
    }
    return p * x;
}

/* erf(a): N. Juffa's single-precision form (< 1 ulp): exp-based branch for |a| > 0.9277, odd polynomial below */
__device__ __forceinline__ float erf_fast(float a) {
    const float t = __builtin_fabsf(a), s = a * a;
    float r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = __builtin_fmaf(r, s, u);
    r = __builtin_fmaf(r, t, -1.06777877e-1f);
    r = __builtin_fmaf(r, t, -6.34846687e-1f);
    r = __builtin_fmaf(r, t, -1.28717512e-1f);
    r = __builtin_fmaf(r, t, -t);
    const float big = __builtin_copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896341f), a);
    float q = -5.96761703e-4f;
    q = __builtin_fmaf(q, s, 4.99119423e-3f);
    q = __builtin_fmaf(q, s, -2.67681349e-2f);
    q = __builtin_fmaf(q, s, 1.12819925e-1f);
    q = __builtin_fmaf(q, s, -3.76125336e-1f);
    q = __builtin_fmaf(q, s, 1.28379166e-1f);
    const float small = __builtin_fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}
