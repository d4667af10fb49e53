/* bgk_rqs_vjp.h -- VJP of ONE rational-quadratic spline element (sample, dim) w.r.t. its input and its 3K (+1) unnormalised
 * parameters, for first-order losses (device only).  Same math as oracle/bgo_impl.h::bgo_rqs_backward (autograd of
 * nn/flow/transformer/spline.py:109-188 + nflows' rational_quadratic_spline).  Shared by bgk_rqs_bwd.hip (one launch per
 * transformer) and bgk_dense_bwd.hip (fused with the conditioner's input-gradient chain): identical results by construction. */
#ifndef BGK_RQS_VJP_H
#define BGK_RQS_VJP_H

#include "bgk_common.h"

/* the wave's largest gradient magnitude -> dst[0] (non-negative floats order like their bit patterns; NaNs do not take part: the
 * maximum describes the finite values the consumers scale) */
__device__ __forceinline__ void bgk_publish_absmax(float* dst, float m) {
    if (dst == nullptr) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off));
    const unsigned mb = __builtin_bit_cast(unsigned, m);
    if ((threadIdx.x & 63) == 0 && mb > *reinterpret_cast<volatile unsigned*>(dst)) atomicMax(reinterpret_cast<unsigned*>(dst), mb);
}

/* softmax probabilities p[k] and the K+1 knots of one parameter set */
template <int KT>
__device__ __forceinline__ void bgk_softmax_knots(const float (&u)[KT], float mn, float sc, float span, float low, float high,
                                                  float (&p)[KT], float (&kn)[KT + 1]) {
    float m = u[0];
#pragma unroll
    for (int k = 1; k < KT; ++k) m = u[k] > m ? u[k] : m;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) { p[k] = bgk_expf(u[k] - m); s += p[k]; }
    float c = 0.0f;
    kn[0] = low;
    const float rs = bgk_rcp_refined(s);      /* the forward's form of p / s (bgk_common.h): the correctly rounded quotient */
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        p[k] = bgk_div_r(p[k], s, rs);
        c += mn + sc * p[k];
        kn[k + 1] = span * c + low;
    }
    kn[KT] = high;
}

/* BGK_VJP_FAST: the knots on the hardware forms too, in the regrouped form of the fused forward kernels (bgk_fused2.hip::rqs_fast):
 * knot_k = low + span min (k + 1) + (span scale / sum e) prefix_k(e) -- exp2, one Newton-refined reciprocal, 7 adds + 8 fma per set
 * instead of 8 polynomial exps and 8 correctly rounded quotients (~330 fewer instructions per set; the stand-alone backward kernel
 * is VALU-bound: 1.3 k instructions per element x 4.5 M elements = 170 us of the chip's VALU time at 2^18 samples x 17 dims).
 * The knots then differ from the deterministic forward's by ~1e-7: an input within that distance of a knot may be evaluated in the
 * neighbouring bin -- the spline is C1 there, so the output gradient is continuous across the knot (the training forward
 * of the fused layers uses the same hardware forms anyway). */
#ifndef BGK_VJP_FAST
#define BGK_VJP_FAST 0
#endif
template <int KT>
__device__ __forceinline__ void bgk_softmax_knots_fast(const float (&u)[KT], float mn, float sc, float span, float low, float high,
                                                       float (&p)[KT], float (&kn)[KT + 1]) {
    float m = u[0];
#pragma unroll
    for (int k = 1; k < KT; ++k) m = __builtin_fmaxf(m, u[k]);
    const float nm = -(m * 1.44269504088896341f);
    float e[KT], c[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) e[k] = __builtin_amdgcn_exp2f(__builtin_fmaf(u[k], 1.44269504088896341f, nm));
    c[0] = e[0];
#pragma unroll
    for (int k = 1; k < KT; ++k) c[k] = c[k - 1] + e[k];
    const float r0 = __builtin_amdgcn_rcpf(c[KT - 1]);
    const float r = __builtin_fmaf(__builtin_fmaf(-c[KT - 1], r0, 1.0f), r0, r0);
    const float g = (span * sc) * r, dstep = span * mn;
    kn[0] = low;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        p[k] = e[k] * r;
        kn[k + 1] = __builtin_fmaf(c[k], g, __builtin_fmaf(dstep, (float)(k + 1), low));
    }
    kn[KT] = high;
}

/* Behind the knots (which decide the bin and are bit-identical to the forward's) nothing here is compared bit for bit with
 * anything: hardware reciprocal / exp2 / log2 / sqrt (1 ulp) instead of the correctly rounded sequences -- a third of the
 * element's instructions. */
__device__ __forceinline__ float bgk_vjp_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float bgk_vjp_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float bgk_vjp_softplus(float x, float beta, float rbeta) {
    const float z = x * beta;
    const float sp = __builtin_amdgcn_logf(1.0f + bgk_vjp_exp(z < 20.0f ? z : 20.0f)) * (0.693147180559945309f * rbeta);
    return z > 20.0f ? x : sp;
}
__device__ __forceinline__ float bgk_vjp_sigmoid(float z) {
    const float sg = bgk_vjp_rcp(1.0f + bgk_vjp_exp(z > -80.0f ? -z : 80.0f));
    return z > 20.0f ? 1.0f : sg;
}

template <int KT>
__device__ __forceinline__ float bgk_pick(const float (&a)[KT], int i) {
    /* the empty asm keeps every candidate an opaque register value: the compiler otherwise turns the select chain into an
     * indexed load from a scratch copy of the array (+ an s_waitcnt vmcnt(0) that drains every load in flight) */
    float v = a[0];
#pragma unroll
    for (int k = 1; k < KT; ++k) {
        float t = a[k];
        asm("" : "+v"(t));
        v = (i == k) ? t : v;
    }
    return v;
}

/* The bin-local part of the VJP: from the bin's knots (cw_i, cw_n | ch_i, ch_n), its two unnormalised slopes and the clamped input
 * x to the cotangents of the bin width / height W, H, of the left knots cw, ch (G_*), of the two slope PARAMETERS (g0, g1: through
 * softplus) and of the input (gx).  Shared by the register-resident element routine and the memory-walking one. */
struct BgkVjpBin { float G_W, G_cw, G_H, G_ch, g0, g1, gx; };

__device__ __forceinline__ BgkVjpBin bgk_rqs_vjp_bin(const BgkRqsCfg& c, int inverse, float x, float cw_i, float cw_n, float ch_i,
                                                    float ch_n, float s_lo, float s_hi, float gy, float gl) {
    const float rbeta = bgk_vjp_rcp(c.beta);
    const float d0 = c.min_d + bgk_vjp_softplus(s_lo, c.beta, rbeta), d1 = c.min_d + bgk_vjp_softplus(s_hi, c.beta, rbeta);
    const float W_i = cw_n - cw_i, H_i = ch_n - ch_i;
    const float iW = bgk_vjp_rcp(W_i);
    const float delta = H_i * iW, S = d0 + d1 - 2.0f * delta;
    float theta;
    if (!inverse) {
        float dx = x - ch_i;
        float qa = dx * S + H_i * (delta - d0), qb = H_i * d0 - dx * S, qc = -delta * dx;
        theta = (2.0f * qc) * bgk_vjp_rcp(-qb - __builtin_amdgcn_sqrtf(qb * qb - 4.0f * qa * qc));
    } else {
        theta = (x - cw_i) * iW;
    }
    const float t = theta * (1.0f - theta), tp = 1.0f - 2.0f * theta, omt = 1.0f - theta;
    const float N = delta * theta * theta + d0 * t, den = delta + S * t;
    const float iden = bgk_vjp_rcp(den), iden2 = iden * iden;
    const float Q = N * iden;
    const float N_th = 2.0f * delta * theta + d0 * tp, den_th = S * tp;
    const float Q_th = (N_th * den - N * den_th) * iden2;
    const float Q_de = (theta * theta * den - N * (1.0f - 2.0f * t)) * iden2;
    const float Q_d0 = (t * den - N * t) * iden2;
    const float Q_d1 = (-N * t) * iden2;
    const float M = d1 * theta * theta + 2.0f * delta * t + d0 * omt * omt;
    const float iM = bgk_vjp_rcp(M);
    const float lf_th = (2.0f * d1 * theta + 2.0f * delta * tp - 2.0f * d0 * omt) * iM - 2.0f * den_th * iden;
    const float lf_de = 2.0f * bgk_vjp_rcp(delta) + 2.0f * t * iM - 2.0f * (1.0f - 2.0f * t) * iden;
    const float lf_d0 = omt * omt * iM - 2.0f * t * iden;
    const float lf_d1 = theta * theta * iM - 2.0f * t * iden;
    float G_de, G_d0, G_d1;
    BgkVjpBin r;
    if (inverse) {
        const float G_th = gy * H_i * Q_th + gl * lf_th;
        G_de = gy * H_i * Q_de + gl * lf_de;
        G_d0 = gy * H_i * Q_d0 + gl * lf_d0;
        G_d1 = gy * H_i * Q_d1 + gl * lf_d1;
        r.G_H = gy * Q + G_de * iW;
        r.G_W = -G_de * delta * iW - G_th * theta * iW;
        r.G_ch = gy;
        r.G_cw = -G_th * iW;
        r.gx = G_th * iW;
    } else {
        const float A_th = gy * W_i - gl * lf_th;
        const float iQth = bgk_vjp_rcp(Q_th);
        const float inv = bgk_vjp_rcp(H_i) * iQth;
        G_de = -gl * lf_de - A_th * Q_de * iQth;
        G_d0 = -gl * lf_d0 - A_th * Q_d0 * iQth;
        G_d1 = -gl * lf_d1 - A_th * Q_d1 * iQth;
        r.G_H = G_de * iW - A_th * Q * inv;
        r.G_W = -G_de * delta * iW + gy * theta;
        r.G_cw = gy;
        r.G_ch = -A_th * inv;
        r.gx = A_th * inv;
    }
    const bool dead = (gy == 0.0f) & (gl == 0.0f);   /* masked-out sample: exact zeros, never 0 * inf */
    if (dead) { G_d0 = G_d1 = r.G_H = r.G_W = r.G_cw = r.G_ch = r.gx = 0.0f; }
    r.g0 = G_d0 * bgk_vjp_sigmoid(s_lo * c.beta);
    r.g1 = G_d1 * bgk_vjp_sigmoid(s_hi * c.beta);
    return r;
}

/* rw / rh / rs: the element's unnormalised widths, heights, slopes (knots 0..K-1); s_K: the slope at knot K (its own slot for a
 * non-circular dim -- has_slot -- else rs[0]); x: the spline's input (pre-clamp); gy, gl: output and log-det cotangents.
 * Out: ow / oh / os parameter gradients, g_slot (gradient of the slot; 0 without one), gx (input gradient, 0 outside the domain). */
template <int K>
__device__ __forceinline__ void bgk_rqs_vjp_element(const BgkRqsCfg& c, int inverse, const float (&rw)[K], const float (&rh)[K],
                                                    const float (&rs)[K], float s_K, bool has_slot, float x, float gy, float gl,
                                                    float (&ow)[K], float (&oh)[K], float (&os)[K], float& g_slot, float& gx_out) {
    float pw[K], ph[K], cw[K + 1], ch[K + 1];
#if BGK_VJP_FAST
    bgk_softmax_knots_fast<K>(rw, c.min_w, c.w_scale, c.xspan, c.left, c.right, pw, cw);
    bgk_softmax_knots_fast<K>(rh, c.min_h, c.h_scale, c.yspan, c.bottom, c.top, ph, ch);
#else
    bgk_softmax_knots<K>(rw, c.min_w, c.w_scale, c.xspan, c.left, c.right, pw, cw);
    bgk_softmax_knots<K>(rh, c.min_h, c.h_scale, c.yspan, c.bottom, c.top, ph, ch);
#endif
    const bool clamped = (x < c.left) | (x > c.right);
    x = x < c.left ? c.left : (x > c.right ? c.right : x);
    int idx = -1;
#pragma unroll
    for (int k = 0; k <= K; ++k) {
        float kn = inverse ? cw[k] : ch[k];
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
    const float s_lo = bgk_pick<K>(rs, idx);
    float s_hi = s_K;
#pragma unroll
    for (int k = 1; k < K; ++k) s_hi = (idx + 1 == k) ? rs[k] : s_hi;
    const BgkVjpBin b = bgk_rqs_vjp_bin(c, inverse, x, cw_i, cw_n, ch_i, ch_n, s_lo, s_hi, gy, gl);
    gx_out = clamped ? 0.0f : b.gx;
    {
        const float gA = (idx >= 1) ? (b.G_cw - b.G_W) : 0.0f, gB = (idx + 1 <= K - 1) ? b.G_W : 0.0f;
        float gp[K], dot = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) { gp[m] = c.w_scale * c.xspan * ((m < idx ? gA : 0.0f) + (m <= idx ? gB : 0.0f)); dot += pw[m] * gp[m]; }
#pragma unroll
        for (int m = 0; m < K; ++m) ow[m] = pw[m] * (gp[m] - dot);
    }
    {
        const float gA = (idx >= 1) ? (b.G_ch - b.G_H) : 0.0f, gB = (idx + 1 <= K - 1) ? b.G_H : 0.0f;
        float gp[K], dot = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) { gp[m] = c.h_scale * c.yspan * ((m < idx ? gA : 0.0f) + (m <= idx ? gB : 0.0f)); dot += ph[m] * gp[m]; }
#pragma unroll
        for (int m = 0; m < K; ++m) oh[m] = ph[m] * (gp[m] - dot);
    }
    {
        g_slot = (has_slot && hi_last) ? b.g1 : 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float g = 0.0f;
            g += (k == idx) ? b.g0 : 0.0f;
            g += (!hi_last && k == idx + 1) ? b.g1 : 0.0f;
            g += (hi_last && !has_slot && k == 0) ? b.g1 : 0.0f;
            os[k] = g;
        }
    }
}

/* The same VJP for an element whose parameters sit in LDS: u[k * st] (k = 0 .. 3 K: widths | heights | slopes | slot row; true
 * parameter = value * c2) and whose gradients take their places.  Operation for operation the arithmetic of bgk_rqs_vjp_element
 * (identical results with FAST = BGK_VJP_FAST) -- but a parameter set is in registers only while its softmax runs, the two slopes of the bin are READ at their
 * (run-time) row instead of selected out of eight registers, and the gradients are written where they are formed: ~50 live
 * registers instead of ~110 (bgk_fused2.hip::coupling_rqs_bwd_recompute_kernel holds a 64-register B operand across this).
 * `wr`: the lane writes (false: an invalid slot working on a duplicate).  Returns the largest gradient magnitude. */
template <int K, bool FAST>
__device__ __forceinline__ float bgk_rqs_vjp_element_lds(const BgkRqsCfg& c, int inverse, float* u, int st, float c2, bool has_slot, bool wr,
                                                        float x, float gy, float gl, float& gx_out) {
    float pw[K], ph[K], cw[K + 1], ch[K + 1];
    {
        float r[K];
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = u[k * st] * c2;
        if constexpr (FAST) bgk_softmax_knots_fast<K>(r, c.min_w, c.w_scale, c.xspan, c.left, c.right, pw, cw);       /* (see BGK_VJP_FAST) */
        else bgk_softmax_knots<K>(r, c.min_w, c.w_scale, c.xspan, c.left, c.right, pw, cw);
    }
    {
        float r[K];
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = u[(K + k) * st] * c2;
        if constexpr (FAST) bgk_softmax_knots_fast<K>(r, c.min_h, c.h_scale, c.yspan, c.bottom, c.top, ph, ch);
        else bgk_softmax_knots<K>(r, c.min_h, c.h_scale, c.yspan, c.bottom, c.top, ph, ch);
    }
    const bool clamped = (x < c.left) | (x > c.right);
    x = x < c.left ? c.left : (x > c.right ? c.right : x);
    int idx = -1;
#pragma unroll
    for (int k = 0; k <= K; ++k) {
        float kn = inverse ? cw[k] : ch[k];
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
    const float s_lo = u[(2 * K + idx) * st] * c2;
    const float s_hi = u[(hi_last ? (has_slot ? 3 * K : 2 * K) : 2 * K + idx + 1) * st] * c2;
    const BgkVjpBin b = bgk_rqs_vjp_bin(c, inverse, x, cw_i, cw_n, ch_i, ch_n, s_lo, s_hi, gy, gl);
    gx_out = clamped ? 0.0f : b.gx;
    float gm = 0.0f;
    {
        const float gA = (idx >= 1) ? (b.G_cw - b.G_W) : 0.0f, gB = (idx + 1 <= K - 1) ? b.G_W : 0.0f;
        float gp[K], dot = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) { gp[m] = c.w_scale * c.xspan * ((m < idx ? gA : 0.0f) + (m <= idx ? gB : 0.0f)); dot += pw[m] * gp[m]; }
#pragma unroll
        for (int m = 0; m < K; ++m) { const float o = pw[m] * (gp[m] - dot); gm = __builtin_fmaxf(gm, __builtin_fabsf(o)); if (wr) u[m * st] = o; }
    }
    {
        const float gA = (idx >= 1) ? (b.G_ch - b.G_H) : 0.0f, gB = (idx + 1 <= K - 1) ? b.G_H : 0.0f;
        float gp[K], dot = 0.0f;
#pragma unroll
        for (int m = 0; m < K; ++m) { gp[m] = c.h_scale * c.yspan * ((m < idx ? gA : 0.0f) + (m <= idx ? gB : 0.0f)); dot += ph[m] * gp[m]; }
#pragma unroll
        for (int m = 0; m < K; ++m) { const float o = ph[m] * (gp[m] - dot); gm = __builtin_fmaxf(gm, __builtin_fabsf(o)); if (wr) u[(K + m) * st] = o; }
    }
    {
        /* slopes: zeros, then the bin's two entries (0 + g: the sums of bgk_rqs_vjp_element, which turn -0 into +0) */
        const float g0 = 0.0f + b.g0, g1 = 0.0f + b.g1;
        const float g_slot = (has_slot && hi_last) ? b.g1 : 0.0f;
        gm = __builtin_fmaxf(gm, __builtin_fmaxf(__builtin_fabsf(g0), __builtin_fabsf(g1)));       /* (g1 lands in a slope row or in the slot) */
        if (wr) {
#pragma unroll
            for (int k = 0; k < K; ++k) u[(2 * K + k) * st] = 0.0f;
            u[3 * K * st] = g_slot;
            u[(2 * K + idx) * st] = g0;
            if (!hi_last) u[(2 * K + idx + 1) * st] = g1;
            else if (!has_slot) u[2 * K * st] = g1;
        }
    }
    return gm;
}

/* ---- any bin count: the element's parameters stay in memory and are walked (bgk_rqs_bwd.hip::rqs_bwd_direct_kernel) -------------
 * One softmax set in memory (u[0..K), unit stride): its maximum, the sum of its exps and -- second walk -- whatever the caller's
 * visitor wants of (k, p_k = softmax probability, knot k + 1).  The knots repeat the FORWARD's sequence operation for operation
 * (bgk_common.h::bgk_rqs_element<0>: deterministic exp, the correctly rounded quotient, plain running sums up to 64 bins,
 * compensated ones beyond), so the bin found here is the bin the forward evaluated. */
struct BgkSoftmaxSet { const float* u; float m, s, rs; };

__device__ __forceinline__ BgkSoftmaxSet bgk_softmax_set(const float* u, int K, bool comp) {
    BgkSoftmaxSet q;
    q.u = u;
    float m = u[0];
    for (int k = 1; k < K; ++k) { const float v = u[k]; m = v > m ? v : m; }
    q.m = m;
    if (comp) {
        BgkKahan acc = {0.0f, 0.0f};
        for (int k = 0; k < K; ++k) acc.add(bgk_expf(u[k] - m));
        q.s = acc.s;
    } else {
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s += bgk_expf(u[k] - m);
        q.s = s;
    }
    q.rs = bgk_rcp_refined(q.s);
    return q;
}
__device__ __forceinline__ float bgk_softmax_p(const BgkSoftmaxSet& q, int k) { return bgk_div_r(bgk_expf(q.u[k] - q.m), q.s, q.rs); }

/* what the VJP needs of one knot set around bin idx: the knots idx and idx + 1, the probability mass below the bin and the bin's own */
struct BgkKnotPair { float k_i, k_n, below, p_bin; };

/* walk the searched set: finds the bin of x (same comparisons as the forward: knot K carries the +1e-6 of nflows' searchsorted) */
__device__ __forceinline__ BgkKnotPair bgk_walk_search(const BgkSoftmaxSet& q, int K, bool comp, float mn, float sc, float span, float low,
                                                       float high, float x, int& idx_out) {
    BgkKnotPair r = {low, low, 0.0f, 0.0f};
    int idx = -1 + (x >= low ? 1 : 0);
    float cum = 0.0f;
    BgkKahan kc = {0.0f, 0.0f};
    bool hi_set = false;
    for (int k = 0; k < K; ++k) {
        const float p = bgk_softmax_p(q, k);
        const float f = mn + sc * p;
        if (comp) { kc.add(f); cum = kc.s; } else cum += f;
        float kn = span * cum + low;
        if (k == K - 1) kn = high;
        const float ks = (k == K - 1) ? kn + 1e-6f : kn;
        const bool ge = x >= ks;
        idx += ge ? 1 : 0;
        if (ge) { r.k_i = kn; r.below += p; }
        if (!ge && !hi_set) { r.k_n = kn; r.p_bin = p; hi_set = true; }
    }
    idx_out = idx < 0 ? 0 : (idx > K - 1 ? K - 1 : idx);
    return r;
}

/* walk the other set up to the given bin */
__device__ __forceinline__ BgkKnotPair bgk_walk_to(const BgkSoftmaxSet& q, int K, bool comp, float mn, float sc, float span, float low,
                                                   float high, int idx) {
    BgkKnotPair r = {low, low, 0.0f, 0.0f};
    float cum = 0.0f;
    BgkKahan kc = {0.0f, 0.0f};
    for (int k = 0; k <= idx; ++k) {
        const float p = bgk_softmax_p(q, k);
        const float f = mn + sc * p;
        if (comp) { kc.add(f); cum = kc.s; } else cum += f;
        float kn = span * cum + low;
        if (k == K - 1) kn = high;
        if (k + 1 == idx) r.k_i = kn;
        if (k == idx) { r.k_n = kn; r.p_bin = p; } else r.below += p;
    }
    return r;
}

/* gradient of the set's K unnormalised parameters, written to g[0..K): the cotangents of the bin's left knot (G_c) and size (G_s)
 * reach parameter m through knot = span * cumsum(mn + sc * softmax) -- d/du_m = p_m (gp_m - sum_j p_j gp_j) with
 * gp_m = sc * span * ([m < idx] (G_c - G_s) + [m <= idx] G_s), the interior-knot conditions of the register routine included;
 * returns the largest magnitude written */
__device__ __forceinline__ float bgk_walk_grad(const BgkSoftmaxSet& q, int K, float sc, float span, int idx, const BgkKnotPair& kp,
                                               float G_c, float G_s, float* g) {
    const float gA = (idx >= 1) ? (G_c - G_s) : 0.0f, gB = (idx + 1 <= K - 1) ? G_s : 0.0f;
    const float w = sc * span;
    const float dot = w * (gA * kp.below + gB * (kp.below + kp.p_bin));
    float amax = 0.0f;
    for (int m = 0; m < K; ++m) {
        const float gp = w * ((m < idx ? gA : 0.0f) + (m <= idx ? gB : 0.0f));
        const float v = bgk_softmax_p(q, m) * (gp - dot);
        g[m] = v;
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v));
    }
    return amax;          /* largest magnitude written */
}

#endif /* BGK_RQS_VJP_H */
