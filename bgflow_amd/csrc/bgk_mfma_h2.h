/* bgk_mfma_h2.h -- split-f16 GEMM building blocks for gfx950 (device only).
 * f32 operand v = hi + lo, hi = rne_f16(v), lo = rne_f16(v - hi); a product = 3 x v_mfma_f32_32x32x16_f16
 * (lo*hi, hi*lo, hi*hi; f32 accumulate).  Accuracy: tools/ubench/split_gemm.hip (0.47 ulp32 rms of sum|a||b|).
 * Orientation D[feature, sample] = W[feature, k] * X[k, sample]: A = packed weights (host: dense.py::_pack_h2),
 * B = activations of the wave's 32 samples.  Hidden k order = accumulator layout, so a layer's output registers
 * become the next layer's B operand without data movement.
 *   packed A of a layer with NT output tiles and S k16-steps: 1 KiB blocks, block(s, m, p) = (s * NT + m) * 2 + p
 *   (p = 0 hi, 1 lo; lane l = 32 kb + i holds W'[32 m + i][k(s, kb, e)], e = 0..7), then NT bias blocks
 *   (lanes < 32: {b_hi, b_lo, 0...}).
 */
#ifndef BGK_MFMA_H2_H
#define BGK_MFMA_H2_H

#include "bgk_common.h"

typedef float h2_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h2_h16x8 __attribute__((ext_vector_type(8)));

template <int NT>
struct H2A { uint4 v[NT][2]; };

template <int NT>
__device__ __forceinline__ void h2a_load(H2A<NT>& f, const uint4* W, int s, int lane) {
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        f.v[m][0] = W[((s * NT + m) * 2 + 0) * 64 + lane];
        f.v[m][1] = W[((s * NT + m) * 2 + 1) * 64 + lane];
    }
}
template <int NT>
__device__ __forceinline__ void h2a_load_bias(H2A<NT>& f, const uint4* W, int S, int lane) {
#pragma unroll
    for (int m = 0; m < NT; ++m) f.v[m][0] = W[(S * NT * 2 + m) * 64 + lane];
}

__device__ __forceinline__ void h2_split8(const float (&v)[8], h2_h16x8& hi, h2_h16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float c = __builtin_amdgcn_fmed3f(v[e], -65000.0f, 65000.0f);
        const _Float16 h = (_Float16)c;
        hi[e] = h;
        lo[e] = (_Float16)(c - (float)h);
    }
}

/* f32 pair -> f16 hi pair + f16 lo pair in 3 instructions: v_cvt_pk_f16_f32 (RNE), then lo = f16(a - hi) with the mixed-precision
 * FMA reading hi as an f16 operand and writing one half of the destination each.
 * HAZARD (found in round 6, bgk_affine_bwd64.hip): these are VALU writes the compiler's hazard recogniser does not see (inline asm is
 * not classified as VALU), so it inserts none of the two wait states a matrix instruction needs behind a VALU write of one of its
 * source registers.  A v_mfma that reads `hi` / `lo` within two instructions of this sequence gets stale data (run-to-run different
 * results, inf where a stale lo half meets a new hi half).  Every user in this library keeps other instructions between a split and its
 * consumer (the split of k-step s + 1 sits behind the MFMAs of k-step s; operands pass through LDS and a barrier; ...): keep it that
 * way, or split with plain conversions (bgk_affine_bwd64.hip::q_split8s selects the same v_cvt_pk_f16_f32 from C). */
__device__ __forceinline__ void h2_split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a0), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
}

/* 8 values x a power-of-two scale -> hi / lo operand halves, no clamp: the caller's scale puts the largest magnitude of the
 * tensor (tile) into [2^14, 2^15), so nothing overflows and every value down to 2^-17 of that maximum keeps a normal lo part --
 * gradients of 1e-6 are represented to 22 bits like the O(1) activations of the forward (round 5; unscaled they were f16
 * subnormals).  Non-finite values stay non-finite (inf -> lo = NaN): a poisoned gradient poisons the step, which then skips itself. */
__device__ __forceinline__ void h2_split8_scaled(const float (&v)[8], float sc, h2_h16x8& hi, h2_h16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h2_split_pair(v[2 * e] * sc, v[2 * e + 1] * sc, h[e], l[e]);
    hi = __builtin_bit_cast(h2_h16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(h2_h16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

/* the power of two that brings a maximum magnitude m >= 0 into [2^14, 2^15), and its reciprocal; 1 for m = 0, subnormal or not
 * finite; exponents beyond +-100 are clamped (the reciprocal stays a normal number) */
__device__ __forceinline__ float h2_pow2_scale(float m, float& inv) {
    const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
    int sb = 268 - e;                              /* biased exponent of 2^(14 - (e - 127)) */
    sb = sb < 27 ? 27 : (sb > 227 ? 227 : sb);
    sb = (e == 0 || e == 255) ? 127 : sb;
    inv = __builtin_bit_cast(float, (unsigned)(254 - sb) << 23);
    return __builtin_bit_cast(float, (unsigned)sb << 23);
}

/* largest magnitude of a tile set over the whole wave (NaNs are ignored by the maximum; an inf gives inf) */
template <int NT>
__device__ __forceinline__ float h2_wave_absmax(const h2_f32x16 (&t)[NT]) {
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = __builtin_fmaxf(m, __builtin_fabsf(t[i][r]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off));
    return m;
}

/* dst[0] = max(dst[0], m) for non-negative floats (their bit patterns order like unsigned integers); the plain read first keeps
 * the atomic traffic to the handful of waves that actually raise the maximum */
__device__ __forceinline__ void h2_publish_absmax(float* dst, float m) {
    const unsigned mb = __builtin_bit_cast(unsigned, m);
    if (mb > *reinterpret_cast<volatile unsigned*>(dst)) atomicMax(reinterpret_cast<unsigned*>(dst), mb);
}

template <int NT>
__device__ __forceinline__ void h2_mfma3(h2_f32x16 (&out)[NT], const H2A<NT>& a, const h2_h16x8& bhi, const h2_h16x8& blo) {
#pragma unroll
    for (int m = 0; m < NT; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, a.v[m][1]), bhi, out[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < NT; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, a.v[m][0]), blo, out[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < NT; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, a.v[m][0]), bhi, out[m], 0, 0, 0);
}

/* a 32*HT-wide activation vector (accumulator layout, already activated) as B operands of 2*HT k16-steps */
template <int HT>
struct H2B { h2_h16x8 hi[2 * HT], lo[2 * HT]; };

template <int HT>
__device__ __forceinline__ void h2_make_b(H2B<HT>& b, const h2_f32x16 (&in)[HT]) {
#pragma unroll
    for (int s = 0; s < 2 * HT; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in[s >> 1][8 * (s & 1) + e];
        h2_split8(v, b.hi[s], b.lo[s]);
    }
}

/* the same with the values multiplied by a power-of-two scale first (gradient tiles: see h2_split8_scaled) */
template <int HT>
__device__ __forceinline__ void h2_make_b_scaled(H2B<HT>& b, const h2_f32x16 (&in)[HT], float sc) {
#pragma unroll
    for (int s = 0; s < 2 * HT; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in[s >> 1][8 * (s & 1) + e];
        h2_split8_scaled(v, sc, b.hi[s], b.lo[s]);
    }
}

/* out[0..NT) += W' * b + bias'  (K = 32 HT, two-deep A ring) */
template <int NT, int HT>
__device__ __forceinline__ void h2_gemm_hidden(h2_f32x16 (&out)[NT], const H2B<HT>& b, const uint4* W, int lane) {
    constexpr int S = 2 * HT;
    H2A<NT> ring[2];
    h2a_load<NT>(ring[0], W, 0, lane);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) h2a_load<NT>(ring[(s + 1) & 1], W, s + 1, lane);
        else h2a_load_bias<NT>(ring[(s + 1) & 1], W, S, lane);
        __builtin_amdgcn_sched_barrier(0);
        h2_mfma3<NT>(out, ring[s & 1], b.hi[s], b.lo[s]);
    }
    const h2_h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < NT; ++m)
        out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_h16x8, ring[S & 1].v[m][0]), one2, out[m], 0, 0, 0);
}

/* out[0..NT) += W0' * x   with x rows in LDS [16 S0][srow] (constant-1 row carries the bias column) */
template <int NT>
__device__ __forceinline__ void h2_gemm_lds(h2_f32x16 (&out)[NT], const float* s_x, int srow, int S0, const uint4* W, int lane) {
    const int j = lane & 31, hh = lane >> 5;
    for (int s = 0; s < S0; ++s) {
        H2A<NT> fr;
        h2a_load<NT>(fr, W, s, lane);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = s_x[(16 * s + 8 * hh + e) * srow + j];
        h2_h16x8 bhi, blo;
        h2_split8(v, bhi, blo);
        h2_mfma3<NT>(out, fr, bhi, blo);
    }
}

/* t = act(t * c), act: 0 identity, 1 SiLU, 2 ReLU, 3 Tanh (wave-uniform runtime switch) */
__device__ __forceinline__ void h2_act_tile(h2_f32x16& t, float c, int act) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] *= c;
    if (act == 1) {        /* hidden activations: hardware exp / rcp forms (bgk_detmath_pk.h) */
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            bgk_f2 v = bgk_siluf2_fast((bgk_f2){t[r], t[r + 1]});
            t[r] = v.x; t[r + 1] = v.y;
        }
    } else if (act == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = t[r] > 0.0f ? t[r] : 0.0f;
    } else if (act == 3) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            bgk_f2 v = bgk_tanhf2_fast((bgk_f2){t[r], t[r + 1]});
            t[r] = v.x; t[r + 1] = v.y;
        }
    }
}

/* feature row of output tile m held in accumulator register r by this lane (hh = lane >> 5) */
__device__ __forceinline__ int h2_row(int m, int r, int hh) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh; }

/* Write a [128 features x 32 samples] tile set held in accumulator layout to dst[b0 + r][0..128) as FULL rows: the direct
 * form (16-byte pieces, 32 rows per store instruction) leaves the L2 to merge eight partial writes per 128-byte line and
 * was measured at 1.6x the algorithmic HBM write traffic (rocprofv3 WRITE_SIZE); through a wave-private LDS slab
 * [32][H2_SLAB] every store instruction covers two complete 512-byte rows. */
constexpr int H2_SLAB = 132;
__device__ __forceinline__ void h2_store_rows128(const h2_f32x16 (&t)[4], float* dst, float* s_buf, int64_t b0, int rows, int lane) {
    const int j = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(s_buf + j * H2_SLAB + 32 * m + 8 * q + 4 * hh) =
                make_float4(t[m][4 * q], t[m][4 * q + 1], t[m][4 * q + 2], t[m][4 * q + 3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int i = it * 64 + lane, r = i >> 5, q4 = i & 31;
        const float4 v = *reinterpret_cast<const float4*>(s_buf + r * H2_SLAB + 4 * q4);
        if (r < rows) *reinterpret_cast<float4*>(dst + (b0 + r) * 128 + 4 * q4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* the same for the first ZT tiles only, into rows of 32 ZT floats (hidden layers of <= 32 ZT units: the arrays of an affine coupling's
 * 64-unit networks are [B, 64] -- half the bytes of the padded form); every store instruction covers 64 / (8 ZT) complete rows */
template <int ZT>
__device__ __forceinline__ void h2_store_rows(const h2_f32x16 (&t)[4], float* dst, float* s_buf, int64_t b0, int rows, int lane) {
    if constexpr (ZT == 4) { h2_store_rows128(t, dst, s_buf, b0, rows, lane); return; }
    const int j = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int m = 0; m < ZT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(s_buf + j * H2_SLAB + 32 * m + 8 * q + 4 * hh) =
                make_float4(t[m][4 * q], t[m][4 * q + 1], t[m][4 * q + 2], t[m][4 * q + 3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4 * ZT; ++it) {
        const int i = it * 64 + lane, r = i / (8 * ZT), q4 = i % (8 * ZT);
        const float4 v = *reinterpret_cast<const float4*>(s_buf + r * H2_SLAB + 4 * q4);
        if (r < rows) *reinterpret_cast<float4*>(dst + (b0 + r) * (32 * ZT) + 4 * q4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#endif /* BGK_MFMA_H2_H */
