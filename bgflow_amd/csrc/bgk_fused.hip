/* bgk_fused.hip -- one-launch spline coupling layer: DenseNet conditioner on the matrix cores
 * + rational-quadratic spline epilogue.  The conditioner activations and the spline parameters
 * never touch HBM: per sample the kernel reads d_c + d floats and writes d (+1) floats.
 *
 * Two kernels share the tiling and the spline code:
 *   coupling_rqs_dense_kernel     (first half of this file)  f32-input MFMA, reproducible f32 arithmetic throughout:
 *                                 bit-identical to the CPU oracle (gemm_mode "f32");
 *   coupling_rqs_dense_h2_kernel  (second half)  conditioner GEMMs in split-f16 form (or single bf16) on the f16 matrix
 *                                 cores, hidden activations and the spline's exp / log on the hardware transcendentals
 *                                 (SpMath<true>): the shipped default, f32-class accuracy, 2.1x faster;
 *                                 also the training forward (SAVE) that writes pre-activations and spline parameters.
 * The notes below describe the first kernel.
 *
 * Roofline: MFMA (f32-input v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense peak): 2*(n_in*H0 + H0*H1 +
 * H1*NCp) flops per sample against 4*(d_c + 2d + 2) bytes -> arithmetic intensity >> the
 * 19.7 flop/B machine balance.
 *
 * Decomposition (gfx950).  A wave owns 32 samples for the whole layer; a workgroup is 4 independent
 * waves (no workgroup barrier anywhere), 2 workgroups per CU = 2 waves per SIMD.
 * Measured on MI355X (profiles/r01_*): the f32-input MFMA and the f32 VALU do NOT overlap -- time is
 * additive (chunk GEMMs alone run at ~100 % of the f32 MFMA rate; SiLU + spline VALU phases add on
 * top, whether the VALU work sits in the partner wave or is software-pipelined into the same wave's
 * MFMA stream; the BGK_PIPE_* build switches keep that experiment reproducible).  The lever left
 * is VALU instruction count, not scheduling.
 *   GEMM orientation: D[feature, sample] = W[feature, k] * X[k, sample]   (A = weights, B = data)
 *   - A fragments: weights pre-packed on the host so that one k-step of 4 output tiles is ONE
 *     coalesced 16-byte-per-lane load (1 KiB per wave, L1/L2 resident), software-prefetched
 *     PF steps ahead;
 *   - B fragments: layer 0 reads the (featurised) conditioner input from LDS; hidden layers feed the
 *     previous layer's accumulator registers straight back as B (the MFMA C/D layout gives lane
 *     (half h, sample j) the features {(r&3)+8(r>>2)+4h}: visiting k in that order needs no data
 *     movement) -- the oracle reproduces this accumulation order (oracle.mfma_k_order);
 *   - last layer: output columns packed per transformed dim (3K+1 values contiguous), processed in
 *     chunks of 128 columns = 5 dims for K = 8; a chunk goes once through wave-private LDS
 *     ([row][32 samples], conflict-free both ways) to regroup from "lane = column" to
 *     "lane = (sample, dim)" for the spline.
 * In this kernel every f32 operation is the same IEEE op in the same order as oracle/bgo_impl.h -> bit-exact
 * parity (MFMA f32 = k-ordered fma chain, one rounding per product).
 */
#include "bgk_mfma_h2.h"
#include "bgk_fused2.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef BGK_PIPE_SPLINE
#define BGK_PIPE_SPLINE 0
#endif
#ifndef BGK_PIPE_ACT
#define BGK_PIPE_ACT 0
#endif
#ifndef BGK_ABL
#define BGK_ABL 0   /* timing ablations of the split-f16 kernel (tools/ablate_h2.sh): 1 no spline, 2 no chunk GEMMs, 4 no activation,
                     * 8 no layer-2 LDS transpose, 16 A operands always from block 0 (L1-resident / hoisted: -3.5 % only, i.e. the
                     * weight stream from L2 is not what limits the GEMM phases; a 3-deep ring gave nothing either) */
#endif

constexpr int FW = 4;                 /* waves per workgroup */
constexpr int FTHREADS = FW * 64;
constexpr int HID = 128;              /* hidden width (both hidden layers) */
constexpr int KB = 8;                 /* spline bins (template constant of this kernel) */
constexpr int PPD = 3 * KB + 1;       /* packed params per dim = 25 */
constexpr int DPC = 128 / PPD;        /* dims per 128-column chunk = 5 */
constexpr int PF = 4;                 /* A-fragment prefetch distance in k-steps */
constexpr int LDS_P = 128 * 32;       /* floats: one parameter chunk [128 rows][32 samples] */
constexpr int SROW = 33;              /* padded row stride of the small per-wave tiles */

struct FusedArgs {
    const float* cond; int64_t ldc; int d_c; int periodic;
    const float4* W0; int T0;        /* layer 0: T0 = ceil(n_in/2) k-steps rounded up to x4 (+1 bias step) */
    const float4* W1;                /* 64 + 1 steps */
    const float4* W2; int n_chunks;  /* per chunk 64 + 1 steps */
    int last_tiles;                  /* live 32-row tiles of the last chunk */
    int act;
    const float* y; int64_t ldy;
    int64_t B; int d; int inverse;
    uint64_t circ_mask;                              /* bit j set = dim j circular (d <= 64) */
    float* out; int64_t ldo;
    float* dlogp; int accumulate;
    int32_t* bin_idx; int32_t* oob_count;
    int lds_per_wave;                                /* floats */
    BgkRqsCfg cfg;
};

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == 1) return bgk_siluf(v);
    if (ACT == 2) return v > 0.0f ? v : 0.0f;
    return bgk_tanhf(v);
}

/* ---- packed-weight stream --------------------------------------------------------------------
 * One k-step = one 1 KiB block = one global_load_dwordx4 per lane with a wave-uniform SGPR base, a
 * single per-lane VGPR byte offset (lane * 16) for the whole kernel and an immediate step offset.
 * hipcc (ROCm 7.2) sinks ordinary prefetch loads down to their first use (one full L2 round trip
 * per 4 MFMAs) and materialises/spills one 64-bit address per step, so the stream is issued from
 * inline asm and counted by hand (guide 5.7 form (ii)): every destination is named "+v" in the
 * s_waitcnt statement that precedes its first consumer, so no MFMA can be scheduled above its wait. */
typedef float f32x4 __attribute__((ext_vector_type(4)));

/* activation of one accumulator tile, two registers per packed-f32 instruction (same bits as the scalar form) */
template <int ACT>
__device__ __forceinline__ void act_tile(f32x16& t) {
    if constexpr (ACT == 1) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            bgk_f2 v = bgk_siluf2((bgk_f2){t[r], t[r + 1]});
            t[r] = v.x; t[r + 1] = v.y;
        }
    } else if constexpr (ACT == 3) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            bgk_f2 v = bgk_tanhf2((bgk_f2){t[r], t[r + 1]});
            t[r] = v.x; t[r + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = act_fn<ACT>(t[r]);
    }
}

/* hidden activation of the split-f16 / bf16 kernels: hardware exp / rcp (bgk_detmath_pk.h) */
template <int ACT>
__device__ __forceinline__ void act_tile_fast(f32x16& t) {
    if constexpr (ACT == 1 || ACT == 3) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const bgk_f2 v = ACT == 1 ? bgk_siluf2_fast((bgk_f2){t[r], t[r + 1]}) : bgk_tanhf2_fast((bgk_f2){t[r], t[r + 1]});
            t[r] = v.x; t[r + 1] = v.y;
        }
    } else {
        act_tile<ACT>(t);
    }
}

/* row of output tile m held in accumulator register r by this lane (hh = lane >> 5) */
__device__ __forceinline__ int drow(int m, int r, int hh) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int IMM>
__device__ __forceinline__ void wload(f32x4& dst, unsigned voff, const char* sbase) {
    /* s_nop 4: the SGPR base may have been (re)materialised by a VALU v_readlane right before this
     * statement; VALU-written SGPR -> VMEM read needs 5 wait states and hipcc pads nothing inside asm */
    /* no "memory" clobber: the packed weights are read-only for the whole launch, and the spline's LDS reads
     * must be free to be scheduled between these statements */
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(IMM));
}
template <int N>
__device__ __forceinline__ void wwait(f32x4& v) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "i"(N));
}
/* load of k-step S relative to a uniform base (steps are grouped in fours: 13-bit signed immediate) */
template <int S>
__device__ __forceinline__ void wload_step(f32x4& dst, unsigned voff, const char* base) {
    wload<(S % 4) * 1024>(dst, voff + (unsigned)((S / 4) * 4096), base);   /* one SGPR base per GEMM */
}

/* NT = number of live output tiles (the last parameter chunk of a layer may need fewer than 4) */
template <int NT = 4>
__device__ __forceinline__ void mfma4v(f32x16 (&acc)[4], const f32x4& a, float b) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0], 0, 0, 0);
    if constexpr (NT > 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1], 0, 0, 0);
    if constexpr (NT > 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc[2], 0, 0, 0);
    if constexpr (NT > 3) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc[3], 0, 0, 0);
}

/* acc[4] = W[128 x 128] * h + bias   with h in accumulator layout (k order = mfma order).
 * The packed operand has 64 + 1 k-steps: the last one multiplies the bias (A, lower half-wave) by
 * 1.0 (B) -- fma(bias, 1, acc) == acc + bias, the oracle's "bias added last". */
constexpr int HSTEPS = 65;

/* state of one in-flight 65-step GEMM (A stream) */
struct Stream {
    f32x4 ring[PF];
    unsigned voff;
    const char* base;
    float bias_b;
};

__device__ __forceinline__ void gstart(Stream& st, const float4* Wbase, int lane) {
    st.base = reinterpret_cast<const char*>(Wbase);
    st.voff = (unsigned)lane * 16u;
    st.bias_b = lane < 32 ? 1.0f : 0.0f;
    static_assert(PF == 4, "prologue written for PF = 4");
    wload_step<0>(st.ring[0], st.voff, st.base);
    wload_step<1>(st.ring[1], st.voff, st.base);
    wload_step<2>(st.ring[2], st.voff, st.base);
    wload_step<3>(st.ring[3], st.voff, st.base);
}

/* k-step S: wait for its A block, refill the ring slot, 4 MFMAs (B = in[S/16][S%16] or the bias 1/0) */
template <int S, int NT = 4>
__device__ __forceinline__ void gstep(Stream& st, f32x16 (&out)[4], const f32x16 (&in)[4]) {
    constexpr int newer = (HSTEPS - 1 - S) < (PF - 1) ? (HSTEPS - 1 - S) : (PF - 1);
    wwait<newer>(st.ring[S % PF]);
    const f32x4 a = st.ring[S % PF];
    if constexpr (S + PF < HSTEPS) wload_step<S + PF>(st.ring[S % PF], st.voff, st.base);
    if constexpr (S < 64) mfma4v<NT>(out, a, in[S / 16][S % 16]);
    else mfma4v<NT>(out, a, st.bias_b);
}

/* GEMM whose INPUT tiles 1..3 still need their activation: tile 0 must already be activated; tile
 * 1 + S/16 is activated element by element behind k-steps 0..47 (it is first consumed at step 16). */
template <int ACT, int S>
struct ActGemm {
    static __device__ __forceinline__ void run(Stream& st, f32x16 (&out)[4], f32x16 (&in)[4]) {
        gstep<S>(st, out, in);
#if BGK_PIPE_ACT
        if constexpr (S < 48) in[1 + S / 16][S % 16] = act_fn<ACT>(in[1 + S / 16][S % 16]);
#else
        if constexpr (S == 0) {
            act_tile<ACT>(in[1]); act_tile<ACT>(in[2]); act_tile<ACT>(in[3]);
        }
#endif
        if constexpr (S + 1 < HSTEPS) ActGemm<ACT, S + 1>::run(st, out, in);
    }
};

/* plain steps [S, E) */
template <int S, int E, int NT = 4>
struct PlainSteps {
    static __device__ __forceinline__ void run(Stream& st, f32x16 (&out)[4], const f32x16 (&in)[4]) {
        if constexpr (S < E) {
            gstep<S, NT>(st, out, in);
            PlainSteps<S + 1, E, NT>::run(st, out, in);
        }
    }
};

/* hook policies for the pipelined spline: a live GEMM, or nothing (last chunk) */
struct LiveGemm {
    Stream& st; f32x16 (&out)[4]; const f32x16 (&in)[4];
    template <int S> __device__ __forceinline__ void step() { if constexpr (S < HSTEPS) gstep<S>(st, out, in); }
};
struct NoGemm {
    template <int S> __device__ __forceinline__ void step() {}
};

/* Arithmetic policy of the spline element.  HW = false: the reproducible f32 sequences of bgk_detmath*.h (exact-f32 kernel,
 * bit-identical to the oracle).  HW = true (split-f16 / bf16 kernels, whose conditioner output is tolerance-class anyway):
 * v_exp_f32 / v_log_f32 (1 ulp each) for the softmax terms, the softplus and the log-det; divisions stay exactly rounded.
 * Measured on cfg 3 against the reference's f64 evaluation: max relative log-det error 1.8e-6 (exact arithmetic: 1.6e-6; the
 * reference's own f32 path: 4.9e-6), 12 % faster end to end (packed-f32 polynomials cost 2 issue slots per instruction on
 * gfx950, the hardware ops a quarter-rate slot each). */
template <bool HW> struct SpMath;
template <> struct SpMath<false> {
    static __device__ __forceinline__ bgk_f2 exp2(bgk_f2 x) { return bgk_expf2(x); }
    static __device__ __forceinline__ bgk_f2 log2(bgk_f2 x) { return bgk_logf2(x); }
    static __device__ __forceinline__ float div(float n, float d) { return bgk_div_safe(n, d); }
    static __device__ __forceinline__ bgk_f2 divr2(bgk_f2 n, bgk_f2 d, bgk_f2 r) { return bgk_div_r2(n, d, r); }
    static __device__ __forceinline__ float rcp(float d) { return bgk_rcp_refined(d); }
    static __device__ __forceinline__ bgk_f2 softplus2(bgk_f2 x, float beta) { return bgk_softplusf2(x, beta); }
};
#ifndef BGK_HW_EXP
#define BGK_HW_EXP 1
#endif
#ifndef BGK_HW_LOG
#define BGK_HW_LOG 1
#endif
#ifndef BGK_HW_DIV
#define BGK_HW_DIV 0      /* measured (tools/exp_hw_combos.sh, cfg 3 vs the f64 goldens): exp + log cost nothing in accuracy
                           * (1.8e-6 vs 1.6e-6 max relative log-det error) and give the whole speed-up (13.8 -> 12.25 ms);
                           * unrefined reciprocals on top are no faster and 3x less accurate (5.9e-6) */
#endif
template <> struct SpMath<true> {
    static __device__ __forceinline__ bgk_f2 exp2(bgk_f2 x) {
#if BGK_HW_EXP
        const bgk_f2 y = x * bgk_splat2(1.44269504088896341f);
        bgk_f2 r; r.x = __builtin_amdgcn_exp2f(y.x); r.y = __builtin_amdgcn_exp2f(y.y); return r;
#else
        return bgk_expf2(x);
#endif
    }
    static __device__ __forceinline__ bgk_f2 log2(bgk_f2 x) {
#if BGK_HW_LOG
        bgk_f2 r; r.x = __builtin_amdgcn_logf(x.x); r.y = __builtin_amdgcn_logf(x.y); return r * bgk_splat2(0.693147180559945309f);
#else
        return bgk_logf2(x);
#endif
    }
#if BGK_HW_DIV
    static __device__ __forceinline__ float div(float n, float d) { return n * __builtin_amdgcn_rcpf(d); }
    static __device__ __forceinline__ bgk_f2 divr2(bgk_f2 n, bgk_f2, bgk_f2 r) { return n * r; }
    static __device__ __forceinline__ float rcp(float d) { return __builtin_amdgcn_rcpf(d); }
#else
    static __device__ __forceinline__ float div(float n, float d) { return bgk_div_safe(n, d); }
    static __device__ __forceinline__ bgk_f2 divr2(bgk_f2 n, bgk_f2 d, bgk_f2 r) { return bgk_div_r2(n, d, r); }
    static __device__ __forceinline__ float rcp(float d) { return bgk_rcp_refined(d); }
#endif
    static __device__ __forceinline__ bgk_f2 softplus2(bgk_f2 x, float beta) {
#if BGK_HW_EXP && BGK_HW_LOG
        const bgk_f2 z = x * bgk_splat2(beta);
        bgk_f2 l = log2(bgk_splat2(1.0f) + exp2(z)) * bgk_splat2(__builtin_amdgcn_rcpf(beta));
        l.x = z.x > 20.0f ? x.x : l.x; l.y = z.y > 20.0f ? x.y : l.y;
        return l;
#else
        return bgk_softplusf2(x, beta);
#endif
    }
};

/* One spline element with 21 hook points; hook i runs k-step S0 + i of the overlapped GEMM.  Same
 * arithmetic, same order as bgk_rqs_element (bgk_common.h) -- only the instruction placement differs. */
constexpr int HOOKS = 21;
template <int INV, int S0, class G, int ST = 32, bool HW = false>
__device__ __forceinline__ float rqs_element_piped(G& g, float x, const float* pw, const float* ph, const float* ps,
                                                   float s_last, const BgkRqsCfg& c, float* lad, int* bin, int* oob) {
    constexpr int K = KB, st = ST;
    int o = (x < c.left) | (x > c.right);
    x = x < c.left ? c.left : (x > c.right ? c.right : x);
    *oob = o;
    const float* pa = INV ? pw : ph;
    const float* pb = INV ? ph : pw;
    const float minA = INV ? c.min_w : c.min_h, minB = INV ? c.min_h : c.min_w;
    const float scA = INV ? c.w_scale : c.h_scale, scB = INV ? c.h_scale : c.w_scale;
    const float spanA = INV ? c.xspan : c.yspan, spanB = INV ? c.yspan : c.xspan;
    const float lowA = INV ? c.left : c.bottom, lowB = INV ? c.bottom : c.left;
    const float highA = INV ? c.right : c.top, highB = INV ? c.top : c.right;
    float ra[K], e[K];
    /* ---- searched set ---- */
#pragma unroll
    for (int k = 0; k < K; ++k) ra[k] = pa[k * st];
    float mA = ra[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mA = ra[k] > mA ? ra[k] : mA;
    g.template step<S0 + 0>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[0] - mA, ra[1] - mA}); e[0] = t.x; e[1] = t.y; } g.template step<S0 + 1>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[2] - mA, ra[3] - mA}); e[2] = t.x; e[3] = t.y; } g.template step<S0 + 2>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[4] - mA, ra[5] - mA}); e[4] = t.x; e[5] = t.y; } g.template step<S0 + 3>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[6] - mA, ra[7] - mA}); e[6] = t.x; e[7] = t.y; }
    float sA = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) sA += e[k];
    g.template step<S0 + 4>();
    int idx = -1 + (x >= lowA ? 1 : 0);
    float lo = lowA, hi = lowA, cum = 0.0f;
    bool hi_set = false;
    const bgk_f2 rA2 = bgk_splat2(SpMath<HW>::rcp(sA)), sA2 = bgk_splat2(sA);
#pragma unroll
    for (int k = 0; k < K; k += 2) {
        /* two knots per packed op; the running sum stays sequential (same bits as the scalar oracle) */
        bgk_f2 p = SpMath<HW>::divr2((bgk_f2){e[k], e[k + 1]}, sA2, rA2);
        p = bgk_splat2(minA) + bgk_splat2(scA) * p;
        const float c0 = cum + p.x, c1 = c0 + p.y;
        cum = c1;
        bgk_f2 kn2 = bgk_splat2(spanA) * (bgk_f2){c0, c1} + bgk_splat2(lowA);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float kn = u ? kn2.y : kn2.x;
            if (k + u == K - 1) kn = highA;
            float ks = (k + u == K - 1) ? kn + 1e-6f : kn;
            bool ge = x >= ks;
            idx += ge ? 1 : 0;
            if (ge) lo = kn;
            if (!ge && !hi_set) { hi = kn; hi_set = true; }
        }
        if (k == 0) g.template step<S0 + 5>();
        if (k == 2) g.template step<S0 + 6>();
        if (k == 4) g.template step<S0 + 7>();
        if (k == 6) g.template step<S0 + 8>();
    }
    idx = idx < 0 ? 0 : idx;
    *bin = idx;
    const float a_i = lo, A_i = hi - lo;
    /* ---- other set ---- */
#pragma unroll
    for (int k = 0; k < K; ++k) ra[k] = pb[k * st];
    float mB = ra[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mB = ra[k] > mB ? ra[k] : mB;
    g.template step<S0 + 9>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[0] - mB, ra[1] - mB}); e[0] = t.x; e[1] = t.y; } g.template step<S0 + 10>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[2] - mB, ra[3] - mB}); e[2] = t.x; e[3] = t.y; } g.template step<S0 + 11>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[4] - mB, ra[5] - mB}); e[4] = t.x; e[5] = t.y; } g.template step<S0 + 12>();
    { bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[6] - mB, ra[7] - mB}); e[6] = t.x; e[7] = t.y; }
    float sB = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) sB += e[k];
    g.template step<S0 + 13>();
    float b_i = lowB, b_ip1 = lowB;
    cum = 0.0f;
    const bgk_f2 rB2 = bgk_splat2(SpMath<HW>::rcp(sB)), sB2 = bgk_splat2(sB);
#pragma unroll
    for (int k = 0; k < K; k += 2) {
        bgk_f2 p = SpMath<HW>::divr2((bgk_f2){e[k], e[k + 1]}, sB2, rB2);
        p = bgk_splat2(minB) + bgk_splat2(scB) * p;
        const float c0 = cum + p.x, c1 = c0 + p.y;
        cum = c1;
        bgk_f2 kn2 = bgk_splat2(spanB) * (bgk_f2){c0, c1} + bgk_splat2(lowB);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float kn = u ? kn2.y : kn2.x;
            if (k + u == K - 1) kn = highB;
            if (k + u + 1 == idx) b_i = kn;
            if (k + u == idx) b_ip1 = kn;
        }
        if (k == 0) g.template step<S0 + 14>();
        if (k == 2) g.template step<S0 + 15>();
        if (k == 4) g.template step<S0 + 16>();
        if (k == 6) g.template step<S0 + 17>();
    }
    const float B_i = b_ip1 - b_i;
    /* ---- gathered derivatives ---- */
    float s_lo = ps[idx * st];
    float s_hi = (idx + 1 < K) ? ps[(idx + 1 < K ? idx + 1 : 0) * st] : s_last;
    const bgk_f2 sp = SpMath<HW>::softplus2((bgk_f2){s_lo, s_hi}, c.beta);
    g.template step<S0 + 18>();
    float d_i = c.min_d + sp.x;
    float d_ip1 = c.min_d + sp.y;
    g.template step<S0 + 19>();
    float cw_i, W_i, ch_i, H_i;
    if (INV) { cw_i = a_i; W_i = A_i; ch_i = b_i; H_i = B_i; }
    else { ch_i = a_i; H_i = A_i; cw_i = b_i; W_i = B_i; }
    float delta = SpMath<HW>::div(H_i, W_i);
    float S = d_i + d_ip1 - 2.0f * delta;
    float outv, l;
    if (!INV) {
        float dx = x - ch_i;
        float a = dx * S + H_i * (delta - d_i);
        float b = H_i * d_i - dx * S;
        float cc = -delta * dx;
        float disc = b * b - 4.0f * a * cc;
        float root = SpMath<HW>::div(2.0f * cc, -b - __builtin_sqrtf(disc));
        outv = root * W_i + cw_i;
        float t1mt = root * (1.0f - root);
        float den = delta + S * t1mt;
        float omr = 1.0f - root;
        float num = (delta * delta) * (d_ip1 * (root * root) + 2.0f * delta * t1mt + d_i * (omr * omr));
        { const bgk_f2 lg = SpMath<HW>::log2((bgk_f2){num, den}); l = -(lg.x - 2.0f * lg.y); }
    } else {
        float theta = SpMath<HW>::div(x - cw_i, W_i);
        float t1mt = theta * (1.0f - theta);
        float numer = H_i * (delta * (theta * theta) + d_i * t1mt);
        float den = delta + S * t1mt;
        outv = ch_i + SpMath<HW>::div(numer, den);
        float omt = 1.0f - theta;
        float num = (delta * delta) * (d_ip1 * (theta * theta) + 2.0f * delta * t1mt + d_i * (omt * omt));
        { const bgk_f2 lg = SpMath<HW>::log2((bgk_f2){num, den}); l = lg.x - 2.0f * lg.y; }
    }
    g.template step<S0 + 20>();
    *lad = l;
    return outv;
}

__device__ __forceinline__ void zero4(f32x16 (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
}

/* Spline of one parameter chunk held in LDS: 3 element evaluations per lane (q = hh, hh+2, hh+4),
 * written branch-free so that the whole routine is ONE basic block and can be software-pipelined
 * under the next chunk's MFMAs.  Invalid (q >= nd) slots evaluate dim 0 of the chunk and are
 * discarded (their output goes to the dummy row d of s_y). */
template <int INV, int IT, class G, int ST = 32, bool HW = false>
__device__ __forceinline__ void spline_slot(G& g, const FusedArgs& a, const float* s_p, float* s_y, int c, int nd,
                                            int hh, int j, int rows, float& run, int& oob_local, int (&bins)[3]) {
    const int q = 2 * IT + hh;
    const bool valid = q < nd;
    const int qq = valid ? q : 0;
    const int dim = c * DPC + qq;
    const float* pw = s_p + (qq * PPD) * ST + j;
    const float* ph = pw + KB * ST;
    const float* ps = ph + KB * ST;
    const bool circ = (a.circ_mask >> dim) & 1ull;
    const float s_nc = ps[KB * ST];
    const float s_last = circ ? ps[0] : s_nc;
    int bin, oob;
    float lad;
    const float x = s_y[dim * SROW + j];
    const float o = rqs_element_piped<INV, IT * HOOKS, G, ST, HW>(g, x, pw, ph, ps, s_last, a.cfg, &lad, &bin, &oob);
    s_y[(valid ? dim : a.d) * SROW + j] = o;
    oob_local += (valid && j < rows) ? oob : 0;
    bins[IT] = bin;
    lad = valid ? lad : 0.0f;
    /* dim 2*IT lives in the lower half-wave, dim 2*IT+1 in the upper one: exchange the two log-dets and let
     * BOTH halves add them in ascending dim order (same bits as the oracle) */
    const float lad_other = __shfl_xor(lad, 32);
    const float l0 = hh ? lad_other : lad;
    const float l1 = hh ? lad : lad_other;
    const float r0 = run + l0;
    run = (2 * IT < nd) ? r0 : run;
    const float r1 = run + l1;
    run = (2 * IT + 1 < nd) ? r1 : run;
}

/* Spline of one parameter chunk held in LDS: 3 element evaluations per lane (q = hh, hh+2, hh+4), branch-free
 * (invalid slots evaluate dim 0 of the chunk and are discarded into the dummy row d of s_y), with the hook
 * policy G running the overlapped GEMM's k-steps 0..62 in between. */
template <int INV, class G, int ST = 32, bool HW = false>
__device__ __forceinline__ void spline_chunk(G& g, const FusedArgs& a, const float* s_p, float* s_y, int c, int nd,
                                             int hh, int j, int rows, float& run, int& oob_local, int (&bins)[3]) {
    spline_slot<INV, 0, G, ST, HW>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
#if BGK_PIPE_SPLINE
    spline_slot<INV, 1, G, ST, HW>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
    spline_slot<INV, 2, G, ST, HW>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
#else
    /* slots whose two dims both lie beyond the chunk's last dim are skipped (wave-uniform branch) */
    bins[1] = bins[2] = 0;
    if (nd > 2) spline_slot<INV, 1, G, ST, HW>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
    if (nd > 4) spline_slot<INV, 2, G, ST, HW>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
#endif
}

/* ---- any even bin count K (the K = 8 routines above carry hand-placed hook points for the software-pipelined exact-f32 kernel;
 * this is the same arithmetic in the same order written as loops) ---- */
template <int INV, int K, int ST, bool HW>
__device__ __forceinline__ float rqs_element_k(float x, const float* pw, const float* ph, const float* ps, float s_last,
                                               const BgkRqsCfg& c, float* lad, int* bin, int* oob) {
    static_assert(K % 2 == 0, "bin pairs");
    constexpr int st = ST;
    int o = (x < c.left) | (x > c.right);
    x = x < c.left ? c.left : (x > c.right ? c.right : x);
    *oob = o;
    const float* pa = INV ? pw : ph;
    const float* pb = INV ? ph : pw;
    const float minA = INV ? c.min_w : c.min_h, minB = INV ? c.min_h : c.min_w;
    const float scA = INV ? c.w_scale : c.h_scale, scB = INV ? c.h_scale : c.w_scale;
    const float spanA = INV ? c.xspan : c.yspan, spanB = INV ? c.yspan : c.xspan;
    const float lowA = INV ? c.left : c.bottom, lowB = INV ? c.bottom : c.left;
    const float highA = INV ? c.right : c.top, highB = INV ? c.top : c.right;
    float ra[K], e[K];
    /* ---- searched set ---- */
#pragma unroll
    for (int k = 0; k < K; ++k) ra[k] = pa[k * st];
    float mA = ra[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mA = ra[k] > mA ? ra[k] : mA;
#pragma unroll
    for (int k = 0; k < K; k += 2) { const bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[k] - mA, ra[k + 1] - mA}); e[k] = t.x; e[k + 1] = t.y; }
    float sA = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) sA += e[k];
    int idx = -1 + (x >= lowA ? 1 : 0);
    float lo = lowA, hi = lowA, cum = 0.0f;
    bool hi_set = false;
    const bgk_f2 rA2 = bgk_splat2(SpMath<HW>::rcp(sA)), sA2 = bgk_splat2(sA);
#pragma unroll
    for (int k = 0; k < K; k += 2) {
        bgk_f2 p = SpMath<HW>::divr2((bgk_f2){e[k], e[k + 1]}, sA2, rA2);
        p = bgk_splat2(minA) + bgk_splat2(scA) * p;
        const float c0 = cum + p.x, c1 = c0 + p.y;
        cum = c1;
        const bgk_f2 kn2 = bgk_splat2(spanA) * (bgk_f2){c0, c1} + bgk_splat2(lowA);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float kn = u ? kn2.y : kn2.x;
            if (k + u == K - 1) kn = highA;
            const float ks = (k + u == K - 1) ? kn + 1e-6f : kn;
            const bool ge = x >= ks;
            idx += ge ? 1 : 0;
            if (ge) lo = kn;
            if (!ge && !hi_set) { hi = kn; hi_set = true; }
        }
    }
    idx = idx < 0 ? 0 : idx;
    *bin = idx;
    const float a_i = lo, A_i = hi - lo;
    /* ---- other set ---- */
#pragma unroll
    for (int k = 0; k < K; ++k) ra[k] = pb[k * st];
    float mB = ra[0];
#pragma unroll
    for (int k = 1; k < K; ++k) mB = ra[k] > mB ? ra[k] : mB;
#pragma unroll
    for (int k = 0; k < K; k += 2) { const bgk_f2 t = SpMath<HW>::exp2((bgk_f2){ra[k] - mB, ra[k + 1] - mB}); e[k] = t.x; e[k + 1] = t.y; }
    float sB = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) sB += e[k];
    float b_i = lowB, b_ip1 = lowB;
    cum = 0.0f;
    const bgk_f2 rB2 = bgk_splat2(SpMath<HW>::rcp(sB)), sB2 = bgk_splat2(sB);
#pragma unroll
    for (int k = 0; k < K; k += 2) {
        bgk_f2 p = SpMath<HW>::divr2((bgk_f2){e[k], e[k + 1]}, sB2, rB2);
        p = bgk_splat2(minB) + bgk_splat2(scB) * p;
        const float c0 = cum + p.x, c1 = c0 + p.y;
        cum = c1;
        const bgk_f2 kn2 = bgk_splat2(spanB) * (bgk_f2){c0, c1} + bgk_splat2(lowB);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float kn = u ? kn2.y : kn2.x;
            if (k + u == K - 1) kn = highB;
            if (k + u + 1 == idx) b_i = kn;
            if (k + u == idx) b_ip1 = kn;
        }
    }
    const float B_i = b_ip1 - b_i;
    /* ---- gathered derivatives ---- */
    const float s_lo = ps[idx * st];
    const float s_hi = (idx + 1 < K) ? ps[(idx + 1 < K ? idx + 1 : 0) * st] : s_last;
    const bgk_f2 sp = SpMath<HW>::softplus2((bgk_f2){s_lo, s_hi}, c.beta);
    const float d_i = c.min_d + sp.x, d_ip1 = c.min_d + sp.y;
    float cw_i, W_i, ch_i, H_i;
    if (INV) { cw_i = a_i; W_i = A_i; ch_i = b_i; H_i = B_i; }
    else { ch_i = a_i; H_i = A_i; cw_i = b_i; W_i = B_i; }
    const float delta = SpMath<HW>::div(H_i, W_i);
    const float S = d_i + d_ip1 - 2.0f * delta;
    float outv, l;
    if (!INV) {
        const float dx = x - ch_i;
        const float a = dx * S + H_i * (delta - d_i);
        const float b = H_i * d_i - dx * S;
        const float cc = -delta * dx;
        const float disc = b * b - 4.0f * a * cc;
        const float root = SpMath<HW>::div(2.0f * cc, -b - __builtin_sqrtf(disc));
        outv = root * W_i + cw_i;
        const float t1mt = root * (1.0f - root);
        const float den = delta + S * t1mt;
        const float omr = 1.0f - root;
        const float num = (delta * delta) * (d_ip1 * (root * root) + 2.0f * delta * t1mt + d_i * (omr * omr));
        { const bgk_f2 lg = SpMath<HW>::log2((bgk_f2){num, den}); l = -(lg.x - 2.0f * lg.y); }
    } else {
        const float theta = SpMath<HW>::div(x - cw_i, W_i);
        const float t1mt = theta * (1.0f - theta);
        const float numer = H_i * (delta * (theta * theta) + d_i * t1mt);
        const float den = delta + S * t1mt;
        outv = ch_i + SpMath<HW>::div(numer, den);
        const float omt = 1.0f - theta;
        const float num = (delta * delta) * (d_ip1 * (theta * theta) + 2.0f * delta * t1mt + d_i * (omt * omt));
        { const bgk_f2 lg = SpMath<HW>::log2((bgk_f2){num, den}); l = lg.x - 2.0f * lg.y; }
    }
    *lad = l;
    return outv;
}

/* spline of one parameter chunk for K bins: the chunk holds DPC = 128 / (3 K + 1) dims, lane (j, hh) evaluates dims hh, hh + 2, ...
 * of sample j; log-dets are added in ascending dim order by both half-waves (as spline_slot does) */
template <int INV, int K, int ST, bool HW>
__device__ __forceinline__ void spline_chunk_k(const FusedArgs& a, const float* s_p, float* s_y, int c, int nd, int hh, int j, int rows,
                                               int64_t b0, float& run, int& oob_local) {
    constexpr int PPDK = 3 * K + 1, DPCK = 128 / PPDK;
    for (int it = 0; 2 * it < nd; ++it) {                  /* wave-uniform trip count */
        const int q = 2 * it + hh;
        const bool valid = q < nd;
        const int qq = valid ? q : 0;
        const int dim = c * DPCK + qq;
        const float* pw = s_p + (qq * PPDK) * ST + j;
        const float* ph = pw + K * ST;
        const float* ps = ph + K * ST;
        const bool circ = (a.circ_mask >> dim) & 1ull;
        const float s_nc = ps[K * ST];
        const float s_last = circ ? ps[0] : s_nc;
        int bin, oob;
        float lad;
        const float x = s_y[dim * SROW + j];
        const float o = rqs_element_k<INV, K, ST, HW>(x, pw, ph, ps, s_last, a.cfg, &lad, &bin, &oob);
        s_y[(valid ? dim : a.d) * SROW + j] = o;
        oob_local += (valid && j < rows) ? oob : 0;
        if (a.bin_idx && valid && j < rows) a.bin_idx[(b0 + j) * a.d + dim] = bin;
        lad = valid ? lad : 0.0f;
        const float lad_other = __shfl_xor(lad, 32);
        const float l0 = hh ? lad_other : lad;
        const float l1 = hh ? lad : lad_other;
        const float r0 = run + l0;
        run = (2 * it < nd) ? r0 : run;
        const float r1 = run + l1;
        run = (2 * it + 1 < nd) ? r1 : run;
    }
}

template <int ACT, int INV>
__global__ __launch_bounds__(FTHREADS, 2) void coupling_rqs_dense_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* s_p = smem + (size_t)wave * a.lds_per_wave;   /* parameter chunk [128][32]; aliases the layer-0 input X0 [2*T0][SROW] */
    float* s_y = s_p + LDS_P;                            /* y / out tile [d][SROW] */
    const int d = a.d;
    const int64_t n_tiles = (a.B + 31) / 32;

    /* one 32-sample tile per wave (no grid-stride loop: hipcc peels/duplicates the 13k-instruction body) */
    const int64_t tile = (int64_t)blockIdx.x * FW + wave;
    if (tile < n_tiles) {
        const int64_t b0 = tile * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);

        /* ---- stage conditioner input (featurised) and y, transposed to [feature][sample] ---- */
        const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
        for (int i = lane; i < 32 * a.d_c; i += 64) {
            const int r = i / a.d_c, c = i - r * a.d_c;
            float v = r < rows ? a.cond[(b0 + r) * a.ldc + c] : 0.0f;
            if (a.periodic) {
                float sv, cv;
                bgk_sincos2pif(v, &sv, &cv);
                s_p[c * SROW + r] = cv;
                s_p[(a.d_c + c) * SROW + r] = sv;
            } else {
                s_p[c * SROW + r] = v;
            }
        }
        for (int i = lane; i < (2 * a.T0 - n_in) * 32; i += 64) s_p[(n_in + (i >> 5)) * SROW + (i & 31)] = 0.0f;   /* pad rows (zero weights) */
        for (int i = lane; i < 32 * d; i += 64) {
            const int r = i / d, c = i - r * d;
            s_y[c * SROW + r] = r < rows ? a.y[(b0 + r) * a.ldy + c] : 0.5f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        /* ---- layer 0: h = act(W0 * x + b0), B operand from LDS, natural k order, bias = last step.
         * T0 (k-steps) is padded to a multiple of 4 with zero weights on the host. ---- */
        f32x16 h[4], acc[4];
        zero4(h);
        {
            const char* base = reinterpret_cast<const char*>(a.W0);
            const unsigned voff = (unsigned)lane * 16u;
            const int G = a.T0 >> 2;
            /* one group = 4 k-steps: 4 loads, then counted waits -- loads and the waits that retire them stay
             * in ONE basic block (asm destinations must not be live across a branch: hipcc may copy them while
             * the data is still in flight).  The exposed L2 round trip per group is ~2 % of the tile time. */
            for (int g = 0; g < G; ++g) {
                f32x4 A0, A1, A2, A3;
                const char* bn = base + (size_t)g * 4096;
                wload<0>(A0, voff, bn); wload<1024>(A1, voff, bn); wload<2048>(A2, voff, bn); wload<3072>(A3, voff, bn);
                const float* xr = s_p + (8 * g + hh) * SROW + j;
                const float x0 = xr[0 * 2 * SROW], x1 = xr[1 * 2 * SROW], x2 = xr[2 * 2 * SROW], x3 = xr[3 * 2 * SROW];
                wwait<3>(A0); mfma4v(h, A0, x0);
                wwait<2>(A1); mfma4v(h, A1, x1);
                wwait<1>(A2); mfma4v(h, A2, x2);
                wwait<0>(A3); mfma4v(h, A3, x3);
            }
            /* bias step (index T0) */
            const f32x4 ab = reinterpret_cast<const f32x4*>(base + (size_t)a.T0 * 1024)[lane];
            mfma4v(h, ab, lane < 32 ? 1.0f : 0.0f);
            /* activation of tile 0 only; tiles 1..3 are activated behind the next GEMM's k-steps */
            act_tile<ACT>(h[0]);
        }
        /* ---- layer 1: acc = W1 * act(h) + b1 (B operand = registers) ---- */
        Stream st;
        zero4(acc);
        gstart(st, a.W1, lane);
        ActGemm<ACT, 0>::run(st, acc, h);
        act_tile<ACT>(acc[0]);

        /* ---- layer 2 in chunks of 128 packed columns + spline, software-pipelined inside the wave:
         *   GEMM(chunk 0) [activating its own input tiles 1..3 on the fly];
         *   for c: { h -> LDS;  [ GEMM(chunk c+1) || spline(chunk c) ] }
         * From here on the roles are swapped: `acc` (layer-1 output, activated in place) is the B operand,
         * `h` (dead layer-0 activations) is the accumulator.  The spline is threaded through the GEMM's
         * k-steps (21 hook points per element): the f32 MFMA holds the matrix pipe for 64 cycles per
         * instruction, during which the same wave issues the spline's VALU work. ---- */
        float run = 0.0f;          /* running sum of log-dets of sample j (kept identical in both half-waves) */
        int oob_local = 0;
        zero4(h);
        gstart(st, a.W2, lane);
        ActGemm<ACT, 0>::run(st, h, acc);
        for (int c = 0; c < a.n_chunks; ++c) {
            /* spline(c-1) reads of s_p are complete (same wave, program order) */
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_p[drow(m, r, hh) * 32 + j] = h[m][r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int nd = (d - c * DPC) < DPC ? (d - c * DPC) : DPC;
            int bins[3];
            if (c + 1 < a.n_chunks) {
                zero4(h);
                const float4* Wn = a.W2 + (size_t)(c + 1) * HSTEPS * 64;
#if BGK_PIPE_SPLINE
                gstart(st, Wn, lane);
                LiveGemm g{st, h, acc};
                spline_chunk<INV>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
                PlainSteps<3 * HOOKS, HSTEPS>::run(st, h, acc);
#else
                /* the last chunk holds d - 5 (n_chunks - 1) dims = ceil(25 nd / 32) live row tiles: skip the rest */
                /* NB: the asm loads of gstart() and the waits that retire them must sit in ONE basic block --
                 * across a branch hipcc may copy the (still in flight) destination registers */
                if (c + 2 == a.n_chunks && a.last_tiles <= 2) { gstart(st, Wn, lane); PlainSteps<0, HSTEPS, 2>::run(st, h, acc); }
                else { gstart(st, Wn, lane); PlainSteps<0, HSTEPS>::run(st, h, acc); }
                NoGemm g;
                spline_chunk<INV>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
#endif
            } else {
                NoGemm g;
                spline_chunk<INV>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
            }
            if (a.bin_idx) {
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int q = 2 * it + hh;
                    if (q < nd && j < rows) a.bin_idx[(b0 + j) * d + c * DPC + q] = bins[it];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (hh == 0 && j < rows) {
            if (a.accumulate) a.dlogp[b0 + j] += run; else a.dlogp[b0 + j] = run;
        }
        for (int i = lane; i < rows * d; i += 64) {
            const int r = i / d, cc = i - r * d;
            a.out[(b0 + r) * a.ldo + cc] = s_y[cc * SROW + r];
        }
        if (a.oob_count) {
            for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
            if (lane == 0 && oob_local) atomicAdd(a.oob_count, oob_local);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


/* =====================================================================================================
 * Split-f16 variant: the same layer with the three GEMMs on the f16 matrix cores.
 * Every f32 operand v (weight, bias, activation) is represented EXACTLY-to-f32-rounding as hi + lo with
 * hi = rne_f16(v), lo = rne_f16(v - hi) (22-24 significant bits); a product needs three MFMAs
 * (lo*hi, hi*lo, hi*hi -- lo*lo is below f32 resolution), accumulated in f32 by v_mfma_f32_32x32x16_f16.
 * Measured on MI355X (tools/ubench/split_gemm.hip, 128-term dot products of N(0, 0.1) weights with
 * SiLU-distributed activations incl. 1e-6-scaled ones): rms error 0.47 ulp32 of sum|a||b| (max 3.5) versus
 * 0.65 (max 5.7) for the exact-f32 fma chain -- the same accuracy class, 16/3 times the MFMA rate
 * (v_mfma_f32_32x32x16_f16: 32 cycles for 16 k, f32-input 32x32x2: 64 cycles for 2 k).
 * Weights are scaled by a per-layer power of two (exact) into the upper f16 range on the host; the
 * accumulator is scaled back (exact) before the activation / the spline.  Activations beyond +-65000 are
 * clamped (f16 range) -- far outside anything a trained conditioner produces.
 * Not bit-identical to the oracle's fma chain (the accumulation order inside the instruction is the
 * hardware's); parity is asserted to the reference at the north-star tolerance instead.
 *   packed A operand (host: dense.py::pack_dense_for_fused_h2): per layer / per 128-row chunk
 *     block(s, m, p) = [(s * 4 + m) * 2 + p] of 1 KiB: lane l = 32 kb + i holds W'[32 m + i][k(s, kb, e)], e = 0..7,
 *     part p (0 = hi, 1 = lo); after the 8 (S0 for layer 0) k-steps: bias block m (lanes < 32: {b_hi, b_lo, 0...}).
 *   k(s, kb, e): layer 0 natural 16 s + 8 kb + e (LDS rows); hidden layers = the accumulator layout, so that the
 *     previous layer's registers are the B operand without data movement.
 */
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

struct FusedArgsH2 {
    FusedArgs f;                 /* shared part (W0/W1/W2/T0 unused) */
    const uint4* A0; int S0;     /* layer 0: S0 k16-steps */
    const uint4* A1;             /* layer 1: 8 steps */
    const uint4* A2;             /* layer 2: n_chunks x (8 steps + bias) */
    float c0, c1, c2;            /* 2^-s of the three layers */
    const float* cs_dev;         /* if non-null: {2^s0, 2^-s0, 2^s1, 2^-s1, 2^s2, 2^-s2} on the device (bgk_pack_dense_h2) */
    /* training forward (SAVE): pre-activations and spline parameters for the backward pass */
    float* z0; float* z1;        /* [B, 128] each */
    float* params; int64_t ldp;  /* [B, P] in the reference layout [w | h | s | s_nc] */
    const int32_t* src_col;      /* device copy of bgk_pack_rqs_columns' table: packed row -> params column, -1 = padding */
};

constexpr int H2_STEPS = 8;                            /* 128 hidden units / 16 */
constexpr int H2_BLOCKS = (H2_STEPS * 4 * 2 + 4);      /* 1 KiB blocks per 128-row GEMM incl. bias */

struct AFrag { uint4 v[4][2]; };     /* one k16-step: 4 output tiles x {hi, lo} */
struct BFrag { h16x8 hi[H2_STEPS], lo[H2_STEPS]; };   /* a 128-wide activation vector as B operands: 64 VGPRs */
typedef short s16x8 __attribute__((ext_vector_type(8)));

/* BF = true: single-bf16 operands (gemm_mode "bf16": bf16 parameter storage + bf16 GEMM inputs, f32 accumulate --
 * the reduced-precision variant of BASELINE config 5); same operand layout, only the hi blocks are read and hold
 * bf16 instead of f16, one v_mfma_f32_32x32x16_bf16 per product. */
__device__ __forceinline__ uint16_t bf16_rne(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool BF>
__device__ __forceinline__ void h2_load(AFrag& f, const uint4* W, int s, int lane) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#if (BGK_ABL & 16)   /* timing experiment: every k-step re-reads block 0 (L1-resident): is the A stream the limiter? */
        f.v[m][0] = W[((0 * 4 + m) * 2 + 0) * 64 + lane + (s & 0)];
        if constexpr (!BF) f.v[m][1] = W[((0 * 4 + m) * 2 + 1) * 64 + lane + (s & 0)];
#else
        f.v[m][0] = W[((s * 4 + m) * 2 + 0) * 64 + lane];
        if constexpr (!BF) f.v[m][1] = W[((s * 4 + m) * 2 + 1) * 64 + lane];
#endif
    }
}
__device__ __forceinline__ void h2_load_bias(AFrag& f, const uint4* W, int lane) {
#pragma unroll
    for (int m = 0; m < 4; ++m) f.v[m][0] = W[(H2_STEPS * 8 + m) * 64 + lane];
}

template <bool BF>
__device__ __forceinline__ void h2_split(const float (&v)[8], h16x8& hi, h16x8& lo) {
    if constexpr (BF) {
        s16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (short)bf16_rne(v[e]);
        hi = __builtin_bit_cast(h16x8, t);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float c = __builtin_amdgcn_fmed3f(v[e], -65000.0f, 65000.0f);
            const _Float16 h = (_Float16)c;
            hi[e] = h;
            lo[e] = (_Float16)(c - (float)h);
        }
    }
}

/* activated f32 tiles (accumulator layout) -> B operands of the 8 k16-steps; done ONCE per layer input */
template <bool BF>
__device__ __forceinline__ void h2_make_b(BFrag& b, const f32x16 (&in)[4]) {
#pragma unroll
    for (int s = 0; s < H2_STEPS; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in[s >> 1][8 * (s & 1) + e];
        h2_split<BF>(v, b.hi[s], b.lo[s]);
    }
}

template <bool BF>
__device__ __forceinline__ void h2_mfma(f32x16 (&out)[4], const AFrag& a, const h16x8& bhi, const h16x8& blo) {
    if constexpr (BF) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a.v[m][0]), __builtin_bit_cast(s16x8, bhi), out[m], 0, 0, 0);
    } else {
        /* small terms first; tiles interleaved so that consecutive MFMAs are independent */
#pragma unroll
        for (int m = 0; m < 4; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a.v[m][1]), bhi, out[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a.v[m][0]), blo, out[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a.v[m][0]), bhi, out[m], 0, 0, 0);
    }
}

/* A-operand ring of one 128-row GEMM: the first H2_RING - 1 steps are requested by h2_gemm_start (possibly long before the
 * GEMM runs -- e.g. ahead of the previous chunk's spline, which hides the L2 round trip), step s + H2_RING - 1
 * while step s computes. */
constexpr int H2_RING = 2;
struct H2Ring { AFrag f[H2_RING]; };

template <bool BF>
__device__ __forceinline__ void h2_gemm_start(H2Ring& r, const uint4* W, int lane) {
#pragma unroll
    for (int s = 0; s < H2_RING - 1; ++s) h2_load<BF>(r.f[s], W, s, lane);
    __builtin_amdgcn_sched_barrier(0);
}

/* out[0..4) += W' * b + b'   (ring already started) */
template <bool BF>
__device__ __forceinline__ void h2_gemm_run(f32x16 (&out)[4], H2Ring& r, const BFrag& b, const uint4* W, int lane) {
#pragma unroll
    for (int s = 0; s < H2_STEPS; ++s) {
        constexpr int D = H2_RING - 1;
#if (BGK_ABL & 32)   /* timing experiment: the A stream of steps 2.. is not loaded (is the vector-memory path the limiter?) */
        if (s == 0) h2_load<BF>(r.f[(s + D) % H2_RING], W, s + D, lane);
#else
        if (s + D < H2_STEPS) h2_load<BF>(r.f[(s + D) % H2_RING], W, s + D, lane);
        else if (s + D == H2_STEPS) h2_load_bias(r.f[(s + D) % H2_RING], W, lane);
#endif
        __builtin_amdgcn_sched_barrier(0);   /* keep the prefetch above this step's MFMAs */
        h2_mfma<BF>(out, r.f[s % H2_RING], b.hi[s], b.lo[s]);
    }
    if constexpr (BF) {
        const s16x8 one2 = {(short)0x3f80, (short)0x3f80, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m)
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, r.f[H2_STEPS % H2_RING].v[m][0]), one2, out[m], 0, 0, 0);
    } else {
        const h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m)
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, r.f[H2_STEPS % H2_RING].v[m][0]), one2, out[m], 0, 0, 0);
    }
}

/* activation of x = t * c (c = exact power-of-two unscale) */
template <int ACT>
__device__ __forceinline__ void act_tile_scaled(f32x16& t, float c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] *= c;
#if !(BGK_ABL & 4)
    act_tile_fast<ACT>(t);
#endif
}

template <int ACT, int INV, bool SAVE, bool BF, int KT = KB>
__global__ __launch_bounds__(FTHREADS, 2) void coupling_rqs_dense_h2_kernel(FusedArgsH2 ah) {
    constexpr int DPCT = 128 / (3 * KT + 1);              /* dims per 128-column parameter chunk */
    const FusedArgs& a = ah.f;
    if (ah.cs_dev) { ah.c0 = ah.cs_dev[1]; ah.c1 = ah.cs_dev[3]; ah.c2 = ah.cs_dev[5]; }   /* wave-uniform scalar loads */
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    /* parameter chunk [128][ST]: stride 33 in the training variant so that the chunk can also be read row-wise
     * (lane = row) without bank conflicts for the coalesced parameter write-out */
    constexpr int ST = SAVE ? 33 : 32;
    float* s_p = smem + (size_t)wave * a.lds_per_wave;
    float* s_y = s_p + 128 * ST;
    const int d = a.d;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * FW + wave;
    if (tile < n_tiles) {
        const int64_t b0 = tile * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);

        /* ---- stage the (featurised) conditioner input [feature][sample], a constant-1 row for the bias, zero pad rows ---- */
        const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
        for (int i = lane; i < 32 * a.d_c; i += 64) {
            const int r = i / a.d_c, c = i - r * a.d_c;
            float v = r < rows ? a.cond[(b0 + r) * a.ldc + c] : 0.0f;
            if (a.periodic) {
                float sv, cv;
                bgk_sincos2pif(v, &sv, &cv);
                s_p[c * SROW + r] = cv;
                s_p[(a.d_c + c) * SROW + r] = sv;
            } else {
                s_p[c * SROW + r] = v;
            }
        }
        for (int i = lane; i < (16 * ah.S0 - n_in) * 32; i += 64)
            s_p[(n_in + (i >> 5)) * SROW + (i & 31)] = (i >> 5) == 0 ? 1.0f : 0.0f;
        for (int i = lane; i < 32 * d; i += 64) {
            const int r = i / d, c = i - r * d;
            s_y[c * SROW + r] = r < rows ? a.y[(b0 + r) * a.ldy + c] : 0.5f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        /* ---- layer 0 (bias = weight column of the constant-1 feature) ---- */
        f32x16 h[4], acc[4];
        zero4(h);
        for (int s = 0; s < ah.S0; ++s) {
            AFrag fr;
            h2_load<BF>(fr, ah.A0, s, lane);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = s_p[(16 * s + 8 * hh + e) * SROW + j];
            h16x8 bhi, blo;
            h2_split<BF>(v, bhi, blo);
            h2_mfma<BF>(h, fr, bhi, blo);
        }
        H2Ring ring;
        h2_gemm_start<BF>(ring, ah.A1, lane);          /* layer-1 operands in flight during the activation */
        if constexpr (SAVE) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[m][r] *= ah.c0;
            h2_store_rows128(h, ah.z0, s_p, b0, rows, lane);
#pragma unroll
            for (int m = 0; m < 4; ++m) act_tile_fast<ACT>(h[m]);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) act_tile_scaled<ACT>(h[m], ah.c0);
        }

        /* ---- layer 1 ---- */
        BFrag bf;
        h2_make_b<BF>(bf, h);
        zero4(acc);
        h2_gemm_run<BF>(acc, ring, bf, ah.A1, lane);
        h2_gemm_start<BF>(ring, ah.A2, lane);
        if constexpr (SAVE) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] *= ah.c1;
            h2_store_rows128(acc, ah.z1, s_p, b0, rows, lane);
#pragma unroll
            for (int m = 0; m < 4; ++m) act_tile_fast<ACT>(acc[m]);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) act_tile_scaled<ACT>(acc[m], ah.c1);
        }
        h2_make_b<BF>(bf, acc);                         /* the layer-2 B operands, shared by all chunks */

        /* ---- layer 2 in chunks of 128 packed columns + spline:
         *   GEMM(0);  for c: { h -> LDS;  request A(c+1);  spline(c);  GEMM(c+1) } ---- */
        float run = 0.0f;
        int oob_local = 0;
        zero4(h);
        h2_gemm_run<BF>(h, ring, bf, ah.A2, lane);
        for (int c = 0; c < a.n_chunks; ++c) {
#if !(BGK_ABL & 8)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_p[drow(m, r, hh) * ST + j] = h[m][r] * ah.c2;
#else
            s_p[lane] = h[0][0] + h[1][1] + h[2][2] + h[3][3];
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (SAVE) {
                /* parameters of this chunk -> params[b][col]: lane = packed row (two passes of 64), one sample row per
                 * store instruction: contiguous 32-byte runs (the 8 bins of a (dim, component)) instead of a 32-line scatter */
                const int col_lo = ah.src_col[c * 128 + lane], col_hi = ah.src_col[c * 128 + 64 + lane];
                for (int jj = 0; jj < rows; ++jj) {
                    float* prow = ah.params + (b0 + jj) * ah.ldp;
                    if (col_lo >= 0) prow[col_lo] = s_p[lane * ST + jj];
                    if (col_hi >= 0) prow[col_hi] = s_p[(64 + lane) * ST + jj];
                }
            }
            const int nd = (d - c * DPCT) < DPCT ? (d - c * DPCT) : DPCT;
            const uint4* Wn = ah.A2 + (size_t)(c + 1) * H2_BLOCKS * 64;
            const bool more = c + 1 < a.n_chunks;
            if (more) h2_gemm_start<BF>(ring, Wn, lane);
            if constexpr (KT == KB) {
                int bins[3] = {0, 0, 0};
                NoGemm g;
#if !(BGK_ABL & 1)
                spline_chunk<INV, NoGemm, ST, true>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
#else
                run += s_p[(lane & 127) * ST + j];
#endif
                if (a.bin_idx) {
#pragma unroll
                    for (int it = 0; it < 3; ++it) {
                        const int q = 2 * it + hh;
                        if (q < nd && j < rows) a.bin_idx[(b0 + j) * d + c * DPC + q] = bins[it];
                    }
                }
            } else {
                spline_chunk_k<INV, KT, ST, true>(a, s_p, s_y, c, nd, hh, j, rows, b0, run, oob_local);
            }
            if (more) {
#if !(BGK_ABL & 2)
                zero4(h);
                h2_gemm_run<BF>(h, ring, bf, Wn, lane);
#else
                h[0][0] += ring.f[0].v[0][0].x;
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (hh == 0 && j < rows) {
            if (a.accumulate) a.dlogp[b0 + j] += run; else a.dlogp[b0 + j] = run;
        }
        for (int i = lane; i < rows * d; i += 64) {
            const int r = i / d, cc = i - r * d;
            a.out[(b0 + r) * a.ldo + cc] = s_y[cc * SROW + r];
        }
        if (a.oob_count) {
            for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
            if (lane == 0 && oob_local) atomicAdd(a.oob_count, oob_local);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


/* ---- conditioners with hidden layers of 129 .. 256 units (zero-padded to 256): coupling_rqs_dense_w256_kernel ---------------------
 * The same decomposition -- a wave owns 32 samples for the whole layer, the previous layer's accumulators are the next GEMM's B
 * operand without data movement -- at twice the width: 8 accumulator tiles (128 registers) + 16 k16-steps of hi / lo B operands
 * (128 registers) + the A ring.  That is more than the 256 registers two waves per SIMD leave each other, so the kernel runs ONE wave
 * per SIMD on the unified 512-entry file (launch bound 1).  Every GEMM produces 128 output rows at a time (four tiles, as in the
 * width-128 kernel): layer 0 and layer 1 run as two 128-row halves over the same B operand, the parameter chunks as before.  Packed
 * operands (dense.py::pack_dense_for_fused_w256): per 128-row GEMM the blocks (s, m, p) of the width-128 layout with s = 0 .. 15,
 * followed by the 4 bias blocks; k order of the hidden layers = the accumulator layout of 8 tiles.  Inference only (both directions):
 * training of such a layer runs the conditioner layer by layer. */
constexpr int W_T = 8;                           /* hidden tiles: 256 units */
constexpr int W_STEPS = 16;                      /* k16-steps over 256 inputs */
constexpr int W_BLOCKS = W_STEPS * 4 * 2 + 4;    /* 1 KiB blocks of one 128-row GEMM over 256 inputs, incl. bias */
#ifndef BGK_W_RING
#define BGK_W_RING 3
#endif
constexpr int W_RING = BGK_W_RING;                        /* A fragments in flight: two k-steps (24 MFMAs) ahead of their use */
struct BFragW { h16x8 hi[W_STEPS], lo[W_STEPS]; };

template <int O, int N>
__device__ __forceinline__ void w_mfma(f32x16 (&out)[N], const AFrag& a, const h16x8& bhi, const h16x8& blo) {
#pragma unroll
    for (int m = 0; m < 4; ++m) out[O + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a.v[m][1]), bhi, out[O + m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) out[O + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a.v[m][0]), blo, out[O + m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) out[O + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a.v[m][0]), bhi, out[O + m], 0, 0, 0);
}

__device__ __forceinline__ void w_make_b(BFragW& b, const f32x16 (&in)[W_T]) {
#pragma unroll
    for (int s = 0; s < W_STEPS; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in[s >> 1][8 * (s & 1) + e];
        h2_split<false>(v, b.hi[s], b.lo[s]);
    }
}

/* out[O .. O + 4) = W' b + b': 128 output rows over the 256 inputs held in b; N = tiles of the out array */
template <int O, int N>
__device__ __forceinline__ void w_gemm(f32x16 (&out)[N], const BFragW& b, const uint4* W, int lane) {
    AFrag ring[W_RING];
#pragma unroll
    for (int s = 0; s < W_RING - 1; ++s) h2_load<false>(ring[s], W, s, lane);
#pragma unroll
    for (int s = 0; s < W_STEPS; ++s) {
        constexpr int D = W_RING - 1;
        if (s + D < W_STEPS) h2_load<false>(ring[(s + D) % W_RING], W, s + D, lane);
        else if (s + D == W_STEPS) {
#pragma unroll
            for (int m = 0; m < 4; ++m) ring[(s + D) % W_RING].v[m][0] = W[(W_STEPS * 8 + m) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);   /* keep the prefetch above this step's MFMAs */
        w_mfma<O, N>(out, ring[s % W_RING], b.hi[s], b.lo[s]);
    }
    const h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 4; ++m)
        out[O + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ring[W_STEPS % W_RING].v[m][0]), one2, out[O + m], 0, 0, 0);
}

/* activation of x = t * c, the type chosen at run time (wave-uniform): one kernel instance per (direction, bin count) */
__device__ __forceinline__ void w_act(f32x16 (&t)[W_T], float c, int act) {
#pragma unroll
    for (int m = 0; m < W_T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[m][r] *= c;
    if (act == 1) {
#pragma unroll
        for (int m = 0; m < W_T; ++m) act_tile_fast<1>(t[m]);
    } else if (act == 2) {
#pragma unroll
        for (int m = 0; m < W_T; ++m) act_tile_fast<2>(t[m]);
    } else {
#pragma unroll
        for (int m = 0; m < W_T; ++m) act_tile_fast<3>(t[m]);
    }
}

template <int INV, int KT>
__global__ __launch_bounds__(FTHREADS, 1) void coupling_rqs_dense_w256_kernel(FusedArgsH2 ah) {
    constexpr int DPCT = 128 / (3 * KT + 1);              /* dims per 128-column parameter chunk */
    constexpr int ST = 32;
    const FusedArgs& a = ah.f;
    if (ah.cs_dev) { ah.c0 = ah.cs_dev[1]; ah.c1 = ah.cs_dev[3]; ah.c2 = ah.cs_dev[5]; }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* s_p = smem + (size_t)wave * a.lds_per_wave;
    float* s_y = s_p + 128 * ST;
    const int d = a.d;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * FW + wave;
    if (tile >= n_tiles) return;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);

    /* ---- stage the (featurised) conditioner input [feature][sample], a constant-1 row for the bias, zero pad rows ---- */
    const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
    for (int i = lane; i < 32 * a.d_c; i += 64) {
        const int r = i / a.d_c, c = i - r * a.d_c;
        float v = r < rows ? a.cond[(b0 + r) * a.ldc + c] : 0.0f;
        if (a.periodic) {
            float sv, cv;
            bgk_sincos2pif(v, &sv, &cv);
            s_p[c * SROW + r] = cv;
            s_p[(a.d_c + c) * SROW + r] = sv;
        } else {
            s_p[c * SROW + r] = v;
        }
    }
    for (int i = lane; i < (16 * ah.S0 - n_in) * 32; i += 64)
        s_p[(n_in + (i >> 5)) * SROW + (i & 31)] = (i >> 5) == 0 ? 1.0f : 0.0f;
    for (int i = lane; i < 32 * d; i += 64) {
        const int r = i / d, c = i - r * d;
        s_y[c * SROW + r] = r < rows ? a.y[(b0 + r) * a.ldy + c] : 0.5f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    /* ---- layer 0: both 128-row halves per k-step (bias = weight column of the constant-1 feature), next step's A fragments in flight ---- */
    f32x16 h[W_T];
#pragma unroll
    for (int m = 0; m < W_T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = 0.0f;
    {
        const uint4* A0b = ah.A0 + (size_t)ah.S0 * 8 * 64;      /* second half: rows 128 .. 255 */
        AFrag fa, fb;
        h2_load<false>(fa, ah.A0, 0, lane);
        h2_load<false>(fb, A0b, 0, lane);
        for (int s = 0; s < ah.S0; ++s) {
            AFrag na = fa, nb = fb;
            if (s + 1 < ah.S0) {
                h2_load<false>(na, ah.A0, s + 1, lane);
                h2_load<false>(nb, A0b, s + 1, lane);
            }
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = s_p[(16 * s + 8 * hh + e) * SROW + j];
            h16x8 bhi, blo;
            h2_split<false>(v, bhi, blo);
            w_mfma<0, W_T>(h, fa, bhi, blo);
            w_mfma<4, W_T>(h, fb, bhi, blo);
            fa = na; fb = nb;
        }
    }
    w_act(h, ah.c0, a.act);

    /* ---- layer 1: two 128-row halves over the same B operand ---- */
    BFragW bf;
    w_make_b(bf, h);
#pragma unroll
    for (int m = 0; m < W_T; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = 0.0f;
    w_gemm<0, W_T>(h, bf, ah.A1, lane);
    w_gemm<4, W_T>(h, bf, ah.A1 + (size_t)W_BLOCKS * 64, lane);
    w_act(h, ah.c1, a.act);
    w_make_b(bf, h);                                /* the layer-2 B operands, shared by all chunks */

    /* ---- layer 2 in chunks of 128 packed columns + spline ---- */
    float run = 0.0f;
    int oob_local = 0;
    for (int c = 0; c < a.n_chunks; ++c) {
        f32x16 p[4];
        zero4(p);
        w_gemm<0, 4>(p, bf, ah.A2 + (size_t)c * W_BLOCKS * 64, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_p[drow(m, r, hh) * ST + j] = p[m][r] * ah.c2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nd = (d - c * DPCT) < DPCT ? (d - c * DPCT) : DPCT;
        if constexpr (KT == KB) {
            int bins[3] = {0, 0, 0};
            NoGemm g;
            spline_chunk<INV, NoGemm, ST, true>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
            if (a.bin_idx) {
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int q = 2 * it + hh;
                    if (q < nd && j < rows) a.bin_idx[(b0 + j) * d + c * DPC + q] = bins[it];
                }
            }
        } else {
            spline_chunk_k<INV, KT, ST, true>(a, s_p, s_y, c, nd, hh, j, rows, b0, run, oob_local);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (hh == 0 && j < rows) {
        if (a.accumulate) a.dlogp[b0 + j] += run; else a.dlogp[b0 + j] = run;
    }
    for (int i = lane; i < rows * d; i += 64) {
        const int r = i / d, cc = i - r * d;
        a.out[(b0 + r) * a.ldo + cc] = s_y[cc * SROW + r];
    }
    if (a.oob_count) {
        for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
        if (lane == 0 && oob_local) atomicAdd(a.oob_count, oob_local);
    }
}


/* ---- conditioners with ANY number of hidden layers (width <= 128, zero-padded): coupling_rqs_dense_deep_kernel ----------------------
 * conditioner_factory.py:76-80 takes any `hidden` tuple; the second-generation kernel is written for two hidden layers.  This is the
 * first-generation decomposition with the hidden -> hidden GEMM in a loop over n_hh = n_hidden - 1 packed layers (A1 = their operands
 * back to back, c1s[l] their unscale factors), n_hh = 0 (one hidden layer) .. DEEP_MAX_HH.  Two waves per SIMD; activation chosen at run
 * time.  Inference only (both directions): training of such a layer runs the conditioner layer by layer. */
constexpr int DEEP_MAX_HH = 7;
struct DeepArgs { FusedArgsH2 h; int n_hh; float c1s[DEEP_MAX_HH]; };

__device__ __forceinline__ void deep_act(f32x16 (&t)[4], float c, int act) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[m][r] *= c;
    if (act == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m) act_tile_fast<1>(t[m]);
    } else if (act == 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) act_tile_fast<2>(t[m]);
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) act_tile_fast<3>(t[m]);
    }
}

template <int INV, int KT>
__global__ __launch_bounds__(FTHREADS, 2) void coupling_rqs_dense_deep_kernel(DeepArgs da) {
    constexpr int DPCT = 128 / (3 * KT + 1);              /* dims per 128-column parameter chunk */
    constexpr int ST = 32;
    FusedArgsH2& ah = da.h;
    const FusedArgs& a = ah.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* s_p = smem + (size_t)wave * a.lds_per_wave;
    float* s_y = s_p + 128 * ST;
    const int d = a.d;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * FW + wave;
    if (tile >= n_tiles) return;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);

    /* ---- stage the (featurised) conditioner input [feature][sample], a constant-1 row for the bias, zero pad rows ---- */
    const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
    for (int i = lane; i < 32 * a.d_c; i += 64) {
        const int r = i / a.d_c, c = i - r * a.d_c;
        float v = r < rows ? a.cond[(b0 + r) * a.ldc + c] : 0.0f;
        if (a.periodic) {
            float sv, cv;
            bgk_sincos2pif(v, &sv, &cv);
            s_p[c * SROW + r] = cv;
            s_p[(a.d_c + c) * SROW + r] = sv;
        } else {
            s_p[c * SROW + r] = v;
        }
    }
    for (int i = lane; i < (16 * ah.S0 - n_in) * 32; i += 64)
        s_p[(n_in + (i >> 5)) * SROW + (i & 31)] = (i >> 5) == 0 ? 1.0f : 0.0f;
    for (int i = lane; i < 32 * d; i += 64) {
        const int r = i / d, c = i - r * d;
        s_y[c * SROW + r] = r < rows ? a.y[(b0 + r) * a.ldy + c] : 0.5f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    /* ---- layer 0 (bias = weight column of the constant-1 feature) ---- */
    f32x16 h[4];
    zero4(h);
    for (int s = 0; s < ah.S0; ++s) {
        AFrag fr;
        h2_load<false>(fr, ah.A0, s, lane);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = s_p[(16 * s + 8 * hh + e) * SROW + j];
        h16x8 bhi, blo;
        h2_split<false>(v, bhi, blo);
        h2_mfma<false>(h, fr, bhi, blo);
    }
    deep_act(h, ah.c0, a.act);

    /* ---- hidden -> hidden layers ---- */
    BFrag bf;
    H2Ring ring;
    for (int l = 0; l < da.n_hh; ++l) {
        const uint4* Wl = ah.A1 + (size_t)l * H2_BLOCKS * 64;
        h2_gemm_start<false>(ring, Wl, lane);
        h2_make_b<false>(bf, h);
        zero4(h);
        h2_gemm_run<false>(h, ring, bf, Wl, lane);
        deep_act(h, da.c1s[l], a.act);
    }
    h2_make_b<false>(bf, h);                          /* the output layer's B operands, shared by all chunks */

    /* ---- output layer in chunks of 128 packed columns + spline ---- */
    float run = 0.0f;
    int oob_local = 0;
    for (int c = 0; c < a.n_chunks; ++c) {
        const uint4* Wc = ah.A2 + (size_t)c * H2_BLOCKS * 64;
        h2_gemm_start<false>(ring, Wc, lane);
        zero4(h);
        h2_gemm_run<false>(h, ring, bf, Wc, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_p[drow(m, r, hh) * ST + j] = h[m][r] * ah.c2;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nd = (d - c * DPCT) < DPCT ? (d - c * DPCT) : DPCT;
        if constexpr (KT == KB) {
            int bins[3] = {0, 0, 0};
            NoGemm g;
            spline_chunk<INV, NoGemm, ST, true>(g, a, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
            if (a.bin_idx) {
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int q = 2 * it + hh;
                    if (q < nd && j < rows) a.bin_idx[(b0 + j) * d + c * DPC + q] = bins[it];
                }
            }
        } else {
            spline_chunk_k<INV, KT, ST, true>(a, s_p, s_y, c, nd, hh, j, rows, b0, run, oob_local);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (hh == 0 && j < rows) {
        if (a.accumulate) a.dlogp[b0 + j] += run; else a.dlogp[b0 + j] = run;
    }
    for (int i = lane; i < rows * d; i += 64) {
        const int r = i / d, cc = i - r * d;
        a.out[(b0 + r) * a.ldo + cc] = s_y[cc * SROW + r];
    }
    if (a.oob_count) {
        for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
        if (lane == 0 && oob_local) atomicAdd(a.oob_count, oob_local);
    }
}

}  // namespace

extern "C" int32_t bgk_pack_rqs_columns(int32_t d, int32_t K, const int32_t* nc_slot_host, int32_t* src_col) {
    if (d <= 0 || K <= 0 || 3 * K + 1 > 128) return BGK_EINVAL;
    const int ppd = 3 * K + 1, dpc = 128 / ppd;
    const int n_chunks = (d + dpc - 1) / dpc;
    const int ncp = n_chunks * 128;
    if (src_col) {
        for (int i = 0; i < ncp; ++i) src_col[i] = -1;
        for (int jd = 0; jd < d; ++jd) {
            const int c = jd / dpc, q = jd - c * dpc;
            int32_t* dst = src_col + c * 128 + q * ppd;
            for (int k = 0; k < K; ++k) {
                dst[k] = jd * K + k;
                dst[K + k] = d * K + jd * K + k;
                dst[2 * K + k] = 2 * d * K + jd * K + k;
            }
            const int slot = nc_slot_host ? nc_slot_host[jd] : -1;
            dst[3 * K] = slot >= 0 ? 3 * d * K + slot : -1;
        }
    }
    return ncp;
}

extern "C" int bgk_coupling_rqs_dense(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                      const float* W0p, const float* W1p, const float* W2p,
                                      int32_t H0, int32_t H1, int32_t act, const float* y,
                                      int64_t ldy, int64_t B, int32_t d, int32_t K, uint64_t circ_mask,
                                      int32_t inverse,
                                      double left, double right, double bottom, double top,
                                      double min_bin_width, double min_bin_height,
                                      double min_derivative, int32_t identity_init, float* out,
                                      int64_t ldo, float* dlogp, int32_t accumulate,
                                      int32_t* bin_idx, int32_t* oob_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(cond && W0p && W1p && W2p && y && out && dlogp, "bgk_coupling_rqs_dense: null pointer");
    BGK_CHECK_ARG(B >= 0 && d > 0 && d_c > 0, "bgk_coupling_rqs_dense: bad sizes");
    if (H0 != HID || H1 != HID || K != KB || d > 64 || act < 1 || act > 3) {
        bgk_set_error("bgk_coupling_rqs_dense: only hidden=(128,128), n_bins=8, d<=64, act in {SiLU,ReLU,Tanh} are fused "
                      "(got H0=%d H1=%d K=%d d=%d act=%d)", H0, H1, K, d, act);
        return BGK_EUNSUPPORTED;
    }
    const int n_in = periodic ? 2 * d_c : d_c;
    if ((n_in + 9) * SROW > LDS_P) {     /* outside the envelope, not an error: the caller runs the conditioner layer by layer */
        bgk_set_error("bgk_coupling_rqs_dense: conditioner input of %d features does not fit the layer-0 tile", n_in);
        return BGK_EUNSUPPORTED;
    }
    BGK_CHECK_ARG(min_bin_width * K <= 1.0 && min_bin_height * K <= 1.0,
                  "Minimal bin width/height too large for the number of bins");
    if (B == 0) return 0;
    FusedArgs a;
    a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic;
    a.W0 = reinterpret_cast<const float4*>(W0p); a.T0 = ((n_in + 1) / 2 + 3) & ~3;   /* padded to x4 by the packer */
    a.W1 = reinterpret_cast<const float4*>(W1p);
    a.W2 = reinterpret_cast<const float4*>(W2p); a.n_chunks = (d + DPC - 1) / DPC;
    a.last_tiles = ((d - (a.n_chunks - 1) * DPC) * PPD + 31) / 32;
    a.act = act; a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.inverse = inverse;
    a.circ_mask = circ_mask;
    a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.bin_idx = bin_idx; a.oob_count = oob_count;
    a.lds_per_wave = LDS_P + (d + 1) * SROW;   /* + dummy row for discarded spline slots */
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
    size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
    int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_coupling_rqs_dense: batch too large for one launch");
    int grid = (int)n_wg;
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH(A, I) hipLaunchKernelGGL((coupling_rqs_dense_kernel<A, I>), dim3(grid), dim3(FTHREADS), shmem, st, a)
    if (act == 1) { if (inverse) BGK_LAUNCH(1, 1); else BGK_LAUNCH(1, 0); }
    else if (act == 2) { if (inverse) BGK_LAUNCH(2, 1); else BGK_LAUNCH(2, 0); }
    else { if (inverse) BGK_LAUNCH(3, 1); else BGK_LAUNCH(3, 0); }
#undef BGK_LAUNCH
    return bgk_launch_status("bgk_coupling_rqs_dense");
}

namespace {
int launch_h2(const char* what, const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
              const void* A0p, const void* A1p, const void* A2p, float c0, float c1, float c2, const float* cs_dev,
              int32_t operand_dtype, int32_t H0, int32_t H1, int32_t act, const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K,
              uint64_t circ_mask, int32_t inverse, double left, double right, double bottom, double top,
              double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
              float* out, int64_t ldo, float* dlogp, int32_t accumulate, int32_t* bin_idx, int32_t* oob_count,
              float* z0, float* z1, float* params, int64_t ldp, const int32_t* src_col, void* stream, const BgkCondSegs* segs = nullptr,
              int params_layout = 0) {
    if (segs && segs->n >= 1) { cond = segs->ptr[0]; ldc = segs->ld[0]; }
    BGK_CHECK_ARG(cond && A0p && A1p && A2p && y && out && dlogp, "%s: null pointer", what);
    BGK_CHECK_ARG(B >= 0 && d > 0 && d_c > 0, "%s: bad sizes", what);
    const bool other_k = (K == 4 || K == 12 || K == 16 || K == 32) && operand_dtype == 0;     /* K != 8: split-f16 only (inference and training forward) */
    const bool wide = H0 == 32 * W_T && H1 == 32 * W_T;      /* hidden width 256 (129 .. 255 zero-padded by the packer): split-f16 inference */
    if (wide && (operand_dtype != 0 || z0 || z1 || params || (segs && segs->n > 1))) {
        bgk_set_error("%s: hidden width 256 runs fused in split-f16 inference from one conditioning tensor only", what);
        return BGK_EUNSUPPORTED;
    }
    if ((!wide && (H0 != HID || H1 != HID)) || (K != KB && !other_k) || d > 64 || act < 1 || act > 3) {
        bgk_set_error("%s: only hidden=(128,128), n_bins=8 (4 | 12 | 16 | 32: split-f16 form only), d<=64, act in {SiLU,ReLU,Tanh} are fused "
                      "(got H0=%d H1=%d K=%d d=%d act=%d)", what, H0, H1, K, d, act);
        return BGK_EUNSUPPORTED;
    }
    const int n_in = periodic ? 2 * d_c : d_c;
    const int S0 = (n_in + 1 + 15) / 16;
    if (16 * S0 * SROW > LDS_P) {        /* outside the envelope, not an error: the caller runs the conditioner layer by layer */
        bgk_set_error("%s: conditioner input of %d features does not fit the layer-0 tile (at most 111)", what, n_in);
        return BGK_EUNSUPPORTED;
    }
    BGK_CHECK_ARG(min_bin_width * K <= 1.0 && min_bin_height * K <= 1.0,
                  "Minimal bin width/height too large for the number of bins");
    if (B == 0) return 0;
    if (wide) {
        FusedArgsH2 ah;
        FusedArgs& a = ah.f;
        a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic;
        a.W0 = nullptr; a.W1 = nullptr; a.W2 = nullptr; a.T0 = 0;
        const int ppd_k = 3 * K + 1, dpc_k = 128 / ppd_k;
        a.n_chunks = (d + dpc_k - 1) / dpc_k;
        a.last_tiles = ((d - (a.n_chunks - 1) * dpc_k) * ppd_k + 31) / 32;
        a.act = act; a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.inverse = inverse;
        a.circ_mask = circ_mask;
        a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
        a.bin_idx = bin_idx; a.oob_count = oob_count;
        a.lds_per_wave = 128 * 32 + (d + 1) * SROW;
        a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
        ah.A0 = reinterpret_cast<const uint4*>(A0p); ah.S0 = S0;
        ah.A1 = reinterpret_cast<const uint4*>(A1p);
        ah.A2 = reinterpret_cast<const uint4*>(A2p);
        ah.c0 = c0; ah.c1 = c1; ah.c2 = c2; ah.cs_dev = cs_dev;
        ah.z0 = nullptr; ah.z1 = nullptr; ah.params = nullptr; ah.ldp = 0; ah.src_col = nullptr;
        const size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
        const int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
        BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "%s: batch too large for one launch", what);
        const int grid = (int)n_wg;
        hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCHW(I, KK) hipLaunchKernelGGL((coupling_rqs_dense_w256_kernel<I, KK>), dim3(grid), dim3(FTHREADS), shmem, st, ah)
#define BGK_LAUNCHW2(KK) do { if (inverse) BGK_LAUNCHW(1, KK); else BGK_LAUNCHW(0, KK); } while (0)
        if (K == 8) BGK_LAUNCHW2(8); else if (K == 4) BGK_LAUNCHW2(4); else if (K == 12) BGK_LAUNCHW2(12);
        else if (K == 16) BGK_LAUNCHW2(16); else BGK_LAUNCHW2(32);
#undef BGK_LAUNCHW2
#undef BGK_LAUNCHW
        return bgk_launch_status(what);
    }
    /* second-generation kernels: their staging index math uses 24-bit multiplies (row strides below 2^24 floats) */
    const bool v2_ok = bgk_h2_variant == 2 && K == KB && ldc < (1 << 24) && ldy < (1 << 24) && ldo < (1 << 24);
    if (segs && segs->n > 1 && !v2_ok) return BGK_EUNSUPPORTED;     /* several conditioning tensors: second-generation kernels only */
    if (params_layout == 1 && !(v2_ok && z0 != nullptr && operand_dtype == 0 && params && z1)) return BGK_EUNSUPPORTED;   /* element-major parameters: second-generation kernel only */
    if (params_layout == 2 && !(v2_ok && z0 != nullptr && operand_dtype == 0 && z1)) return BGK_EUNSUPPORTED;             /* no parameter write-out: likewise */
    if (v2_ok && z0 != nullptr && operand_dtype == 0 && (src_col || params_layout >= 1) && (params || params_layout == 2) && z1)   /* training forward */
        return bgk_launch_rqs_dense_h2v2_train(what, z0, z1, params_layout == 2 ? nullptr : params, ldp, params_layout >= 1 ? nullptr : src_col, cond, ldc, d_c, periodic, A0p, A1p, A2p, c0, c1, c2, cs_dev,
                                               act, y, ldy, B, d, circ_mask, inverse, left, right, bottom, top, min_bin_width,
                                               min_bin_height, min_derivative, identity_init, out, ldo, dlogp, accumulate, bin_idx,
                                               oob_count, stream, segs);
    if (v2_ok && z0 == nullptr && operand_dtype == 1)   /* reduced-precision bf16 mode on the second-generation kernel */
        return bgk_launch_rqs_dense_h2v2_bf16(what, cond, ldc, d_c, periodic, A0p, A1p, A2p, c0, c1, c2, cs_dev, act, y, ldy, B, d, circ_mask,
                                              inverse, left, right, bottom, top, min_bin_width, min_bin_height, min_derivative,
                                              identity_init, out, ldo, dlogp, accumulate, bin_idx, oob_count, stream, segs);
    if (v2_ok && z0 == nullptr && operand_dtype == 0)   /* split-f16 inference: the second-generation kernel */
        return bgk_launch_rqs_dense_h2v2(what, cond, ldc, d_c, periodic, A0p, A1p, A2p, c0, c1, c2, cs_dev, act, y, ldy, B, d, circ_mask,
                                         inverse, left, right, bottom, top, min_bin_width, min_bin_height, min_derivative,
                                         identity_init, out, ldo, dlogp, accumulate, bin_idx, oob_count, stream, segs);
    FusedArgsH2 ah;
    FusedArgs& a = ah.f;
    a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic;
    a.W0 = nullptr; a.W1 = nullptr; a.W2 = nullptr; a.T0 = 0;
    const int ppd_k = 3 * K + 1, dpc_k = 128 / ppd_k;
    a.n_chunks = (d + dpc_k - 1) / dpc_k;
    a.last_tiles = ((d - (a.n_chunks - 1) * dpc_k) * ppd_k + 31) / 32;
    a.act = act; a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.inverse = inverse;
    a.circ_mask = circ_mask;
    a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.bin_idx = bin_idx; a.oob_count = oob_count;
    const bool save = z0 != nullptr;
    a.lds_per_wave = 128 * (save ? 33 : 32) + (d + 1) * SROW;
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
    ah.A0 = reinterpret_cast<const uint4*>(A0p); ah.S0 = S0;
    ah.A1 = reinterpret_cast<const uint4*>(A1p);
    ah.A2 = reinterpret_cast<const uint4*>(A2p);
    ah.c0 = c0; ah.c1 = c1; ah.c2 = c2; ah.cs_dev = cs_dev;
    ah.z0 = z0; ah.z1 = z1; ah.params = params; ah.ldp = ldp; ah.src_col = src_col;
    size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
#ifdef BGK_DBG_EXTRA_LDS
    shmem += BGK_DBG_EXTRA_LDS;   /* occupancy experiments (tools/ablate_h2.sh): e.g. 20000 -> one workgroup per CU */
#endif
    int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "%s: batch too large for one launch", what);
    int grid = (int)n_wg;
    hipStream_t st = (hipStream_t)stream;
    BGK_CHECK_ARG(operand_dtype == 0 || (operand_dtype == 1 && !save), "%s: operand_dtype %d (0 = split-f16, 1 = bf16; the training "
                  "forward is split-f16 only)", what, operand_dtype);
#define BGK_LAUNCH(A, I, S, F) hipLaunchKernelGGL((coupling_rqs_dense_h2_kernel<A, I, S, F>), dim3(grid), dim3(FTHREADS), shmem, st, ah)
#define BGK_LAUNCH2(A, I) do { if (save) BGK_LAUNCH(A, I, true, false); else if (operand_dtype == 1) BGK_LAUNCH(A, I, false, true); \
                               else BGK_LAUNCH(A, I, false, false); } while (0)
#define BGK_LAUNCHK(A, I, S, KK) hipLaunchKernelGGL((coupling_rqs_dense_h2_kernel<A, I, S, false, KK>), dim3(grid), dim3(FTHREADS), shmem, st, ah)
#define BGK_LAUNCHK3(A, I, KK) do { if (save) BGK_LAUNCHK(A, I, true, KK); else BGK_LAUNCHK(A, I, false, KK); } while (0)
#define BGK_LAUNCHK2(A, I) do { if (K == 4) BGK_LAUNCHK3(A, I, 4); else if (K == 12) BGK_LAUNCHK3(A, I, 12); \
                                else if (K == 16) BGK_LAUNCHK3(A, I, 16); else BGK_LAUNCHK3(A, I, 32); } while (0)
    if (K != KB) {
        if (act == 1) { if (inverse) BGK_LAUNCHK2(1, 1); else BGK_LAUNCHK2(1, 0); }
        else if (act == 2) { if (inverse) BGK_LAUNCHK2(2, 1); else BGK_LAUNCHK2(2, 0); }
        else { if (inverse) BGK_LAUNCHK2(3, 1); else BGK_LAUNCHK2(3, 0); }
    }
    else if (act == 1) { if (inverse) BGK_LAUNCH2(1, 1); else BGK_LAUNCH2(1, 0); }
    else if (act == 2) { if (inverse) BGK_LAUNCH2(2, 1); else BGK_LAUNCH2(2, 0); }
    else { if (inverse) BGK_LAUNCH2(3, 1); else BGK_LAUNCH2(3, 0); }
#undef BGK_LAUNCHK2
#undef BGK_LAUNCHK3
#undef BGK_LAUNCHK
#undef BGK_LAUNCH2
#undef BGK_LAUNCH
    return bgk_launch_status(what);
}
}  // namespace

extern "C" int bgk_coupling_rqs_dense_deep(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                           const void* A0p, const void* A1p, const void* A2p, float c0, const float* c1s, float c2,
                                           int32_t n_hidden, int32_t act, const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K,
                                           uint64_t circ_mask, int32_t inverse,
                                           double left, double right, double bottom, double top,
                                           double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
                                           float* out, int64_t ldo, float* dlogp, int32_t accumulate,
                                           int32_t* bin_idx, int32_t* oob_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    const char* what = "bgk_coupling_rqs_dense_deep";
    BGK_CHECK_ARG(cond && A0p && A2p && y && out && dlogp, "%s: null pointer", what);
    BGK_CHECK_ARG(B > 0 && d > 0 && d_c > 0, "%s: bad sizes", what);
    BGK_CHECK_ARG(n_hidden == 1 || (A1p && c1s), "%s: null hidden-layer operands", what);
    if (n_hidden < 1 || n_hidden > DEEP_MAX_HH + 1 || !(K == 4 || K == 8 || K == 12 || K == 16 || K == 32) || d > 64 || act < 1 || act > 3) {
        bgk_set_error("%s: 1 .. %d hidden layers of width 128, n_bins in {4, 8, 12, 16, 32}, d<=64, act in {SiLU,ReLU,Tanh} are fused "
                      "(got n_hidden=%d K=%d d=%d act=%d)", what, DEEP_MAX_HH + 1, n_hidden, K, d, act);
        return BGK_EUNSUPPORTED;
    }
    const int n_in = periodic ? 2 * d_c : d_c;
    const int S0 = (n_in + 1 + 15) / 16;
    if (16 * S0 * SROW > LDS_P) {        /* outside the envelope, not an error: the caller runs the conditioner layer by layer */
        bgk_set_error("%s: conditioner input of %d features does not fit the layer-0 tile (at most 111)", what, n_in);
        return BGK_EUNSUPPORTED;
    }
    BGK_CHECK_ARG(min_bin_width * K <= 1.0 && min_bin_height * K <= 1.0, "Minimal bin width/height too large for the number of bins");
    DeepArgs da;
    FusedArgsH2& ah = da.h;
    FusedArgs& a = ah.f;
    a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic;
    a.W0 = nullptr; a.W1 = nullptr; a.W2 = nullptr; a.T0 = 0;
    const int ppd_k = 3 * K + 1, dpc_k = 128 / ppd_k;
    a.n_chunks = (d + dpc_k - 1) / dpc_k;
    a.last_tiles = ((d - (a.n_chunks - 1) * dpc_k) * ppd_k + 31) / 32;
    a.act = act; a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.inverse = inverse;
    a.circ_mask = circ_mask;
    a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.bin_idx = bin_idx; a.oob_count = oob_count;
    a.lds_per_wave = 128 * 32 + (d + 1) * SROW;
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
    ah.A0 = reinterpret_cast<const uint4*>(A0p); ah.S0 = S0;
    ah.A1 = reinterpret_cast<const uint4*>(A1p);
    ah.A2 = reinterpret_cast<const uint4*>(A2p);
    ah.c0 = c0; ah.c1 = 1.0f; ah.c2 = c2; ah.cs_dev = nullptr;
    ah.z0 = nullptr; ah.z1 = nullptr; ah.params = nullptr; ah.ldp = 0; ah.src_col = nullptr;
    da.n_hh = n_hidden - 1;
    for (int l = 0; l < DEEP_MAX_HH; ++l) da.c1s[l] = l < da.n_hh ? c1s[l] : 1.0f;
    const size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
    const int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "%s: batch too large for one launch", what);
    const int grid = (int)n_wg;
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCHD(I, KK) hipLaunchKernelGGL((coupling_rqs_dense_deep_kernel<I, KK>), dim3(grid), dim3(FTHREADS), shmem, st, da)
#define BGK_LAUNCHD2(KK) do { if (inverse) BGK_LAUNCHD(1, KK); else BGK_LAUNCHD(0, KK); } while (0)
    if (K == 8) BGK_LAUNCHD2(8); else if (K == 4) BGK_LAUNCHD2(4); else if (K == 12) BGK_LAUNCHD2(12);
    else if (K == 16) BGK_LAUNCHD2(16); else BGK_LAUNCHD2(32);
#undef BGK_LAUNCHD2
#undef BGK_LAUNCHD
    return bgk_launch_status(what);
}

extern "C" int bgk_coupling_rqs_dense_h2(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                         const void* A0p, const void* A1p, const void* A2p,
                                         float c0, float c1, float c2, const float* cs_dev, int32_t operand_dtype,
                                         int32_t H0, int32_t H1, int32_t act, const float* y,
                                         int64_t ldy, int64_t B, int32_t d, int32_t K, uint64_t circ_mask,
                                         int32_t inverse,
                                         double left, double right, double bottom, double top,
                                         double min_bin_width, double min_bin_height,
                                         double min_derivative, int32_t identity_init, float* out,
                                         int64_t ldo, float* dlogp, int32_t accumulate,
                                         int32_t* bin_idx, int32_t* oob_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    return launch_h2("bgk_coupling_rqs_dense_h2", cond, ldc, d_c, periodic, A0p, A1p, A2p, c0, c1, c2, cs_dev, operand_dtype, H0, H1, act, y, ldy, B, d, K,
                     circ_mask, inverse, left, right, bottom, top, min_bin_width, min_bin_height, min_derivative,
                     identity_init, out, ldo, dlogp, accumulate, bin_idx, oob_count,
                     nullptr, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int bgk_coupling_rqs_dense_h2_mc(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                            const void* A0p, const void* A1p, const void* A2p,
                                            float c0, float c1, float c2, const float* cs_dev, int32_t operand_dtype,
                                            int32_t H0, int32_t H1, int32_t act, const float* y,
                                            int64_t ldy, int64_t B, int32_t d, int32_t K, uint64_t circ_mask,
                                            int32_t inverse,
                                            double left, double right, double bottom, double top,
                                            double min_bin_width, double min_bin_height,
                                            double min_derivative, int32_t identity_init, float* out,
                                            int64_t ldo, float* dlogp, int32_t accumulate,
                                            int32_t* bin_idx, int32_t* oob_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(cond && ldc && width && n_cond >= 1 && n_cond <= BGK_MAX_COND, "bgk_coupling_rqs_dense_h2_mc: 1..%d conditioning tensors", BGK_MAX_COND);
    BgkCondSegs segs{};
    int d_c = 0;
    for (int i = 0; i < n_cond; ++i) { segs.ptr[i] = cond[i]; segs.ld[i] = ldc[i]; segs.w[i] = width[i]; d_c += width[i]; }
    segs.n = n_cond;
    return launch_h2("bgk_coupling_rqs_dense_h2_mc", cond[0], ldc[0], d_c, periodic, A0p, A1p, A2p, c0, c1, c2, cs_dev, operand_dtype, H0, H1, act, y, ldy,
                     B, d, K, circ_mask, inverse, left, right, bottom, top, min_bin_width, min_bin_height, min_derivative,
                     identity_init, out, ldo, dlogp, accumulate, bin_idx, oob_count,
                     nullptr, nullptr, nullptr, 0, nullptr, stream, &segs);
}

extern "C" int bgk_coupling_rqs_dense_h2_train(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                               const void* A0p, const void* A1p, const void* A2p,
                                               float c0, float c1, float c2, const float* cs_dev,
                                               int32_t H0, int32_t H1, int32_t act, const float* y,
                                               int64_t ldy, int64_t B, int32_t d, int32_t K, uint64_t circ_mask,
                                               int32_t inverse,
                                               double left, double right, double bottom, double top,
                                               double min_bin_width, double min_bin_height,
                                               double min_derivative, int32_t identity_init, float* out,
                                               int64_t ldo, float* dlogp, int32_t accumulate,
                                               int32_t* oob_count, float* z0, float* z1, float* params, int64_t ldp,
                                               const int32_t* src_col_dev, int32_t params_layout, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(params_layout >= 0 && params_layout <= 2, "bgk_coupling_rqs_dense_h2_train: params_layout %d (0 = the reference's columns, 1 = element-major, "
                  "2 = not written: bgk_coupling_rqs_dense_h2_backward recomputes them)", params_layout);
    BGK_CHECK_ARG(z0 && z1 && (params || params_layout == 2) && (src_col_dev || params_layout >= 1), "bgk_coupling_rqs_dense_h2_train: null save buffer");
    const int n_nc = d - __builtin_popcountll(circ_mask & (d >= 64 ? ~0ull : ((1ull << d) - 1)));
    BGK_CHECK_ARG(params_layout == 2 || ldp >= (params_layout == 1 ? (3 * K + 1) * d + 3 : 3 * K * d + n_nc), "bgk_coupling_rqs_dense_h2_train: params row stride %lld too small", (long long)ldp);
    return launch_h2("bgk_coupling_rqs_dense_h2_train", cond, ldc, d_c, periodic, A0p, A1p, A2p, c0, c1, c2, cs_dev, 0, H0, H1, act, y, ldy, B, d,
                     K, circ_mask, inverse, left, right, bottom, top, min_bin_width, min_bin_height, min_derivative,
                     identity_init, out, ldo, dlogp, accumulate, nullptr, oob_count, z0, z1, params, ldp, src_col_dev, stream, nullptr, params_layout);
}

/* Backward of the spline transformer of a layer whose training forward ran with params_layout = 2 (no parameter write-out): the
 * parameters are recomputed from the saved z1 [B, 128] (contiguous) with the forward's packed output-layer operand A2p / scale c2
 * (cs_dev: the forward's device scale table or NULL; circ_mask as there), then the VJP of bgk_rqs_backward.  g_params [B, P] comes out in the
 * reference's column order (slot of a non-circular dim = its rank among them), g_y [B, d]; g_absmax as in bgk_rqs_backward.  Fused envelope: hidden width 128, 8 bins, d <= 64. */
extern "C" int bgk_coupling_rqs_dense_h2_backward(const float* z1, const void* A2p, float c2, const float* cs_dev, int32_t H1, int32_t act,
                                                  const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K, int32_t P,
                                                  uint64_t circ_mask, int32_t inverse,
                                                  double left, double right, double bottom, double top,
                                                  double min_bin_width, double min_bin_height, double min_derivative,
                                                  int32_t identity_init, const float* g_out, int64_t ldgo, const float* g_dlogp,
                                                  float* g_y, int64_t ldgy, float* g_params, int64_t ldgp, float* g_absmax, void* stream) {
    if (B == 0) return 0;
    const char* what = "bgk_coupling_rqs_dense_h2_backward";
    BGK_CHECK_ARG(z1 && A2p && y && g_out && g_dlogp && g_y && g_params, "%s: null pointer", what);
    BGK_CHECK_ARG(B > 0 && d > 0, "%s: bad sizes", what);
    if (H1 != HID || K != KB || d > 64 || act < 1 || act > 3 || bgk_h2_variant != 2) {
        bgk_set_error("%s: only hidden width 128, n_bins=8, d<=64, act in {SiLU,ReLU,Tanh} (got H1=%d K=%d d=%d act=%d)", what, H1, K, d, act);
        return BGK_EUNSUPPORTED;
    }
    const int n_nc = d - __builtin_popcountll(circ_mask & (d >= 64 ? ~0ull : ((1ull << d) - 1)));
    BGK_CHECK_ARG(P == 3 * K * d + n_nc && ldgp >= P && ldy >= d && ldgo >= d && ldgy >= d, "%s: bad widths / row strides", what);
    BGK_CHECK_ARG(((uintptr_t)z1 & 15) == 0, "%s: z1 must be 16-byte aligned", what);
    BGK_CHECK_ARG(min_bin_width * K <= 1.0 && min_bin_height * K <= 1.0, "Minimal bin width/height too large for the number of bins");
    return bgk_launch_rqs_bwd_recompute(what, z1, A2p, c2, cs_dev, act, y, ldy, B, d, circ_mask, inverse, left, right, bottom, top,
                                        min_bin_width, min_bin_height, min_derivative, identity_init, g_out, ldgo, g_dlogp, g_y, ldgy,
                                        g_params, ldgp, g_absmax, stream);
}
