/* bgk_fused2.hip -- second-generation one-launch spline coupling layer (inference, split-f16 conditioner GEMMs).
 *
 * Same contract, same packed operands (dense.py::pack_dense_for_fused_h2 / bgk_pack_dense_h2) and the same tiling as
 * coupling_rqs_dense_h2_kernel (bgk_fused.hip): a wave owns 32 samples for the whole layer, 4 independent waves per
 * workgroup, 2 workgroups per CU.  What changed, and why (measurements: tools/ubench/issue_bench, profiles/README.md r02):
 *
 *  1. MFMA and VALU time were ADDITIVE in the first kernel (30 % matrix pipe busy + 52 % VALU active + waits).  On
 *     gfx950 a wave issues one instruction per ~4-5 cycles, the f16 MFMA holds the matrix pipe for 32 cycles, and plain
 *     (non-packed) f32 VALU work placed in the same wave between consecutive MFMAs hides ~60-75 % of the smaller of
 *     the two -- packed-f32 VALU ops (v_pk_*) hide nothing.  So every GEMM after layer 0 is issued as a stream of
 *     single MFMA "events" threaded through the VALU work that is independent of it:
 *        layer-1 GEMM          <-> activation + f16 hi/lo split of the layer-0 tiles 1..3   (k-step s needs tile s/2 only)
 *        layer-2 chunk-0 GEMM  <-> activation + split of the layer-1 tiles 1..3
 *        layer-2 chunk c+1     <-> spline of chunk c (3 elements per lane, 34 hook points each)
 *     The placement is pinned with sched_barrier(0) fences; the weight (A) fragments are loaded per (k-step, tile)
 *     through a 4-deep register ring (32 VGPRs instead of 64), SGPR-base addressing, no per-load address arithmetic.
 *  2. Fewer VALU instructions (first kernel: 6.3 k per 32-sample tile; packed ops avoided altogether):
 *     - the spline's normalisation is algebraically regrouped: knot_k = low + span*min*(k+1) + (span*scale / sum e) * prefix_k(e):
 *       7 adds + 7 fma per set instead of 8 exactly rounded divisions + scale + cumsum + affine map; the other set only
 *       evaluates the two knots of the bin found (select chains on the comparison masks);
 *     - the exact power-of-two unscale of the last GEMM is folded into the exp2 / softplus constants;
 *     - accumulators start from the inline-constant 0 C operand (no zero fills), dead tiles of the last chunk are
 *       not computed, staging index math uses a magic-number division.
 *  Accuracy class: like the first split-f16 kernel (hardware exp2 / log2 / rcp / sqrt with one Newton step on the
 *  reciprocals): not bit-identical to the oracle; parity is asserted per sample at 1e-5 relative on log|det J| and
 *  bin indices may differ only where x is within rounding distance of a knot (tests/test_gpu_parity.py).
 *  The exact-f32 kernel (bgk_fused.hip, gemm_mode "f32") stays the bit-exact reference path.
 */
#include "bgk_common.h"
#include "bgk_fused2.h"

#ifndef BGK_V2_SAVE
#define BGK_V2_SAVE 0                /* 1 (bgk_fused2_train.hip): the training forward -- the same kernel also writes what the backward
                                      * needs: the scaled pre-activations z0, z1 [B, 128] and the spline parameters [B, P] */
#endif
#if BGK_V2_SAVE
#include "bgk_mfma_h2.h"             /* h2_store_rows128 */
#include "bgk_rqs_vjp.h"             /* the backward twin of the training forward (coupling_rqs_bwd_recompute_kernel) */
#define coupling_rqs_dense_h2v2_kernel coupling_rqs_dense_h2v2_train_kernel
#endif
#ifndef BGK_V2_BF16
#define BGK_V2_BF16 0                /* 1 (bgk_fused2_bf16.hip): REDUCED-PRECISION mode "bf16" -- bf16 weights and GEMM inputs, ONE
                                      * v_mfma_f32_32x32x16_bf16 per product instead of the three f16 ones; spline arithmetic unchanged */
#endif
#if BGK_V2_BF16
#define coupling_rqs_dense_h2v2_kernel coupling_rqs_dense_h2v2_bf16_kernel
typedef short s16x8 __attribute__((ext_vector_type(8)));
#endif
#ifndef BGK_V2_AFFTRAIN
#define BGK_V2_AFFTRAIN 0            /* 1 (bgk_fused2_afftrain.hip): the training forward of the AFFINE coupling layer -- coupling_affine_dense_v2_kernel also
                                      * writes the scaled pre-activations of both networks and their outputs (mu, the pre-tanh scale values) */
#endif
#if BGK_V2_AFFTRAIN
#define coupling_affine_dense_v2_kernel coupling_affine_dense_v2_train_kernel
#endif
constexpr int NPROD = BGK_V2_BF16 ? 1 : 3;      /* matrix instructions per (k-step, tile) */

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) char* gptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr4_t;

constexpr int FW = 4;                 /* waves per workgroup */
constexpr int FTHREADS = FW * 64;
constexpr int KB = 8;                 /* spline bins */
constexpr int PPD = 3 * KB + 1;       /* packed rows per transformed dim = 25 */
constexpr int DPC = 128 / PPD;        /* dims per 128-row chunk = 5 */
constexpr int SROW = 33;              /* padded row stride of the y / input tiles */
constexpr int ST = BGK_V2_SAVE ? 33 : 32;   /* row stride of the parameter chunk in LDS (33: the training variant also reads the chunk row-wise) */
constexpr int KS = 8;                 /* k16-steps of a 128-wide hidden layer */
constexpr int GBLK = KS * 8 + 4;      /* 1 KiB blocks per packed 128-row GEMM incl. the 4 bias blocks */
#ifndef BGK_V2_ABL
#define BGK_V2_ABL 0                /* timing ablations (wrong results): 1 no spline, 2 no activation math, 4 no LDS transposition, 8 no staging / output */
#endif
#ifndef BGK_V2_ASMSPLIT
#define BGK_V2_ASMSPLIT 1            /* f16 hi / lo split of the activations: v_cvt_pk_f16_f32 + 2 x v_fma_mix (3 instructions per pair) */
#endif
#ifndef BGK_V2_ACT2
#define BGK_V2_ACT2 1                /* hidden activations: the two values of a pair interleaved (0: one chain after the other) */
#endif
#ifndef BGK_V2_KARG
#define BGK_V2_KARG 2                /* spline constants: 0 SGPR-resident, 1 scalar loads at their uses, 2 per direction (see rqs_fast) */
#endif
#ifndef BGK_V2_RD
#define BGK_V2_RD 4
#endif
#ifndef BGK_V2_PINSEL
#define BGK_V2_PINSEL 1
#endif
#ifndef BGK_V2_BSEARCH
#define BGK_V2_BSEARCH 1             /* bin search: 1 = 3-level binary search with windowed selects, 0 = linear count + select chains */
#endif
#ifndef BGK_V2_OVFL
#define BGK_V2_OVFL 1                 /* 1: the kernel sets MODE.FP16_OVFL -- f16 conversions saturate at +-65504 instead of producing inf (whose lo part
                                       * would be inf - inf), so the activation code carries no clamp instructions */
#endif
#ifndef BGK_V2_TS
#define BGK_V2_TS 0                   /* profiling build (tools/r04_phase_ts.py): lane 0 of every wave stores s_memtime at phase boundaries into bin_idx[tile * 32 * d + k] instead of the bin indices */
#endif
#if BGK_V2_TS
#define V2_TS(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0 && a.bin_idx) a.bin_idx[b0 * d + (k)] = (int)(unsigned)t_; } while (0)
#else
#define V2_TS(k) do { } while (0)
#endif
#ifndef BGK_V2_EORD
#define BGK_V2_EORD 0                 /* event order of a GEMM: 0 = the three products of a tile-step back to back, 1 = part-major over the tiles of a k-step */
#endif
#ifndef BGK_V2_BUF
#define BGK_V2_BUF 1                  /* A stream through buffer loads: one s_mov per 4 KiB group instead of a 64-bit SALU add per tile-step */
#endif
constexpr int RD = BGK_V2_RD;         /* A-fragment ring depth in tile-steps */
constexpr int EH = 34;                /* hook points per spline element */

/* conditioning input as up to BGK_MAX_COND tensors [B, w_i] that stand for their concatenation (CouplingFlow's torch.cat of the
 * conditioning tensors, nn/flow/coupling.py:162-165, without the copy): segment i fills feature rows off[i] .. off[i] + w[i] */
struct CondSegs {
    const float* ptr[BGK_MAX_COND]; int64_t ld[BGK_MAX_COND]; int w[BGK_MAX_COND]; int off[BGK_MAX_COND]; uint32_t magic[BGK_MAX_COND];
    int n;
};

struct SetK { float gnum, low, high, dstep; float kc[7]; };   /* gnum = span * scale, dstep = span * min_bin, kc[k] = low + dstep (k + 1) */

struct SpC {                          /* spline constants */
    float left, right;
    SetK sa, sb;                      /* searched set / other set */
    float beta, kout, min_d;          /* softplus: log(1 + exp(beta s)) / beta = log2(1 + exp2(beta log2e s)) * kout */
};

struct V2Args {
    CondSegs cs;                      /* the conditioning tensor(s): up to BGK_MAX_COND column blocks, each with its own rows */
    int d_c; int periodic;            /* total conditioning width; 1 = featurise as [cos 2 pi c | sin 2 pi c] */
    const float* y; int64_t ldy; float* out; int64_t ldo; int d; uint32_t magic_d;
    int64_t B; float* dlogp; int accumulate; int32_t* bin_idx; int32_t* oob_count;
    const uint4* A0; const uint4* A1; const uint4* A2; int S0; int n_chunks; int last_tiles;
    float c0, c1, c2; const float* cs_dev;
    uint64_t circ_mask;
    SpC sc;
    int lds_per_wave;
    /* tile images in LDS (row-major, one row per sample): the conditioner features [32][nfs] at the head of the parameter-chunk
     * buffer, y / out [32][ys].  stage: 0 = per-lane loads (any strides / several tensors / periodic), 1 = the contiguous [32][d_c]
     * tile copied by the DMA path IS the feature tile (nfs = d_c), 2 = DMA of the raw tile + an elementwise cos / sin pass */
    int nfs, ys, stage, y_dma, out_lin;
#if BGK_V2_SAVE
    float* z0; float* z1;             /* scaled pre-activations [B, 128] */
    float* params; int64_t ldp;       /* spline parameters [B, P] in the reference's column order, or (src_col == NULL) element-major [B][d][3 K + 1] */
    const int32_t* src_col;           /* packed column -> parameter column (-1: padding); NULL: element-major layout */
#endif
};
/* the kernel argument block in the constant address space: the spline constants are (re)read with scalar loads where they are
 * used instead of occupying ~30 SGPRs for the whole persistent loop */
typedef const __attribute__((address_space(4))) V2Args* kargs_t;

struct BFrag { h16x8 hi[KS], lo[KS]; };     /* a 128-wide activation vector as B operands: 64 VGPRs */
struct TFrag { u32x4 hi, lo; };             /* A operand of one (k-step, tile): 8 VGPRs */

__device__ __forceinline__ int drow(int m, int r, int hh) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh; }

/* block BLK (1 KiB) of a packed operand: uniform SGPR base (advanced by SALU in 4 KiB steps) + per-lane VGPR offset + immediate */
template <int BLK>
__device__ __forceinline__ u32x4 ld_block(const uint4* base, unsigned voff) {
    unsigned long long gb = (unsigned long long)base + (unsigned long long)((BLK * 1024) & ~4095);
    asm volatile("" : "+s"(gb));
    return *(gptr4_t)((gptr_t)gb + voff + ((BLK * 1024) & 4095));
}
/* blocks BLK (even) and BLK + 1: the hi / lo parts of one (k-step, tile) share their 4 KiB group and its SGPR base */
template <int BLK>
__device__ __forceinline__ void ld_pair(u32x4& hi, u32x4& lo, const uint4* base, unsigned voff) {
    static_assert((BLK & 1) == 0, "hi / lo pairs start at even blocks");
    unsigned long long gb = (unsigned long long)base + (unsigned long long)((BLK * 1024) & ~4095);
    asm volatile("" : "+s"(gb));
    hi = *(gptr4_t)((gptr_t)gb + voff + ((BLK * 1024) & 4095));
    lo = *(gptr4_t)((gptr_t)gb + voff + ((BLK * 1024) & 4095) + 1024);
}

/* ---- one 128-row (NT live tiles) x 128 GEMM as a stream of single-MFMA events ----------------------------------------
 * tile-step T = s * NT + m (k-step s, tile m), then NT bias tile-steps; event E = 3 T + p: p = 0 lo*hi, 1 hi*lo, 2 hi*hi
 * (small terms first), bias events one MFMA each.  The ring slot of tile-step T is refilled with T + RD right after its
 * last MFMA has been issued. */
template <int NT, int MS = 4, int RDX = RD>       /* MS: tiles per k-step in the packed operand (spline layers: 4; affine output layer: NT); RDX: ring depth */
struct Live {
    static constexpr int NTS = KS * NT;           /* product tile-steps */
    static constexpr int NEV = NPROD * NTS + NT;  /* events */
    f32x16 (&out)[4];
    const BFrag& b;
    const uint4* W;
    unsigned voff;
    TFrag (&ring)[RDX];
#if BGK_V2_BUF
    __amdgpu_buffer_rsrc_t rs;
#endif

    template <int BLK>
    __device__ __forceinline__ u32x4 blk() {
#if BGK_V2_BUF
        return __builtin_amdgcn_raw_buffer_load_b128(rs, voff + ((BLK * 1024) & 4095), (BLK * 1024) & ~4095, 0);
#else
        return ld_block<BLK>(W, voff);
#endif
    }
    /* the hi / the lo block of tile-step T alone (event order 1 refills the two halves of a ring slot at different times) */
    template <int T>
    __device__ __forceinline__ void load_hi() {
        if constexpr (T < NTS) ring[T % RDX].hi = blk<((T / NT) * MS + T % NT) * 2>();
        else if constexpr (T < NTS + NT) ring[T % RDX].hi = blk<KS * MS * 2 + (T - NTS)>();
    }
    template <int T>
    __device__ __forceinline__ void load_lo() {
        if constexpr (T < NTS) ring[T % RDX].lo = blk<((T / NT) * MS + T % NT) * 2 + 1>();
    }
    template <int T>
    __device__ __forceinline__ void load() {
#ifdef BGK_V2_ABL_NOLOAD     /* timing experiment: only the prologue of each GEMM loads A (wrong results) */
        if constexpr (T >= RDX) return;
#endif
        if constexpr (T < NTS) {
            constexpr int s = T / NT, m = T % NT;
#if BGK_V2_BF16
            ring[T % RDX].hi = blk<(s * MS + m) * 2>();         /* the bf16 values sit in the "hi" blocks of the same layout */
#elif BGK_V2_BUF
            ring[T % RDX].hi = blk<(s * MS + m) * 2>();
            ring[T % RDX].lo = blk<(s * MS + m) * 2 + 1>();
#else
            ld_pair<(s * MS + m) * 2>(ring[T % RDX].hi, ring[T % RDX].lo, W, voff);
#endif
        } else if constexpr (T < NTS + NT) {
            ring[T % RDX].hi = blk<KS * MS * 2 + (T - NTS)>();
        }
    }
    template <int T0, int T1>
    __device__ __forceinline__ void loads() {
        if constexpr (T0 < T1) { load<T0>(); loads<T0 + 1, T1>(); }
    }
    __device__ __forceinline__ void start() {
        __builtin_amdgcn_sched_barrier(0);
#if BGK_V2_BUF
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);   /* raw buffer, dword3 = gfx9 default format */
#endif
        loads<0, RDX>();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int E>
    __device__ __forceinline__ void event() {
        __builtin_amdgcn_sched_barrier(0);      /* pin the MFMA and the ring refill: the scheduler would sink the loads to their uses */
#if BGK_V2_BF16
        if constexpr (E < NTS) {
            constexpr int T = E, s = T / NT, m = T % NT;
            const s16x8 a = __builtin_bit_cast(s16x8, ring[T % RDX].hi), bb = __builtin_bit_cast(s16x8, b.hi[s]);
            if constexpr (s == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                out[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, z, 0, 0, 0);
            } else {
                out[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, out[m], 0, 0, 0);
            }
            load<T + RDX>();
        } else if constexpr (E < NEV) {
            constexpr int m = E - NTS, T = NTS + m;
            const s16x8 one2 = {(short)0x3f80, (short)0x3f80, 0, 0, 0, 0, 0, 0};
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, ring[T % RDX].hi), one2, out[m], 0, 0, 0);
        }
#elif BGK_V2_EORD
        /* event order 1: within a group of G consecutive tile-steps the products run part-major -- lo*hi of the G tiles, hi*lo of
         * the G tiles, hi*hi of the G tiles -- so that consecutive MFMAs never write the same accumulator: a dependent MFMA that is
         * not issued back to back with its predecessor (VALU work is threaded between the events) waits for the predecessor's
         * write-back instead of using the matrix pipe's accumulator forwarding (MI355X_MICROARCH.md: +43 cycles).  The lo half of a
         * ring slot is free after the first part, the hi half after the third: they are refilled separately. */
        if constexpr (E < 3 * NTS) {
            constexpr int G = (RDX < NT ? RDX : NT);
            static_assert(NTS % G == 0, "groups tile the product tile-steps");
            constexpr int grp = E / (3 * G), rem = E % (3 * G), p = rem / G, T = grp * G + rem % G, s = T / NT, m = T % NT;
            const TFrag& f = ring[T % RDX];
            const h16x8 a = __builtin_bit_cast(h16x8, p == 0 ? f.lo : f.hi);
            const h16x8 bb = p == 1 ? b.lo[s] : b.hi[s];
            if constexpr (s == 0 && p == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, z, 0, 0, 0);
            } else {
                out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, out[m], 0, 0, 0);
            }
            if constexpr (p == 0) load_lo<T + RDX>();
            if constexpr (p == 2) load_hi<T + RDX>();
        } else if constexpr (E < NEV) {
            constexpr int m = E - 3 * NTS, T = NTS + m;
            const h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ring[T % RDX].hi), one2, out[m], 0, 0, 0);
        }
#else
        if constexpr (E < 3 * NTS) {
            constexpr int T = E / 3, p = E % 3, s = T / NT, m = T % NT;
            const TFrag& f = ring[T % RDX];
            const h16x8 a = __builtin_bit_cast(h16x8, p == 0 ? f.lo : f.hi);
            const h16x8 bb = p == 1 ? b.lo[s] : b.hi[s];
            if constexpr (s == 0 && p == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, z, 0, 0, 0);
            } else {
                out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, out[m], 0, 0, 0);
            }
            if constexpr (p == 2) load<T + RDX>();
        } else if constexpr (E < NEV) {
            constexpr int m = E - 3 * NTS, T = NTS + m;
            const h16x8 one2 = {(_Float16)1.0f, (_Float16)1.0f, 0, 0, 0, 0, 0, 0};
            out[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ring[T % RDX].hi), one2, out[m], 0, 0, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int E0, int E1>
    __device__ __forceinline__ void events() {
        if constexpr (E0 < E1) {
            event<E0>();
            events<E0 + 1, E1>();
        }
    }
    /* hook HK of NH: its share of the NEV events, fenced so that neither the MFMAs nor the surrounding VALU work move */
    template <int HK, int NH>
    __device__ __forceinline__ void hook() {
        constexpr int e0 = (HK * NEV) / NH, e1 = ((HK + 1) * NEV) / NH;
        if constexpr (e0 < e1) events<e0, e1>();
    }
};
struct NoLive {
    template <int HK, int NH> __device__ __forceinline__ void hook() {}
};

/* hook adapters: map the hook index of a code region (activation of one tile: 8 hooks; spline slot: EH hooks) to the
 * hook space of the overlapped GEMM */
template <class G, int BASE, int NH>
struct Hooks {
    G& g;
    template <int I> __device__ __forceinline__ void at() { g.template hook<BASE + I, NH>(); }
};

/* ---- hidden activation (hardware exp2 / rcp) of x = t * c and split into f16 hi + lo ---- */
template <int ACT>
__device__ __forceinline__ float act_hw(float x) {
#if (BGK_V2_ABL & 2)
    return x;
#endif
    if constexpr (ACT == 1) {           /* SiLU */
        const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896341f);
        return x * __builtin_amdgcn_rcpf(1.0f + e);
    } else if constexpr (ACT == 2) {    /* ReLU */
        return x > 0.0f ? x : 0.0f;
    } else {                            /* tanh = 1 - 2 / (1 + exp(2x)) */
        const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681f);
        return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
    }
}

/* tile T of a layer output (accumulator layout, 16 values per lane) -> B operands of k-steps 2T and 2T + 1; three hooks per pair */
template <int ACT, int T, int P, class H>
__device__ __forceinline__ void act_split_pair(H& hk, const f32x16& t, float c, BFrag& bf) {
    constexpr int r = 2 * P;
#if BGK_V2_ACT2 && !(BGK_V2_ABL & 2)
    /* both values of the pair advance together: exp2 of both | MFMA | reciprocals + products of both | MFMA.  The two chains are
     * independent, so no consumer sits directly behind its transcendental (no hazard nop, no wait for the transcendental's latency) */
    float a0, a1;
    if constexpr (ACT == 2) {
        a0 = t[r] * c; a1 = t[r + 1] * c;
        hk.template at<3 * P>();
        a0 = a0 > 0.0f ? a0 : 0.0f; a1 = a1 > 0.0f ? a1 : 0.0f;
        hk.template at<3 * P + 1>();
    } else {
        /* t = UNSCALED accumulator value, x = t c.  SiLU: x / (1 + e^-x) = t / ((1 + e) / c): the unscale factor rides on the exponent's
         * constant and on the fma that forms the denominator (5 instructions per value instead of 7); tanh needs x only inside the exponent */
        constexpr float kk = ACT == 1 ? -1.44269504088896341f : 2.88539008177792681f;
        const float ck = c * kk, rc = 1.0f / c;
        const float e0 = __builtin_amdgcn_exp2f(t[r] * ck), e1 = __builtin_amdgcn_exp2f(t[r + 1] * ck);
        hk.template at<3 * P>();
        if constexpr (ACT == 1) {
            const float q0 = __builtin_amdgcn_rcpf(__builtin_fmaf(e0, rc, rc)), q1 = __builtin_amdgcn_rcpf(__builtin_fmaf(e1, rc, rc));
            a0 = t[r] * q0; a1 = t[r + 1] * q1;
        } else {
            const float q0 = __builtin_amdgcn_rcpf(1.0f + e0), q1 = __builtin_amdgcn_rcpf(1.0f + e1);
            a0 = 1.0f - 2.0f * q0; a1 = 1.0f - 2.0f * q1;
        }
        hk.template at<3 * P + 1>();
    }
#else
    float a0 = act_hw<ACT>(t[r] * c);
    hk.template at<3 * P>();
    float a1 = act_hw<ACT>(t[r + 1] * c);
    hk.template at<3 * P + 1>();
#endif
#if !BGK_V2_OVFL
    if constexpr (ACT != 3) {       /* SiLU / ReLU outputs are bounded below; keep the f16 conversion finite above */
        a0 = __builtin_fminf(a0, 65000.0f); a1 = __builtin_fminf(a1, 65000.0f);
    }
#endif
    constexpr int s = 2 * T + (r >> 3), e = r & 7;
#if BGK_V2_BF16
    typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const bf2v pb = __builtin_convertvector((f2v){a0, a1}, bf2v);           /* v_cvt_pk_bf16_f32, round to nearest even */
    const h2v ph = __builtin_bit_cast(h2v, pb);
    bf.hi[s][e] = ph[0]; bf.hi[s][e + 1] = ph[1];
#elif BGK_V2_ASMSPLIT
    /* hi pair = v_cvt_pk_f16_f32 (RNE); lo = f16(a - hi) by the mixed-precision FMA reading hi as an f16 operand and writing one half
     * of the destination: a - hi is exact in f32, so the single rounding equals (_Float16)(a - (float)hi).  3 instructions per pair. */
    unsigned uh, ul;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(uh) : "v"(a0), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ul) : "v"(uh), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ul) : "v"(uh), "v"(a1));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const h2v ph = __builtin_bit_cast(h2v, uh), pl = __builtin_bit_cast(h2v, ul);
    bf.hi[s][e] = ph[0]; bf.hi[s][e + 1] = ph[1];
    bf.lo[s][e] = pl[0]; bf.lo[s][e + 1] = pl[1];
#else
    const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
    bf.hi[s][e] = h0; bf.hi[s][e + 1] = h1;
    bf.lo[s][e] = (_Float16)(a0 - (float)h0); bf.lo[s][e + 1] = (_Float16)(a1 - (float)h1);
#endif
    hk.template at<3 * P + 2>();
}
template <int ACT, int T, class H>
__device__ __forceinline__ void act_split_tile(H hk, const f32x16& t, float c, BFrag& bf) {
    act_split_pair<ACT, T, 0>(hk, t, c, bf); act_split_pair<ACT, T, 1>(hk, t, c, bf);
    act_split_pair<ACT, T, 2>(hk, t, c, bf); act_split_pair<ACT, T, 3>(hk, t, c, bf);
    act_split_pair<ACT, T, 4>(hk, t, c, bf); act_split_pair<ACT, T, 5>(hk, t, c, bf);
    act_split_pair<ACT, T, 6>(hk, t, c, bf); act_split_pair<ACT, T, 7>(hk, t, c, bf);
}

/* ---- spline element, hardware-transcendental / regrouped form ------------------------------------------------------
 * pa / pb / ps: the element's 8 unnormalised searched-set / other-set / slope rows in the LDS chunk (row stride ST), UNSCALED
 * accumulator values (true parameter = value * c2; kL = c2 log2 e, kz = c2 beta log2 e fold the factor).  ps[8 ST] is the
 * non-circular extra slope row.  Follows nflows' rational_quadratic_spline like bgk_rqs_element. */
struct SpK { float kL, kz, c2; };

__device__ __forceinline__ float fdiv_hw(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ float rcp_nr(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}

template <bool KARG> struct ScSrc;
template <> struct ScSrc<true> {
    kargs_t ka;
    __device__ __forceinline__ ScSrc(const V2Args&) { ka = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(ka)); }
    __device__ __forceinline__ const __attribute__((address_space(4))) SpC& get() const { return ka->sc; }
};
template <> struct ScSrc<false> {
    const V2Args& a;
    __device__ __forceinline__ ScSrc(const V2Args& a_) : a(a_) {}
    __device__ __forceinline__ const SpC& get() const { return a.sc; }
};

template <int INV, class H>
__device__ __forceinline__ float rqs_fast(H hk, float x, const float* pa, const float* pb, const float* ps, bool circ,
                                          const V2Args& a, const SpK& k, float* lad, int* bin, int* oob) {
    /* the spline constants: (re)read from the kernel argument block by scalar loads inside the element (inverse instance: fewer
     * live SGPRs win there), or kept in SGPRs for the whole tile (forward instance: ~470 fewer s_load + s_waitcnt per tile; A/B
     * on one box: forward -2 % with registers, inverse +2 %) */
    const ScSrc<(BGK_V2_KARG == 2 ? INV != 0 : BGK_V2_KARG != 0)> scs(a);
#define SC scs.get()
    float va[KB], vb[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) va[i] = pa[i * ST];
#pragma unroll
    for (int i = 0; i < KB; ++i) vb[i] = pb[i * ST];
    *oob = (x < SC.left) | (x > SC.right);
    x = __builtin_amdgcn_fmed3f(x, SC.left, SC.right);
    /* ---- searched set: softmax numerators, running sums, the 7 interior knots, comparison masks ---- */
    float mA = __builtin_fmaxf(__builtin_fmaxf(va[0], va[1]), va[2]);
    mA = __builtin_fmaxf(__builtin_fmaxf(mA, va[3]), va[4]);
    mA = __builtin_fmaxf(__builtin_fmaxf(mA, va[5]), va[6]);
    mA = __builtin_fmaxf(mA, va[7]);
    const float nmA = -(mA * k.kL);
    hk.template at<0>();
    float E[KB];
    E[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[0], k.kL, nmA));
    E[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[1], k.kL, nmA));
    E[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[2], k.kL, nmA));
    hk.template at<1>();
    E[3] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[3], k.kL, nmA));
    E[4] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[4], k.kL, nmA));
    E[5] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[5], k.kL, nmA));
    hk.template at<2>();
    E[6] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[6], k.kL, nmA));
    E[7] = __builtin_amdgcn_exp2f(__builtin_fmaf(va[7], k.kL, nmA));
    E[1] += E[0]; E[2] += E[1];
    hk.template at<3>();
    E[3] += E[2]; E[4] += E[3]; E[5] += E[4]; E[6] += E[5]; E[7] += E[6];
    hk.template at<4>();
    const float gA = SC.sa.gnum * rcp_nr(E[7]);
    float kn[7];
    kn[0] = __builtin_fmaf(E[0], gA, SC.sa.kc[0]);
    kn[1] = __builtin_fmaf(E[1], gA, SC.sa.kc[1]);
    hk.template at<5>();
    kn[2] = __builtin_fmaf(E[2], gA, SC.sa.kc[2]);
    kn[3] = __builtin_fmaf(E[3], gA, SC.sa.kc[3]);
    kn[4] = __builtin_fmaf(E[4], gA, SC.sa.kc[4]);
    kn[5] = __builtin_fmaf(E[5], gA, SC.sa.kc[5]);
    kn[6] = __builtin_fmaf(E[6], gA, SC.sa.kc[6]);
    hk.template at<6>();
#if BGK_V2_BSEARCH
    /* Bin search as a 3-level binary search on the (monotone) knots: idx = #{k : x >= kn[k]} exactly as the linear count (the knots are
     * non-decreasing: fma of non-decreasing prefix sums with a positive gain onto increasing offsets), with the window of candidate
     * knots / other-set prefix sums halved by selects at every level: 3 compares + 20 selects instead of 7 + 36. */
    /* (knots and end points as opaque register values: with a load or an fma on one arm only, the compiler turns the selects into
     * divergent branches, which also cuts the pinned MFMA / VALU interleaving) */
    float low_a = SC.sa.low, high_a = SC.sa.high;
    asm volatile("" : "+s"(low_a), "+s"(high_a));
#pragma unroll
    for (int i = 0; i < 7; ++i) asm volatile("" : "+v"(kn[i]));
    const bool g3 = x >= kn[3];
    hk.template at<7>();
    float w0 = g3 ? kn[3] : low_a, w1 = g3 ? kn[4] : kn[0], w2 = g3 ? kn[5] : kn[1], w3 = g3 ? kn[6] : kn[2];
    hk.template at<8>();
    float w4 = g3 ? high_a : kn[3];
    const bool gm = x >= w2;
    const float v0 = gm ? w2 : w0, v1 = gm ? w3 : w1, v2 = gm ? w4 : w2;
    hk.template at<9>();
    const bool gl = x >= v1;
    const float lo = gl ? v1 : v0, hi = gl ? v2 : v1;
    float idxf = g3 ? 4.0f : 0.0f;
    hk.template at<10>();
    idxf += gm ? 2.0f : 0.0f;
    idxf += gl ? 1.0f : 0.0f;
    const int idx = (int)idxf;
    *bin = idx;
    hk.template at<11>();
    /* the two slopes of the bin (dynamic LDS rows; circular dims wrap the last knot's slope to row 0) */
    const int j1 = circ ? ((idx + 1) & 7) : (idx + 1);
    const float s_lo = ps[idx * ST], s_hi = ps[j1 * ST];
    const float A_i = hi - lo;
    /* ---- other set: only knot[idx], knot[idx + 1] ---- */
    float mB = __builtin_fmaxf(__builtin_fmaxf(vb[0], vb[1]), vb[2]);
    hk.template at<12>();
    mB = __builtin_fmaxf(__builtin_fmaxf(mB, vb[3]), vb[4]);
    mB = __builtin_fmaxf(__builtin_fmaxf(mB, vb[5]), vb[6]);
    mB = __builtin_fmaxf(mB, vb[7]);
    const float nmB = -(mB * k.kL);
    float F[KB];
    F[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[0], k.kL, nmB));
    hk.template at<13>();
    F[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[1], k.kL, nmB));
    F[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[2], k.kL, nmB));
    F[3] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[3], k.kL, nmB));
    hk.template at<14>();
    F[4] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[4], k.kL, nmB));
    F[5] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[5], k.kL, nmB));
    F[6] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[6], k.kL, nmB));
    hk.template at<15>();
    F[7] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[7], k.kL, nmB));
    F[1] += F[0]; F[2] += F[1]; F[3] += F[2];
    hk.template at<16>();
    F[4] += F[3]; F[5] += F[4]; F[6] += F[5]; F[7] += F[6];
    const float rB = __builtin_amdgcn_rcpf(F[7]);
    hk.template at<17>();
    const float gB = SC.sb.gnum * __builtin_fmaf(__builtin_fmaf(-F[7], rB, 1.0f), rB, rB);
    /* prefix sums (0, F0 .. F7) of the other set, windowed by the same three decisions */
    const float f0 = g3 ? F[3] : 0.0f, f1 = g3 ? F[4] : F[0], f2 = g3 ? F[5] : F[1];
    hk.template at<18>();
    const float f3 = g3 ? F[6] : F[2], f4 = g3 ? F[7] : F[3];
    const float u0 = gm ? f2 : f0, u1 = gm ? f3 : f1, u2 = gm ? f4 : f2;
    hk.template at<19>();
    const float Flo = gl ? u1 : u0;
    float Fhi = gl ? u2 : u1;
    hk.template at<20>();
    const float cb = __builtin_fmaf(idxf, SC.sb.dstep, SC.sb.low);
    hk.template at<21>();
    const float b_i = __builtin_fmaf(Flo, gB, cb);
    float b_ip1 = __builtin_fmaf(Fhi, gB, cb + SC.sb.dstep);
    float top_b = SC.sb.high;
    asm volatile("" : "+s"(top_b));                         /* (a load on one arm would turn the select into a branch) */
    b_ip1 = (g3 && gm && gl) ? top_b : b_ip1;
    const float B_i = b_ip1 - b_i;
#else
    const bool g0 = x >= kn[0], g1 = x >= kn[1], g2 = x >= kn[2], g3 = x >= kn[3], g4 = x >= kn[4], g5 = x >= kn[5], g6 = x >= kn[6];
    hk.template at<7>();
    int idx = (g0 ? 1 : 0) + (g1 ? 1 : 0) + (g2 ? 1 : 0) + (g3 ? 1 : 0) + (g4 ? 1 : 0) + (g5 ? 1 : 0) + (g6 ? 1 : 0);
    *bin = idx;
    hk.template at<8>();
    /* the two slopes of the bin (dynamic LDS rows; circular dims wrap the last knot's slope to row 0) */
    const int j1 = circ ? ((idx + 1) & 7) : (idx + 1);
    const float s_lo = ps[idx * ST], s_hi = ps[j1 * ST];
    float lo = SC.sa.low;
    lo = g0 ? kn[0] : lo; lo = g1 ? kn[1] : lo; lo = g2 ? kn[2] : lo;
    hk.template at<9>();
    lo = g3 ? kn[3] : lo; lo = g4 ? kn[4] : lo; lo = g5 ? kn[5] : lo; lo = g6 ? kn[6] : lo;
    hk.template at<10>();
    float hi = SC.sa.high;
    hi = g6 ? hi : kn[6]; hi = g5 ? hi : kn[5]; hi = g4 ? hi : kn[4]; hi = g3 ? hi : kn[3];
    hk.template at<11>();
    hi = g2 ? hi : kn[2]; hi = g1 ? hi : kn[1]; hi = g0 ? hi : kn[0];
    const float A_i = hi - lo;
    /* ---- other set: only knot[idx], knot[idx + 1] ---- */
    float mB = __builtin_fmaxf(__builtin_fmaxf(vb[0], vb[1]), vb[2]);
    hk.template at<12>();
    mB = __builtin_fmaxf(__builtin_fmaxf(mB, vb[3]), vb[4]);
    mB = __builtin_fmaxf(__builtin_fmaxf(mB, vb[5]), vb[6]);
    mB = __builtin_fmaxf(mB, vb[7]);
    const float nmB = -(mB * k.kL);
    float F[KB];
    F[0] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[0], k.kL, nmB));
    hk.template at<13>();
    F[1] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[1], k.kL, nmB));
    F[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[2], k.kL, nmB));
    F[3] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[3], k.kL, nmB));
    hk.template at<14>();
    F[4] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[4], k.kL, nmB));
    F[5] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[5], k.kL, nmB));
    F[6] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[6], k.kL, nmB));
    hk.template at<15>();
    F[7] = __builtin_amdgcn_exp2f(__builtin_fmaf(vb[7], k.kL, nmB));
    F[1] += F[0]; F[2] += F[1]; F[3] += F[2];
    hk.template at<16>();
    F[4] += F[3]; F[5] += F[4]; F[6] += F[5]; F[7] += F[6];
    const float rB = __builtin_amdgcn_rcpf(F[7]);
    hk.template at<17>();
    const float gB = SC.sb.gnum * __builtin_fmaf(__builtin_fmaf(-F[7], rB, 1.0f), rB, rB);
    float Flo = 0.0f;
    Flo = g0 ? F[0] : Flo; Flo = g1 ? F[1] : Flo; Flo = g2 ? F[2] : Flo;
    hk.template at<18>();
    Flo = g3 ? F[3] : Flo; Flo = g4 ? F[4] : Flo; Flo = g5 ? F[5] : Flo; Flo = g6 ? F[6] : Flo;
    hk.template at<19>();
    float Fhi = F[7];
    Fhi = g6 ? Fhi : F[6]; Fhi = g5 ? Fhi : F[5]; Fhi = g4 ? Fhi : F[4]; Fhi = g3 ? Fhi : F[3];
    hk.template at<20>();
    Fhi = g2 ? Fhi : F[2]; Fhi = g1 ? Fhi : F[1]; Fhi = g0 ? Fhi : F[0];
    const float cb = __builtin_fmaf((float)idx, SC.sb.dstep, SC.sb.low);
    hk.template at<21>();
    const float b_i = __builtin_fmaf(Flo, gB, cb);
    float b_ip1 = __builtin_fmaf(Fhi, gB, cb + SC.sb.dstep);
    b_ip1 = g6 ? SC.sb.high : b_ip1;
    const float B_i = b_ip1 - b_i;
#endif
    /* ---- derivatives: min_d + softplus(beta s) / beta ---- */
    const float z0 = s_lo * k.kz;                           /* log2(e) beta s_true */
    hk.template at<22>();
    const float ez0 = __builtin_amdgcn_exp2f(z0);
    float lg0 = __builtin_amdgcn_logf(1.0f + ez0);          /* v_log_f32 = log2 */
    asm volatile("" : "+v"(lg0));                           /* keep the select below a select (no divergent branch around the log) */
    const float sm0 = ez0 * __builtin_fmaf(ez0, -0.5f, 1.0f) * 1.44269504088896341f;   /* log1p for tiny arguments */
    hk.template at<23>();
    float l0 = (ez0 < 2.44140625e-4f ? sm0 : lg0) * SC.kout;
#if BGK_V2_PINSEL
    asm volatile("" : "+v"(l0));                            /* keeps the identity select below a select where SC.kout is a scalar load */
#endif
    l0 = z0 > 28.8539008177792681f ? s_lo * k.c2 : l0;      /* beta s > 20: identity (torch softplus threshold) */
    const float d_i = SC.min_d + l0;
    const float z1 = s_hi * k.kz;
    hk.template at<24>();
    const float ez1 = __builtin_amdgcn_exp2f(z1);
    float lg1 = __builtin_amdgcn_logf(1.0f + ez1);
    asm volatile("" : "+v"(lg1));
    const float sm1 = ez1 * __builtin_fmaf(ez1, -0.5f, 1.0f) * 1.44269504088896341f;
    hk.template at<25>();
    float l1 = (ez1 < 2.44140625e-4f ? sm1 : lg1) * SC.kout;
#if BGK_V2_PINSEL
    asm volatile("" : "+v"(l1));                            /* keeps the identity select below a select where SC.kout is a scalar load */
#endif
    l1 = z1 > 28.8539008177792681f ? s_hi * k.c2 : l1;
    const float d_ip1 = SC.min_d + l1;
    float cw_i, W_i, ch_i, H_i;
    if (INV) { cw_i = lo; W_i = A_i; ch_i = b_i; H_i = B_i; }
    else { ch_i = lo; H_i = A_i; cw_i = b_i; W_i = B_i; }
    hk.template at<26>();
    const float delta = fdiv_hw(H_i, W_i);
    const float S = d_i + d_ip1 - 2.0f * delta;
    hk.template at<27>();
    float outv, l;
    if (!INV) {
        const float dx = x - ch_i;
        const float dxS = dx * S;
        const float qa = dxS + H_i * (delta - d_i);
        const float qb = H_i * d_i - dxS;
        hk.template at<28>();
        const float qc = -delta * dx;
        const float disc = qb * qb - 4.0f * qa * qc;
        const float sq = __builtin_amdgcn_sqrtf(disc);
        hk.template at<29>();
        const float root = fdiv_hw(2.0f * qc, -qb - sq);
        outv = __builtin_fmaf(root, W_i, cw_i);
        hk.template at<30>();
        const float omr = 1.0f - root;
        const float t1mt = root * omr;
        const float den = delta + S * t1mt;
        const float lden = __builtin_amdgcn_logf(den);
        hk.template at<31>();
        const float num = (delta * delta) * (d_ip1 * (root * root) + 2.0f * delta * t1mt + d_i * (omr * omr));
        hk.template at<32>();
        l = (2.0f * lden - __builtin_amdgcn_logf(num)) * 0.693147180559945309f;
    } else {
        const float theta = fdiv_hw(x - cw_i, W_i);
        const float omt = 1.0f - theta;
        hk.template at<28>();
        const float t1mt = theta * omt;
        const float numer = H_i * (delta * (theta * theta) + d_i * t1mt);
        const float den = delta + S * t1mt;
        hk.template at<29>();
        outv = ch_i + fdiv_hw(numer, den);
        const float lden = __builtin_amdgcn_logf(den);
        hk.template at<30>();
        const float inner = d_ip1 * (theta * theta) + 2.0f * delta * t1mt;
        hk.template at<31>();
        const float num = (delta * delta) * (inner + d_i * (omt * omt));
        hk.template at<32>();
        l = (__builtin_amdgcn_logf(num) - 2.0f * lden) * 0.693147180559945309f;
    }
    hk.template at<33>();
    *lad = l;
    return outv;
}

#undef SC

/* slot IT of a chunk: element q = 2 IT + hh of sample j; invalid slots evaluate dim 0 of the chunk and are discarded */
template <int INV, int IT, int NHK, class G>
__device__ __forceinline__ void spline_slot(G& g, const V2Args& a, const SpK& k, const float* s_p, float* s_y, int c, int nd,
                                            int hh, int j, int rows, float& run, int& oob_local, int (&bins)[3]) {
    const int q = 2 * IT + hh;
    const bool valid = q < nd;
    const int qq = valid ? q : 0;
    const int dim = c * DPC + qq;
    const float* pw = s_p + (qq * PPD) * ST + j;
    const float* ph = pw + KB * ST;
    const float* ps = ph + KB * ST;
    const bool circ = (a.circ_mask >> dim) & 1ull;
    int bin, oob;
    float lad;
    const float x = s_y[j * a.ys + dim];
    Hooks<G, IT * EH, NHK> hk{g};
    const float o = rqs_fast<INV>(hk, x, INV ? pw : ph, INV ? ph : pw, ps, circ, a, k, &lad, &bin, &oob);
    s_y[valid ? j * a.ys + dim : 32 * a.ys + j] = o;         /* (32 spare floats behind the tile take the results of invalid slots) */
    oob_local += (valid && j < rows) ? oob : 0;
    bins[IT] = bin;
    lad = valid ? lad : 0.0f;
    /* dim 2 IT lives in the lower half-wave, dim 2 IT + 1 in the upper one: both halves add them in ascending dim order */
    /* v_permlane32_swap (lanes 32..63 of the first register <-> lanes 0..31 of the second): afterwards l0 holds the lower
     * half-wave's value in both halves and l1 the upper one's.  Inline asm: hipcc 7.2's __builtin_amdgcn_permlane32_swap
     * returns the first result for both elements (checked in the ISA); s_nop 1 = the VALU-write -> permlane-read wait states. */
    float l0 = lad, l1 = lad;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(l0), "+v"(l1));
    run += l0;
    run += l1;
}

/* accumulator layout [row = feature][lane = (half, sample)] -> LDS [row][sample] (conflict-free both ways) */
__device__ __forceinline__ void chunk_to_lds(const f32x16 (&h)[4], float* s_p, int hh, int j) {
#if (BGK_V2_ABL & 4)
    s_p[threadIdx.x & 63] = h[0][0] + h[1][1] + h[2][2] + h[3][3];
    return;
#endif
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_p[drow(m, r, hh) * ST + j] = h[m][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* chunk c: registers -> LDS (transposition), then the spline of this chunk threaded through the next chunk's GEMM */
#if BGK_V2_SAVE
/* parameters of chunk c (in s_p; UNSCALED accumulator values: true parameter = value * c2) -> params[b][col]: lane = packed row (two
 * passes of 64), one sample row per store instruction: contiguous 32-byte runs (the 8 bins of a (dim, component)).  Called right
 * after the chunk reached LDS and AFTER the next GEMM's first operand loads were requested: vmcnt is one in-order counter for
 * loads and stores on gfx9, so a load requested behind these 64 stores would wait for every one of them to be acknowledged. */
__device__ __forceinline__ void save_chunk_params(const V2Args& a, const float* s_p, int c, int lane, int64_t b0, int rows) {
#ifdef BGK_V2_ABL_NOPSAVE      /* timing experiment: the parameters are not written (wrong gradients) */
    return;
#endif
    if (a.params == nullptr) return;      /* the backward recomputes them from z1 (coupling_rqs_bwd_recompute_kernel below) */
    if (a.src_col == nullptr) {
        /* element-major layout [B][d][3 K + 1] (round 5): the chunk's rows ARE that order, so a sample's share of the chunk is one
         * contiguous run of DPC * PPD = 125 floats -- 32 lanes x 16 bytes, two sample rows per store instruction, 16 instructions
         * per chunk instead of 64 (the last lane of a row carries the first 3 floats of the next chunk's run, which that chunk
         * overwrites; the row pitch leaves room behind the last one).  0.29 -> 0.25 ms per layer at 2^18 samples. */
        typedef float f4s __attribute__((ext_vector_type(4), aligned(4)));
        const int l = lane & 31, half = lane >> 5;
        const int nd = (a.d - c * DPC) < DPC ? (a.d - c * DPC) : DPC;
        const bool mine = 4 * l < nd * PPD;
        for (int jj = 0; jj < rows; jj += 2) {
            const int r = jj + half;
            if (r < rows && mine) {
                float* prow = a.params + (b0 + r) * a.ldp + (DPC * PPD) * c;
                const f4s v = {s_p[(4 * l) * ST + r] * a.c2, s_p[(4 * l + 1) * ST + r] * a.c2, s_p[(4 * l + 2) * ST + r] * a.c2, s_p[(4 * l + 3) * ST + r] * a.c2};
                *reinterpret_cast<f4s*>(prow + 4 * l) = v;
            }
        }
        return;
    }
    const int col_lo = a.src_col[c * 128 + lane], col_hi = a.src_col[c * 128 + 64 + lane];
    for (int jj = 0; jj < rows; ++jj) {
        float* prow = a.params + (b0 + jj) * a.ldp;
        if (col_lo >= 0) prow[col_lo] = s_p[lane * ST + jj] * a.c2;
        if (col_hi >= 0) prow[col_hi] = s_p[(64 + lane) * ST + jj] * a.c2;
    }
}
#endif

template <int INV, int NT, bool SAVEP>
__device__ __forceinline__ void chunk_piped(const V2Args& a, const SpK& k, float* s_p, float* s_y, int c, int hh, int j, int rows,
                                            float& run, int& oob_local, int (&bins)[3], f32x16 (&h)[4], const BFrag& bf,
                                            TFrag (&ring)[RD], unsigned voff, int64_t b0) {
    Live<NT> g{h, bf, a.A2 + (size_t)(c + 1) * GBLK * 64, voff, ring};
    g.start();                       /* the next GEMM's first A fragments travel while this chunk goes through LDS */
    chunk_to_lds(h, s_p, hh, j);
#if BGK_V2_SAVE
    if constexpr (SAVEP) save_chunk_params(a, s_p, c, (int)threadIdx.x & 63, b0, rows);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if (BGK_V2_ABL & 1)
    g.template events<0, Live<NT>::NEV>();
    run += s_p[(threadIdx.x & 127) * ST + j];
#else
    spline_slot<INV, 0, 3 * EH>(g, a, k, s_p, s_y, c, DPC, hh, j, rows, run, oob_local, bins);
    spline_slot<INV, 1, 3 * EH>(g, a, k, s_p, s_y, c, DPC, hh, j, rows, run, oob_local, bins);
    spline_slot<INV, 2, 3 * EH>(g, a, k, s_p, s_y, c, DPC, hh, j, rows, run, oob_local, bins);
#endif
}

/* linear DMA copy of a wave's contiguous, 16-byte aligned tile of n floats (a multiple of 4) into LDS; only the first `valid`
 * floats exist in memory (last tile of a batch): the others are left alone (lanes masked) -- nothing downstream depends on them */
typedef const __attribute__((address_space(1))) void* gvp_t;
typedef __attribute__((address_space(3))) void* lvp_t;
__device__ __forceinline__ void dma_tile32(float* dst, const float* __restrict__ src, int n, int valid, int lane) {
    for (int c = 0; c < n; c += 256) {                /* 1 KiB per instruction */
        const int e = c + lane * 4;
        if (e + 3 < valid) __builtin_amdgcn_global_load_lds((gvp_t)(src + e), (lvp_t)(dst + c), 16, 0, 0);
    }
    if (valid & 3) {                                  /* partial tile whose length is no multiple of 4: its last 1..3 floats, one dword each */
        const int edge = valid & ~3;
        if (lane < (valid & 3)) __builtin_amdgcn_global_load_lds((gvp_t)(src + edge + lane), (lvp_t)(dst + edge), 4, 0, 0);
    }
}

/* sin / cos of 2 pi x for the periodic featuriser of these tolerance-class kernels: the hardware forms take their argument in
 * revolutions (max abs error 1.24e-7 on [-2, 2], tools/ubench/hw_sincos.hip: the resolution of the split-f16 operands they feed);
 * 3 instructions instead of the ~30 of the reproducible polynomial form (bgk_sincos2pif, kept by the exact-f32 kernel) */
__device__ __forceinline__ void v2_sincos2pi(float x, float* s, float* c) {
    const float f = __builtin_amdgcn_fractf(x);
    *s = __builtin_amdgcn_sinf(f);
    *c = __builtin_amdgcn_cosf(f);
}

/* ---- staging of a tile's inputs as row-major images (one row per sample) in LDS: y -> s_y [32][ys], the (featurised) conditioner
 * input -> s_p [32][nfs].  Contiguous, 16-byte aligned tensors (what a flow over separate field tensors hands over) travel by the
 * DMA path: a linear copy, no staging registers, no index arithmetic, every request in flight from the first cycle.  stage_issue
 * only issues those requests; the caller follows them with the operand loads that do not depend on the inputs (layer 0's first A
 * fragments, the first ring slots of layer 1), so that the one wait inside stage_finish covers all of them.  (Round 3 staged
 * [feature][sample] tiles through registers: ~50 instructions per element of index arithmetic, exec-masked LDS writes and branches --
 * 10 k of the 58 k cycles a wave spends on its tile, tools/r04_phase_ts.py.)
 * stage: 0 = per-lane loads (any strides / several tensors / periodic), 1 = the contiguous [32][d_c] tile copied by the DMA path IS
 * the feature tile (nfs = d_c), 2 = DMA of the raw tile behind the feature tile + an elementwise cos / sin pass. */
struct StageTiles { int d_c, periodic, nfs, ys, stage, y_dma, d; uint32_t magic_d; int rows, lane; };

__device__ __forceinline__ void stage_issue(const StageTiles& t, const CondSegs& cs, int64_t b0, const float* y_t, float* s_p, float* s_y) {
    if (t.y_dma) dma_tile32(s_y, y_t, 32 * t.d, t.rows * t.d, t.lane);
    if (t.stage == 1) dma_tile32(s_p, cs.ptr[0] + b0 * cs.ld[0], 32 * t.d_c, t.rows * t.d_c, t.lane);
    else if (t.stage == 2) dma_tile32(s_p + 32 * t.nfs, cs.ptr[0] + b0 * cs.ld[0], 32 * t.d_c, t.rows * t.d_c, t.lane);
}

#ifndef BGK_V2_SB
#define BGK_V2_SB 8
#endif
/* KA: the kernel's argument struct (its CondSegs member `cs` first): the segment table of the general path is indexed at run time and
 * read from the kernel-argument block by scalar loads -- indexing the by-value struct would make the compiler copy it to scratch */
template <class KA>
__device__ __forceinline__ void stage_finish(const StageTiles& t, const CondSegs& cs, int64_t b0, const float* y_t, int ldy32,
                                             float* s_p, float* s_y, float y_fill) {
    constexpr int SB = BGK_V2_SB;
    const int lane = t.lane, rows = t.rows, d = t.d, nfs = t.nfs, n_y = 32 * t.d;
    if (!t.y_dma) {                                  /* y with a row stride / unaligned: per-lane loads, SB in flight per lane */
        for (int base = 0; base < n_y; base += 64 * SB) {
            float vy[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int i = base + u * 64 + lane;
                const int r = (int)(__umul24((unsigned)i, t.magic_d) >> 20), c = i - (int)__umul24((unsigned)r, (unsigned)d);
                vy[u] = (i < n_y && r < rows) ? y_t[(int)__umul24((unsigned)r, (unsigned)ldy32) + c] : y_fill;
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int i = base + u * 64 + lane;
                const int r = (int)(__umul24((unsigned)i, t.magic_d) >> 20), c = i - (int)__umul24((unsigned)r, (unsigned)d);
                if (i < n_y) s_y[(int)__umul24((unsigned)r, (unsigned)t.ys) + c] = vy[u];
            }
        }
    }
    if (t.stage == 0) {
        typedef const __attribute__((address_space(4))) KA* ka_t;
        const ka_t kseg = (ka_t)__builtin_amdgcn_kernarg_segment_ptr();
        for (int sg = 0; sg < cs.n; ++sg) {
            const float* cond_t = kseg->cs.ptr[sg] + b0 * kseg->cs.ld[sg];
            const int ldc32 = (int)kseg->cs.ld[sg], w_c = kseg->cs.w[sg], n_c = 32 * w_c, col0 = kseg->cs.off[sg];
            const uint32_t magic_c = kseg->cs.magic[sg];
            for (int base = 0; base < n_c; base += 64 * SB) {
                float vc[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int i = base + u * 64 + lane;
                    const int r = (int)(__umul24((unsigned)i, magic_c) >> 20), c = i - (int)__umul24((unsigned)r, (unsigned)w_c);
                    vc[u] = (i < n_c && r < rows) ? cond_t[(int)__umul24((unsigned)r, (unsigned)ldc32) + c] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int i = base + u * 64 + lane;
                    const int r = (int)(__umul24((unsigned)i, magic_c) >> 20), c = i - (int)__umul24((unsigned)r, (unsigned)w_c);
                    if (i < n_c) {
                        float* f = s_p + (int)__umul24((unsigned)r, (unsigned)nfs) + col0 + c;
                        if (t.periodic) {                               /* WrapPeriodic featuriser (nn/periodic.py:30-37) */
                            float sv, cv;
                            v2_sincos2pi(vc[u], &sv, &cv);
                            f[0] = cv;
                            f[t.d_c] = sv;
                        } else {
                            f[0] = vc[u];
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* DMA pieces land in request order: everything requested so far has arrived */
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (t.stage == 2) {
        /* [cos 2 pi c | sin 2 pi c] of the raw tile (behind the feature tile) in its memory order: element i = (sample i / d_c, column i % d_c) */
        const float* raw = s_p + 32 * nfs;
        const uint32_t magic_c = cs.magic[0];
        for (int i = lane; i < 32 * t.d_c; i += 64) {
            const int r = (int)(__umul24((unsigned)i, magic_c) >> 20), c = i - (int)__umul24((unsigned)r, (unsigned)t.d_c);
            float sv, cv;
            v2_sincos2pi(raw[i], &sv, &cv);
            float* f = s_p + (int)__umul24((unsigned)r, (unsigned)nfs) + c;
            f[0] = cv;
            f[t.d_c] = sv;
        }
    }
    if (t.stage != 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

/* the finished tile image -> out rows.  lin: the image IS the memory image (contiguous, aligned rows): 16-byte pieces, the last 1..3
 * floats of a partial tile singly */
__device__ __forceinline__ void store_tile32(float* out_t, int ldo32, const float* s_y, int ys, int d, uint32_t magic_d, int rows, int lane, int lin) {
    const int n = rows * d;
    if (lin) {
        for (int i = lane * 4; i + 3 < n; i += 256) *reinterpret_cast<float4*>(out_t + i) = *reinterpret_cast<const float4*>(s_y + i);
        if (lane < (n & 3)) out_t[(n & ~3) + lane] = s_y[(n & ~3) + lane];
    } else {
        for (int i = lane; i < n; i += 64) {
            const int r = (int)(__umul24((unsigned)i, magic_d) >> 20), cc = i - (int)__umul24((unsigned)r, (unsigned)d);
            out_t[(int)__umul24((unsigned)r, (unsigned)ldo32) + cc] = s_y[(int)__umul24((unsigned)r, (unsigned)ys) + cc];
        }
    }
}

/* layer 0: A fragments of one k-step (4 tiles x {hi, lo}) */
struct L0Frag { uint4 v[4][2]; };
__device__ __forceinline__ void l0_request(L0Frag& f, const uint4* A0, int s, int lane) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f.v[m][0] = A0[((s * 4 + m) * 2 + 0) * 64 + lane];
#if !BGK_V2_BF16
        f.v[m][1] = A0[((s * 4 + m) * 2 + 1) * 64 + lane];
#endif
    }
}
/* one k-step: f = the lane's 8 consecutive features; `last`: the k-step that holds the constant-1 (bias) feature, rel0 = index of
 * element 0 relative to it */
__device__ __forceinline__ void l0_step(f32x16 (&h)[4], const L0Frag& fr, const float* f, bool last, int rel0) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f[e];
    if (last) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rel0 + e == 0 ? 1.0f : (rel0 + e > 0 ? 0.0f : v[e]);
    }
#if BGK_V2_BF16
    s16x8 bb;
#pragma unroll
    for (int e = 0; e < 8; ++e) bb[e] = __builtin_bit_cast(short, (__bf16)v[e]);
#pragma unroll
    for (int m = 0; m < 4; ++m)
        h[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, fr.v[m][0]), bb, h[m], 0, 0, 0);
#else
    h16x8 bhi, blo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#if BGK_V2_OVFL
        const float c = v[e];
#else
        const float c = __builtin_amdgcn_fmed3f(v[e], -65000.0f, 65000.0f);
#endif
        const _Float16 hv = (_Float16)c;
        bhi[e] = hv;
        blo[e] = (_Float16)(c - (float)hv);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, fr.v[m][1]), bhi, h[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, fr.v[m][0]), blo, h[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) h[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, fr.v[m][0]), bhi, h[m], 0, 0, 0);
#endif
}

/* SAVEP (training variant only): the spline parameters are written out.  The instance without it is the default training forward since
 * the backward recomputes them: with the write-out compiled in (and skipped at run time) its loop-invariant row pointers and LDS
 * addresses cost 18 spilled registers, one of them reloaded -- behind a vmcnt(0) -- in front of every chunk's transposition. */
template <int ACT, int INV, bool SAVEP = true>
__global__ __launch_bounds__(FTHREADS, 2) void coupling_rqs_dense_h2v2_kernel(V2Args a) {
    if (a.cs_dev) { a.c0 = a.cs_dev[1]; a.c1 = a.cs_dev[3]; a.c2 = a.cs_dev[5]; }   /* wave-uniform scalar loads */
#if BGK_V2_OVFL
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");                 /* MODE.FP16_OVFL */
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6;
    float* s_p = smem + (size_t)wave * a.lds_per_wave;   /* parameter chunk [128][ST]; first the conditioner feature tile [32][nfs] */
    float* s_y = s_p + 128 * ST;                          /* y / out tile [32][ys] + 32 spare floats */
    const int d = a.d;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * FW + wave;
    if (tile >= n_tiles) return;
    const SpK k{a.c2 * 1.44269504088896341f, a.c2 * a.sc.beta * 1.44269504088896341f, a.c2};
    const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
    const int ldy32 = (int)a.ldy, ldo32 = (int)a.ldo;
    const int n_y = 32 * d;

  {
    const int lane = (int)threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
    const unsigned voff = (unsigned)lane * 16u;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
    const float* y_t = a.y + b0 * a.ldy;            /* wave-uniform tile bases, 32-bit per-lane offsets */
    float* out_t = a.out + b0 * a.ldo;
    V2_TS(0);

    /* ---- stage the tile's inputs as row-major images (one row per sample) in LDS: y -> s_y [32][ys], the (featurised) conditioner
     * input -> s_p [32][nfs].  Contiguous, 16-byte aligned tensors (what a flow over separate field tensors hands over) travel by
     * the DMA path: a linear copy, no staging registers, no index arithmetic, every request in flight from the first cycle.  The
     * requests are followed at once by the operand loads that do not depend on the inputs (layer 0's first A fragments, the
     * first ring slots of layer 1), so that one wait covers all of them.  (Round 3 staged [feature][sample] tiles through
     * registers: ~50 instructions per element of index arithmetic, exec-masked LDS writes and branches -- 10 k of the 58 k cycles
     * a wave spends on its tile, tools/r04_phase_ts.py.) ---- */
    const int nfs = a.nfs, ys = a.ys;
    const StageTiles stg{a.d_c, a.periodic, nfs, ys, a.stage, a.y_dma, d, a.magic_d, rows, lane};
    stage_issue(stg, a.cs, b0, y_t, s_p, s_y);
    f32x16 h[4], acc[4];
    TFrag ring[RD];
    BFrag bf;
    L0Frag fa, fb;
    l0_request(fa, a.A0, 0, lane);
    Live<4> g1{acc, bf, a.A1, voff, ring};
    g1.start();
    V2_TS(12);
    stage_finish<V2Args>(stg, a.cs, b0, y_t, ldy32, s_p, s_y, 0.5f);
    V2_TS(1);

    /* ---- layer 0; B operand = 8 consecutive features of the lane's sample row, split on the fly.  The bias is the weight column of
     * a constant-1 feature n_in, which sits in the last k-step: it (and zeros behind it) are selected into the operand there, the
     * tile itself holds the n_in real features only.  A fragments of k-step s + 1 are requested before the MFMAs of k-step s. ---- */
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[m][r] = 0.0f;
    {
        const float* frow = s_p + j * nfs + 8 * hh;
        const int rel0 = 16 * (a.S0 - 1) + 8 * hh - n_in;       /* feature index of element 0 of the last k-step, relative to the constant-1 feature */
        for (int s = 0; s < a.S0; s += 2) {
            if (s + 1 < a.S0) l0_request(fb, a.A0, s + 1, lane);
            l0_step(h, fa, frow + 16 * s, s + 1 == a.S0, rel0);
            if (s + 1 < a.S0) {
                if (s + 2 < a.S0) l0_request(fa, a.A0, s + 2, lane);
                l0_step(h, fb, frow + 16 * (s + 1), s + 2 == a.S0, rel0);
            }
        }
    }

    V2_TS(2);
    /* ---- layer 1: events of k-steps 2t, 2t + 1 behind the activation of tile t + 1 ---- */
    {
        Live<4>& g = g1;
#if BGK_V2_SAVE
        /* z0 = layer-0 pre-activations, as full rows through the (now free) parameter-chunk buffer: [32][132] = 128 * ST floats;
         * after the ring start: the first MFMAs then wait for their operand loads only, not for these 16 stores (vmcnt is in order) */
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[m][r] *= a.c0;
#ifndef BGK_V2_ABL_NOZSAVE
        h2_store_rows128(h, a.z0, s_p, b0, rows, lane);
#endif
        const float c0_act = 1.0f;
#else
        const float c0_act = a.c0;
#endif
        NoLive none;
        act_split_tile<ACT, 0>(Hooks<NoLive, 0, 1>{none}, h[0], c0_act, bf);
        __builtin_amdgcn_sched_barrier(0);
        V2_TS(14);
        act_split_tile<ACT, 1>(Hooks<Live<4>, 0, 100>{g}, h[1], c0_act, bf);      /* hook i = event i (100 events) */
        act_split_tile<ACT, 2>(Hooks<Live<4>, 24, 100>{g}, h[2], c0_act, bf);
        act_split_tile<ACT, 3>(Hooks<Live<4>, 48, 100>{g}, h[3], c0_act, bf);
        __builtin_amdgcn_sched_barrier(0);
        V2_TS(15);
        g.template events<(72 * Live<4>::NEV) / 100, Live<4>::NEV>();
    }

    V2_TS(3);
    /* ---- layer 2, chunk 0: the same behind the activation of the layer-1 tiles ---- */
    float run = 0.0f;
    int oob_local = 0;
    {
        Live<4> g{h, bf, a.A2, voff, ring};
        g.start();
#if BGK_V2_SAVE
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] *= a.c1;
#ifndef BGK_V2_ABL_NOZSAVE
        h2_store_rows128(acc, a.z1, s_p, b0, rows, lane);
#endif
        const float c1_act = 1.0f;
#else
        const float c1_act = a.c1;
#endif
        NoLive none;
        act_split_tile<ACT, 0>(Hooks<NoLive, 0, 1>{none}, acc[0], c1_act, bf);
        __builtin_amdgcn_sched_barrier(0);
        act_split_tile<ACT, 1>(Hooks<Live<4>, 0, 100>{g}, acc[1], c1_act, bf);
        act_split_tile<ACT, 2>(Hooks<Live<4>, 24, 100>{g}, acc[2], c1_act, bf);
        act_split_tile<ACT, 3>(Hooks<Live<4>, 48, 100>{g}, acc[3], c1_act, bf);
        __builtin_amdgcn_sched_barrier(0);
        g.template events<(72 * Live<4>::NEV) / 100, Live<4>::NEV>();
    }
    V2_TS(4);
    /* ---- chunks: h -> LDS; spline(c) threaded through GEMM(c + 1) ---- */
    for (int c = 0; c < a.n_chunks; ++c) {
        int bins[3] = {0, 0, 0};
        const int nd = (d - c * DPC) < DPC ? (d - c * DPC) : DPC;
        if (c + 2 < a.n_chunks || (c + 2 == a.n_chunks && a.last_tiles > 2)) {
            chunk_piped<INV, 4, SAVEP>(a, k, s_p, s_y, c, hh, j, rows, run, oob_local, bins, h, bf, ring, voff, b0);
        } else if (c + 2 == a.n_chunks) {
            chunk_piped<INV, 2, SAVEP>(a, k, s_p, s_y, c, hh, j, rows, run, oob_local, bins, h, bf, ring, voff, b0);
        } else {
            NoLive none;
            chunk_to_lds(h, s_p, hh, j);
#if BGK_V2_SAVE
            if constexpr (SAVEP) save_chunk_params(a, s_p, c, lane, b0, rows);
#endif
#if (BGK_V2_ABL & 1)
            run += s_p[(threadIdx.x & 127) * ST + j];
#else
            spline_slot<INV, 0, 1>(none, a, k, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
            if (nd > 2) spline_slot<INV, 1, 1>(none, a, k, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
            if (nd > 4) spline_slot<INV, 2, 1>(none, a, k, s_p, s_y, c, nd, hh, j, rows, run, oob_local, bins);
#endif
        }
#if BGK_V2_TS
        V2_TS(5 + c);
#else
        if (a.bin_idx) {
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int q = 2 * it + hh;
                if (q < nd && j < rows) a.bin_idx[(b0 + j) * d + c * DPC + q] = bins[it];
            }
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (hh == 0 && j < rows) {
        if (a.accumulate) a.dlogp[b0 + j] += run; else a.dlogp[b0 + j] = run;
    }
    store_tile32(out_t, ldo32, s_y, ys, d, a.magic_d, rows, lane, a.out_lin);
    if (a.oob_count && __builtin_amdgcn_ballot_w64(oob_local != 0)) {     /* rare: inputs outside the spline domain */
        for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
        if (lane == 0) atomicAdd(a.oob_count, oob_local);
    }
    V2_TS(5 + a.n_chunks);
  }
}

/* tile images: DMA copies need a contiguous, 16-byte aligned tensor and a row length whose LDS bank pattern is harmless (the rows of a
 * tile are read one per lane: a row stride that is a multiple of 8 dwords would serialise every access 8-fold or worse) */
struct TilePlan { int nfs, ys, stage, y_dma, out_lin; };
TilePlan plan_tiles(const CondSegs& cs, int d_c, int periodic, const float* y, int64_t ldy, const float* out, int64_t ldo, int d, int tile_floats) {
    const int n_in = periodic ? 2 * d_c : d_c;
    const auto dma_ok = [](const float* p, int64_t ld, int w) { return ld == w && ((uintptr_t)p & 15) == 0 && (w & 7) != 0; };
    TilePlan t;
    t.y_dma = dma_ok(y, ldy, d) ? 1 : 0;
    t.ys = t.y_dma ? d : (d | 1);
    t.out_lin = (t.y_dma && ldo == d && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    const bool c_dma = cs.n == 1 && dma_ok(cs.ptr[0], cs.ld[0], d_c);
    t.stage = (c_dma && !periodic) ? 1 : ((c_dma && periodic && 32 * ((n_in | 1) + d_c) <= tile_floats) ? 2 : 0);
    t.nfs = t.stage == 1 ? d_c : (n_in | 1);
    return t;
}

SetK make_set(double low, double high, double min_bin, int K) {
    SetK s;
    const double span = high - low, scale = 1.0 - min_bin * K;
    /* the products of f32-rounded factors, like the first kernel's cfg (bgk_make_rqs_cfg) composes them */
    s.gnum = (float)((double)(float)span * (double)(float)scale);
    s.low = (float)low; s.high = (float)high;
    s.dstep = (float)((double)(float)span * (double)(float)min_bin);
    for (int k = 0; k < 7; ++k) s.kc[k] = (float)((double)(float)low + (double)(float)span * (double)(float)min_bin * (k + 1));
    return s;
}

/* i / d for i < 4096, d <= 127 as (i * M) >> 20 with the full-rate 24-bit multiply (v_mul_u32_u24; the 32-bit v_mul_lo / v_mul_hi
 * run at quarter rate): M = ceil(2^20 / d), error i (M d - 2^20) / (d 2^20) < 4096 / 2^20 < 1 / d, product < 2^32.  The row
 * strides multiplied the same way must stay below 2^24 (launch_h2 checks) */
uint32_t magic_div(int d) { return (uint32_t)(((1u << 20) + (uint32_t)d - 1) / (uint32_t)d); }

/* fill the kernel's segment table: `segs` (several conditioning tensors) or the single tensor (cond, ldc, d_c); false = bad input */
bool make_cond_segs(CondSegs& cs, const float* cond, int64_t ldc, int d_c, const BgkCondSegs* segs) {
    for (int i = 0; i < BGK_MAX_COND; ++i) { cs.ptr[i] = nullptr; cs.ld[i] = 0; cs.w[i] = 0; cs.off[i] = 0; cs.magic[i] = 0; }
    if (!segs || segs->n <= 1) {
        const float* p = segs && segs->n == 1 ? segs->ptr[0] : cond;
        const int64_t ld = segs && segs->n == 1 ? segs->ld[0] : ldc;
        if (!p || d_c <= 0 || ld >= (1 << 24)) return false;
        cs.n = 1; cs.ptr[0] = p; cs.ld[0] = ld; cs.w[0] = d_c; cs.magic[0] = magic_div(d_c);
        return true;
    }
    if (segs->n > BGK_MAX_COND) return false;
    int off = 0;
    for (int i = 0; i < segs->n; ++i) {
        if (!segs->ptr[i] || segs->w[i] <= 0 || segs->ld[i] < segs->w[i] || segs->ld[i] >= (1 << 24)) return false;
        cs.ptr[i] = segs->ptr[i]; cs.ld[i] = segs->ld[i]; cs.w[i] = segs->w[i]; cs.off[i] = off; cs.magic[i] = magic_div(segs->w[i]);
        off += segs->w[i];
    }
    cs.n = segs->n;
    return off == d_c;
}


#if !BGK_V2_SAVE && !BGK_V2_BF16
/* ---- affine (RealNVP) coupling layer with two conditioner networks of width 128 on the same event-threaded GEMM stream ----------
 * CouplingFlow(AffineTransformer(shift = DenseNet, scale = DenseNet)), nn/flow/transformer/affine.py:41-70.  Same contract and packed
 * operands as coupling_affine_dense_kernel<4, OT> (bgk_fused_affine.hip), which streams its A fragments through a 2-deep ring with
 * GEMM and activation phases alternating (0.85 ms per layer of cfg 5 at 2^20 samples, 22 % of it on the matrix pipe).  Here, per
 * network:  layer 0 from the staged conditioner tile;  layer-1 events threaded through the activation of the layer-0 tiles;  the
 * output layer's events (OT tiles per k-step) through the activation of the layer-1 tiles.  The shift network's result waits in LDS
 * ([dim][sample], in the conditioner tile's place) while the scale network runs on the same registers; y and the result pass
 * through a [dim][sample] LDS tile, so that global rows are read and written coalesced (one row-strided 4-byte access per lane and
 * dim costs 64 cache-line requests per instruction). */
struct AffV2Net { const uint4* A0; const uint4* A1; const uint4* A1b; const uint4* A2; float c0, c1, c1b, c2;    /* A1b: third hidden layer or NULL */
#if BGK_V2_AFFTRAIN
                  const float* cs;      /* device scale table {2^s, 2^-s} x 3 of bgk_pack_mlp_h2 (NULL: c0 .. c2 as given) */
                  float* z0; float* z1; /* the scaled pre-activations of the two hidden layers [B, 128], written for the backward */
#endif
};
struct AffV2Args {
    CondSegs cs; int d_c; int periodic; int S0;
    AffV2Net shift, scale; int has_shift, has_scale;
    const float* log_alpha; int preserve_volume, is_circular, inverse;
    const float* y; int64_t ldy; int64_t B; int d;
    float* out; int64_t ldo; float* dlogp; int accumulate;
    uint32_t magic_d;
    int lds_tile, lds_per_wave;  /* floats: conditioner / shift tile, whole wave slice (+ y / out tile) */
    int nfs, ys, stage, y_dma, out_lin;   /* the tile images (see StageTiles) */
#if BGK_V2_AFFTRAIN
    float* mu_out; float* s_out; int64_t ldms;     /* the networks' outputs [B, ldms] (ldms = 32 OT): shift values, scale values before tanh */
    int slab;                                      /* float offset of the 16-row store slab [16][32] inside the wave's conditioner-tile slice */
    int zt;                                        /* 32-unit tiles per row of the z arrays: 4 ([B, 128]) or 2 ([B, 64]: hidden layers of <= 64 units) */
#endif
};

#if BGK_V2_AFFTRAIN
/* NT tiles held in accumulator layout -> dst[b0 + r][32 m ..] as complete 128-byte lines: half a tile (16 rows) at a time through a
 * wave-private LDS slab [16][32] whose 16-byte pieces are XOR-swizzled by the row; every store instruction covers 8 rows x 128 B.
 * (The direct form -- 16-byte pieces of 32 rows per instruction -- costs 1.6 x the write traffic, bgk_mfma_h2.h.)  Stores go through a
 * buffer descriptor of the tile's rows: rows past the batch are out of range, no store is conditional (the in-order memory counter
 * stays exact for the operand loads around them). */
template <int NT>
__device__ __forceinline__ void aff_store_tiles(const f32x16 (&t)[4], float* dst, int pitch, float* slab, int64_t b0, int rows, int lane) {
    const int j = lane & 31, hh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + b0 * pitch), 0, rows * pitch * 4, 0x00020000);
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((j >> 4) == half) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(slab + (j & 15) * 32 + 4 * ((2 * q + hh) ^ (j & 7))) =
                        make_float4(t[m][4 * q], t[m][4 * q + 1], t[m][4 * q + 2], t[m][4 * q + 3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * 64 + lane, r = i >> 3, p = i & 7;
                const float4 v = *reinterpret_cast<const float4*>(slab + r * 32 + 4 * (p ^ (r & 7)));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ((16 * half + r) * pitch + 32 * m + 4 * p) * 4, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
}
struct AffSave { float* slab; int64_t b0; int rows, lane, zt; };      /* zt: tiles of a hidden layer that leave (2: [B, 64] arrays, 4: [B, 128]) */
#endif

/* layer 0: X = A0' * [features; 1] (l0_step); the A fragments of k-step s + 1 are requested before the MFMAs of k-step s (two fragment
 * sets, loop unrolled by two).  requested: `fa` already holds (or has in flight) the fragments of k-step 0. */
__device__ __forceinline__ void aff_layer0(const AffV2Net& n, int S0, const float* s_p, int nfs, int n_in, int lane, int j, int hh,
                                           f32x16 (&X)[4], L0Frag& fa, bool requested) {
    L0Frag fb;
    if (!requested) l0_request(fa, n.A0, 0, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) X[m][r] = 0.0f;
    const float* frow = s_p + j * nfs + 8 * hh;
    const int rel0 = 16 * (S0 - 1) + 8 * hh - n_in;
    for (int s = 0; s < S0; s += 2) {
        if (s + 1 < S0) l0_request(fb, n.A0, s + 1, lane);
        l0_step(X, fa, frow + 16 * s, s + 1 == S0, rel0);
        if (s + 1 < S0) {
            if (s + 2 < S0) l0_request(fa, n.A0, s + 2, lane);
            l0_step(X, fb, frow + 16 * (s + 1), s + 2 == S0, rel0);
        }
    }
}

/* one H x H layer: Y = A' * act(c X) + b', its events threaded through the activation of X's tiles 1..3 */
#if BGK_V2_AFFTRAIN
#define AFF_SAVE_PARAMS , float* zdst, const AffSave& sv
#define AFF_SAVE_Z(X) do { _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) X[m_][r_] *= c; \
                           if (sv.zt == 2) aff_store_tiles<2>(X, zdst, 64, sv.slab, sv.b0, sv.rows, sv.lane); \
                           else aff_store_tiles<4>(X, zdst, 128, sv.slab, sv.b0, sv.rows, sv.lane); \
                           c = 1.0f; } while (0)
#else
#define AFF_SAVE_PARAMS
#define AFF_SAVE_Z(X) do { } while (0)
#endif
template <int ACT>
__device__ __forceinline__ void aff_hidden(const uint4* A, float c, f32x16 (&X)[4], f32x16 (&Y)[4], BFrag& bf, TFrag (&ring)[RD], unsigned voff AFF_SAVE_PARAMS) {
    NoLive none;
    Live<4> g{Y, bf, A, voff, ring};
    g.start();
    AFF_SAVE_Z(X);       /* (training: behind the ring start -- the first MFMAs wait for their operand loads, not for these stores) */
    act_split_tile<ACT, 0>(Hooks<NoLive, 0, 1>{none}, X[0], c, bf);
    __builtin_amdgcn_sched_barrier(0);
    act_split_tile<ACT, 1>(Hooks<Live<4>, 0, 100>{g}, X[1], c, bf);       /* hook i = event i (100 events) */
    act_split_tile<ACT, 2>(Hooks<Live<4>, 24, 100>{g}, X[2], c, bf);
    act_split_tile<ACT, 3>(Hooks<Live<4>, 48, 100>{g}, X[3], c, bf);
    __builtin_amdgcn_sched_barrier(0);
    g.template events<(72 * Live<4>::NEV) / 100, Live<4>::NEV>();
}
/* the output layer: Y[0 .. OT) = A2' * act(c X) + b2' (unscaled), OT tiles per k-step */
template <int ACT, int OT>
__device__ __forceinline__ void aff_output(const uint4* A, float c, f32x16 (&X)[4], f32x16 (&Y)[4], BFrag& bf, TFrag (&ring)[RD], unsigned voff AFF_SAVE_PARAMS) {
    NoLive none;
    typedef Live<OT, OT> GO;                                                   /* 25 OT events: 6 OT per activated tile */
    GO g{Y, bf, A, voff, ring};
    g.start();
    AFF_SAVE_Z(X);
    act_split_tile<ACT, 0>(Hooks<NoLive, 0, 1>{none}, X[0], c, bf);
    __builtin_amdgcn_sched_barrier(0);
    act_split_tile<ACT, 1>(Hooks<GO, 0, 100>{g}, X[1], c, bf);
    act_split_tile<ACT, 2>(Hooks<GO, 24, 100>{g}, X[2], c, bf);
    act_split_tile<ACT, 3>(Hooks<GO, 48, 100>{g}, X[3], c, bf);
    __builtin_amdgcn_sched_barrier(0);
    g.template events<(72 * GO::NEV) / 100, GO::NEV>();
}
/* the layers behind layer 0.  Two hidden layers: X -> Y -> X[0 .. OT);  three (DEEP): X -> Y -> X -> Y[0 .. OT) */
#if BGK_V2_AFFTRAIN
template <int ACT, int OT, bool DEEP>
__device__ __forceinline__ void aff_layers(const AffV2Net& n, f32x16 (&X)[4], f32x16 (&Y)[4], BFrag& bf, TFrag (&ring)[RD], unsigned voff, const AffSave& sv) {
    static_assert(!DEEP, "the training forward takes two hidden layers");
    aff_hidden<ACT>(n.A1, n.c0, X, Y, bf, ring, voff, n.z0, sv);
    aff_output<ACT, OT>(n.A2, n.c1, Y, X, bf, ring, voff, n.z1, sv);
}
#else
template <int ACT, int OT, bool DEEP>
__device__ __forceinline__ void aff_layers(const AffV2Net& n, f32x16 (&X)[4], f32x16 (&Y)[4], BFrag& bf, TFrag (&ring)[RD], unsigned voff) {
    aff_hidden<ACT>(n.A1, n.c0, X, Y, bf, ring, voff);
    if constexpr (DEEP) {
        aff_hidden<ACT>(n.A1b, n.c1, Y, X, bf, ring, voff);
        aff_output<ACT, OT>(n.A2, n.c1b, X, Y, bf, ring, voff);
    } else {
        aff_output<ACT, OT>(n.A2, n.c1, Y, X, bf, ring, voff);
    }
}
#endif

/* tanh for the OUTPUT layer (log sigma): hardware exp2 + Newton-refined rcp above 0.625 (abs error ~1e-7), odd polynomial below
 * (the form of the other fused affine kernels, bgk_fused_affine.hip::r_tanh_out) */
__device__ __forceinline__ float aff_tanh_out(float x) {
    const float ax = __builtin_fabsf(x);
    const float dn = 1.0f + __builtin_amdgcn_exp2f(ax * 2.88539008177792681f);
    const float big = __builtin_copysignf(__builtin_fmaf(-2.0f, rcp_nr(dn), 1.0f), x);
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    const float small = __builtin_fmaf(p * z, x, x);
    return ax >= 0.625f ? big : small;
}

#ifndef BGK_V2_AFF_TS
#define BGK_V2_AFF_TS 0               /* profiling build (tools/r06_aff_ts.py): lane 0 stamps s_memtime at the affine kernel's phase boundaries and writes the stamps over the tile's first output row */
#endif
#if BGK_V2_AFF_TS
#define AFF_TS(k) do { ats_[k] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define AFF_TS(k) do { } while (0)
#endif

template <int ACT_S, int ACT_T, int OT, bool DEEP>
__global__ __launch_bounds__(FTHREADS, 2) void coupling_affine_dense_v2_kernel(AffV2Args a) {
#if BGK_V2_AFF_TS
    unsigned ats_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#if BGK_V2_OVFL
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");                 /* MODE.FP16_OVFL (act_split_pair carries no clamps) */
#endif
#if BGK_V2_AFFTRAIN
    if (a.shift.cs) { a.shift.c0 = a.shift.cs[1]; a.shift.c1 = a.shift.cs[3]; a.shift.c2 = a.shift.cs[5]; }    /* wave-uniform scalar loads */
    if (a.scale.cs) { a.scale.c0 = a.scale.cs[1]; a.scale.c1 = a.scale.cs[3]; a.scale.c2 = a.scale.cs[5]; }
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6;
    float* s_p = smem + (size_t)wave * a.lds_per_wave;   /* the conditioner feature tile [32][nfs] (+ the raw tile behind it); later the shift values [d][SROW] */
    float* s_y = s_p + a.lds_tile;                        /* y / out tile [32][ys]: the memory image of the rows */
    const int d = a.d;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * FW + wave;
    if (tile >= n_tiles) return;
    const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
    const int ldy32 = (int)a.ldy, ldo32 = (int)a.ldo;
    const int n_y = 32 * d;
    const int lane = (int)threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
    const unsigned voff = (unsigned)lane * 16u;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
    const float* y_t = a.y + b0 * a.ldy;
    float* out_t = a.out + b0 * a.ldo;

    /* ---- stage the tile images (see StageTiles): y -> s_y [32][ys], conditioner features -> s_p [32][nfs]; the first network's
     * layer-0 fragments travel with them ---- */
    const int nfs = a.nfs, ys = a.ys;
    const StageTiles stg{a.d_c, a.periodic, nfs, ys, a.stage, a.y_dma, d, a.magic_d, rows, lane};
    AFF_TS(0);
    stage_issue(stg, a.cs, b0, y_t, s_p, s_y);
    L0Frag fa;
    l0_request(fa, a.has_shift ? a.shift.A0 : a.scale.A0, 0, lane);
    stage_finish<AffV2Args>(stg, a.cs, b0, y_t, ldy32, s_p, s_y, 0.0f);
    AFF_TS(1);

    f32x16 h[4], acc[4];
    TFrag ring[RD];
    BFrag bf;
    /* ---- shift network: result (unscaled) in mu = h[0 .. OT) (three hidden layers: acc[0 .. OT)) ---- */
    f32x16 (&mu)[4] = DEEP ? acc : h;
    f32x16 (&t0)[4] = DEEP ? h : acc;          /* the scale network's layer-0 output: the array that does not hold mu */
#if BGK_V2_AFFTRAIN
    const AffSave sv{s_p + a.slab, b0, rows, lane, a.zt};
    if (a.has_shift) {
        aff_layer0(a.shift, a.S0, s_p, nfs, n_in, lane, j, hh, h, fa, true);
        aff_layers<ACT_S, OT, DEEP>(a.shift, h, acc, bf, ring, voff, sv);
        /* the shift values leave for the backward (the scale network's first fragments are requested in front of the stores) */
        if (a.has_scale) l0_request(fa, a.scale.A0, 0, lane);
#pragma unroll
        for (int m = 0; m < OT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) mu[m][r] *= a.shift.c2;
        aff_store_tiles<OT>(mu, a.mu_out, (int)a.ldms, sv.slab, b0, rows, lane);
        a.shift.c2 = 1.0f;
    }
    if (a.has_scale) aff_layer0(a.scale, a.S0, s_p, nfs, n_in, lane, j, hh, t0, fa, true);
#else
    if (a.has_shift) {
        aff_layer0(a.shift, a.S0, s_p, nfs, n_in, lane, j, hh, h, fa, true);
        AFF_TS(2);
        aff_layers<ACT_S, OT, DEEP>(a.shift, h, acc, bf, ring, voff);
        AFF_TS(3);
    }
    /* ---- scale network: layer 0 while the other array still holds the shift values; then they are parked in the (now free) tile.
     * (Its first fragments requested any earlier stay live across the GEMMs above: spills.) ---- */
    if (a.has_scale) aff_layer0(a.scale, a.S0, s_p, nfs, n_in, lane, j, hh, t0, fa, !a.has_shift);
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (a.has_shift) {
#pragma unroll
        for (int m = 0; m < OT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dim = drow(m, r, hh);
                if (dim < d) s_p[dim * SROW + j] = mu[m][r] * a.shift.c2;
            }
    }
#if BGK_V2_AFFTRAIN
    if (a.has_scale) {
        aff_layers<ACT_T, OT, DEEP>(a.scale, t0, mu, bf, ring, voff, sv);
        /* the scale values before tanh (what bgk_affine_backward differentiates) */
#pragma unroll
        for (int m = 0; m < OT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] *= a.scale.c2;
        aff_store_tiles<OT>(acc, a.s_out, (int)a.ldms, sv.slab, b0, rows, lane);
        a.scale.c2 = 1.0f;
    }
#else
    AFF_TS(4);
    if (a.has_scale) aff_layers<ACT_T, OT, DEEP>(a.scale, t0, mu, bf, ring, voff);       /* result in acc[0 .. OT) either way */
#endif
    AFF_TS(5);

    /* ---- affine tail (affine.py:41-70): lane (j, hh) owns sample j, dims drow(m, r, hh) ---- */
    const float alpha = a.has_scale ? bgk_expf(a.log_alpha[0]) : 0.0f;
    float lsum = 0.0f;
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float l = (a.has_scale && drow(m, r, hh) < d) ? aff_tanh_out(acc[m][r] * a.scale.c2) * alpha : 0.0f;
            acc[m][r] = l;
            lsum += l;
        }
    float total = lsum + __shfl_xor(lsum, 32);
    if (a.preserve_volume && a.has_scale) {
        const float mean = total / (float)d;
        lsum = 0.0f;
#pragma unroll
        for (int m = 0; m < OT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ls = drow(m, r, hh) < d ? acc[m][r] - mean : 0.0f;
                acc[m][r] = ls;
                lsum += ls;
            }
        total = lsum + __shfl_xor(lsum, 32);
    }
#pragma unroll
    for (int m = 0; m < OT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dim = drow(m, r, hh);
            if (dim < d) {
                const float yv = s_y[j * ys + dim];
                const float mm = a.has_shift ? s_p[dim * SROW + j] : 0.0f;
                const float ls = acc[m][r];
                const float sg = __builtin_amdgcn_exp2f((a.inverse ? -ls : ls) * 1.44269504088896341f);     /* |ls| <= exp(log_alpha): 1 ulp */
                float t = a.inverse ? sg * (yv - mm) : sg * yv + mm;
                if (a.is_circular) { t = t - __builtin_truncf(t); if (t < 0.0f) t = t + 1.0f; }
                s_y[j * ys + dim] = t;
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    AFF_TS(6);
    if (hh == 0 && j < rows) {
        const float dl = a.inverse ? -total : total;
        if (a.accumulate) a.dlogp[b0 + j] += dl; else a.dlogp[b0 + j] = dl;
    }
    store_tile32(out_t, ldo32, s_y, ys, d, a.magic_d, rows, lane, a.out_lin);
#if BGK_V2_AFF_TS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AFF_TS(7);
    if (lane == 0)
        for (int q = 0; q < 8; ++q) reinterpret_cast<unsigned*>(out_t)[q] = ats_[q];
#endif
}
#endif   /* affine layer */

#if BGK_V2_SAVE
/* ---- spline backward of a fused training layer with the parameters RECOMPUTED (round 5) ------------------------------------------
 * The training forward wrote the layer's 3 K + 1 = 25 spline parameters per element for the backward: 446 MB per layer at 2^18
 * samples x 17 dims, 0.11 of its 0.27 ms (the write-out does not overlap: stores and operand loads share one in-order counter),
 * and bgk_rqs_backward read them back.  They are one GEMM away from z1, which is saved anyway (the weight-gradient and input-gradient
 * kernels need it): this kernel redoes the conditioner's output layer on the matrix cores -- the SAME operand blocks, activation,
 * f16 split and MFMA event order as the forward (Live<>, act_split_tile: the parameters come out bit-identical) -- chunk by chunk
 * into the LDS chunk buffer, and runs bgk_rqs_vjp_element (the arithmetic of bgk_rqs_backward) on every element of the chunk from
 * there.  Reads z1 (512 B / sample) + y, g_out, g_dlogp; writes g_params in the reference's column order (what
 * bgk_dense_backward_dx and bgk_dense_weight_grad read) and g_y.  nn/flow/transformer/spline.py:109-188 + nn/dense.py:47-48. */
struct RcArgs {
    const float* z1; const uint4* A2; float c2; const float* cs_dev; int n_chunks, last_tiles;
    const float* y; int64_t ldy; int64_t B; int d; uint64_t nc_mask; int inverse;      /* nc_mask: bit per non-circular dim (its slot = its rank among them) */
    const float* g_out; int64_t ldgo; const float* g_dlogp; float* g_y; int64_t ldgy; float* g_params; int64_t ldgp;
    float* g_absmax; BgkRqsCfg cfg; int lds_per_wave;
};

typedef float rc_f4u __attribute__((ext_vector_type(4), aligned(4)));

/* slot IT of chunk c: element q = 2 IT + hh of sample j (the forward's assignment, spline_slot); invalid slots work on dim 0 of
 * the chunk and write nothing */
/* -DBGK_RC_TS=1: s_memtime phase sums of a wave's tile (chunk -> LDS | VJP | next GEMM | gradient stores), written over g_y[b0][0..7] */
#ifndef BGK_RC_TS
#define BGK_RC_TS 0
#endif
#ifndef BGK_RC_RD
#define BGK_RC_RD 4           /* (6 / 8: no difference, call 52) */
#endif

#if BGK_RC_TS
#define RC_Q(v) do { __builtin_amdgcn_sched_barrier(0); v = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RC_ADD(acc, t1, t0) acc += (t1) - (t0)
#else
#define RC_Q(v) do { } while (0)
#define RC_ADD(acc, t1, t0) do { } while (0)
#endif
struct RcTs { unsigned lds, vjp, gemm, st; };
struct RcIn { float x, gy; };      /* the element's input and output cotangent, requested one slot ahead */
/* Memory operations of this kernel and the one in-order counter (vmcnt: loads AND stores).  The compiler's wait for a load allows
 * exactly the operations it KNOWS were issued after it; a conditionally issued one in between (a store under an exec-mask branch, a
 * uniform `if` around a slot) makes that number uncertain and the wait becomes vmcnt(0) -- a wait for every store before it as well,
 * i.e. for the HBM write round trip of the previous chunk's 16 KB of gradients (first form of this kernel: 66 k of a tile's 165 k
 * cycles).  Hence: every global access here is unconditional (buffer descriptors of the tile: lanes with nothing to read or write
 * point out of range), the slot count of a chunk is a template parameter (straight-line code between a request and its use), the
 * slot table of the non-circular dims is a bit mask (no load), and the first slot's inputs are consumed BEFORE the chunk's store
 * pass is issued. */
struct RcTile { __amdgpu_buffer_rsrc_t y, go, gy, gp; int oy, ogo, ogy; };
constexpr int RC_OOB = 0x7ffffff0;
__device__ __forceinline__ RcIn rc_request(const RcTile& t, int c, int it, int nd, int hh) {
    const int q = 2 * it + hh, dim = c * DPC + (q < nd ? q : 0);
    return RcIn{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.y, t.oy + dim * 4, 0, 0)),
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.go, t.ogo + dim * 4, 0, 0))};
}

/* slot IT of chunk c: element q = 2 IT + hh of sample j (the forward's assignment, spline_slot); invalid slots work on dim 0 of the
 * chunk and write nothing.  MORE: another slot of this chunk follows (its inputs are requested here); otherwise the first slot of
 * chunk c + 1 (requested here as well, past the last chunk: dim 0 again, never used). */
template <int IT, bool MORE, bool FAST>
__device__ __forceinline__ void rc_vjp_slot(const RcArgs& a, const RcTile& tl, float* s_p, int c, int nd, int hh, int j, int rows, float gl, float& gmax, RcIn& in) {
    const int q = 2 * IT + hh;
    const bool valid = q < nd && j < rows;
    const int qq = q < nd ? q : 0;
    const int dim = c * DPC + qq, d = a.d;
    const float x = in.x, gy = in.gy;
    {   /* the next slot's inputs travel while this one computes */
        const int cn = MORE ? c : (c + 1 < a.n_chunks ? c + 1 : 0), itn = MORE ? IT + 1 : 0;
        const int ndn = MORE ? nd : ((d - cn * DPC) < DPC ? (d - cn * DPC) : DPC);
        in = rc_request(tl, cn, itn, ndn, hh);
    }
    float* pe = s_p + (qq * PPD) * ST + j;
    const bool has_slot = (a.nc_mask >> dim) & 1ull;
    float gx;
    /* the spline constants, opaque per slot: the VJP forms ~20 products and sums of them (the knots' base terms low + span min (k + 1),
     * scale factors, 1 / beta ..) that are the same for every element -- hoisted out of the chunk loop they are 20 more live
     * registers, which went to scratch (every scratch reload: a vmcnt(0)); recomputed per slot they are 20 instructions of ~700 */
    BgkRqsCfg cf = a.cfg;
    asm volatile("" : "+s"(cf.left), "+s"(cf.right), "+s"(cf.bottom), "+s"(cf.top), "+s"(cf.xspan), "+s"(cf.yspan));
    asm volatile("" : "+s"(cf.min_w), "+s"(cf.min_h), "+s"(cf.min_d), "+s"(cf.w_scale), "+s"(cf.h_scale), "+s"(cf.beta));
    /* (the gradients take the parameters' places in the chunk: every lane rewrites what it alone read) */
    const float m = bgk_rqs_vjp_element_lds<KB, FAST>(cf, a.inverse, pe, ST, a.c2, has_slot, q < nd, x, gy, gl, gx);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gx), tl.gy, valid ? tl.ogy + dim * 4 : RC_OOB, 0, 0);
    gmax = valid ? __builtin_fmaxf(gmax, m) : gmax;
    /* the maximum is formed HERE: left to the scheduler, the 25 max operations of a slot sink behind the next chunk's GEMM (their result is
     * not needed before the kernel ends) and keep the slot's 25 gradient values -- 75 registers per chunk -- alive across it: the
     * allocator then parks the B operand in scratch, eight registers per ring slot beyond two, and reloads it inside the GEMMs
     * (0.277 -> 0.252 ms per launch, call 53) */
    asm volatile("" : "+v"(gmax));
}

/* the chunk's gradients, LDS -> g_params in the reference's column order [w | h | s | slots]: 16-byte pieces (row r, dim q, set t,
 * half e) = chunk rows 25 q + 8 t + 4 e .. + 3 of column r; an instruction writes the <= 30 pieces of two sample rows.  Issued BEHIND
 * the next chunk's GEMM: its operand loads then have no stores in front of them (one in-order counter for loads and stores), and the
 * stores drain while the next chunk's VJP computes. */
__device__ __forceinline__ void rc_store_chunk(const RcArgs& a, const RcTile& tl, const float* s_p, int c, int nd, int lane, int rows) {
    const int p = lane & 31, half = lane >> 5, d = a.d;
    const int q = p / 6, te = p - 6 * q, t = te >> 1, e = te & 1;
    const bool mine = p < 6 * nd;
    const float* src = s_p + ((mine ? q : 0) * PPD + 8 * t + 4 * e) * ST;
    const int col = t * d * KB + (c * DPC + q) * KB + 4 * e;
    /* buffer stores on a descriptor of the tile's rows: a 32-bit offset per lane instead of a 64-bit address per store (sixteen address
     * pairs spilled to scratch here otherwise -- and every scratch reload is a vmcnt(0)); rows past the batch are out of range */
    const int vo = mine ? col * 4 : RC_OOB;
    const int pitch = (int)a.ldgp * 4;
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += 8) {           /* four row pairs per round: 16 LDS reads in flight, then 4 stores */
        rc_f4u v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 2 * u + half;
            v[u] = (rc_f4u){src[r], src[ST + r], src[2 * ST + r], src[3 * ST + r]};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[u]), tl.gp, vo, (r0 + 2 * u + half) * pitch, 0);
    }
    /* the non-circular dims' slot gradients: lane -> (row lane & 31, dims 2 i + (lane >> 5)); three rounds whatever nd */
    const int r = lane & 31;
#pragma unroll
    for (int q0 = 0; q0 < 6; q0 += 2) {
        const int qs = q0 + half, dim = c * DPC + qs;
        const bool has = qs < nd && ((a.nc_mask >> dim) & 1ull);
        const int slot = __builtin_popcountll(a.nc_mask & ((1ull << dim) - 1ull));
        const float v = s_p[((qs < nd ? qs : 0) * PPD + 3 * KB) * ST + r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), tl.gp, has ? r * pitch + (3 * d * KB + slot) * 4 : RC_OOB, 0, 0);
    }
}

/* chunk c (in h; NS slots): h -> LDS, the VJP of the chunk's elements (gradients back into the chunk), the next chunk's GEMM (NT tiles
 * per k-step; 0: none), then the chunk's gradients out */
constexpr int RC_RD = BGK_RC_RD;      /* A-fragment ring depth of this kernel's GEMMs: nothing is threaded through them here, so a tile-step costs operand
                                       * latency / ring depth (depth 4: 8 k of a 100-event GEMM's 11 k cycles are waits) */
template <int NT, int NS, bool FAST>
__device__ __forceinline__ void rc_chunk(const RcArgs& a, const RcTile& tl, float* s_p, int c, int nd, int lane_in, int rows, float gl, float& gmax, RcIn& in,
                                         f32x16 (&h)[4], const BFrag& bf, TFrag (&ring)[RC_RD], unsigned voff, RcTs& ts) {
    /* the lane index is made opaque per chunk: otherwise every per-lane address of the chunk body (75 parameter reads and write-backs,
     * the store pass) is hoisted out of the chunk loop as a loop invariant and spilled (72 registers) -- and scratch reloads are
     * vmcnt(0) waits */
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int j = lane & 31, hh = lane >> 5;
    Live<(NT > 0 ? NT : 4), 4, RC_RD> g{h, bf, a.A2 + (size_t)(c + 1) * GBLK * 64, voff, ring};
#if BGK_RC_TS
    unsigned t0, t1, t2, t3, t4;
#endif
    RC_Q(t0);
    chunk_to_lds(h, s_p, hh, j);
    RC_Q(t1); RC_ADD(ts.lds, t1, t0);
    __builtin_amdgcn_sched_barrier(0);     /* (the slots one after the other: interleaved by the scheduler they need twice the registers) */
    rc_vjp_slot<0, (NS > 1), FAST>(a, tl, s_p, c, nd, hh, j, rows, gl, gmax, in);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NS > 1) rc_vjp_slot<1, (NS > 2), FAST>(a, tl, s_p, c, nd, hh, j, rows, gl, gmax, in);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NS > 2) rc_vjp_slot<2, false, FAST>(a, tl, s_p, c, nd, hh, j, rows, gl, gmax, in);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    RC_Q(t2); RC_ADD(ts.vjp, t2, t1);
    /* Behind the VJP.  Measured alternatives (profiles/r05_ab_runs.txt, calls 45 / 47 / 52): the GEMM's events threaded through hook points
     * of the VJP the way the forward threads them through the spline: 313 us per launch instead of 271 (each event's operand wait also
     * waits for the g_y stores and input requests queued in between: one in-order counter); ring depth 6 / 8 and / or the ring started in
     * front of the VJP: within 1 % (once the allocator stopped parking the B operand in scratch: see rc_vjp_slot). */
    if constexpr (NT > 0) g.start();
    if constexpr (NT > 0) g.template events<0, Live<(NT > 0 ? NT : 4), 4, RC_RD>::NEV>();
    asm volatile("" : "+v"(in.x), "+v"(in.gy));      /* the next chunk's first inputs are awaited here, in front of the stores below */
    RC_Q(t3); RC_ADD(ts.gemm, t3, t2);
    rc_store_chunk(a, tl, s_p, c, nd, lane, rows);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    RC_Q(t4); RC_ADD(ts.st, t4, t3);
}

template <int ACT, bool FAST>
__global__ __launch_bounds__(FTHREADS, 2) void coupling_rqs_bwd_recompute_kernel(RcArgs a) {
    if (a.cs_dev) a.c2 = a.cs_dev[5];
#if BGK_V2_OVFL
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");                 /* MODE.FP16_OVFL, as in the forward */
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      /* uniform: the tile's row bases live in scalar registers */
    float* s_p = smem + (size_t)wave * a.lds_per_wave;   /* first the z1 tile [32][128] (16-byte pieces XOR-swizzled by the row), then the parameter chunks [128][ST] */
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * FW + wave;      /* (tiles in reverse order, so that the consumers start with the rows written last: no difference, call 43) */
    if (tile >= n_tiles) return;
    const int lane = (int)threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
    const unsigned voff = (unsigned)lane * 16u;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
    RcTs ts{0u, 0u, 0u, 0u};
#if BGK_RC_TS
    unsigned q0, q1, q2, q3;
#endif
    RC_Q(q0);

    /* z1 tile by DMA: instruction i = rows 2 i, 2 i + 1; lane -> row 2 i + (lane >> 5), 16-byte piece (lane & 31) ^ (row & 15) of it.
     * A lane then reads pieces 8 m + 2 q + hh of row j (accumulator layout: rows 32 m + 8 q + 4 hh .. + 3 of the layer output). */
    {
        const float* zt = a.z1 + b0 * 128;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 2 * i + (lane >> 5), rr = row < rows ? row : 0;
            const int piece = (lane & 31) ^ (row & 15);
            __builtin_amdgcn_global_load_lds((gvp_t)(zt + rr * 128 + piece * 4), (lvp_t)(s_p + i * 256), 16, 0, 0);
        }
    }
    f32x16 h[4];
    TFrag ring[RC_RD];
    BFrag bf;
    Live<4, 4, RC_RD> g0{h, bf, a.A2, voff, ring};
    g0.start();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RC_Q(q1);
    {
        f32x16 z[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int piece = (8 * m + 2 * q + hh) ^ (j & 15);
                const float4 v = *reinterpret_cast<const float4*>(s_p + j * 128 + piece * 4);
                z[m][4 * q] = v.x; z[m][4 * q + 1] = v.y; z[m][4 * q + 2] = v.z; z[m][4 * q + 3] = v.w;
            }
        NoLive none;
        act_split_tile<ACT, 0>(Hooks<NoLive, 0, 1>{none}, z[0], 1.0f, bf);
        act_split_tile<ACT, 1>(Hooks<NoLive, 0, 1>{none}, z[1], 1.0f, bf);
        act_split_tile<ACT, 2>(Hooks<NoLive, 0, 1>{none}, z[2], 1.0f, bf);
        act_split_tile<ACT, 3>(Hooks<NoLive, 0, 1>{none}, z[3], 1.0f, bf);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();             /* every lane has its z1 values: the tile's space takes the parameter chunks */
    RC_Q(q2);
    g0.template events<0, Live<4, 4, RC_RD>::NEV>();
    RC_Q(q3);
    const int jr = j < rows ? j : rows - 1;
    const float gl = (a.g_dlogp + b0)[jr];
    float gmax = 0.0f;
    const RcTile tl{__builtin_amdgcn_make_buffer_rsrc((void*)(a.y + b0 * a.ldy), 0, (int)(rows * a.ldy * 4), 0x00020000),
                    __builtin_amdgcn_make_buffer_rsrc((void*)(a.g_out + b0 * a.ldgo), 0, (int)(rows * a.ldgo * 4), 0x00020000),
                    __builtin_amdgcn_make_buffer_rsrc((void*)(a.g_y + b0 * a.ldgy), 0, (int)(rows * a.ldgy * 4), 0x00020000),
                    __builtin_amdgcn_make_buffer_rsrc((void*)(a.g_params + b0 * a.ldgp), 0, (int)(rows * a.ldgp * 4), 0x00020000),
                    jr * (int)a.ldy * 4, jr * (int)a.ldgo * 4, j * (int)a.ldgy * 4};
    RcIn in = rc_request(tl, 0, 0, a.d < DPC ? a.d : DPC, hh);
    for (int c = 0; c + 1 < a.n_chunks; ++c) {       /* whole chunks: 5 dims, 3 slots */
        if (c + 2 < a.n_chunks || a.last_tiles > 2) rc_chunk<4, 3, FAST>(a, tl, s_p, c, DPC, lane, rows, gl, gmax, in, h, bf, ring, voff, ts);
        else rc_chunk<2, 3, FAST>(a, tl, s_p, c, DPC, lane, rows, gl, gmax, in, h, bf, ring, voff, ts);
    }
    {
        const int c = a.n_chunks - 1, nd = a.d - c * DPC;
        if (nd > 4) rc_chunk<0, 3, FAST>(a, tl, s_p, c, nd, lane, rows, gl, gmax, in, h, bf, ring, voff, ts);
        else if (nd > 2) rc_chunk<0, 2, FAST>(a, tl, s_p, c, nd, lane, rows, gl, gmax, in, h, bf, ring, voff, ts);
        else rc_chunk<0, 1, FAST>(a, tl, s_p, c, nd, lane, rows, gl, gmax, in, h, bf, ring, voff, ts);
    }
    bgk_publish_absmax(a.g_absmax, gmax);
#if BGK_RC_TS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned q4;
    RC_Q(q4);
    if (lane == 0) {
        unsigned* o = reinterpret_cast<unsigned*>(a.g_y + b0 * a.ldgy);
        o[0] = q1 - q0; o[1] = q2 - q1; o[2] = q3 - q2; o[3] = ts.lds; o[4] = ts.vjp; o[5] = ts.gemm; o[6] = ts.st; o[7] = q4 - q0;
    }
#endif
}
#endif

}  // namespace

#if !BGK_V2_SAVE && !BGK_V2_BF16 && !BGK_V2_AFFTRAIN
int bgk_h2_variant = 2;
#endif

#if BGK_V2_SAVE
int bgk_rc_vjp_variant = 2;      /* 2 (default): the element VJP's knots on the hardware forms (BGK_VJP_FAST); 1: the deterministic forms of bgk_rqs_backward */
int bgk_launch_rqs_bwd_recompute(const char* what, const float* z1, const void* A2p, float c2, const float* cs_dev, int32_t act,
                                 const float* y, int64_t ldy, int64_t B, int32_t d, uint64_t circ_mask, int32_t inverse,
                                 double left, double right, double bottom, double top,
                                 double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
                                 const float* g_out, int64_t ldgo, const float* g_dlogp, float* g_y, int64_t ldgy,
                                 float* g_params, int64_t ldgp, float* g_absmax, void* stream) {
    RcArgs a;
    a.z1 = z1; a.A2 = reinterpret_cast<const uint4*>(A2p); a.c2 = c2; a.cs_dev = cs_dev;
    a.n_chunks = (d + DPC - 1) / DPC;
    a.last_tiles = ((d - (a.n_chunks - 1) * DPC) * PPD + 31) / 32;
    a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.nc_mask = ~circ_mask & (d >= 64 ? ~0ull : ((1ull << d) - 1ull)); a.inverse = inverse;
    BGK_CHECK_ARG(ldy < (1 << 24) && ldgo < (1 << 24) && ldgy < (1 << 24) && ldgp < (1 << 24), "%s: row stride too large", what);
    a.g_out = g_out; a.ldgo = ldgo; a.g_dlogp = g_dlogp; a.g_y = g_y; a.ldgy = ldgy; a.g_params = g_params; a.ldgp = ldgp;
    a.g_absmax = g_absmax;
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, KB);
    a.lds_per_wave = ((128 * ST + 3) / 4) * 4;
    static_assert(128 * ST >= 32 * 128, "the chunk buffer holds the z1 tile");
    const size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
    const int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "%s: batch too large for one launch", what);
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH(A, F) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(coupling_rqs_bwd_recompute_kernel<A, F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                              hipLaunchKernelGGL((coupling_rqs_bwd_recompute_kernel<A, F>), dim3((int)n_wg), dim3(FTHREADS), shmem, st, a); } while (0)
#define BGK_LAUNCH_F(A) do { if (bgk_rc_vjp_variant == 1) BGK_LAUNCH(A, false); else BGK_LAUNCH(A, true); } while (0)
    if (act == 1) BGK_LAUNCH_F(1); else if (act == 2) BGK_LAUNCH_F(2); else BGK_LAUNCH_F(3);
#undef BGK_LAUNCH_F
#undef BGK_LAUNCH
    return bgk_launch_status(what);
}
#endif

#if !BGK_V2_AFFTRAIN      /* (the affine training unit holds the affine kernel only) */
#if BGK_V2_SAVE
int bgk_launch_rqs_dense_h2v2_train(const char* what, float* z0, float* z1, float* params, int64_t ldp, const int32_t* src_col,
                                    const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
#elif BGK_V2_BF16
int bgk_launch_rqs_dense_h2v2_bf16(const char* what, const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
#else
int bgk_launch_rqs_dense_h2v2(const char* what, const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
#endif
                              const void* A0p, const void* A1p, const void* A2p, float c0, float c1, float c2, const float* cs_dev,
                              int32_t act, const float* y, int64_t ldy, int64_t B, int32_t d, uint64_t circ_mask, int32_t inverse,
                              double left, double right, double bottom, double top,
                              double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
                              float* out, int64_t ldo, float* dlogp, int32_t accumulate, int32_t* bin_idx, int32_t* oob_count,
                              void* stream, const BgkCondSegs* segs) {
    V2Args a;
    const int n_in = periodic ? 2 * d_c : d_c;
    BGK_CHECK_ARG(make_cond_segs(a.cs, cond, ldc, d_c, segs), "%s: bad conditioning segments", what);
    a.d_c = d_c; a.periodic = periodic;
    a.y = y; a.ldy = ldy; a.out = out; a.ldo = ldo; a.d = d; a.magic_d = magic_div(d);
    a.B = B; a.dlogp = dlogp; a.accumulate = accumulate; a.bin_idx = bin_idx; a.oob_count = oob_count;
    a.A0 = reinterpret_cast<const uint4*>(A0p); a.A1 = reinterpret_cast<const uint4*>(A1p); a.A2 = reinterpret_cast<const uint4*>(A2p);
    a.S0 = (n_in + 1 + 15) / 16;
    a.n_chunks = (d + DPC - 1) / DPC;
    a.last_tiles = ((d - (a.n_chunks - 1) * DPC) * PPD + 31) / 32;
    a.c0 = c0; a.c1 = c1; a.c2 = c2; a.cs_dev = cs_dev;
    a.circ_mask = circ_mask;
    a.sc.left = (float)left; a.sc.right = (float)right;
    /* bgflow forward = nflows inverse: the heights are searched, the widths evaluated; bgflow inverse: the other way round */
    const SetK sw = make_set(left, right, min_bin_width, KB), sh = make_set(bottom, top, min_bin_height, KB);
    a.sc.sa = inverse ? sw : sh;
    a.sc.sb = inverse ? sh : sw;
    const double beta = identity_init ? (0.6931471805599453 / (1.0 - min_derivative)) : 1.0;
    a.sc.beta = (float)beta; a.sc.kout = (float)(0.6931471805599453 / (double)(float)beta); a.sc.min_d = (float)min_derivative;
    const TilePlan tp = plan_tiles(a.cs, d_c, periodic, y, ldy, out, ldo, d, 128 * ST);
    a.nfs = tp.nfs; a.ys = tp.ys; a.stage = tp.stage; a.y_dma = tp.y_dma; a.out_lin = tp.out_lin;
    BGK_CHECK_ARG(32 * a.nfs + 16 <= 128 * ST, "%s: %d conditioner input features do not fit the LDS tile", what, n_in);
    a.lds_per_wave = ((128 * ST + 32 * a.ys + 32 + 3) / 4) * 4;
#if BGK_V2_SAVE
    a.z0 = z0; a.z1 = z1; a.params = params; a.ldp = ldp; a.src_col = src_col;
#endif
    const size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
    const int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "%s: batch too large for one launch", what);
    BGK_CHECK_ARG((int64_t)32 * (d > d_c ? d : d_c) < (1 << 16), "%s: tile index range", what);
    BGK_CHECK_ARG(ldy < (1 << 24) && ldo < (1 << 24), "%s: row stride too large", what);
    const int grid = (int)n_wg;
    hipStream_t st = (hipStream_t)stream;
#if BGK_V2_SAVE
#define BGK_LAUNCH(A, I) do { if (params) hipLaunchKernelGGL((coupling_rqs_dense_h2v2_kernel<A, I, true>), dim3(grid), dim3(FTHREADS), shmem, st, a); \
                              else hipLaunchKernelGGL((coupling_rqs_dense_h2v2_kernel<A, I, false>), dim3(grid), dim3(FTHREADS), shmem, st, a); } while (0)
#else
#define BGK_LAUNCH(A, I) hipLaunchKernelGGL((coupling_rqs_dense_h2v2_kernel<A, I>), dim3(grid), dim3(FTHREADS), shmem, st, a)
#endif
    if (act == 1) { if (inverse) BGK_LAUNCH(1, 1); else BGK_LAUNCH(1, 0); }
    else if (act == 2) { if (inverse) BGK_LAUNCH(2, 1); else BGK_LAUNCH(2, 0); }
    else { if (inverse) BGK_LAUNCH(3, 1); else BGK_LAUNCH(3, 0); }
#undef BGK_LAUNCH
    return bgk_launch_status(what);
}
#endif   /* !BGK_V2_AFFTRAIN */

#if !BGK_V2_SAVE && !BGK_V2_BF16
/* hidden width 128, two or three hidden layers; activations (shift, scale): both SiLU, both ReLU, both Tanh, or ReLU / Tanh.
 * Returns BGK_EUNSUPPORTED (no error text) for any other combination: bgk_fused_affine.hip::affine_dense_launch then runs its
 * streaming kernel. */
#if BGK_V2_AFFTRAIN
int bgk_launch_affine_dense_v2_train(const BgkAffTrainSave* save,
                               const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
#else
int bgk_launch_affine_dense_v2(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
#endif
                               const void* sA0, const void* sA1, const void* sA1b, const void* sA2, float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                               const void* tA0, const void* tA1, const void* tA1b, const void* tA2, float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                               const float* log_alpha, int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                               const float* y, int64_t ldy, int64_t B, int32_t d,
                               float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream, const BgkCondSegs* segs) {
#if BGK_V2_AFFTRAIN
    const char* what = "bgk_coupling_affine_dense_h2_train";
#else
    const char* what = "bgk_coupling_affine_dense_h2";
#endif
    AffV2Args a;
    const int n_in = periodic ? 2 * d_c : d_c;
    BGK_CHECK_ARG(make_cond_segs(a.cs, cond, ldc, d_c, segs), "%s: bad conditioning segments", what);
    a.d_c = d_c; a.periodic = periodic; a.S0 = (n_in + 1 + 15) / 16;
#if BGK_V2_AFFTRAIN
    a.shift = AffV2Net{(const uint4*)sA0, (const uint4*)sA1, (const uint4*)sA1b, (const uint4*)sA2, sc0, sc1, sc1b, sc2, save->s_cs, save->s_z0, save->s_z1};
    a.scale = AffV2Net{(const uint4*)tA0, (const uint4*)tA1, (const uint4*)tA1b, (const uint4*)tA2, tc0, tc1, tc1b, tc2, save->t_cs, save->t_z0, save->t_z1};
    a.mu_out = save->mu; a.s_out = save->s_raw; a.ldms = save->ldms; a.zt = save->ldz == 64 ? 2 : 4;
#else
    a.shift = AffV2Net{(const uint4*)sA0, (const uint4*)sA1, (const uint4*)sA1b, (const uint4*)sA2, sc0, sc1, sc1b, sc2};
    a.scale = AffV2Net{(const uint4*)tA0, (const uint4*)tA1, (const uint4*)tA1b, (const uint4*)tA2, tc0, tc1, tc1b, tc2};
#endif
    a.has_shift = sA0 != nullptr; a.has_scale = tA0 != nullptr;
    const bool deep = a.has_shift ? sA1b != nullptr : tA1b != nullptr;
#if BGK_V2_AFFTRAIN
    if (deep) return BGK_EUNSUPPORTED;
    BGK_CHECK_ARG((!a.has_shift || (save->s_z0 && save->s_z1 && save->mu)) && (!a.has_scale || (save->t_z0 && save->t_z1 && save->s_raw)),
                  "%s: null save buffer", what);
    BGK_CHECK_ARG(save->ldz == 64 || save->ldz == 128, "%s: ldz = %lld (64 | 128)", what, (long long)save->ldz);
    BGK_CHECK_ARG(save->ldms >= 32 * ((d + 31) / 32) && save->ldms % 4 == 0 && save->ldms < (1 << 20), "%s: ldms = %lld (a multiple of 4, >= 32 ceil(d / 32))",
                  what, (long long)save->ldms);
#endif
    if (a.has_shift && a.has_scale && (sA1b != nullptr) != (tA1b != nullptr)) return BGK_EUNSUPPORTED;
    const int as = a.has_shift ? s_act : t_act, at = a.has_scale ? t_act : s_act;
    if (!((as == at && as >= 1 && as <= 3) || (as == 2 && at == 3))) return BGK_EUNSUPPORTED;
    a.log_alpha = log_alpha; a.preserve_volume = preserve_volume; a.is_circular = is_circular; a.inverse = inverse;
    a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.magic_d = magic_div(d);
    const int OT = (d + 31) / 32;
    TilePlan tp = plan_tiles(a.cs, d_c, periodic, y, ldy, out, ldo, d, 32 * 128);
    a.nfs = tp.nfs; a.ys = tp.ys; a.stage = tp.stage; a.y_dma = tp.y_dma; a.out_lin = tp.out_lin;
    const int tile_f = 32 * a.nfs + (a.stage == 2 ? 32 * d_c : 0) + 16, park_f = d * SROW;
    a.lds_tile = (((tile_f > park_f ? tile_f : park_f) + 3) / 4) * 4;
#if BGK_V2_AFFTRAIN
    a.slab = a.lds_tile;             /* behind the conditioner tile AND the parked shift values: both are live while some network's z leaves */
    a.lds_tile += 16 * 32;
#endif
    a.lds_per_wave = a.lds_tile + ((32 * a.ys + 3) / 4) * 4;
    const size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
    const int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "%s: batch too large for one launch", what);
    BGK_CHECK_ARG(ldy < (1 << 24) && ldo < (1 << 24) && (int64_t)32 * d_c < 4096 && (int64_t)32 * d < 4096
                  && OT >= 1 && OT <= 3, "%s: outside the kernel's envelope", what);
    if (shmem > 160 * 1024) {
        bgk_set_error("%s: %d input features / %d dims do not fit the LDS tiles", what, n_in, d);
        return BGK_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH(S, T, O, D) do { if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(coupling_affine_dense_v2_kernel<S, T, O, D>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                    hipLaunchKernelGGL((coupling_affine_dense_v2_kernel<S, T, O, D>), dim3((int)n_wg), dim3(FTHREADS), shmem, st, a); } while (0)
#if BGK_V2_AFFTRAIN
#define BGK_LAUNCH_D(S, T, O) BGK_LAUNCH(S, T, O, false)
#else
#define BGK_LAUNCH_D(S, T, O) do { if (deep) BGK_LAUNCH(S, T, O, true); else BGK_LAUNCH(S, T, O, false); } while (0)
#endif
#define BGK_LAUNCH_O(S, T) do { if (OT == 1) BGK_LAUNCH_D(S, T, 1); else if (OT == 2) BGK_LAUNCH_D(S, T, 2); else BGK_LAUNCH_D(S, T, 3); } while (0)
    if (as == 1) BGK_LAUNCH_O(1, 1); else if (as == 3) BGK_LAUNCH_O(3, 3); else if (at == 2) BGK_LAUNCH_O(2, 2); else BGK_LAUNCH_O(2, 3);
#undef BGK_LAUNCH_D
#undef BGK_LAUNCH_O
#undef BGK_LAUNCH
    return bgk_launch_status(what);
}
#endif
