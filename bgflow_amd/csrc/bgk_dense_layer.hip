/* bgk_dense_layer.hip -- ONE Linear layer of a conditioner, y = act(x W^T + b)  (nn/dense.py:30-48: Linear (+ activation) per entry of
 * DenseNet._layers), for the networks the one-launch coupling kernels do not take: other depths than two / three hidden layers, hidden
 * layers wider than 256, more than 64 transformed dims, the README flow's [1, 4, 1] nets.  The layer-by-layer path of such a coupling is
 *   bgk_dense_layer x n_layers  ->  bgk_rqs_transform | bgk_affine_transform
 * with the activations of a layer in HBM between the launches.
 *
 * Roofline: HBM for the builder's shapes (4 (n_in + n_out) B per sample against 2 n_in n_out flops x 3 MFMAs per product: 256 -> 425
 * is 2.7 KB and 0.65 MFLOP executed per sample, 0.36 ms of traffic and 0.27 ms of matrix work at 2^20), matrix cores beyond ~ 400 x 400.
 *
 * Decomposition (gfx950): as in the coupling kernels a wave owns 32 samples and computes EVERY output feature for them, so x is read
 * once and y written once; D[feature, sample] = W[feature, k] X[k, sample] with v_mfma_f32_32x32x16_f16 in split-f16 form (hi + lo
 * f16 pairs, three MFMAs per product, f32 accumulate: f32-class).
 *   - B operand: the wave's [32 samples][n_in] tile arrives as row-contiguous 16-byte pieces (a wave instruction = 1 KB of consecutive
 *     row bytes) in LDS, is read back in fragment order (lane (kb, j): the 8 consecutive features 16 s + 8 kb .. + 7 of row j, two
 *     16-byte reads per k16-step; row stride == 4 mod 64 banks), brought under a per-SAMPLE power-of-two scale (the row's largest magnitude into
 *     [2^14, 2^15): inputs of any range, unlike the coupling kernels' bounded activations), split once and kept in registers as
 *     hi / lo fragments (8 per k-step) for every 128-row group of output features.  Loaded in fragment order straight from global
 *     memory (32-byte pieces of 32 different rows per instruction, every 128-byte line touched by 8 instructions) the 256-input
 *     layers ran 1.7 ms at 2^20 whether the fragments then sat in LDS or in registers (profiles/r05_ab_runs.txt calls 58 / 59);
 *   - A operand: the weights packed on the host (dense.py::pack_linear_layer) under a per-layer power-of-two scale, per 128 output
 *     rows the blocks (s, m, p) of the coupling kernels' layout (natural k order), streamed from L2 through a ring of 2 - 6 k-steps;
 *   - epilogue on the accumulators: unscale, + y (accumulate: the second and later 256-column passes of a wider input), + bias,
 *     activation (the reproducible polynomial SiLU / Tanh of bgk_detmath_pk.h); the 128-feature group leaves through the tile's LDS
 *     space as row-contiguous 16-byte pieces (32 lanes per 512-byte row segment).
 * k-steps per pass: template constant S in {1, 2, 4, 8, 12, 16} (the packer pads with zero columns); inputs wider than 256 columns run
 * as passes of <= 256 columns that accumulate into y.  Registers: 8 S (operand) + 64 (accumulators) + 96 (A ring): two waves per
 * SIMD up to S = 8, one wave on the unified 512-register file beyond.  LDS: 32 x max(16 S + 4, 132) floats per wave.
 */
#include "bgk_mfma_h2.h"
#include "bgk_detmath_pk.h"

namespace {

constexpr int LW = 4;                 /* waves per workgroup */
#ifndef BGK_LAYER_RING
#define BGK_LAYER_RING 6               /* A-ring depth of the one-wave-per-SIMD instances (S = 12, 16): 3 -> 6 is - 3 % (call 62) */
#endif

struct LayerArgs {
    const float* x; int64_t ldx; int n_in;       /* this pass' input columns (<= 16 S) */
    const uint4* A; int G;                       /* packed weights: G groups of 128 output rows x S k16-steps x 4 tiles x {hi, lo} */
    float c; const float* c_dev;                 /* 2^-s of the weights' scale; c_dev (if non-null): read it from c_dev[1] (bgk_pack_linear_layer) */
    const float* bias; int act;                  /* applied by the last pass: bias [n_out] or null; 0 none, 1 SiLU, 2 ReLU, 3 Tanh */
    float* y; int64_t ldy; int n_out; int64_t B;
    int accumulate;                              /* y = act(y + x W^T + b) */
    int x_al, y_al;                              /* rows of x / y start on 16-byte boundaries */
    int bias_lds;                                /* the bias vector (zero-padded to 128 G floats) sits in LDS behind the waves' tiles */
};

/* SiLU / tanh of the epilogue: hardware exp2 + a Newton-refined reciprocal (1 - 2 ulp; the reproducible polynomial forms of
 * bgk_detmath_pk.h cost 3 x the instructions and the epilogue was 40 % of the kernel: tools/r06_layer_ts.py).  tanh keeps the odd
 * polynomial below 0.625, where 1 - 2 / (e + 1) cancels. */
__device__ __forceinline__ float layer_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float layer_silu(float x) {
    return x * layer_rcp(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float layer_tanh(float x) {
    const float ax = __builtin_fabsf(x);
    const float big = __builtin_copysignf(__builtin_fmaf(-2.0f, layer_rcp(1.0f + __builtin_amdgcn_exp2f(ax * 2.88539008177792681f)), 1.0f), x);
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    return ax >= 0.625f ? big : __builtin_fmaf(p * z, x, x);
}

__device__ __forceinline__ float layer_act(float v, int act) {
    if (act == 2) return v > 0.0f ? v : 0.0f;
    return v;
}

/* item lane, lane + 64, lane + 128, ... of a row-major [rows][w] tile as (row, column) without a division per item */
struct RowWalk {
    int r, c, dr, dc, w;
    __device__ __forceinline__ RowWalk(int lane, int w_) : w(w_) { r = lane / w; c = lane - r * w; dr = 64 / w; dc = 64 - dr * w; }
    __device__ __forceinline__ void next() { r += dr; c += dc; if (c >= w) { c -= w; ++r; } }
};

#ifndef BGK_LAYER_TS
#define BGK_LAYER_TS 0              /* 1: lane 0 stamps s_memtime at the phase boundaries and writes the stamps over the tile's first output row (tools/r06_layer_ts.py) */
#endif
#if BGK_LAYER_TS
#define LAYER_TS(k) do { ts_[k] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define LAYER_TS(k) do { } while (0)
#endif

template <int S>
__global__ __launch_bounds__(LW * 64, S <= 8 ? 2 : 1) void dense_layer_kernel(LayerArgs a) {
#if BGK_LAYER_TS
    unsigned ts_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    constexpr int XS = 16 * S + 4;                  /* LDS row stride of the input tile, floats (== 4 mod 64 banks: 16-byte reads of 16 rows hit 64 banks) */
    constexpr int YS = 128 + 4;                     /* ... of a 128-feature output group */
    constexpr int PER_WAVE = 32 * (XS > YS ? XS : YS);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* s_t = smem + (size_t)wave * PER_WAVE;
    const float* s_bias = smem + (size_t)LW * PER_WAVE;             /* [128 G] (bias_lds) */
    if (a.bias_lds) {                                               /* the whole workgroup, before any wave leaves */
        float* sb = smem + (size_t)LW * PER_WAVE;
        for (int i = threadIdx.x; i < 128 * a.G; i += LW * 64) sb[i] = (a.bias && i < a.n_out) ? a.bias[i] : 0.0f;
        __syncthreads();
    }
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * LW + wave;
    if (tile >= n_tiles) return;
    const int64_t b0 = tile * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
    const bool live = j < rows;
    LAYER_TS(0);

    /* ---- the tile's rows: row-contiguous 16-byte pieces (a wave instruction = 1 KB of consecutive row bytes) into LDS [row][XS] ---- */
    if (a.x_al && (a.n_in & 3) == 0) {
        /* every request of the tile in flight before the first LDS write: 2 S pieces per lane at most, no branch (pieces past the
         * tile's end repeat a piece of its last row: same data to the same place) */
        RowWalk wk(lane, a.n_in >> 2);
        float4 t[2 * S];
        int off[2 * S];
#pragma unroll
        for (int it = 0; it < 2 * S; ++it) {
            const int r = wk.r < rows ? wk.r : rows - 1;
            t[it] = *reinterpret_cast<const float4*>(a.x + (b0 + r) * a.ldx + 4 * wk.c);
            off[it] = r * XS + 4 * wk.c;
            wk.next();
        }
#pragma unroll
        for (int it = 0; it < 2 * S; ++it) *reinterpret_cast<float4*>(s_t + off[it]) = t[it];
    } else {
        /* rows that do not start on 16-byte boundaries (17 conditioning features, a column slice): dword requests, consecutive lanes
         * on consecutive floats, 16 per lane in flight */
        RowWalk wk(lane, a.n_in);
        const int n_chunks = (rows * a.n_in + 1023) >> 10;
        for (int ch = 0; ch < n_chunks; ++ch) {
            float t[16];
            int off[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = wk.r < rows ? wk.r : rows - 1;
                t[it] = a.x[(b0 + r) * a.ldx + wk.c];
                off[it] = r * XS + wk.c;
                wk.next();
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) s_t[off[it]] = t[it];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    LAYER_TS(1);
    /* ---- fragment order (lane (kb, j): features 16 s + 8 kb .. + 7 of row j), the row's largest magnitude, split under its power-of-two scale ---- */
    h2_h16x8 bhi[S], blo[S];
    float inv_tile;
    {
        float v[S][8];
        float m = 0.0f;
        const bool full = a.n_in == 16 * S;          /* no padded k columns: nothing to mask (a row past the batch may hold anything: it only reaches its own, unwritten, outputs) */
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int col = 16 * s + 8 * hh + 4 * g;
                const float4 q = *reinterpret_cast<const float4*>(s_t + j * XS + col);       /* beyond the data: whatever LDS holds, discarded below */
                v[s][4 * g + 0] = (full || (live && col + 0 < a.n_in)) ? q.x : 0.0f;
                v[s][4 * g + 1] = (full || (live && col + 1 < a.n_in)) ? q.y : 0.0f;
                v[s][4 * g + 2] = (full || (live && col + 2 < a.n_in)) ? q.z : 0.0f;
                v[s][4 * g + 3] = (full || (live && col + 3 < a.n_in)) ? q.w : 0.0f;
            }
            /* the row's largest magnitude (per sample since round 6: an inf / NaN entry takes only its own row's scale along -- to 1 --
             * and that row's outputs are poisoned by the entry anyway) */
#pragma unroll
            for (int e = 0; e < 8; e += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[s][e]), __builtin_fabsf(v[s][e + 1])));
        }
        /* the scale is per SAMPLE (round 6; per 32-sample tile before): lanes j and j + 32 hold the two halves of row j's k range, the
         * accumulator column of lane (j, hh) is sample j again, so the unscale factor below is a per-lane value -- a row's precision
         * does not depend on which other rows share its tile (torch.nn.Linear's rows do not either) */
        m = __builtin_fmaxf(m, __shfl_xor(m, 32));
        const float sc = h2_pow2_scale(m, inv_tile);
#pragma unroll
        for (int s = 0; s < S; ++s) h2_split8_scaled(v[s], sc, bhi[s], blo[s]);
    }
    const float cu = (a.c_dev ? a.c_dev[1] : a.c) * inv_tile;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                             /* the tile's LDS space now carries the output groups */

    /* ---- every 128-row group of output features: GEMM over the S k-steps, epilogue on the accumulators, rows out through LDS ---- */
    LAYER_TS(2);
    for (int g = 0; g < a.G; ++g) {
        const uint4* Wg = a.A + (size_t)g * (S * 8 * 64);
        h2_f32x16 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
        constexpr int R = S == 8 ? 2 : (S >= 12 ? BGK_LAYER_RING : (S >= 3 ? 3 : S));         /* ring depth (S = 8: 64 + 64 + 2 x 32 registers leave room at two waves per SIMD) */
        H2A<4> ring[R];
#pragma unroll
        for (int s = 0; s < R - 1; ++s) h2a_load<4>(ring[s], Wg, s, lane);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s + R - 1 < S) h2a_load<4>(ring[(s + R - 1) % R], Wg, s + R - 1, lane);
            __builtin_amdgcn_sched_barrier(0);                   /* keep the prefetch above this step's MFMAs */
            h2_mfma3<4>(acc, ring[s % R], bhi[s], blo[s]);
        }
        if (g == 0) LAYER_TS(3);
        const float* yrow = a.y + (b0 + (live ? j : 0)) * a.ldy;
        const int width = a.n_out - 128 * g < 128 ? a.n_out - 128 * g : 128;      /* live features of this group */
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (32 * m >= width) continue;                       /* wave-uniform: padded tiles of the last group */
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fl = 32 * m + 8 * q + 4 * hh, f0 = 128 * g + fl;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[m][4 * q + e] * cu;
                if (a.accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += (live && f0 + e < a.n_out) ? yrow[f0 + e] : 0.0f;
                }
                if (a.bias_lds) {
                    const float4 bv = *reinterpret_cast<const float4*>(s_bias + f0);
                    o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
                } else if (a.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += f0 + e < a.n_out ? a.bias[f0 + e] : 0.0f;
                }
                if (a.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = layer_silu(o[e]);
                } else if (a.act == 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = layer_tanh(o[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = layer_act(o[e], a.act);
                }
                *reinterpret_cast<float4*>(s_t + j * YS + fl) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (g == 0) LAYER_TS(4);
        float* yg = a.y + b0 * a.ldy + 128 * g;
        if (a.y_al && (width & 3) == 0) {                        /* row-contiguous 16-byte pieces: 32 lanes per 512-byte row segment */
            RowWalk wk(lane, width >> 2);
            const int n_it = (rows * (width >> 2) + 63) >> 6;
            for (int it = 0; it < n_it; ++it) {
                if (wk.r < rows) *reinterpret_cast<float4*>(yg + wk.r * a.ldy + 4 * wk.c) = *reinterpret_cast<const float4*>(s_t + wk.r * YS + 4 * wk.c);
                wk.next();
            }
        } else {                                                 /* e.g. 425 spline parameters per row: dword stores, a row segment per two instructions */
            for (int r = 0; r < rows; ++r) {
                if (lane < width) yg[r * a.ldy + lane] = s_t[r * YS + lane];
                if (lane + 64 < width) yg[r * a.ldy + lane + 64] = s_t[r * YS + lane + 64];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (g == 0) LAYER_TS(5);
    }
#if BGK_LAYER_TS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LAYER_TS(6);
    if (lane == 0)
        for (int q = 0; q < 8; ++q) reinterpret_cast<unsigned*>(a.y + b0 * a.ldy)[q] = ts_[q];
#endif
}

/* ---- device-side operand packing (the weights of a training run change every step; dense.py::pack_linear_layer is the layout's
 * reference, tested against this): cs[0] = max |W| over the column block (atomicMax on the bit pattern, zeroed by the launcher);
 * every packing thread derives the scale 2^e, e = clamp(floor(log2(32768 / max)), -16, 24), from it, thread 0 publishes 2^-e in cs[1] ---- */
__global__ __launch_bounds__(256) void layer_max_kernel(const float* W, int64_t ldw, int n_out, int n_in, float* cs) {
    float m = 0.0f;
    const int64_t n = (int64_t)n_out * n_in;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / n_in;
        m = fmaxf(m, fabsf(W[r * ldw + (i - r * n_in)]));
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) {
        if (!(m < 3.0e38f)) m = 3.4e38f;                       /* inf / NaN weights: largest finite pattern (scale exponent 0 below) */
        atomicMax(reinterpret_cast<unsigned int*>(cs), __builtin_bit_cast(unsigned int, m));
    }
}

__global__ __launch_bounds__(256) void layer_pack_kernel(const float* W, int64_t ldw, int n_out, int n_in, int S, int G, uint4* out, float* cs) {
    const float m = cs[0];
    int e = 0;
    if (m > 0.0f && m < 3.0e38f) {
        e = (int)floorf(log2f(32768.0f / m));
        e = e < -16 ? -16 : (e > 24 ? 24 : e);
    }
    const float scale = ldexpf(1.0f, e);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) cs[1] = ldexpf(1.0f, -e);
    if (t >= (int64_t)G * S * 8 * 64) return;
    const int lane = (int)(t & 63), blk = (int)(t >> 6);
    const int i = lane & 31, kb = lane >> 5;
    const int p = blk & 1, mt = (blk >> 1) & 3, s = (blk >> 3) % S, g = (blk >> 3) / S;
    const int row = 128 * g + 32 * mt + i;
    uint16_t o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = 16 * s + 8 * kb + q;
        const float v = (row < n_out && k < n_in) ? W[(int64_t)row * ldw + k] * scale : 0.0f;
        const _Float16 h = (_Float16)v;
        const _Float16 r = p ? (_Float16)(v - (float)h) : h;
        o[q] = __builtin_bit_cast(uint16_t, r);
    }
    out[t] = *reinterpret_cast<const uint4*>(o);
}

/* Packed operands that follow the weights without a host-side version key (round 6): ONE workgroup reads the column block, forms a
 * 64-bit fingerprint of its bit patterns (sum over elements of a per-position mix: any changed element changes it, up to a 2^-64
 * collision) and its largest magnitude; equal to the fingerprint stored with the operands -> done; otherwise it re-packs the block
 * itself (the arithmetic of layer_max_kernel + layer_pack_kernel) and stores the new fingerprint.  A launch of a few microseconds in
 * front of every bgk_dense_layer call replaces a cache keyed on (data_ptr, torch's version counter), which updates through `.data`,
 * old-style optimizers or kernels writing through a view do not bump: stale weights, silently. */
/* transposed: the operands of the TRANSPOSED matrix -- element (row, k) = W[k * ldw + row] -- for dX = g W, the input gradient of the
 * Linear layer, as one more bgk_dense_layer call on g (autograd of nn/dense.py:47-48 without a library GEMM). */
__global__ __launch_bounds__(1024) void layer_refresh_kernel(const float* W, int64_t ldw, int n_out, int n_in, int S, int G, uint4* out, float* cs,
                                                             unsigned long long* state, int transposed) {
    __shared__ unsigned long long s_fp[16];
    __shared__ float s_m[16];
    __shared__ int s_same;
    const int tid = threadIdx.x;
    const int64_t n = (int64_t)n_out * n_in;
    unsigned long long fp = 0ull;
    float m = 0.0f;
    for (int64_t i = tid; i < n; i += 1024) {
        const int64_t r = i / n_in, k = i - r * n_in;
        const float v = transposed ? W[k * ldw + r] : W[r * ldw + k];
        unsigned long long x = ((unsigned long long)__builtin_bit_cast(unsigned, v) + 0x9E3779B9ull) * (unsigned long long)(2 * i + 1);
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        fp += x;
        m = fmaxf(m, fabsf(v));
    }
    for (int off = 32; off > 0; off >>= 1) { fp += __shfl_xor(fp, off); m = fmaxf(m, __shfl_xor(m, off)); }
    if ((tid & 63) == 0) { s_fp[tid >> 6] = fp; s_m[tid >> 6] = m; }
    __syncthreads();
    if (tid == 0) {
        fp = 0ull; m = 0.0f;
        for (int w = 0; w < 16; ++w) { fp += s_fp[w]; m = fmaxf(m, s_m[w]); }
        if (!(m < 3.0e38f)) m = 3.4e38f;
        s_same = state[1] == 1ull && state[0] == fp;
        if (!s_same) { state[0] = fp; state[1] = 1ull; cs[0] = m; }
        s_m[0] = m;
    }
    __syncthreads();
    if (s_same) return;
    m = s_m[0];
    int e = 0;
    if (m > 0.0f && m < 3.0e38f) {
        e = (int)floorf(log2f(32768.0f / m));
        e = e < -16 ? -16 : (e > 24 ? 24 : e);
    }
    const float scale = ldexpf(1.0f, e);
    if (tid == 0) cs[1] = ldexpf(1.0f, -e);
    for (int64_t t = tid; t < (int64_t)G * S * 8 * 64; t += 1024) {
        const int lane = (int)(t & 63), blk = (int)(t >> 6);
        const int i = lane & 31, kb = lane >> 5;
        const int p = blk & 1, mt = (blk >> 1) & 3, s = (blk >> 3) % S, g = (blk >> 3) / S;
        const int row = 128 * g + 32 * mt + i;
        uint16_t o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = 16 * s + 8 * kb + q;
            const float v = (row < n_out && k < n_in) ? (transposed ? W[(int64_t)k * ldw + row] : W[(int64_t)row * ldw + k]) * scale : 0.0f;
            const _Float16 h = (_Float16)v;
            const _Float16 r = p ? (_Float16)(v - (float)h) : h;
            o[q] = __builtin_bit_cast(uint16_t, r);
        }
        out[t] = *reinterpret_cast<const uint4*>(o);
    }
}

template <int S>
int launch_layer(const LayerArgs& a, hipStream_t st) {
    constexpr int XS = 16 * S + 4, YS = 128 + 4;
    size_t shmem = sizeof(float) * (size_t)LW * 32 * (XS > YS ? XS : YS);
    LayerArgs b = a;
    b.bias_lds = a.bias != nullptr && shmem + sizeof(float) * 128 * (size_t)a.G <= 160 * 1024;      /* (else: per-element loads in the epilogue) */
    if (b.bias_lds) shmem += sizeof(float) * 128 * (size_t)a.G;
    if (shmem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_layer_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int64_t n_wg = ((a.B + 31) / 32 + LW - 1) / LW;
    hipLaunchKernelGGL((dense_layer_kernel<S>), dim3((unsigned)n_wg), dim3(LW * 64), shmem, st, b);
    return bgk_launch_status("bgk_dense_layer");
}

}  // namespace

extern "C" int bgk_dense_layer_steps(int32_t n_in) {
    if (n_in <= 0 || n_in > 256) return -1;
    const int s = (n_in + 15) / 16;
    return s <= 2 ? s : (s <= 4 ? 4 : (s <= 8 ? 8 : (s <= 12 ? 12 : 16)));
}

extern "C" int bgk_pack_linear_layer(const float* W, int64_t ldw, int32_t n_out, int32_t n_in, void* Ap, float* cs, void* stream) {
    BGK_CHECK_ARG(W && Ap && cs && n_out > 0 && n_in > 0 && n_in <= 256 && ldw >= n_in, "bgk_pack_linear_layer: bad arguments (a column block of at most 256)");
    BGK_CHECK_ARG(((uintptr_t)Ap & 15) == 0, "bgk_pack_linear_layer: the operand buffer must be 16-byte aligned");
    const int S = bgk_dense_layer_steps(n_in), G = (n_out + 127) / 128;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(cs, 0, 2 * sizeof(float), st) != hipSuccess) return bgk_launch_status("bgk_pack_linear_layer");
    const int64_t n = (int64_t)n_out * n_in;
    const int g_max = (int)((n + 255) / 256 < 64 ? (n + 255) / 256 : 64);
    hipLaunchKernelGGL(layer_max_kernel, dim3(g_max), dim3(256), 0, st, W, ldw, n_out, n_in, cs);
    const int64_t threads = (int64_t)G * S * 8 * 64;
    hipLaunchKernelGGL(layer_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, W, ldw, n_out, n_in, S, G,
                       reinterpret_cast<uint4*>(Ap), cs);
    return bgk_launch_status("bgk_pack_linear_layer");
}

extern "C" int bgk_refresh_linear_layer(const float* W, int64_t ldw, int32_t n_out, int32_t n_in, int32_t transposed, void* Ap, float* cs, void* state,
                                        void* stream) {
    BGK_CHECK_ARG(W && Ap && cs && state && n_out > 0 && n_in > 0 && n_in <= 256 && ldw >= (transposed ? n_out : n_in),
                  "bgk_refresh_linear_layer: bad arguments (a column block of at most 256)");
    BGK_CHECK_ARG(((uintptr_t)Ap & 15) == 0 && ((uintptr_t)state & 7) == 0, "bgk_refresh_linear_layer: the operand buffer must be 16-byte, the state 8-byte aligned");
    const int S = bgk_dense_layer_steps(n_in), G = (n_out + 127) / 128;
    hipLaunchKernelGGL(layer_refresh_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, W, ldw, n_out, n_in, S, G, reinterpret_cast<uint4*>(Ap), cs,
                       reinterpret_cast<unsigned long long*>(state), transposed ? 1 : 0);
    return bgk_launch_status("bgk_refresh_linear_layer");
}

extern "C" int bgk_dense_layer(const float* x, int64_t ldx, int64_t B, int32_t n_in, const void* Ap, int32_t S, float c, const float* c_dev,
                               const float* bias, int32_t n_out, int32_t act, float* y, int64_t ldy, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(x && Ap && y && B > 0 && n_in > 0 && n_out > 0 && ldx >= n_in && ldy >= n_out, "bgk_dense_layer: bad arguments");
    BGK_CHECK_ARG(act >= 0 && act <= 3, "bgk_dense_layer: act %d (0 none, 1 SiLU, 2 ReLU, 3 Tanh)", act);
    BGK_CHECK_ARG(n_in <= 256 && S == bgk_dense_layer_steps(n_in), "bgk_dense_layer: %d input columns per pass (at most 256: wider inputs run as "
                  "accumulating passes) in %d k-steps (bgk_dense_layer_steps says %d)", n_in, S, bgk_dense_layer_steps(n_in));
    BGK_CHECK_ARG(((uintptr_t)Ap & 15) == 0, "bgk_dense_layer: packed weights must be 16-byte aligned");
    const int64_t n_wg = ((B + 31) / 32 + LW - 1) / LW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_dense_layer: batch too large for one launch");
    LayerArgs a;
    a.x = x; a.ldx = ldx; a.n_in = n_in;
    a.A = reinterpret_cast<const uint4*>(Ap); a.G = (n_out + 127) / 128;
    a.c = c; a.c_dev = c_dev; a.bias = bias; a.act = act;
    a.y = y; a.ldy = ldy; a.n_out = n_out; a.B = B; a.accumulate = accumulate;
    a.x_al = ((uintptr_t)x & 15) == 0 && (ldx & 3) == 0;
    a.y_al = ((uintptr_t)y & 15) == 0 && (ldy & 3) == 0;
    hipStream_t st = (hipStream_t)stream;
    switch (S) {
        case 1: return launch_layer<1>(a, st);
        case 2: return launch_layer<2>(a, st);
        case 4: return launch_layer<4>(a, st);
        case 8: return launch_layer<8>(a, st);
        case 12: return launch_layer<12>(a, st);
        default: return launch_layer<16>(a, st);
    }
}
