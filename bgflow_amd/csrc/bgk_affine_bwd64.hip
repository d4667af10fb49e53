/* bgk_affine_bwd64.hip -- backward of ONE conditioner network of an affine coupling with hidden layers of <= 64 units, input-gradient
 * chain AND weight / bias gradients in one launch (round 6; BASELINE cfg 2: DenseNet([32, 64, 64, 32]) shift / scale networks,
 * nn/flow/transformer/affine.py:35-43 + nn/dense.py:30-48 under loss.backward(), nn/training/trainers.py:156-163).
 *
 * The three-kernel form (bgk_mlp_backward_dx -> g_z1, g_z0 in HBM -> bgk_mlp_weight_grad reading them and z1, z0 again) moves
 * 4 (d + 2 n_in + 8 H) B per sample = 2.4 KB for cfg 2's networks; the weight gradients contract over the BATCH, so a wave that holds a
 * 32-sample tile of g_z and h = act(z) on chip can form its share of g^T h right there: this kernel reads g, z1, z0, x once
 * (4 (d + n_in + 2 H) = 0.77 KB per sample), writes the conditioner-input gradient, and keeps the three weight gradients of the network
 * (32 x 64 + 64 x 64 + 64 x 32 values = 8 accumulator tiles, 128 registers) in registers across all its tiles -- the "batch-contracting"
 * form.  Per tile:
 *   g_h1 = W2^T g            (B operand: the tile's rows of g; A: the transposed operands of bgk_pack_mlp_h2_t, staged once per workgroup in LDS)
 *   g_z1 = g_h1 act'(z1), h1 = act(z1);     dW2 += g^T h1,   db2 += sum g
 *   g_h0 = W1^T g_z1, g_z0 = g_h0 act'(z0), h0 = act(z0);   dW1 += g_z1^T h0, db1 += sum g_z1
 *   g_x  = W0^T g_z0 (+ what the caller adds);                dW0 += g_z0^T x,  db0 += sum g_z0
 * The batch contraction needs its operands with the SAMPLE index along k: what a lane holds in accumulator layout (lane = sample) is
 * transposed through a wave-private LDS tile [unit][sample]; columns of g and x are gathered from the tile's rows (cache hits).  All
 * products are split-f16 (hi + lo, three MFMAs, f32 accumulate) under power-of-two scales: g under the tensor's (bgk_affine_backward
 * publishes max |g|), g_z1 / g_z0 under the scale of the largest tile maximum the wave has seen so far -- the gradients accumulate in that
 * scaled domain and are rescaled (exactly, by a power of two) on the few occasions the running scale shrinks.  Per-wave partial gradients go to a workspace and
 * are summed in fixed order by wgrad_reduce_kernel's twin below (deterministic, no atomics).
 * Envelope: d <= 32, n_in <= 32 (not periodic), H0, H1 <= 64; everything else runs the three-kernel form.
 */
#include "bgk_mfma_h2.h"

namespace {

constexpr int QW = 4;              /* waves per workgroup */
#ifndef BGK_BWD64_OCC
#define BGK_BWD64_OCC 1            /* workgroups per CU = waves per SIMD: 1 -- the 8 gradient tiles (128 registers) + the chain's working set need the 512-register file */
#endif
constexpr int TP = 36;             /* row pitch (floats) of the transposed tile [unit][sample]: 16-byte aligned rows */
constexpr int OPB = 8 + 16 + 8;    /* 1 KiB operand blocks in LDS: T2 (2 k-steps x 2 tiles x {hi, lo}), T1 (4 x 2 x 2), T0 (4 x 1 x 2) */

struct Bwd64Args {
    const float* g; int64_t ldg; int d;
    const float* z1; const float* z0;
    const float* x; int64_t ldc; int n_in;
    const uint4* T2; const uint4* T1; const uint4* T0; const float* cs;
    int act; int64_t B;
    float* g_x; int64_t ldgx; const float* g_x_add; int64_t ldga;
    const float* g_absmax;
    float* pw2; float* pw1; float* pw0; float* pb2; float* pb1; float* pb0;
    int H1, H0, n_slabs;
    int vec_gx;          /* rows of g_cond (and of its addend) start on 16-byte boundaries and n_in is a multiple of 4 */
    /* RECOMP: the network's FORWARD operands (bgk_pack_mlp_h2, HT = 2) -- z0 / z1 are recomputed from the conditioner rows */
    const uint4* A0; const uint4* A1; const uint4* A2; int S0;
    /* TAIL (the scale network of a forward-direction coupling): the affine tail's backward in front of the network's; s_raw [B, lds]: the
     * network's saved output (!RECOMP) */
    const float* s_raw; int64_t lds;
    const float* y; int64_t ldy; const float* g_dl; const float* log_alpha; float* g_y; int64_t ldgy; float* pa; int vec_gy;
    float* g_mu;                                 /* TAIL + INVERSE (an inverse-direction layer): y = the layer's OUTPUT; g_mu [B, d] (pitch ldgy) = - g_y leaves for the shift network's launch */
};
constexpr int FWB = 12 + 18 + 9;   /* forward operand blocks in LDS (RECOMP): A0 (<= 3 k-steps x 2 tiles x {hi, lo}), A1 (4 x 2 x 2 + 2 bias), A2 (4 x 1 x 2 + 1) */

/* tanh of the scale network's OUTPUT (log sigma): the form of the forward kernels (bgk_affine_fwd64.hip::f64_tanh_out) */
__device__ __forceinline__ float q_tanh_out(float x) {
    const float ax = __builtin_fabsf(x);
    const float dn = 1.0f + __builtin_amdgcn_exp2f(ax * 2.88539008177792681f);
    const float rc = __builtin_amdgcn_rcpf(dn);
    const float big = __builtin_copysignf(__builtin_fmaf(-2.0f, __builtin_fmaf(__builtin_fmaf(-dn, rc, 1.0f), rc, rc), 1.0f), x);
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    const float small = __builtin_fmaf(p * z, x, x);
    return ax >= 0.625f ? big : small;
}

/* d = g * act'(z), h = act(z) for a pair (hardware exp / rcp; the forms of bgk_dense_backward_dx) */
__device__ __forceinline__ void q_act_grad2(int act, bgk_f2 z, bgk_f2 g, bgk_f2& gz, bgk_f2& h) {
    if (act == 1) {
        const bgk_f2 y = z * bgk_splat2(-1.44269504088896341f);
        bgk_f2 e; e.x = __builtin_amdgcn_exp2f(y.x); e.y = __builtin_amdgcn_exp2f(y.y);
        e = e + bgk_splat2(1.0f);
        bgk_f2 s; s.x = __builtin_amdgcn_rcpf(e.x); s.y = __builtin_amdgcn_rcpf(e.y);
        h = z * s;
        gz = g * (s * (bgk_splat2(1.0f) + z * (bgk_splat2(1.0f) - s)));
    } else if (act == 2) {
        h.x = z.x > 0.0f ? z.x : 0.0f; h.y = z.y > 0.0f ? z.y : 0.0f;
        gz.x = z.x > 0.0f ? g.x : 0.0f; gz.y = z.y > 0.0f ? g.y : 0.0f;
    } else {
        h = bgk_tanhf2_fast(z);
        gz = g * (bgk_splat2(1.0f) - h * h);
    }
}

/* acc (2 tiles, accumulator layout) * c -> g_z = . * act'(z) in place, h = act(z); z rows of 64 floats */
__device__ __forceinline__ void q_load_z(float4 (&zz)[8], const float* zrow, int hh) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) zz[4 * m + q] = *reinterpret_cast<const float4*>(zrow + 32 * m + 8 * q + 4 * hh);
}
__device__ __forceinline__ void q_act_backward(h2_f32x16 (&t)[2], h2_f32x16 (&hv)[2], float c, int act, const float4 (&zz)[8]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 z = zz[4 * m + q];
            bgk_f2 g0, g1, a0, a1;
            q_act_grad2(act, (bgk_f2){z.x, z.y}, (bgk_f2){t[m][4 * q] * c, t[m][4 * q + 1] * c}, g0, a0);
            q_act_grad2(act, (bgk_f2){z.z, z.w}, (bgk_f2){t[m][4 * q + 2] * c, t[m][4 * q + 3] * c}, g1, a1);
            t[m][4 * q] = g0.x; t[m][4 * q + 1] = g0.y; t[m][4 * q + 2] = g1.x; t[m][4 * q + 3] = g1.y;
            hv[m][4 * q] = a0.x; hv[m][4 * q + 1] = a0.y; hv[m][4 * q + 2] = a1.x; hv[m][4 * q + 3] = a1.y;
        }
}

/* two tiles held in accumulator layout (lane = sample j) -> LDS [unit][sample] */
__device__ __forceinline__ void q_transpose_out(const h2_f32x16 (&t)[2], float* sx, int j, int hh) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) sx[(32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh) * TP + j] = t[m][r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
/* the 8 samples 16 s + 8 kb .. + 7 of unit `u` out of the transposed tile */
__device__ __forceinline__ void q_read8(const float* sx, int u, int s, int kb, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(sx + u * TP + 16 * s + 8 * kb);
    const float4 b = *reinterpret_cast<const float4*>(sx + u * TP + 16 * s + 8 * kb + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void q_release(void) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct QFrag { h2_h16x8 hi, lo; };
__device__ __forceinline__ h2_f32x16 q_mfma3(h2_f32x16 c, const QFrag& a, const QFrag& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, c, 0, 0, 0);
}
__device__ __forceinline__ QFrag q_lds_frag(const uint4* s_op, int blk, int lane) {
    QFrag f;
    f.hi = __builtin_bit_cast(h2_h16x8, s_op[(blk + 0) * 64 + lane]);
    f.lo = __builtin_bit_cast(h2_h16x8, s_op[(blk + 1) * 64 + lane]);
    return f;
}

/* raw buffer loads on a descriptor of the tile's rows: a 32-bit per-lane offset + a wave-uniform offset per request instead of a 64-bit
 * address per element (the first form of this kernel spent 600 of its 4 000 instructions per tile on address arithmetic), and rows past
 * the batch / masked columns read as 0 from the hardware range check -- no per-element compare / select */
constexpr int Q_OOB = 0x7ffffff0;
__device__ __forceinline__ float q_ld1(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
typedef unsigned q_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 q_ld4(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const q_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    const unsigned w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];      /* (by value: __builtin_bit_cast of a vector ELEMENT picks element 0 every time) */
    return make_float4(__uint_as_float(w0), __uint_as_float(w1), __uint_as_float(w2), __uint_as_float(w3));
}
/* f32 -> f16 hi / lo operand halves WITHOUT inline assembly.  The asm form of the other kernels (h2_split_pair: v_cvt_pk_f16_f32 + two
 * v_fma_mix) is invisible to the compiler's hazard recogniser: a matrix instruction that reads a register such an asm statement wrote a
 * cycle or two earlier gets the OLD contents.  Elsewhere dozens of instructions sit between the split and its consumer; here (short ReLU
 * activation code, one wave per SIMD, nothing else to issue) they met: run-to-run different gradients, occasionally inf (the lo half
 * of an overflowed value), tools/r06_dbg_bwd64_direct.py.  Written with conversions and an fma the compiler knows, it selects the same
 * instructions (v_cvt_pk_f16_f32, v_fma_mix*) AND keeps the required distance. */
typedef _Float16 q_h2 __attribute__((ext_vector_type(2)));
typedef float q_f2 __attribute__((ext_vector_type(2)));
#ifndef BGK_Q_ASMSPLIT
#define BGK_Q_ASMSPLIT 0      /* experiment: 1 = the three-instruction asm split with an s_nop 1 behind the last pair (the wait states the compiler cannot know about) */
#endif
__device__ __forceinline__ void q_split8s(const float (&v)[8], float sc, QFrag& f) {
#if BGK_Q_ASMSPLIT
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h2_split_pair(v[2 * e] * sc, v[2 * e + 1] * sc, h[e], l[e]);
    asm volatile("s_nop 1" : "+v"(h[3]), "+v"(l[3]), "+v"(h[0]), "+v"(l[0]), "+v"(h[1]), "+v"(l[1]), "+v"(h[2]), "+v"(l[2]));
    f.hi = __builtin_bit_cast(h2_h16x8, make_uint4(h[0], h[1], h[2], h[3]));
    f.lo = __builtin_bit_cast(h2_h16x8, make_uint4(l[0], l[1], l[2], l[3]));
    return;
#endif
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const float a0 = v[e] * sc, a1 = v[e + 1] * sc;
        const q_h2 h = __builtin_convertvector((q_f2){a0, a1}, q_h2);
        f.hi[e] = h[0]; f.hi[e + 1] = h[1];
        f.lo[e] = (_Float16)__builtin_fmaf((float)h[0], -1.0f, a0);
        f.lo[e + 1] = (_Float16)__builtin_fmaf((float)h[1], -1.0f, a1);
    }
}
/* 8 unscaled values (activations, conditioner inputs): clamped to the f16 range like the forward's */
__device__ __forceinline__ void q_split8(const float (&v)[8], QFrag& f) { h2_split8(v, f.hi, f.lo); }

/* two tiles in accumulator layout x a power-of-two scale -> the B operands of the four k-steps over their 64 rows (h2_make_b_scaled) */
__device__ __forceinline__ void q_make_b(QFrag (&b)[4], const h2_f32x16 (&in)[2], float sc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in[s >> 1][8 * (s & 1) + e];
        q_split8s(v, sc, b[s]);
    }
}

/* the running power-of-two scale of a gradient operand whose products accumulate across tiles: the scale of the largest tile maximum
 * seen so far (monotone: it only ever shrinks); when it shrinks the accumulators (NA tiles, held in the scaled domain) follow by the
 * exact power-of-two ratio -- a handful of times per launch, against one multiply-add per accumulator element and tile for the
 * alternative (a temporary accumulator per tile, unscaled and added: 256 accumulator-register moves + 128 fma per tile) */
template <int NA>
__device__ __forceinline__ void q_running_scale(float tile_max, float& s_run, float& inv_run, h2_f32x16 (&acc)[NA]) {
    if (tile_max > 0.0f) {
        float inv_t;
        const float s_t = h2_pow2_scale(tile_max, inv_t);
        if (s_run == 0.0f || s_t < s_run) {
            if (s_run != 0.0f) {
                const float ratio = s_t * inv_run;
#pragma unroll
                for (int k = 0; k < NA; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[k][r] *= ratio;
            }
            s_run = s_t; inv_run = inv_t;
        }
    } else if (s_run == 0.0f) { s_run = 1.0f; inv_run = 1.0f; }
}

/* MODE = RECOMP + 2 TAIL + 4 INVERSE (a compile-time flag: as a run-time one in the tile loop it pushed the kernel's 104 spilled scalar
 * registers into a miscompile -- wrong bias / log_alpha sums in the forward-direction instance, tools history of round 6).
 * MODE = RECOMP + 2 TAIL (+ 4 INVERSE).
 * !RECOMP: z1 / z0 saved by the forward.  RECOMP (round 6): NOTHING is saved -- the wave redoes the network's forward on its tile (the
 * forward kernels' products in the forward kernels' order, operands in LDS beside the transposed ones: 44 - 57 matrix instructions) and
 * the training forward writes only its outputs.
 * !TAIL: g = the gradient w.r.t. the network's output, under the tensor's scale (g_absmax) or, without one, a running scale like s1 / s0
 * (the shift network of a forward-direction coupling: g_mu = g_out, read as it is).  TAIL: the scale network of a forward-direction
 * coupling -- g_s_raw = (g_out e^s y + g_dlogp) alpha (1 - tanh^2 s_raw) is formed on chip from y, g_out, g_dlogp and s_raw (saved, or
 * recomputed: the output layer too) with bgk_affine_backward's arithmetic (nn/flow/transformer/affine.py:41-70 under autograd),
 * g_y = g_out e^s leaves from here, the log_alpha gradient as one partial per workgroup: bgk_affine_backward's launch, its g_mu / g_s
 * arrays (2 x 4 d B per sample written and read again) and its atomics do not exist. */
template <int ACT, int MODE>
__global__ __launch_bounds__(QW * 64, BGK_BWD64_OCC) void affine_net_bwd64_kernel(Bwd64Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4* s_op = reinterpret_cast<uint4*>(smem);                         /* [OPB][64] operand blocks */
    const int lane_in = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool RECOMP = (MODE & 1) != 0, TAIL = (MODE & 2) != 0, INVERSE = (MODE & 4) != 0;     /* (INVERSE: with TAIL only) */
    const uint4* s_fw = s_op + OPB * 64;                                  /* RECOMP: [FWB][64] forward operand blocks */
    float* sx = smem + (OPB + (RECOMP ? FWB : 0)) * 256 + wave * (64 * TP); /* this wave's transposed tile */
    if (RECOMP) {
        const int nb0 = a.S0 * 4;
        for (int b = wave; b < FWB; b += QW) {
            const uint4* src = b < 12 ? (b < nb0 ? a.A0 + b * 64 : nullptr) : (b < 30 ? a.A1 + (b - 12) * 64 : (TAIL ? a.A2 + (b - 30) * 64 : nullptr));
            if (src) s_op[(OPB + b) * 64 + lane_in] = src[lane_in];
        }
    }
    /* ---- the network's transposed operands, once per workgroup: T2 blocks (s < 2, m < 2), T1 (s < 4, m < 2), T0 (s < 4; FT = 1) ---- */
    for (int b = wave; b < OPB; b += QW) {
        const uint4* src;
        if (b < 8) { const int p = b & 1, m = (b >> 1) & 1, s = b >> 2; src = a.T2 + ((s * 4 + m) * 2 + p) * 64; }
        else if (b < 24) { const int c = b - 8, p = c & 1, m = (c >> 1) & 1, s = c >> 2; src = a.T1 + ((s * 4 + m) * 2 + p) * 64; }
        else { const int c = b - 24; src = a.T0 + c * 64; }
        s_op[b * 64 + lane_in] = src[lane_in];
    }
    __syncthreads();
    const float c2 = a.cs[5], c1 = a.cs[3], c0 = a.cs[1];
    float inv_sg = 1.0f;
    const bool run_g = TAIL || a.g_absmax == nullptr;                      /* g under a running scale like s1 / s0 (no tensor maximum published) */
    float sg = run_g ? 0.0f : h2_pow2_scale(a.g_absmax[0], inv_sg);
    const float alpha = TAIL ? bgk_expf(a.log_alpha[0]) : 0.0f;
    float ga_sum = 0.0f;                                                   /* TAIL: the lane's share of d loss / d log_alpha / alpha */
    const int wslab = blockIdx.x * QW + wave;              /* the wave's position among all waves: its tiles are wslab, wslab + n_waves, .. */
    const int64_t n_tiles = (a.B + 31) / 32;
    const int d = a.d, n_in = a.n_in;
    const int ldg4 = (int)a.ldg * 4, ldc4 = (int)a.ldc * 4;

    /* the network's weight gradients, in the scaled domain of their g operand: dW2 under sg, dW1 / dW0 under their running scales */
    h2_f32x16 dW2[2], dW1[4], dW0[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dW2[0][r] = dW2[1][r] = 0.0f; dW1[0][r] = dW1[1][r] = dW1[2][r] = dW1[3][r] = 0.0f; dW0[0][r] = dW0[1][r] = 0.0f; }
    float bs2 = 0.0f, bs1[2] = {0.0f, 0.0f}, bs0[2] = {0.0f, 0.0f};
    float s1 = 0.0f, inv1 = 1.0f, s0 = 0.0f, inv0 = 1.0f;
    const h2_f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int64_t tile = wslab; tile < n_tiles; tile += (int64_t)a.n_slabs * QW) {
        /* the lane index is made opaque per tile: otherwise every per-lane offset of the loop body is hoisted out of the loop as an
         * invariant and kept live */
        int lane = lane_in;
        asm volatile("" : "+v"(lane));
        const int j = lane & 31, hh = lane >> 5;
        const int64_t b0 = tile * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + b0 * a.ldg), 0, (rows - 1) * ldg4 + d * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + b0 * a.ldc), 0, (rows - 1) * ldc4 + n_in * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z1 = __builtin_amdgcn_make_buffer_rsrc((void*)((RECOMP ? a.x : a.z1) + b0 * 64), 0, RECOMP ? 0 : rows * 256, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_z0 = __builtin_amdgcn_make_buffer_rsrc((void*)((RECOMP ? a.x : a.z0) + b0 * 64), 0, RECOMP ? 0 : rows * 256, 0x00020000);
        /* ---- every global request of the tile up front (one exposed round trip per tile: one wave per SIMD hides none): the lane's row
         * of g (columns 16 s + 8 hh ..), its rows of z1 / z0, its column of g and of the conditioner input (samples 16 s + 8 hh ..) ---- */
        float grow_v[2][8], gcol_v[2][8], xcol_v[2][8];
        float4 zz1[8], zz0[8];
        float xr[2][8];                          /* RECOMP: the lane's row of the conditioner input, features 16 s + 8 hh .. (the forward's layer-0 B operand) */
        float srv[16];                           /* TAIL && !RECOMP: the saved s_raw, accumulator layout */
        float yv[16], gv[16], gl = 0.0f;         /* TAIL: y and g_out in ACCUMULATOR layout (dims (r & 3) + 8 (r >> 2) + 4 hh of sample j), g_dlogp of sample j */
        {
            const int vrow = j * ldg4 + hh * 32;                                   /* row j, column 8 hh */
            const int vgc = j < d ? hh * 8 * ldg4 + j * 4 : Q_OOB;                  /* sample 8 hh, column j */
            const int vxc = j < n_in ? hh * 8 * ldc4 + j * 4 : Q_OOB;
            const int vz = j * 256 + hh * 16;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (!TAIL) {
                        grow_v[s][e] = (16 * s + 8 * hh + e < d) ? q_ld1(rs_g, vrow, (16 * s + e) * 4) : 0.0f;     /* (a row's columns past d belong to the next row) */
                        gcol_v[s][e] = q_ld1(rs_g, vgc, (16 * s + e) * ldg4);
                    }
                    xcol_v[s][e] = q_ld1(rs_x, vxc, (16 * s + e) * ldc4);
                    if (RECOMP) xr[s][e] = (16 * s + 8 * hh + e < n_in) ? q_ld1(rs_x, j * ldc4 + hh * 32, (16 * s + e) * 4) : 0.0f;
                }
            if (TAIL) {
                const int ldy4 = (int)a.ldy * 4;
                const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + b0 * a.ldy), 0, (rows - 1) * ldy4 + d * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g_dl + b0), 0, rows * 4, 0x00020000);
                gl = q_ld1(rs_l, j * 4, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool in = 8 * q + 4 * hh + e < d;
                        yv[4 * q + e] = in ? q_ld1(rs_y, j * ldy4 + hh * 16, (8 * q + e) * 4) : 0.0f;
                        gv[4 * q + e] = in ? q_ld1(rs_g, j * ldg4 + hh * 16, (8 * q + e) * 4) : 0.0f;
                    }
                if (!RECOMP) {
                    const int lds4 = (int)a.lds * 4;
                    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(a.s_raw + b0 * a.lds), 0, (rows - 1) * lds4 + d * 4, 0x00020000);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) srv[4 * q + e] = (8 * q + 4 * hh + e < d) ? q_ld1(rs_s, j * lds4 + hh * 16, (8 * q + e) * 4) : 0.0f;
                }
            }
            if (!RECOMP) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        zz1[4 * m + q] = q_ld4(rs_z1, vz, (32 * m + 8 * q) * 4);
                        zz0[4 * m + q] = q_ld4(rs_z0, vz, (32 * m + 8 * q) * 4);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float sraw[16];                          /* TAIL: the scale network's output before tanh (accumulator layout) */
        if (RECOMP) {
            /* ---- the network's forward on this tile, as bgk_affine_fwd64.hip::f64_network runs it: layer 0 from the conditioner rows (the
             * constant-1 feature n_in carries the bias), hidden layer, (TAIL) output layer; the scaled pre-activations stay in zz0 / zz1 ---- */
            const uint4* A0s = s_fw;
            const uint4* A1s = s_fw + 12 * 64;
            const uint4* A2s = s_fw + 30 * 64;
            h2_f32x16 fa[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) fa[m][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (s < a.S0) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int f = 16 * s + 8 * hh + e;
                        v[e] = f == n_in ? 1.0f : (s < 2 ? xr[s < 2 ? s : 0][e] : 0.0f);
                    }
                    h2_h16x8 bhi, blo;
                    h2_split8(v, bhi, blo);
                    H2A<2> fr;
                    h2a_load<2>(fr, A0s, s, lane);
                    h2_mfma3<2>(fa, fr, bhi, blo);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) fa[m][r] *= c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) zz0[4 * m + q] = make_float4(fa[m][4 * q], fa[m][4 * q + 1], fa[m][4 * q + 2], fa[m][4 * q + 3]);
                h2_act_tile(fa[m], 1.0f, ACT);
            }
            H2B<2> fb;
            h2_make_b<2>(fb, fa);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) fa[m][r] = 0.0f;
            h2_gemm_hidden<2, 2>(fa, fb, A1s, lane);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) fa[m][r] *= c1;
#pragma unroll
                for (int q = 0; q < 4; ++q) zz1[4 * m + q] = make_float4(fa[m][4 * q], fa[m][4 * q + 1], fa[m][4 * q + 2], fa[m][4 * q + 3]);
            }
            if (TAIL) {
                h2_act_tile(fa[0], 1.0f, ACT);
                h2_act_tile(fa[1], 1.0f, ACT);
                h2_make_b<2>(fb, fa);
                h2_f32x16 so[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) so[0][r] = 0.0f;
                h2_gemm_hidden<1, 2>(so, fb, A2s, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) sraw[r] = so[0][r] * c2;
            }
        } else if (TAIL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sraw[r] = srv[r];
        }
        if (TAIL) {
            __builtin_amdgcn_sched_barrier(0);
            /* ---- the affine tail's backward (forward direction, no volume preservation): per element th = tanh(s_raw), s = alpha th,
             * g_y = g e^s, g_s = g e^s y + g_dlogp, g_s_raw = g_s alpha (1 - th^2), d / d log_alpha += g_s th alpha ---- */
            float gs[16], gy[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool in = (r & 3) + 8 * (r >> 2) + 4 * hh < d;
                const float th = q_tanh_out(sraw[r]);
                const float ex = __builtin_amdgcn_exp2f((INVERSE ? -th : th) * alpha * 1.44269504088896341f);
                gy[r] = gv[r] * ex;
                /* forward: out = y e^s + mu, dlogp = sum s;  inverse: out = (y - mu) e^-s, dlogp = - sum s, and (y - mu) e^-s IS the output (yv) */
                const float gls = INVERSE ? -__builtin_fmaf(gv[r], yv[r], gl) : __builtin_fmaf(gy[r], yv[r], gl);
                gs[r] = in ? gls * alpha * __builtin_fmaf(-th, th, 1.0f) : 0.0f;
                ga_sum += in ? gls * th : 0.0f;
            }
            if (j < rows) {
                float* grow_y = a.g_y + (b0 + j) * a.ldgy;
                float* grow_m = INVERSE ? a.g_mu + (b0 + j) * a.ldgy : nullptr;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 8 * q + 4 * hh;
                    if (a.vec_gy && f0 + 3 < d) {
                        *reinterpret_cast<float4*>(grow_y + f0) = make_float4(gy[4 * q], gy[4 * q + 1], gy[4 * q + 2], gy[4 * q + 3]);
                        if (grow_m) *reinterpret_cast<float4*>(grow_m + f0) = make_float4(-gy[4 * q], -gy[4 * q + 1], -gy[4 * q + 2], -gy[4 * q + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (f0 + e < d) { grow_y[f0 + e] = gy[4 * q + e]; if (grow_m) grow_m[f0 + e] = -gy[4 * q + e]; }
                    }
                }
            }
            /* g_s_raw in the two layouts the network's backward reads g in: rows (lane (j, hh): dims 16 s + 8 hh ..: the quads a lane lacks
             * sit in its partner lane j, 1 - hh) and columns (lane = dim j, samples 16 s + 8 hh ..: through the transposed tile) */
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float own_lo = gs[4 * (2 * s2) + e], own_hi = gs[4 * (2 * s2 + 1) + e];     /* quads q = 2 s, 2 s + 1: dims 16 s + 4 hh + e, 16 s + 8 + 4 hh + e */
                    const float send = hh ? own_lo : own_hi;                                       /* hh = 0 keeps dims 16 s + 0..3, hh = 1 keeps 16 s + 12..15 */
                    const float recv = __shfl_xor(send, 32);
                    grow_v[s2][e] = hh ? recv : own_lo;                                              /* dims 16 s + 8 hh + e */
                    grow_v[s2][4 + e] = hh ? own_hi : recv;                                          /* dims 16 s + 8 hh + 4 + e */
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) sx[((r & 3) + 8 * (r >> 2) + 4 * hh) * TP + j] = gs[r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) q_read8(sx, j, s2, hh, gcol_v[s2]);
            q_release();
        }
        if (run_g) {
            /* the running scale of g (its products accumulate in dW2 across the wave's tiles) */
            float tm = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 8; ++e) tm = __builtin_fmaxf(tm, __builtin_fabsf(grow_v[s2][e]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tm = __builtin_fmaxf(tm, __shfl_xor(tm, off));
            q_running_scale<2>(tm, sg, inv_sg, dW2);
            __builtin_amdgcn_sched_barrier(0);
        }
        /* ---- g_h1 = W2^T g ---- */
        h2_f32x16 acc[2] = {zero16, zero16};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            QFrag bq;
            q_split8s(grow_v[s], sg, bq);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = q_mfma3(acc[m], q_lds_frag(s_op, ((s * 2 + m) * 2), lane), bq);
        }
        __builtin_amdgcn_sched_barrier(0);
        /* ---- g_z1, h1 ---- */
        h2_f32x16 hv[2];
        q_act_backward(acc, hv, c2 * inv_sg, ACT, zz1);
        __builtin_amdgcn_sched_barrier(0);
        /* ---- dW2 += g^T h1 (scaled by sg), db2: A = the lane's column of g, B = h1 through the transposed tile ---- */
        {
            QFrag aq[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bs2 += gcol_v[s][e];
                q_split8s(gcol_v[s], sg, aq[s]);
            }
            q_transpose_out(hv, sx, j, hh);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
                    q_read8(sx, 32 * t + j, s, hh, v);
                    QFrag bq;
                    q_split8(v, bq);
                    dW2[t] = q_mfma3(dW2[t], aq[s], bq);
                }
            q_release();
        }
        __builtin_amdgcn_sched_barrier(0);
        /* ---- g_h0 = W1^T g_z1; the same g_z1 as the A operand of dW1 (through the transposed tile), both under the running scale s1 ---- */
        QFrag a1[2][2];
        {
            q_running_scale<4>(h2_wave_absmax<2>(acc), s1, inv1, dW1);
            QFrag bf[4];
            q_make_b(bf, acc, s1);
            q_transpose_out(acc, sx, j, hh);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
                    q_read8(sx, 32 * m + j, s, hh, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs1[m] += v[e];
                    q_split8s(v, s1, a1[m][s]);
                }
            q_release();
            acc[0] = zero16; acc[1] = zero16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = q_mfma3(acc[m], q_lds_frag(s_op, 8 + ((s * 2 + m) * 2), lane), bf[s]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        q_act_backward(acc, hv, c1 * inv1, ACT, zz0);
        __builtin_amdgcn_sched_barrier(0);
        /* ---- dW1 += g_z1^T h0 (scaled by s1) ---- */
        q_transpose_out(hv, sx, j, hh);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            QFrag bq[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
                q_read8(sx, 32 * t + j, s, hh, v);
                q_split8(v, bq[s]);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 2; ++s) dW1[2 * m + t] = q_mfma3(dW1[2 * m + t], a1[m][s], bq[s]);
        }
        q_release();
        __builtin_amdgcn_sched_barrier(0);
        /* ---- g_x = W0^T g_z0 (+ the caller's addend); g_z0 as the A operand of dW0, under the running scale s0 ---- */
        QFrag a0[2][2];
        h2_f32x16 gx = zero16;
        {
            q_running_scale<2>(h2_wave_absmax<2>(acc), s0, inv0, dW0);
            QFrag bf[4];
            q_make_b(bf, acc, s0);
            q_transpose_out(acc, sx, j, hh);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float v[8];
                    q_read8(sx, 32 * m + j, s, hh, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs0[m] += v[e];
                    q_split8s(v, s0, a0[m][s]);
                }
            q_release();
            if (a.g_x) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    gx = q_mfma3(gx, q_lds_frag(s_op, 24 + s * 2, lane), bf[s]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (a.g_x && j < rows) {
            float* orow = a.g_x + (b0 + j) * a.ldgx;
            const float* arow = a.g_x_add ? a.g_x_add + (b0 + j) * a.ldga : nullptr;
            const float cu = c0 * inv0;
            if (a.vec_gx) {                 /* rows on 16-byte boundaries, n_in a multiple of 4: the four features 8 q + 4 hh .. of a register quad as one store */
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 8 * q + 4 * hh;
                    if (f0 < n_in) {
                        float4 o = make_float4(gx[4 * q] * cu, gx[4 * q + 1] * cu, gx[4 * q + 2] * cu, gx[4 * q + 3] * cu);
                        if (arow) { const float4 p = *reinterpret_cast<const float4*>(arow + f0); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
                        *reinterpret_cast<float4*>(orow + f0) = o;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (f < n_in) orow[f] = gx[r] * cu + (arow ? arow[f] : 0.0f);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        /* ---- dW0 += g_z0^T x (scaled by s0): B = the lane's column of the conditioner input ---- */
        {
            QFrag bq[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) q_split8(xcol_v[s], bq[s]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 2; ++s) dW0[m] = q_mfma3(dW0[m], a0[m][s], bq[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    /* ---- the workgroup's partial gradients: every wave leaves the scaled domain, waves 1 .. 3 hand their tiles to wave 0 through LDS (the
     * operand blocks are dead by now), wave 0 adds them in wave order and writes the slab: a quarter of the partial sums for the
     * reduction kernel to read (35 -> 9 MB per network at 2^20 samples), still a fixed summation order ---- */
    const int lane = lane_in, j = lane & 31, hh = lane >> 5;
    const int H1 = a.H1, H0 = a.H0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dW2[0][r] *= inv_sg; dW2[1][r] *= inv_sg;
        dW1[0][r] *= inv1; dW1[1][r] *= inv1; dW1[2][r] *= inv1; dW1[3][r] *= inv1;
        dW0[0][r] *= inv0; dW0[1][r] *= inv0;
    }
    __syncthreads();
    float* red = smem;                                   /* [3 waves][16 registers][64 lanes] */
    auto wg_sum = [&](h2_f32x16& t) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = t[r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < QW - 1; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[r] += red[(w * 16 + r) * 64 + lane];
        }
        __syncthreads();
    };
    wg_sum(dW2[0]); wg_sum(dW2[1]); wg_sum(dW1[0]); wg_sum(dW1[1]); wg_sum(dW1[2]); wg_sum(dW1[3]); wg_sum(dW0[0]); wg_sum(dW0[1]);
    {
        h2_f32x16 bsv;
#pragma unroll
        for (int r = 0; r < 16; ++r) bsv[r] = 0.0f;
        bsv[0] = bs2; bsv[1] = bs1[0]; bsv[2] = bs1[1]; bsv[3] = bs0[0]; bsv[4] = bs0[1]; bsv[5] = ga_sum;
        wg_sum(bsv);
        bs2 = bsv[0]; bs1[0] = bsv[1]; bs1[1] = bsv[2]; bs0[0] = bsv[3]; bs0[1] = bsv[4]; ga_sum = bsv[5];
    }
    if (wave != 0) return;
    const int slab = blockIdx.x;                          /* one slab per workgroup */
    if (TAIL) {                                           /* the workgroup's share of d loss / d log_alpha (fixed shuffle tree; the slabs are summed in order by the reduction) */
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ga_sum += __shfl_xor(ga_sum, off);
        if (lane == 0) a.pa[slab] = ga_sum * alpha;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 32 * t + j;
            if (row < d && col < H1) a.pw2[((int64_t)slab * d + row) * H1 + col] = dW2[t][r];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (32 * m + row < H1 && col < H0) a.pw1[((int64_t)slab * H1 + 32 * m + row) * H0 + col] = dW1[2 * m + t][r];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
            if (32 * m + row < H0 && j < n_in) a.pw0[((int64_t)slab * H0 + 32 * m + row) * n_in + j] = dW0[m][r];
    }
    if (j < d) a.pb2[((int64_t)slab * 2 + hh) * d + j] = bs2;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (32 * m + j < H1) a.pb1[((int64_t)slab * 2 + hh) * H1 + 32 * m + j] = bs1[m];
        if (32 * m + j < H0) a.pb0[((int64_t)slab * 2 + hh) * H0 + 32 * m + j] = bs0[m];
    }
}

/* fixed-order sum of the per-wave partials (the scheme of wgrad_reduce_kernel, bgk_wgrad.hip): element i of a gradient [n, k] (+ its
 * bias [n] behind it) = sum over slabs; 8 lanes per element each take every 8th slab, combined in ascending order through LDS */
struct QRed { const float* pw; const float* pb; int n, k; float* gW; float* gb; };
struct QRedGroup { QRed r[3]; int64_t first[4]; int n_slabs; const float* pa; float* g_log_alpha; };     /* pa: one partial of the log_alpha gradient per slab (or NULL) */
struct QRedPair { QRedGroup g[2]; int blocks0; };     /* two networks' partial sets in one launch: blocks [0, blocks0) belong to g[0] */
__device__ __forceinline__ void bwd64_reduce_body(const QRedGroup& rg, int block, int accumulate);
__global__ __launch_bounds__(256) void bwd64_reduce_kernel(QRedGroup rg, int accumulate) { bwd64_reduce_body(rg, (int)blockIdx.x, accumulate); }
__global__ __launch_bounds__(256) void bwd64_reduce2_kernel(QRedPair rp, int accumulate) {
    if ((int)blockIdx.x < rp.blocks0) bwd64_reduce_body(rp.g[0], (int)blockIdx.x, accumulate);
    else bwd64_reduce_body(rp.g[1], (int)blockIdx.x - rp.blocks0, accumulate);
}
__device__ __forceinline__ void bwd64_reduce_body(const QRedGroup& rg, int block, int accumulate) {
    __shared__ float s_part[8][32];
    if ((int64_t)block * 32 >= rg.first[3]) {                 /* the block behind the gradients: the log_alpha partials, summed in slab order by one thread */
        if (threadIdx.x == 0 && rg.pa && rg.g_log_alpha) {
            float t = 0.0f;
            for (int k = 0; k < rg.n_slabs; ++k) t += rg.pa[k];
            rg.g_log_alpha[0] = accumulate ? rg.g_log_alpha[0] + t : t;
        }
        return;
    }
    const int sub = threadIdx.x >> 5, el = threadIdx.x & 31;
    const int64_t gi = (int64_t)block * 32 + el;
    const int q = gi >= rg.first[2] ? 2 : (gi >= rg.first[1] ? 1 : 0);
    const QRed& o = rg.r[q];
    const int64_t i = gi - rg.first[q];
    const int64_t nk = (int64_t)o.n * o.k;
    const int64_t total = gi < rg.first[3] ? nk + o.n : 0;
    float acc = 0.0f;
    if (i < nk && i < total) {
        float a0 = 0.0f, a1 = 0.0f;
        int s = sub;
        for (; s + 8 < rg.n_slabs; s += 16) { a0 += o.pw[(int64_t)s * nk + i]; a1 += o.pw[(int64_t)(s + 8) * nk + i]; }
        if (s < rg.n_slabs) a0 += o.pw[(int64_t)s * nk + i];
        acc = a0 + a1;
    } else if (i < total) {
        const int col = (int)(i - nk);
        float a0 = 0.0f, a1 = 0.0f;
        int s = sub;
        for (; s + 8 < 2 * rg.n_slabs; s += 16) { a0 += o.pb[(int64_t)s * o.n + col]; a1 += o.pb[(int64_t)(s + 8) * o.n + col]; }
        if (s < 2 * rg.n_slabs) a0 += o.pb[(int64_t)s * o.n + col];
        acc = a0 + a1;
    }
    s_part[sub][el] = acc;
    __syncthreads();
    if (sub == 0) {
        float t = s_part[0][el];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += s_part[u][el];
        if (i < nk && i < total) { if (o.gW) o.gW[i] = accumulate ? o.gW[i] + t : t; }
        else if (i < total) { if (o.gb) o.gb[i - nk] = accumulate ? o.gb[i - nk] + t : t; }
    }
}

int bwd64_slabs(int64_t B) {
    const int64_t tiles = (B + 31) / 32;
    int64_t want = 256 * BGK_BWD64_OCC * QW;            /* BGK_BWD64_OCC workgroups per CU */
    if (tiles < want) want = ((tiles + QW - 1) / QW) * QW;
    return (int)((want < QW ? QW : want) / QW);          /* workgroups = slabs of partial sums (the waves of a workgroup are summed on chip) */
}

}  // namespace

extern "C" int64_t bgk_affine_net_backward64_workspace(int64_t B, int32_t d, int32_t H1, int32_t H0, int32_t n_in) {
    const int64_t s = bwd64_slabs(B);
    return s * ((int64_t)d * H1 + (int64_t)H1 * H0 + (int64_t)H0 * n_in) + 2 * s * ((int64_t)d + H1 + H0);
}

/* one network's launch + reduction (a: everything but the workspace pointers; mode = the kernel's MODE = RECOMP + 2 TAIL) */
static int bwd64_run(Bwd64Args a, int mode, float* workspace, float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0,
                     float* g_log_alpha, int accumulate, hipStream_t st, const char* what, QRedGroup* defer = nullptr, unsigned* defer_blocks = nullptr) {
    const int d = a.d, H1 = a.H1, H0 = a.H0, n_in = a.n_in;
    const int n_slabs = bwd64_slabs(a.B);
    a.n_slabs = n_slabs;
    a.vec_gx = a.g_x && n_in % 4 == 0 && ((uintptr_t)a.g_x & 15) == 0 && a.ldgx % 4 == 0 && (!a.g_x_add || (((uintptr_t)a.g_x_add & 15) == 0 && a.ldga % 4 == 0));
    float* p = workspace;
    a.pw2 = p; p += (int64_t)n_slabs * d * H1;
    a.pw1 = p; p += (int64_t)n_slabs * H1 * H0;
    a.pw0 = p; p += (int64_t)n_slabs * H0 * n_in;
    a.pb2 = p; p += (int64_t)2 * n_slabs * d;
    a.pb1 = p; p += (int64_t)2 * n_slabs * H1;
    a.pb0 = p; p += (int64_t)2 * n_slabs * H0;
    a.pa = p;
    const size_t shmem = sizeof(float) * ((size_t)(OPB + ((mode & 1) ? FWB : 0)) * 256 + (size_t)QW * 64 * TP);
#define BGK_LAUNCH_Q(A, M) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(affine_net_bwd64_kernel<A, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                hipLaunchKernelGGL((affine_net_bwd64_kernel<A, M>), dim3(n_slabs), dim3(QW * 64), shmem, st, a); } while (0)
#define BGK_LAUNCH_QA(M) do { if (a.act == 1) BGK_LAUNCH_Q(1, M); else if (a.act == 2) BGK_LAUNCH_Q(2, M); else BGK_LAUNCH_Q(3, M); } while (0)
    if (mode == 0) BGK_LAUNCH_QA(0); else if (mode == 1) BGK_LAUNCH_QA(1); else if (mode == 2) BGK_LAUNCH_QA(2); else if (mode == 3) BGK_LAUNCH_QA(3);
    else if (mode == 6) BGK_LAUNCH_QA(6); else BGK_LAUNCH_QA(7);
#undef BGK_LAUNCH_QA
#undef BGK_LAUNCH_Q
    QRedGroup rg;
    rg.r[0] = QRed{a.pw2, a.pb2, d, H1, gW2, gb2};
    rg.r[1] = QRed{a.pw1, a.pb1, H1, H0, gW1, gb1};
    rg.r[2] = QRed{a.pw0, a.pb0, H0, n_in, gW0, gb0};
    rg.first[0] = 0;
    rg.first[1] = (int64_t)d * H1 + d;
    rg.first[2] = rg.first[1] + (int64_t)H1 * H0 + H1;
    rg.first[3] = rg.first[2] + (int64_t)H0 * n_in + H0;
    rg.n_slabs = n_slabs;
    rg.pa = (mode & 2) ? a.pa : nullptr; rg.g_log_alpha = (mode & 2) ? g_log_alpha : nullptr;
    const unsigned blocks = (unsigned)((rg.first[3] + 31) / 32) + (rg.pa && rg.g_log_alpha ? 1u : 0u);
    if (defer) { *defer = rg; *defer_blocks = blocks; return bgk_launch_status(what); }       /* the caller reduces this set together with another */
    hipLaunchKernelGGL(bwd64_reduce_kernel, dim3(blocks), dim3(256), 0, st, rg, accumulate);
    return bgk_launch_status(what);
}

extern "C" int bgk_affine_net_backward64(const float* g, int64_t ldg, int32_t d, const float* z1, const float* z0,
                                         const float* cond, int64_t ldc, int32_t n_in, int32_t H1, int32_t H0,
                                         const void* T0, const void* T1, const void* T2, const float* cs, int32_t act, int64_t B,
                                         float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga, const float* g_absmax,
                                         float* workspace, int64_t workspace_floats,
                                         float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0, int32_t accumulate,
                                         void* stream) {
    if (B == 0) return 0;
    BGK_CHECK_ARG(g && z1 && z0 && cond && T0 && T1 && T2 && cs && workspace, "bgk_affine_net_backward64: null pointer");
    BGK_CHECK_ARG(B > 0 && d > 0 && n_in > 0 && H1 > 0 && H0 > 0 && ldg >= d && ldc >= n_in && act >= 1 && act <= 3 && (accumulate == 0 || accumulate == 1),
                  "bgk_affine_net_backward64: bad sizes");
    if (d > 32 || n_in > 32 || H1 > 64 || H0 > 64) {
        bgk_set_error("bgk_affine_net_backward64: d = %d, n_in = %d, hidden (%d, %d): the kernel takes d, n_in <= 32 and hidden layers of <= 64 units", d, n_in, H0, H1);
        return BGK_EUNSUPPORTED;
    }
    BGK_CHECK_ARG(workspace_floats >= bgk_affine_net_backward64_workspace(B, d, H1, H0, n_in), "bgk_affine_net_backward64: workspace too small");
    BGK_CHECK_ARG(ldg < (1 << 20) && ldc < (1 << 20), "bgk_affine_net_backward64: row stride too large");
    Bwd64Args a{};
    a.g = g; a.ldg = ldg; a.d = d; a.z1 = z1; a.z0 = z0; a.x = cond; a.ldc = ldc; a.n_in = n_in;
    a.T2 = (const uint4*)T2; a.T1 = (const uint4*)T1; a.T0 = (const uint4*)T0; a.cs = cs; a.act = act; a.B = B;
    a.g_x = g_cond; a.ldgx = ldgc; a.g_x_add = g_cond ? g_cond_add : nullptr; a.ldga = ldga; a.g_absmax = g_absmax;
    a.H1 = H1; a.H0 = H0;
    return bwd64_run(a, 0, workspace, gW2, gb2, gW1, gb1, gW0, gb0, nullptr, accumulate, (hipStream_t)stream, "bgk_affine_net_backward64");
}

extern "C" int64_t bgk_affine_coupling_backward64_workspace(int64_t B, int32_t d, int32_t n_in, int32_t sH1, int32_t sH0, int32_t tH1, int32_t tH0) {
    const int64_t a = bgk_affine_net_backward64_workspace(B, d, sH1, sH0, n_in), b = bgk_affine_net_backward64_workspace(B, d, tH1, tH0, n_in);
    return a + b + bwd64_slabs(B);         /* the two networks' partial sets side by side (one reduction launch for both) + the log_alpha partials */
}

extern "C" int bgk_affine_coupling_backward64(const float* cond, int64_t ldc, int32_t n_in, const float* y, int64_t ldy, int32_t d,
                                              const float* g_out, int64_t ldgo, const float* g_dlogp,
                                              const float* s_z0, const float* s_z1, const float* t_z0, const float* t_z1, const float* s_raw, int64_t lds,
                                              const void* sA0, const void* sA1, const void* sT0, const void* sT1, const void* sT2,
                                              const float* s_cs, int32_t s_act, int32_t sH1, int32_t sH0,
                                              const void* tA0, const void* tA1, const void* tA2, const void* tT0, const void* tT1, const void* tT2,
                                              const float* t_cs, int32_t t_act, int32_t tH1, int32_t tH0,
                                              const float* log_alpha, int32_t inverse, int64_t B,
                                              float* g_y, int64_t ldgy, float* g_mu, float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga,
                                              float* g_log_alpha, float* workspace, int64_t workspace_floats,
                                              float* const* s_grads, float* const* t_grads, int32_t accumulate, void* stream) {
    if (B == 0) return 0;
    const char* what = "bgk_affine_coupling_backward64";
    BGK_CHECK_ARG(cond && y && g_out && g_dlogp && g_y && log_alpha && workspace && s_grads && t_grads, "%s: null pointer", what);
    BGK_CHECK_ARG((inverse == 0 || inverse == 1) && (!inverse || g_mu), "%s: an inverse-direction layer needs the g_mu [B, d] buffer", what);
    BGK_CHECK_ARG(sA0 && sA1 && sT0 && sT1 && sT2 && s_cs && tA0 && tA1 && tA2 && tT0 && tT1 && tT2 && t_cs, "%s: null operand", what);
    BGK_CHECK_ARG(B > 0 && d > 0 && n_in > 0 && sH1 > 0 && sH0 > 0 && tH1 > 0 && tH0 > 0 && ldc >= n_in && ldy >= d && ldgo >= d && ldgy >= d
                  && s_act >= 1 && s_act <= 3 && t_act >= 1 && t_act <= 3 && (accumulate == 0 || accumulate == 1), "%s: bad sizes", what);
    if (d > 32 || n_in > 32 || sH1 > 64 || sH0 > 64 || tH1 > 64 || tH0 > 64) {
        bgk_set_error("%s: d = %d, n_in = %d, hidden (%d, %d) / (%d, %d): the kernel takes d, n_in <= 32 and hidden layers of <= 64 units", what, d, n_in, sH0, sH1, tH0, tH1);
        return BGK_EUNSUPPORTED;
    }
    BGK_CHECK_ARG(workspace_floats >= bgk_affine_coupling_backward64_workspace(B, d, n_in, sH1, sH0, tH1, tH0), "%s: workspace too small", what);
    BGK_CHECK_ARG(ldc < (1 << 20) && ldy < (1 << 20) && ldgo < (1 << 20), "%s: row stride too large", what);
    const bool saved = s_z0 || s_z1 || t_z0 || t_z1 || s_raw;          /* the forward's arrays: all five, or none (the networks are recomputed) */
    BGK_CHECK_ARG(!saved || (s_z0 && s_z1 && t_z0 && t_z1 && s_raw && lds >= d && lds < (1 << 20)), "%s: saved arrays: all five (z0, z1 of both networks, s_raw), or none", what);
    const int rec = saved ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;
    Bwd64Args a{};
    a.g = g_out; a.ldg = ldgo; a.d = d; a.x = cond; a.ldc = ldc; a.n_in = n_in; a.B = B; a.S0 = (n_in + 1 + 15) / 16;
    a.g_x = g_cond; a.ldgx = ldgc; a.ldga = ldga;
    /* the scale network first: tail backward (g_y, the log_alpha gradient) + its chain and weight gradients; g_cond = W0^T g_z0 + g_cond_add */
    a.T2 = (const uint4*)tT2; a.T1 = (const uint4*)tT1; a.T0 = (const uint4*)tT0; a.cs = t_cs; a.act = t_act; a.H1 = tH1; a.H0 = tH0;
    a.A0 = (const uint4*)tA0; a.A1 = (const uint4*)tA1; a.A2 = (const uint4*)tA2;
    a.g_x_add = g_cond ? g_cond_add : nullptr;
    a.y = y; a.ldy = ldy; a.g_dl = g_dlogp; a.log_alpha = log_alpha; a.g_y = g_y; a.ldgy = ldgy;
    a.vec_gy = ((uintptr_t)g_y & 15) == 0 && ldgy % 4 == 0 && (!inverse || ((uintptr_t)g_mu & 15) == 0);
    a.g_mu = g_mu;
    a.z1 = t_z1; a.z0 = t_z0; a.s_raw = s_raw; a.lds = lds;
    QRedPair rp;
    unsigned nb0 = 0, nb1 = 0;
    int rc = bwd64_run(a, 2 + rec + (inverse ? 4 : 0), workspace, t_grads[0], t_grads[1], t_grads[2], t_grads[3], t_grads[4], t_grads[5], g_log_alpha, accumulate, st, what,
                       &rp.g[0], &nb0);
    if (rc) return rc;
    float* ws_shift = workspace + bgk_affine_net_backward64_workspace(B, d, tH1, tH0, n_in) + bwd64_slabs(B);
    /* the shift network: g_mu = g_out (forward direction) or - g_out e^-s (inverse: written by the launch above); its conditioner-input
     * gradient is added to the scale network's */
    if (inverse) { a.g = g_mu; a.ldg = ldgy; }
    a.T2 = (const uint4*)sT2; a.T1 = (const uint4*)sT1; a.T0 = (const uint4*)sT0; a.cs = s_cs; a.act = s_act; a.H1 = sH1; a.H0 = sH0;
    a.A0 = (const uint4*)sA0; a.A1 = (const uint4*)sA1; a.A2 = nullptr;
    a.g_x_add = g_cond; a.ldga = ldgc;
    a.z1 = s_z1; a.z0 = s_z0;
    rc = bwd64_run(a, rec, ws_shift, s_grads[0], s_grads[1], s_grads[2], s_grads[3], s_grads[4], s_grads[5], nullptr, accumulate, st, what, &rp.g[1], &nb1);
    if (rc) return rc;
    rp.blocks0 = (int)nb0;
    hipLaunchKernelGGL(bwd64_reduce2_kernel, dim3(nb0 + nb1), dim3(256), 0, st, rp, accumulate);      /* both networks' partial sums in one launch */
    return bgk_launch_status(what);
}
