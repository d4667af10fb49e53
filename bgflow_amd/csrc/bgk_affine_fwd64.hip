/* bgk_affine_fwd64.hip -- training forward of an affine coupling layer whose conditioner networks have hidden layers of <= 64 units
 * (round 6; BASELINE cfg 2: AffineTransformer(shift = DenseNet([32, 64, 64, 32], ReLU), scale = DenseNet([32, 64, 64, 32], Tanh)),
 * nn/flow/transformer/affine.py:35-70 + nn/dense.py:30-48 + nn/flow/coupling.py:152-182) -- the arithmetic of
 * bgk_coupling_affine_dense_h2 plus what the backward reads: the scaled pre-activations z0, z1 [B, 64] of both networks, the shift values
 * and the scale network's values before tanh [B, ldms].
 *
 * bgk_coupling_affine_dense_h2_train runs such a layer on the width-128 kernel with the hidden layers zero-padded inside the operands:
 * four times the matrix work of the 64 x 64 layers and twice that of the others, 0.57 ms per layer of cfg 2 at 2^20 samples for 1.7 GB
 * of traffic.  This kernel is sized for the small networks: operands packed for 64 hidden rows (bgk_pack_mlp_h2, HT = 2; 39 KB per
 * network) sit in LDS for the life of a persistent workgroup of 12 waves (3 per SIMD, <= 170 registers per lane: a 64-unit layer is 32
 * accumulator registers), a wave owns 32 samples per tile, reads its rows of the conditioner input and of y straight from global memory
 * and writes z through a 2 KB LDS slab as complete 128-byte lines.  Same products in the same order as the other split-f16 kernels
 * (k-steps ascending, lo hi / hi lo / hi hi, bias last; hidden activations on the hardware exp2 / rcp forms): bit-identical outputs.
 * Roofline: HBM, 4 (n_in + 2 d + 4 * 64 + 2 ldms + 2) B per sample = 1.7 KB for cfg 2.
 * Envelope: d <= 32, n_in <= 32 (one tensor, not periodic), hidden layers <= 64; everything else: bgk_coupling_affine_dense_h2_train.
 */
#include "bgk_mfma_h2.h"

namespace {

#ifndef BGK_FWD64_WAVES
#define BGK_FWD64_WAVES 12              /* 3 per SIMD: 170 registers per lane (the kernel needs 169; 16 waves: 40 spilled) */
#endif
constexpr int FWW = BGK_FWD64_WAVES;        /* waves per workgroup (one workgroup per CU) */
constexpr int NETB = 12 + 18 + 9;           /* 1 KiB operand blocks per network: A0 (<= 3 k-steps x 2 tiles x {hi, lo}), A1 (4 x 2 x 2 + 2 bias), A2 (4 x 1 x 2 + 1) */

struct Fwd64Net { const uint4* A0; const uint4* A1; const uint4* A2; const float* cs; int act; float* z0; float* z1; };
struct Fwd64Args {
    const float* x; int64_t ldc; int n_in; int S0;
    Fwd64Net net[2];                        /* shift, scale (A0 == NULL: absent) */
    const float* log_alpha; int preserve_volume, is_circular, inverse;
    const float* y; int64_t ldy; int64_t B; int d;
    float* out; int64_t ldo; float* dlogp; int accumulate;
    float* mu; float* s_raw; int64_t ldms;
    int vec_y, vec_o;                       /* rows of y / out start on 16-byte boundaries (float4 accesses) */
};

__device__ __forceinline__ float f64_rcp_nr(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
/* tanh of the OUTPUT layer (log sigma): the form of the other fused affine kernels (bgk_fused2.hip::aff_tanh_out) */
__device__ __forceinline__ float f64_tanh_out(float x) {
    const float ax = __builtin_fabsf(x);
    const float dn = 1.0f + __builtin_amdgcn_exp2f(ax * 2.88539008177792681f);
    const float big = __builtin_copysignf(__builtin_fmaf(-2.0f, f64_rcp_nr(dn), 1.0f), x);
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    const float small = __builtin_fmaf(p * z, x, x);
    return ax >= 0.625f ? big : small;
}

/* NT tiles held in accumulator layout -> dst[b0 + r][32 m ..] (row pitch `pitch` floats) as complete 128-byte lines, half a tile at a
 * time through the wave's LDS slab [16][32] (16-byte pieces XOR-swizzled by the row); buffer stores: rows past the batch are out of range */
typedef unsigned f64_u32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void f64_store_tiles(const h2_f32x16 (&t)[NT], float* dst, int pitch, float* slab, int64_t b0, int rows, int lane) {
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f64_u32x4, v), rs, ((16 * half + r) * pitch + 32 * m + 4 * p) * 4, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
}

struct F64Frag { h2_h16x8 hi, lo; };

/* one network on the tile whose layer-0 B operands are xb[0 .. S0): result (scaled by 1: the true output values) in o */
__device__ __forceinline__ void f64_network(const Fwd64Net& n, const uint4* ops, const F64Frag (&xb)[3], int S0, h2_f32x16& o, float* slab,
                                            int64_t b0, int rows, int lane) {
    const uint4* A0 = ops;
    const uint4* A1 = ops + 12 * 64;
    const uint4* A2 = ops + 30 * 64;
    const float c0 = n.cs[1], c1 = n.cs[3], c2 = n.cs[5];
    h2_f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < S0) {
            H2A<2> fr;
            h2a_load<2>(fr, A0, s, lane);
            h2_mfma3<2>(acc, fr, xb[s].hi, xb[s].lo);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] *= c0;
    if (n.z0) f64_store_tiles<2>(acc, n.z0, 64, slab, b0, rows, lane);          /* (NULL: nothing is saved -- bgk_affine_coupling_backward64 recomputes) */
    H2B<2> bf;
    h2_act_tile(acc[0], 1.0f, n.act);
    h2_act_tile(acc[1], 1.0f, n.act);
    h2_make_b<2>(bf, acc);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    h2_gemm_hidden<2, 2>(acc, bf, A1, lane);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] *= c1;
    if (n.z1) f64_store_tiles<2>(acc, n.z1, 64, slab, b0, rows, lane);
    h2_act_tile(acc[0], 1.0f, n.act);
    h2_act_tile(acc[1], 1.0f, n.act);
    h2_make_b<2>(bf, acc);
    h2_f32x16 out1[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) out1[0][r] = 0.0f;
    h2_gemm_hidden<1, 2>(out1, bf, A2, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = out1[0][r] * c2;
}

__global__ __launch_bounds__(FWW * 64, 1) void coupling_affine_fwd64_train_kernel(Fwd64Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint4* s_op = reinterpret_cast<uint4*>(smem);                          /* [2][NETB][64] */
    const int lane_in = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* slab = smem + 2 * NETB * 256 + wave * 512;
    const int nb0 = a.S0 * 4;
    for (int n = 0; n < 2; ++n) {
        if (!a.net[n].A0) continue;
        for (int b = wave; b < NETB; b += FWW) {
            const uint4* src = b < 12 ? (b < nb0 ? a.net[n].A0 + b * 64 : nullptr) : (b < 30 ? a.net[n].A1 + (b - 12) * 64 : a.net[n].A2 + (b - 30) * 64);
            if (src) s_op[(n * NETB + b) * 64 + lane_in] = src[lane_in];
        }
    }
    __syncthreads();
    const bool has_shift = a.net[0].A0 != nullptr, has_scale = a.net[1].A0 != nullptr;
    const float alpha = has_scale ? bgk_expf(a.log_alpha[0]) : 0.0f;
    const int64_t n_tiles = (a.B + 31) / 32;
    const int d = a.d, n_in = a.n_in;
    for (int64_t tile = (int64_t)blockIdx.x * FWW + wave; tile < n_tiles; tile += (int64_t)gridDim.x * FWW) {
        int lane = lane_in;
        asm volatile("" : "+v"(lane));           /* per-lane addresses are recomputed per tile, not kept live as loop invariants */
        const int j = lane & 31, hh = lane >> 5;
        const int64_t b0 = tile * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
        const int jr = j < rows ? j : rows - 1;
        /* ---- layer-0 B operands: the lane's row of the conditioner input, features 16 s + 8 hh ..; the constant-1 feature n_in carries the bias ---- */
        F64Frag xb[3];
        const float* xrow = a.x + (b0 + jr) * a.ldc;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = 16 * s + 8 * hh + e;
                v[e] = (s < a.S0 && f < n_in) ? xrow[f] : (f == n_in ? 1.0f : 0.0f);
            }
            h2_split8(v, xb[s].hi, xb[s].lo);
        }
        h2_f32x16 mu, sr;
#pragma unroll
        for (int r = 0; r < 16; ++r) { mu[r] = 0.0f; sr[r] = 0.0f; }
        if (has_shift) {
            f64_network(a.net[0], s_op, xb, a.S0, mu, slab, b0, rows, lane);
            h2_f32x16 t1[1] = {mu};
            if (a.mu) f64_store_tiles<1>(t1, a.mu, (int)a.ldms, slab, b0, rows, lane);
        }
        if (has_scale) {
            f64_network(a.net[1], s_op + NETB * 64, xb, a.S0, sr, slab, b0, rows, lane);
            h2_f32x16 t1[1] = {sr};
            if (a.s_raw) f64_store_tiles<1>(t1, a.s_raw, (int)a.ldms, slab, b0, rows, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        /* y values of the lane's dims (accumulator layout: dims (r & 3) + 8 (r >> 2) + 4 hh of sample j), requested behind the networks (their registers are the networks' while those run) */
        float yv[16];
        const float* yrow = a.y + (b0 + jr) * a.ldy;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 8 * q + 4 * hh;
            if (a.vec_y && f0 + 3 < d) {
                const float4 t = *reinterpret_cast<const float4*>(yrow + f0);
                yv[4 * q] = t.x; yv[4 * q + 1] = t.y; yv[4 * q + 2] = t.z; yv[4 * q + 3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[4 * q + e] = f0 + e < d ? yrow[f0 + e] : 0.0f;
            }
        }
        /* ---- affine tail (affine.py:41-70) ---- */
        float lsum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dim = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float l = (has_scale && dim < d) ? f64_tanh_out(sr[r]) * alpha : 0.0f;
            sr[r] = l;
            lsum += l;
        }
        float total = lsum + __shfl_xor(lsum, 32);
        if (a.preserve_volume && has_scale) {
            const float mean = total / (float)d;
            lsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dim = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float ls = dim < d ? sr[r] - mean : 0.0f;
                sr[r] = ls;
                lsum += ls;
            }
            total = lsum + __shfl_xor(lsum, 32);
        }
        float ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ls = sr[r];
            const float sg = __builtin_amdgcn_exp2f((a.inverse ? -ls : ls) * 1.44269504088896341f);
            float t = a.inverse ? sg * (yv[r] - mu[r]) : sg * yv[r] + mu[r];
            if (a.is_circular) { t = t - __builtin_truncf(t); if (t < 0.0f) t = t + 1.0f; }
            ov[r] = t;
        }
        if (j < rows) {
            float* orow = a.out + (b0 + j) * a.ldo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 8 * q + 4 * hh;
                if (a.vec_o && f0 + 3 < d) {
                    *reinterpret_cast<float4*>(orow + f0) = make_float4(ov[4 * q], ov[4 * q + 1], ov[4 * q + 2], ov[4 * q + 3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (f0 + e < d) orow[f0 + e] = ov[4 * q + e];
                }
            }
            if (hh == 0) {
                const float dl = a.inverse ? -total : total;
                if (a.accumulate) a.dlogp[b0 + j] += dl; else a.dlogp[b0 + j] = dl;
            }
        }
    }
}

}  // namespace

extern "C" int bgk_coupling_affine_dense_fwd64_train(const float* cond, int64_t ldc, int32_t n_in,
                                                     const void* sA0, const void* sA1, const void* sA2, const float* s_cs, int32_t s_act,
                                                     const void* tA0, const void* tA1, const void* tA2, const float* t_cs, int32_t t_act,
                                                     const float* log_alpha, int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                                                     const float* y, int64_t ldy, int64_t B, int32_t d, float* out, int64_t ldo,
                                                     float* dlogp, int32_t accumulate,
                                                     float* s_z0, float* s_z1, float* t_z0, float* t_z1, float* mu, float* s_raw, int64_t ldms,
                                                     void* stream) {
    if (B == 0) return 0;
    const char* what = "bgk_coupling_affine_dense_fwd64_train";
    BGK_CHECK_ARG(cond && y && out && dlogp && B > 0 && d > 0 && n_in > 0 && ldc >= n_in && ldy >= d && ldo >= d, "%s: bad arguments", what);
    const bool no_save = !s_z0 && !s_z1 && !t_z0 && !t_z1 && !mu && !s_raw;       /* all six NULL: the forward alone (the backward recomputes: bgk_affine_coupling_backward64) */
    BGK_CHECK_ARG((sA0 || tA0) && (!sA0 || (sA1 && sA2 && s_cs)) && (!tA0 || (tA1 && tA2 && t_cs && log_alpha)), "%s: null operand", what);
    BGK_CHECK_ARG(no_save || ((!sA0 || (s_z0 && s_z1)) && (!tA0 || (t_z0 && t_z1 && s_raw))), "%s: save buffers: all of a network's (mu may be left out), or none at all", what);
    BGK_CHECK_ARG(no_save || (ldms >= 32 && ldms % 4 == 0 && ldms < (1 << 20)), "%s: ldms = %lld (a multiple of 4, >= 32)", what, (long long)ldms);
    if (d > 32 || n_in > 32) { bgk_set_error("%s: d = %d, n_in = %d: the kernel takes d, n_in <= 32", what, d, n_in); return BGK_EUNSUPPORTED; }
    const auto act_ok = [](const void* A, int act) { return !A || (act >= 1 && act <= 3); };
    BGK_CHECK_ARG(act_ok(sA0, s_act) && act_ok(tA0, t_act), "%s: act: 1 SiLU, 2 ReLU, 3 Tanh", what);
    Fwd64Args a;
    a.x = cond; a.ldc = ldc; a.n_in = n_in; a.S0 = (n_in + 1 + 15) / 16;
    a.net[0] = Fwd64Net{(const uint4*)sA0, (const uint4*)sA1, (const uint4*)sA2, s_cs, s_act, s_z0, s_z1};
    a.net[1] = Fwd64Net{(const uint4*)tA0, (const uint4*)tA1, (const uint4*)tA2, t_cs, t_act, t_z0, t_z1};
    a.log_alpha = log_alpha; a.preserve_volume = preserve_volume; a.is_circular = is_circular; a.inverse = inverse;
    a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.mu = mu; a.s_raw = s_raw; a.ldms = ldms;
    a.vec_y = ((uintptr_t)y & 15) == 0 && (ldy & 3) == 0;
    a.vec_o = ((uintptr_t)out & 15) == 0 && (ldo & 3) == 0;
    const size_t shmem = sizeof(float) * ((size_t)2 * NETB * 256 + (size_t)FWW * 512);
    const int64_t n_tiles = (B + 31) / 32;
    const int64_t want = (n_tiles + FWW - 1) / FWW;
    const int grid = (int)(want < 256 ? want : 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(coupling_affine_fwd64_train_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(coupling_affine_fwd64_train_kernel, dim3(grid), dim3(FWW * 64), shmem, (hipStream_t)stream, a);
    return bgk_launch_status(what);
}
