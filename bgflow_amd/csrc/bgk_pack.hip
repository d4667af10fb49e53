/* bgk_pack.hip -- device-side operand packing for the split-f16 fused kernels (bgk_fused.hip, bgk_mfma_h2.h).
 * Replaces the host/torch packer (bgflow_amd/dense.py::_pack_h2, kept as the layout's reference and tested
 * against this one) where the weights change every step: a KL / NLL training step re-packs 16 conditioners, which
 * as ~30 small torch ops + 3 host syncs per layer cost more than the fused forward itself.
 *   pass 1 (32 workgroups per layer + atomicMax): m = max(|W|, |b|) -> scale 2^e with e = clamp(floor(log2(32768 / m)), -16, 24);
 *           cs[2 l] = 2^e, cs[2 l + 1] = 2^-e (the kernels read the unscale factor from device memory)
 *   pass 2: one thread per (1 KiB block, lane): 8 weights -> hi = rne_f16(v), lo = rne_f16(v - hi)
 */
#include "bgk_common.h"

namespace {

__device__ __forceinline__ uint16_t pk_bf16_rne(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float pk_bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
/* 16-bit pattern of part p (0 hi, 1 lo) of v in the layer's operand type */
__device__ __forceinline__ uint16_t pk_part(float v, int p, int bf16) {
    if (bf16) {
        const uint16_t h = pk_bf16_rne(v);
        return p ? pk_bf16_rne(v - pk_bf16_to_f32(h)) : h;
    }
    const _Float16 h = (_Float16)v;
    const _Float16 r = p ? (_Float16)(v - (float)h) : h;
    return __builtin_bit_cast(uint16_t, r);
}

struct PackLayer {
    const float* W; const float* b;   /* source Linear: weight [rows, K] row-major, bias [rows] */
    int rows, K;
    const int32_t* row_map;           /* packed row -> source row (-1 = padding), NULL = identity */
    int n_groups;                     /* independent 32*NT-row groups (layer-2 chunks), each followed by its bias blocks */
    int NT, S, natural;               /* natural = 1: k = 16 s + 8 kb + e with the bias as column K; 0: accumulator order */
    _Float16* out;
    int bf16;                         /* 1: hi = rne_bf16(v) (no scaling), lo = rne_bf16(v - hi) [lo is read for the bias blocks only] */
};

/* pass 1a: max |value| of each layer, PACK_SPLIT workgroups per layer, combined with atomicMax on the f32 bit pattern
 * (non-negative floats order like unsigned integers) in cs[2 l] (zeroed by the launcher); pass 1b turns the maxima into
 * the scale pair in place.  (One workgroup per layer took 46 us -- 0.85 ms of every training step.) */
constexpr int PACK_SPLIT = 32;
__global__ __launch_bounds__(256) void pack_max_kernel(PackLayer L0, PackLayer L1, PackLayer L2, float* cs) {
    const int layer = blockIdx.x / PACK_SPLIT, part = blockIdx.x % PACK_SPLIT;
    const PackLayer& L = layer == 0 ? L0 : (layer == 1 ? L1 : L2);
    __shared__ float red[4];
    float m = 0.0f;
    const int nW = L.rows * L.K;
    for (int i = part * 256 + threadIdx.x; i < nW; i += PACK_SPLIT * 256) m = fmaxf(m, fabsf(L.W[i]));
    for (int i = part * 256 + threadIdx.x; i < L.rows; i += PACK_SPLIT * 256) m = fmaxf(m, fabsf(L.b[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (!(m < 3.0e38f)) m = 3.4e38f;                      /* inf / NaN weights: largest finite pattern (scale exponent 0 below) */
        atomicMax(reinterpret_cast<unsigned int*>(cs + 2 * layer), __builtin_bit_cast(unsigned int, m));
    }
}
__global__ void pack_scale_kernel(PackLayer L0, PackLayer L1, PackLayer L2, float* cs) {
    const int layer = threadIdx.x;
    if (layer >= 3) return;
    const PackLayer& L = layer == 0 ? L0 : (layer == 1 ? L1 : L2);
    const float m = cs[2 * layer];
    int e = 0;
    if (!L.bf16 && m > 0.0f && m < 3.0e38f) {
        e = (int)floorf(log2f(32768.0f / m));
        e = e < -16 ? -16 : (e > 24 ? 24 : e);
    }
    cs[2 * layer] = ldexpf(1.0f, e);
    cs[2 * layer + 1] = ldexpf(1.0f, -e);
}

/* all three layers in one launch: workgroups [first[q], first[q + 1]) pack layer q */
struct PackGroup { PackLayer L[3]; int first[4]; };

__device__ __forceinline__ void pack_blocks_body(const PackGroup& g, const float* cs, int bx) {
    const int layer = bx >= g.first[2] ? 2 : (bx >= g.first[1] ? 1 : 0);
    const PackLayer& L = g.L[layer];
    const int blocks_per_group = L.S * L.NT * 2 + (L.natural ? 0 : L.NT);
    const int64_t t = (int64_t)(bx - g.first[layer]) * 256 + threadIdx.x;
    const int64_t total = (int64_t)L.n_groups * blocks_per_group * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int blk = (int)((t >> 6) % blocks_per_group);
    const int grp = (int)((t >> 6) / blocks_per_group);
    const int i = lane & 31, kb = lane >> 5;
    const float scale = cs[2 * layer];
    uint16_t o[8];
    if (blk < L.S * L.NT * 2) {
        const int p = blk & 1, m = (blk >> 1) % L.NT, s = (blk >> 1) / L.NT;
        const int prow = grp * 32 * L.NT + 32 * m + i;
        const int srow = L.row_map ? L.row_map[prow] : (prow < L.rows ? prow : -1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = L.natural ? 16 * s + 8 * kb + e : 32 * (s >> 1) + (e & 3) + 8 * (2 * (s & 1) + (e >> 2)) + 4 * kb;
            float v = 0.0f;
            if (srow >= 0) {
                if (k < L.K) v = L.W[(int64_t)srow * L.K + k];
                else if (L.natural && k == L.K) v = L.b[srow];
            }
            o[e] = pk_part(v * scale, p, L.bf16);
        }
    } else {
        const int m = blk - L.S * L.NT * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0;
        if (kb == 0) {
            const int prow = grp * 32 * L.NT + 32 * m + i;
            const int srow = L.row_map ? L.row_map[prow] : (prow < L.rows ? prow : -1);
            if (srow >= 0) {
                const float v = L.b[srow] * scale;
                o[0] = pk_part(v, 0, L.bf16);
                o[1] = pk_part(v, 1, L.bf16);
            }
        }
    }
    _Float16* dst = L.out + ((int64_t)(grp * blocks_per_group + blk) * 64 + lane) * 8;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
}

__global__ __launch_bounds__(256) void pack_blocks_kernel(PackGroup g, const float* cs) { pack_blocks_body(g, cs, (int)blockIdx.x); }

/* ---- the packs of several conditioners in two launches (a training step re-packs every coupling layer after the optimizer
 * step: 16 x (memset + 3 launches) of ~5 us each were 0.4 ms of a 17 ms step) ---- */
constexpr int PACK_MANY = 16;                 /* conditioners per launch: 16 x 216 B of descriptors in the kernel arguments */
struct PackOne { PackGroup g; float* cs; };
struct PackMany { PackOne c[PACK_MANY]; };
typedef const __attribute__((address_space(4))) PackMany* packmany_t;     /* run-time indexed: scalar loads from the argument block */

/* one workgroup per (layer, conditioner): max |W|, |b| -> the scale pair, written directly (no atomics, no memset) */
__global__ __launch_bounds__(1024) void pack_maxscale_many_kernel(PackMany) {
    const packmany_t ka = (packmany_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int layer = blockIdx.x, ci = blockIdx.y;
    const float* W = ka->c[ci].g.L[layer].W;
    const float* b = ka->c[ci].g.L[layer].b;
    const int rows = ka->c[ci].g.L[layer].rows, nW = rows * ka->c[ci].g.L[layer].K, bf16 = ka->c[ci].g.L[layer].bf16;
    float* cs = ka->c[ci].cs;
    __shared__ float red[16];
    float m = 0.0f;
    for (int i = threadIdx.x; i < nW; i += 1024) m = fmaxf(m, fabsf(W[i]));
    for (int i = threadIdx.x; i < rows; i += 1024) m = fmaxf(m, fabsf(b[i]));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
        if (!(m < 3.0e38f)) m = 3.4e38f;
        int e = 0;
        if (!bf16 && m > 0.0f && m < 3.0e38f) {
            e = (int)floorf(log2f(32768.0f / m));
            e = e < -16 ? -16 : (e > 24 ? 24 : e);
        }
        cs[2 * layer] = ldexpf(1.0f, e);
        cs[2 * layer + 1] = ldexpf(1.0f, -e);
    }
}

__global__ __launch_bounds__(256) void pack_blocks_many_kernel(PackMany) {
    const PackMany* km = (const PackMany*)__builtin_amdgcn_kernarg_segment_ptr();      /* run-time indexed: read in place */
    const int ci = blockIdx.y;
    if ((int)blockIdx.x >= km->c[ci].g.first[3]) return;
    pack_blocks_body(km->c[ci].g, km->c[ci].cs, (int)blockIdx.x);
}

int fill_group(PackGroup& g, const PackLayer& L0, const PackLayer& L1, const PackLayer& L2) {
    g.L[0] = L0; g.L[1] = L1; g.L[2] = L2;
    int blocks = 0;
    for (int q = 0; q < 3; ++q) {
        const PackLayer& L = g.L[q];
        const int blocks_per_group = L.S * L.NT * 2 + (L.natural ? 0 : L.NT);
        const int64_t total = (int64_t)L.n_groups * blocks_per_group * 64;
        g.first[q] = blocks;
        blocks += (int)((total + 255) / 256);
    }
    g.first[3] = blocks;
    return blocks;
}

void launch_pack(const PackLayer& L0, const PackLayer& L1, const PackLayer& L2, const float* cs, hipStream_t st) {
    PackGroup g;
    const int blocks = fill_group(g, L0, L1, L2);
    hipLaunchKernelGGL(pack_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, cs);
}

}  // namespace

extern "C" int bgk_pack_dense_h2(const float* W0, const float* b0, int32_t n_in, int32_t H,
                                 const float* W1, const float* b1,
                                 const float* W2, const float* b2, int32_t rows2,
                                 const int32_t* row_map2_dev, int32_t n_groups2, int32_t NT2, int32_t operand_dtype,
                                 void* A0, void* A1, void* A2, float* cs, void* stream) {
    BGK_CHECK_ARG(W0 && b0 && W1 && b1 && W2 && b2 && A0 && A1 && A2 && cs, "bgk_pack_dense_h2: null pointer");
    BGK_CHECK_ARG(n_in > 0 && (H == 32 || H == 64 || H == 96 || H == 128) && rows2 > 0 && n_groups2 > 0 && NT2 > 0 && NT2 <= 4,
                  "bgk_pack_dense_h2: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int HT = H / 32;
    BGK_CHECK_ARG(operand_dtype == 0 || operand_dtype == 1, "bgk_pack_dense_h2: operand_dtype %d (0 = split-f16, 1 = bf16)", operand_dtype);
    PackLayer L0{W0, b0, H, n_in, nullptr, 1, HT, (n_in + 1 + 15) / 16, 1, (_Float16*)A0, operand_dtype};
    PackLayer L1{W1, b1, H, H, nullptr, 1, HT, 2 * HT, 0, (_Float16*)A1, operand_dtype};
    PackLayer L2{W2, b2, rows2, H, row_map2_dev, n_groups2, NT2, 2 * HT, 0, (_Float16*)A2, operand_dtype};
    if (hipMemsetAsync(cs, 0, 6 * sizeof(float), st) != hipSuccess) { bgk_set_error("bgk_pack_dense_h2: memset failed"); return BGK_EINVAL; }
    hipLaunchKernelGGL(pack_max_kernel, dim3(3 * PACK_SPLIT), dim3(256), 0, st, L0, L1, L2, cs);
    hipLaunchKernelGGL(pack_scale_kernel, dim3(1), dim3(64), 0, st, L0, L1, L2, cs);
    launch_pack(L0, L1, L2, cs, st);
    return bgk_launch_status("bgk_pack_dense_h2");
}

/* bgk_pack_dense_h2 for hidden layers of H0 / H1 <= 32 HT units: the operands are packed for a kernel that runs 32 HT hidden rows
 * (rows past H0 / H1 and their bias entries are zero: padded units hold act(0) = 0 and feed zero columns of the next layer) -- no
 * padded copy of the weights exists.  The output layer: rows2 rows in n_groups2 groups of 32 NT2 (row_map2_dev as bgk_pack_dense_h2). */
extern "C" int bgk_pack_mlp_h2(const float* W0, const float* b0, int32_t n_in, int32_t H0,
                               const float* W1, const float* b1, int32_t H1,
                               const float* W2, const float* b2, int32_t rows2,
                               const int32_t* row_map2_dev, int32_t n_groups2, int32_t NT2, int32_t HT,
                               void* A0, void* A1, void* A2, float* cs, void* stream) {
    BGK_CHECK_ARG(W0 && b0 && W1 && b1 && W2 && b2 && A0 && A1 && A2 && cs, "bgk_pack_mlp_h2: null pointer");
    BGK_CHECK_ARG(n_in > 0 && HT >= 1 && HT <= 4 && H0 > 0 && H0 <= 32 * HT && H1 > 0 && H1 <= 32 * HT && rows2 > 0 && n_groups2 > 0
                  && NT2 > 0 && NT2 <= 4, "bgk_pack_mlp_h2: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    PackLayer L0{W0, b0, H0, n_in, nullptr, 1, HT, (n_in + 1 + 15) / 16, 1, (_Float16*)A0, 0};
    PackLayer L1{W1, b1, H1, H0, nullptr, 1, HT, 2 * HT, 0, (_Float16*)A1, 0};
    PackLayer L2{W2, b2, rows2, H1, row_map2_dev, n_groups2, NT2, 2 * HT, 0, (_Float16*)A2, 0};
    if (hipMemsetAsync(cs, 0, 6 * sizeof(float), st) != hipSuccess) { bgk_set_error("bgk_pack_mlp_h2: memset failed"); return BGK_EINVAL; }
    hipLaunchKernelGGL(pack_max_kernel, dim3(3 * PACK_SPLIT), dim3(256), 0, st, L0, L1, L2, cs);
    hipLaunchKernelGGL(pack_scale_kernel, dim3(1), dim3(64), 0, st, L0, L1, L2, cs);
    launch_pack(L0, L1, L2, cs, st);
    return bgk_launch_status("bgk_pack_mlp_h2");
}

/* bgk_pack_mlp_h2 of n conditioners in two launches per 16; H0 / H1 / NT2 / HT may be NULL: 128 / 128 / 4 / 4 for every conditioner
 * (= bgk_pack_dense_h2_many) */
extern "C" int bgk_pack_mlp_h2_many(int32_t n, const float* const* W0, const float* const* b0, const int32_t* n_in, const int32_t* H0,
                                    const float* const* W1, const float* const* b1, const int32_t* H1,
                                    const float* const* W2, const float* const* b2,
                                    const int32_t* rows2, const int32_t* const* row_map2_dev, const int32_t* n_groups2, const int32_t* NT2,
                                    const int32_t* HT,
                                    void* const* A0, void* const* A1, void* const* A2, float* const* cs, void* stream) {
    BGK_CHECK_ARG(n >= 0 && W0 && b0 && n_in && W1 && b1 && W2 && b2 && rows2 && row_map2_dev && n_groups2 && A0 && A1 && A2 && cs,
                  "bgk_pack_mlp_h2_many: null pointer");
    if (n == 0) return 0;       /* nothing to do (and no launch status to ask a GPU-less box for) */
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += PACK_MANY) {
        const int cnt = n - base < PACK_MANY ? n - base : PACK_MANY;
        PackMany M;
        int max_blocks = 0;
        for (int c = 0; c < cnt; ++c) {
            const int i = base + c;
            const int h0 = H0 ? H0[i] : 128, h1 = H1 ? H1[i] : 128, nt2 = NT2 ? NT2[i] : 4, ht = HT ? HT[i] : 4;
            BGK_CHECK_ARG(W0[i] && b0[i] && W1[i] && b1[i] && W2[i] && b2[i] && A0[i] && A1[i] && A2[i] && cs[i] && n_in[i] > 0 && rows2[i] > 0
                          && n_groups2[i] > 0 && ht >= 1 && ht <= 4 && h0 > 0 && h0 <= 32 * ht && h1 > 0 && h1 <= 32 * ht && nt2 >= 1 && nt2 <= 4,
                          "bgk_pack_mlp_h2_many: bad conditioner %d", i);
            PackLayer L0{W0[i], b0[i], h0, n_in[i], nullptr, 1, ht, (n_in[i] + 1 + 15) / 16, 1, (_Float16*)A0[i], 0};
            PackLayer L1{W1[i], b1[i], h1, h0, nullptr, 1, ht, 2 * ht, 0, (_Float16*)A1[i], 0};
            PackLayer L2{W2[i], b2[i], rows2[i], h1, row_map2_dev[i], n_groups2[i], nt2, 2 * ht, 0, (_Float16*)A2[i], 0};
            const int blocks = fill_group(M.c[c].g, L0, L1, L2);
            M.c[c].cs = cs[i];
            max_blocks = blocks > max_blocks ? blocks : max_blocks;
        }
        for (int c = cnt; c < PACK_MANY; ++c) M.c[c] = M.c[0];
        hipLaunchKernelGGL(pack_maxscale_many_kernel, dim3(3, (unsigned)cnt), dim3(1024), 0, st, M);
        hipLaunchKernelGGL(pack_blocks_many_kernel, dim3((unsigned)max_blocks, (unsigned)cnt), dim3(256), 0, st, M);
    }
    return bgk_launch_status("bgk_pack_mlp_h2_many");
}

extern "C" int bgk_pack_dense_h2_many(int32_t n, const float* const* W0, const float* const* b0, const int32_t* n_in,
                                      const float* const* W1, const float* const* b1, const float* const* W2, const float* const* b2,
                                      const int32_t* rows2, const int32_t* const* row_map2_dev, const int32_t* n_groups2,
                                      void* const* A0, void* const* A1, void* const* A2, float* const* cs, void* stream) {
    return bgk_pack_mlp_h2_many(n, W0, b0, n_in, nullptr, W1, b1, nullptr, W2, b2, rows2, row_map2_dev, n_groups2, nullptr, nullptr, A0, A1, A2, cs, stream);
}
