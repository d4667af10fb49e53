/* bgk_philox.hip -- prior sampling in one launch: the tensors of a ProductDistribution / NormalDistribution /
 * UniformDistribution sample (bgflow/distribution/normal.py:74-92, distributions.py:100-117, product.py:84-117: `torch.randn` /
 * `Uniform.sample` per component + the shift / scale ops) AND the prior energy of the sample (normal.py:61-72; the uniform
 * components contribute their constant) from a counter-based generator.  An OPT-IN path (`sample_fused=True`): the default priors
 * keep drawing from torch's generator, so that `torch.manual_seed` reproduces the sample stream users know.
 *
 * Generator: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 known-answer vectors
 * are pinned in tests/test_oracle_golden.py through oracle/philox.py).  One 128-bit counter = (row low, row high, field << 20 |
 * 4-column block, call offset), key = seed: the stream is a pure function of (seed, offset, GLOBAL row, field, column) -- independent
 * of the launch geometry and of how a batch is sharded over ranks (`row0` = first global row of this launch).
 *   uniform: u = ((x >> 8) + 0.5) 2^-24 in (0, 1), then low + u (high - low)
 *   normal:  Box-Muller on two such uniforms: r = sqrt(-2 ln u1), (r cos 2 pi u2, r sin 2 pi u2); value = mean + scale * n
 *            (scale = sigma sqrt(T)); ln / sincos are the deterministic forms of bgk_detmath.h (same bits as the C oracle)
 * Layout: lane = row of a 64-row tile; a field's [64][d] tile is assembled in LDS (the tile's memory image when rows are contiguous)
 * and leaves as coalesced stores; the energy 0.5 sum n^2 accumulates per lane in column order (deterministic). */
#include "bgk_common.h"

namespace {

constexpr int PW = 4;                 /* waves per workgroup */
constexpr int PH_MAXF = BGK_MAX_ENERGY_FIELDS;

struct PField { float* out; int64_t ldo; const float* p0; const float* p1; int d, kind; float scale, e_const; };
struct PArgs {
    PField f[PH_MAXF]; int n; int64_t B, row0;
    uint32_t seed_lo, seed_hi, offset;
    float c_out; float* energy;       /* energy [B] or NULL: sum_f e_f + c_out (the caller folds 1 / T and log Z into scale / constants) */
    int lds_per_wave;
};
typedef const __attribute__((address_space(4))) PArgs* pargs_t;

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f; }   /* 2^-24 */

__global__ __launch_bounds__(PW * 64) void philox_fields_kernel(PArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const pargs_t ka = (pargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * PW + wave;
    if (tile >= ((a.B + 63) >> 6)) return;
    float* s_t = smem + (size_t)wave * a.lds_per_wave;
    const int64_t b0 = tile * 64;
    const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    const uint64_t grow = (uint64_t)(a.row0 + b0 + lane);
    const uint32_t r_lo = (uint32_t)grow, r_hi = (uint32_t)(grow >> 32);
    float e = a.c_out;
    for (int fi = 0; fi < a.n; ++fi) {
        const int d = ka->f[fi].d, kind = ka->f[fi].kind;
        const float* __restrict__ p0 = ka->f[fi].p0;
        const float* __restrict__ p1 = ka->f[fi].p1;
        const float scale = ka->f[fi].scale;
        float* row = s_t + lane * d;
        float ef = 0.0f;
        for (int cb = 0; 4 * cb < d; ++cb) {
            uint32_t o[4];
            philox4x32_10(r_lo, r_hi, ((uint32_t)fi << 20) | (uint32_t)cb, a.offset, a.seed_lo, a.seed_hi, o);
            float v[4];
            if (kind == 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float rad = __builtin_sqrtf(-2.0f * bgk_logf(u01(o[2 * h])));
                    float sn, cs;
                    bgk_sincos2pif(u01(o[2 * h + 1]), &sn, &cs);
                    v[2 * h] = rad * cs; v[2 * h + 1] = rad * sn;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = u01(o[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 4 * cb + q;
                if (col < d) {
                    float y;
                    if (kind == 1) { ef = __builtin_fmaf(v[q], v[q], ef); y = __builtin_fmaf(v[q], scale, p0 ? p0[col] : 0.0f); }
                    else { const float lo = p0 ? p0[col] : 0.0f, hi = p1 ? p1[col] : 1.0f; y = lo + v[q] * (hi - lo); }
                    row[col] = y;
                }
            }
        }
        e += kind == 1 ? 0.5f * ef + ka->f[fi].e_const : ka->f[fi].e_const;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        /* tile out: the LDS tile is the memory image of contiguous rows; any row stride otherwise */
        float* out = ka->f[fi].out;
        const int64_t ldo = ka->f[fi].ldo;
        const int total = rows * d;
        if (ldo == d) {
            float* dst = out + b0 * d;
            for (int q = lane; q < total; q += 64) dst[q] = s_t[q];
        } else {
            for (int q = lane; q < total; q += 64) { const int r = q / d, c = q - r * d; out[(b0 + r) * ldo + c] = s_t[q]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (a.energy && lane < rows) a.energy[b0 + lane] = e;
}

}  // namespace

extern "C" int bgk_philox_fields(uint64_t seed, uint32_t offset, int64_t row0, int32_t n_fields, float* const* out, const int64_t* ldo,
                                 const int32_t* d, const int32_t* kind, const float* const* p0, const float* const* p1,
                                 const float* scale, const float* e_const, double c_out, int64_t B, float* energy, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(n_fields >= 1 && n_fields <= PH_MAXF && out && ldo && d && kind && B >= 0 && row0 >= 0, "bgk_philox_fields: bad arguments");
    if (B == 0) return 0;
    PArgs a{};
    int dmax = 1;
    for (int i = 0; i < n_fields; ++i) {
        BGK_CHECK_ARG(out[i] && d[i] > 0 && d[i] < (1 << 22) && ldo[i] >= d[i] && (kind[i] == 0 || kind[i] == 1), "bgk_philox_fields: field %d", i);
        a.f[i] = PField{out[i], ldo[i], p0 ? p0[i] : nullptr, p1 ? p1[i] : nullptr, d[i], kind[i], scale ? scale[i] : 1.0f, e_const ? e_const[i] : 0.0f};
        dmax = d[i] > dmax ? d[i] : dmax;
    }
    a.n = n_fields; a.B = B; a.row0 = row0; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.offset = offset;
    a.c_out = (float)c_out; a.energy = energy; a.lds_per_wave = 64 * dmax;
    const size_t shmem = sizeof(float) * (size_t)PW * a.lds_per_wave;
    BGK_CHECK_ARG(shmem <= 160 * 1024, "bgk_philox_fields: a field of %d columns does not fit the LDS tile", dmax);
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(philox_fields_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int64_t n_wg = (((B + 63) >> 6) + PW - 1) / PW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_philox_fields: batch too large for one launch");
    hipLaunchKernelGGL(philox_fields_kernel, dim3((unsigned)n_wg), dim3(PW * 64), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_philox_fields");
}
