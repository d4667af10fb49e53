/* bgk_ic.hip -- Z-matrix <-> Cartesian internal-coordinate transform (relative to fixed atoms,
 * optional PCA whitening of the fixed block) with log|det J|.
 *
 * One lane owns one sample and walks the Z-matrix rows (xyz->IC) or the placement table
 * (IC->xyz, NeRF-style sequential placement); the index tables are wave-uniform (scalar loads).
 * Per-sample rows are staged through LDS so every HBM access is a coalesced stream over the
 * tile: algorithmic bytes 4*(3*n_atoms + 3n + keep + 2) per sample.  Row strides in LDS are odd
 * -> the lane-per-row accesses are bank-conflict-free.
 *
 * Arithmetic follows SURVEY.md Appendix B / oracle/bgo_impl.h (explicit 3x3 Jacobian determinant
 * like the reference, eps clamps included); transcendental functions are the precise OCML ones.
 */
#include "bgk_common.h"
#include "bgk_dual.h"
#include <stdlib.h>

namespace {

constexpr int IC_THREADS = 128;
#define PI_F 3.14159265358979323846f

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float norm(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 divs(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float det3(V3 r0, V3 r1, V3 r2) { return dot(cross(r0, r1), r2); }

struct IcArgs {
    float* x; int64_t ldx;                  /* in (xyz2ic) or out (ic2xyz) */
    float* bonds; float* angles; float* torsions; int64_t ldic;
    float* xfix; int64_t ldf;
    const int32_t* table;                   /* zmat [n,4] or place [n,5] */
    const int32_t* fixed;
    int n, n_fixed, n_atoms, keep;
    int normalize, enforce;
    float eps;
    const float* wh_mean; const float* T;   /* Twhiten [3nf, keep] or Tblacken [keep, 3nf] */
    float jac_xz;
    int64_t B;
    float* dlogp; int accumulate;
    int32_t* warn_count;
    int sx;                                 /* LDS row stride of the xyz tile (odd) */
    int sic;                                /* LDS row stride of one IC tile (odd)  */
    int sfx;                                /* LDS row stride of the xfix tile (odd) */
};

__device__ __forceinline__ float clamp_min_flag(float v, float eps, int enforce, int& warn) {
    if (v < eps) { warn += 1; if (enforce) v = eps; }
    return v;
}

/* cooperative coalesced copy of a [rows, cols] global tile (row stride ld) <-> LDS (row stride s) */
__device__ __forceinline__ void tile_load(float* dst, int s, const float* src, int64_t ld, int rows, int cols) {
    for (int i = threadIdx.x; i < rows * cols; i += (int)blockDim.x) {
        int r = i / cols, c = i - r * cols;
        dst[r * s + c] = src[(int64_t)r * ld + c];
    }
}
__device__ __forceinline__ void tile_store(float* dst, int64_t ld, const float* src, int s, int rows, int cols) {
    for (int i = threadIdx.x; i < rows * cols; i += (int)blockDim.x) {
        int r = i / cols, c = i - r * cols;
        dst[(int64_t)r * ld + c] = src[r * s + c];
    }
}

__global__ __launch_bounds__(IC_THREADS) void ic_xyz2ic_kernel(IcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = (int)blockDim.x, n = a.n, nf3 = 3 * a.n_fixed;
    float* s_x = smem;                       /* [TS][sx]  */
    float* s_b = s_x + TS * a.sx;            /* [TS][sic] */
    float* s_a = s_b + TS * a.sic;
    float* s_t = s_a + TS * a.sic;
    float* s_f = s_t + TS * a.sic;           /* [TS][sfx] */
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    int warn = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        tile_load(s_x, a.sx, a.x + b0 * a.ldx, a.ldx, rows, 3 * a.n_atoms);
        __syncthreads();
        if (tid < rows) {
            const float* xr = s_x + tid * a.sx;
            float acc = 0.0f;
            for (int i = 0; i < n; ++i) {
                const int i1 = a.table[4 * i], i2 = a.table[4 * i + 1], i3 = a.table[4 * i + 2], i4 = a.table[4 * i + 3];
                V3 x1 = ld3(xr + 3 * i1), x2 = ld3(xr + 3 * i2), x3 = ld3(xr + 3 * i3), x4 = ld3(xr + 3 * i4);
                /* dist_deriv (ic_helper.py:148-165) */
                V3 r = sub(x2, x1);
                float rn = clamp_min_flag(norm(r), a.eps, a.enforce, warn);
                V3 Jb = {-r.x / rn, -r.y / rn, -r.z / rn};
                /* angle_deriv (ic_helper.py:168-210) */
                V3 r12 = sub(x1, x2);
                float n12 = clamp_min_flag(norm(r12), a.eps, a.enforce, warn);
                V3 u12 = divs(r12, n12);
                V3 r32 = sub(x3, x2);
                float n32 = clamp_min_flag(norm(r32), a.eps, a.enforce, warn);
                V3 u32 = divs(r32, n32);
                float cosa = dot(u12, u32);
                /* J = u32^T (I - u12 u12^T) / n12 : column c = sum_k u32[k] * (delta_kc - u12[k] u12[c]) / n12 */
                float u12v[3] = {u12.x, u12.y, u12.z}, u32v[3] = {u32.x, u32.y, u32.z}, Jav[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float s = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float Pkc = ((k == c ? 1.0f : 0.0f) - u12v[k] * u12v[c]) / n12;
                        s += u32v[k] * Pkc;
                    }
                    Jav[c] = s;
                }
                if (a.enforce) { cosa = cosa < -1.0f + a.eps ? -1.0f + a.eps : cosa; cosa = cosa > 1.0f - a.eps ? 1.0f - a.eps : cosa; }
                float ang = acosf(cosa);
                float sq = __builtin_sqrtf(1.0f - cosa * cosa);
                V3 Ja = {-Jav[0] / sq, -Jav[1] / sq, -Jav[2] / sq};
                /* torsion_deriv (ic_helper.py:213-293) */
                V3 b0v = {-(x2.x - x1.x), -(x2.y - x1.y), -(x2.z - x1.z)};
                V3 b1 = sub(x3, x2), b2 = sub(x4, x3);
                float b1n = clamp_min_flag(norm(b1), a.eps, a.enforce, warn);
                V3 u = divs(b1, b1n);
                float b0u = dot(b0v, u), b2u = dot(b2, u);
                V3 v = {b0v.x - b0u * u.x, b0v.y - b0u * u.y, b0v.z - b0u * u.z};
                V3 w = {b2.x - b2u * u.x, b2.y - b2u * u.y, b2.z - b2u * u.z};
                float xx = dot(v, w);
                float yy = dot(cross(u, v), w);
                float tor = atan2f(yy, xx);
                float q = clamp_min_flag(xx * xx + yy * yy, a.eps, a.enforce, warn);
                float dadx = -yy / q, dady = xx / q;
                V3 wxu = cross(w, u);
                V3 g = {dadx * w.x + dady * wxu.x, dadx * w.y + dady * wxu.y, dadx * w.z + dady * wxu.z};
                float gu = dot(g, u);
                V3 Jt = {g.x - gu * u.x, g.y - gu * u.y, g.z - gu * u.z};
                float det = det3(Jb, Ja, Jt);
                acc += logf(fabsf(det));
                if (a.normalize) { ang = ang / PI_F; tor = (tor + PI_F) / (2.0f * PI_F); }
                s_b[tid * a.sic + i] = rn;
                s_a[tid * a.sic + i] = ang;
                s_t[tid * a.sic + i] = tor;
            }
            if (a.normalize) acc += -(float)n * logf(PI_F) - (float)n * logf(2.0f * PI_F);
            if (a.T) {
                for (int k = 0; k < a.keep; ++k) {
                    float s = 0.0f;
                    for (int c = 0; c < nf3; ++c) {
                        float xc = xr[3 * a.fixed[c / 3] + c % 3] - a.wh_mean[c];
                        s += xc * a.T[c * a.keep + k];
                    }
                    s_f[tid * a.sfx + k] = s;
                }
                acc += a.jac_xz;
            } else {
                for (int c = 0; c < nf3; ++c) s_f[tid * a.sfx + c] = xr[3 * a.fixed[c / 3] + c % 3];
            }
            if (a.accumulate) a.dlogp[b0 + tid] += acc; else a.dlogp[b0 + tid] = acc;
        }
        __syncthreads();
        tile_store(a.bonds + b0 * a.ldic, a.ldic, s_b, a.sic, rows, n);
        tile_store(a.angles + b0 * a.ldic, a.ldic, s_a, a.sic, rows, n);
        tile_store(a.torsions + b0 * a.ldic, a.ldic, s_t, a.sic, rows, n);
        tile_store(a.xfix + b0 * a.ldf, a.ldf, s_f, a.sfx, rows, a.keep);
        __syncthreads();
    }
    if (warn && a.warn_count) atomicAdd(a.warn_count, warn);
}

/* IC -> xyz.  Only the growing position table (dynamic atom indexing) lives in LDS (one odd-stride row per
 * lane, 8 waves / CU); bonds / angles / torsions are read straight from global memory -- a wave walks a
 * contiguous [64, n] block column by column, so every line is fetched once and served from L1 afterwards.
 * With normalised angles the trigonometry uses the exact-quadrant sincos(2 pi x) of bgk_detmath.h on the
 * NORMALISED value (sin(pi a) = sin(2 pi a/2), sin(2 pi t - pi) = -sin(2 pi t)) instead of OCML sinf/cosf
 * with their general argument reduction. */
constexpr int IC2_THREADS = 64;
__global__ __launch_bounds__(IC2_THREADS) void ic_ic2xyz_kernel(IcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = (int)blockDim.x, n = a.n, nf3 = 3 * a.n_fixed;
    float* s_x = smem;                  /* [TS][sx] */
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    int warn = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        if (tid < rows) {
            const int64_t b = b0 + tid;
            float* xr = s_x + tid * a.sx;
            const float* fx = a.xfix + b * a.ldf;
            float acc = 0.0f;
            if (a.T) {
                for (int c = 0; c < nf3; ++c) {
                    float s = 0.0f;
                    for (int k = 0; k < a.keep; ++k) s += fx[k] * a.T[k * nf3 + c];
                    xr[3 * a.fixed[c / 3] + c % 3] = s + a.wh_mean[c];
                }
                acc += -a.jac_xz;
            } else {
                for (int c = 0; c < nf3; ++c) xr[3 * a.fixed[c / 3] + c % 3] = fx[c];
            }
            if (a.normalize) acc += (float)n * logf(PI_F) + (float)n * logf(2.0f * PI_F);
            const float* pb = a.bonds + b * a.ldic;
            const float* pa = a.angles + b * a.ldic;
            const float* pt = a.torsions + b * a.ldic;
            for (int i = 0; i < n; ++i) {
                const int at = a.table[5 * i], i1 = a.table[5 * i + 1], i2 = a.table[5 * i + 2],
                          i3 = a.table[5 * i + 3], zr = a.table[5 * i + 4];
                V3 p1 = ld3(xr + 3 * i1), p2 = ld3(xr + 3 * i2), p3 = ld3(xr + 3 * i3);
                const float dd = pb[zr];
                float st, ct, sa, ca;
                if (a.normalize) {
                    bgk_sincos2pif(0.5f * pa[zr], &sa, &ca);
                    bgk_sincos2pif(pt[zr], &st, &ct);
                    st = -st; ct = -ct;
                } else {
                    const float an = pa[zr], t = pt[zr];
                    st = sinf(t); ct = cosf(t); sa = sinf(an); ca = cosf(an);
                }
                /* ic2xyz_deriv (ic_helper.py:372-452) */
                V3 v1 = sub(p1, p2), v2 = sub(p1, p3);
                V3 nv = cross(v1, v2), nn = cross(v1, nv);
                float nvn = clamp_min_flag(norm(nv), a.eps, a.enforce, warn);
                float nnn = clamp_min_flag(norm(nn), a.eps, a.enforce, warn);
                V3 nh = divs(nv, nvn), nnh = divs(nn, nnn);
                V3 v3 = {nh.x * (-st) + nnh.x * ct, nh.y * (-st) + nnh.y * ct, nh.z * (-st) + nnh.z * ct};
                float v3n = clamp_min_flag(norm(v3), a.eps, a.enforce, warn);
                V3 v3h = divs(v3, v3n);
                float v1n = clamp_min_flag(norm(v1), a.eps, a.enforce, warn);
                V3 v1h = divs(v1, v1n);
                V3 pos = {p1.x + v3h.x * dd * sa - v1h.x * dd * ca, p1.y + v3h.y * dd * sa - v1h.y * dd * ca,
                          p1.z + v3h.z * dd * sa - v1h.z * dd * ca};
                V3 Jd = {v3h.x * sa - v1h.x * ca, v3h.y * sa - v1h.y * ca, v3h.z * sa - v1h.z * ca};
                V3 Ja = {v3h.x * dd * ca + v1h.x * dd * sa, v3h.y * dd * ca + v1h.y * dd * sa,
                         v3h.z * dd * ca + v1h.z * dd * sa};
                V3 Jt3 = {nh.x * (-ct) + nnh.x * (-st), nh.y * (-ct) + nnh.y * (-st), nh.z * (-ct) + nnh.z * (-st)};
                float jt1 = dd * sa, h3 = dot(v3h, Jt3), inv = 1.0f / v3n;
                V3 Jt = {jt1 * inv * (Jt3.x - v3h.x * h3), jt1 * inv * (Jt3.y - v3h.y * h3), jt1 * inv * (Jt3.z - v3h.z * h3)};
                /* rows of J = stack([Jd, Ja, Jt], dim=-1) */
                V3 R0 = {Jd.x, Ja.x, Jt.x}, R1 = {Jd.y, Ja.y, Jt.y}, R2 = {Jd.z, Ja.z, Jt.z};
                float det = det3(R0, R1, R2);
                acc += bgk_logf(fabsf(det));
                xr[3 * at] = pos.x; xr[3 * at + 1] = pos.y; xr[3 * at + 2] = pos.z;
            }
            if (a.accumulate) a.dlogp[b] += acc; else a.dlogp[b] = acc;
        }
        __syncthreads();
        for (int i = tid; i < rows * 3 * a.n_atoms; i += (int)blockDim.x) {
            int r = i / (3 * a.n_atoms), c = i - r * 3 * a.n_atoms;
            a.x[(b0 + r) * a.ldx + c] = s_x[r * a.sx + c];
        }
        __syncthreads();
    }
    if (warn && a.warn_count) atomicAdd(a.warn_count, warn);
}

/* ---- generation path: icdf domain maps fused into the IC -> xyz prologue -----------------------------------------------
 * bgk_icdf_ic2xyz: the four CDFTransform blocks the builder puts in front of the coordinate transform
 * (generator_builder.py:443-459, nn/flow/cdf.py:36-45, marginals icmarginals.py:41-77) applied on the fly to the values the
 * placement loop reads anyway -- no [B, 60] round trip, no 4 extra launches.  Channel descriptors are
 * wave-uniform (scalar loads, no divergence).  Arithmetic of this variant: reciprocal square roots for the normalisations
 * and the closed-form log|det J| = 2 ln d + ln|sin a| of a placement (what the explicit 3x3 determinant of ic_helper.py:372-452
 * evaluates to; exact away from the eps clamps, and closer to the f64 value than the f32 determinant). */
#define SQRT2_F 1.41421356237309504880f
#define LOG_SQRT_2PI_F 0.91893853320467274178f

struct IcGenArgs {
    IcArgs ic;
    const float* dsc_b; const float* dsc_a; const float* dsc_t; const float* dsc_f;   /* [n][6] x 3, [keep][6]; NULL = identity */
    int use_eps; float cdf_eps;
};

/* y = icdf(u) of one channel, ld += -log_prob(y) (cdf_kernel's inverse branch, bgk_cdf.hip; ds[5] = the channel's
 * element-independent log-normalisation, precomputed in f64 on the host: no logf per element) */
__device__ __forceinline__ float icdf_channel(float v, const float* ds, int use_eps, float eps, float& ld_acc) {
    const int kind = (int)ds[0];
    if (kind < 0) return v;               /* no map on this field */
    if (use_eps) v = v < eps ? eps : (v > 1.0f - eps ? 1.0f - eps : v);
    float y, ld;
    if (kind == 0) {
        y = ds[1] + v * (ds[2] - ds[1]);
        ld = ds[5];
    } else {
        const float r0 = kind == 1 ? v : ds[4] * v + ds[3];
        const float z = erfinvf(2.0f * r0 - 1.0f) * SQRT2_F;
        y = z * ds[2] + ds[1];
        ld = 0.5f * z * z + ds[5];
    }
    if (use_eps) ld = ld < -1.0f / eps ? -1.0f / eps : ld;
    ld_acc += ld;
    return y;
}

/* 1 / max(|v|, eps) with the clamp counted like clamp_min_flag */
__device__ __forceinline__ float inv_norm_flag(V3 v, float eps, int enforce, int& warn) {
    float n2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (n2 < eps * eps) { warn += 1; if (enforce) n2 = eps * eps; }
    float r = __builtin_amdgcn_rsqf(n2);
    return r * (1.5f - 0.5f * n2 * r * r);       /* one Newton step: ~1 ulp */
}

/* log|det J| of one placement exactly as the reference evaluates it (explicit 3x3 determinant of ic2xyz_deriv,
 * ic_helper.py:372-452, with its eps clamps) -- only taken when a norm of the placement had to be clamped: there the
 * clamped vectors are no longer unit vectors and the determinant differs from d^2 sin a. */
__device__ __noinline__ float explicit_placement_logdet(V3 p1, V3 p2, V3 p3, float dd, float st, float ct, float sa, float ca,
                                                        float eps, int enforce) {
    int w = 0;
    V3 v1 = sub(p1, p2), v2 = sub(p1, p3);
    V3 nv = cross(v1, v2), nn = cross(v1, nv);
    float nvn = clamp_min_flag(norm(nv), eps, enforce, w);
    float nnn = clamp_min_flag(norm(nn), eps, enforce, w);
    V3 nh = divs(nv, nvn), nnh = divs(nn, nnn);
    V3 v3 = {nh.x * (-st) + nnh.x * ct, nh.y * (-st) + nnh.y * ct, nh.z * (-st) + nnh.z * ct};
    float v3n = clamp_min_flag(norm(v3), eps, enforce, w);
    V3 v3h = divs(v3, v3n);
    float v1n = clamp_min_flag(norm(v1), eps, enforce, w);
    V3 v1h = divs(v1, v1n);
    V3 Jd = {v3h.x * sa - v1h.x * ca, v3h.y * sa - v1h.y * ca, v3h.z * sa - v1h.z * ca};
    V3 Ja = {v3h.x * dd * ca + v1h.x * dd * sa, v3h.y * dd * ca + v1h.y * dd * sa, v3h.z * dd * ca + v1h.z * dd * sa};
    V3 Jt3 = {nh.x * (-ct) + nnh.x * (-st), nh.y * (-ct) + nnh.y * (-st), nh.z * (-ct) + nnh.z * (-st)};
    float jt1 = dd * sa, h3 = dot(v3h, Jt3), inv = 1.0f / v3n;
    V3 Jt = {jt1 * inv * (Jt3.x - v3h.x * h3), jt1 * inv * (Jt3.y - v3h.y * h3), jt1 * inv * (Jt3.z - v3h.z * h3)};
    V3 R0 = {Jd.x, Ja.x, Jt.x}, R1 = {Jd.y, Ja.y, Jt.y}, R2 = {Jd.z, Ja.z, Jt.z};
    return bgk_logf(fabsf(det3(R0, R1, R2)));
}

__global__ __launch_bounds__(IC2_THREADS) void icdf_ic2xyz_kernel(IcGenArgs g) {
    const IcArgs& a = g.ic;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = (int)blockDim.x, n = a.n, nf3 = 3 * a.n_fixed, keep = a.keep;
    /* LDS: position table [TS][sx] | Tblacken [keep][nf3] | mean [nf3] | LDS offsets of the fixed coordinates [nf3] | channel
     * descriptors [3 n + keep][6].  The small wave-uniform tables are staged once per workgroup: read from global memory
     * inside the per-sample loops they were vector loads with a full round trip each (58 % of the wave time in s_waitcnt). */
    float* s_x = smem;
    float* s_T = s_x + TS * a.sx;
    float* s_mean = s_T + keep * nf3;
    int* s_off = reinterpret_cast<int*>(s_mean + nf3);
    float* s_dsc = reinterpret_cast<float*>(s_off + nf3);
    /* s_cb[c]: LDS offset (3 * atom) of the atom that Z-matrix row c places.  The tile's inputs are fetched with row-contiguous
     * (coalesced) loads and parked IN the position table: the (bond, angle, torsion) of a placement sit in the three slots its
     * atom's coordinates will overwrite, the fixed block in the first slots of the fixed atoms -- no extra LDS, and none of the
     * 4-byte-per-row strided loads whose lines were fetched again and again by placements far apart in time. */
    int* s_cb = reinterpret_cast<int*>(s_dsc + (3 * n + keep) * 6);
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += TS) s_cb[a.table[5 * i + 4]] = 3 * a.table[5 * i];
    for (int i = tid; i < nf3; i += TS) {
        s_off[i] = 3 * a.fixed[i / 3] + i % 3;
        s_mean[i] = a.T ? a.wh_mean[i] : 0.0f;
    }
    if (a.T) for (int i = tid; i < keep * nf3; i += TS) s_T[i] = a.T[i];
    for (int i = tid; i < (3 * n + keep) * 6; i += TS) {
        const int ch = i / 6, e = i - 6 * ch;
        const float* src = ch < n ? g.dsc_b : (ch < 2 * n ? g.dsc_a : (ch < 3 * n ? g.dsc_t : g.dsc_f));
        const int local = ch < n ? ch : (ch < 2 * n ? ch - n : (ch < 3 * n ? ch - 2 * n : ch - 3 * n));
        s_dsc[i] = src ? src[6 * local + e] : (e == 0 ? -1.0f : 0.0f);      /* kind -1 = identity */
    }
    __syncthreads();
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    /* e / n and e / keep for e < 2^16 by multiply-high: floor(e m / 2^32) with m = ceil(2^32 / n) is exact while e n < 2^32 */
    const unsigned magic_n = (unsigned)((0x100000000ull + (unsigned)n - 1) / (unsigned)n);
    const unsigned magic_k = (unsigned)((0x100000000ull + (unsigned)keep - 1) / (unsigned)keep);
    const int na3 = 3 * a.n_atoms;
    const unsigned magic_a = (unsigned)((0x100000000ull + (unsigned)na3 - 1) / (unsigned)na3);
    const int ldic32 = (int)a.ldic, ldf32 = (int)a.ldf, ldx32 = (int)a.ldx;
    int warn = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        const bool park_fixed = keep <= 16 && keep <= nf3;
        {   /* element e = it * TS + tid of a field's [rows x n] block: 8 loads in flight, then their 8 LDS writes */
            const float* srcs[3] = {a.bonds, a.angles, a.torsions};
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const float* src = srcs[f] + b0 * a.ldic;
                for (int it0 = 0; it0 * TS < rows * n; it0 += 8) {
                    float v[8];
                    int off[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = (it0 + u) * TS + tid, r = (int)__umulhi((unsigned)e, magic_n), c = e - (int)__umul24((unsigned)r, (unsigned)n);
                        const bool ok = e < rows * n;
                        v[u] = ok ? src[(int)__umul24((unsigned)r, (unsigned)ldic32) + c] : 0.0f;      /* 24-bit multiplies: full rate (r < 64; strides checked by the launcher) */
                        off[u] = ok ? (int)__umul24((unsigned)r, (unsigned)a.sx) + s_cb[c] + f : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (off[u] >= 0) s_x[off[u]] = v[u];
                }
            }
            if (park_fixed) {
                const float* src = a.xfix + b0 * a.ldf;
                float v[16];
                int off[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int e = u * TS + tid, r = (int)__umulhi((unsigned)e, magic_k), c = e - (int)__umul24((unsigned)r, (unsigned)keep);
                    const bool ok = e < rows * keep;
                    v[u] = ok ? src[(int)__umul24((unsigned)r, (unsigned)ldf32) + c] : 0.0f;
                    off[u] = ok ? (int)__umul24((unsigned)r, (unsigned)a.sx) + s_off[c] : -1;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) if (off[u] >= 0) s_x[off[u]] = v[u];
            }
        }
        __syncthreads();
        if (tid < rows) {
            const int64_t b = b0 + tid;
            float* xr = s_x + tid * a.sx;
            const float* fx = a.xfix + b * a.ldf;
            float fxv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) fxv[k] = (park_fixed && k < keep) ? xr[s_off[k]] : 0.0f;      /* all read before any is overwritten */
            float acc = 0.0f;
            if (a.T) {
                if (park_fixed) {
                    /* the whitened coordinates first (registers), then each output coordinate as one running sum in k order -- the
                     * same additions as the read-modify-write form below, without 9 dependent LDS round trips per coordinate */
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < keep) fxv[k] = icdf_channel(fxv[k], s_dsc + 6 * (3 * n + k), g.use_eps, g.cdf_eps, acc);
                    for (int c = 0; c < nf3; ++c) {
                        float xc = s_mean[c];
#pragma unroll
                        for (int k = 0; k < 16; ++k)
                            if (k < keep) xc += fxv[k] * s_T[k * nf3 + c];
                        xr[s_off[c]] = xc;
                    }
                } else {
                    for (int c = 0; c < nf3; ++c) xr[s_off[c]] = s_mean[c];
                    for (int k = 0; k < keep; ++k) {
                        const float zk = icdf_channel(fx[k], s_dsc + 6 * (3 * n + k), g.use_eps, g.cdf_eps, acc);
                        for (int c = 0; c < nf3; ++c) xr[s_off[c]] += zk * s_T[k * nf3 + c];
                    }
                }
                acc += -a.jac_xz;
            } else {
                if (park_fixed) {
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if (c < nf3) xr[s_off[c]] = icdf_channel(fxv[c], s_dsc + 6 * (3 * n + c), g.use_eps, g.cdf_eps, acc);
                } else {
                    for (int c = 0; c < nf3; ++c) xr[s_off[c]] = icdf_channel(fx[c], s_dsc + 6 * (3 * n + c), g.use_eps, g.cdf_eps, acc);
                }
            }
            if (a.normalize) acc += (float)n * logf(PI_F) + (float)n * logf(2.0f * PI_F);
            for (int i = 0; i < n; ++i) {
                const int at = a.table[5 * i], i1 = a.table[5 * i + 1], i2 = a.table[5 * i + 2], i3 = a.table[5 * i + 3];
                const int zc = a.table[5 * i + 4];
                const float vb = xr[3 * at], va = xr[3 * at + 1], vt = xr[3 * at + 2];      /* parked inputs of this placement */
                const float dd = icdf_channel(vb, s_dsc + 6 * zc, g.use_eps, g.cdf_eps, acc);
                const float an = icdf_channel(va, s_dsc + 6 * (n + zc), g.use_eps, g.cdf_eps, acc);
                const float tn = icdf_channel(vt, s_dsc + 6 * (2 * n + zc), g.use_eps, g.cdf_eps, acc);
                V3 p1 = ld3(xr + 3 * i1), p2 = ld3(xr + 3 * i2), p3 = ld3(xr + 3 * i3);
                float st, ct, sa, ca;
                if (a.normalize) {
                    bgk_sincos2pif(0.5f * an, &sa, &ca);
                    bgk_sincos2pif(tn, &st, &ct);
                    st = -st; ct = -ct;
                } else {
                    st = sinf(tn); ct = cosf(tn); sa = sinf(an); ca = cosf(an);
                }
                V3 v1 = sub(p1, p2), v2 = sub(p1, p3);
                V3 nv = cross(v1, v2), nn = cross(v1, nv);
                const int warn0 = warn;
                const float inv_nv = inv_norm_flag(nv, a.eps, a.enforce, warn), inv_nn = inv_norm_flag(nn, a.eps, a.enforce, warn);
                V3 nh = {nv.x * inv_nv, nv.y * inv_nv, nv.z * inv_nv}, nnh = {nn.x * inv_nn, nn.y * inv_nn, nn.z * inv_nn};
                V3 v3 = {nh.x * (-st) + nnh.x * ct, nh.y * (-st) + nnh.y * ct, nh.z * (-st) + nnh.z * ct};
                const float inv_v3 = inv_norm_flag(v3, a.eps, a.enforce, warn), inv_v1 = inv_norm_flag(v1, a.eps, a.enforce, warn);
                const float ks = dd * sa * inv_v3, kc = dd * ca * inv_v1;
                xr[3 * at] = p1.x + v3.x * ks - v1.x * kc;
                xr[3 * at + 1] = p1.y + v3.y * ks - v1.y * kc;
                xr[3 * at + 2] = p1.z + v3.z * ks - v1.z * kc;
                if (warn == warn0) acc += (2.0f * __builtin_amdgcn_logf(dd) + __builtin_amdgcn_logf(fabsf(sa))) * 0.693147180559945309f;
                else acc += explicit_placement_logdet(p1, p2, p3, dd, st, ct, sa, ca, a.eps, a.enforce);   /* rare: degenerate geometry */
            }
            if (a.accumulate) a.dlogp[b] += acc; else a.dlogp[b] = acc;
        }
        __syncthreads();
        {   /* full rows out: i / na3 by multiply-high (exact while i na3 < 2^32), the products by the full-rate 24-bit multiply */
            float* x_t = a.x + b0 * a.ldx;
            for (int i = tid; i < rows * na3; i += (int)blockDim.x) {
                const int r = (int)__umulhi((unsigned)i, magic_a), c = i - (int)__umul24((unsigned)r, (unsigned)na3);
                x_t[(int)__umul24((unsigned)r, (unsigned)ldx32) + c] = s_x[(int)__umul24((unsigned)r, (unsigned)a.sx) + c];
            }
        }
        __syncthreads();
    }
    if (warn && a.warn_count) atomicAdd(a.warn_count, warn);
}

/* ---- backward (VJP) of ic_ic2xyz_kernel: reverse sweep over the placement table ------------------
 * Same math as oracle/bgo_impl.h::bgo_ic_ic2xyz_backward (hand-derived adjoint of ic2xyz_deriv,
 * log|det J| = 2 ln d + ln|sin a|).  Lane = sample; x (forward output) and the running position
 * adjoints live in per-lane LDS rows; IC tiles are overwritten in place by their gradients. */
#include "bgk_dma.h"
constexpr int ICB_THREADS = 64;

struct IcBwdArgs {
    const float* bonds; const float* angles; const float* torsions; int64_t ldic;
    const float* x; int64_t ldx;
    const float* g_x; int64_t ldgx;
    const float* g_dlogp;
    const int32_t* place; const int32_t* fixed;
    int n, n_fixed, n_atoms, keep, normalize;
    const float* T;                     /* Tblacken [keep, 3nf] or NULL */
    int64_t B;
    float* g_bonds; float* g_angles; float* g_torsions; int64_t ldgic;
    float* g_xfix; int64_t ldgf;
    int sx, sic, sfx;
    float eps; int enforce;             /* the forward's norm clamps (ic_helper.py:372-452): where one fired the adjoint is evaluated on dual numbers */
    int32_t* fix;                       /* [1 + B]: count and indices of the samples a sweep kernel hands to ic_ic2xyz_bwd_fix_kernel */
};

__device__ __forceinline__ void tile_load64(float* dst, int s, const float* src, int64_t ld, int rows, int cols) {
    for (int i = threadIdx.x; i < rows * cols; i += (int)blockDim.x) {
        int r = i / cols, c = i - r * cols;
        dst[r * s + c] = src[(int64_t)r * ld + c];
    }
}
__device__ __forceinline__ void tile_store64(float* dst, int64_t ld, const float* src, int s, int rows, int cols) {
    for (int i = threadIdx.x; i < rows * cols; i += (int)blockDim.x) {
        int r = i / cols, c = i - r * cols;
        dst[(int64_t)r * ld + c] = src[r * s + c];
    }
}
__device__ __forceinline__ V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 proj_out(V3 g, V3 u, float inv_norm) {   /* (g - u (u.g)) / |v|  : adjoint of v -> v/|v| */
    float p = dot(u, g);
    return {(g.x - u.x * p) * inv_norm, (g.y - u.y * p) * inv_norm, (g.z - u.z * p) * inv_norm};
}

/* 1 / |v| from the squared norm: hardware rsq + one Newton step (the adjoint of a normalisation is amplified by 1 / |v| twice over when
 * the placement's reference atoms nearly coincide -- a bond of 5e-4 nm makes |v1 x (v1 x v2)| ~ 1e-7) */
__device__ __forceinline__ float inv_sqrt_refined(float n2) {
    const float r = __builtin_amdgcn_rsqf(n2);
    return r * __builtin_fmaf(-0.5f * n2, r * r, 1.5f);
}
/* sin(2 pi f), f in [0, 1): the hardware form (|error| <= 1.3e-7 ABSOLUTE) except within 1/32 revolution of a zero, where the
 * log-det term gl cos a / sin a needs the sine to RELATIVE accuracy (an angle 1e-5 from pi: 1 % off on the hardware form) --
 * there the odd series in the exact distance to the zero */
__device__ __forceinline__ float sin_rev_near_zero_exact(float f) {
    const float k = __builtin_rintf(2.0f * f);                 /* 0, 1 or 2 half revolutions */
    const float r = __builtin_fmaf(k, -0.5f, f);               /* exact */
    const float x = r * 6.28318530717958647692f, z = x * x;
    float p = __builtin_fmaf(z, 8.33333333e-3f, -1.66666667e-1f);
    p = __builtin_fmaf(p * z, x, x);
    p = ((int)k & 1) ? -p : p;
    return __builtin_fabsf(r) < 0.03125f ? p : __builtin_amdgcn_sinf(f);
}

/* VJP of ONE placement exactly as the reference differentiates it -- torch autograd through ic2xyz_deriv (ic_helper.py:372-452) and
 * log|det J| of its explicit Jacobian (ic.py:435-513), eps clamps with torch.clamp's derivative (a clamped norm passes none) -- on
 * forward-mode dual numbers: 12 inputs, 4 passes of Dual<3>.  Taken only where a norm of the placement was clamped (atoms that
 * nearly coincide: a bond drawn 1e-4 from the lower end of its truncated-normal marginal): there the clamped vectors are no unit
 * vectors, log|det J| is not 2 ln d + ln|sin a|, and the closed-form adjoint of the sweep below is off by O(1). */
struct PlaceAdj { V3 g1, g2, g3; float gd, ga, gt; };

/* sin / cos of a dual whose VALUE's sine and cosine are known (they do not change from pass to pass) */
template <int N> __device__ __forceinline__ Dual<N> dsin_pre(Dual<N> a, float s, float c) { Dual<N> r; r.v = s; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
template <int N> __device__ __forceinline__ Dual<N> dcos_pre(Dual<N> a, float s, float c) { Dual<N> r; r.v = c; for (int i = 0; i < N; ++i) r.d[i] = -(a.d[i] * s); return r; }

template <int N>
__device__ __forceinline__ Dual<N> place_scalar_dual(DV3<N> p1, DV3<N> p2, DV3<N> p3, Dual<N> d, Dual<N> a, Dual<N> t, V3 g, float gl, float eps,
                                                     float sin_t, float cos_t, float sin_a, float cos_a) {
    typedef Dual<N> D;
    typedef DV3<N> W;
    const W v1 = dsub(p1, p2), v2 = dsub(p1, p3);
    const W nv = dcross(v1, v2), nn = dcross(v1, nv);
    const D nvn = dclamp_min(dnorm(nv), eps), nnn = dclamp_min(dnorm(nn), eps);
    const W nh = ddivs(nv, nvn), nnh = ddivs(nn, nnn);
    const D st = dsin_pre(t, sin_t, cos_t), ct = dcos_pre(t, sin_t, cos_t), sa = dsin_pre(a, sin_a, cos_a), ca = dcos_pre(a, sin_a, cos_a);
    const W v3 = dadd(dscale(nh, -st), dscale(nnh, ct));
    const D v3n = dclamp_min(dnorm(v3), eps);
    const W v3h = ddivs(v3, v3n);
    const D v1n = dclamp_min(dnorm(v1), eps);
    const W v1h = ddivs(v1, v1n);
    const W pos = dadd(p1, dsub(dscale(v3h, d * sa), dscale(v1h, d * ca)));
    const W Jd = dsub(dscale(v3h, sa), dscale(v1h, ca));
    const W Ja = dadd(dscale(v3h, d * ca), dscale(v1h, d * sa));
    const W Jt3 = dadd(dscale(nh, -ct), dscale(nnh, -st));
    const D h3 = ddot(v3h, Jt3);
    const W Jt = dscale(dsub(Jt3, dscale(v3h, h3)), (d * sa) / v3n);
    const W R0 = {Jd.x, Ja.x, Jt.x}, R1 = {Jd.y, Ja.y, Jt.y}, R2 = {Jd.z, Ja.z, Jt.z};
    const D det = ddot(dcross(R0, R1), R2);
    return pos.x * g.x + pos.y * g.y + pos.z * g.z + dlog(dabs(det)) * gl;
}

/* N directional derivatives per pass, 12 / N passes in a rolled loop (the VALUES are recomputed by every pass: the fix-up kernel,
 * whose registers are free, takes four directions at a time) */
template <int N>
__device__ __forceinline__ PlaceAdj placement_vjp_dual(V3 p1, V3 p2, V3 p3, float dd, float a_rad, float t_rad, V3 g, float gl, float eps) {
    static_assert(12 % N == 0, "whole passes");
    float o[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = 0.0f;
    const float sin_t = sinf(t_rad), cos_t = cosf(t_rad), sin_a = sinf(a_rad), cos_a = cosf(a_rad);
#pragma unroll 1
    for (int pass = 0; pass < 12 / N; ++pass) {
        const int b = N * pass;
        DV3<N> q1 = {dseed<N>(p1.x, 0 - b), dseed<N>(p1.y, 1 - b), dseed<N>(p1.z, 2 - b)};
        DV3<N> q2 = {dseed<N>(p2.x, 3 - b), dseed<N>(p2.y, 4 - b), dseed<N>(p2.z, 5 - b)};
        DV3<N> q3 = {dseed<N>(p3.x, 6 - b), dseed<N>(p3.y, 7 - b), dseed<N>(p3.z, 8 - b)};
        const Dual<N> s = place_scalar_dual<N>(q1, q2, q3, dseed<N>(dd, 9 - b), dseed<N>(a_rad, 10 - b), dseed<N>(t_rad, 11 - b), g, gl, eps,
                                               sin_t, cos_t, sin_a, cos_a);
#pragma unroll
        for (int k = 0; k < 12; ++k)
#pragma unroll
            for (int e = 0; e < N; ++e) o[k] = (k == b + e) ? s.d[e] : o[k];
    }
    PlaceAdj r;
    r.g1 = {o[0], o[1], o[2]}; r.g2 = {o[3], o[4], o[5]}; r.g3 = {o[6], o[7], o[8]};
    r.gd = o[9]; r.ga = o[10]; r.gt = o[11];
    return r;
}

/* The same twelve directional derivatives, one LANE each (lanes 0 .. 11 of a wave whose lanes all hold the SAME placement; the other
 * lanes compute a zero direction): one pass of Dual<1> instead of three of Dual<4>, the results collected with v_readlane -- the
 * one-wave-per-sample fix-up kernel (round 6). */
__device__ __forceinline__ PlaceAdj placement_vjp_dual_lanes(V3 p1, V3 p2, V3 p3, float dd, float a_rad, float t_rad, V3 g, float gl, float eps) {
    const int b = (int)(threadIdx.x & 63);
    const float sin_t = sinf(t_rad), cos_t = cosf(t_rad), sin_a = sinf(a_rad), cos_a = cosf(a_rad);
    DV3<1> q1 = {dseed<1>(p1.x, 0 - b), dseed<1>(p1.y, 1 - b), dseed<1>(p1.z, 2 - b)};
    DV3<1> q2 = {dseed<1>(p2.x, 3 - b), dseed<1>(p2.y, 4 - b), dseed<1>(p2.z, 5 - b)};
    DV3<1> q3 = {dseed<1>(p3.x, 6 - b), dseed<1>(p3.y, 7 - b), dseed<1>(p3.z, 8 - b)};
    const Dual<1> s = place_scalar_dual<1>(q1, q2, q3, dseed<1>(dd, 9 - b), dseed<1>(a_rad, 10 - b), dseed<1>(t_rad, 11 - b), g, gl, eps,
                                           sin_t, cos_t, sin_a, cos_a);
    float o[12];
    const int bits = __builtin_bit_cast(int, s.d[0]);
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bits, k));
    PlaceAdj r;
    r.g1 = {o[0], o[1], o[2]}; r.g2 = {o[3], o[4], o[5]}; r.g3 = {o[6], o[7], o[8]};
    r.gd = o[9]; r.ga = o[10]; r.gt = o[11];
    return r;
}

/* adjoint of one placement: cotangents of the three reference atoms (g1 includes the pass-through of g), of the bond, the angle and
 * the torsion (in the caller's units: normalised angles get their pi / 2 pi).  Angles go to the hardware sin / cos in revolutions
 * (tools/ubench/hw_sincos.hip); reciprocals on the hardware form. */
/* DUAL: a placement whose norms the forward clamped is differentiated on dual numbers right here (placement_vjp_dual); !DUAL: the
 * closed form is evaluated regardless and `bad` is raised -- the sweep kernels hand such samples to ic_ic2xyz_bwd_fix_kernel, so that
 * the rare path's registers and code stay out of their loops (inlined there it doubled the loop: the position arrays went to AGPRs
 * and their wave-uniform indexing from v_movrel to select chains). */
template <bool DUAL, int ND = 1, bool LANES = false>
__device__ __forceinline__ PlaceAdj placement_adjoint(V3 p1, V3 p2, V3 p3, float dd, float an, float t, V3 g, float gl, int normalize,
                                                      float eps, int enforce, bool& bad) {
    PlaceAdj o;
    const float an_rev = normalize ? 0.5f * an : an * (0.5f / PI_F);
    const float t_rev = normalize ? t - 0.5f : t * (0.5f / PI_F);
    V3 v1 = sub(p1, p2), v2 = sub(p1, p3);
    V3 nv = cross(v1, v2), nn = cross(v1, nv);
    const float n2_nv = dot(nv, nv), n2_nn = dot(nn, nn), n2_v1 = dot(v1, v1), e2 = eps * eps;
    if (enforce && (n2_nv < e2 || n2_nn < e2 || n2_v1 < e2)) {       /* rare: a norm of this placement was clamped by the forward */
        if constexpr (DUAL) {
            const float a_rad = normalize ? an * PI_F : an, t_rad = normalize ? t * (2.0f * PI_F) - PI_F : t;
            if constexpr (LANES) o = placement_vjp_dual_lanes(p1, p2, p3, dd, a_rad, t_rad, g, gl, eps);       /* (the condition is wave-uniform there) */
            else o = placement_vjp_dual<ND>(p1, p2, p3, dd, a_rad, t_rad, g, gl, eps);
            if (normalize) { o.ga *= PI_F; o.gt *= 2.0f * PI_F; }
            return o;
        } else {
            bad = true;
        }
    }
    float inv_nv = inv_sqrt_refined(n2_nv), inv_nn = inv_sqrt_refined(n2_nn), inv_v1 = inv_sqrt_refined(n2_v1);
    V3 nh = scale(nv, inv_nv), nnh = scale(nn, inv_nn), v1h = scale(v1, inv_v1);
    const float tf = __builtin_amdgcn_fractf(t_rev), af = __builtin_amdgcn_fractf(an_rev);
    float st = __builtin_amdgcn_sinf(tf), ct = __builtin_amdgcn_cosf(tf), sa = sin_rev_near_zero_exact(af), ca = __builtin_amdgcn_cosf(af);
    V3 v3 = add(scale(nh, -st), scale(nnh, ct));
    float inv_v3 = inv_sqrt_refined(dot(v3, v3));
    V3 v3h = scale(v3, inv_v3);
    float gd = dot(g, add(scale(v3h, sa), scale(v1h, -ca))) + gl * 2.0f * __builtin_amdgcn_rcpf(dd);
    float ga = dot(g, add(scale(v3h, dd * ca), scale(v1h, dd * sa))) + gl * ca * __builtin_amdgcn_rcpf(sa);
    V3 g_v3 = proj_out(scale(g, dd * sa), v3h, inv_v3);
    float gt = dot(g_v3, add(scale(nh, -ct), scale(nnh, -st)));
    V3 g_n = proj_out(scale(g_v3, -st), nh, inv_nv);
    V3 g_nn = proj_out(scale(g_v3, ct), nnh, inv_nn);
    V3 g_v1 = cross(nv, g_nn);
    g_n = add(g_n, cross(g_nn, v1));
    g_v1 = add(g_v1, cross(v2, g_n));
    V3 g_v2 = cross(g_n, v1);
    g_v1 = add(g_v1, proj_out(scale(g, -dd * ca), v1h, inv_v1));
    o.g1 = {g.x + g_v1.x + g_v2.x, g.y + g_v1.y + g_v2.y, g.z + g_v1.z + g_v2.z};
    o.g2 = {-g_v1.x, -g_v1.y, -g_v1.z};
    o.g3 = {-g_v2.x, -g_v2.y, -g_v2.z};
    if (normalize) { ga = ga * PI_F; gt = gt * (2.0f * PI_F); }
    o.gd = gd; o.ga = ga; o.gt = gt;
    return o;
}

/* a sample one of whose placements had a clamped norm: the sweep kernels (closed-form adjoint) append it to the list the fix-up
 * kernel works through */
__device__ __forceinline__ void flag_for_fixup(int32_t* fix, int64_t b) {
    const int k = atomicAdd(fix, 1);
    fix[1 + k] = (int32_t)b;
}

template <bool DUAL>
__global__ __launch_bounds__(ICB_THREADS) void ic_ic2xyz_bwd_kernel(IcBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = (int)blockDim.x, n = a.n, nf3 = 3 * a.n_fixed;
    float* s_x = smem;                        /* [TS][sx] forward positions */
    float* s_g = s_x + TS * a.sx;             /* [TS][sx] position adjoints */
    float* s_b = s_g + TS * a.sx;             /* [TS][sic] bonds -> g_bonds */
    float* s_a = s_b + TS * a.sic;
    float* s_t = s_a + TS * a.sic;
    float* s_f = s_t + TS * a.sic;            /* [TS][sfx] g_xfix */
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        tile_load64(s_x, a.sx, a.x + b0 * a.ldx, a.ldx, rows, 3 * a.n_atoms);
        tile_load64(s_g, a.sx, a.g_x + b0 * a.ldgx, a.ldgx, rows, 3 * a.n_atoms);
        tile_load64(s_b, a.sic, a.bonds + b0 * a.ldic, a.ldic, rows, n);
        tile_load64(s_a, a.sic, a.angles + b0 * a.ldic, a.ldic, rows, n);
        tile_load64(s_t, a.sic, a.torsions + b0 * a.ldic, a.ldic, rows, n);
        __syncthreads();
        if (tid < rows) {
            const float* xr = s_x + tid * a.sx;
            float* gp = s_g + tid * a.sx;
            const float gl = a.g_dlogp[b0 + tid];
            /* a sample whose upstream adjoints are all zero (e.g. masked out of the loss because its geometry
             * is degenerate and its log-det is -inf) must get exactly zero gradients, not 0 * inf = NaN */
            bool live = gl != 0.0f;
            bool bad = false;
            for (int c = 0; c < 3 * a.n_atoms; ++c) live = live || (gp[c] != 0.0f);
            for (int i = n - 1; i >= 0; --i) {
                if (!live) { s_b[tid * a.sic + i] = 0.0f; s_a[tid * a.sic + i] = 0.0f; s_t[tid * a.sic + i] = 0.0f; continue; }
                const int at = a.place[5 * i], i1 = a.place[5 * i + 1], i2 = a.place[5 * i + 2],
                          i3 = a.place[5 * i + 3], zr = a.place[5 * i + 4];
                V3 p1 = ld3(xr + 3 * i1), p2 = ld3(xr + 3 * i2), p3 = ld3(xr + 3 * i3);
                float dd = s_b[tid * a.sic + zr], an = s_a[tid * a.sic + zr], t = s_t[tid * a.sic + zr];
                const V3 g = ld3(gp + 3 * at);
                const PlaceAdj q = placement_adjoint<DUAL>(p1, p2, p3, dd, an, t, g, gl, a.normalize, a.eps, a.enforce, bad);
                gp[3 * i1] += q.g1.x; gp[3 * i1 + 1] += q.g1.y; gp[3 * i1 + 2] += q.g1.z;
                gp[3 * i2] += q.g2.x; gp[3 * i2 + 1] += q.g2.y; gp[3 * i2 + 2] += q.g2.z;
                gp[3 * i3] += q.g3.x; gp[3 * i3 + 1] += q.g3.y; gp[3 * i3 + 2] += q.g3.z;
                const float gd = q.gd, ga = q.ga, gt = q.gt;
                s_b[tid * a.sic + zr] = gd; s_a[tid * a.sic + zr] = ga; s_t[tid * a.sic + zr] = gt;
            }
            if (a.T) {
                for (int k = 0; k < a.keep; ++k) {
                    float s = 0.0f;
                    for (int c = 0; c < nf3; ++c) s += gp[3 * a.fixed[c / 3] + c % 3] * a.T[k * nf3 + c];
                    s_f[tid * a.sfx + k] = s;
                }
            } else {
                for (int c = 0; c < nf3; ++c) s_f[tid * a.sfx + c] = gp[3 * a.fixed[c / 3] + c % 3];
            }
            if (!DUAL && bad) flag_for_fixup(a.fix, b0 + tid);
        }
        __syncthreads();
        tile_store64(a.g_bonds + b0 * a.ldgic, a.ldgic, s_b, a.sic, rows, n);
        tile_store64(a.g_angles + b0 * a.ldgic, a.ldgic, s_a, a.sic, rows, n);
        tile_store64(a.g_torsions + b0 * a.ldgic, a.ldgic, s_t, a.sic, rows, n);
        tile_store64(a.g_xfix + b0 * a.ldgf, a.ldgf, s_f, a.sfx, rows, a.keep);
        __syncthreads();
    }
}

/* The same reverse sweep with the positions in REGISTERS (3 x NA values, indexed by the wave-uniform atom ids of the placement
 * table; the adjoints stay LDS rows: a second register set of that size is not promoted by the compiler) and ONE LDS region shared
 * by the x tile (staging only) and the three IC tiles: the rows of the kernel above (2 x 67 + 3 x 23 + 10 floats per lane at ala2
 * size = 54 KB per wave) let two waves live on a CU, and the sweep is a latency chain -- 0.40 ms for 2^18 samples = 0.5 TB/s.  Here
 * 38 KB per wave: four waves per CU. */
template <int NA>
__global__ __launch_bounds__(ICB_THREADS) void ic_ic2xyz_bwd_reg_kernel(IcBwdArgs a) {
    constexpr bool DUAL = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.n, nf3 = 3 * a.n_fixed, n_atoms = a.n_atoms, na3 = 3 * a.n_atoms;
    const int sreg = a.sx;                    /* launcher: max(3 n_atoms, 3 sic) | 1 */
    float* s_r = smem;                        /* [64][sreg]: x rows (staging), then bonds | angles | torsions rows (-> their gradients) */
    float* s_g = s_r + 64 * sreg;             /* [64][sreg] position adjoints */
    float* s_f = s_g + 64 * sreg;             /* [64][sfx] g_xfix */
    const int tid = threadIdx.x;
    typedef const __attribute__((address_space(4))) int32_t* ci32_t;     /* scalar loads: the atom ids index the register arrays */
    const ci32_t place = (ci32_t)a.place, fixed = (ci32_t)a.fixed;
    const int64_t n_tiles = (a.B + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * 64;
        const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
        float px[NA], py[NA], pz[NA];
        tile_load64(s_r, sreg, a.x + b0 * a.ldx, a.ldx, rows, na3);
        tile_load64(s_g, sreg, a.g_x + b0 * a.ldgx, a.ldgx, rows, na3);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (k < n_atoms) { px[k] = s_r[tid * sreg + 3 * k]; py[k] = s_r[tid * sreg + 3 * k + 1]; pz[k] = s_r[tid * sreg + 3 * k + 2]; }
        __syncthreads();
        float* s_b = s_r;
        float* s_a = s_r + a.sic;
        float* s_t = s_r + 2 * a.sic;            /* row stride sreg for all three */
        tile_load64(s_b, sreg, a.bonds + b0 * a.ldic, a.ldic, rows, n);
        tile_load64(s_a, sreg, a.angles + b0 * a.ldic, a.ldic, rows, n);
        tile_load64(s_t, sreg, a.torsions + b0 * a.ldic, a.ldic, rows, n);
        __syncthreads();
        if (tid < rows) {
            float* gp = s_g + tid * sreg;
            const float gl = a.g_dlogp[b0 + tid];
            /* a sample whose upstream adjoints are all zero (e.g. masked out of the loss because its geometry is degenerate and its
             * log-det is -inf) must get exactly zero gradients, not 0 * inf = NaN */
            bool live = gl != 0.0f;
            bool bad = false;
            for (int c = 0; c < na3; ++c) live = live || (gp[c] != 0.0f);
            for (int i = n - 1; i >= 0; --i) {
                const int at = place[5 * i], i1 = place[5 * i + 1], i2 = place[5 * i + 2], i3 = place[5 * i + 3], zr = place[5 * i + 4];
                float dd = s_b[tid * sreg + zr], an = s_a[tid * sreg + zr], t = s_t[tid * sreg + zr];
                if (!live) { s_b[tid * sreg + zr] = 0.0f; s_a[tid * sreg + zr] = 0.0f; s_t[tid * sreg + zr] = 0.0f; continue; }
                const V3 p1 = {px[i1], py[i1], pz[i1]}, p2 = {px[i2], py[i2], pz[i2]}, p3 = {px[i3], py[i3], pz[i3]};
                const V3 g = ld3(gp + 3 * at);
                const PlaceAdj q = placement_adjoint<DUAL>(p1, p2, p3, dd, an, t, g, gl, a.normalize, a.eps, a.enforce, bad);
                gp[3 * i1] += q.g1.x; gp[3 * i1 + 1] += q.g1.y; gp[3 * i1 + 2] += q.g1.z;
                gp[3 * i2] += q.g2.x; gp[3 * i2 + 1] += q.g2.y; gp[3 * i2 + 2] += q.g2.z;
                gp[3 * i3] += q.g3.x; gp[3 * i3 + 1] += q.g3.y; gp[3 * i3 + 2] += q.g3.z;
                const float gd = q.gd, ga = q.ga, gt = q.gt;
                s_b[tid * sreg + zr] = gd; s_a[tid * sreg + zr] = ga; s_t[tid * sreg + zr] = gt;
            }
            if (a.T) {
                for (int k = 0; k < a.keep; ++k) {
                    float s = 0.0f;
                    for (int c = 0; c < nf3; ++c) s += gp[3 * fixed[c / 3] + c % 3] * a.T[k * nf3 + c];
                    s_f[tid * a.sfx + k] = s;
                }
            } else {
                for (int c = 0; c < nf3; ++c) s_f[tid * a.sfx + c] = gp[3 * fixed[c / 3] + c % 3];
            }
            if (bad) flag_for_fixup(a.fix, b0 + tid);
        }
        __syncthreads();
        tile_store64(a.g_bonds + b0 * a.ldgic, a.ldgic, s_b, sreg, rows, n);
        tile_store64(a.g_angles + b0 * a.ldgic, a.ldgic, s_a, sreg, rows, n);
        tile_store64(a.g_torsions + b0 * a.ldgic, a.ldgic, s_t, sreg, rows, n);
        tile_store64(a.g_xfix + b0 * a.ldgf, a.ldgf, s_f, a.sfx, rows, a.keep);
        __syncthreads();
    }
}

/* The sweep for CONTIGUOUS tensors (what the flow hands over: x / g_x [B, 3 n_atoms], the three IC fields [B, n] each, g_xfix
 * [B, keep], all 16-byte aligned) -- round 5.  The two kernels above fetch their tiles with one dword per lane and loop iteration
 * (an integer division and a dependent global load each: 183 serial round trips per tile at ala2 size, which is where their time
 * went: 0.30 ms for 2^18 samples = 0.9 TB/s).  Here a wave owns 64 samples and
 *   1. x and g_x arrive as linear DMA copies (bgk_dma.h: 18 requests per tile, all in flight at once); the lane lifts ITS x row
 *      into registers (px / py / pz indexed by the wave-uniform atom ids), g_x stays in LDS as the running position adjoints;
 *   2. the three IC tiles are copied by DMA into the x tile's place (memory image [64][n]: a lane's row has an odd stride);
 *   3. the reverse sweep overwrites the IC values by their gradients in place, g_xfix goes behind them ([64][keep]);
 *   4. the four results leave as 16-byte coalesced stores of the tile images.
 * LDS: 2 x 64 x 3 n_atoms floats per wave (33 KB at ala2 size): four waves per CU. */
/* -DBGK_ICB_TS=1: s_memtime stamps of the tile's phases (tools/r05_icb_ts.py); lane 0 writes them over the tile's first g_xfix row */
#ifndef BGK_ICB_TS
#define BGK_ICB_TS 0
#endif
#if BGK_ICB_TS
#define ICB_TS(k) do { __builtin_amdgcn_sched_barrier(0); ts_[k] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ICB_TS(k) do { } while (0)
#endif

template <int NA>
__global__ __launch_bounds__(64) void ic_ic2xyz_bwd_dma_kernel(IcBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.n, nf3 = 3 * a.n_fixed, n_atoms = a.n_atoms, na3 = 3 * a.n_atoms, keep = a.keep;
    const int lane = threadIdx.x;
    const int region = a.sx;                   /* launcher: 64 * max(3 n_atoms, 3 n + keep), a multiple of 4 */
    float* s_r = smem;                         /* x tile -> bonds | angles | torsions | g_xfix tiles */
    float* s_g = smem + region;                /* g_x tile = position adjoints, row stride 3 n_atoms */
    float* s_T = s_g + 64 * na3;               /* Tblacken transposed and padded: [3 n_fixed][16] (column c of the whitened fixed block as four 16-byte reads) */
    int* s_fo = reinterpret_cast<int*>(s_T + (a.T ? 16 * nf3 : 0));        /* offset of fixed coordinate c in a position row */
    typedef const __attribute__((address_space(4))) int32_t* ci32_t;
    const ci32_t place = (ci32_t)a.place;
    /* the wave-uniform tables of the fixed block once per wave: read per sample and coordinate from global memory they were 135
     * dependent scalar round trips per tile */
    for (int c = lane; c < nf3; c += 64) s_fo[c] = 3 * a.fixed[c / 3] + c % 3;
    if (a.T) for (int e = lane; e < 16 * nf3; e += 64) { const int c = e >> 4, kk = e & 15; s_T[e] = (kk < keep && keep <= 16) ? a.T[kk * nf3 + c] : 0.0f; }
    const int64_t n_tiles = (a.B + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * 64;
        const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
#if BGK_ICB_TS
        unsigned ts_[8];
#endif
        ICB_TS(0);
        dma_tile(s_r, a.x + b0 * na3, na3, rows, lane);
        dma_tile(s_g, a.g_x + b0 * na3, na3, rows, lane);
        const float gl = lane < rows ? a.g_dlogp[b0 + lane] : 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ICB_TS(1);
        float px[NA], py[NA], pz[NA];
        /* all NA slots unconditionally (slots beyond the molecule read the next row's first floats: in the region, never used): one
         * run of independent LDS reads instead of a branch and a wait per atom */
#pragma unroll
        for (int k = 0; k < NA; ++k) { px[k] = s_r[lane * na3 + 3 * k]; py[k] = s_r[lane * na3 + 3 * k + 1]; pz[k] = s_r[lane * na3 + 3 * k + 2]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float* s_b = s_r;
        float* s_a = s_r + 64 * n;
        float* s_t = s_r + 128 * n;
        float* s_f = s_r + 192 * n;            /* [64][keep] */
        dma_tile(s_b, a.bonds + b0 * n, n, rows, lane);
        dma_tile(s_a, a.angles + b0 * n, n, rows, lane);
        dma_tile(s_t, a.torsions + b0 * n, n, rows, lane);
        float* gp = s_g + lane * na3;
        /* a sample whose upstream adjoints are all zero (e.g. masked out of the loss because its geometry is degenerate and its
         * log-det is -inf) must get exactly zero gradients, not 0 * inf = NaN.  (g != 0 on the raw bits, sign dropped: a NaN counts.) */
        unsigned any = 0u;
        if ((na3 & 1) == 0) {
            const uint2* r2 = reinterpret_cast<const uint2*>(gp);
            for (int c = 0; c < (na3 >> 1); ++c) { const uint2 u = r2[c]; any |= (u.x | u.y); }
        } else {
            for (int c = 0; c < na3; ++c) any |= __builtin_bit_cast(unsigned, gp[c]);
        }
        const bool live = (gl != 0.0f) || ((any & 0x7fffffffu) != 0u);
        bool bad = false;
        ICB_TS(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ICB_TS(3);
        /* reverse sweep; the record and the three IC values of placement i - 1 are requested while placement i is evaluated */
        int at = place[5 * (n - 1)], i1 = place[5 * (n - 1) + 1], i2 = place[5 * (n - 1) + 2], i3 = place[5 * (n - 1) + 3], zr = place[5 * (n - 1) + 4];
        float dd = s_b[lane * n + zr], an = s_a[lane * n + zr], t = s_t[lane * n + zr];
        for (int i = n - 1; i >= 0; --i) {
            const int in = i > 0 ? i - 1 : 0;
            const int at_n = place[5 * in], i1_n = place[5 * in + 1], i2_n = place[5 * in + 2], i3_n = place[5 * in + 3], zr_n = place[5 * in + 4];
            const float dd_n = s_b[lane * n + zr_n], an_n = s_a[lane * n + zr_n], t_n = s_t[lane * n + zr_n];   /* (a different row than zr unless i == 0) */
            const V3 p1 = {px[i1], py[i1], pz[i1]}, p2 = {px[i2], py[i2], pz[i2]}, p3 = {px[i3], py[i3], pz[i3]};
            /* the adjoints of the placed atom and of its three reference atoms in ONE round of LDS reads (four distinct atoms): written
             * as nine read-modify-writes the compiler has to keep them in order -- they might alias -- and the sweep paid nine LDS
             * round trips per placement, half of its time */
            const V3 g = ld3(gp + 3 * at), a1 = ld3(gp + 3 * i1), a2 = ld3(gp + 3 * i2), a3 = ld3(gp + 3 * i3);
            PlaceAdj q = placement_adjoint<false>(p1, p2, p3, dd, an, t, g, gl, a.normalize, a.eps, a.enforce, bad);
            if (!live) { q.g1 = q.g2 = q.g3 = V3{0.0f, 0.0f, 0.0f}; q.gd = q.ga = q.gt = 0.0f; }
            gp[3 * i1] = a1.x + q.g1.x; gp[3 * i1 + 1] = a1.y + q.g1.y; gp[3 * i1 + 2] = a1.z + q.g1.z;
            gp[3 * i2] = a2.x + q.g2.x; gp[3 * i2 + 1] = a2.y + q.g2.y; gp[3 * i2 + 2] = a2.z + q.g2.z;
            gp[3 * i3] = a3.x + q.g3.x; gp[3 * i3 + 1] = a3.y + q.g3.y; gp[3 * i3 + 2] = a3.z + q.g3.z;
            s_b[lane * n + zr] = q.gd; s_a[lane * n + zr] = q.ga; s_t[lane * n + zr] = q.gt;
            at = at_n; i1 = i1_n; i2 = i2_n; i3 = i3_n; zr = zr_n; dd = dd_n; an = an_n; t = t_n;
        }
        ICB_TS(4);
        if (a.T && keep <= 16) {
            /* every whitened coordinate as its own running sum over the fixed coordinates c (ascending, product then sum: the order
             * and the operations of the other kernels): 16 independent chains instead of keep x 3 n_fixed dependent round trips */
            float acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
#pragma unroll 3
            for (int c = 0; c < nf3; ++c) {
                const float gc = gp[s_fo[c]];
                const float4* tc = reinterpret_cast<const float4*>(s_T + 16 * c);
                const float4 t0 = tc[0], t1 = tc[1], t2 = tc[2], t3 = tc[3];
                const float tv[16] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w};
#pragma unroll
                for (int k = 0; k < 16; ++k) { const float pr = gc * tv[k]; acc[k] = acc[k] + pr; }      /* (rows >= keep of the table are 0) */
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < keep) s_f[lane * keep + k] = acc[k];
        } else if (a.T) {                          /* more than 16 whitened coordinates: from global memory */
            for (int k = 0; k < keep; ++k) {
                float s = 0.0f;
                for (int c = 0; c < nf3; ++c) s += gp[s_fo[c]] * a.T[k * nf3 + c];
                s_f[lane * keep + k] = s;
            }
        } else {
            for (int c = 0; c < nf3; ++c) s_f[lane * keep + c] = gp[s_fo[c]];
        }
        if (bad && live && lane < rows) flag_for_fixup(a.fix, b0 + lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ICB_TS(5);
        {   /* the tile images out: 16-byte pieces, then the 1..3 floats a partial tile may leave over */
            float* const outs[4] = {a.g_bonds + b0 * n, a.g_angles + b0 * n, a.g_torsions + b0 * n, a.g_xfix + b0 * keep};
            const float* const srcs[4] = {s_b, s_a, s_t, s_f};
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int total = rows * (f < 3 ? n : keep), total4 = total >> 2;
                const float4* s4 = reinterpret_cast<const float4*>(srcs[f]);
                float4* g4 = reinterpret_cast<float4*>(outs[f]);
                for (int q = lane; q < total4; q += 64) g4[q] = s4[q];
                for (int q = (total4 << 2) + lane; q < total; q += 64) outs[f][q] = srcs[f][q];
            }
        }
        ICB_TS(6);
#if BGK_ICB_TS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ICB_TS(7);
        if (lane == 0)
            for (int q = 0; q < 8; ++q) reinterpret_cast<unsigned*>(a.g_xfix + b0 * keep)[q] = ts_[q];
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

/* The samples the sweep kernels flagged (fix[0] of them, indices behind it), one lane each: the lane copies its sample's rows (g_x, x,
 * bonds, angles, torsions) into a private LDS row -- sixteen independent loads per round trip -- and runs the same sweep with the
 * dual-number adjoint where a norm was clamped.  A few hundred samples of 2^18 at cfg 3's uniform prior: the launch costs its latency.
 * blockDim.x = lanes per workgroup (64, fewer for molecules whose rows do not fit 160 KB of LDS). */
__device__ __forceinline__ void copy_row16(float* dst, const float* __restrict__ src, int w) {
    constexpr int CW = 16;                /* loads in flight per round trip (48: no faster, measured) */
    for (int c0 = 0; c0 < w; c0 += CW) {
        float v[CW];
#pragma unroll
        for (int u = 0; u < CW; ++u) v[u] = src[c0 + u < w ? c0 + u : w - 1];
#pragma unroll
        for (int u = 0; u < CW; ++u)
            if (c0 + u < w) dst[c0 + u] = v[u];
    }
}

__global__ __launch_bounds__(64) void ic_ic2xyz_bwd_fix_kernel(IcBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.n, nf3 = 3 * a.n_fixed, na3 = 3 * a.n_atoms, keep = a.keep;
    const int count = a.fix[0], TS = (int)blockDim.x;
    if ((int)blockIdx.x * TS >= count) return;
    typedef const __attribute__((address_space(4))) int32_t* ci32_t;
    const ci32_t place = (ci32_t)a.place;
    /* the wave-uniform tables of the fixed block in LDS when they fit (a.sfx: the launcher's verdict), as in the DMA sweep */
    float* s_T = smem + (size_t)TS * a.sx;
    int* s_fo = reinterpret_cast<int*>(s_T + (a.T ? keep * nf3 : 0));
    const bool tables = a.sfx != 0;
    if (tables) {
        for (int c = threadIdx.x; c < nf3; c += TS) s_fo[c] = 3 * a.fixed[c / 3] + c % 3;
        if (a.T) for (int c = threadIdx.x; c < keep * nf3; c += TS) s_T[c] = a.T[c];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int k = blockIdx.x * TS + threadIdx.x; k < count; k += gridDim.x * TS) {
        const int64_t b = a.fix[1 + k];
        float* gp = smem + (size_t)threadIdx.x * a.sx;      /* row: position adjoints | positions | bonds | angles | torsions */
        float* xr = gp + na3;
        float* rb = xr + na3;
        float* ra = rb + n;
        float* rt = ra + n;
        copy_row16(gp, a.g_x + b * a.ldgx, na3);
        copy_row16(xr, a.x + b * a.ldx, na3);
        copy_row16(rb, a.bonds + b * a.ldic, n);
        copy_row16(ra, a.angles + b * a.ldic, n);
        copy_row16(rt, a.torsions + b * a.ldic, n);
        const float gl = a.g_dlogp[b];
        bool bad = false;
        for (int i = n - 1; i >= 0; --i) {
            const int at = place[5 * i], i1 = place[5 * i + 1], i2 = place[5 * i + 2], i3 = place[5 * i + 3], zr = place[5 * i + 4];
            const V3 p1 = ld3(xr + 3 * i1), p2 = ld3(xr + 3 * i2), p3 = ld3(xr + 3 * i3);
            const float dd = rb[zr], an = ra[zr], t = rt[zr];
            const V3 g = ld3(gp + 3 * at);
            const PlaceAdj q = placement_adjoint<true, 4>(p1, p2, p3, dd, an, t, g, gl, a.normalize, a.eps, a.enforce, bad);
            gp[3 * i1] += q.g1.x; gp[3 * i1 + 1] += q.g1.y; gp[3 * i1 + 2] += q.g1.z;
            gp[3 * i2] += q.g2.x; gp[3 * i2 + 1] += q.g2.y; gp[3 * i2 + 2] += q.g2.z;
            gp[3 * i3] += q.g3.x; gp[3 * i3 + 1] += q.g3.y; gp[3 * i3 + 2] += q.g3.z;
            a.g_bonds[b * a.ldgic + zr] = q.gd; a.g_angles[b * a.ldgic + zr] = q.ga; a.g_torsions[b * a.ldgic + zr] = q.gt;
        }
        if (a.T) {
            for (int kk = 0; kk < keep; ++kk) {
                float s = 0.0f;
                if (tables) for (int c = 0; c < nf3; ++c) s += gp[s_fo[c]] * s_T[kk * nf3 + c];
                else for (int c = 0; c < nf3; ++c) s += gp[3 * a.fixed[c / 3] + c % 3] * a.T[kk * nf3 + c];
                a.g_xfix[b * a.ldgf + kk] = s;
            }
        } else {
            for (int c = 0; c < nf3; ++c) a.g_xfix[b * a.ldgf + c] = gp[3 * a.fixed[c / 3] + c % 3];
        }
    }
}

/* Round 6: the same fix-up with ONE WAVE per listed sample.  The lane-per-sample kernel above pays for divergence: a listed sample has
 * one clamped placement (rarely two), but the 64 samples of a wave have theirs at different steps of the sweep, so the wave runs the
 * dual-number path (three passes of Dual<4>, ~1.8 k instructions) at nearly every step -- 0.139 ms for a few hundred samples at
 * cfg 3's prior.  Here all lanes of a wave hold the same sample: the clamp condition is wave-uniform (the dual path runs only at the
 * sample's own clamped steps), its twelve directions are twelve lanes (one pass of Dual<1>), the rows arrive by coalesced loads:
 * 0.026 ms.  4096 single-wave workgroups stride over the list, so a long list (a collapsed flow) is served no worse than before. */
__global__ __launch_bounds__(64) void ic_ic2xyz_bwd_fix_wave_kernel(IcBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = a.n, nf3 = 3 * a.n_fixed, na3 = 3 * a.n_atoms, keep = a.keep;
    const int count = a.fix[0], lane = (int)threadIdx.x;
    if ((int)blockIdx.x >= count) return;
    typedef const __attribute__((address_space(4))) int32_t* ci32_t;
    const ci32_t place = (ci32_t)a.place;
    float* gp = smem;                           /* position adjoints | positions | bonds | angles | torsions */
    float* xr = gp + na3;
    float* rb = xr + na3;
    float* ra = rb + n;
    float* rt = ra + n;
    for (int k = blockIdx.x; k < count; k += gridDim.x) {
        const int64_t b = a.fix[1 + k];
        for (int c = lane; c < na3; c += 64) { gp[c] = a.g_x[b * a.ldgx + c]; xr[c] = a.x[b * a.ldx + c]; }
        for (int c = lane; c < n; c += 64) { rb[c] = a.bonds[b * a.ldic + c]; ra[c] = a.angles[b * a.ldic + c]; rt[c] = a.torsions[b * a.ldic + c]; }
        const float gl = a.g_dlogp[b];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool bad = false;
        for (int i = n - 1; i >= 0; --i) {
            const int at = place[5 * i], i1 = place[5 * i + 1], i2 = place[5 * i + 2], i3 = place[5 * i + 3], zr = place[5 * i + 4];
            const V3 p1 = ld3(xr + 3 * i1), p2 = ld3(xr + 3 * i2), p3 = ld3(xr + 3 * i3);
            const float dd = rb[zr], an = ra[zr], t = rt[zr];
            const V3 g = ld3(gp + 3 * at);
            const PlaceAdj q = placement_adjoint<true, 1, true>(p1, p2, p3, dd, an, t, g, gl, a.normalize, a.eps, a.enforce, bad);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                gp[3 * i1] += q.g1.x; gp[3 * i1 + 1] += q.g1.y; gp[3 * i1 + 2] += q.g1.z;
                gp[3 * i2] += q.g2.x; gp[3 * i2 + 1] += q.g2.y; gp[3 * i2 + 2] += q.g2.z;
                gp[3 * i3] += q.g3.x; gp[3 * i3 + 1] += q.g3.y; gp[3 * i3 + 2] += q.g3.z;
                a.g_bonds[b * a.ldgic + zr] = q.gd; a.g_angles[b * a.ldgic + zr] = q.ga; a.g_torsions[b * a.ldgic + zr] = q.gt;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (a.T) {
            for (int kk = lane; kk < keep; kk += 64) {
                float s = 0.0f;
                for (int c = 0; c < nf3; ++c) s += gp[3 * a.fixed[c / 3] + c % 3] * a.T[kk * nf3 + c];
                a.g_xfix[b * a.ldgf + kk] = s;
            }
        } else {
            for (int c = lane; c < nf3; c += 64) a.g_xfix[b * a.ldgf + c] = gp[3 * a.fixed[c / 3] + c % 3];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

/* ---- global reference frame of the first three atoms (ReferenceSystemTransformation) -------------
 * 9 floats in, 9 floats out per sample; log|det J_9x9| in closed form -(2 ln d01 + 2 ln d12 + ln sin a012)
 * (the reference uses a batched autograd Jacobian + 24-term permutation expansion; see oracle bgo_refsys). */
struct RefSysArgs { const float* in; float* out; float* dlogp; int64_t B; int inverse, normalize, enforce, accumulate; float eps; };

__global__ __launch_bounds__(256) void ic_refsys_kernel(RefSysArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const float* v = a.in + 9 * b;
    float* o = a.out + 9 * b;
    int warn = 0;
    float dl;
    if (!a.inverse) {
        V3 x0 = ld3(v), x1 = ld3(v + 3), x2 = ld3(v + 6);
        V3 r01 = sub(x1, x0), r12 = sub(x2, x1);
        float d01 = clamp_min_flag(norm(r01), a.eps, a.enforce, warn);
        float d12 = clamp_min_flag(norm(r12), a.eps, a.enforce, warn);
        V3 aa = sub(x0, x1), cc = sub(x2, x1);
        float an = clamp_min_flag(norm(aa), a.eps, a.enforce, warn), cn = clamp_min_flag(norm(cc), a.eps, a.enforce, warn);
        float cosang = (aa.x / an) * (cc.x / cn) + (aa.y / an) * (cc.y / cn) + (aa.z / an) * (cc.z / cn);
        if (a.enforce) { cosang = cosang < -1.0f + a.eps ? -1.0f + a.eps : cosang; cosang = cosang > 1.0f - a.eps ? 1.0f - a.eps : cosang; }
        float a012 = acosf(cosang);
        float e1n = clamp_min_flag(norm(r01), a.eps, a.enforce, warn);
        V3 e1 = divs(r01, e1n);
        V3 e2 = cross(sub(x2, x0), e1);
        float e2n = clamp_min_flag(norm(e2), a.eps, a.enforce, warn);
        e2 = divs(e2, e2n);
        V3 e3 = cross(e2, e1);
        float alpha = atan2f(e1.x, -e1.y), beta = e1.z, gamma = atan2f(-e3.z, -e2.z);
        dl = -(2.0f * logf(d01) + 2.0f * logf(d12) + logf(sinf(a012)));
        if (a.normalize) {
            a012 = a012 / PI_F; alpha = (alpha + PI_F) / (2.0f * PI_F); gamma = (gamma + PI_F) / (2.0f * PI_F);
            dl += -logf(PI_F) - 2.0f * logf(2.0f * PI_F);
        }
        o[0] = x0.x; o[1] = x0.y; o[2] = x0.z; o[3] = d01; o[4] = d12; o[5] = a012; o[6] = alpha; o[7] = beta; o[8] = gamma;
    } else {
        V3 x0 = ld3(v);
        float d01 = v[3], d12 = v[4], a012 = v[5], alpha = v[6], beta = v[7], gamma = v[8];
        dl = 0.0f;
        if (a.normalize) {
            alpha = alpha * (2.0f * PI_F) - PI_F; gamma = gamma * (2.0f * PI_F) - PI_F; a012 = a012 * PI_F;
            dl += logf(PI_F) + 2.0f * logf(2.0f * PI_F);
        }
        dl += 2.0f * logf(d01) + 2.0f * logf(d12) + logf(sinf(a012));
        V3 p1 = {0.0f, 0.0f, d01}, p0 = {0.0f, 0.0f, 0.0f}, p3 = {0.0f, -1.0f, 0.0f};
        V3 v1 = sub(p1, p0), v2 = sub(p1, p3);
        V3 nv = cross(v1, v2), nn = cross(v1, nv);
        float nvn = clamp_min_flag(norm(nv), a.eps, a.enforce, warn), nnn = clamp_min_flag(norm(nn), a.eps, a.enforce, warn);
        float tq = 0.5f * PI_F, st = sinf(tq), ct = cosf(tq), sa = sinf(a012), ca = cosf(a012);
        V3 nh = divs(nv, nvn), nnh = divs(nn, nnn);
        V3 v3 = {nh.x * (-st) + nnh.x * ct, nh.y * (-st) + nnh.y * ct, nh.z * (-st) + nnh.z * ct};
        float v3n = clamp_min_flag(norm(v3), a.eps, a.enforce, warn), v1n = clamp_min_flag(norm(v1), a.eps, a.enforce, warn);
        V3 v3h = divs(v3, v3n), v1h = divs(v1, v1n);
        V3 p2 = {p1.x + v3h.x * d12 * sa - v1h.x * d12 * ca, p1.y + v3h.y * d12 * sa - v1h.y * d12 * ca,
                 p1.z + v3h.z * d12 * sa - v1h.z * d12 * ca};
        float bA = acosf(beta);
        float caA = cosf(alpha), saA = sinf(alpha), cb = cosf(bA), sb = sinf(bA), cg = cosf(gamma), sg = sinf(gamma);
        float Rz1[9] = {caA, -saA, 0, saA, caA, 0, 0, 0, 1}, Rx[9] = {1, 0, 0, 0, cb, -sb, 0, sb, cb}, Rz2[9] = {cg, -sg, 0, sg, cg, 0, 0, 0, 1};
        float T[9], R[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { float s = 0.0f; for (int k = 0; k < 3; ++k) s += Rz1[3 * i + k] * Rx[3 * k + j]; T[3 * i + j] = s; }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) { float s = 0.0f; for (int k = 0; k < 3; ++k) s += T[3 * i + k] * Rz2[3 * k + j]; R[3 * i + j] = s; }
        const float p1v[3] = {p1.x, p1.y, p1.z}, p2v[3] = {p2.x, p2.y, p2.z}, x0v[3] = {x0.x, x0.y, x0.z};
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            float s1 = 0.0f, s2 = 0.0f;
            for (int dd = 0; dd < 3; ++dd) { s1 += p1v[dd] * R[3 * e + dd]; s2 += p2v[dd] * R[3 * e + dd]; }
            o[e] = x0v[e]; o[3 + e] = s1 + x0v[e]; o[6 + e] = s2 + x0v[e];
        }
    }
    if (a.accumulate) a.dlogp[b] += dl; else a.dlogp[b] = dl;
    (void)warn;
}

/* samples per workgroup: the kernels keep one sample's atoms per thread in LDS; the default tile is halved until it fits 160 KB, so
 * molecules of several thousand atoms run (partially filled waves) instead of being refused */
int fit_tile(int default_ts, size_t floats_per_sample, size_t extra_floats) {
    int ts = default_ts;
    while (ts > 1 && sizeof(float) * ((size_t)ts * floats_per_sample + extra_floats) > 160 * 1024) ts >>= 1;
    return ts;
}

int ic_launch(bool to_ic, IcArgs& a, void* stream, const char* what) {
    a.n_atoms = a.n + a.n_fixed;
    a.sx = (3 * a.n_atoms) | 1;
    a.sic = a.n | 1;
    a.sfx = a.keep | 1;
    if (to_ic) {
        const size_t per = (size_t)(a.sx + 3 * a.sic + a.sfx);
        const int ts = fit_tile(IC_THREADS, per, 0);
        const size_t shmem = sizeof(float) * (size_t)ts * per;
        if (shmem > 160 * 1024) { bgk_set_error("%s: %d atoms do not fit the LDS tile", what, a.n_atoms); return BGK_EUNSUPPORTED; }
        const int64_t n_tiles = (a.B + ts - 1) / ts;
        const int grid = (int)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
        if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ic_xyz2ic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(ic_xyz2ic_kernel, dim3(grid), dim3(ts), shmem, (hipStream_t)stream, a);
    } else {
        const int ts = fit_tile(IC2_THREADS, (size_t)a.sx, 0);
        const size_t shmem2 = sizeof(float) * (size_t)ts * (size_t)a.sx;
        if (shmem2 > 160 * 1024) { bgk_set_error("%s: %d atoms do not fit the LDS tile", what, a.n_atoms); return BGK_EUNSUPPORTED; }
        const int64_t nt2 = (a.B + ts - 1) / ts;
        const int grid2 = (int)(nt2 < 256 * 32 ? nt2 : 256 * 32);
        if (shmem2 > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ic_ic2xyz_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(ic_ic2xyz_kernel, dim3(grid2), dim3(ts), shmem2, (hipStream_t)stream, a);
    }
    return bgk_launch_status(what);
}

}  // namespace

extern "C" int bgk_ic_xyz2ic(const float* x, int64_t ldx, const int32_t* zmat, int32_t n,
                             const int32_t* fixed, int32_t n_fixed, int32_t normalize_angles,
                             float eps, int32_t enforce_boundaries, const float* wh_mean,
                             const float* Twhiten, int32_t keep, float jac_xz, int64_t B,
                             float* bonds, float* angles, float* torsions, int64_t ldic,
                             float* xfix, int64_t ldf, float* dlogp, int32_t accumulate,
                             int32_t* warn_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_ic_xyz2ic: bad sizes");
    BGK_CHECK_ARG(x && zmat && fixed && bonds && angles && torsions && xfix && dlogp, "bgk_ic_xyz2ic: null pointer");
    BGK_CHECK_ARG(Twhiten ? (wh_mean != nullptr && keep > 0) : (keep == 3 * n_fixed), "bgk_ic_xyz2ic: bad whitening arguments");
    if (B == 0) return 0;
    IcArgs a{};
    a.x = const_cast<float*>(x); a.ldx = ldx; a.bonds = bonds; a.angles = angles; a.torsions = torsions;
    a.ldic = ldic; a.xfix = xfix; a.ldf = ldf; a.table = zmat; a.fixed = fixed; a.n = n; a.n_fixed = n_fixed;
    a.keep = keep; a.normalize = normalize_angles; a.enforce = enforce_boundaries; a.eps = eps;
    a.wh_mean = wh_mean; a.T = Twhiten; a.jac_xz = jac_xz; a.B = B; a.dlogp = dlogp; a.accumulate = accumulate;
    a.warn_count = warn_count;
    return ic_launch(true, a, stream, "bgk_ic_xyz2ic");
}

extern "C" int bgk_ic_ic2xyz(const float* bonds, const float* angles, const float* torsions,
                             int64_t ldic, const float* xfix, int64_t ldf, const int32_t* place,
                             int32_t n, const int32_t* fixed, int32_t n_fixed,
                             int32_t normalize_angles, float eps, int32_t enforce_boundaries,
                             const float* wh_mean, const float* Tblacken, int32_t keep,
                             float jac_xz, int64_t B, float* x, int64_t ldx, float* dlogp,
                             int32_t accumulate, int32_t* warn_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_ic_ic2xyz: bad sizes");
    BGK_CHECK_ARG(x && place && fixed && bonds && angles && torsions && xfix && dlogp, "bgk_ic_ic2xyz: null pointer");
    BGK_CHECK_ARG(Tblacken ? (wh_mean != nullptr && keep > 0) : (keep == 3 * n_fixed), "bgk_ic_ic2xyz: bad whitening arguments");
    if (B == 0) return 0;
    IcArgs a{};
    a.x = x; a.ldx = ldx; a.bonds = const_cast<float*>(bonds); a.angles = const_cast<float*>(angles);
    a.torsions = const_cast<float*>(torsions); a.ldic = ldic; a.xfix = const_cast<float*>(xfix); a.ldf = ldf;
    a.table = place; a.fixed = fixed; a.n = n; a.n_fixed = n_fixed; a.keep = keep;
    a.normalize = normalize_angles; a.enforce = enforce_boundaries; a.eps = eps;
    a.wh_mean = wh_mean; a.T = Tblacken; a.jac_xz = jac_xz; a.B = B; a.dlogp = dlogp; a.accumulate = accumulate;
    a.warn_count = warn_count;
    return ic_launch(false, a, stream, "bgk_ic_ic2xyz");
}

extern "C" int bgk_icdf_ic2xyz(const float* bonds, const float* angles, const float* torsions, int64_t ldic,
                               const float* xfix, int64_t ldf,
                               const float* desc_bonds, const float* desc_angles, const float* desc_torsions, const float* desc_fixed,
                               int32_t use_eps, float cdf_eps,
                               const int32_t* place, int32_t n, const int32_t* fixed, int32_t n_fixed,
                               int32_t normalize_angles, float eps, int32_t enforce_boundaries,
                               const float* wh_mean, const float* Tblacken, int32_t keep, float jac_xz, int64_t B,
                               float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_icdf_ic2xyz: bad sizes");
    BGK_CHECK_ARG(x && place && fixed && bonds && angles && torsions && xfix && dlogp, "bgk_icdf_ic2xyz: null pointer");
    BGK_CHECK_ARG(Tblacken ? (wh_mean != nullptr && keep > 0) : (keep == 3 * n_fixed), "bgk_icdf_ic2xyz: bad whitening arguments");
    BGK_CHECK_ARG(ldic < (1 << 24) && ldf < (1 << 24) && ldx < (1 << 24), "bgk_icdf_ic2xyz: row stride too large");
    if (B == 0) return 0;
    IcGenArgs g{};
    IcArgs& a = g.ic;
    a.x = x; a.ldx = ldx; a.bonds = const_cast<float*>(bonds); a.angles = const_cast<float*>(angles);
    a.torsions = const_cast<float*>(torsions); a.ldic = ldic; a.xfix = const_cast<float*>(xfix); a.ldf = ldf;
    a.table = place; a.fixed = fixed; a.n = n; a.n_fixed = n_fixed; a.keep = keep;
    a.normalize = normalize_angles; a.enforce = enforce_boundaries; a.eps = eps;
    a.wh_mean = wh_mean; a.T = Tblacken; a.jac_xz = jac_xz; a.B = B; a.dlogp = dlogp; a.accumulate = accumulate;
    a.warn_count = warn_count;
    a.n_atoms = n + n_fixed;
    a.sx = (3 * a.n_atoms) | 1;
    g.dsc_b = desc_bonds; g.dsc_a = desc_angles; g.dsc_t = desc_torsions; g.dsc_f = desc_fixed; g.use_eps = use_eps; g.cdf_eps = cdf_eps;
    const int nf3 = 3 * n_fixed;
    const size_t extra = (size_t)keep * nf3 + 2 * (size_t)nf3 + (size_t)(3 * n + keep) * 6 + (size_t)n;
    const int ts = fit_tile(IC2_THREADS, (size_t)a.sx, extra);
    size_t shmem = sizeof(float) * ((size_t)ts * (size_t)a.sx + extra);
    if (shmem > 160 * 1024) { bgk_set_error("bgk_icdf_ic2xyz: %d atoms do not fit the LDS tile", a.n_atoms); return BGK_EUNSUPPORTED; }
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(icdf_ic2xyz_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int64_t nt = (B + ts - 1) / ts;
    int grid = (int)(nt < 256 * 32 ? nt : 256 * 32);
    hipLaunchKernelGGL(icdf_ic2xyz_kernel, dim3(grid), dim3(ts), shmem, (hipStream_t)stream, g);
    return bgk_launch_status("bgk_icdf_ic2xyz");
}

extern "C" int bgk_ic_ic2xyz_backward(const float* bonds, const float* angles, const float* torsions,
                                      int64_t ldic, const float* x, int64_t ldx, const int32_t* place,
                                      int32_t n, const int32_t* fixed, int32_t n_fixed,
                                      int32_t normalize_angles, float eps, int32_t enforce_boundaries,
                                      const float* Tblacken, int32_t keep,
                                      int64_t B, const float* g_x, int64_t ldgx, const float* g_dlogp,
                                      float* g_bonds, float* g_angles, float* g_torsions, int64_t ldgic,
                                      float* g_xfix, int64_t ldgf, int32_t* fix_ws, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_ic_ic2xyz_backward: bad sizes");
    BGK_CHECK_ARG(bonds && angles && torsions && x && place && fixed && g_x && g_dlogp && g_bonds && g_angles &&
                  g_torsions && g_xfix, "bgk_ic_ic2xyz_backward: null pointer");
    BGK_CHECK_ARG(Tblacken ? keep > 0 : keep == 3 * n_fixed, "bgk_ic_ic2xyz_backward: bad whitening arguments");
    BGK_CHECK_ARG(B < (int64_t)0x7fffffff, "bgk_ic_ic2xyz_backward: batch too large for the fix-up list");
    IcBwdArgs a{};
    a.bonds = bonds; a.angles = angles; a.torsions = torsions; a.ldic = ldic; a.x = x; a.ldx = ldx;
    a.g_x = g_x; a.ldgx = ldgx; a.g_dlogp = g_dlogp; a.place = place; a.fixed = fixed; a.n = n; a.n_fixed = n_fixed;
    a.n_atoms = n + n_fixed; a.keep = keep; a.normalize = normalize_angles; a.T = Tblacken; a.B = B;
    a.g_bonds = g_bonds; a.g_angles = g_angles; a.g_torsions = g_torsions; a.ldgic = ldgic; a.g_xfix = g_xfix; a.ldgf = ldgf;
    a.sx = (3 * a.n_atoms) | 1; a.sic = n | 1; a.sfx = keep | 1;
    a.eps = eps; a.enforce = enforce_boundaries; a.fix = fix_ws;
    hipStream_t st = (hipStream_t)stream;
    const char* what = "bgk_ic_ic2xyz_backward";
    /* the sweep over all samples (closed-form adjoint; samples with a clamped norm are listed in fix_ws), then the listed samples again
     * with the dual-number adjoint.  Without a list (fix_ws == NULL) the generic kernel evaluates the dual numbers in line. */
    auto fixup = [&]() -> int {
        if (!enforce_boundaries) return bgk_launch_status(what);           /* nothing is ever clamped */
        IcBwdArgs f = a;
        f.sx = (6 * a.n_atoms + 3 * n) | 1;                    /* a lane's row: position adjoints | positions | bonds | angles | torsions */
        size_t tab = (size_t)(Tblacken ? keep * 3 * n_fixed : 0) + 3 * (size_t)n_fixed;      /* floats: Tblacken + the offsets of the fixed coordinates */
        f.sfx = tab <= 4096 ? 1 : 0;                           /* the fixed block's tables in LDS (else read from global memory) */
        if (!f.sfx) tab = 0;
        const int ts = fit_tile(64, (size_t)f.sx, tab);
        const size_t shm = sizeof(float) * ((size_t)ts * (size_t)f.sx + tab);
        if (shm > 160 * 1024) { bgk_set_error("%s: %d atoms do not fit the LDS rows of the fix-up launch", what, a.n_atoms); return BGK_EUNSUPPORTED; }
        if (shm > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ic_ic2xyz_bwd_fix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        /* one wave per listed sample (round 6); BGK_IC_FIX_LANES=1 keeps round 5's lane-per-sample kernel (the A/B of
         * profiles/r06_ab_runs.txt: 0.144 -> 0.026 ms on the few hundred listed samples of a cfg 3 KL step) */
        const bool lanes_only = getenv("BGK_IC_FIX_LANES") != nullptr;
        const size_t shw = sizeof(float) * (size_t)f.sx;
        if (!lanes_only && shw <= 160 * 1024) {
            if (shw > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ic_ic2xyz_bwd_fix_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(ic_ic2xyz_bwd_fix_wave_kernel, dim3(4096), dim3(64), shw, st, f);
            return bgk_launch_status(what);
        }
        hipLaunchKernelGGL(ic_ic2xyz_bwd_fix_kernel, dim3(ts == 64 ? 64 : 256), dim3(ts), shm, st, f);
        return bgk_launch_status(what);
    };
    if (fix_ws && enforce_boundaries) {
        const hipError_t e = hipMemsetAsync(fix_ws, 0, sizeof(int32_t), st);
        if (e != hipSuccess) { bgk_set_error("%s: %s", what, hipGetErrorString(e)); return (int)e; }
    }
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool contiguous = ldx == 3 * a.n_atoms && ldgx == 3 * a.n_atoms && ldic == n && ldgic == n && ldgf == keep &&
                            al16(x) && al16(g_x) && al16(bonds) && al16(angles) && al16(torsions) &&
                            al16(g_bonds) && al16(g_angles) && al16(g_torsions) && al16(g_xfix);
    const bool listed = fix_ws != nullptr || !enforce_boundaries;
    if (listed && a.n_atoms <= 32 && contiguous && !getenv("BGK_IC_BWD_LDS") && !getenv("BGK_IC_BWD_NODMA")) {      /* tiles by DMA, positions in registers */
        const int w = 3 * a.n_atoms > 3 * n + keep ? 3 * a.n_atoms : 3 * n + keep;
        IcBwdArgs b = a;
        b.sx = 64 * ((w + 3) & ~3);                                /* floats of the x / IC region; the g_x tile follows */
        const size_t shm = sizeof(float) * ((size_t)b.sx + 64 * (size_t)(3 * a.n_atoms) + (size_t)(Tblacken ? 16 * 3 * n_fixed : 0) + 3 * (size_t)n_fixed);
        int64_t nt = (B + 63) / 64;
        int grid = (int)(nt < 256 * 4 ? nt : 256 * 4);          /* one wave per SIMD (LDS): every wave walks its share of the tiles */
        if (a.n_atoms <= 24) hipLaunchKernelGGL(ic_ic2xyz_bwd_dma_kernel<24>, dim3(grid), dim3(64), shm, st, b);
        else hipLaunchKernelGGL(ic_ic2xyz_bwd_dma_kernel<32>, dim3(grid), dim3(64), shm, st, b);
        return fixup();
    }
    if (listed && a.n_atoms <= 32 && !getenv("BGK_IC_BWD_LDS")) {          /* positions in registers: 38 instead of 54 KB of LDS per wave */
        const int sreg = (a.sx > 3 * a.sic ? a.sx : 3 * a.sic) | 1;
        const size_t shm = sizeof(float) * 64 * (size_t)(2 * sreg + a.sfx);
        IcBwdArgs b = a;
        b.sx = sreg;                                             /* kernel: sreg = max(sx, 3 sic) -> pass it through sx */
        int64_t nt = (B + 63) / 64;
        int grid = (int)(nt < 256 * 28 ? nt : 256 * 28);
        if (a.n_atoms <= 24) hipLaunchKernelGGL(ic_ic2xyz_bwd_reg_kernel<24>, dim3(grid), dim3(ICB_THREADS), shm, st, b);
        else hipLaunchKernelGGL(ic_ic2xyz_bwd_reg_kernel<32>, dim3(grid), dim3(ICB_THREADS), shm, st, b);
        return fixup();
    }
    const size_t per = (size_t)(2 * a.sx + 3 * a.sic + a.sfx);
    const int ts = fit_tile(ICB_THREADS, per, 0);
    size_t shmem = sizeof(float) * (size_t)ts * per;
    if (shmem > 160 * 1024) { bgk_set_error("bgk_ic_ic2xyz_backward: %d atoms do not fit the LDS tile", a.n_atoms); return BGK_EUNSUPPORTED; }
    int64_t n_tiles = (B + ts - 1) / ts;
    int grid = (int)(n_tiles < 256 * 12 ? n_tiles : 256 * 12);
    if (listed) {
        if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ic_ic2xyz_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(ic_ic2xyz_bwd_kernel<false>, dim3(grid), dim3(ts), shmem, st, a);
        return fixup();
    }
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ic_ic2xyz_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(ic_ic2xyz_bwd_kernel<true>, dim3(grid), dim3(ts), shmem, st, a);
    return bgk_launch_status(what);
}

extern "C" int bgk_ic_refsys(const float* in, int64_t B, int32_t inverse, int32_t normalize_angles, float eps,
                             int32_t enforce_boundaries, float* out, float* dlogp, int32_t accumulate, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && in && out && dlogp, "bgk_ic_refsys: bad arguments");
    if (B == 0) return 0;
    RefSysArgs a{in, out, dlogp, B, inverse, normalize_angles, enforce_boundaries, accumulate, eps};
    hipLaunchKernelGGL(ic_refsys_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_ic_refsys");
}
