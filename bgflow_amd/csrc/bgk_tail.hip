/* bgk_tail.hip -- the sampling tail of a builder flow in one launch, second generation: the icdf domain maps of the four IC fields
 * (CDFTransform._inverse, nn/flow/cdf.py:36-45, marginals factory/icmarginals.py:41-77) + IC -> xyz
 * (RelativeInternalCoordinateTransformation._inverse / MixedCoordinateTransformation._inverse, nn/flow/crd_transform/ic.py:435-513,
 * 862-884, ic2xyz_deriv ic_helper.py:372-452) + PCA blackening (pca.py:85-93) + the running log|det J|.
 *
 * What changed against icdf_ic2xyz_kernel (bgk_ic.hip: 10.9 k VALU instructions per 64-sample tile, a third of them f32 arithmetic,
 * 157 spilled SGPRs, one wave per 17 KB of LDS, 0.48 ms per 2^20 samples = 14 % of the HBM rate):
 *   * the growing position table lives in REGISTERS (px / py / pz [NA]): the atom indices of a placement are wave-uniform, so the
 *     dynamic accesses are uniform register indexing (s_set_gpr_idx + v_mov) -- no LDS table, no per-lane address arithmetic, no
 *     LDS round trips on the dependent chain of the sequential placements, and LDS shrinks to one [64][n] staging slab per wave
 *     (4.4 KB: occupancy is limited by registers, 4 waves per SIMD, instead of by LDS);
 *   * every wave-uniform table (placement rows, channel descriptors, Tblacken, mean) is read by SCALAR loads: the channel kind is a
 *     scalar branch instead of a select chain over all kinds;
 *   * the field tiles are contiguous [64][n] blocks: staging is a linear copy (no index arithmetic at all), a lane then reads its own
 *     row with stride n (odd for the builder's fields: conflict free);
 *   * branch-free erfinv (M. Giles' single-precision polynomials, the tail branch behind a wave-level ballot) instead of the OCML
 *     routine; near a bound of a truncated-normal marginal the DISTANCE to the bound is evaluated directly by the reverted Taylor
 *     series of the normal CDF (host-side f64 coefficients), which removes the cancellation mu + sigma z that dominates the f32
 *     error of log|sin a| and 2 log d for near-degenerate samples;
 *   * |nn| = |v1| |nv| and |v3| = 1 hold away from the eps clamps: two reciprocal square roots per placement instead of four;
 *     a lane whose geometry hits an eps clamp (rare) is recomputed with the reference's explicit arithmetic, clamp by clamp;
 *   * the placement's log-det 2 ln d + ln|sin a| = ln|d (d sin a)| is one v_log, taken while the angle is mapped;
 *   * output rows are written straight from the registers (one 12-byte store per atom and lane).
 * Envelope: n_atoms <= 32, keep <= 16, contiguous field rows, normalised angles; anything else runs icdf_ic2xyz_kernel.
 */
#include "bgk_common.h"
#include "bgk_erf.h"

namespace {

#ifndef BGK_TAIL_ABL
#define BGK_TAIL_ABL 0                 /* timing ablations (wrong results): 1 one output store per lane, 2 no input loads, 4 one placement only */
#endif
constexpr int TW = 4;                  /* waves per workgroup (independent: they share nothing but the launch) */
constexpr int DSC = 20;                /* floats per channel descriptor (bgflow_amd/cdf.py::tail_descriptor) */
#ifndef BGK_TAIL_KMAX
#define BGK_TAIL_KMAX 16
#endif
constexpr int KMAX = BGK_TAIL_KMAX;    /* whitened / fixed coordinates held in registers */
constexpr float SMAX = 0.03f;          /* reverted-series window: s = (distance in cdf units) / pdf(bound) below this */
#define LN2_F 0.693147180559945309f

/* channel descriptor: [0] kind as int32 BITS (-1 none, 0 uniform, 1 normal, 2 truncated normal)
 *   uniform:  [1] low  [2] high - low  [5] log(high - low)
 *   normal / truncated normal:  y = mu + sigma sqrt2 e,  e = erfinv(xs v + xo),  -log_prob = e^2 + cst
 *     [1] mu  [2] sigma sqrt2  [3] xs  [4] xo  [5] cst  [6] 1 / (sigma sqrt2)  [13] sigma
 *     lower bound: [7] k = Z / pdf(alpha) (1e30: none)  [8..11] c2..c5  [12] y0 = mu + sigma alpha
 *     upper bound: [14] k  [15..18] c2..c5  [19] y0 = mu + sigma beta */

/* wave-uniform tables (channel descriptors, placement rows, whitening matrices) are read through the CONSTANT address space: loads
 * from it are invariant by definition, so they stay scalar loads (s_load_*) wherever they sit in the control flow, and what they
 * return lives in SGPRs -- register indices derived from them need no waterfall loop */
typedef const __attribute__((address_space(4))) float* cf32_t;
typedef const __attribute__((address_space(4))) int32_t* ci32_t;
struct Desc { float f[DSC]; };                                           /* one channel descriptor: s_load_dwordx16 + x4 */
struct Rec { int32_t at, i1, i2, i3, zr, pad0, pad1, pad2; };             /* one placement: atom, its three reference atoms, Z row */
/* element-wise copies out of the constant address space (adjacent scalar loads are merged into s_load_dwordx8 / x16) */
__device__ __forceinline__ Desc load_desc(cf32_t base, int row) {
    Desc d;
#pragma unroll
    for (int k = 0; k < DSC; ++k) d.f[k] = base[row * DSC + k];
    return d;
}
__device__ __forceinline__ Rec load_rec(ci32_t base, int row) {
    return Rec{base[8 * row], base[8 * row + 1], base[8 * row + 2], base[8 * row + 3], base[8 * row + 4], 0, 0, 0};
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}

struct CdfClamp { float lo, hi, ld_min; };      /* [eps, 1 - eps] and -1 / eps of CDFTransform (cdf.py:28-46); (-inf, inf, -inf) without eps */

struct TailArgs {
    const float* bonds; const float* angles; const float* torsions; const float* xfix;   /* contiguous rows [B, n] x 3, [B, keep] */
    const float* desc;                   /* [3 n + keep][DSC]: bonds | angles | torsions rows IN PLACEMENT ORDER (row f n + i = the channel of
                                          * field f that placement i consumes), then the keep fixed channels */
    const int32_t* place;                /* [n][8]: atom, p1, p2, p3, zrow, 0, 0, 0 (placement order) */
    const int32_t* fixed;                /* [n_fixed] atom ids */
    const float* mean; const float* T;   /* whitening mean [3 n_fixed], Tblacken [keep][3 n_fixed]; T == NULL: xfix = coordinates */
    int n, n_fixed, keep, enforce;
    CdfClamp cl;
    float eps, const_ld;                 /* const_ld: n (ln pi + ln 2 pi) - jac_xz, f64 on the host */
    int64_t B; float* x; int64_t ldx; float* dlogp; int accumulate; int32_t* warn_count;
    int lds_per_wave;
    float* o_bonds; float* o_angles; float* o_torsions; float* o_fixed;      /* outputs of the inverse-direction kernel */
    /* KL epilogue of the training tail (icdf_ic2xyz_uni_kernel<NA, true, true>): target energy of a normal target about kl_mean and the
     * [sum (u - dlogp), kept] partial sums of the KL loss, per 64-sample tile */
    const float* kl_mean;                /* [3 (n + n_fixed)] or NULL (0) */
    const float* kl_dl_in;               /* [B] log-det of the flow in front of the tail (read only), or NULL */
    float kl_inv_t, kl_c_in, kl_c_out; int kl_drop;
    float* kl_u; float* kl_dl; float* kl_partial;      /* [B] target energy, [B] total log-det, [tiles][2] */
};


/* y = icdf(v) of one channel (descriptor `ds`: wave-uniform, scalar loads); ld += -log_prob(y) */
__device__ __forceinline__ float icdf_chan(float v, const Desc& dsc, const CdfClamp& cl, float& ld_acc) {
    const float* ds = dsc.f;
    const int kind = __builtin_bit_cast(int, ds[0]);
    if (kind < 0) return v;                                             /* no map on this field */
    v = __builtin_amdgcn_fmed3f(v, cl.lo, cl.hi);
    if (kind == 0) {
        ld_acc += ds[5];
        return __builtin_fmaf(v, ds[2], ds[1]);
    }
    float e = erfinv_fast(__builtin_fmaf(v, ds[3], ds[4]));
    float y = __builtin_fmaf(e, ds[2], ds[1]);
    if (kind == 2) {
        /* close to a bound the result is (bound + a small distance): evaluate the distance itself.  With s = Z v / pdf(alpha) the
         * distance in units of sigma is h = s + c2 s^2 + ... + c5 s^5 (reverted Taylor series of the normal cdf around alpha). */
        const float s_lo = v * ds[7], s_hi = (1.0f - v) * ds[14];
        const bool lo = s_lo < SMAX, hi = s_hi < SMAX;
        if (__builtin_amdgcn_ballot_w64(lo)) {
            float h = __builtin_fmaf(ds[11], s_lo, ds[10]);
            h = __builtin_fmaf(h, s_lo, ds[9]);
            h = __builtin_fmaf(h, s_lo, ds[8]);
            h = __builtin_fmaf(h, s_lo, 1.0f) * s_lo;
            const float yt = __builtin_fmaf(ds[13], h, ds[12]);
            y = lo ? yt : y;
            e = lo ? (yt - ds[1]) * ds[6] : e;
        }
        if (__builtin_amdgcn_ballot_w64(hi)) {
            float h = __builtin_fmaf(ds[18], s_hi, ds[17]);
            h = __builtin_fmaf(h, s_hi, ds[16]);
            h = __builtin_fmaf(h, s_hi, ds[15]);
            h = __builtin_fmaf(h, s_hi, 1.0f) * s_hi;
            const float yt = __builtin_fmaf(-ds[13], h, ds[19]);
            y = hi ? yt : y;
            e = hi ? (yt - ds[1]) * ds[6] : e;
        }
    }
    ld_acc += __builtin_fmaxf(__builtin_fmaf(e, e, ds[5]), cl.ld_min);
    return y;
}

/* cos(2 pi x), sin(2 pi x): exact quadrant reduction + Cephes polynomials (bgk_detmath.h::bgk_sincos2pif), selects instead of
 * the quadrant branches */
#ifndef BGK_TAIL_HWSIN
#define BGK_TAIL_HWSIN 1       /* 1: the hardware sin / cos (argument in revolutions; max abs error 1.24e-7 on [-2, 2], tools/ubench/hw_sincos.hip):
                                * 3 instructions instead of 26, twice per placement.  0: the reproducible polynomial form */
#endif
__device__ __forceinline__ void sincos2pi(float x, float& so, float& co) {
#if BGK_TAIL_HWSIN
    const float fr = __builtin_amdgcn_fractf(x);
    so = __builtin_amdgcn_sinf(fr);
    co = __builtin_amdgcn_cosf(fr);
    return;
#endif
    const float magic = 12582912.0f;
    const float t = __builtin_fmaf(x, 4.0f, magic);
    const float kq = t - magic;
    const unsigned q = __builtin_bit_cast(unsigned, t);                 /* low two bits = quadrant */
    const float f = __builtin_fmaf(kq, -0.25f, x);
    const float th = f * 6.28318530717958647692f;
    const float z = th * th;
    float sp = -1.9515295891e-4f;
    sp = __builtin_fmaf(sp, z, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    sp = sp * z;
    const float sn = __builtin_fmaf(sp, th, th);
    float cp = 2.443315711809948e-5f;
    cp = __builtin_fmaf(cp, z, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    cp = cp * z;
    cp = cp * z;
    const float cs = __builtin_fmaf(z, -0.5f, cp) + 1.0f;
    const bool odd = q & 1u, neg_s = q & 2u, neg_c = (q + 1u) & 2u;      /* q: 0 (s, c)  1 (c, -s)  2 (-s, -c)  3 (-c, s) */
    const float a = odd ? cs : sn, b = odd ? sn : cs;
    so = neg_s ? -a : a;
    co = neg_c ? -b : b;
}

__device__ __forceinline__ float rcp1(float d) {                          /* 1 / d, one Newton step: ~1 ulp */
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float rsq_nr(float q) {                       /* 1 / sqrt(q), one Newton step: ~1 ulp */
    const float r = __builtin_amdgcn_rsqf(q);
    return r * __builtin_fmaf(-0.5f * q, r * r, 1.5f);
}

/* one placement exactly as the reference evaluates it (ic2xyz_deriv, ic_helper.py:372-452: four norms clamped at eps, explicit
 * 3 x 3 determinant) -- only for lanes whose geometry hits an eps clamp, where the clamped vectors are no unit vectors any more
 * and neither |nn| = |v1||nv| nor the closed-form log-det holds.  Returns log|det J|; counts the clamps like the reference warns. */
__device__ __forceinline__ float placement_reference(V3 p1, V3 p2, V3 p3, float dd, float st, float ct, float sa, float ca,
                                                  float eps, int enforce, V3* pos_out, int* warn) {
    auto clampn = [&](V3 v) {
        float nrm = __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        if (nrm < eps) { *warn += 1; if (enforce) nrm = eps; }
        return nrm;
    };
    const V3 v1 = sub(p1, p2), v2 = sub(p1, p3);
    const V3 nv = {v1.y * v2.z - v1.z * v2.y, v1.z * v2.x - v1.x * v2.z, v1.x * v2.y - v1.y * v2.x};
    const V3 nn = {v1.y * nv.z - v1.z * nv.y, v1.z * nv.x - v1.x * nv.z, v1.x * nv.y - v1.y * nv.x};
    const float nvn = clampn(nv), nnn = clampn(nn);
    const V3 nh = {nv.x / nvn, nv.y / nvn, nv.z / nvn}, nnh = {nn.x / nnn, nn.y / nnn, nn.z / nnn};
    const V3 v3 = {nh.x * (-st) + nnh.x * ct, nh.y * (-st) + nnh.y * ct, nh.z * (-st) + nnh.z * ct};
    const float v3n = clampn(v3);
    const V3 v3h = {v3.x / v3n, v3.y / v3n, v3.z / v3n};
    const float v1n = clampn(v1);
    const V3 v1h = {v1.x / v1n, v1.y / v1n, v1.z / v1n};
    *pos_out = {p1.x + v3h.x * dd * sa - v1h.x * dd * ca, p1.y + v3h.y * dd * sa - v1h.y * dd * ca, p1.z + v3h.z * dd * sa - v1h.z * dd * ca};
    const V3 Jd = {v3h.x * sa - v1h.x * ca, v3h.y * sa - v1h.y * ca, v3h.z * sa - v1h.z * ca};
    const V3 Ja = {v3h.x * dd * ca + v1h.x * dd * sa, v3h.y * dd * ca + v1h.y * dd * sa, v3h.z * dd * ca + v1h.z * dd * sa};
    const V3 Jt3 = {nh.x * (-ct) + nnh.x * (-st), nh.y * (-ct) + nnh.y * (-st), nh.z * (-ct) + nnh.z * (-st)};
    const float jt1 = dd * sa, h3 = v3h.x * Jt3.x + v3h.y * Jt3.y + v3h.z * Jt3.z, inv = 1.0f / v3n;
    const V3 Jt = {jt1 * inv * (Jt3.x - v3h.x * h3), jt1 * inv * (Jt3.y - v3h.y * h3), jt1 * inv * (Jt3.z - v3h.z * h3)};
    const V3 R0 = {Jd.x, Ja.x, Jt.x}, R1 = {Jd.y, Ja.y, Jt.y}, R2 = {Jd.z, Ja.z, Jt.z};
    const V3 c01 = {R0.y * R1.z - R0.z * R1.y, R0.z * R1.x - R0.x * R1.z, R0.x * R1.y - R0.y * R1.x};
    return bgk_logf(__builtin_fabsf(c01.x * R2.x + c01.y * R2.y + c01.z * R2.z));
}

struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };

/* copy the wave's [rows][w] block of a contiguous field into the staging slab (linear: element e of the block -> s_in[e]) */
__device__ __forceinline__ void stage_field(float* s_in, const float* __restrict__ src, int w, int rows, int lane) {
    const int total = rows * w;
    for (int k = 0; k < w; ++k) {
        const int e = k * 64 + lane;
#if (BGK_TAIL_ABL & 2)
        s_in[e] = 0.25f + 0.001f * (float)(e & 255);
#else
        s_in[e] = e < total ? src[e] : 0.5f;
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int NA>
__global__ __launch_bounds__(TW * 64) void icdf_ic2xyz_reg_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * TW + wave;
    if (tile >= ((a.B + 63) >> 6)) return;
    float* s_in = smem + (size_t)wave * a.lds_per_wave;
    const int64_t b0 = tile * 64;
    const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    const int n = a.n, keep = a.keep, nf3 = 3 * a.n_fixed;
    const cf32_t desc = (cf32_t)a.desc;
    const ci32_t recs = (ci32_t)a.place;
    const CdfClamp cl = a.cl;

    float px[NA], py[NA], pz[NA];            /* every slot that is read (atoms 0 .. n_atoms - 1) is written first */
    float acc = a.const_ld;

    /* ---- fixed block: icdf maps, blackening, into the fixed atoms' slots ---- */
    {
        stage_field(s_in, a.xfix + b0 * keep, keep, rows, lane);
        float fz[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < keep) {
                const Desc d = load_desc(desc, 3 * n + k);
                fz[k] = icdf_chan(s_in[lane * keep + k], d, cl, acc);
            }
        const cf32_t T = (cf32_t)a.T, mean = (cf32_t)a.mean;
        for (int fa = 0; fa < a.n_fixed; ++fa) {
            float cx, cy, cz;
            if (T) {
                cx = mean[3 * fa]; cy = mean[3 * fa + 1]; cz = mean[3 * fa + 2];
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < keep) {
                        cx = __builtin_fmaf(fz[k], T[k * nf3 + 3 * fa], cx);
                        cy = __builtin_fmaf(fz[k], T[k * nf3 + 3 * fa + 1], cy);
                        cz = __builtin_fmaf(fz[k], T[k * nf3 + 3 * fa + 2], cz);
                    }
            } else {
                cx = 0.0f; cy = 0.0f; cz = 0.0f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {                         /* keep == 3 n_fixed: coordinate k = 3 fa + c */
                    cx = (k == 3 * fa) ? fz[k] : cx;
                    cy = (k == 3 * fa + 1) ? fz[k] : cy;
                    cz = (k == 3 * fa + 2) ? fz[k] : cz;
                }
            }
            const int at = ((ci32_t)a.fixed)[fa];
            px[at] = cx; py[at] = cy; pz[at] = cz;
        }
        __builtin_amdgcn_wave_barrier();
    }

    /* Every iteration of the loops below needs one placement record and one channel descriptor, both wave-uniform and both
     * addressed by the loop counter alone (the descriptor rows are stored in placement order): their scalar loads are issued
     * together with the lane's LDS read at the top of the iteration -- one wait per iteration instead of a chain of dependent ones
     * (the record of iteration i + 1 is requested one iteration ahead, because the LDS address needs its Z row). */
    /* ---- bonds: d parked in px[atom] ---- */
    stage_field(s_in, a.bonds + b0 * n, n, rows, lane);
    {
        Rec r = load_rec(recs, 0);
        for (int i = 0; i < n; ++i) {
            const Rec rn = load_rec(recs, i + 1 < n ? i + 1 : i);
            const Desc d = load_desc(desc, i);
            const float v = s_in[lane * n + r.zr];
            px[r.at] = icdf_chan(v, d, cl, acc);
            r = rn;
        }
    }
    __builtin_amdgcn_wave_barrier();
    /* ---- angles: (d sin a, d cos a) parked in (px, py); log|det| of the placement = ln|d (d sin a)| ---- */
    stage_field(s_in, a.angles + b0 * n, n, rows, lane);
    {
        Rec r = load_rec(recs, 0);
        for (int i = 0; i < n; ++i) {
            const Rec rn = load_rec(recs, i + 1 < n ? i + 1 : i);
            const Desc d = load_desc(desc, n + i);
            const float v = s_in[lane * n + r.zr];
            const float an = icdf_chan(v, d, cl, acc);
            float sa, ca;
            sincos2pi(0.5f * an, sa, ca);                                /* sin / cos (pi a) */
            const float dd = px[r.at];
            const float dsa = dd * sa;
            acc = __builtin_fmaf(LN2_F, __builtin_amdgcn_logf(__builtin_fabsf(dd * dsa)), acc);
            px[r.at] = dsa; py[r.at] = dd * ca;
            r = rn;
        }
    }
    __builtin_amdgcn_wave_barrier();
    /* ---- torsions: parked in pz ---- */
    stage_field(s_in, a.torsions + b0 * n, n, rows, lane);
    {
        Rec r = load_rec(recs, 0);
        for (int i = 0; i < n; ++i) {
            const Rec rn = load_rec(recs, i + 1 < n ? i + 1 : i);
            const Desc d = load_desc(desc, 2 * n + i);
            const float v = s_in[lane * n + r.zr];
            pz[r.at] = icdf_chan(v, d, cl, acc);
            r = rn;
        }
    }

    /* ---- sequential placement ---- */
    int warn = 0;
    const float eps2 = a.eps * a.eps;
    Rec r = load_rec(recs, 0);
    for (int i = 0; i < ((BGK_TAIL_ABL & 4) ? 1 : n); ++i) {
        const Rec rn = load_rec(recs, i + 1 < n ? i + 1 : i);
        const int at = r.at, i1 = r.i1, i2 = r.i2, i3 = r.i3;
        r = rn;
        const V3 p1 = {px[i1], py[i1], pz[i1]}, p2 = {px[i2], py[i2], pz[i2]}, p3 = {px[i3], py[i3], pz[i3]};
        const float dsa = px[at], dca = py[at], tn = pz[at];
        float st, ct;
        sincos2pi(tn, st, ct);
        st = -st; ct = -ct;                                              /* sin / cos (2 pi t - pi) */
        const V3 v1 = sub(p1, p2), v2 = sub(p1, p3);
        const V3 nv = cross(v1, v2), nn = cross(v1, nv);
        const float q1 = dot(v1, v1), qn = dot(nv, nv);
        const bool bad = (q1 < eps2) || (qn < eps2) || (q1 * qn < eps2);
        const float r1 = rsq_nr(q1), rn_ = rsq_nr(qn);
        const float ka = -st * rn_ * dsa, kb = ct * (r1 * rn_) * dsa, kc = dca * r1;
        V3 pos = {__builtin_fmaf(ka, nv.x, __builtin_fmaf(kb, nn.x, __builtin_fmaf(-kc, v1.x, p1.x))),
                  __builtin_fmaf(ka, nv.y, __builtin_fmaf(kb, nn.y, __builtin_fmaf(-kc, v1.y, p1.y))),
                  __builtin_fmaf(ka, nv.z, __builtin_fmaf(kb, nn.z, __builtin_fmaf(-kc, v1.z, p1.z)))};
        if (__builtin_amdgcn_ballot_w64(bad)) {                          /* rare: degenerate geometry */
            if (bad) {
                const float d = __builtin_sqrtf(dsa * dsa + dca * dca);
                const float sa = dsa / d, ca = dca / d;
                V3 pr;
                const float ld_ref = placement_reference(p1, p2, p3, d, st, ct, sa, ca, a.eps, a.enforce, &pr, &warn);
                acc += ld_ref - LN2_F * __builtin_amdgcn_logf(__builtin_fabsf(d * dsa));
                pos = pr;
            }
        }
        px[at] = pos.x; py[at] = pos.y; pz[at] = pos.z;
    }

    if (lane < rows) {
        const int64_t b = b0 + lane;
        if (a.accumulate) a.dlogp[b] += acc; else a.dlogp[b] = acc;
        float* row = a.x + b * a.ldx;
        const int n_atoms = n + a.n_fixed;
#if (BGK_TAIL_ABL & 1)
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int k = 0; k < NA; ++k) if (k < n_atoms) { sx += px[k]; sy += py[k]; sz += pz[k]; }
        *reinterpret_cast<F3*>(row) = F3{sx, sy, sz};
#else
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (k < n_atoms) *reinterpret_cast<F3*>(row + 3 * k) = F3{px[k], py[k], pz[k]};
#endif
    }
    if (warn && a.warn_count) atomicAdd(a.warn_count, warn);
}


/* ---- field-uniform marginals (what the builder installs: one distribution object per IC field) -------------------------------
 * When every channel of a field shares one descriptor, the icdf maps need no per-channel table at all: they run ELEMENTWISE over
 * the field tiles in their memory layout (element e = k 64 + lane of the [64][n] tile), the descriptor sits in SGPRs for the whole
 * pass, there is no per-iteration scalar load, no LDS read on the critical path and all n elements of a lane are independent
 * (instruction-level parallelism across k).  Data movement:
 *   1. the four field tiles are copied global -> LDS by the DMA path (global_load_lds: no staging registers, 16 instructions per
 *      tile set, all in flight from the first cycle; the tiles are contiguous, so the copy is linear);
 *   2. fixed pass: icdf in place, then every lane reads its own row and blackens it into the fixed atoms' registers;
 *   3. bonds / angles / torsions pass: the three values of element e -> (d sin a, d cos a, t, sum of the log-dets incl. the
 *      placement's ln|d (d sin a)|), written back in place (+ the log-det slab in the fixed tile's place);
 *   4. every lane sums its row of the log-det slab, then walks the placements: (d sin a, d cos a, t) of placement i + 1 are
 *      requested from LDS one iteration ahead;
 *   5. the finished rows go to LDS ([64][3 n_atoms], the tile's memory image) and leave as 16-byte coalesced stores. */
#include "bgk_dma.h"

#ifndef BGK_TAIL_TS
#define BGK_TAIL_TS 0                 /* profiling build (tools/r06_tail_ts.py): lane 0 stamps s_memtime at the sampling tail's phase boundaries and writes the stamps over the tile's first output row */
#endif
#if BGK_TAIL_TS
#define TAIL_TS(k) do { tts_[k] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TAIL_TS(k) do { } while (0)
#endif

template <int NA, bool EMIT = false, bool KL = false>
__global__ __launch_bounds__(TW * 64) void icdf_ic2xyz_uni_kernel(TailArgs a) {
#if BGK_TAIL_TS
    unsigned tts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * TW + wave;
    if (tile >= ((a.B + 63) >> 6)) return;
    const int n = a.n, keep = a.keep, nf3 = 3 * a.n_fixed, n_atoms = n + a.n_fixed;
    const int W = n > keep ? n : keep;                /* slab width */
    float* R0 = smem + (size_t)wave * a.lds_per_wave; /* bonds   -> d sin a */
    float* R1 = R0 + 64 * W;                          /* angles  -> d cos a */
    float* R2 = R1 + 64 * W;                          /* torsions */
    float* R3 = R2 + 64 * W;                          /* fixed   -> log-det slab */
    const int64_t b0 = tile * 64;
    const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    const cf32_t desc = (cf32_t)a.desc;               /* [4][DSC]: bonds, angles, torsions, fixed */
    const ci32_t recs = (ci32_t)a.place;
    const CdfClamp cl = a.cl;

    TAIL_TS(0);
    dma_tile(R3, a.xfix + b0 * keep, keep, rows, lane);
    dma_tile(R0, a.bonds + b0 * n, n, rows, lane);
    dma_tile(R1, a.angles + b0 * n, n, rows, lane);
    dma_tile(R2, a.torsions + b0 * n, n, rows, lane);

    float px[NA], py[NA], pz[NA];
    float acc = a.const_ld;

    /* ---- fixed pass ---- */
    {
        wait_vmcnt((n / 4 + n % 4) * 3);
        TAIL_TS(1);             /* the fixed tile has landed; the 3 (n / 4 + n % 4) later requests may still fly */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const Desc df = load_desc(desc, 3);
        for (int k = 0; k < keep; ++k) {                 /* (the log-dets are re-derived per row below: -log_prob = e^2 + cst) */
            const int e = k * 64 + lane;
            float dummy = 0.0f;
            R3[e] = icdf_chan(R3[e], df, cl, dummy);
            if (EMIT && e < rows * keep) a.o_fixed[b0 * keep + e] = R3[e];      /* training: the mapped field for the backward kernels */
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float fz[KMAX];
        const int kind_f = __builtin_bit_cast(int, df.f[0]);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < keep) {
                const float z = R3[lane * keep + k];
                fz[k] = z;
                if (kind_f > 0) {                     /* normal / truncated normal: e = (y - mu) / (sigma sqrt2) */
                    const float ee = (z - df.f[1]) * df.f[6];
                    acc += __builtin_fmaxf(__builtin_fmaf(ee, ee, df.f[5]), cl.ld_min);
                } else if (kind_f == 0) {
                    acc += df.f[5];
                }
            }
        const cf32_t T = (cf32_t)a.T, mean = (cf32_t)a.mean;
        for (int fa = 0; fa < a.n_fixed; ++fa) {
            float cx, cy, cz;
            if (T) {
                cx = mean[3 * fa]; cy = mean[3 * fa + 1]; cz = mean[3 * fa + 2];
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < keep) {
                        cx = __builtin_fmaf(fz[k], T[k * nf3 + 3 * fa], cx);
                        cy = __builtin_fmaf(fz[k], T[k * nf3 + 3 * fa + 1], cy);
                        cz = __builtin_fmaf(fz[k], T[k * nf3 + 3 * fa + 2], cz);
                    }
            } else {
                cx = 0.0f; cy = 0.0f; cz = 0.0f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    cx = (k == 3 * fa) ? fz[k] : cx;
                    cy = (k == 3 * fa + 1) ? fz[k] : cy;
                    cz = (k == 3 * fa + 2) ? fz[k] : cz;
                }
            }
            const int at = ((ci32_t)a.fixed)[fa];
            px[at] = cx; py[at] = cy; pz[at] = cz;
        }
        __builtin_amdgcn_wave_barrier();
    }

    TAIL_TS(2);
    /* ---- bonds / angles / torsions, elementwise in the tiles' own layout ---- */
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const Desc db = load_desc(desc, 0), da = load_desc(desc, 1), dt = load_desc(desc, 2);
#pragma unroll 2
        for (int k = 0; k < n; ++k) {
            const int e = k * 64 + lane;
            float ld = 0.0f;
            const float d = icdf_chan(R0[e], db, cl, ld);
            const float an = icdf_chan(R1[e], da, cl, ld);
            const float tn = icdf_chan(R2[e], dt, cl, ld);
            if (EMIT && e < rows * n) { a.o_bonds[b0 * n + e] = d; a.o_angles[b0 * n + e] = an; a.o_torsions[b0 * n + e] = tn; }
            float sa, ca;
            sincos2pi(0.5f * an, sa, ca);             /* sin / cos (pi a) */
            const float dsa = d * sa;
            ld = __builtin_fmaf(LN2_F, __builtin_amdgcn_logf(__builtin_fabsf(d * dsa)), ld);   /* ln|det| of the placement */
            R0[e] = dsa; R1[e] = d * ca; R2[e] = tn; R3[e] = ld;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < n; ++c) acc += R3[lane * n + c];
    }

    TAIL_TS(3);
    /* ---- sequential placement: the inputs of placement i + 1 are requested from LDS one iteration ahead ---- */
    int warn = 0;
    const float eps2 = a.eps * a.eps;
    Rec r = load_rec(recs, 0);
    float dsa = R0[lane * n + r.zr], dca = R1[lane * n + r.zr], tn = R2[lane * n + r.zr];
    for (int i = 0; i < n; ++i) {
        const Rec rn = load_rec(recs, i + 1 < n ? i + 1 : i);
        const float dsa_n = R0[lane * n + rn.zr], dca_n = R1[lane * n + rn.zr], tn_n = R2[lane * n + rn.zr];
        const int at = r.at, i1 = r.i1, i2 = r.i2, i3 = r.i3;
        const V3 p1 = {px[i1], py[i1], pz[i1]}, p2 = {px[i2], py[i2], pz[i2]}, p3 = {px[i3], py[i3], pz[i3]};
        float st, ct;
        sincos2pi(tn, st, ct);
        st = -st; ct = -ct;                                              /* sin / cos (2 pi t - pi) */
        const V3 v1 = sub(p1, p2), v2 = sub(p1, p3);
        const V3 nv = cross(v1, v2), nn = cross(v1, nv);
        const float q1 = dot(v1, v1), qn = dot(nv, nv);
        const bool bad = (q1 < eps2) || (qn < eps2) || (q1 * qn < eps2);
        const float r1 = rsq_nr(q1), rn_ = rsq_nr(qn);
        const float ka = -st * rn_ * dsa, kb = ct * (r1 * rn_) * dsa, kc = dca * r1;
        V3 pos = {__builtin_fmaf(ka, nv.x, __builtin_fmaf(kb, nn.x, __builtin_fmaf(-kc, v1.x, p1.x))),
                  __builtin_fmaf(ka, nv.y, __builtin_fmaf(kb, nn.y, __builtin_fmaf(-kc, v1.y, p1.y))),
                  __builtin_fmaf(ka, nv.z, __builtin_fmaf(kb, nn.z, __builtin_fmaf(-kc, v1.z, p1.z)))};
        if (__builtin_amdgcn_ballot_w64(bad)) {                          /* rare: degenerate geometry */
            if (bad) {
                const float d = __builtin_sqrtf(dsa * dsa + dca * dca);
                const float sa = dsa / d, ca = dca / d;
                V3 pr;
                const float ld_ref = placement_reference(p1, p2, p3, d, st, ct, sa, ca, a.eps, a.enforce, &pr, &warn);
                acc += ld_ref - LN2_F * __builtin_amdgcn_logf(__builtin_fabsf(d * dsa));
                pos = pr;
            }
        }
        px[at] = pos.x; py[at] = pos.y; pz[at] = pos.z;
        r = rn; dsa = dsa_n; dca = dca_n; tn = tn_n;
    }

    TAIL_TS(4);
    if (!KL && lane < rows) {
        const int64_t b = b0 + lane;
        if (a.accumulate) a.dlogp[b] += acc; else a.dlogp[b] = acc;
    }
    if constexpr (KL) {
        /* the KL integrand of the lane's sample while its coordinates are in registers (BoltzmannGenerator.kldiv, bg.py:140-147, for a
         * normal target: u = (|x - mean|^2 / 2 + c_in) / T + c_out, distribution/normal.py:61-72) and the tile's share of
         * [sum (u - dlogp), samples kept]: no second pass over x, no per-sample loss tensor.  Lane sums in a fixed shuffle tree. */
        const cf32_t km = (cf32_t)a.kl_mean;
        float e = 0.0f;
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (k < n_atoms) {
                const float dx = px[k] - (km ? km[3 * k] : 0.0f), dy = py[k] - (km ? km[3 * k + 1] : 0.0f), dz = pz[k] - (km ? km[3 * k + 2] : 0.0f);
                e += dx * dx; e += dy * dy; e += dz * dz;
            }
        const float u = (0.5f * e + a.kl_c_in) * a.kl_inv_t + a.kl_c_out;
        const bool valid = lane < rows;
        const float dl_tot = acc + ((valid && a.kl_dl_in) ? a.kl_dl_in[b0 + lane] : 0.0f);
        const float loss = u - dl_tot;
        const bool ok = valid && (!a.kl_drop || __builtin_isfinite(loss));
        if (valid) { a.kl_u[b0 + lane] = u; a.kl_dl[b0 + lane] = dl_tot; }
        float s_l = ok ? loss : 0.0f, s_c = ok ? 1.0f : 0.0f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s_l += __shfl_xor(s_l, off); s_c += __shfl_xor(s_c, off); }
        if (lane == 0) { a.kl_partial[2 * tile] = s_l; a.kl_partial[2 * tile + 1] = s_c; }
    }
    /* ---- rows -> LDS ([64][3 n_atoms] = the tile's memory image) -> 16-byte coalesced stores ---- */
    __builtin_amdgcn_wave_barrier();
    {
        const int ld_row = 3 * n_atoms;
        float* orow = R0 + lane * ld_row;
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (k < n_atoms) { orow[3 * k] = px[k]; orow[3 * k + 1] = py[k]; orow[3 * k + 2] = pz[k]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int total4 = (rows * ld_row) >> 2;      /* float4s of the tile image (64 ld_row is a multiple of 4) */
        const float4* s4 = reinterpret_cast<const float4*>(R0);
        float4* g4 = reinterpret_cast<float4*>(a.x + b0 * a.ldx);
        for (int q = lane; q < total4; q += 64) g4[q] = s4[q];
        const int tail0 = total4 << 2;                /* partial last tile with rows * ld_row not a multiple of 4 */
        for (int q = tail0 + lane; q < rows * ld_row; q += 64) a.x[b0 * a.ldx + q] = R0[q];
    }
    if (warn && a.warn_count) atomicAdd(a.warn_count, warn);
#if BGK_TAIL_TS
    TAIL_TS(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TAIL_TS(6);
    if (lane == 0)
        for (int q = 0; q < 8; ++q) reinterpret_cast<unsigned*>(a.x + b0 * a.ldx)[q] = tts_[q];
#endif
}


/* ==== the inverse (NLL) direction of the tail: xyz -> IC + whitening + the four cdf domain maps in one launch ===================
 * RelativeInternalCoordinateTransformation._forward / MixedCoordinateTransformation._forward (nn/flow/crd_transform/ic.py:386-433,
 * 838-860; dist / angle / torsion of ic_helper.py:148-293; whitening pca.py:74-83) followed by CDFTransform._forward x 4
 * (nn/flow/cdf.py:28-35) -- five launches of the block path (bgk_ic_xyz2ic + 4 x bgk_cdf_transform) and a [B, 60] round trip.
 * Same layout ideas as the sampling direction: the x tile arrives by DMA, every lane lifts its row into registers (px / py / pz,
 * uniform register indexing by the Z-matrix rows), the Z rows are independent of each other (instruction-level parallelism across
 * rows), results go to LDS row-major = the memory image of the contiguous [B, n] output tiles and leave as coalesced stores.
 * Arithmetic: the angle is atan2(|r12 x r32|, r12 . r32) instead of acos of the normalised dot product (same value, but accurate
 * for small and near-straight angles, where acos amplifies the rounding of the cosine by 1 / sin a); log|det J| of a row in closed
 * form -(2 ln d + ln sin a) (what the explicit 3 x 3 determinant of the reference evaluates to away from its eps clamps; rows
 * that hit a clamp are recomputed with the reference's explicit arithmetic); erf (N. Juffa's single-precision form, < 1 ulp) and
 * atan (Cephes) as branch-free polynomials.  Field-uniform marginals only (desc4); otherwise the block path runs. */

/* atan2(y, x) in (-pi, pi]: atan of min / max on [0, 1] (Cephes atanf: reduction at tan(pi / 8), odd polynomial), octant fix-ups */
__device__ __forceinline__ float atan2_fast(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    const float q = mn * rcp1(mx);                                      /* in [0, 1]; 0 / 0 -> handled by the callers' clamps */
    const bool big = q > 0.41421356237309503f;
    const float xr = big ? (q - 1.0f) * rcp1(q + 1.0f) : q;
    const float z = xr * xr;
    float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    float r = __builtin_fmaf(p * z, xr, xr);
    r = big ? r + 0.78539816339744831f : r;
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.0f ? 3.14159265358979324f - r : r;
    return __builtin_copysignf(r, y);
}

/* u = cdf(v) of one field-uniform channel (CDFTransform._forward: clamp to [eps, 1 - eps], log-det = log_prob(v) >= -1 / eps) */
__device__ __forceinline__ float cdf_chan(float v, const Desc& dsc, float inv_xs, const CdfClamp& cl, float& ld_acc) {
    const float* ds = dsc.f;
    const int kind = __builtin_bit_cast(int, ds[0]);
    if (kind < 0) return v;
    float u, ld;
    if (kind == 0) {
        u = __builtin_amdgcn_fmed3f((v - ds[1]) * inv_xs, 0.0f, 1.0f);   /* inv_xs = 1 / (high - low) for a uniform channel */
        ld = -ds[5];
    } else {
        const float e = (v - ds[1]) * ds[6];                             /* (v - mu) / (sigma sqrt2) */
        u = (erf_fast(e) - ds[4]) * inv_xs;                              /* (Phi - cdf_lower) / Z with xs = 2 Z, xo = 2 cdf_lower - 1 */
        ld = -__builtin_fmaf(e, e, ds[5]);
    }
    ld_acc += __builtin_fmaxf(ld, cl.ld_min);
    return __builtin_amdgcn_fmed3f(u, cl.lo, cl.hi);
}

/* one Z row exactly like the reference (dist_deriv / angle_deriv / torsion_deriv with their eps clamps and the explicit 3 x 3
 * determinant, ic_helper.py:148-293) -- only for lanes that hit a clamp */
__device__ __forceinline__ void zrow_reference(V3 x1, V3 x2, V3 x3, V3 x4, float eps, int enforce, float* d_out, float* a_out, float* t_out,
                                               float* ld_out, int* warn) {
    auto cl = [&](float v) { if (v < eps) { *warn += 1; if (enforce) v = eps; } return v; };
    auto nrm = [](V3 v) { return __builtin_sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); };
    const V3 r = sub(x2, x1);
    const float rn = cl(nrm(r));
    const V3 Jb = {-r.x / rn, -r.y / rn, -r.z / rn};
    const V3 r12 = sub(x1, x2), r32 = sub(x3, x2);
    const float n12 = cl(nrm(r12)), n32 = cl(nrm(r32));
    const V3 u12 = {r12.x / n12, r12.y / n12, r12.z / n12}, u32 = {r32.x / n32, r32.y / n32, r32.z / n32};
    float cosa = u12.x * u32.x + u12.y * u32.y + u12.z * u32.z;
    const float u12v[3] = {u12.x, u12.y, u12.z}, u32v[3] = {u32.x, u32.y, u32.z};
    float Jav[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sacc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) sacc += u32v[k] * (((k == c ? 1.0f : 0.0f) - u12v[k] * u12v[c]) / n12);
        Jav[c] = sacc;
    }
    if (enforce) cosa = __builtin_fminf(__builtin_fmaxf(cosa, -1.0f + eps), 1.0f - eps);
    const float ang = acosf(cosa), sq = __builtin_sqrtf(1.0f - cosa * cosa);
    const V3 Ja = {-Jav[0] / sq, -Jav[1] / sq, -Jav[2] / sq};
    const V3 b0 = r12, b1 = r32, b2 = sub(x4, x3);
    const float b1n = cl(nrm(b1));
    const V3 u = {b1.x / b1n, b1.y / b1n, b1.z / b1n};
    const float b0u = b0.x * u.x + b0.y * u.y + b0.z * u.z, b2u = b2.x * u.x + b2.y * u.y + b2.z * u.z;
    const V3 v = {b0.x - b0u * u.x, b0.y - b0u * u.y, b0.z - b0u * u.z}, w = {b2.x - b2u * u.x, b2.y - b2u * u.y, b2.z - b2u * u.z};
    const float xx = v.x * w.x + v.y * w.y + v.z * w.z;
    const V3 uxv = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
    const float yy = uxv.x * w.x + uxv.y * w.y + uxv.z * w.z;
    const float tor = atan2f(yy, xx);
    const float q = cl(xx * xx + yy * yy);
    const float dadx = -yy / q, dady = xx / q;
    const V3 wxu = {w.y * u.z - w.z * u.y, w.z * u.x - w.x * u.z, w.x * u.y - w.y * u.x};
    const V3 g = {dadx * w.x + dady * wxu.x, dadx * w.y + dady * wxu.y, dadx * w.z + dady * wxu.z};
    const float gu = g.x * u.x + g.y * u.y + g.z * u.z;
    const V3 Jt = {g.x - gu * u.x, g.y - gu * u.y, g.z - gu * u.z};
    const V3 c01 = {Jb.y * Ja.z - Jb.z * Ja.y, Jb.z * Ja.x - Jb.x * Ja.z, Jb.x * Ja.y - Jb.y * Ja.x};
    *ld_out = bgk_logf(__builtin_fabsf(c01.x * Jt.x + c01.y * Jt.y + c01.z * Jt.z));
    *d_out = rn; *a_out = ang; *t_out = tor;
}

template <int NA>
__global__ __launch_bounds__(TW * 64) void xyz2ic_cdf_uni_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * TW + wave;
    if (tile >= ((a.B + 63) >> 6)) return;
    const int n = a.n, keep = a.keep, nf3 = 3 * a.n_fixed, n_atoms = n + a.n_fixed, ldr = 3 * n_atoms;
    float* S = smem + (size_t)wave * a.lds_per_wave;  /* x tile [64][3 n_atoms]; afterwards bonds | angles | torsions | fixed tiles */
    float* R0 = S;
    float* R1 = R0 + 64 * n;
    float* R2 = R1 + 64 * n;
    float* R3 = R2 + 64 * n;
    const int64_t b0 = tile * 64;
    const int rows = (int)((a.B - b0) < 64 ? (a.B - b0) : 64);
    const cf32_t desc = (cf32_t)a.desc;               /* [4][DSC]: bonds, angles, torsions, fixed */
    const ci32_t recs = (ci32_t)a.place;              /* Z rows: [n][8] = (a, b, c, d, 0, 0, 0, 0) */
    const CdfClamp cl = a.cl;

    dma_tile(S, a.x + b0 * ldr, ldr, rows, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float px[NA], py[NA], pz[NA];
    {
        const float* row = S + lane * ldr;
#pragma unroll
        for (int k = 0; k < NA; ++k)
            if (k < n_atoms) { px[k] = row[3 * k]; py[k] = row[3 * k + 1]; pz[k] = row[3 * k + 2]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                  /* every lane holds its row: the tile's place in LDS is free */
    float acc = a.const_ld;                           /* -n (ln pi + ln 2 pi) + jac_xz */

    /* ---- fixed atoms: whitening + cdf map ---- */
    {
        const Desc df = load_desc(desc, 3);
        const float inv_f = rcp1(__builtin_bit_cast(int, df.f[0]) == 0 ? df.f[2] : df.f[3]);
        const cf32_t T = (cf32_t)a.T, mean = (cf32_t)a.mean;
        float fz[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) fz[k] = 0.0f;
        for (int fa = 0; fa < a.n_fixed; ++fa) {
            const int at = ((ci32_t)a.fixed)[fa];
            const float cx = px[at], cy = py[at], cz = pz[at];
            if (T) {
                const float dx = cx - mean[3 * fa], dy = cy - mean[3 * fa + 1], dz = cz - mean[3 * fa + 2];
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < keep) {
                        fz[k] = __builtin_fmaf(dx, T[(3 * fa) * keep + k], fz[k]);
                        fz[k] = __builtin_fmaf(dy, T[(3 * fa + 1) * keep + k], fz[k]);
                        fz[k] = __builtin_fmaf(dz, T[(3 * fa + 2) * keep + k], fz[k]);
                    }
            } else {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    fz[k] = (k == 3 * fa) ? cx : fz[k];
                    fz[k] = (k == 3 * fa + 1) ? cy : fz[k];
                    fz[k] = (k == 3 * fa + 2) ? cz : fz[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < keep) R3[lane * keep + k] = cdf_chan(fz[k], df, inv_f, cl, acc);
    }

    /* ---- Z rows (independent of each other) ---- */
    int warn = 0;
    const float eps2 = a.eps * a.eps;
    const Desc db = load_desc(desc, 0), da = load_desc(desc, 1), dt = load_desc(desc, 2);
    const float inv_b = rcp1(__builtin_bit_cast(int, db.f[0]) == 0 ? db.f[2] : db.f[3]);
    const float inv_a = rcp1(__builtin_bit_cast(int, da.f[0]) == 0 ? da.f[2] : da.f[3]);
    const float inv_t = rcp1(__builtin_bit_cast(int, dt.f[0]) == 0 ? dt.f[2] : dt.f[3]);
    Rec r = load_rec(recs, 0);
    for (int i = 0; i < n; ++i) {
        const Rec rn = load_rec(recs, i + 1 < n ? i + 1 : i);
        const V3 x1 = {px[r.at], py[r.at], pz[r.at]}, x2 = {px[r.i1], py[r.i1], pz[r.i1]};
        const V3 x3 = {px[r.i2], py[r.i2], pz[r.i2]}, x4 = {px[r.i3], py[r.i3], pz[r.i3]};
        r = rn;
        const V3 r12 = sub(x1, x2), r32 = sub(x3, x2), b2 = sub(x4, x3);
        const float q12 = dot(r12, r12), q32 = dot(r32, r32);
        const float i12 = rsq_nr(q12), i32 = rsq_nr(q32);
        float d = q12 * i12;                                             /* |x2 - x1| */
        const V3 cr = cross(r12, r32);
        const float qc = dot(cr, cr);
        const float sn = qc * rsq_nr(qc) * (i12 * i32);                  /* sin a = |r12 x r32| / (|r12| |r32|) */
        const float cs = dot(r12, r32) * (i12 * i32);                    /* cos a */
        float ang = atan2_fast(sn, cs);
        /* torsion (ic_helper.py:213-293): u = r32 / |r32|, v = r12 - (r12.u) u, w = b2 - (b2.u) u, t = atan2((u x v).w, v.w) */
        const V3 u = {r32.x * i32, r32.y * i32, r32.z * i32};
        const float b0u = dot(r12, u), b2u = dot(b2, u);
        const V3 v = {__builtin_fmaf(-b0u, u.x, r12.x), __builtin_fmaf(-b0u, u.y, r12.y), __builtin_fmaf(-b0u, u.z, r12.z)};
        const V3 w = {__builtin_fmaf(-b2u, u.x, b2.x), __builtin_fmaf(-b2u, u.y, b2.y), __builtin_fmaf(-b2u, u.z, b2.z)};
        const float xx = dot(v, w), yy = dot(cross(u, v), w);
        float tor = atan2_fast(yy, xx);
        float ld = -LN2_F * __builtin_amdgcn_logf(q12 * sn);             /* -(2 ln d + ln sin a) */
        const bool bad = (q12 < eps2) || (q32 < eps2) || (__builtin_fabsf(cs) > 1.0f - a.eps) || (xx * xx + yy * yy < a.eps);
        if (__builtin_amdgcn_ballot_w64(bad)) {                          /* rare: degenerate geometry, the reference's clamps */
            if (bad) zrow_reference(x1, x2, x3, x4, a.eps, a.enforce, &d, &ang, &tor, &ld, &warn);
        }
        acc += ld;
        ang = ang * 0.318309886183790672f;                               /* / pi */
        tor = __builtin_fmaf(tor, 0.159154943091895336f, 0.5f);          /* (t + pi) / (2 pi) */
        R0[lane * n + i] = cdf_chan(d, db, inv_b, cl, acc);
        R1[lane * n + i] = cdf_chan(ang, da, inv_a, cl, acc);
        R2[lane * n + i] = cdf_chan(tor, dt, inv_t, cl, acc);
    }
    if (lane < rows) {
        const int64_t b = b0 + lane;
        if (a.accumulate) a.dlogp[b] += acc; else a.dlogp[b] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    /* ---- the four tiles leave as their memory images ---- */
    {
        float* outs[4] = {a.o_bonds + b0 * n, a.o_angles + b0 * n, a.o_torsions + b0 * n, a.o_fixed + b0 * keep};
        const float* srcs[4] = {R0, R1, R2, R3};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int w = f < 3 ? n : keep;
            const int total = rows * w, total4 = total >> 2;
            const float4* s4 = reinterpret_cast<const float4*>(srcs[f]);
            float4* g4 = reinterpret_cast<float4*>(outs[f]);
            for (int q = lane; q < total4; q += 64) g4[q] = s4[q];
            for (int q = (total4 << 2) + lane; q < total; q += 64) outs[f][q] = srcs[f][q];
        }
    }
    if (warn && a.warn_count) atomicAdd(a.warn_count, warn);
}

}  // namespace

extern "C" int bgk_icdf_ic2xyz_reg(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                                   const float* desc20, int32_t use_eps, float cdf_eps,
                                   const int32_t* place, int32_t n, const int32_t* fixed, int32_t n_fixed,
                                   float eps, int32_t enforce_boundaries,
                                   const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                                   float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_icdf_ic2xyz_reg: bad sizes");
    BGK_CHECK_ARG(x && place && fixed && bonds && angles && torsions && xfix && dlogp && desc20, "bgk_icdf_ic2xyz_reg: null pointer");
    BGK_CHECK_ARG(Tblacken ? (wh_mean != nullptr && keep > 0) : (keep == 3 * n_fixed), "bgk_icdf_ic2xyz_reg: bad whitening arguments");
    if (n + n_fixed > 32 || keep > KMAX) return BGK_EUNSUPPORTED;       /* the register-resident kernel's envelope */
    if (B == 0) return 0;
    TailArgs a{};
    a.bonds = bonds; a.angles = angles; a.torsions = torsions; a.xfix = xfix; a.desc = desc20; a.place = place; a.fixed = fixed;
    a.mean = wh_mean; a.T = Tblacken; a.n = n; a.n_fixed = n_fixed; a.keep = keep; a.enforce = enforce_boundaries;
    const float inf = __builtin_inff();
    a.cl = use_eps ? CdfClamp{cdf_eps, 1.0f - cdf_eps, -1.0f / cdf_eps} : CdfClamp{-inf, inf, -inf};
    a.eps = eps; a.const_ld = (float)const_ld;
    a.B = B; a.x = x; a.ldx = ldx; a.dlogp = dlogp; a.accumulate = accumulate; a.warn_count = warn_count;
    a.lds_per_wave = 64 * (n > keep ? n : keep);
    const size_t shmem = sizeof(float) * (size_t)TW * a.lds_per_wave;
    const int64_t n_wg = (((B + 63) >> 6) + TW - 1) / TW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_icdf_ic2xyz_reg: batch too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (n + n_fixed <= 24) hipLaunchKernelGGL(icdf_ic2xyz_reg_kernel<24>, dim3((unsigned)n_wg), dim3(TW * 64), shmem, st, a);
    else hipLaunchKernelGGL(icdf_ic2xyz_reg_kernel<32>, dim3((unsigned)n_wg), dim3(TW * 64), shmem, st, a);
    return bgk_launch_status("bgk_icdf_ic2xyz_reg");
}

/* the same tail for FIELD-UNIFORM marginals (every channel of a field shares one descriptor -- what the builder installs):
 * desc4 [4][20] = the bonds / angles / torsions / fixed descriptor; x must be contiguous (ldx = 3 (n + n_fixed)) and all five
 * tensors 16-byte aligned.  See icdf_ic2xyz_uni_kernel. */
static int icdf_ic2xyz_uni_launch(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                                  const float* desc4, int32_t use_eps, float cdf_eps,
                                  const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                                  float eps, int32_t enforce_boundaries,
                                  const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                                  float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count,
                                  float* y_bonds, float* y_angles, float* y_torsions, float* y_fixed, void* stream,
                                  const TailArgs* kl = nullptr) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_icdf_ic2xyz_uni: bad sizes");
    BGK_CHECK_ARG(x && place8 && fixed && bonds && angles && torsions && xfix && (dlogp || kl) && desc4, "bgk_icdf_ic2xyz_uni: null pointer");
    BGK_CHECK_ARG(Tblacken ? (wh_mean != nullptr && keep > 0) : (keep == 3 * n_fixed), "bgk_icdf_ic2xyz_uni: bad whitening arguments");
    const int n_atoms = n + n_fixed;
    if (n_atoms > 32 || keep > KMAX || n > 29 || ldx != 3 * n_atoms) return BGK_EUNSUPPORTED;
    if ((((uintptr_t)bonds | (uintptr_t)angles | (uintptr_t)torsions | (uintptr_t)xfix | (uintptr_t)x) & 15) != 0) return BGK_EUNSUPPORTED;
    if (B == 0) return 0;
    TailArgs a{};
    a.bonds = bonds; a.angles = angles; a.torsions = torsions; a.xfix = xfix; a.desc = desc4; a.place = place8; a.fixed = fixed;
    a.mean = wh_mean; a.T = Tblacken; a.n = n; a.n_fixed = n_fixed; a.keep = keep; a.enforce = enforce_boundaries;
    const float inf = __builtin_inff();
    a.cl = use_eps ? CdfClamp{cdf_eps, 1.0f - cdf_eps, -1.0f / cdf_eps} : CdfClamp{-inf, inf, -inf};
    a.eps = eps; a.const_ld = (float)const_ld;
    a.B = B; a.x = x; a.ldx = ldx; a.dlogp = dlogp; a.accumulate = accumulate; a.warn_count = warn_count;
    a.o_bonds = y_bonds; a.o_angles = y_angles; a.o_torsions = y_torsions; a.o_fixed = y_fixed;
    if (kl) {
        a.kl_mean = kl->kl_mean; a.kl_dl_in = kl->kl_dl_in; a.kl_inv_t = kl->kl_inv_t; a.kl_c_in = kl->kl_c_in; a.kl_c_out = kl->kl_c_out;
        a.kl_drop = kl->kl_drop; a.kl_u = kl->kl_u; a.kl_dl = kl->kl_dl; a.kl_partial = kl->kl_partial;
    }
    const int W = n > keep ? n : keep;
    a.lds_per_wave = 64 * (4 * W > 3 * n_atoms ? 4 * W : 3 * n_atoms);
    const size_t shmem = sizeof(float) * (size_t)TW * a.lds_per_wave;
    if (shmem > 160 * 1024) return BGK_EUNSUPPORTED;
    const int64_t n_wg = (((B + 63) >> 6) + TW - 1) / TW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_icdf_ic2xyz_uni: batch too large for one launch");
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH(NA_, EM_) do { if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(icdf_ic2xyz_uni_kernel<NA_, EM_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                  hipLaunchKernelGGL((icdf_ic2xyz_uni_kernel<NA_, EM_>), dim3((unsigned)n_wg), dim3(TW * 64), shmem, st, a); } while (0)
    if (kl) {
#define BGK_LAUNCH_KL(NA_) do { if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(icdf_ic2xyz_uni_kernel<NA_, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                hipLaunchKernelGGL((icdf_ic2xyz_uni_kernel<NA_, true, true>), dim3((unsigned)n_wg), dim3(TW * 64), shmem, st, a); } while (0)
        if (n_atoms <= 24) BGK_LAUNCH_KL(24); else BGK_LAUNCH_KL(32);
#undef BGK_LAUNCH_KL
    }
    else if (y_bonds) { if (n_atoms <= 24) BGK_LAUNCH(24, true); else BGK_LAUNCH(32, true); }
    else { if (n_atoms <= 24) BGK_LAUNCH(24, false); else BGK_LAUNCH(32, false); }
#undef BGK_LAUNCH
    return bgk_launch_status("bgk_icdf_ic2xyz_uni");
}

extern "C" int bgk_icdf_ic2xyz_uni(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                                   const float* desc4, int32_t use_eps, float cdf_eps,
                                   const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                                   float eps, int32_t enforce_boundaries,
                                   const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                                   float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream) {
    return icdf_ic2xyz_uni_launch(bonds, angles, torsions, xfix, desc4, use_eps, cdf_eps, place8, n, fixed, n_fixed, eps, enforce_boundaries,
                                  wh_mean, Tblacken, keep, const_ld, B, x, ldx, dlogp, accumulate, warn_count, nullptr, nullptr, nullptr,
                                  nullptr, stream);
}

/* the same launch for a TRAINING forward: additionally writes the four mapped fields (the outputs of the icdf maps = the inputs of
 * the coordinate transform), contiguous [B, n] x 3 and [B, keep] -- what bgk_ic_ic2xyz_backward and bgk_cdf_backward read, so the
 * backward pass runs on the existing kernels while the forward stays one launch instead of five */
extern "C" int bgk_icdf_ic2xyz_uni_train(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                                         const float* desc4, int32_t use_eps, float cdf_eps,
                                         const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                                         float eps, int32_t enforce_boundaries,
                                         const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                                         float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count,
                                         float* y_bonds, float* y_angles, float* y_torsions, float* y_fixed, void* stream) {
    BGK_CHECK_ARG(y_bonds && y_angles && y_torsions && y_fixed, "bgk_icdf_ic2xyz_uni_train: null pointer");
    return icdf_ic2xyz_uni_launch(bonds, angles, torsions, xfix, desc4, use_eps, cdf_eps, place8, n, fixed, n_fixed, eps, enforce_boundaries,
                                  wh_mean, Tblacken, keep, const_ld, B, x, ldx, dlogp, accumulate, warn_count, y_bonds, y_angles, y_torsions,
                                  y_fixed, stream);
}

/* ... and with the KL integrand formed in the same launch (BoltzmannGenerator.kldiv, bg.py:140-147, for a target that is a normal
 * distribution about t_mean -- distribution/normal.py:61-72: u = (|x - t_mean|^2 / 2 + c_in) / temperature + c_out): every lane has its
 * sample's coordinates in registers when the placements are done, so the target energy costs no second pass over x and the loss no
 * per-sample tensor.  dlogp_in [B] (may be NULL): log-det of the flow in front of the tail, read only.  Written: x, the four mapped
 * fields (as bgk_icdf_ic2xyz_uni_train), u [B], dlogp_total [B] = dlogp_in + the tail's log-det, partial [ceil(B / 64)][2] (per tile:
 * sum of u - dlogp_total over the samples kept, their number; drop_nonfinite: samples with a non-finite integrand are not kept) and
 * loss_sums [2] (f64: the partials added in a fixed order by a second small launch). */
extern "C" int bgk_icdf_ic2xyz_uni_train_kl(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                                            const float* desc4, int32_t use_eps, float cdf_eps,
                                            const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                                            float eps, int32_t enforce_boundaries,
                                            const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                                            float* x, int64_t ldx, const float* dlogp_in, int32_t* warn_count,
                                            float* y_bonds, float* y_angles, float* y_torsions, float* y_fixed,
                                            const float* t_mean, double temperature, double c_in, double c_out, int32_t drop_nonfinite,
                                            float* u, float* dlogp_total, float* partial, double* loss_sums, void* stream) {
    if (B == 0) return 0;
    BGK_CHECK_ARG(y_bonds && y_angles && y_torsions && y_fixed && u && dlogp_total && partial && loss_sums && temperature > 0.0,
                  "bgk_icdf_ic2xyz_uni_train_kl: bad arguments");
    TailArgs kl{};
    kl.kl_mean = t_mean; kl.kl_dl_in = dlogp_in; kl.kl_inv_t = (float)(1.0 / temperature); kl.kl_c_in = (float)c_in; kl.kl_c_out = (float)c_out;
    kl.kl_drop = drop_nonfinite; kl.kl_u = u; kl.kl_dl = dlogp_total; kl.kl_partial = partial;
    const int st = icdf_ic2xyz_uni_launch(bonds, angles, torsions, xfix, desc4, use_eps, cdf_eps, place8, n, fixed, n_fixed, eps, enforce_boundaries,
                                          wh_mean, Tblacken, keep, const_ld, B, x, ldx, nullptr, 0, warn_count, y_bonds, y_angles, y_torsions,
                                          y_fixed, stream, &kl);
    if (st != 0) return st;
    return bgk_loss_partial_reduce(partial, (int)((B + 63) >> 6), loss_sums, stream);
}

/* The inverse (NLL) direction of the builder tail in one launch: x [B, 3 (n + n_fixed)] -> cdf-mapped bonds / angles / torsions [B, n]
 * and (whitened) fixed coordinates [B, keep], all contiguous, + log|det J| (replaces bgk_ic_xyz2ic + 4 x bgk_cdf_transform:
 * RelativeInternalCoordinateTransformation._forward / MixedCoordinateTransformation._forward, crd_transform/ic.py:386-433, 838-860,
 * then CDFTransform._forward x 4, nn/flow/cdf.py:28-35).  zmat8 [n, 8] int32 rows (a, b, c, d, 0, 0, 0, 0); desc4 [4, 20] as in
 * bgk_icdf_ic2xyz_uni (field-uniform marginals); Twhiten [3 n_fixed, keep]; const_ld = -n (ln pi + ln 2 pi) + jac_xz. */
extern "C" int bgk_xyz2ic_cdf_uni(const float* x, const float* desc4, int32_t use_eps, float cdf_eps,
                                  const int32_t* zmat8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                                  float eps, int32_t enforce_boundaries,
                                  const float* wh_mean, const float* Twhiten, int32_t keep, double const_ld, int64_t B,
                                  float* bonds, float* angles, float* torsions, float* xfix,
                                  float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_xyz2ic_cdf_uni: bad sizes");
    BGK_CHECK_ARG(x && zmat8 && fixed && bonds && angles && torsions && xfix && dlogp && desc4, "bgk_xyz2ic_cdf_uni: null pointer");
    BGK_CHECK_ARG(Twhiten ? (wh_mean != nullptr && keep > 0) : (keep == 3 * n_fixed), "bgk_xyz2ic_cdf_uni: bad whitening arguments");
    const int n_atoms = n + n_fixed;
    if (n_atoms > 32 || keep > KMAX || 3 * n + keep > 3 * n_atoms) return BGK_EUNSUPPORTED;
    if ((((uintptr_t)bonds | (uintptr_t)angles | (uintptr_t)torsions | (uintptr_t)xfix | (uintptr_t)x) & 15) != 0) return BGK_EUNSUPPORTED;
    if (B == 0) return 0;
    TailArgs a{};
    a.x = const_cast<float*>(x); a.desc = desc4; a.place = zmat8; a.fixed = fixed;
    a.mean = wh_mean; a.T = Twhiten; a.n = n; a.n_fixed = n_fixed; a.keep = keep; a.enforce = enforce_boundaries;
    const float inf = __builtin_inff();
    a.cl = use_eps ? CdfClamp{cdf_eps, 1.0f - cdf_eps, -1.0f / cdf_eps} : CdfClamp{-inf, inf, -inf};
    a.eps = eps; a.const_ld = (float)const_ld;
    a.B = B; a.ldx = 3 * n_atoms; a.dlogp = dlogp; a.accumulate = accumulate; a.warn_count = warn_count;
    a.o_bonds = bonds; a.o_angles = angles; a.o_torsions = torsions; a.o_fixed = xfix;
    a.lds_per_wave = 64 * 3 * n_atoms;
    const size_t shmem = sizeof(float) * (size_t)TW * a.lds_per_wave;
    if (shmem > 160 * 1024) return BGK_EUNSUPPORTED;
    const int64_t n_wg = (((B + 63) >> 6) + TW - 1) / TW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_xyz2ic_cdf_uni: batch too large for one launch");
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH(NA_) do { if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xyz2ic_cdf_uni_kernel<NA_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                             hipLaunchKernelGGL(xyz2ic_cdf_uni_kernel<NA_>, dim3((unsigned)n_wg), dim3(TW * 64), shmem, st, a); } while (0)
    if (n_atoms <= 24) BGK_LAUNCH(24); else BGK_LAUNCH(32);
#undef BGK_LAUNCH
    return bgk_launch_status("bgk_xyz2ic_cdf_uni");
}
