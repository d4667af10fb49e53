/* bgk_ic_bwd.hip -- backward (VJP) kernels of the coordinate transforms that still lacked one after round 1:
 *   bgk_ic_xyz2ic_backward   xyz -> internal coordinates (RelativeInternalCoordinateTransformation._forward, crd_transform/ic.py:386-433,
 *                            row Jacobians ic_helper.py:148-293; PCA whitening of the fixed block pca.py:74-82)
 *   bgk_ic_refsys_backward   global reference system of the first three atoms, both directions (ic.py:162-265, ic_helper.py:480-680)
 * Both evaluate the SAME formulas as the forward kernels (bgk_ic.hip; explicit row-Jacobian determinant, eps clamps) on dual
 * numbers (bgk_dual.h), i.e. they differentiate exactly what the reference's autograd differentiates.  Lane = sample; a row
 * of the Z-matrix depends on 4 atoms = 12 inputs -> 4 passes of Dual<3>; the reference system has 9 inputs -> 3 passes.
 * HBM traffic: 4 * (2 * 3 n_atoms + 3 n + keep + 1) B per sample (xyz2ic), 4 * (9 + 9 + 1 + 9) B (refsys); the arithmetic
 * (17 rows x 4 passes x ~1 kflop at ala2 size) is noticeable only against the forward kernels, not against a training step.
 * Used by NLL training / force matching; the KL direction differentiates ic -> xyz (bgk_ic_ic2xyz_backward, bgk_ic.hip). */
#include "bgk_common.h"
#include "bgk_dual.h"

namespace {

#define PI_F 3.14159265358979323846f
typedef Dual<3> D3;
typedef DV3<3> V;

__device__ __forceinline__ D3 clampmin(D3 v, float eps, int enforce) { return (enforce && v.v < eps) ? dconst<3>(eps) : v; }

struct RowOut { D3 rn, ang, tor, logdet; };

/* one Z-matrix row (atom x1 placed relative to x2, x3, x4): same operations as ic_xyz2ic_kernel */
__device__ __forceinline__ RowOut row_eval(V x1, V x2, V x3, V x4, float eps, int enforce, int normalize) {
    RowOut o;
    /* dist_deriv */
    V r = dsub(x2, x1);
    D3 rn = clampmin(dnorm(r), eps, enforce);
    V Jb = {-(r.x / rn), -(r.y / rn), -(r.z / rn)};
    /* angle_deriv */
    V r12 = dsub(x1, x2);
    D3 n12 = clampmin(dnorm(r12), eps, enforce);
    V u12 = ddivs(r12, n12);
    V r32 = dsub(x3, x2);
    D3 n32 = clampmin(dnorm(r32), eps, enforce);
    V u32 = ddivs(r32, n32);
    D3 cosa = ddot(u12, u32);
    D3 u12v[3] = {u12.x, u12.y, u12.z}, u32v[3] = {u32.x, u32.y, u32.z}, Jav[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        D3 s = dconst<3>(0.0f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            D3 Pkc = ((k == c ? 1.0f : 0.0f) - u12v[k] * u12v[c]) / n12;
            s = s + u32v[k] * Pkc;
        }
        Jav[c] = s;
    }
    if (enforce) cosa = dclamp(cosa, -1.0f + eps, 1.0f - eps);
    D3 ang = dacos(cosa);
    D3 sq = dsqrt(1.0f - cosa * cosa);
    V Ja = {-(Jav[0] / sq), -(Jav[1] / sq), -(Jav[2] / sq)};
    /* torsion_deriv */
    V b0v = {-(x2.x - x1.x), -(x2.y - x1.y), -(x2.z - x1.z)};
    V b1 = dsub(x3, x2), b2 = dsub(x4, x3);
    D3 b1n = clampmin(dnorm(b1), eps, enforce);
    V u = ddivs(b1, b1n);
    D3 b0u = ddot(b0v, u), b2u = ddot(b2, u);
    V v = {b0v.x - b0u * u.x, b0v.y - b0u * u.y, b0v.z - b0u * u.z};
    V w = {b2.x - b2u * u.x, b2.y - b2u * u.y, b2.z - b2u * u.z};
    D3 xx = ddot(v, w);
    D3 yy = ddot(dcross(u, v), w);
    D3 tor = datan2(yy, xx);
    D3 q = clampmin(xx * xx + yy * yy, eps, enforce);
    D3 dadx = -(yy / q), dady = xx / q;
    V wxu = dcross(w, u);
    V g = {dadx * w.x + dady * wxu.x, dadx * w.y + dady * wxu.y, dadx * w.z + dady * wxu.z};
    D3 gu = ddot(g, u);
    V Jt = {g.x - gu * u.x, g.y - gu * u.y, g.z - gu * u.z};
    D3 det = ddot(dcross(Jb, Ja), Jt);
    o.logdet = dlog(dabs(det));
    if (normalize) { ang = ang / PI_F; tor = (tor + PI_F) / (2.0f * PI_F); }
    o.rn = rn; o.ang = ang; o.tor = tor;
    return o;
}

constexpr int XB_THREADS = 64;

struct Xyz2IcBwdArgs {
    const float* x; int64_t ldx;
    const float* g_bonds; const float* g_angles; const float* g_torsions; int64_t ldgic;
    const float* g_xfix; int64_t ldgf;
    const float* g_dlogp;
    const int32_t* zmat; const int32_t* fixed;
    int n, n_fixed, n_atoms, keep, normalize, enforce;
    float eps;
    const float* T;             /* Twhiten [3 nf, keep] or NULL */
    int64_t B;
    float* g_x; int64_t ldgx;
    int sx, sic, sfx;
};

__device__ __forceinline__ void tload(float* dst, int s, const float* src, int64_t ld, int rows, int cols) {
    for (int i = threadIdx.x; i < rows * cols; i += (int)blockDim.x) {
        int r = i / cols, c = i - r * cols;
        dst[r * s + c] = src[(int64_t)r * ld + c];
    }
}

__device__ __forceinline__ V seed_atom(const float* p, bool active) {
    V a;
    a.x = dseed<3>(p[0], active ? 0 : -1); a.y = dseed<3>(p[1], active ? 1 : -1); a.z = dseed<3>(p[2], active ? 2 : -1);
    return a;
}

__global__ __launch_bounds__(XB_THREADS) void ic_xyz2ic_bwd_kernel(Xyz2IcBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = (int)blockDim.x, n = a.n, nf3 = 3 * a.n_fixed, na3 = 3 * a.n_atoms;
    float* s_x = smem;                    /* [TS][sx] positions */
    float* s_g = s_x + TS * a.sx;         /* [TS][sx] position adjoints */
    float* s_b = s_g + TS * a.sx;         /* [TS][sic] upstream g_bonds / g_angles / g_torsions */
    float* s_a = s_b + TS * a.sic;
    float* s_t = s_a + TS * a.sic;
    float* s_f = s_t + TS * a.sic;        /* [TS][sfx] upstream g_xfix */
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        tload(s_x, a.sx, a.x + b0 * a.ldx, a.ldx, rows, na3);
        tload(s_b, a.sic, a.g_bonds + b0 * a.ldgic, a.ldgic, rows, n);
        tload(s_a, a.sic, a.g_angles + b0 * a.ldgic, a.ldgic, rows, n);
        tload(s_t, a.sic, a.g_torsions + b0 * a.ldgic, a.ldgic, rows, n);
        tload(s_f, a.sfx, a.g_xfix + b0 * a.ldgf, a.ldgf, rows, a.keep);
        __syncthreads();
        if (tid < rows) {
            const float* xr = s_x + tid * a.sx;
            float* gp = s_g + tid * a.sx;
            for (int c = 0; c < na3; ++c) gp[c] = 0.0f;
            const float gl = a.g_dlogp[b0 + tid];
            for (int i = 0; i < n; ++i) {
                const int idx[4] = {a.zmat[4 * i], a.zmat[4 * i + 1], a.zmat[4 * i + 2], a.zmat[4 * i + 3]};
                const float gb = s_b[tid * a.sic + i], ga = s_a[tid * a.sic + i], gt = s_t[tid * a.sic + i];
                if (gb == 0.0f && ga == 0.0f && gt == 0.0f && gl == 0.0f) continue;    /* masked-out sample: exact zeros, no 0 * inf */
                for (int k = 0; k < 4; ++k) {
                    const RowOut o = row_eval(seed_atom(xr + 3 * idx[0], k == 0), seed_atom(xr + 3 * idx[1], k == 1),
                                              seed_atom(xr + 3 * idx[2], k == 2), seed_atom(xr + 3 * idx[3], k == 3),
                                              a.eps, a.enforce, a.normalize);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        gp[3 * idx[k] + c] += gb * o.rn.d[c] + ga * o.ang.d[c] + gt * o.tor.d[c] + gl * o.logdet.d[c];
                }
            }
            /* fixed block: pass-through or z = (x_fixed - mean) Twhiten */
            if (a.T) {
                for (int c = 0; c < nf3; ++c) {
                    float s = 0.0f;
                    for (int k = 0; k < a.keep; ++k) s += s_f[tid * a.sfx + k] * a.T[c * a.keep + k];
                    gp[3 * a.fixed[c / 3] + c % 3] += s;
                }
            } else {
                for (int c = 0; c < nf3; ++c) gp[3 * a.fixed[c / 3] + c % 3] += s_f[tid * a.sfx + c];
            }
        }
        __syncthreads();
        for (int i = tid; i < rows * na3; i += (int)blockDim.x) {
            int r = i / na3, c = i - r * na3;
            a.g_x[(b0 + r) * a.ldgx + c] = s_g[r * a.sx + c];
        }
        __syncthreads();
    }
}

/* ---- global reference system on duals: the same operations as ic_refsys_kernel (bgk_ic.hip) ---- */
struct RefOut { D3 o[9]; D3 dl; };

__device__ __forceinline__ RefOut refsys_fwd(const D3 (&v)[9], float eps, int enforce, int normalize) {
    RefOut r;
    V x0 = {v[0], v[1], v[2]}, x1 = {v[3], v[4], v[5]}, x2 = {v[6], v[7], v[8]};
    V r01 = dsub(x1, x0), r12 = dsub(x2, x1);
    D3 d01 = clampmin(dnorm(r01), eps, enforce), d12 = clampmin(dnorm(r12), eps, enforce);
    V aa = dsub(x0, x1), cc = dsub(x2, x1);
    D3 an = clampmin(dnorm(aa), eps, enforce), cn = clampmin(dnorm(cc), eps, enforce);
    D3 cosang = (aa.x / an) * (cc.x / cn) + (aa.y / an) * (cc.y / cn) + (aa.z / an) * (cc.z / cn);
    if (enforce) cosang = dclamp(cosang, -1.0f + eps, 1.0f - eps);
    D3 a012 = dacos(cosang);
    D3 e1n = clampmin(dnorm(r01), eps, enforce);
    V e1 = ddivs(r01, e1n);
    V e2 = dcross(dsub(x2, x0), e1);
    D3 e2n = clampmin(dnorm(e2), eps, enforce);
    e2 = ddivs(e2, e2n);
    V e3 = dcross(e2, e1);
    D3 alpha = datan2(e1.x, -e1.y), beta = e1.z, gamma = datan2(-e3.z, -e2.z);
    r.dl = -(2.0f * dlog(d01) + 2.0f * dlog(d12) + dlog(dsin(a012)));
    if (normalize) {
        a012 = a012 / PI_F; alpha = (alpha + PI_F) / (2.0f * PI_F); gamma = (gamma + PI_F) / (2.0f * PI_F);
        r.dl = r.dl + (-logf(PI_F) - 2.0f * logf(2.0f * PI_F));
    }
    r.o[0] = x0.x; r.o[1] = x0.y; r.o[2] = x0.z; r.o[3] = d01; r.o[4] = d12; r.o[5] = a012; r.o[6] = alpha; r.o[7] = beta; r.o[8] = gamma;
    return r;
}

__device__ __forceinline__ RefOut refsys_inv(const D3 (&v)[9], float eps, int enforce, int normalize) {
    RefOut r;
    V x0 = {v[0], v[1], v[2]};
    D3 d01 = v[3], d12 = v[4], a012 = v[5], alpha = v[6], beta = v[7], gamma = v[8];
    r.dl = dconst<3>(0.0f);
    if (normalize) {
        alpha = alpha * (2.0f * PI_F) - PI_F; gamma = gamma * (2.0f * PI_F) - PI_F; a012 = a012 * PI_F;
        r.dl = r.dl + (logf(PI_F) + 2.0f * logf(2.0f * PI_F));
    }
    r.dl = r.dl + 2.0f * dlog(d01) + 2.0f * dlog(d12) + dlog(dsin(a012));
    const D3 zero = dconst<3>(0.0f);
    V p1 = {zero, zero, d01}, p0 = {zero, zero, zero}, p3 = {zero, dconst<3>(-1.0f), zero};
    V v1 = dsub(p1, p0), v2 = dsub(p1, p3);
    V nv = dcross(v1, v2), nn = dcross(v1, nv);
    D3 nvn = clampmin(dnorm(nv), eps, enforce), nnn = clampmin(dnorm(nn), eps, enforce);
    const float tq = 0.5f * PI_F, st = sinf(tq), ct = cosf(tq);
    D3 sa = dsin(a012), ca = dcos(a012);
    V nh = ddivs(nv, nvn), nnh = ddivs(nn, nnn);
    V v3 = {nh.x * (-st) + nnh.x * ct, nh.y * (-st) + nnh.y * ct, nh.z * (-st) + nnh.z * ct};
    D3 v3n = clampmin(dnorm(v3), eps, enforce), v1n = clampmin(dnorm(v1), eps, enforce);
    V v3h = ddivs(v3, v3n), v1h = ddivs(v1, v1n);
    V p2 = {p1.x + v3h.x * d12 * sa - v1h.x * d12 * ca, p1.y + v3h.y * d12 * sa - v1h.y * d12 * ca,
            p1.z + v3h.z * d12 * sa - v1h.z * d12 * ca};
    D3 bA = dacos(beta);
    D3 caA = dcos(alpha), saA = dsin(alpha), cb = dcos(bA), sb = dsin(bA), cg = dcos(gamma), sg = dsin(gamma);
    const D3 one = dconst<3>(1.0f);
    D3 Rz1[9] = {caA, -saA, zero, saA, caA, zero, zero, zero, one}, Rx[9] = {one, zero, zero, zero, cb, -sb, zero, sb, cb},
       Rz2[9] = {cg, -sg, zero, sg, cg, zero, zero, zero, one};
    D3 T[9], R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { D3 s = zero; for (int k = 0; k < 3; ++k) s = s + Rz1[3 * i + k] * Rx[3 * k + j]; T[3 * i + j] = s; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { D3 s = zero; for (int k = 0; k < 3; ++k) s = s + T[3 * i + k] * Rz2[3 * k + j]; R[3 * i + j] = s; }
    const D3 p1v[3] = {p1.x, p1.y, p1.z}, p2v[3] = {p2.x, p2.y, p2.z}, x0v[3] = {x0.x, x0.y, x0.z};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        D3 s1 = zero, s2 = zero;
        for (int dd = 0; dd < 3; ++dd) { s1 = s1 + p1v[dd] * R[3 * e + dd]; s2 = s2 + p2v[dd] * R[3 * e + dd]; }
        r.o[e] = x0v[e]; r.o[3 + e] = s1 + x0v[e]; r.o[6 + e] = s2 + x0v[e];
    }
    return r;
}

struct RefBwdArgs { const float* in; const float* g_out; const float* g_dlogp; float* g_in; int64_t B; int inverse, normalize, enforce; float eps; };

__global__ __launch_bounds__(128) void ic_refsys_bwd_kernel(RefBwdArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    float in[9], go[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { in[i] = a.in[9 * b + i]; go[i] = a.g_out[9 * b + i]; }
    const float gl = a.g_dlogp[b];
    bool live = gl != 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) live = live || (go[i] != 0.0f);
    for (int pass = 0; pass < 3; ++pass) {
        float g3[3] = {0.0f, 0.0f, 0.0f};
        if (live) {
            D3 v[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) v[i] = dseed<3>(in[i], i - 3 * pass);
            const RefOut r = a.inverse ? refsys_inv(v, a.eps, a.enforce, a.normalize) : refsys_fwd(v, a.eps, a.enforce, a.normalize);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = gl * r.dl.d[c];
#pragma unroll
                for (int i = 0; i < 9; ++i) s += go[i] * r.o[i].d[c];
                g3[c] = s;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) a.g_in[9 * b + 3 * pass + c] = g3[c];
    }
}

}  // namespace

extern "C" int bgk_ic_xyz2ic_backward(const float* x, int64_t ldx, const int32_t* zmat, int32_t n,
                                      const int32_t* fixed, int32_t n_fixed, int32_t normalize_angles, float eps,
                                      int32_t enforce_boundaries, const float* Twhiten, int32_t keep, int64_t B,
                                      const float* g_bonds, const float* g_angles, const float* g_torsions, int64_t ldgic,
                                      const float* g_xfix, int64_t ldgf, const float* g_dlogp,
                                      float* g_x, int64_t ldgx, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && n > 0 && n_fixed > 0, "bgk_ic_xyz2ic_backward: bad sizes");
    BGK_CHECK_ARG(x && zmat && fixed && g_bonds && g_angles && g_torsions && g_xfix && g_dlogp && g_x, "bgk_ic_xyz2ic_backward: null pointer");
    BGK_CHECK_ARG(Twhiten ? keep > 0 : keep == 3 * n_fixed, "bgk_ic_xyz2ic_backward: bad whitening arguments");
    if (B == 0) return 0;
    Xyz2IcBwdArgs a{};
    a.x = x; a.ldx = ldx; a.g_bonds = g_bonds; a.g_angles = g_angles; a.g_torsions = g_torsions; a.ldgic = ldgic;
    a.g_xfix = g_xfix; a.ldgf = ldgf; a.g_dlogp = g_dlogp; a.zmat = zmat; a.fixed = fixed; a.n = n; a.n_fixed = n_fixed;
    a.n_atoms = n + n_fixed; a.keep = keep; a.normalize = normalize_angles; a.enforce = enforce_boundaries; a.eps = eps;
    a.T = Twhiten; a.B = B; a.g_x = g_x; a.ldgx = ldgx;
    a.sx = (3 * a.n_atoms) | 1; a.sic = n | 1; a.sfx = keep | 1;
    int ts = XB_THREADS;              /* halve the tile until one sample's atoms per thread fit the LDS (big molecules) */
    while (ts > 1 && sizeof(float) * (size_t)ts * (size_t)(2 * a.sx + 3 * a.sic + a.sfx) > 160 * 1024) ts >>= 1;
    size_t shmem = sizeof(float) * (size_t)ts * (size_t)(2 * a.sx + 3 * a.sic + a.sfx);
    if (shmem > 160 * 1024) { bgk_set_error("bgk_ic_xyz2ic_backward: %d atoms do not fit the LDS tile", a.n_atoms); return BGK_EUNSUPPORTED; }
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ic_xyz2ic_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int64_t n_tiles = (B + ts - 1) / ts;
    int grid = (int)(n_tiles < 256 * 12 ? n_tiles : 256 * 12);
    hipLaunchKernelGGL(ic_xyz2ic_bwd_kernel, dim3(grid), dim3(ts), shmem, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_ic_xyz2ic_backward");
}

extern "C" int bgk_ic_refsys_backward(const float* in, const float* g_out, const float* g_dlogp, int64_t B, int32_t inverse,
                                      int32_t normalize_angles, float eps, int32_t enforce_boundaries, float* g_in, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && in && g_out && g_dlogp && g_in, "bgk_ic_refsys_backward: bad arguments");
    if (B == 0) return 0;
    RefBwdArgs a{in, g_out, g_dlogp, g_in, B, inverse, normalize_angles, enforce_boundaries, eps};
    hipLaunchKernelGGL(ic_refsys_bwd_kernel, dim3((unsigned)((B + 127) / 128)), dim3(128), 0, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_ic_refsys_backward");
}
