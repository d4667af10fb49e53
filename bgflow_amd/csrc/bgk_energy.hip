/* bgk_energy.hip -- the energies that bracket the flow: the target end of the KL integrand
 * u(F(z)) - log|det J| (bgflow/bg.py:13-17) and the prior end of the NLL integrand (bg.py:20-22), for the distributions of the
 * BASELINE configs, as ONE launch over the tensors of a sample:
 *   kind 0  NormalDistribution without `cov` (bgflow/distribution/normal.py:61-72):  0.5 sum_j (x_j - mean_j)^2
 *   kind 1  DoubleWellEnergy (bgflow/distribution/energy/double_well.py:17-22):      a x_0 + b x_0^2 + c x_0^4 + 0.5 sum_{j>=1} x_j^2
 *   kind 2  a constant (UniformDistribution, bgflow/distribution/distributions.py:100-117: sum_j log(high_j - low_j))
 * A ProductEnergy / ProductDistribution (bgflow/distribution/product.py:13-117: the sum of the components' energies) is the list of
 * its components' fields: u = (sum_f e_f(x_f) + c_in) / T + c_out.  Optionally the kernel also forms the per-sample KL loss
 * u - dlogp and its block partial sums [sum, n] (non-finite samples dropped on request), reduced in fixed order by a second tiny
 * launch -- dp.global_mean then all-reduces a ready 2-vector.  The backward launch writes g_x of every field (and g_dlogp).
 *
 * Roofline: HBM, 4 (sum_f d_f + 1) B per sample forward, 4 (2 sum_f d_f + 1) B backward.  Rows of a tile are staged coalesced
 * through LDS (odd stride) in column chunks of 96 (any width fits), one lane per row adds its terms in ascending column order
 * (deterministic).  Rows that are multiples of 4 wide at 16-byte aligned addresses take the float4 kernels (energy_rows4_kernel, round 6:
 * L lanes per row, fixed xor tree); BGK_ENERGY_STAGED=1 keeps the staging kernels for every shape (the A/B). */
#include "bgk_common.h"

namespace {

constexpr int NE_THREADS = 256;
constexpr int NE_ROWS = 128;          /* rows per tile */
constexpr int NE_CW = 96;             /* columns per LDS chunk: 128 x 97 floats = 49.7 KB */
constexpr int NE_MAXF = BGK_MAX_ENERGY_FIELDS;

struct EField {
    const float* x; int64_t ldx; const float* p;        /* p: mean [d] of a normal field or NULL */
    int d, kind; float a, b, c;
    uint32_t magic_cw, magic_last; int last_w;          /* i / w by multiply-high for w = NE_CW and for the last (partial) chunk */
};

struct EArgs {
    EField f[NE_MAXF]; int n; int64_t B;
    float inv_t, c_in, c_out; float* u;
    const float* dlogp; int drop_nonfinite; float* partial;     /* partial [gridDim.x][2] or NULL */
};
typedef const __attribute__((address_space(4))) EArgs* eargs_t;   /* run-time indexed field table: scalar loads from the argument block */

__device__ __forceinline__ uint32_t magic_of(int w) { return (uint32_t)(((1ull << 32) + (uint64_t)w - 1) / (uint64_t)w); }

__global__ __launch_bounds__(NE_THREADS) void energy_fields_kernel(EArgs a) {
    __shared__ float s_x[NE_ROWS * (NE_CW | 1)];
    __shared__ float s_red[2 * NE_ROWS];
    constexpr int S = NE_CW | 1;
    const eargs_t ka = (eargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.B + NE_ROWS - 1) / NE_ROWS;
    float bsum = 0.0f, bcnt = 0.0f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * NE_ROWS;
        const int rows = (int)((a.B - b0) < NE_ROWS ? (a.B - b0) : NE_ROWS);
        float e = 0.0f;
        for (int fi = 0; fi < a.n; ++fi) {
            const int kind = ka->f[fi].kind, d = ka->f[fi].d;
            if (kind == 2) { e += ka->f[fi].a; continue; }
            const float* __restrict__ x = ka->f[fi].x;
            const float* __restrict__ mean = ka->f[fi].p;
            const int64_t ldx = ka->f[fi].ldx;
            const float ca = ka->f[fi].a, cb = ka->f[fi].b, cc = ka->f[fi].c;
            float acc = 0.0f, acc0 = 0.0f;
            for (int c0 = 0; c0 < d; c0 += NE_CW) {
                const int w = (d - c0) < NE_CW ? (d - c0) : NE_CW;
                const uint32_t magic = w == NE_CW ? ka->f[fi].magic_cw : ka->f[fi].magic_last;
                for (int i = tid; i < rows * w; i += NE_THREADS) {
                    const int r = w == 1 ? i : (int)__umulhi((unsigned)i, magic), c = i - r * w, col = c0 + c;   /* (2^32 / 1 has no 32-bit magic) */
                    float v = x[(b0 + r) * ldx + col];
                    float s;
                    if (kind == 0) {
                        v -= mean ? mean[col] : 0.0f;
                        s = v * v;
                    } else {
                        const float v2 = v * v;
                        s = col == 0 ? (ca * v + cb * v2) + cc * (v2 * v2) : v2;
                    }
                    s_x[r * S + c] = s;
                }
                __syncthreads();
                if (tid < rows) {
                    for (int c = 0; c < w; ++c) {
                        const float s = s_x[tid * S + c];
                        if (kind == 1 && c0 + c == 0) acc0 = s; else acc += s;
                    }
                }
                __syncthreads();
            }
            e += acc0 + 0.5f * acc;
        }
        if (tid < rows) {
            const float u = (e + a.c_in) * a.inv_t + a.c_out;
            a.u[b0 + tid] = u;
            if (a.partial) {
                const float loss = u - a.dlogp[b0 + tid];
                const bool ok = !a.drop_nonfinite || __builtin_isfinite(loss);
                bsum += ok ? loss : 0.0f;
                bcnt += ok ? 1.0f : 0.0f;
            }
        }
    }
    if (a.partial) {                   /* block partial: fixed order over the 128 row lanes */
        if (tid < NE_ROWS) { s_red[tid] = bsum; s_red[NE_ROWS + tid] = bcnt; }
        __syncthreads();
        if (tid == 0) {
            float s = 0.0f, c = 0.0f;
            for (int i = 0; i < NE_ROWS; ++i) { s += s_red[i]; c += s_red[NE_ROWS + i]; }
            a.partial[2 * blockIdx.x] = s; a.partial[2 * blockIdx.x + 1] = c;
        }
    }
}

/* Rows of widths that are multiples of 4 at 16-byte aligned addresses (round 6; cfg 2's DoubleWellEnergy(64) at 2^20 samples ran the
 * staging kernel above at 1 TB/s: 0.26 ms for 268 MB): a row belongs to L = 2^k >= d / 4 consecutive lanes, every lane reads ONE float4
 * of the row (a wave instruction = 64 / L whole rows, consecutive in memory), adds its four terms in ascending order and the lanes of
 * the row combine through a fixed xor tree -- no LDS, no division, deterministic.  Same terms as above (the first column of a double
 * well apart, 0.5 x the rest). */
__global__ __launch_bounds__(NE_THREADS) void energy_rows4_kernel(EArgs a, int L) {
    __shared__ float s_red[2 * NE_THREADS];
    const eargs_t ka = (eargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x, l = tid & (L - 1), grp = tid / L, rpb = NE_THREADS / L;
    float bsum = 0.0f, bcnt = 0.0f;
    for (int64_t row = (int64_t)blockIdx.x * rpb + grp; row < a.B; row += (int64_t)gridDim.x * rpb) {
        float e = 0.0f;
        for (int fi = 0; fi < a.n; ++fi) {
            const int kind = ka->f[fi].kind, d = ka->f[fi].d;
            if (kind == 2) { e += ka->f[fi].a; continue; }
            float part = 0.0f;
            if (4 * l < d) {
                const float4 v = *reinterpret_cast<const float4*>(ka->f[fi].x + row * ka->f[fi].ldx + 4 * l);
                if (kind == 0) {
                    const float* mean = ka->f[fi].p;
                    const float4 m = mean ? *reinterpret_cast<const float4*>(mean + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float t0 = v.x - m.x, t1 = v.y - m.y, t2 = v.z - m.z, t3 = v.w - m.w;
                    part = 0.5f * (((t0 * t0 + t1 * t1) + t2 * t2) + t3 * t3);
                } else {
                    const float v2 = v.x * v.x;
                    const float first = l == 0 ? (ka->f[fi].a * v.x + ka->f[fi].b * v2) + ka->f[fi].c * (v2 * v2) : 0.5f * v2;
                    part = first + 0.5f * ((v.y * v.y + v.z * v.z) + v.w * v.w);
                }
            }
            for (int off = L >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off);
            e += part;
        }
        if (l == 0) {
            const float u = (e + a.c_in) * a.inv_t + a.c_out;
            a.u[row] = u;
            if (a.partial) {
                const float loss = u - a.dlogp[row];
                const bool ok = !a.drop_nonfinite || __builtin_isfinite(loss);
                bsum += ok ? loss : 0.0f;
                bcnt += ok ? 1.0f : 0.0f;
            }
        }
    }
    if (a.partial) {                   /* block partial: fixed order over the row lanes */
        s_red[tid] = bsum; s_red[NE_THREADS + tid] = bcnt;
        __syncthreads();
        if (tid == 0) {
            float s = 0.0f, c = 0.0f;
            for (int i = 0; i < NE_THREADS; i += L) { s += s_red[i]; c += s_red[NE_THREADS + i]; }
            a.partial[2 * blockIdx.x] = s; a.partial[2 * blockIdx.x + 1] = c;
        }
    }
}

/* out[0] = sum of the block sums, out[1] = sum of the block counts, both in double and in a fixed order (deterministic): lane t adds
 * the blocks t, t + 64, ... in ascending order, lane 0 then adds the 64 lane sums in ascending order.  (One lane walking all ~2000
 * partials through dependent global loads took 0.13 ms.) */
__global__ __launch_bounds__(64) void energy_partial_reduce_kernel(const float* partial, int nblk, double* out) {
    __shared__ double s_s[64], s_c[64];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) { s += (double)partial[2 * i]; c += (double)partial[2 * i + 1]; }
    s_s[threadIdx.x] = s; s_c[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tc = 0.0;
        for (int t = 0; t < 64; ++t) { ts += s_s[t]; tc += s_c[t]; }
        out[0] = ts; out[1] = tc;
    }
}

struct EBwdField { const float* x; int64_t ldx; const float* p; float* g_x; int64_t ldg; int d, kind; float a, b, c; };
struct EBwdArgs {
    EBwdField f[NE_MAXF]; int n; int64_t B; float inv_t;
    const float* g_u;                   /* [B] upstream gradient of u, or NULL: then g_u[b] = g_scalar[0] * mask[b] (loss-sum form) */
    const float* g_scalar; const float* u; const float* dlogp; int drop_nonfinite; float* g_dlogp;
};
typedef const __attribute__((address_space(4))) EBwdArgs* ebargs_t;

__global__ __launch_bounds__(NE_THREADS) void energy_fields_bwd_kernel(EBwdArgs a) {
    const ebargs_t ka = (ebargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const float gs = a.g_scalar ? a.g_scalar[0] : 0.0f;
    auto row_grad = [&](int64_t r) -> float {
        if (a.g_u) return a.g_u[r];
        const bool ok = !a.drop_nonfinite || __builtin_isfinite(a.u[r] - a.dlogp[r]);
        return ok ? gs : 0.0f;
    };
    if (a.g_dlogp)                      /* d(sum_i (u_i - dlogp_i)) / d dlogp_i = -1 for the kept samples */
        for (int64_t r = (int64_t)blockIdx.x * NE_THREADS + threadIdx.x; r < a.B; r += (int64_t)gridDim.x * NE_THREADS)
            a.g_dlogp[r] = -row_grad(r);
    for (int fi = 0; fi < a.n; ++fi) {
        const int kind = ka->f[fi].kind, d = ka->f[fi].d;
        float* g_x = ka->f[fi].g_x;
        if (kind == 2 || !g_x) continue;
        const float* __restrict__ x = ka->f[fi].x;
        const float* __restrict__ mean = ka->f[fi].p;
        const int64_t ldx = ka->f[fi].ldx, ldg = ka->f[fi].ldg;
        const float ca = ka->f[fi].a, cb = ka->f[fi].b, cc = ka->f[fi].c;
        const int64_t total = a.B * d;
        for (int64_t i = (int64_t)blockIdx.x * NE_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * NE_THREADS) {
            const int64_t r = i / d;
            const int col = (int)(i - r * d);
            const float v = x[r * ldx + col];
            float de;
            if (kind == 0) de = v - (mean ? mean[col] : 0.0f);
            else de = col == 0 ? ca + 2.0f * cb * v + 4.0f * cc * (v * v * v) : v;
            g_x[r * ldg + col] = row_grad(r) * de * a.inv_t;
        }
    }
}

/* the float4 form of the backward for the same class of rows (see energy_rows4_kernel): a row = L lanes, no division per element */
__global__ __launch_bounds__(NE_THREADS) void energy_rows4_bwd_kernel(EBwdArgs a, int L) {
    const ebargs_t ka = (ebargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    const float gs = a.g_scalar ? a.g_scalar[0] : 0.0f;
    const int tid = threadIdx.x, l = tid & (L - 1), grp = tid / L, rpb = NE_THREADS / L;
    for (int64_t row = (int64_t)blockIdx.x * rpb + grp; row < a.B; row += (int64_t)gridDim.x * rpb) {
        float gr;
        if (a.g_u) gr = a.g_u[row];
        else gr = (!a.drop_nonfinite || __builtin_isfinite(a.u[row] - a.dlogp[row])) ? gs : 0.0f;
        if (a.g_dlogp && l == 0) a.g_dlogp[row] = -gr;
        for (int fi = 0; fi < a.n; ++fi) {
            const int kind = ka->f[fi].kind, d = ka->f[fi].d;
            float* g_x = ka->f[fi].g_x;
            if (kind == 2 || !g_x || 4 * l >= d) continue;
            const float4 v = *reinterpret_cast<const float4*>(ka->f[fi].x + row * ka->f[fi].ldx + 4 * l);
            float4 de;
            if (kind == 0) {
                const float* mean = ka->f[fi].p;
                const float4 m = mean ? *reinterpret_cast<const float4*>(mean + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
                de = make_float4(v.x - m.x, v.y - m.y, v.z - m.z, v.w - m.w);
            } else {
                de = v;
                if (l == 0) de.x = ka->f[fi].a + 2.0f * ka->f[fi].b * v.x + 4.0f * ka->f[fi].c * (v.x * v.x * v.x);
            }
            *reinterpret_cast<float4*>(g_x + row * ka->f[fi].ldg + 4 * l) =
                make_float4(gr * de.x * a.inv_t, gr * de.y * a.inv_t, gr * de.z * a.inv_t, gr * de.w * a.inv_t);
        }
    }
}

/* lanes per row of the float4 kernels for these fields (a power of two, <= 64), or 0: some field is not a multiple of 4 wide / wider
 * than 256 / off 16-byte boundaries */
template <typename F>
int rows4_lanes(const F* f, int n, const float* const* g_x, const int64_t* ldg) {
    int L = 1;
    const auto al = [](const void* p, int64_t ld) { return ((uintptr_t)p & 15) == 0 && ld % 4 == 0; };
    for (int i = 0; i < n; ++i) {
        if (f[i].kind == 2) continue;
        if (f[i].d % 4 != 0 || f[i].d > 256 || !al(f[i].x, f[i].ldx) || (f[i].kind == 0 && f[i].p && ((uintptr_t)f[i].p & 15) != 0)) return 0;
        if (g_x && g_x[i] && !al(g_x[i], ldg[i])) return 0;
        while (4 * L < f[i].d) L <<= 1;
    }
    return L;
}

int fill_fields(const char* what, EField* f, int32_t n_fields, const float* const* x, const int64_t* ldx, const int32_t* d, const int32_t* kind,
                const float* const* param, const float* coef) {
    BGK_CHECK_ARG(n_fields >= 1 && n_fields <= NE_MAXF && x && ldx && d && kind, "%s: 1..%d fields", what, NE_MAXF);
    for (int i = 0; i < n_fields; ++i) {
        BGK_CHECK_ARG(kind[i] >= 0 && kind[i] <= 2 && d[i] > 0, "%s: field %d: bad kind / width", what, i);
        BGK_CHECK_ARG(kind[i] == 2 || (x[i] && ldx[i] >= d[i]), "%s: field %d: null tensor / row stride", what, i);
        f[i].x = x[i]; f[i].ldx = ldx[i]; f[i].p = param ? param[i] : nullptr; f[i].d = d[i]; f[i].kind = kind[i];
        f[i].a = coef ? coef[3 * i] : 0.0f; f[i].b = coef ? coef[3 * i + 1] : 0.0f; f[i].c = coef ? coef[3 * i + 2] : 0.0f;
        const int last = d[i] % NE_CW == 0 ? NE_CW : d[i] % NE_CW;
        f[i].magic_cw = (uint32_t)(((1ull << 32) + NE_CW - 1) / NE_CW);
        f[i].magic_last = (uint32_t)(((1ull << 32) + (uint64_t)last - 1) / (uint64_t)last);
        f[i].last_w = last;
    }
    return 0;
}

}  // namespace

extern "C" int bgk_energy_fields(const float* const* x, const int64_t* ldx, const int32_t* d, const int32_t* kind,
                                 const float* const* param, const float* coef, int32_t n_fields, int64_t B,
                                 double temperature, double c_in, double c_out, float* u,
                                 const float* dlogp, int32_t drop_nonfinite, float* partial, int32_t nblk, double* loss_sums,
                                 void* stream) {
    BGK_CHECK_ARG((u || B == 0) && B >= 0 && temperature > 0.0, "bgk_energy_fields: bad arguments");
    BGK_CHECK_ARG(!loss_sums || (dlogp && partial && nblk >= 1), "bgk_energy_fields: the loss sums need dlogp and a [nblk, 2] workspace");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {               /* an empty batch (its tensors have no storage: null pointers): the loss sums are zero */
        if (loss_sums) { hipError_t e = hipMemsetAsync(loss_sums, 0, 2 * sizeof(double), s); if (e != hipSuccess) return (int)e; }
        return 0;
    }
    EArgs a{};
    const int st = fill_fields("bgk_energy_fields", a.f, n_fields, x, ldx, d, kind, param, coef);
    if (st) return st;
    a.n = n_fields; a.B = B; a.inv_t = (float)(1.0 / temperature); a.c_in = (float)c_in; a.c_out = (float)c_out; a.u = u;
    a.dlogp = loss_sums ? dlogp : nullptr; a.drop_nonfinite = drop_nonfinite; a.partial = loss_sums ? partial : nullptr;
    const int L = getenv("BGK_ENERGY_STAGED") ? 0 : rows4_lanes(a.f, n_fields, (const float* const*)nullptr, (const int64_t*)nullptr);
    const int64_t n_tiles = L ? (B + NE_THREADS / L - 1) / (NE_THREADS / L) : (B + NE_ROWS - 1) / NE_ROWS;
    int grid = (int)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
    if (loss_sums && grid > nblk) grid = nblk;
    if (L) hipLaunchKernelGGL(energy_rows4_kernel, dim3(grid), dim3(NE_THREADS), 0, s, a, L);
    else hipLaunchKernelGGL(energy_fields_kernel, dim3(grid), dim3(NE_THREADS), 0, s, a);
    if (loss_sums) hipLaunchKernelGGL(energy_partial_reduce_kernel, dim3(1), dim3(64), 0, s, partial, grid, loss_sums);
    return bgk_launch_status("bgk_energy_fields");
}

/* [sum, count] partials -> loss_sums [2] (f64, fixed order): shared with the KL epilogue of the training tail (bgk_tail.hip) */
int bgk_loss_partial_reduce(const float* partial, int n_partials, double* loss_sums, void* stream) {
    hipLaunchKernelGGL(energy_partial_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, n_partials, loss_sums);
    return bgk_launch_status("bgk_loss_partial_reduce");
}

extern "C" int bgk_energy_fields_backward(const float* const* x, const int64_t* ldx, const int32_t* d, const int32_t* kind,
                                          const float* const* param, const float* coef, int32_t n_fields, int64_t B,
                                          double temperature, const float* g_u,
                                          const float* g_scalar, const float* u, const float* dlogp, int32_t drop_nonfinite, float* g_dlogp,
                                          float* const* g_x, const int64_t* ldg, void* stream) {
    BGK_CHECK_ARG(B >= 0 && temperature > 0.0 && g_x && ldg, "bgk_energy_fields_backward: bad arguments");
    BGK_CHECK_ARG(g_u || (g_scalar && u && dlogp), "bgk_energy_fields_backward: need g_u [B] or (g_scalar, u, dlogp)");
    if (B == 0) return 0;
    EField tmp[NE_MAXF];
    const int st = fill_fields("bgk_energy_fields_backward", tmp, n_fields, x, ldx, d, kind, param, coef);
    if (st) return st;
    EBwdArgs a{};
    int64_t total = 0;
    for (int i = 0; i < n_fields; ++i) {
        BGK_CHECK_ARG(!g_x[i] || ldg[i] >= d[i], "bgk_energy_fields_backward: field %d: gradient row stride", i);
        a.f[i] = EBwdField{tmp[i].x, tmp[i].ldx, tmp[i].p, g_x[i], ldg[i], tmp[i].d, tmp[i].kind, tmp[i].a, tmp[i].b, tmp[i].c};
        total += (int64_t)d[i] * B;
    }
    a.n = n_fields; a.B = B; a.inv_t = (float)(1.0 / temperature);
    a.g_u = g_u; a.g_scalar = g_scalar; a.u = u; a.dlogp = dlogp; a.drop_nonfinite = drop_nonfinite; a.g_dlogp = g_dlogp;
    const int L = getenv("BGK_ENERGY_STAGED") ? 0 : rows4_lanes(tmp, n_fields, (const float* const*)g_x, ldg);
    if (L) {
        const int64_t blocks = (B + NE_THREADS / L - 1) / (NE_THREADS / L);
        const int grid = (int)(blocks < 256 * 16 ? blocks : 256 * 16);
        hipLaunchKernelGGL(energy_rows4_bwd_kernel, dim3(grid), dim3(NE_THREADS), 0, (hipStream_t)stream, a, L);
        return bgk_launch_status("bgk_energy_fields_backward");
    }
    const int64_t blocks = (total + NE_THREADS * 4 - 1) / (NE_THREADS * 4);
    const int grid = (int)(blocks < 256 * 16 ? (blocks < 1 ? 1 : blocks) : 256 * 16);
    hipLaunchKernelGGL(energy_fields_bwd_kernel, dim3(grid), dim3(NE_THREADS), 0, (hipStream_t)stream, a);
    return bgk_launch_status("bgk_energy_fields_backward");
}

/* the single-field forms of round 2 (ABI kept) */
extern "C" int bgk_normal_energy(const float* x, int64_t ldx, const float* mean, int32_t d, int64_t B,
                                 double temperature, double log_z, float* u, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(x && u, "bgk_normal_energy: null pointer");
    const int32_t kind = 0;
    return bgk_energy_fields(&x, &ldx, &d, &kind, &mean, nullptr, 1, B, temperature, 0.0, log_z, u, nullptr, 0, nullptr, 0, nullptr, stream);
}

extern "C" int bgk_normal_energy_backward(const float* x, int64_t ldx, const float* mean, int32_t d, int64_t B,
                                          double temperature, const float* g_u, float* g_x, int64_t ldg, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(x && g_u && g_x, "bgk_normal_energy_backward: null pointer");
    const int32_t kind = 0;
    return bgk_energy_fields_backward(&x, &ldx, &d, &kind, &mean, nullptr, 1, B, temperature, g_u, nullptr, nullptr, nullptr, 0, nullptr,
                                      &g_x, &ldg, stream);
}
