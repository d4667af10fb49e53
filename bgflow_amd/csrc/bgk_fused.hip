/* bgk_fused.hip -- one-launch spline coupling layer: DenseNet conditioner on the f32 matrix cores
 * + rational-quadratic spline epilogue.  The conditioner activations and the spline parameters
 * never touch HBM: per sample the kernel reads d_c + d floats and writes d (+1) floats.
 *
 * Roofline: MFMA (f32-input v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s dense peak): 2*(n_in*H0 + H0*H1 +
 * H1*NCp) flops per sample against 4*(d_c + 2d + 2) bytes -> arithmetic intensity >> the
 * 19.7 flop/B machine balance.
 *
 * Decomposition (gfx950).  A wave owns 32 samples for the whole layer; a workgroup is 4 independent
 * waves (no workgroup barrier anywhere), 2 workgroups per CU = 2 waves per SIMD so that one wave's
 * VALU phases (SiLU, spline) overlap the other's MFMA phases.
 *   GEMM orientation: D[feature, sample] = W[feature, k] * X[k, sample]   (A = weights, B = data)
 *   - A fragments: weights pre-packed on the host so that one k-step of 4 output tiles is ONE
 *     coalesced 16-byte-per-lane load (1 KiB per wave, L1/L2 resident), software-prefetched
 *     PF steps ahead;
 *   - B fragments: layer 0 reads the (featurised) conditioner input from LDS; hidden layers feed the
 *     previous layer's accumulator registers straight back as B (the MFMA C/D layout gives lane
 *     (half h, sample j) the features {(r&3)+8(r>>2)+4h}: visiting k in that order needs no data
 *     movement) -- the oracle reproduces this accumulation order (oracle.mfma_k_order);
 *   - last layer: output columns packed per transformed dim (3K+1 values contiguous), processed in
 *     chunks of 128 columns = 5 dims for K = 8; a chunk goes once through wave-private LDS
 *     ([row][32 samples], conflict-free both ways) to regroup from "lane = column" to
 *     "lane = (sample, dim)" for the spline.
 * Every f32 operation is the same IEEE op in the same order as oracle/bgo_impl.h -> bit-exact
 * parity (MFMA f32 = k-ordered fma chain, one rounding per product).
 */
#include "bgk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FW = 4;                 /* waves per workgroup */
constexpr int FTHREADS = FW * 64;
constexpr int HID = 128;              /* hidden width (both hidden layers) */
constexpr int KB = 8;                 /* spline bins (template constant of this kernel) */
constexpr int PPD = 3 * KB + 1;       /* packed params per dim = 25 */
constexpr int DPC = 128 / PPD;        /* dims per 128-column chunk = 5 */
constexpr int PF = 4;                 /* A-fragment prefetch distance in k-steps */
constexpr int LDS_P = 128 * 32;       /* floats: one parameter chunk [128 rows][32 samples] */
constexpr int SROW = 33;              /* padded row stride of the small per-wave tiles */

struct FusedArgs {
    const float* cond; int64_t ldc; int d_c; int periodic;
    const float4* W0; int T0;        /* layer 0: T0 = ceil(n_in/2) k-steps (+1 bias step) */
    const float4* W1;                /* 64 + 1 steps */
    const float4* W2; int n_chunks;  /* per chunk 64 + 1 steps */
    int act;
    const float* y; int64_t ldy;
    int64_t B; int d; int inverse;
    uint64_t circ_mask;                              /* bit j set = dim j circular (d <= 64) */
    float* out; int64_t ldo;
    float* dlogp; int accumulate;
    int32_t* bin_idx; int32_t* oob_count;
    int lds_per_wave;                                /* floats */
    BgkRqsCfg cfg;
};

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == 1) return bgk_siluf(v);
    if (ACT == 2) return v > 0.0f ? v : 0.0f;
    return bgk_tanhf(v);
}

/* Packed-weight fetch: one wave-uniform base pointer + ONE per-lane byte offset (lane * 16) for the
 * whole kernel; the k-step part (step * 1 KiB) folds into the instruction's immediate offset, so
 * there is no per-load 64-bit address arithmetic (the compiler otherwise materialises one address
 * pair per step and spills them). */
struct WBuf {
    const char* base;
    unsigned voff;
};
__device__ __forceinline__ WBuf wbuf_make(const float4* base, int /*n_steps*/, int lane) {
    WBuf w;
    w.base = reinterpret_cast<const char*>(base);
    w.voff = (unsigned)lane * 16u;
    return w;
}
__device__ __forceinline__ float4 wbuf_ld(const WBuf& w, int step) {
    return *reinterpret_cast<const float4*>(w.base + (size_t)step * 1024 + w.voff);
}

/* row of output tile m held in accumulator register r by this lane (hh = lane >> 5) */
__device__ __forceinline__ int drow(int m, int r, int hh) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh; }

/* 4 output tiles x one k-step */
__device__ __forceinline__ void mfma4(f32x16 (&acc)[4], const float4& a, float b) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc[3], 0, 0, 0);
}

/* acc[4] = W[128 x 128] * h + bias   with h in accumulator layout (k order = mfma order).
 * The packed operand has 64 + 1 k-steps: the last one multiplies the bias (A, lower half-wave) by
 * 1.0 (B) -- fma(bias, 1, acc) == acc + bias, the oracle's "bias added last". */
constexpr int HSTEPS = 65;
__device__ __forceinline__ void gemm_hidden(f32x16 (&acc)[4], const f32x16 (&h)[4], const float4* Wbase, int lane) {
    const WBuf w = wbuf_make(Wbase, HSTEPS, lane);
    float4 ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p] = wbuf_ld(w, p);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = kb * 16 + r;
            float4 a = ring[s % PF];
            if (s + PF < HSTEPS) ring[s % PF] = wbuf_ld(w, s + PF);
            mfma4(acc, a, h[kb][r]);
        }
    }
    mfma4(acc, ring[64 % PF], lane < 32 ? 1.0f : 0.0f);
}

__device__ __forceinline__ void zero4(f32x16 (&acc)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
}

template <int ACT>
__global__ __launch_bounds__(FTHREADS, 2) void coupling_rqs_dense_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, hh = lane >> 5;
    float* s_p = smem + (size_t)wave * a.lds_per_wave;   /* parameter chunk [128][32]; aliases the layer-0 input X0 [2*T0][SROW] */
    float* s_y = s_p + LDS_P;                            /* y / out tile [d][SROW] */
    const int d = a.d;
    const int64_t n_tiles = (a.B + 31) / 32;

    for (int64_t tile = (int64_t)blockIdx.x * FW + wave; tile < n_tiles; tile += (int64_t)gridDim.x * FW) {
        const int64_t b0 = tile * 32;
        const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);

        /* ---- stage conditioner input (featurised) and y, transposed to [feature][sample] ---- */
        const int n_in = a.periodic ? 2 * a.d_c : a.d_c;
        for (int i = lane; i < 32 * a.d_c; i += 64) {
            const int r = i / a.d_c, c = i - r * a.d_c;
            float v = r < rows ? a.cond[(b0 + r) * a.ldc + c] : 0.0f;
            if (a.periodic) {
                float sv, cv;
                bgk_sincos2pif(v, &sv, &cv);
                s_p[c * SROW + r] = cv;
                s_p[(a.d_c + c) * SROW + r] = sv;
            } else {
                s_p[c * SROW + r] = v;
            }
        }
        if (n_in & 1) { if (lane < 32) s_p[n_in * SROW + lane] = 0.0f; }   /* zero pad row for the odd k */
        for (int i = lane; i < 32 * d; i += 64) {
            const int r = i / d, c = i - r * d;
            s_y[c * SROW + r] = r < rows ? a.y[(b0 + r) * a.ldy + c] : 0.5f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        /* ---- layer 0: h = act(W0 * x + b0), B operand from LDS, natural k order, bias = last step ---- */
        f32x16 h[4], acc[4];
        zero4(h);
        {
            const int T0 = a.T0;
            const WBuf w = wbuf_make(a.W0, T0 + 1, lane);
            float4 ring[PF];
#pragma unroll
            for (int p = 0; p < PF; ++p) ring[p] = wbuf_ld(w, p <= T0 ? p : T0);
            for (int t0 = 0; t0 < T0; t0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int t = t0 + u;
                    if (t < T0) {
                        float4 av = ring[u];
                        const int tn = t + PF;
                        ring[u] = wbuf_ld(w, tn <= T0 ? tn : T0);
                        float bv = s_p[(2 * t + hh) * SROW + j];
                        mfma4(h, av, bv);
                    }
                }
            }
            /* bias step (index T0): by construction ring[T0 % PF] holds it */
            float4 ab = wbuf_ld(w, T0);
            mfma4(h, ab, lane < 32 ? 1.0f : 0.0f);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[m][r] = act_fn<ACT>(h[m][r]);
        }
        /* ---- layer 1: h = act(W1 * h + b1), B operand = registers ---- */
        zero4(acc);
        gemm_hidden(acc, h, a.W1, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[m][r] = act_fn<ACT>(acc[m][r]);

        /* ---- layer 2 in chunks of 128 packed columns + spline ---- */
        float run = 0.0f;          /* running sum of log-dets of sample j (handed between the two half-waves) */
        int oob_local = 0;
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < a.n_chunks; ++c) {
            zero4(acc);
            gemm_hidden(acc, h, a.W2 + (size_t)c * HSTEPS * 64, lane);
            /* previous chunk's spline reads of s_p are complete (same wave, program order) */
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_p[drow(m, r, hh) * 32 + j] = acc[m][r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int nd = (d - c * DPC) < DPC ? (d - c * DPC) : DPC;
            /* elements (q, j): q = hh, hh+2, hh+4; log-det sum in ascending dim order via half-wave hand-off */
#pragma unroll 1
            for (int it = 0; it < (DPC + 1) / 2; ++it) {
                const int q = 2 * it + hh;
                const int dim = c * DPC + q;
                float lad = 0.0f;
                if (q < nd) {
                    const float* pw = s_p + (q * PPD) * 32 + j;
                    const float* ph = pw + KB * 32;
                    const float* ps = ph + KB * 32;
                    const bool circ = (a.circ_mask >> dim) & 1ull;
                    const float s_last = circ ? ps[0] : ps[KB * 32];
                    int bin, oob;
                    float x = s_y[dim * SROW + j];
                    float o = bgk_rqs_element<KB>(x, pw, ph, ps, 32, s_last, KB, a.inverse, a.cfg, &lad, &bin, &oob);
                    s_y[dim * SROW + j] = o;
                    oob_local += (j < rows) ? oob : 0;
                    if (a.bin_idx && j < rows) a.bin_idx[(b0 + j) * d + dim] = bin;
                }
                /* dim 2*it lives in the lower half-wave, dim 2*it+1 in the upper one: exchange the two
                 * log-dets and let BOTH halves add them in ascending dim order (same bits as the oracle) */
                const float lad_other = __shfl_xor(lad, 32);
                const float l0 = hh ? lad_other : lad;
                const float l1 = hh ? lad : lad_other;
                if (2 * it < nd) run = run + l0;
                if (2 * it + 1 < nd) run = run + l1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (hh == 0 && j < rows) {
            if (a.accumulate) a.dlogp[b0 + j] += run; else a.dlogp[b0 + j] = run;
        }
        for (int i = lane; i < rows * d; i += 64) {
            const int r = i / d, cc = i - r * d;
            a.out[(b0 + r) * a.ldo + cc] = s_y[cc * SROW + r];
        }
        if (a.oob_count) {
            for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
            if (lane == 0 && oob_local) atomicAdd(a.oob_count, oob_local);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int32_t bgk_pack_rqs_columns(int32_t d, int32_t K, const int32_t* nc_slot_host, int32_t* src_col) {
    if (d <= 0 || K <= 0 || 3 * K + 1 > 128) return BGK_EINVAL;
    const int ppd = 3 * K + 1, dpc = 128 / ppd;
    const int n_chunks = (d + dpc - 1) / dpc;
    const int ncp = n_chunks * 128;
    if (src_col) {
        for (int i = 0; i < ncp; ++i) src_col[i] = -1;
        for (int jd = 0; jd < d; ++jd) {
            const int c = jd / dpc, q = jd - c * dpc;
            int32_t* dst = src_col + c * 128 + q * ppd;
            for (int k = 0; k < K; ++k) {
                dst[k] = jd * K + k;
                dst[K + k] = d * K + jd * K + k;
                dst[2 * K + k] = 2 * d * K + jd * K + k;
            }
            const int slot = nc_slot_host ? nc_slot_host[jd] : -1;
            dst[3 * K] = slot >= 0 ? 3 * d * K + slot : -1;
        }
    }
    return ncp;
}

extern "C" int bgk_coupling_rqs_dense(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                      const float* W0p, const float* W1p, const float* W2p,
                                      int32_t H0, int32_t H1, int32_t act, const float* y,
                                      int64_t ldy, int64_t B, int32_t d, int32_t K, uint64_t circ_mask,
                                      int32_t inverse,
                                      double left, double right, double bottom, double top,
                                      double min_bin_width, double min_bin_height,
                                      double min_derivative, int32_t identity_init, float* out,
                                      int64_t ldo, float* dlogp, int32_t accumulate,
                                      int32_t* bin_idx, int32_t* oob_count, void* stream) {
    BGK_CHECK_ARG(cond && W0p && W1p && W2p && y && out && dlogp, "bgk_coupling_rqs_dense: null pointer");
    BGK_CHECK_ARG(B >= 0 && d > 0 && d_c > 0, "bgk_coupling_rqs_dense: bad sizes");
    if (H0 != HID || H1 != HID || K != KB || d > 64 || act < 1 || act > 3) {
        bgk_set_error("bgk_coupling_rqs_dense: only hidden=(128,128), n_bins=8, d<=64, act in {SiLU,ReLU,Tanh} are fused "
                      "(got H0=%d H1=%d K=%d d=%d act=%d)", H0, H1, K, d, act);
        return BGK_EUNSUPPORTED;
    }
    const int n_in = periodic ? 2 * d_c : d_c;
    BGK_CHECK_ARG((n_in + 1) * SROW <= LDS_P, "bgk_coupling_rqs_dense: conditioner input of %d features too wide", n_in);
    BGK_CHECK_ARG(min_bin_width * K <= 1.0 && min_bin_height * K <= 1.0,
                  "Minimal bin width/height too large for the number of bins");
    if (B == 0) return 0;
    FusedArgs a;
    a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic;
    a.W0 = reinterpret_cast<const float4*>(W0p); a.T0 = (n_in + 1) / 2;
    a.W1 = reinterpret_cast<const float4*>(W1p);
    a.W2 = reinterpret_cast<const float4*>(W2p); a.n_chunks = (d + DPC - 1) / DPC;
    a.act = act; a.y = y; a.ldy = ldy; a.B = B; a.d = d; a.inverse = inverse;
    a.circ_mask = circ_mask;
    a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.bin_idx = bin_idx; a.oob_count = oob_count;
    a.lds_per_wave = LDS_P + d * SROW;
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
    size_t shmem = sizeof(float) * (size_t)FW * a.lds_per_wave;
    int64_t n_wg = ((B + 31) / 32 + FW - 1) / FW;
    int grid = (int)(n_wg < 256 * 2 * 8 ? n_wg : 256 * 2 * 8);
    hipStream_t st = (hipStream_t)stream;
    if (act == 1) hipLaunchKernelGGL(coupling_rqs_dense_kernel<1>, dim3(grid), dim3(FTHREADS), shmem, st, a);
    else if (act == 2) hipLaunchKernelGGL(coupling_rqs_dense_kernel<2>, dim3(grid), dim3(FTHREADS), shmem, st, a);
    else hipLaunchKernelGGL(coupling_rqs_dense_kernel<3>, dim3(grid), dim3(FTHREADS), shmem, st, a);
    return bgk_launch_status("bgk_coupling_rqs_dense");
}
