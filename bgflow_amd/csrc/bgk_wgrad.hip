/* bgk_wgrad.hip -- weight and bias gradients of the conditioner MLP's Linear layers in the training step
 * (autograd of nn/dense.py:47-48:  dW = g^T h,  db = sum_rows g  for g [B, n], h [B, k], B = batch):
 *   bgk_dense_weight_grad   one call per coupling layer: (g_params, h1) -> dW2, db2;  (g_z1, h0) -> dW1, db1;
 *                           (g_z0, featurised conditioner input) -> dW0, db0
 * replacing, per layer, three split-K batched hipBLASLt GEMMs + three partial-sum reductions + six column-sum launches
 * (8.4 ms of a 28 ms KL step at 2^18 samples, profiles/r01_kl_step_kernel_stats.csv) by three GEMM launches + one reduction.
 *
 * Roofline: HBM -- every operand element is read once (h re-read once per 128-row block of g^T, from L2): 4 (n + k) B per
 * sample and GEMM, ~1 GB per B|A layer at 2^18 samples; the contraction runs over the BATCH, so the matrix cores need
 * 2 (n k) flops per sample = 0.15 GFLOP/MB: two orders of magnitude below the machine balance even in the 3-product split
 * form used here.
 * Decomposition: workgroup = 128 rows of the output (4 waves x one 32-row tile) x all k <= 128 columns x one slab of batch
 * rows (split-K); per step of 16 batch rows every thread fetches 8 consecutive batch rows of ONE column of g and of h
 * (each wave-level load = 32 | 64 consecutive floats of a row: coalesced), splits them into f16 hi + lo parts -- g under a
 * per-tensor power-of-two scale that puts its largest magnitude (published by the kernel that wrote it: bgk_rqs_backward,
 * bgk_dense_backward_dx) into [2^14, 2^15), h clamped to the f16 range like the forward's activations: hi*hi + hi*lo + lo*hi carries
 * 22 significant bits per product, the forward's f32-class arithmetic (rounds 1-4 split into bf16 pairs, 16 bits per product: the
 * flat KL gradient was 1.3e-4 off its f64 value).  The g values a thread fetches ARE its A fragment (lane = (column, 8-row block)) and stay
 * in registers; the h parts (shared by the four waves) go to LDS as ONE 16-byte value per part -- already the B operand layout;
 * column stride 48 B keeps the ds_read_b128 conflict-free.  LDS is double-buffered (one barrier per step).  The bias gradient is the running sum of the g values a thread loads anyway.  Partials per slab go
 * to a workspace and are summed in fixed order by wgrad_reduce_kernel: deterministic, no atomics.
 */
#include "bgk_mfma_h2.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_THREADS = 256;
constexpr int COLS = 128;            /* columns of g (output rows) per workgroup, and max columns of h */
constexpr int CSTRIDE = 48;          /* bytes per column in an LDS operand array: 2 x 16 B (two 8-row blocks) + 16 B pad */
constexpr int ARR = COLS * CSTRIDE;  /* bytes of one operand array */

struct WgArgs {
    const float* g; int64_t ldg; int n;        /* [B, n] */
    const float* h; int64_t ldh; int k;        /* [B, k] (k <= 128) or, featurise != 0, the raw conditioner input [B, k / 2] */
    int featurise;                             /* 1: h columns are cos(2 pi x_c) for c < k/2, sin(2 pi x_c) after (nn/periodic.py:30-37) */
    int act;                                   /* 0: h as given; 1 SiLU, 2 ReLU, 3 Tanh: the array holds pre-activations z, h = act(z) (hardware
                                                * exp2 / rcp forms, the same as bgk_dense_backward_dx's activation recomputation) */
    int64_t B; int64_t rows_per_slab; int n_slabs; int n_blocks;
    float* part_w;                             /* [n_slabs][n][k] */
    float* part_b;                             /* [2 n_slabs][n] */
    const float* g_absmax;                     /* [1] largest |g| (device; NULL: g is split unscaled -- values below 6e-5 lose bits) */
};

/* 8 values -> f16 hi and lo parts (16 B each), 1.5 VALU instructions per value + the caller's scale / clamp (h2_split_pair:
 * v_cvt_pk_f16_f32 + 2 x v_fma_mix) */
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) h2_split_pair(v[2 * e], v[2 * e + 1], h[e], l[e]);
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

/* the (up to) three GEMMs of one coupling layer in ONE launch: blocks [first[q], first[q + 1]) work on GEMM q.  Alone, the two
 * 128 x 128 GEMMs fill one workgroup per CU and wait on HBM latency; together with the [P x 128] one the chip is full. */
struct WgGroup { WgArgs g[3]; int first[4]; };

__global__ __launch_bounds__(WG_THREADS, 3) void wgrad_kernel(WgGroup grp_args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   /* 2 buffers x {h_hi, h_lo} */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = (int)blockIdx.x >= grp_args.first[2] ? 2 : ((int)blockIdx.x >= grp_args.first[1] ? 1 : 0);
    const WgArgs& a = grp_args.g[q];
    const int block = (int)blockIdx.x - grp_args.first[q];
    /* block -> (slab, n-block): blocks that share a slab (and re-read the same rows of h) are 8 apart = on the same XCD */
    const int per8 = 8 * a.n_blocks;
    const int grp = block / per8, rem = block - grp * per8;
    const int slab = grp * 8 + (rem & 7), nb = rem >> 3;
    if (slab >= a.n_slabs) return;
    const int64_t r0 = (int64_t)slab * a.rows_per_slab;
    const int64_t r1 = (r0 + a.rows_per_slab) < a.B ? (r0 + a.rows_per_slab) : a.B;
    /* g: a thread fetches exactly the A fragment its MFMAs consume (lane = (column i of the wave's 32-column tile, 8-row block
     * kb)): no LDS round trip for g.  h (shared by the four waves): thread -> (column c, 8-row block rg), through LDS. */
    const int gi = lane & 31, gkb = lane >> 5;
    const int gcol = nb * COLS + wave * 32 + gi;
    const int c = tid & 127, rg = tid >> 7;
    const bool g_ok = gcol < a.n, h_ok = c < a.k;
    const int kh = a.k >> 1;
    const int hc = a.featurise ? (c < kh ? c : c - kh) : c;  /* source column of h */
    const int KT = (a.k + 31) >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    float bsum = 0.0f;
    float g_inv;
    const float g_scale = h2_pow2_scale(a.g_absmax ? a.g_absmax[0] : 0.0f, g_inv);     /* max |g| -> [2^14, 2^15) */
    const int my_off = c * CSTRIDE + rg * 16;                /* where my 16-byte h values go */
    /* software pipeline, two 16-row steps deep: the 16 values of steps s + 1 and s + 2 are in flight while step s is converted /
     * multiplied (one step of distance left every step waiting on HBM: a step takes ~500 cycles, a loaded round trip > 2000).
     * Loads are raw buffer loads: base = the slab's first row, per-lane byte offsets computed ONCE (8 + 8 VGPRs), the step's row
     * offset in an SGPR, rows past the slab / the batch and padded columns return 0 from the hardware range check -- no per-element
     * 64-bit address arithmetic or compares (they were 2/3 of the kernel's VALU instructions: 230 per wave and step, VALU 58 % busy). */
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + r0 * a.ldg), 0, (int)((r1 - r0) * a.ldg * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)(a.h + r0 * a.ldh), 0, (int)((r1 - r0) * a.ldh * 4), 0x00020000);
    unsigned vg[8], vh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        vg[e] = g_ok ? (unsigned)(((8 * gkb + e) * a.ldg + gcol) * 4) : 0x40000000u;      /* beyond num_records: reads as 0 */
        vh[e] = h_ok ? (unsigned)(((8 * rg + e) * a.ldh + hc) * 4) : 0x40000000u;
    }
    const int step_g = (int)(16 * a.ldg * 4), step_h = (int)(16 * a.ldh * 4);
    auto fetch = [&](int64_t r, float (&gv)[8], float (&hv)[8]) {
        const int sg = (int)((r - r0) >> 4) * step_g, sh = (int)((r - r0) >> 4) * step_h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, vg[e], sg, 0));
            hv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_h, vh[e], sh, 0));
        }
    };
    auto step = [&](int64_t r, float (&gq)[8], float (&hq)[8], int buf) {
        float gv[8], hv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gv[e] = gq[e]; hv[e] = hq[e]; }
        if (r + 32 < r1) fetch(r + 32, gq, hq);
        if (a.featurise && h_ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sv, cv;
                bgk_sincos2pif(hv[e], &sv, &cv);
                hv[e] = (r + 8 * rg + e < r1) ? (c < kh ? cv : sv) : 0.0f;
            }
        }
        if (a.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = hv[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hv[e] * -1.44269504088896341f));
        } else if (a.act == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = hv[e] > 0.0f ? hv[e] : 0.0f;
        } else if (a.act == 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hv[e] * 2.88539008177792681f));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bsum += gv[e];
            gv[e] *= g_scale;
            hv[e] = __builtin_amdgcn_fmed3f(hv[e], -65000.0f, 65000.0f);
        }
        uint4 ghi, glo, hhi, hlo;
        split8(gv, ghi, glo);
        split8(hv, hhi, hlo);
        unsigned char* base = smem + buf * 2 * ARR;
        *reinterpret_cast<uint4*>(base + 0 * ARR + my_off) = hhi;
        *reinterpret_cast<uint4*>(base + 1 * ARR + my_off) = hlo;
        __syncthreads();
        const h16x8 ahi = __builtin_bit_cast(h16x8, ghi), alo = __builtin_bit_cast(h16x8, glo);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < KT) {
                const int rd_b = (m * 32 + (lane & 31)) * CSTRIDE + (lane >> 5) * 16;
                const h16x8 bhi = *reinterpret_cast<const h16x8*>(base + 0 * ARR + rd_b);
                const h16x8 blo = *reinterpret_cast<const h16x8*>(base + 1 * ARR + rd_b);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[m], 0, 0, 0);
            }
        }
        /* the other buffer is written next; its readers finished before the barrier above */
    };
    float g0[8], h0[8], g1[8], h1[8];
    fetch(r0, g0, h0);
    fetch(r0 + 16, g1, h1);
    for (int64_t r = r0; r < r1; r += 32) {
        step(r, g0, h0, 0);
        if (r + 16 < r1) step(r + 16, g1, h1, 1);
    }
    /* partial dW of this slab: accumulator layout -> [n][k] */
    const int j = lane & 31, hh = lane >> 5;
    float* pw = a.part_w + (int64_t)slab * a.n * a.k;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m < KT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = nb * COLS + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, col = m * 32 + j;
                if (row < a.n && col < a.k) pw[(int64_t)row * a.k + col] = acc[m][r] * g_inv;
            }
        }
    }
    if (g_ok) a.part_b[((int64_t)slab * 2 + gkb) * a.n + gcol] = bsum;
}

/* fixed-order sum of the slab partials: 8 lanes per output element each sum every 8th slab (2 accumulators), combined in ascending
 * order through LDS.  The 8 lanes of an element sit in 8 different half-waves: a half-wave reads 32 CONSECUTIVE elements of one slab
 * (128 B, coalesced) -- with the 8 lanes adjacent every load touched 8 slabs x 32 B. */
struct RedOne { const float* part_w; const float* part_b; int n_slabs, n, k; float* gW; float* gb;
                int ldo; };     /* row stride of gW (k: a whole contiguous [n, k] gradient; larger: a column block of a wider one) */
struct RedGroup { RedOne r[3]; int64_t first[4]; };      /* first[q]: first output element of GEMM q */

__device__ __forceinline__ void wgrad_reduce_body(const RedGroup& rg, int accumulate, int bx) {
    __shared__ float s_part[8][32];
    const int sub = threadIdx.x >> 5, el = threadIdx.x & 31;
    const int64_t gi = (int64_t)bx * 32 + el;
    const int q = gi >= rg.first[2] ? 2 : (gi >= rg.first[1] ? 1 : 0);
    const RedOne& o = rg.r[q];
    const int64_t i = gi - rg.first[q];
    const int n = o.n, n_slabs = o.n_slabs;
    const int64_t nk = (int64_t)n * o.k;
    const int64_t total = gi < rg.first[3] ? nk + (o.gb ? n : 0) : 0;
    float acc = 0.0f;
    if (i < nk && i < total) {
        float a0 = 0.0f, a1 = 0.0f;
        int s = sub;
        for (; s + 8 < n_slabs; s += 16) { a0 += o.part_w[(int64_t)s * nk + i]; a1 += o.part_w[(int64_t)(s + 8) * nk + i]; }
        if (s < n_slabs) a0 += o.part_w[(int64_t)s * nk + i];
        acc = a0 + a1;
    } else if (i < total) {
        const int col = (int)(i - nk);
        float a0 = 0.0f, a1 = 0.0f;
        int s = sub;
        for (; s + 8 < 2 * n_slabs; s += 16) { a0 += o.part_b[(int64_t)s * n + col]; a1 += o.part_b[(int64_t)(s + 8) * n + col]; }
        if (s < 2 * n_slabs) a0 += o.part_b[(int64_t)s * n + col];
        acc = a0 + a1;
    }
    s_part[sub][el] = acc;
    __syncthreads();
    if (sub == 0) {
        float t = s_part[0][el];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += s_part[u][el];
        if (i < nk && i < total) {
            const int64_t at = o.ldo == o.k ? i : (i / o.k) * (int64_t)o.ldo + (i % o.k);
            o.gW[at] = accumulate ? o.gW[at] + t : t;
        }
        else if (i < total) o.gb[i - nk] = accumulate ? o.gb[i - nk] + t : t;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(RedGroup rg, int accumulate) { wgrad_reduce_body(rg, accumulate, (int)blockIdx.x); }

/* the reductions of several coupling layers in one launch (bgk_dense_weight_grad_reduce_many): a 16-layer backward pass ends with
 * ONE reduction over all partial sets instead of sixteen 25 us launches at a third of the memory rate */
constexpr int RED_MANY = 16;
struct RedMany { RedGroup r[RED_MANY]; };
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(RedMany, int accumulate) {
    const RedMany* km = (const RedMany*)__builtin_amdgcn_kernarg_segment_ptr();     /* run-time indexed: read in place */
    const int li = blockIdx.y;
    if ((int64_t)blockIdx.x * 32 >= km->r[li].first[3]) return;
    wgrad_reduce_body(km->r[li], accumulate, (int)blockIdx.x);
}

int slabs_for(int64_t B, int n) {
    /* ~2 workgroups per CU, slabs of at least 1024 rows, a multiple of 8 slabs (XCD-aware block map) */
    const int n_blocks = (n + COLS - 1) / COLS;
    int n_slabs = (int)((B + 1023) / 1024);
#ifndef BGK_WG_TARGET
#define BGK_WG_TARGET 512            /* workgroups per GEMM (~2 per CU) */
#endif
    const int want = (BGK_WG_TARGET + n_blocks - 1) / n_blocks;
    n_slabs = n_slabs > want ? want : n_slabs;
    return ((n_slabs + 7) / 8) * 8;
}
int64_t ws_need(int64_t B, int n, int k) {
    const int n_slabs = slabs_for(B, n);
    return (int64_t)n_slabs * n * k + (int64_t)2 * n_slabs * n;
}

struct GemmSpec { const char* what; const float* g; int64_t ldg; int n; const float* h; int64_t ldh; int k; int featurise; int act; float* gW; float* gb;
                  const float* g_absmax; int ldo = 0; };      /* ldo: row stride of gW (0: k) */

/* mode bits: 1 launch the GEMMs, 2 launch the reduction (red_out != NULL: also / only hand the reduction's descriptor out) */
int gemm_group(const GemmSpec* specs, int count, int64_t B, float* ws, int64_t ws_floats, int accumulate, hipStream_t st, int mode = 3,
               RedGroup* red_out = nullptr) {
    WgGroup grp;
    RedGroup red;
    int blocks = 0;
    int64_t used = 0, outs = 0;
    for (int q = 0; q < 3; ++q) {
        grp.first[q] = blocks;
        red.first[q] = outs;
        if (q >= count) { grp.g[q] = WgArgs{}; red.r[q] = RedOne{}; continue; }
        const GemmSpec& sp = specs[q];
        BGK_CHECK_ARG(sp.n > 0 && sp.k > 0 && sp.k <= COLS && (!sp.featurise || (sp.k % 2 == 0)), "%s: n = %d, k = %d not supported (k <= 128)",
                      sp.what, sp.n, sp.k);
        const int n_blocks = (sp.n + COLS - 1) / COLS, n_slabs = slabs_for(B, sp.n);
        int64_t rows = (B + n_slabs - 1) / n_slabs;
        rows = ((rows + 15) / 16) * 16;
        const int64_t need = ws_need(B, sp.n, sp.k);
        BGK_CHECK_ARG(ws_floats >= used + need, "%s: workspace of %lld floats needed, %lld given", sp.what, (long long)(used + need), (long long)ws_floats);
        float* pw = ws + used;
        float* pb = pw + (int64_t)n_slabs * sp.n * sp.k;
        grp.g[q] = WgArgs{sp.g, sp.ldg, sp.n, sp.h, sp.ldh, sp.k, sp.featurise, sp.act, B, rows, n_slabs, n_blocks, pw, pb, sp.g_absmax};
        red.r[q] = RedOne{pw, pb, n_slabs, sp.n, sp.k, sp.gW, sp.gb, sp.ldo > 0 ? sp.ldo : sp.k};
        blocks += n_slabs * n_blocks;
        used += need;
        outs += (int64_t)sp.n * sp.k + sp.n;
    }
    grp.first[3] = blocks;
    red.first[3] = outs;
    if (mode & 1) hipLaunchKernelGGL(wgrad_kernel, dim3(blocks), dim3(WG_THREADS), 2 * 2 * ARR, st, grp);
    if (mode & 2) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((outs + 31) / 32)), dim3(256), 0, st, red, accumulate);
    if (red_out) *red_out = red;
    return 0;
}

}  // namespace

/* ---- the general form: hidden layers of H1 / H0 <= 128 units whose pre-activations / gradients sit in arrays of row pitch ldz ----
 * (an affine coupling's shift / scale networks with 64 hidden units; the conditioners the fused kernels run zero-padded): the
 * gradients come out in the parameters' own shapes [P, H1], [H1, H0], [H0, n_in] -- no padded copies, nothing to slice. */
extern "C" int64_t bgk_mlp_weight_grad_workspace(int64_t B, int32_t P, int32_t H1, int32_t H0, int32_t n_in) {
    /* floats: the partial sets of the three GEMMs (they run in one launch) */
    return ws_need(B, P, H1) + ws_need(B, H1, H0) + ws_need(B, H0, n_in);
}

extern "C" int bgk_mlp_weight_grad(const float* g_out, int64_t ldg, int32_t P, const float* g_z1, const float* g_z0,
                                   const float* h1, const float* h0, int64_t ldz, int32_t H1, int32_t H0, int32_t h_act,
                                   const float* cond, int64_t ldc, int32_t d_c, int32_t periodic, int64_t B,
                                   float* workspace, int64_t workspace_floats,
                                   float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0, int32_t accumulate,
                                   const float* g_absmax, void* stream) {
    BGK_CHECK_ARG(g_out && g_z1 && g_z0 && h1 && h0 && cond && workspace, "bgk_mlp_weight_grad: null pointer");
    BGK_CHECK_ARG(B > 0 && P > 0 && d_c > 0 && h_act >= 0 && h_act <= 3 && accumulate >= 0 && accumulate <= 2 && H1 > 0 && H1 <= COLS
                  && H0 > 0 && H0 <= COLS && ldz >= H1 && ldz >= H0, "bgk_mlp_weight_grad: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int n_in = periodic ? 2 * d_c : d_c;
    GemmSpec specs[3];
    int count = 0;
    const float* am = g_absmax;      /* {max |g_out|, max |g_z1|, max |g_z0|} on the device, or NULL */
    if (gW2) specs[count++] = GemmSpec{"bgk_mlp_weight_grad (layer 2)", g_out, ldg, P, h1, ldz, H1, 0, h_act, gW2, gb2, am};
    if (gW1) specs[count++] = GemmSpec{"bgk_mlp_weight_grad (layer 1)", g_z1, ldz, H1, h0, ldz, H0, 0, h_act, gW1, gb1, am ? am + 1 : nullptr};
    if (gW0) specs[count++] = GemmSpec{"bgk_mlp_weight_grad (layer 0)", g_z0, ldz, H0, cond, ldc, n_in, periodic, 0, gW0, gb0, am ? am + 2 : nullptr};
    if (count == 0) return 0;
    /* accumulate == 2: the partial sums only -- the caller reduces them later with bgk_mlp_weight_grad_reduce_many */
    const int rc = gemm_group(specs, count, B, workspace, workspace_floats, accumulate, st, accumulate == 2 ? 1 : 3);
    if (rc != 0) return rc;
    return bgk_launch_status("bgk_mlp_weight_grad");
}

/* H1 / H0 may be NULL: 128 hidden units everywhere (= bgk_dense_weight_grad_reduce_many) */
extern "C" int bgk_mlp_weight_grad_reduce_many(int32_t n, const int64_t* B, const int32_t* P, const int32_t* H1, const int32_t* H0,
                                               const int32_t* n_in, float* const* workspace, float* const* gW2, float* const* gb2,
                                               float* const* gW1, float* const* gb1, float* const* gW0, float* const* gb0,
                                               int32_t accumulate, void* stream) {
    BGK_CHECK_ARG(n >= 0 && B && P && n_in && workspace && gW2 && gb2 && gW1 && gb1 && gW0 && gb0 && (accumulate == 0 || accumulate == 1),
                  "bgk_mlp_weight_grad_reduce_many: bad arguments");
    if (n == 0) return 0;       /* nothing to do (and no launch status to ask a GPU-less box for) */
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += RED_MANY) {
        const int cnt = n - base < RED_MANY ? n - base : RED_MANY;
        RedMany M;
        int64_t max_outs = 0;
        for (int c = 0; c < cnt; ++c) {
            const int i = base + c;
            const int h1 = H1 ? H1[i] : 128, h0 = H0 ? H0[i] : 128;
            BGK_CHECK_ARG(B[i] > 0 && P[i] > 0 && n_in[i] > 0 && n_in[i] <= COLS && h1 > 0 && h1 <= COLS && h0 > 0 && h0 <= COLS
                          && workspace[i] && gW2[i] && gW1[i] && gW0[i], "bgk_mlp_weight_grad_reduce_many: bad layer %d", i);
            GemmSpec specs[3] = {GemmSpec{"bgk_mlp_weight_grad_reduce_many (layer 2)", nullptr, 0, P[i], nullptr, 0, h1, 0, 0, gW2[i], gb2[i], nullptr},
                                 GemmSpec{"bgk_mlp_weight_grad_reduce_many (layer 1)", nullptr, 0, h1, nullptr, 0, h0, 0, 0, gW1[i], gb1[i], nullptr},
                                 GemmSpec{"bgk_mlp_weight_grad_reduce_many (layer 0)", nullptr, 0, h0, nullptr, 0, n_in[i], 0, 0, gW0[i], gb0[i], nullptr}};
            const int rc = gemm_group(specs, 3, B[i], workspace[i], (int64_t)1 << 60, accumulate, st, 0, &M.r[c]);
            if (rc != 0) return rc;
            max_outs = M.r[c].first[3] > max_outs ? M.r[c].first[3] : max_outs;
        }
        for (int c = cnt; c < RED_MANY; ++c) M.r[c] = M.r[0];
        hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((unsigned)((max_outs + 31) / 32), (unsigned)cnt), dim3(256), 0, st, M, accumulate);
    }
    return bgk_launch_status("bgk_mlp_weight_grad_reduce_many");
}

/* ONE Linear layer of any width (round 6): gW [n, k] = g^T h, gb [n] = sum_rows g for g [B, n], h [B, k] -- the input columns in blocks of
 * 128 (three blocks per launch), each block's partial sums reduced into its columns of gW.  The weight / bias gradient of a Linear
 * outside the fused training envelopes (a conditioner with other depths / widths, a stand-alone DenseNet): before round 6
 * torch.bmm + a sum on hipBLASLt (dense._gram_tn). */
extern "C" int64_t bgk_linear_weight_grad_workspace(int64_t B, int32_t n, int32_t k) {
    (void)k;
    return 3 * ws_need(B, n, COLS);
}

extern "C" int bgk_linear_weight_grad(const float* g, int64_t ldg, int32_t n, const float* h, int64_t ldh, int32_t k, int64_t B,
                                      float* workspace, int64_t workspace_floats, float* gW, float* gb, int32_t accumulate,
                                      const float* g_absmax, void* stream) {
    if (B == 0) return 0;
    BGK_CHECK_ARG(g && h && workspace && gW && B > 0 && n > 0 && k > 0 && ldg >= n && ldh >= k && (accumulate == 0 || accumulate == 1),
                  "bgk_linear_weight_grad: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int k0 = 0; k0 < k; k0 += 3 * COLS) {
        GemmSpec specs[3];
        int count = 0;
        for (int kb = k0; kb < k && count < 3; kb += COLS) {
            const int kw = k - kb < COLS ? k - kb : COLS;
            specs[count] = GemmSpec{"bgk_linear_weight_grad", g, ldg, n, h + kb, ldh, kw, 0, 0, gW + kb, kb == 0 ? gb : nullptr, g_absmax, k};
            ++count;
        }
        const int rc = gemm_group(specs, count, B, workspace, workspace_floats, accumulate, st);
        if (rc != 0) return rc;
    }
    return bgk_launch_status("bgk_linear_weight_grad");
}

extern "C" int64_t bgk_dense_weight_grad_workspace(int64_t B, int32_t P, int32_t n_in) { return bgk_mlp_weight_grad_workspace(B, P, 128, 128, n_in); }

extern "C" int bgk_dense_weight_grad(const float* g_params, int64_t ldg, int32_t P, const float* g_z1, const float* g_z0,
                                     const float* h1, const float* h0, int32_t h_act, const float* cond, int64_t ldc, int32_t d_c,
                                     int32_t periodic, int64_t B, float* workspace, int64_t workspace_floats,
                                     float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0, int32_t accumulate,
                                     const float* g_absmax, void* stream) {
    return bgk_mlp_weight_grad(g_params, ldg, P, g_z1, g_z0, h1, h0, 128, 128, 128, h_act, cond, ldc, d_c, periodic, B, workspace, workspace_floats,
                               gW2, gb2, gW1, gb1, gW0, gb0, accumulate, g_absmax, stream);
}

extern "C" int bgk_dense_weight_grad_reduce_many(int32_t n, const int64_t* B, const int32_t* P, const int32_t* n_in,
                                                 float* const* workspace, float* const* gW2, float* const* gb2, float* const* gW1,
                                                 float* const* gb1, float* const* gW0, float* const* gb0, int32_t accumulate, void* stream) {
    return bgk_mlp_weight_grad_reduce_many(n, B, P, nullptr, nullptr, n_in, workspace, gW2, gb2, gW1, gb1, gW0, gb0, accumulate, stream);
}
