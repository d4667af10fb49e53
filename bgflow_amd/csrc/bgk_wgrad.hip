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
 * (each wave-level load = 64 consecutive floats of a row: coalesced), splits them into bf16 hi + lo (bf16 keeps the f32
 * exponent: gradients of 1e-8 stay normal; hi*hi + hi*lo + lo*hi leaves a 2^-16 relative error per product, below the f32
 * accumulation noise of a 2^18-term sum), and stores them as ONE 16-byte LDS value per part -- already the MFMA operand
 * layout (lane = (column, 8-row block)); column stride 48 B keeps the ds_read_b128 conflict-free.  LDS is double-buffered
 * (one barrier per step).  The bias gradient is the running sum of the g values a thread loads anyway.  Partials per slab go
 * to a workspace and are summed in fixed order by wgrad_reduce_kernel: deterministic, no atomics.
 */
#include "bgk_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_THREADS = 256;
constexpr int COLS = 128;            /* columns of g (output rows) per workgroup, and max columns of h */
constexpr int CSTRIDE = 48;          /* bytes per column in an LDS operand array: 2 x 16 B (two 8-row blocks) + 16 B pad */
constexpr int ARR = COLS * CSTRIDE;  /* bytes of one operand array */

struct WgArgs {
    const float* g; int64_t ldg; int n;        /* [B, n] */
    const float* h; int64_t ldh; int k;        /* [B, k] (k <= 128) or, featurise != 0, the raw conditioner input [B, k / 2] */
    int featurise;                             /* 1: h columns are cos(2 pi x_c) for c < k/2, sin(2 pi x_c) after (nn/periodic.py:30-37) */
    int64_t B; int64_t rows_per_slab; int n_slabs; int n_blocks;
    float* part_w;                             /* [n_slabs][n][k] */
    float* part_b;                             /* [2 n_slabs][n] */
};

__device__ __forceinline__ unsigned short bf16_rne_bits(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

/* 8 values -> bf16 hi and lo parts (16 B each) */
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    unsigned short h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = bf16_rne_bits(v[e]);
        l[e] = bf16_rne_bits(v[e] - bf16_to_f32(h[e]));
    }
    hi = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
    lo = make_uint4(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16));
}

__global__ __launch_bounds__(WG_THREADS, 3) void wgrad_kernel(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   /* 2 buffers x {g_hi, g_lo, h_hi, h_lo} */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    /* block -> (slab, n-block): blocks that share a slab (and re-read the same rows of h) are 8 apart = on the same XCD */
    const int per8 = 8 * a.n_blocks;
    const int grp = blockIdx.x / per8, rem = blockIdx.x - grp * per8;
    const int slab = grp * 8 + (rem & 7), nb = rem >> 3;
    if (slab >= a.n_slabs) return;
    const int64_t r0 = (int64_t)slab * a.rows_per_slab;
    const int64_t r1 = (r0 + a.rows_per_slab) < a.B ? (r0 + a.rows_per_slab) : a.B;
    const int c = tid & 127, rg = tid >> 7;                 /* my column and 8-row block */
    const int gcol = nb * COLS + c;
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
    const int my_off = c * CSTRIDE + rg * 16;                /* where my 16-byte values go */
    const int rd_a = (wave * 32 + (lane & 31)) * CSTRIDE + (lane >> 5) * 16;
    /* software pipeline: the 16 values of step s + 1 are requested before step s is converted / multiplied, so the HBM latency
     * overlaps the LDS + matrix-core work (without it every 16-row step paid a full round trip: 144 us per GEMM) */
    auto fetch = [&](int64_t r, float (&gv)[8], float (&hv)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t row = r + 8 * rg + e;
            const bool in = row < r1;
            gv[e] = (g_ok && in) ? a.g[row * a.ldg + gcol] : 0.0f;
            hv[e] = (h_ok && in) ? a.h[row * a.ldh + hc] : 0.0f;
        }
    };
    float gn[8], hn[8];
    fetch(r0, gn, hn);
    int buf = 0;
    for (int64_t r = r0; r < r1; r += 16, buf ^= 1) {
        float gv[8], hv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { gv[e] = gn[e]; hv[e] = hn[e]; }
        if (r + 16 < r1) fetch(r + 16, gn, hn);
        if (a.featurise && h_ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sv, cv;
                bgk_sincos2pif(hv[e], &sv, &cv);
                hv[e] = (r + 8 * rg + e < r1) ? (c < kh ? cv : sv) : 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum += gv[e];
        uint4 ghi, glo, hhi, hlo;
        split8(gv, ghi, glo);
        split8(hv, hhi, hlo);
        unsigned char* base = smem + buf * 4 * ARR;
        *reinterpret_cast<uint4*>(base + 0 * ARR + my_off) = ghi;
        *reinterpret_cast<uint4*>(base + 1 * ARR + my_off) = glo;
        *reinterpret_cast<uint4*>(base + 2 * ARR + my_off) = hhi;
        *reinterpret_cast<uint4*>(base + 3 * ARR + my_off) = hlo;
        __syncthreads();
        const s16x8 ahi = *reinterpret_cast<const s16x8*>(base + 0 * ARR + rd_a);
        const s16x8 alo = *reinterpret_cast<const s16x8*>(base + 1 * ARR + rd_a);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < KT) {
                const int rd_b = (m * 32 + (lane & 31)) * CSTRIDE + (lane >> 5) * 16;
                const s16x8 bhi = *reinterpret_cast<const s16x8*>(base + 2 * ARR + rd_b);
                const s16x8 blo = *reinterpret_cast<const s16x8*>(base + 3 * ARR + rd_b);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[m], 0, 0, 0);
            }
        }
        /* the other buffer is written next; its readers finished before the barrier above */
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
                if (row < a.n && col < a.k) pw[(int64_t)row * a.k + col] = acc[m][r];
            }
        }
    }
    if (g_ok) a.part_b[((int64_t)slab * 2 + rg) * a.n + gcol] = bsum;
}

/* fixed-order sum of the slab partials: 8 lanes per output element each sum every 8th slab (4 accumulators), combined
 * by a fixed shuffle tree -- 8x the parallelism of one thread per element (the partial sets are only a few MB: latency-bound) */
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part_w, const float* part_b, int n_slabs, int n, int k,
                                                           float* gW, float* gb, int accumulate) {
    const int64_t nk = (int64_t)n * k;
    const int64_t total = nk + (gb ? n : 0);
    const int sub = threadIdx.x & 7;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    float acc = 0.0f;
    if (i < nk) {
        float a0 = 0.0f, a1 = 0.0f;
        int s = sub;
        for (; s + 8 < n_slabs; s += 16) { a0 += part_w[(int64_t)s * nk + i]; a1 += part_w[(int64_t)(s + 8) * nk + i]; }
        if (s < n_slabs) a0 += part_w[(int64_t)s * nk + i];
        acc = a0 + a1;
    } else if (i < total) {
        const int col = (int)(i - nk);
        float a0 = 0.0f, a1 = 0.0f;
        int s = sub;
        for (; s + 8 < 2 * n_slabs; s += 16) { a0 += part_b[(int64_t)s * n + col]; a1 += part_b[(int64_t)(s + 8) * n + col]; }
        if (s < 2 * n_slabs) a0 += part_b[(int64_t)s * n + col];
        acc = a0 + a1;
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (sub == 0) {
        if (i < nk) gW[i] = accumulate ? gW[i] + acc : acc;
        else if (i < total) gb[i - nk] = accumulate ? gb[i - nk] + acc : acc;
    }
}

int one_gemm(const char* what, const float* g, int64_t ldg, int n, const float* h, int64_t ldh, int k, int featurise, int64_t B,
             float* ws, int64_t ws_floats, float* gW, float* gb, int accumulate, hipStream_t st) {
    BGK_CHECK_ARG(n > 0 && k > 0 && k <= COLS && (!featurise || (k % 2 == 0)), "%s: n = %d, k = %d not supported (k <= 128)", what, n, k);
    const int n_blocks = (n + COLS - 1) / COLS;
    /* ~2 workgroups per CU, slabs of at least 1024 rows, a multiple of 8 slabs (XCD-aware block map) */
    int n_slabs = (int)((B + 1023) / 1024);
    const int want = (512 + n_blocks - 1) / n_blocks;
    n_slabs = n_slabs > want ? want : n_slabs;
    n_slabs = ((n_slabs + 7) / 8) * 8;
    int64_t rows = (B + n_slabs - 1) / n_slabs;
    rows = ((rows + 15) / 16) * 16;
    const int64_t need = (int64_t)n_slabs * n * k + (int64_t)2 * n_slabs * n;
    BGK_CHECK_ARG(ws_floats >= need, "%s: workspace of %lld floats needed, %lld given", what, (long long)need, (long long)ws_floats);
    WgArgs a{g, ldg, n, h, ldh, k, featurise, B, rows, n_slabs, n_blocks, ws, ws + (int64_t)n_slabs * n * k};
    hipLaunchKernelGGL(wgrad_kernel, dim3(n_slabs * n_blocks), dim3(WG_THREADS), 2 * 4 * ARR, st, a);
    const int64_t total = ((int64_t)n * k + n) * 8;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.part_w, a.part_b, n_slabs, n, k, gW, gb, accumulate);
    return 0;
}

}  // namespace

extern "C" int64_t bgk_dense_weight_grad_workspace(int64_t B, int32_t P, int32_t n_in) {
    /* floats: the largest of the three GEMMs' partial sets (they run one after the other on the stream) */
    int64_t best = 0;
    const int ns[3] = {P, 128, 128}, ks[3] = {128, 128, n_in};
    for (int i = 0; i < 3; ++i) {
        const int n_blocks = (ns[i] + COLS - 1) / COLS;
        int n_slabs = (int)((B + 1023) / 1024);
        const int want = (512 + n_blocks - 1) / n_blocks;
        n_slabs = n_slabs > want ? want : n_slabs;
        n_slabs = ((n_slabs + 7) / 8) * 8;
        const int64_t need = (int64_t)n_slabs * ns[i] * ks[i] + (int64_t)2 * n_slabs * ns[i];
        best = need > best ? need : best;
    }
    return best;
}

extern "C" int bgk_dense_weight_grad(const float* g_params, int64_t ldg, int32_t P, const float* g_z1, const float* g_z0,
                                     const float* h1, const float* h0, const float* cond, int64_t ldc, int32_t d_c,
                                     int32_t periodic, int64_t B, float* workspace, int64_t workspace_floats,
                                     float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0, int32_t accumulate, void* stream) {
    BGK_CHECK_ARG(g_params && g_z1 && g_z0 && h1 && h0 && cond && workspace, "bgk_dense_weight_grad: null pointer");
    BGK_CHECK_ARG(B > 0 && P > 0 && d_c > 0, "bgk_dense_weight_grad: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int n_in = periodic ? 2 * d_c : d_c;
    int rc = 0;
    if (gW2) rc = one_gemm("bgk_dense_weight_grad (layer 2)", g_params, ldg, P, h1, 128, 128, 0, B, workspace, workspace_floats, gW2, gb2, accumulate, st);
    if (rc == 0 && gW1) rc = one_gemm("bgk_dense_weight_grad (layer 1)", g_z1, 128, 128, h0, 128, 128, 0, B, workspace, workspace_floats, gW1, gb1, accumulate, st);
    if (rc == 0 && gW0) rc = one_gemm("bgk_dense_weight_grad (layer 0)", g_z0, 128, 128, cond, ldc, n_in, periodic, B, workspace, workspace_floats, gW0, gb0, accumulate, st);
    if (rc != 0) return rc;
    return bgk_launch_status("bgk_dense_weight_grad");
}
