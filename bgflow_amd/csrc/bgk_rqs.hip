/* bgk_rqs.hip -- rational-quadratic spline transformer, conditioner output read from HBM.
 *
 * HBM-bound kernel: per (sample, dim) it consumes 3K(+1) parameter floats and 1 input float and
 * produces 1 output float (+ 1/d of a dlogp float): 4*(P + 2d + 2) algorithmic bytes per sample.
 *
 * Work decomposition (gfx950): a workgroup of 256 threads (4 waves) owns a tile of TS consecutive
 * samples.  The tile's parameter rows are contiguous in HBM ([TS, P] floats), so they are streamed
 * with 16-byte-per-lane coalesced loads into LDS (row stride padded to an odd number of dwords).
 * Elements are then assigned lane -> sample (e % TS), so the K raw widths/heights/slopes of a lane
 * sit at LDS stride (odd) -> bank-conflict-free ds_read_b32.  Per-element log-dets go through LDS
 * and are summed per sample in ascending-dim order (bit-reproducible, same order as the oracle).
 */
#include "bgk_common.h"

namespace {

constexpr int RQS_THREADS = 256;

struct RqsArgs {
    const float* y; int64_t ldy;
    const float* params; int64_t ldp;
    const int32_t* nc_slot;
    int64_t B; int d; int K; int P; int inverse;
    float* out; int64_t ldo;
    float* dlogp; int accumulate;
    int32_t* bin_idx; int32_t* oob_count;
    int TS;      /* samples per tile */
    int Pp;      /* padded LDS row stride (odd) */
    uint32_t magicP; /* ceil(2^32 / P) for idx / P */
    BgkRqsCfg cfg;
};

template <int KT>
__global__ __launch_bounds__(RQS_THREADS) void rqs_kernel(RqsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TS = a.TS, d = a.d, P = a.P, Pp = a.Pp;
    const int K = KT ? KT : a.K;
    float* s_par = smem;                    /* [TS][Pp] */
    float* s_lad = s_par + TS * Pp;         /* [TS][d]  */
    float* s_out = s_lad + TS * d;          /* [TS][d]  */
    __shared__ int s_oob;
    const int tid = threadIdx.x;
    const int64_t n_tiles = (a.B + TS - 1) / TS;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int rows = (int)((a.B - b0) < TS ? (a.B - b0) : TS);
        if (tid == 0) s_oob = 0;
        /* ---- stage the parameter rows ---- */
        if (a.ldp == P && ((((uintptr_t)(a.params + b0 * a.ldp)) & 15) == 0)) {
            const float4* src = reinterpret_cast<const float4*>(a.params + b0 * a.ldp);
            const int n4 = (rows * P) >> 2;
            for (int q = tid; q < n4; q += RQS_THREADS) {
                float4 v = src[q];
                uint32_t i = (uint32_t)q << 2;
                uint32_t r = __umulhi(i, a.magicP);
                uint32_t cidx = i - r * (uint32_t)P;
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s_par[r * Pp + cidx] = vv[u];
                    ++cidx;
                    if (cidx == (uint32_t)P) { cidx = 0; ++r; }
                }
            }
            for (int i = (n4 << 2) + tid; i < rows * P; i += RQS_THREADS) {
                uint32_t r = __umulhi((uint32_t)i, a.magicP);
                s_par[r * Pp + (i - r * P)] = a.params[b0 * a.ldp + i];
            }
        } else {
            for (int i = tid; i < rows * P; i += RQS_THREADS) {
                uint32_t r = __umulhi((uint32_t)i, a.magicP);
                uint32_t cidx = i - r * P;
                s_par[r * Pp + cidx] = a.params[(b0 + r) * a.ldp + cidx];
            }
        }
        /* ---- stage y (coalesced along the row) into s_out ---- */
        for (int i = tid; i < rows * d; i += RQS_THREADS) {
            int r = i / d, j = i - r * d;
            s_out[i] = a.y[(b0 + r) * a.ldy + j];
        }
        __syncthreads();
        /* ---- elements: lane -> sample ---- */
        int oob_local = 0;
        for (int e = tid; e < TS * d; e += RQS_THREADS) {
            const int j = e / TS, s = e - j * TS;
            if (s < rows) {
                const float* row = s_par + s * Pp;
                const float* pw = row + j * K;
                const float* ph = row + d * K + j * K;
                const float* ps = row + 2 * d * K + j * K;
                const int slot = a.nc_slot[j];
                const float s_last = slot >= 0 ? row[3 * d * K + slot] : ps[0];
                float lad; int bin, oob;
                float x = s_out[s * d + j];
                float o = bgk_rqs_element<KT>(x, pw, ph, ps, 1, s_last, K, a.inverse, a.cfg, &lad, &bin, &oob);
                s_out[s * d + j] = o;
                s_lad[s * d + j] = lad;
                oob_local += oob;
                if (a.bin_idx) a.bin_idx[(b0 + s) * d + j] = bin;
            }
        }
        if (oob_local) atomicAdd(&s_oob, oob_local);
        __syncthreads();
        /* ---- write back: outputs coalesced along the row, dlogp summed in ascending dim order ---- */
        for (int i = tid; i < rows * d; i += RQS_THREADS) {
            int r = i / d, j = i - r * d;
            a.out[(b0 + r) * a.ldo + j] = s_out[i];
        }
        for (int s = tid; s < rows; s += RQS_THREADS) {
            float acc = 0.0f;
            for (int j = 0; j < d; ++j) acc += s_lad[s * d + j];
            if (a.accumulate) a.dlogp[b0 + s] += acc; else a.dlogp[b0 + s] = acc;
        }
        if (tid == 0 && s_oob && a.oob_count) atomicAdd(a.oob_count, s_oob);
        __syncthreads();
    }
}

/* ---- K = 8 streaming variant: no parameter staging --------------------------------------------------------
 * Elements are enumerated dim-fastest (e = s * d + j), so the 8 raw widths (heights, slopes) of 64 consecutive
 * lanes are 64 consecutive 32-byte runs of a parameter row: every lane fetches its 24 parameters with six 16-byte
 * loads straight into registers (4-byte aligned is enough for global_load_dwordx4), y / out / bin indices are
 * plain coalesced accesses, and only the per-element log-dets pass through LDS for the ascending-dim row sum.
 * No barrier between load and compute, ~100 VGPRs -> 4-5 waves / SIMD hide the HBM latency. */
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int RQS2_TS = 128;   /* samples per workgroup tile */

__global__ __launch_bounds__(RQS_THREADS) void rqs_stream_kernel(RqsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_lad[];   /* [RQS2_TS * d] */
    constexpr int K = 8;
    const int d = a.d, tid = threadIdx.x;
    const uint64_t magicd = (0x100000000ull + (uint64_t)d - 1) / (uint64_t)d;   /* 2^32 for d = 1: needs 33 bits */
    const int64_t n_tiles = (a.B + RQS2_TS - 1) / RQS2_TS;
    int oob_local = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * RQS2_TS;
        const int rows = (int)((a.B - b0) < RQS2_TS ? (a.B - b0) : RQS2_TS);
        const int n = rows * d;
        for (int e = tid; e < n; e += RQS_THREADS) {
            const int s = (int)(((uint64_t)(uint32_t)e * magicd) >> 32), j = e - s * d;
            const float* row = a.params + (b0 + s) * a.ldp;
            const float* gw = row + j * K;
            const float* gh = gw + d * K;
            const float* gs = gh + d * K;
            float pw[K], ph[K], ps[K];
            const f4u w0 = *reinterpret_cast<const f4u*>(gw), w1 = *reinterpret_cast<const f4u*>(gw + 4);
            const f4u h0 = *reinterpret_cast<const f4u*>(gh), h1 = *reinterpret_cast<const f4u*>(gh + 4);
            const f4u t0 = *reinterpret_cast<const f4u*>(gs), t1 = *reinterpret_cast<const f4u*>(gs + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { pw[k] = w0[k]; pw[4 + k] = w1[k]; ph[k] = h0[k]; ph[4 + k] = h1[k]; ps[k] = t0[k]; ps[4 + k] = t1[k]; }
            const int slot = a.nc_slot[j];
            const float s_last = slot >= 0 ? row[3 * d * K + slot] : ps[0];
            const float x = a.y[(b0 + s) * a.ldy + j];
            float lad; int bin, oob;
            const float o = bgk_rqs_element<K, true>(x, pw, ph, ps, 1, s_last, K, a.inverse, a.cfg, &lad, &bin, &oob);
            a.out[(b0 + s) * a.ldo + j] = o;
            if (a.bin_idx) a.bin_idx[(b0 + s) * d + j] = bin;
            s_lad[e] = lad;
            oob_local += oob;
        }
        __syncthreads();
        for (int s = tid; s < rows; s += RQS_THREADS) {
            float acc = 0.0f;
            for (int j = 0; j < d; ++j) acc += s_lad[s * d + j];
            if (a.accumulate) a.dlogp[b0 + s] += acc; else a.dlogp[b0 + s] = acc;
        }
        __syncthreads();
    }
    if (a.oob_count) {
        for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
        if ((tid & 63) == 0 && oob_local) atomicAdd(a.oob_count, oob_local);
    }
}

/* ---- any bin count / any parameter-row length: no LDS staging of the parameter rows --------------------------------------------
 * rqs_kernel<0> needs at least four [P]-float rows in LDS (K <= 64 for the cfg-3 shapes).  Beyond that every lane walks the 3 K (+1)
 * parameters of ITS element straight from memory: elements are enumerated dim-fastest, so the K-float runs of 64 consecutive lanes
 * are 64 consecutive runs of a parameter row -- the k-th access of a wave touches addresses 4 K bytes apart (one cache line per 128
 * / (4 K) lanes), every line is used completely over the K accesses and stays in L1 meanwhile.  Not a roofline kernel (the walk is
 * repeated for max / sum / knots): it keeps wide splines off device torch ops. ---- */
__global__ __launch_bounds__(RQS_THREADS) void rqs_direct_kernel(RqsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_lad[];   /* [RQS2_TS * d] */
    const int d = a.d, K = a.K, tid = threadIdx.x;
    const uint64_t magicd = (0x100000000ull + (uint64_t)d - 1) / (uint64_t)d;
    const int64_t n_tiles = (a.B + RQS2_TS - 1) / RQS2_TS;
    int oob_local = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t b0 = tile * RQS2_TS;
        const int rows = (int)((a.B - b0) < RQS2_TS ? (a.B - b0) : RQS2_TS);
        const int n = rows * d;
        for (int e = tid; e < n; e += RQS_THREADS) {
            const int s = (int)(((uint64_t)(uint32_t)e * magicd) >> 32), j = e - s * d;
            const float* row = a.params + (b0 + s) * a.ldp;
            const float* pw = row + (int64_t)j * K;
            const float* ph = pw + (int64_t)d * K;
            const float* ps = ph + (int64_t)d * K;
            const int slot = a.nc_slot[j];
            const float s_last = slot >= 0 ? row[(int64_t)3 * d * K + slot] : ps[0];
            const float x = a.y[(b0 + s) * a.ldy + j];
            float lad; int bin, oob;
            const float o = bgk_rqs_element<0>(x, pw, ph, ps, 1, s_last, K, a.inverse, a.cfg, &lad, &bin, &oob);
            a.out[(b0 + s) * a.ldo + j] = o;
            if (a.bin_idx) a.bin_idx[(b0 + s) * d + j] = bin;
            s_lad[e] = lad;
            oob_local += oob;
        }
        __syncthreads();
        for (int s = tid; s < rows; s += RQS_THREADS) {
            float acc = 0.0f;
            for (int j = 0; j < d; ++j) acc += s_lad[s * d + j];
            if (a.accumulate) a.dlogp[b0 + s] += acc; else a.dlogp[b0 + s] = acc;
        }
        __syncthreads();
    }
    if (a.oob_count) {
        for (int off = 32; off > 0; off >>= 1) oob_local += __shfl_xor(oob_local, off);
        if ((tid & 63) == 0 && oob_local) atomicAdd(a.oob_count, oob_local);
    }
}

}  // namespace

extern "C" int bgk_rqs_transform(const float* y, int64_t ldy, const float* params, int64_t ldp,
                                 int32_t P, const int32_t* nc_slot, int64_t B, int32_t d, int32_t K,
                                 int32_t inverse, double left, double right, double bottom,
                                 double top, double min_bin_width, double min_bin_height,
                                 double min_derivative, int32_t identity_init, float* out,
                                 int64_t ldo, float* dlogp, int32_t accumulate, int32_t* bin_idx,
                                 int32_t* oob_count, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(B >= 0 && d > 0 && K > 0, "bgk_rqs_transform: bad sizes B=%lld d=%d K=%d", (long long)B, d, K);
    BGK_CHECK_ARG(y && params && nc_slot && out && dlogp, "bgk_rqs_transform: null pointer");
    BGK_CHECK_ARG(min_bin_width * K <= 1.0 && min_bin_height * K <= 1.0,
                  "Minimal bin width/height too large for the number of bins");
    if (B == 0) return 0;
    RqsArgs a;
    a.y = y; a.ldy = ldy; a.params = params; a.ldp = ldp; a.nc_slot = nc_slot;
    a.B = B; a.d = d; a.K = K; a.inverse = inverse;
    a.out = out; a.ldo = ldo; a.dlogp = dlogp; a.accumulate = accumulate;
    a.bin_idx = bin_idx; a.oob_count = oob_count;
    a.P = P;
    BGK_CHECK_ARG(P >= 3 * K * d && P <= 3 * K * d + d && ldp >= P,
                  "bgk_rqs_transform: params width %d (ld %lld) not in [3Kd, 3Kd+d] for d=%d K=%d", P, (long long)ldp, d, K);
    a.cfg = bgk_make_rqs_cfg(left, right, bottom, top, min_bin_width, min_bin_height, min_derivative, identity_init, K);
    a.Pp = a.P | 1;
    a.magicP = (uint32_t)((0x100000000ull + (uint64_t)a.P - 1) / (uint64_t)a.P);
    /* tile size: largest multiple of 4 samples whose LDS footprint stays under ~52 KiB (3 workgroups
     * of 4 waves per CU), capped at 64 */
    const size_t budget = 52 * 1024;
    int TS = (int)(budget / (sizeof(float) * (size_t)(a.Pp + 2 * d)));
    TS = TS > 64 ? 64 : TS;
    TS &= ~3;
    if (TS < 4 || K > 64) {     /* parameter rows beyond the LDS tile (K > 64 for ~17 dims): every lane walks its element's parameters in memory */
        BGK_CHECK_ARG((size_t)RQS2_TS * d * sizeof(float) <= 64 * 1024, "bgk_rqs_transform: %d dims do not fit the log-det tile", d);
        const int64_t nt2 = (B + RQS2_TS - 1) / RQS2_TS;
        const int grid2 = (int)(nt2 < 256 * 16 ? nt2 : 256 * 16);
        hipLaunchKernelGGL(rqs_direct_kernel, dim3(grid2), dim3(RQS_THREADS), sizeof(float) * (size_t)RQS2_TS * d, (hipStream_t)stream, a);
        return bgk_launch_status("bgk_rqs_transform");
    }
    a.TS = TS;
    size_t shmem = sizeof(float) * (size_t)TS * (size_t)(a.Pp + 2 * d);
    int64_t n_tiles = (B + TS - 1) / TS;
    int grid = (int)(n_tiles < 256 * 12 ? n_tiles : 256 * 12);
    hipStream_t st = (hipStream_t)stream;
    if (K == 8 && d <= 1024) {
        const int64_t nt2 = (B + RQS2_TS - 1) / RQS2_TS;
        const int grid2 = (int)(nt2 < 256 * 16 ? nt2 : 256 * 16);
        hipLaunchKernelGGL(rqs_stream_kernel, dim3(grid2), dim3(RQS_THREADS), sizeof(float) * (size_t)RQS2_TS * d, st, a);
    } else if (K == 8) hipLaunchKernelGGL(rqs_kernel<8>, dim3(grid), dim3(RQS_THREADS), shmem, st, a);
    else hipLaunchKernelGGL(rqs_kernel<0>, dim3(grid), dim3(RQS_THREADS), shmem, st, a);
    return bgk_launch_status("bgk_rqs_transform");
}
