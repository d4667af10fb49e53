/* bgk_dense_bwd.hip -- input-gradient chain of the conditioner MLP in one launch (training step, autograd of
 * nn/dense.py:47-48 behind ConditionalSplineTransformer, nn/flow/transformer/spline.py:109):
 *   g_h1 = g_params W2            g_z1 = g_h1 * act'(z1)      h1 = act(z1)
 *   g_h0 = g_z1 W1                g_z0 = g_h0 * act'(z0)      h0 = act(z0)
 *   g_feat = g_z0 W0              g_cond = featuriser^T g_feat  (cos / sin featuriser of nn/periodic.py:30-37 or identity)
 * replacing three tall-skinny hipBLASLt GEMMs, two activation recomputations and two activation-backward passes per
 * layer; the weight / bias gradients stay GEMMs over the batch on the tensors written here (g_z1, g_z0, h1, h0).
 * Same split-f16 machinery as the forward (bgk_mfma_h2.h): a wave owns 32 samples; the B operand of the first GEMM is
 * read straight from the row-major g_params (8 consecutive floats per lane and k16-step), later ones are the previous
 * GEMM's accumulator registers.  A operands = transposed weights packed by bgk_pack_dense_h2_t.
 * Roofline: HBM 4 (P + 4*128 + 2*128 + d_c) B per sample (cfg 3 B|A: 4.8 kB); hidden activations' derivative on the
 * hardware exp / rcp forms like the forward.
 */
#include "bgk_mfma_h2.h"

namespace {

#include "bgk_dma.h"

#ifndef BGK_DBWD_DW
#define BGK_DBWD_DW 4
#endif
constexpr int DW = BGK_DBWD_DW;      /* waves (32-row tiles) per workgroup */
constexpr int DSROW = 33;
constexpr int DBWD_PAD = 4;               /* T2 holds a multiple of 4 k-steps (zero blocks behind ceil(P / 16)): part of the ABI */
constexpr int DBWD_DG = DBWD_PAD;         /* a.S2 is a multiple of this: the first GEMM runs whole groups of DBWD_GD k-steps */
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned bgk_u4 __attribute__((ext_vector_type(4)));

struct DenseBwdArgs {
    const float* g; int64_t ldg; int P;           /* gradient w.r.t. the MLP output [B, P] */
    const float* z1; const float* z0;             /* saved pre-activations [B, 128] */
    const float* cond; int64_t ldc; int d_c; int periodic;
    const uint4 *T2, *T1, *T0; int S2;            /* transposed-weight operands; S2 = ceil(P / 16) rounded up to a multiple of 4 */
    const float* cs;                              /* {2^s, 2^-s} x 3 (layer 0, 1, 2) */
    int act; int64_t B;
    float* g_z1; float* g_z0; float* h1; float* h0;
    float* g_cond; int64_t ldgc;
    const float* g_cond_add; int64_t ldga;        /* NULL, or a [B, d_c] tensor added to g_cond on the way out (may BE g_cond: accumulation in place) */
    int lds_per_wave;
    const float* g_absmax;                        /* [1] largest |g| of the launch (device; NULL: g enters the first GEMM unscaled) */
    float* gz_absmax;                             /* [2] raised to the largest |g_z1|, |g_z0| written (NULL: not wanted) */
};

/* -DBGK_SBD_TS=1: s_memtime stamps of the wave's phases (14: first GEMM done, 19 - 29: the chain), written over row b0 of g_z0 at the end */
#ifndef BGK_SBD_TS
#define BGK_SBD_TS 0
#endif
#ifndef BGK_DBWD_DRAIN
#define BGK_DBWD_DRAIN 0      /* experiment: 1 drain the queue behind the z requests, 2 behind the g_z stores, 3 both */
#endif
#if BGK_SBD_TS
#define SBD_TS(k) do { __builtin_amdgcn_sched_barrier(0); if (lane == 0) reinterpret_cast<unsigned*>(s_f)[(k) * H2_SLAB + 130] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SBD_TS(k) do { } while (0)
#endif

/* d = g * act'(z), h = act(z) for a pair (hardware exp / rcp) */
__device__ __forceinline__ void act_grad2(int act, bgk_f2 z, bgk_f2 g, bgk_f2& gz, bgk_f2& h) {
    if (act == 1) {
        const bgk_f2 y = z * bgk_splat2(-1.44269504088896341f);
        bgk_f2 e; e.x = __builtin_amdgcn_exp2f(y.x); e.y = __builtin_amdgcn_exp2f(y.y);
        e = e + bgk_splat2(1.0f);
        bgk_f2 s; s.x = __builtin_amdgcn_rcpf(e.x); s.y = __builtin_amdgcn_rcpf(e.y);
        h = z * s;
        gz = g * (s * (bgk_splat2(1.0f) + z * (bgk_splat2(1.0f) - s)));
    } else if (act == 2) {
        h.x = z.x > 0.0f ? z.x : 0.0f; h.y = z.y > 0.0f ? z.y : 0.0f;
        gz.x = z.x > 0.0f ? g.x : 0.0f; gz.y = z.y > 0.0f ? g.y : 0.0f;
    } else {
        h = bgk_tanhf2_fast(z);
        gz = g * (bgk_splat2(1.0f) - h * h);
    }
}

/* acc (accumulator layout, 4 tiles) -> g_z = acc * c * act'(z), h = act(z); z is read as 16-byte groups, g_z / h leave as
 * full rows through the LDS slab.  All sixteen z requests go out BEFORE the arithmetic (64 registers): left to the compiler
 * they were issued a few at a time between the activation code, one exposed memory round trip after the other -- 50 k of the
 * 170 k cycles a wave lived (s_memtime stamps, -DBGK_SBD_TS=1), 19 k with the requests up front. */
/* ZT: tiles of 32 hidden units the z / g_z / h arrays hold per row (row pitch 32 ZT floats; 4 = the [B, 128] arrays of the spline layers,
 * 2 = [B, 64]: an affine coupling's 64-unit networks); the tiles past ZT belong to units that do not exist: their gradient is 0 */
template <int ZT>
__device__ __forceinline__ void act_backward_tiles(h2_f32x16 (&t)[4], float c, int act, const float* z, float* gz_out, float* h_out,
                                                   float* s_buf, int64_t b0, int lane, int rows, int tsb = 22) {
    float* const s_f = s_buf;
    (void)s_f; (void)tsb;
    const int j = lane & 31, hh = lane >> 5;
    const int64_t row = (b0 + (j < rows ? j : 0)) * (32 * ZT);
    float4 zall[4 * ZT];
#pragma unroll
    for (int m = 0; m < ZT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) zall[4 * m + q] = *reinterpret_cast<const float4*>(z + row + 32 * m + 8 * q + 4 * hh);
    __builtin_amdgcn_sched_barrier(0);
#if BGK_SBD_TS || BGK_DBWD_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SBD_TS(tsb);
#endif
#pragma unroll
    for (int m = ZT; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[m][r] = 0.0f;
#pragma unroll
    for (int m = 0; m < ZT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 zz = zall[4 * m + q];
            bgk_f2 g0, g1, a0, a1;
            act_grad2(act, (bgk_f2){zz.x, zz.y}, (bgk_f2){t[m][4 * q] * c, t[m][4 * q + 1] * c}, g0, a0);
            act_grad2(act, (bgk_f2){zz.z, zz.w}, (bgk_f2){t[m][4 * q + 2] * c, t[m][4 * q + 3] * c}, g1, a1);
            t[m][4 * q] = g0.x; t[m][4 * q + 1] = g0.y; t[m][4 * q + 2] = g1.x; t[m][4 * q + 3] = g1.y;
        }
#if BGK_SBD_TS
    asm volatile("" :: "v"(t[0][0]), "v"(t[3][15]), "v"(t[1][7]), "v"(t[2][9]));
    SBD_TS(tsb + 1);
#endif
    if (h_out) {      /* the activations themselves (NULL: the weight-gradient kernel recomputes act(z) while loading z) */
        h2_f32x16 hv[4];
#pragma unroll
        for (int m = 0; m < ZT; ++m) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hv[m][4 * q] = zall[4 * m + q].x; hv[m][4 * q + 1] = zall[4 * m + q].y;
                hv[m][4 * q + 2] = zall[4 * m + q].z; hv[m][4 * q + 3] = zall[4 * m + q].w;
            }
            h2_act_tile(hv[m], 1.0f, act);
        }
        h2_store_rows<ZT>(hv, h_out, s_buf, b0, rows, lane);
    }
    h2_store_rows<ZT>(t, gz_out, s_buf, b0, rows, lane);
#if BGK_SBD_TS || (BGK_DBWD_DRAIN & 2)
    SBD_TS(tsb + 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SBD_TS(tsb + 3);
#endif
}

/* everything behind the first GEMM: acc = W2^T (g * 2^s) (weights' own scale still on) -> g_z1, g_z0 (+ h1, h0), g_cond.
 * The gradient tiles that feed the next GEMM are split into f16 hi + lo parts under a power-of-two scale taken from the TILE's own
 * largest magnitude (a wave-wide maximum: 32 v_max3 + 6 cross-lane steps per GEMM): whatever the loss scale and the batch size
 * (g ~ 1 / B), every product carries 22 significant bits like the forward's.  inv_g: reciprocal of the first GEMM's scale. */
template <int FT, int ZT>
__device__ __forceinline__ void dx_chain_tail(const DenseBwdArgs& a, h2_f32x16 (&acc)[4], float inv_g, float* s_f, int64_t b0, int lane, int rows) {
    const int j = lane & 31, hh = lane >> 5;
    const float c2 = a.cs[5], c1 = a.cs[3], c0 = a.cs[1];
    act_backward_tiles<ZT>(acc, c2 * inv_g, a.act, a.z1, a.g_z1, a.h1, s_f, b0, lane, rows);
    SBD_TS(19);

    /* ---- g_h0 = W1^T g_z1 ---- */
    H2B<4> bf;
    float inv1, inv0;
    {
        const float m1 = h2_wave_absmax<4>(acc);
        if (a.gz_absmax && lane == 0) h2_publish_absmax(a.gz_absmax, m1);
        h2_make_b_scaled<4>(bf, acc, h2_pow2_scale(m1, inv1));
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    h2_gemm_hidden<4, 4>(acc, bf, a.T1, lane);
#if BGK_SBD_TS
    asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][15]));
#endif
    SBD_TS(20);
    act_backward_tiles<ZT>(acc, c1 * inv1, a.act, a.z0, a.g_z0, a.h0, s_f, b0, lane, rows, 26);
    SBD_TS(21);
    {
        const float m0 = h2_wave_absmax<4>(acc);
        if (a.gz_absmax && lane == 0) h2_publish_absmax(a.gz_absmax + 1, m0);
        /* ---- g_feat = W0^T g_z0, then the featuriser's transpose ---- */
        if (a.g_cond == nullptr) return;
        h2_make_b_scaled<4>(bf, acc, h2_pow2_scale(m0, inv0));
    }
    h2_f32x16 gf[FT];
#pragma unroll
    for (int m = 0; m < FT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) gf[m][r] = 0.0f;
    h2_gemm_hidden<FT, 4>(gf, bf, a.T0, lane);
    if (!a.periodic) {
        if (j < rows) {
            float* orow = a.g_cond + (b0 + j) * a.ldgc;
            const float* arow = a.g_cond_add ? a.g_cond_add + (b0 + j) * a.ldga : nullptr;
#pragma unroll
            for (int m = 0; m < FT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = h2_row(m, r, hh);
                    if (f < a.d_c) orow[f] = gf[m][r] * (c0 * inv0) + (arow ? arow[f] : 0.0f);
                }
        }
    } else {
#pragma unroll
        for (int m = 0; m < FT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_f[h2_row(m, r, hh) * DSROW + j] = gf[m][r] * (c0 * inv0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        /* feats = [cos 2 pi x, sin 2 pi x]  ->  g_x = 2 pi (cos * g_sin - sin * g_cos) */
        for (int i = lane; i < rows * a.d_c; i += 64) {
            const int r = i / a.d_c, c = i - r * a.d_c;
            float sv, cv;
            bgk_sincos2pif(a.cond[(b0 + r) * a.ldc + c], &sv, &cv);
            const float gc = s_f[c * DSROW + r], gs = s_f[(a.d_c + c) * DSROW + r];
            const float prev = a.g_cond_add ? a.g_cond_add[(b0 + r) * a.ldga + c] : 0.0f;
            a.g_cond[(b0 + r) * a.ldgc + c] = 6.28318530717958648f * (cv * gs - sv * gc) + prev;
        }
    }
}

/* First GEMM (g_h1 = W2^T g), LDS plan.  Measured alternatives: profiles/r05_dx_phase_ts.txt (operands streamed from L2 by every wave;
 * the gradient in 32-byte pieces straight into registers; groups of four k-steps; rings 2 / 3; eight waves per workgroup; the loop
 * without the interleave below). */
#ifndef BGK_DBWD_GRING
#define BGK_DBWD_GRING 2       /* gradient groups in flight + in use per wave (LDS ring slots of 4 KB) */
#endif
#ifndef BGK_DBWD_ORING
#define BGK_DBWD_ORING 3       /* operand groups in flight + in use per workgroup (LDS ring slots of 16 KB) */
#endif
constexpr int DBWD_GD = 2;                                           /* k-steps per group */
constexpr int DBWD_OPG = DBWD_GD * 4 * 2 * 64;                        /* 16-byte pieces of an operand group */
constexpr size_t DBWD_GLDS_BYTES = BGK_DBWD_ORING * (size_t)DBWD_OPG * 16 + (size_t)DW * BGK_DBWD_GRING * 4096;
template <int N> __device__ __forceinline__ void dx_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

/* A-operand fragments of one k-step of the first GEMM, read from LDS by inline asm (see the kernel) */
typedef unsigned bgk_u4v __attribute__((ext_vector_type(4)));
struct DxFrag { bgk_u4v r[4][2]; };
__device__ __forceinline__ void dx_frag_wait(DxFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r[0][0]), "+v"(f.r[0][1]), "+v"(f.r[1][0]), "+v"(f.r[1][1]),
                                           "+v"(f.r[2][0]), "+v"(f.r[2][1]), "+v"(f.r[3][0]), "+v"(f.r[3][1]) : : "memory");
}
__device__ __forceinline__ void dx_frag_get(const DxFrag& f, H2A<4>& h) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) h.v[m][pp] = make_uint4(f.r[m][pp][0], f.r[m][pp][1], f.r[m][pp][2], f.r[m][pp][3]);
}

template <int FT, int ZT = 4>
__global__ __launch_bounds__(DW * 64, 8 / DW) void dense_bwd_dx_kernel(DenseBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   /* uniform: buffer descriptor in SGPRs */
    const int j = lane & 31, hh = lane >> 5;
    float* s_f = smem + (size_t)wave * a.lds_per_wave;       /* [32][H2_SLAB] output slab; later the g_feat tile [32 FT][DSROW] */
    const int64_t n_tiles = (a.B + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * DW + wave;
    const bool active = tile < n_tiles;                      /* a wave without a tile still copies its share of the operands and meets the barriers */
    const int64_t b0 = (active ? tile : n_tiles - 1) * 32;
    const int rows = (int)((a.B - b0) < 32 ? (a.B - b0) : 32);
#if BGK_SBD_TS
    const unsigned ts0 = (unsigned)__builtin_amdgcn_s_memtime();       /* (written behind the first GEMM: the slab space holds its operands until then) */
#endif

    /* ---- g_h1 = W2^T g : B operand straight from the row-major gradient ---- */
    h2_f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    float sg, inv_sg;
    {
        /* descriptor of the tile's gradient rows: rows past the batch (and columns past the allocation) are out of range */
        const __amdgpu_buffer_rsrc_t rs_gt = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + b0 * a.ldg), 0, (int)(rows * a.ldg * 4), 0x00020000);
        const int P = a.P;
        const int S2 = a.S2;
        sg = h2_pow2_scale(a.g_absmax ? a.g_absmax[0] : 0.0f, inv_sg);      /* per-tensor power of two: max |g| -> [2^14, 2^15) */
        /* Round 5.  The operand blocks of this GEMM (W2^T as f16 hi + lo: 8 KB per k-step, 229 KB per tile at P = 425) are the same for every
         * tile: the four waves of the workgroup copy a group of GD = 2 k-steps (16 KB) into an LDS ring together -- each wave a quarter, by
         * DMA, in the slab space the chain behind this GEMM uses afterwards (a barrier separates the two uses) -- and read their A fragments
         * from there (streamed from L2 by every wave, each k-step exposed most of an L2 round trip: 76 k of the 147 k cycles a wave lived).
         * The gradient tile goes through LDS as well: loaded straight into registers as the B operand, a load instruction handed the address
         * unit 64 separate 16-byte pieces (32 rows x 2), one piece per cycle, eight waves per CU -- 29 k of the GEMM's 71 k cycles were
         * request ISSUE.  Per group a wave copies its tile's 32 rows x 128 bytes by DMA, eight adjacent lanes per row piece, into a private
         * ring, and reads its B operand (row j, k = 16 s + 8 hh ..) from there; the 16-byte pieces of a row are stored XOR-swizzled by the
         * row so that the 64 lanes' reads spread over the banks.  Bounds come from the buffer descriptor; what out-of-range pieces leave in
         * LDS (zeros) and the columns >= P are masked at the point of USE (a select right behind a load is a wait for the whole batch in
         * front of the current group's matrix work: the operand blocks there are 0 too, but 0 * NaN is not).  All LDS reads of this GEMM
         * are inline asm: behind an LDS-DMA the compiler's wait insertion puts vmcnt(0) in front of every LDS read it sees (possible
         * alias), i.e. in front of the current group's work it waited for the requests of the NEXT groups. */
        constexpr int GD = DBWD_GD, OPG = DBWD_OPG, GRING = BGK_DBWD_GRING, ORING = BGK_DBWD_ORING, PDG = GRING - 1, PDO = ORING - 1;
        constexpr int NG = 4, NO = OPG / DW / 64;                        /* DMA instructions of a wave per group: gradient | operands */
        /* Requests complete in order.  A group needs its gradient (requested PDG groups ago) and its operands (PDO groups ago); the kind
         * with the SHORTER distance is requested first inside a group, so that behind the younger of the two needed batches there are
         * only batches of later groups: REMAIN of them may still be in flight at the top of a group. */
        constexpr bool G_FIRST = PDO >= PDG;
        constexpr int REMAIN = PDO == PDG ? (PDG - 1) * (NG + NO) : PDO > PDG ? NO + (PDG - 1) * (NG + NO) : NG + (PDO - 1) * (NG + NO);
        static_assert(DBWD_DG % GD == 0 && GRING >= 2 && ORING >= 2 && REMAIN < 64, "S2 is a multiple of DBWD_DG");
        const int ngroups = S2 / GD;
        char* const smem_b = reinterpret_cast<char*>(smem);
        const unsigned lds0 = (unsigned)(uintptr_t)(lvp_t)smem;
        const unsigned lds_a = lds0 + (unsigned)lane * 16u;                                   /* + buffer * OPG * 16 + u * 8192 + immediate */
        const int gring0 = ORING * OPG * 16 + wave * (GRING * 4096);                               /* byte offset of this wave's gradient ring */
        /* DMA source: instruction i = rows 8 i .. 8 i + 7, lane -> row 8 i + (lane >> 3), piece (lane & 7) ^ (row & 7) [row & 7 == lane >> 3] */
        const int gvo = (lane >> 3) * (int)a.ldg * 4 + (((lane & 7) ^ (lane >> 3)) * 16);
        const int gstep = 8 * (int)a.ldg * 4;
        /* read addresses: row j of the ring slot, pieces (4 u + 2 hh + e) ^ (j & 7) */
        const unsigned grow = lds0 + (unsigned)gring0 + (unsigned)(j * 128);
        const int gpj = (2 * hh) ^ (j & 7);
        auto dma_op = [&](int gi, int buf) {
            const uint4* src = a.T2 + (size_t)gi * OPG + wave * (OPG / DW);
            char* dst = smem_b + (buf * OPG + wave * (OPG / DW)) * 16;
#pragma unroll
            for (int i = 0; i < NO; ++i)
                __builtin_amdgcn_global_load_lds((gvp_t)(src + i * 64 + lane), (lvp_t)(dst + i * 1024), 16, 0, 0);
        };
        auto dma_grad = [&](int gi, int slot) {
            char* dst = smem_b + gring0 + slot * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_gt, (lvp_t)(dst + i * 1024), 16, gvo, gi * (GD * 64) + i * gstep, 0, 0);
        };
        /* An asm load's destination counts as written where the statement ends: the wait statement names the registers ("+v"), so
         * that every use -- a copy the register allocator might place included -- is ordered behind the wait. */
        auto lds_frag = [&](DxFrag& f, int buf, int u) {
            const unsigned addr = lds_a + (unsigned)buf * (OPG * 16u) + (unsigned)u * (4 * 2 * 1024u);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f.r[m][pp]) : "v"(addr), "i"((m * 2 + pp) * 1024) : "memory");
        };
        const int Pj = j < rows ? P : 0;                                 /* rows past the batch: every column masked */
        /* (past the end the last group is requested again, into ring slots nobody reads any more: the request counts stay uniform) */
        /* The operands of a group are used one turn of the loop after its gradient (LAG = 1: the loop below multiplies group g - 1 while it
         * splits group g), so a turn requests the operands of group t + PDO - 1; the distances between request and use -- and with
         * them REMAIN -- are the same. */
        constexpr int LAG = 1;
        static_assert(!LAG || (G_FIRST && PDO >= 2), "pipelined form: gradient requests first, an operand slot for the group in use");
        auto request_g = [&](int t, int gslot) { if (t + PDG >= 0) dma_grad(t + PDG < ngroups ? t + PDG : ngroups - 1, gslot); };
        auto request_o = [&](int t, int oslot) { if (t + PDO - LAG >= 0) dma_op(t + PDO - LAG < ngroups ? t + PDO - LAG : ngroups - 1, oslot); };
        auto request = [&](int t, int gslot, int oslot) {                /* what turn t requests: gradient t + PDG, operands t + PDO - LAG */
            if (G_FIRST) { request_g(t, gslot); request_o(t, oslot); }
            else { request_o(t, oslot); request_g(t, gslot); }
        };
        constexpr int PDM = PDG > PDO ? PDG : PDO;
#pragma unroll
        for (int t = -PDM; t < 0; ++t) request(t, (t + PDG + GRING) % GRING, (t + PDO - LAG + 2 * ORING) % ORING);
        __builtin_amdgcn_sched_barrier(0);
#if BGK_SBD_TS
        unsigned tsw = 0u, tsb = 0u, tsc = 0u, tss = 0u, tsi = 0u, tq0, tq1, tq3, tq4, tq2 = (unsigned)__builtin_amdgcn_s_memtime();
#define SBD_Q(v) do { __builtin_amdgcn_sched_barrier(0); v = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
        /* the gradient tile of group g out of ring slot `slot`: raw 16-byte pieces (asm reads, see lds_frag) */
        static_assert(GD == 2, "the wait names the four registers");
        auto read_raw = [&](bgk_u4v (&raw)[GD][2], int slot) {
            const unsigned so = (unsigned)slot * 4096u;
#pragma unroll
            for (int u = 0; u < GD; ++u)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const unsigned ad = grow + so + (unsigned)((gpj ^ (4 * u + e)) * 16);
                    asm volatile("ds_read_b128 %0, %1" : "=&v"(raw[u][e]) : "v"(ad) : "memory");
                }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]) : : "memory");
        };
        auto split_step = [&](const bgk_u4v (&raw)[GD][2], int g, int u, h2_h16x8& hi, h2_h16x8& lo) {
            float v[8];
            const int k0 = 16 * (g * GD + u) + 8 * hh;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned w = raw[u][e >> 2][e & 3];                /* (by value: __builtin_bit_cast of a vector ELEMENT picked element 0 every time) */
                v[e] = k0 + e < Pj ? __uint_as_float(w) : 0.0f;
            }
            h2_split8_scaled(v, sg, hi, lo);
        };
        /* Turn g of the loop: [wait, barrier] the gradient tile of group g out of LDS, then the 24 MFMAs of group g - 1, one by one, with
         * the VALU work of group g's f16 split and the eight DMA requests of the turn between them (sched_group_barrier: 1 MFMA, 5 VALU,
         * a request behind every third) -- in the sequential form the matrix pipe idled through 11 k cycles of split and 10 - 13 k cycles
         * of request issue per tile (stamps), a wave's instructions being issued in order. */
        int slot = 0, slot_in = PDG % GRING, ob = (ORING - LAG) % ORING, ob_in = (PDO - LAG) % ORING;
        h2_h16x8 phi[GD], plo[GD];
        auto turn_done = [&]() {
            slot = slot + 1 == GRING ? 0 : slot + 1;
            slot_in = slot_in + 1 == GRING ? 0 : slot_in + 1;
            ob = ob + 1 == ORING ? 0 : ob + 1;
            ob_in = ob_in + 1 == ORING ? 0 : ob_in + 1;
        };
        {   /* turn 0: nothing to multiply yet */
            dx_wait_vm<REMAIN>();
            bgk_u4v raw[GD][2];
            read_raw(raw, slot);
#pragma unroll
            for (int u = 0; u < GD; ++u) split_step(raw, 0, u, phi[u], plo[u]);
            __builtin_amdgcn_sched_barrier(0);
            request(0, slot_in, ob_in);
            turn_done();
        }
#if BGK_SBD_TS
        tq2 = (unsigned)__builtin_amdgcn_s_memtime();
#endif
        for (int gi = 1; gi < ngroups; ++gi) {
#if BGK_SBD_TS
            SBD_Q(tq0); tsc += tq0 - tq2;
#endif
            dx_wait_vm<REMAIN>();                                       /* the gradient tile of group gi and this wave's share of the operands of group gi - 1 have landed */
#if BGK_SBD_TS
            SBD_Q(tq1); tsw += tq1 - tq0;
#endif
            __syncthreads();                                            /* ... everyone's share; and everyone has left the slot this turn's operand request goes to */
#if BGK_SBD_TS
            SBD_Q(tq2); tsb += tq2 - tq1;
#endif
            bgk_u4v raw[GD][2];
            read_raw(raw, slot);
            h2_h16x8 nhi[GD], nlo[GD];
            DxFrag fr[2];
            lds_frag(fr[0], ob, 0);
#pragma unroll
            for (int u = 0; u < GD; ++u) {
                dx_frag_wait(fr[u & 1]);                                /* this step's fragments (requested one step ago) have arrived */
                if (u + 1 < GD) lds_frag(fr[(u + 1) & 1], ob, u + 1);
                __builtin_amdgcn_sched_barrier(0);
                H2A<4> fa;
                dx_frag_get(fr[u & 1], fa);
                h2_mfma3<4>(acc, fa, phi[u], plo[u]);
                split_step(raw, gi, u, nhi[u], nlo[u]);
                if (u == 0) request_g(gi, slot_in); else request_o(gi, ob_in);
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    if (q % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < GD; ++u) { phi[u] = nhi[u]; plo[u] = nlo[u]; }
            turn_done();
        }
        {   /* the last group's MFMAs */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            DxFrag fr[2];
            lds_frag(fr[0], ob, 0);
#pragma unroll
            for (int u = 0; u < GD; ++u) {
                dx_frag_wait(fr[u & 1]);
                if (u + 1 < GD) lds_frag(fr[(u + 1) & 1], ob, u + 1);
                __builtin_amdgcn_sched_barrier(0);
                H2A<4> fa;
                dx_frag_get(fr[u & 1], fa);
                h2_mfma3<4>(acc, fa, phi[u], plo[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#if BGK_SBD_TS
        SBD_Q(tq0); tsc += tq0 - tq2;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                /* the copies past the end */
        __syncthreads();                                                /* the buffers are the waves' output slabs from here on */
        if (!active) return;
#if BGK_SBD_TS
        if (lane == 0) { unsigned* q = reinterpret_cast<unsigned*>(s_f); q[1 * H2_SLAB + 130] = tsw; q[2 * H2_SLAB + 130] = tsb; q[3 * H2_SLAB + 130] = tsc; q[4 * H2_SLAB + 130] = tss; q[5 * H2_SLAB + 130] = tsi; }
#endif
    }
#if BGK_SBD_TS
    if (lane == 0) reinterpret_cast<unsigned*>(s_f)[0 * H2_SLAB + 130] = ts0;
#endif
    SBD_TS(14);
    dx_chain_tail<FT, ZT>(a, acc, inv_sg, s_f, b0, lane, rows);
#if BGK_SBD_TS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SBD_TS(15);
    if (lane == 0)
        for (int k = 0; k < 32; ++k) reinterpret_cast<unsigned*>(a.g_z0)[b0 * (32 * ZT) + k] = reinterpret_cast<unsigned*>(s_f)[k * H2_SLAB + 130];
#endif
}

/* transposed-weight operand blocks: M[i][k] = W[ksrc(k)][i] * scale, i = output row of the backward GEMM (= input feature of
 * the layer), k in natural order (natural = 1: ksrc = k) or in accumulator order (ksrc = hidden unit of slot k) */
struct PackT {
    const float* W; int rows_src, cols_src;   /* the layer's weight [rows_src = out features, cols_src = in features] */
    int NT, S, natural;
    _Float16* out;
};

/* the three layers in one launch: workgroups [first[q], first[q + 1]) pack layer q */
struct PackTGroup { PackT L[3]; int first[4]; };

__device__ __forceinline__ void pack_t_body(const PackTGroup& g, const float* cs, int bx) {
    const int layer = bx >= g.first[2] ? 2 : (bx >= g.first[1] ? 1 : 0);
    const PackT& L = g.L[layer];
    const int blocks = L.S * L.NT * 2 + L.NT;
    const int64_t t = (int64_t)(bx - g.first[layer]) * 256 + threadIdx.x;
    if (t >= (int64_t)blocks * 64) return;
    const int lane = (int)(t & 63), blk = (int)(t >> 6);
    const int i = lane & 31, kb = lane >> 5;
    const float scale = cs[2 * layer];
    _Float16 o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)0.0f;
    if (blk < L.S * L.NT * 2) {
        const int p = blk & 1, m = (blk >> 1) % L.NT, s = (blk >> 1) / L.NT;
        const int col = 32 * m + i;                                  /* input feature of the layer */
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = L.natural ? 16 * s + 8 * kb + e : 32 * (s >> 1) + (e & 3) + 8 * (2 * (s & 1) + (e >> 2)) + 4 * kb;
            float v = 0.0f;
            if (k < L.rows_src && col < L.cols_src) v = L.W[(int64_t)k * L.cols_src + col] * scale;
            const _Float16 h = (_Float16)v;
            o[e] = p ? (_Float16)(v - (float)h) : h;
        }
    }   /* else: zero "bias" blocks (the backward GEMMs have no bias) */
    *reinterpret_cast<uint4*>(L.out + ((int64_t)blk * 64 + lane) * 8) = *reinterpret_cast<const uint4*>(o);
}

__global__ __launch_bounds__(256) void pack_t_kernel(PackTGroup g, const float* cs) { pack_t_body(g, cs, (int)blockIdx.x); }

/* the transposed operands of several conditioners in one launch (bgk_pack_dense_h2_t_many) */
constexpr int PACKT_MANY = 16;
struct PackTOne { PackTGroup g; const float* cs; };
struct PackTMany { PackTOne c[PACKT_MANY]; };
__global__ __launch_bounds__(256) void pack_t_many_kernel(PackTMany) {
    const PackTMany* km = (const PackTMany*)__builtin_amdgcn_kernarg_segment_ptr();     /* run-time indexed: read in place */
    const int ci = blockIdx.y;
    if ((int)blockIdx.x >= km->c[ci].g.first[3]) return;
    pack_t_body(km->c[ci].g, km->c[ci].cs, (int)blockIdx.x);
}

}  // namespace

/* H0 / H1: hidden widths of the source weights (W0 [H0, n_in], W1 [H1, H0], W2 [P, H1]); units past them are zero in the operands */
static int pack_t_fill(PackTGroup& g, const float* W0, int n_in, const float* W1, const float* W2, int P, void* T0, void* T1,
                       void* T2, int H0 = 128, int H1 = 128) {
    const int FT = (n_in + 31) / 32;
    const int S2 = ((P + 15) / 16 + DBWD_PAD - 1) / DBWD_PAD * DBWD_PAD;
    const PackT L2{W2, P, H1, 4, S2, 1, (_Float16*)T2};   /* M[hidden i][k] = W2[col(k)][i], col = output column of the MLP */
    const PackT L1{W1, H1, H0, 4, 8, 0, (_Float16*)T1};      /* M[i][k] = W1[unit(k)][i] */
    const PackT L0{W0, H0, n_in, FT, 8, 0, (_Float16*)T0};    /* M[feature i][k] = W0[unit(k)][i] */
    g.L[0] = L0; g.L[1] = L1; g.L[2] = L2;
    int n_wg = 0;
    for (int l = 0; l < 3; ++l) {
        const int64_t total = (int64_t)(g.L[l].S * g.L[l].NT * 2 + g.L[l].NT) * 64;
        g.first[l] = n_wg;
        n_wg += (int)((total + 255) / 256);
    }
    g.first[3] = n_wg;
    return n_wg;
}

static int pack_t_launch(const float* W0, int n_in, const float* W1, const float* W2, int P, const float* cs, void* T0, void* T1,
                         void* T2, hipStream_t st, int H0 = 128, int H1 = 128) {
    PackTGroup g;
    const int n_wg = pack_t_fill(g, W0, n_in, W1, W2, P, T0, T1, T2, H0, H1);
    hipLaunchKernelGGL(pack_t_kernel, dim3((unsigned)n_wg), dim3(256), 0, st, g, cs);
    return 0;
}

/* bgk_pack_mlp_h2_t of n conditioners in one launch per 16 (after an optimizer step: every coupling layer of the flow); H0 / H1 may be
 * NULL: 128 hidden units everywhere (= bgk_pack_dense_h2_t_many) */
extern "C" int bgk_pack_mlp_h2_t_many(int32_t n, const float* const* W0, const int32_t* n_in, const int32_t* H0, const float* const* W1,
                                      const int32_t* H1, const float* const* W2, const int32_t* P, const float* const* cs,
                                      void* const* T0, void* const* T1, void* const* T2, void* stream) {
    BGK_CHECK_ARG(n >= 0 && W0 && n_in && W1 && W2 && P && cs && T0 && T1 && T2, "bgk_pack_mlp_h2_t_many: null pointer");
    if (n == 0) return 0;       /* nothing to do (and no launch status to ask a GPU-less box for) */
    for (int base = 0; base < n; base += PACKT_MANY) {
        const int cnt = n - base < PACKT_MANY ? n - base : PACKT_MANY;
        PackTMany M;
        int max_wg = 0;
        for (int c = 0; c < cnt; ++c) {
            const int i = base + c;
            const int h0 = H0 ? H0[i] : 128, h1 = H1 ? H1[i] : 128;
            BGK_CHECK_ARG(W0[i] && W1[i] && W2[i] && cs[i] && T0[i] && T1[i] && T2[i] && n_in[i] > 0 && n_in[i] <= 96 && P[i] > 0
                          && h0 > 0 && h0 <= 128 && h1 > 0 && h1 <= 128, "bgk_pack_mlp_h2_t_many: bad conditioner %d", i);
            const int n_wg = pack_t_fill(M.c[c].g, W0[i], n_in[i], W1[i], W2[i], P[i], T0[i], T1[i], T2[i], h0, h1);
            M.c[c].cs = cs[i];
            max_wg = n_wg > max_wg ? n_wg : max_wg;
        }
        for (int c = cnt; c < PACKT_MANY; ++c) M.c[c] = M.c[0];
        hipLaunchKernelGGL(pack_t_many_kernel, dim3((unsigned)max_wg, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, M);
    }
    return bgk_launch_status("bgk_pack_mlp_h2_t_many");
}

extern "C" int bgk_pack_dense_h2_t_many(int32_t n, const float* const* W0, const int32_t* n_in, const float* const* W1,
                                        const float* const* W2, const int32_t* P, const float* const* cs,
                                        void* const* T0, void* const* T1, void* const* T2, void* stream) {
    return bgk_pack_mlp_h2_t_many(n, W0, n_in, nullptr, W1, nullptr, W2, P, cs, T0, T1, T2, stream);
}

extern "C" int bgk_pack_dense_h2_t(const float* W0, int32_t n_in, const float* W1, const float* W2, int32_t P,
                                   const float* cs, void* T0, void* T1, void* T2, void* stream) {
    BGK_CHECK_ARG(W0 && W1 && W2 && cs && T0 && T1 && T2, "bgk_pack_dense_h2_t: null pointer");
    BGK_CHECK_ARG(n_in > 0 && n_in <= 96 && P > 0, "bgk_pack_dense_h2_t: bad sizes (n_in <= 96)");
    pack_t_launch(W0, n_in, W1, W2, P, cs, T0, T1, T2, (hipStream_t)stream);
    return bgk_launch_status("bgk_pack_dense_h2_t");
}

/* bgk_pack_dense_h2_t for hidden layers of H0 / H1 <= 128 units (the backward kernels run 128: the operands' other units are zero) */
extern "C" int bgk_pack_mlp_h2_t(const float* W0, int32_t n_in, int32_t H0, const float* W1, int32_t H1, const float* W2, int32_t P,
                                 const float* cs, void* T0, void* T1, void* T2, void* stream) {
    BGK_CHECK_ARG(W0 && W1 && W2 && cs && T0 && T1 && T2, "bgk_pack_mlp_h2_t: null pointer");
    BGK_CHECK_ARG(n_in > 0 && n_in <= 96 && P > 0 && H0 > 0 && H0 <= 128 && H1 > 0 && H1 <= 128, "bgk_pack_mlp_h2_t: bad sizes (n_in <= 96, H <= 128)");
    pack_t_launch(W0, n_in, W1, W2, P, cs, T0, T1, T2, (hipStream_t)stream, H0, H1);
    return bgk_launch_status("bgk_pack_mlp_h2_t");
}

/* ldz: row pitch of z1 / z0 / g_z1 / g_z0 / h1 / h0 -- 128, or 64 for networks whose hidden layers have <= 64 units (the operands'
 * other units are zero: bgk_pack_mlp_h2_t) */
extern "C" int bgk_mlp_backward_dx(const float* g, int64_t ldg, int32_t P, const float* z1, const float* z0, int64_t ldz,
                                   const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                   const void* T0, const void* T1, const void* T2, const float* cs, int32_t act,
                                   int64_t B, float* g_z1, float* g_z0, float* h1, float* h0,
                                   float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga,
                                   const float* g_absmax, float* gz_absmax, void* stream) {
    if (B == 0) return 0;       /* an empty batch: nothing to do (its tensors have no storage, hence null pointers) */
    BGK_CHECK_ARG(g && z1 && z0 && T0 && T1 && T2 && cs && g_z1 && g_z0, "bgk_dense_backward_dx: null pointer");
    BGK_CHECK_ARG((h1 == nullptr) == (h0 == nullptr), "bgk_dense_backward_dx: h1 and h0 are written both or not at all");
    BGK_CHECK_ARG(B >= 0 && P > 0 && ldg >= P && d_c > 0 && act >= 1 && act <= 3 && (ldz == 128 || ldz == 64), "bgk_dense_backward_dx: bad sizes (ldz: 64 | 128)");
    BGK_CHECK_ARG(!(g_cond && periodic && !cond), "bgk_dense_backward_dx: the periodic featuriser needs the conditioner input");
    const int n_in = periodic ? 2 * d_c : d_c;
    if (n_in > 96) { bgk_set_error("bgk_dense_backward_dx: %d input features > 96", n_in); return BGK_EUNSUPPORTED; }
    if (B == 0) return 0;
    DenseBwdArgs a;
    a.g = g; a.ldg = ldg; a.P = P; a.z1 = z1; a.z0 = z0; a.cond = cond; a.ldc = ldc; a.d_c = d_c; a.periodic = periodic;
    a.T2 = (const uint4*)T2; a.T1 = (const uint4*)T1; a.T0 = (const uint4*)T0; a.S2 = ((P + 15) / 16 + DBWD_PAD - 1) / DBWD_PAD * DBWD_PAD; a.cs = cs; a.act = act; a.B = B;
    a.g_z1 = g_z1; a.g_z0 = g_z0; a.h1 = h1; a.h0 = h0; a.g_cond = g_cond; a.ldgc = ldgc;
    a.g_absmax = g_absmax; a.gz_absmax = gz_absmax;
    a.g_cond_add = g_cond ? g_cond_add : nullptr; a.ldga = ldga;
    BGK_CHECK_ARG(!a.g_cond_add || ldga >= d_c, "bgk_dense_backward_dx: bad row stride of g_cond_add");
    const int FT = (n_in + 31) / 32;
    a.lds_per_wave = 32 * H2_SLAB > 32 * FT * DSROW ? 32 * H2_SLAB : 32 * FT * DSROW;   /* output slab, reused for the g_feat tile */
    size_t shmem = sizeof(float) * (size_t)DW * a.lds_per_wave;
    static_assert(DBWD_GLDS_BYTES * (8 / DW) <= 160 * 1024, "eight waves per CU");
    if (shmem < DBWD_GLDS_BYTES) shmem = DBWD_GLDS_BYTES;            /* the operand ring + the waves' gradient rings (first GEMM) */
    const int64_t n_wg = ((B + 31) / 32 + DW - 1) / DW;
    BGK_CHECK_ARG(n_wg < (int64_t)0x7fffffff, "bgk_dense_backward_dx: batch too large for one launch");
    hipStream_t st = (hipStream_t)stream;
#define BGK_LAUNCH_DXZ(F, Z) do { if (shmem > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_bwd_dx_kernel<F, Z>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((dense_bwd_dx_kernel<F, Z>), dim3((int)n_wg), dim3(DW * 64), shmem, st, a); } while (0)
#define BGK_LAUNCH_DX(F) do { if (ldz == 64) BGK_LAUNCH_DXZ(F, 2); else BGK_LAUNCH_DXZ(F, 4); } while (0)
    if (FT == 1) BGK_LAUNCH_DX(1);
    else if (FT == 2) BGK_LAUNCH_DX(2);
    else BGK_LAUNCH_DX(3);
#undef BGK_LAUNCH_DX
#undef BGK_LAUNCH_DXZ
    return bgk_launch_status("bgk_dense_backward_dx");
}

extern "C" int bgk_dense_backward_dx(const float* g, int64_t ldg, int32_t P, const float* z1, const float* z0,
                                     const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                     const void* T0, const void* T1, const void* T2, const float* cs, int32_t act,
                                     int64_t B, float* g_z1, float* g_z0, float* h1, float* h0,
                                     float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga,
                                     const float* g_absmax, float* gz_absmax, void* stream) {
    return bgk_mlp_backward_dx(g, ldg, P, z1, z0, 128, cond, ldc, d_c, periodic, T0, T1, T2, cs, act, B, g_z1, g_z0, h1, h0, g_cond, ldgc,
                               g_cond_add, ldga, g_absmax, gz_absmax, stream);
}
