// What does one MFMA cost the VALU issue port of its SIMD?  mfma_order.hip: with >= 8 v_fma per MFMA the stream is VALU-issue bound and every
// v_mfma_f32_32x32x16_f16 adds ~18 cycles (two waves per SIMD) -- as much as ~7 plain VALU instructions.  Variants of the SAME stream
// (12 events per iteration, NV fillers behind each, consecutive events on different accumulators):
//   0  accumulators in VGPRs (the round-3 kernel)           1  accumulators in AGPRs
//   2  C operand = inline 0 (no accumulator read), VGPR D     3  two 16x16x32 per event, VGPR accumulators (same flops)
//   4  two 16x16x32 per event, AGPR accumulators              5  no MFMA at all (the fillers alone)
//   6  AGPR accumulators and A / B operands in AGPRs
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_cost mfma_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int NV, int PH>
__device__ __forceinline__ void fill(float (&v)[8], float c1, float c2) {
#pragma unroll
    for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(e + PH) & 7]) : "v"(c1), "v"(c2));
}

template <int VAR, int NV>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    f32x16 acc[4];
    f32x4 acc4[8];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 4; ++r) acc4[m][r] = 0.f;
    h16x8 ahi, alo, bhi, blo;
    for (int e = 0; e < 8; ++e) {
        ahi[e] = (_Float16)(1.0f + 0.001f * (threadIdx.x & 7) + e); alo[e] = (_Float16)(0.001f * e);
        bhi[e] = (_Float16)(0.5f + e); blo[e] = (_Float16)(0.002f * e);
    }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.1f * e + threadIdx.x * 1e-3f;
    const float c1 = 0.999f, c2 = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ev = 0; ev < 12; ++ev) {
            const int p = ev / 4, m = ev % 4;
            const h16x8 a = p == 0 ? alo : ahi, b = p == 1 ? blo : bhi;
            __builtin_amdgcn_sched_barrier(0);
            if (VAR == 0) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
            else if (VAR == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
            else if (VAR == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc[m]) : "v"(a), "v"(b));
            else if (VAR == 3) {
                acc4[2 * m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[2 * m], 0, 0, 0);
                acc4[2 * m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc4[2 * m + 1], 0, 0, 0);
            } else if (VAR == 4) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc4[2 * m]) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc4[2 * m + 1]) : "v"(b), "v"(a));
            } else if (VAR == 6) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m]) : "a"(a), "a"(b));
            __builtin_amdgcn_sched_barrier(0);
            if (ev % 3 == 0) fill<NV, 0>(v, c1, c2);
            else if (ev % 3 == 1) fill<NV, 3>(v, c1, c2);
            else fill<NV, 5>(v, c1, c2);
        }
    }
    float s = 0.f;
    for (int e = 0; e < 8; ++e) s += v[e];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 4; ++r) s += acc4[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VAR, int NV>
float run(float* out, int threads, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<VAR, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<VAR, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int NV>
void row(float* out, int threads) {
    const int iters = 4000;
    printf("NV %2d, %d waves/SIMD | vgpr-acc %.3f  agpr-acc %.3f  C=0 %.3f  2x16x16x32 %.3f  2x16x16x32 agpr %.3f  agpr-acc+AB %.3f  fillers only %.3f ms\n", NV, threads / 256,
           run<0, NV>(out, threads, iters), run<1, NV>(out, threads, iters), run<2, NV>(out, threads, iters), run<3, NV>(out, threads, iters),
           run<4, NV>(out, threads, iters), run<6, NV>(out, threads, iters), run<5, NV>(out, threads, iters));
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    printf("48000 events per wave\n");
    row<0>(out, 512); row<4>(out, 512); row<8>(out, 512); row<12>(out, 512); row<16>(out, 512);
    row<8>(out, 256); row<12>(out, 256); row<16>(out, 256);
    row<8>(out, 768); row<12>(out, 768); row<12>(out, 1024);
    return 0;
}
