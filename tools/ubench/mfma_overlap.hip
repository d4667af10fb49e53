// Does MFMA work overlap with f32 VALU work on gfx950?  (bf16 32x32x16 and f32 32x32x2, same wave / two waves per SIMD)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int KIND>   // MODE bit0: mfma, bit1: valu ; KIND 0: bf16 32x32x16, 1: f32 32x32x2
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + threadIdx.x + e); b[e] = (short)(0x3f00 + e); }
    float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
    float v0 = threadIdx.x * 0.001f, v1 = 0.1f, v2 = 0.2f, v3 = 0.3f, v4 = .4f, v5 = .5f, v6 = .6f, v7 = .7f;
    float c1 = 0.999f + threadIdx.x * 1e-9f, c2 = 1e-3f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7}, pc1 = {c1, c1}, pc2 = {c2, c2};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE & 1) {
                if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
                else acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[u], 0, 0, 0);
            }
            if (MODE & 2) {
                // 8 independent fma chains: 8 VALU instr per MFMA (32 cycles worth for bf16 mfma, half of the f32 mfma)
                if (MODE & 4) {   // packed: 4 v_pk_fma_f32
                    asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(pc1), "v"(pc2));
                    asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p2), "+v"(p3) : "v"(pc1), "v"(pc2));
                } else {          // scalar: 8 v_fma_f32
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c1), "v"(c2));
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c1), "v"(c2));
                }
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int KIND>
float run(float* out, int blocks, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {   // waves per SIMD
        int threads = 256 * wps / 1;       // 4 SIMDs x wps waves
        int blocks = 256;
        printf("waves/SIMD=%d\n", wps);
        printf("  bf16 mfma only            %.3f ms\n", run<1, 0>(out, blocks, threads, iters));
        printf("  8 v_fma only              %.3f ms\n", run<2, 0>(out, blocks, threads, iters));
        printf("  4 v_pk_fma only           %.3f ms\n", run<6, 0>(out, blocks, threads, iters));
        printf("  bf16 mfma + 8 v_fma       %.3f ms\n", run<3, 0>(out, blocks, threads, iters));
        printf("  bf16 mfma + 4 v_pk_fma    %.3f ms\n", run<7, 0>(out, blocks, threads, iters));
        printf("  f32 mfma only             %.3f ms\n", run<1, 1>(out, blocks, threads, iters));
        printf("  f32 mfma + 8 v_fma        %.3f ms\n", run<3, 1>(out, blocks, threads, iters));
        printf("  f32 mfma + 4 v_pk_fma     %.3f ms\n", run<7, 1>(out, blocks, threads, iters));
    }
    // cycles: per iteration 4 mfma (bf16: 4*32 = 128 cyc; f32: 4*64 = 256 cyc) and 32 valu (128 cyc)
    return 0;
}
