// Do the matrix pipe and the VALU of one SIMD overlap ACROSS waves on gfx950?  Workgroup of 8 waves = 2 per SIMD (waves w and w + 4
// share SIMD w % 4): role 0 = every wave runs the MFMA chain, 1 = every wave the VALU chain, 2 = waves 0-3 MFMA and waves 4-7 VALU,
// 3 = every wave MFMA + VALU interleaved (8 v_fma per MFMA: 32 cycles each).  Also 4 waves per workgroup (1 per SIMD) for the
// single-wave rates.   build: hipcc --offload-arch=gfx950 -O3 -o cross_wave cross_wave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int NV>   // VALU instructions per MFMA slot
__global__ __launch_bounds__(512) void k(float* out, int iters, int role) {
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    h16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(1.0f + 0.001f * (threadIdx.x & 7) + e); b[e] = (_Float16)(0.5f + e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.1f * e + threadIdx.x * 1e-3f;
    const float c1 = 0.999f, c2 = 1e-3f;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool do_m = role == 0 || role == 3 || (role == 2 && wave < 4);
    const bool do_v = role == 1 || role == 3 || (role == 2 && wave >= 4);
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e & 7]) : "v"(c1), "v"(c2));
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e & 7]) : "v"(c1), "v"(c2));
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 8; ++e) s += v[e];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
float run(float* out, int threads, int iters, int role) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV>), dim3(256), dim3(threads), 0, 0, out, iters, role);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV>), dim3(256), dim3(threads), 0, 0, out, iters, role);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int NV>
void report() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;   // per wave: 80000 MFMA slots
    printf("NV = %d VALU per MFMA slot (%d cycles of VALU against 32 of matrix pipe)\n", NV, 4 * NV);
    printf("  1 wave/SIMD : mfma %.3f  valu %.3f  interleaved %.3f ms\n", run<NV>(out, 256, iters, 0), run<NV>(out, 256, iters, 1), run<NV>(out, 256, iters, 3));
    printf("  2 waves/SIMD: mfma %.3f  valu %.3f  interleaved %.3f  | one wave mfma + other wave valu %.3f ms\n",
           run<NV>(out, 512, iters, 0), run<NV>(out, 512, iters, 1), run<NV>(out, 512, iters, 3), run<NV>(out, 512, iters, 2));
    hipFree(out);
}

int main() { report<4>(); report<8>(); report<12>(); return 0; }
