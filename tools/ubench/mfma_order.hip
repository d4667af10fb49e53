// Does the ORDER of the three split-f16 products matter when VALU work is threaded between the MFMAs?  A k-step of the coupling kernel's
// GEMMs is 4 tiles x 3 products = 12 MFMAs.  Order 0 (round-3 kernel): the three products of a tile back to back on the SAME accumulator,
// NV VALU instructions between every two MFMAs.  Order 1: part-major -- lo*hi of the 4 tiles, hi*lo of the 4 tiles, hi*hi of the 4 tiles:
// consecutive MFMAs never write the same accumulator (dependency distance 4).  Order 2: two accumulators alternating (distance 2).
// MI355X_MICROARCH.md: a dependent MFMA not issued back to back with its predecessor pays ~+43 cycles (no accumulator forwarding).
// Fillers: FK = 0 v_fma_f32 (8 independent chains), 1 = every fourth filler a v_exp_f32.   1 and 2 waves per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -o mfma_order mfma_order.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int NV, int FK, int PH>
__device__ __forceinline__ void fill(float (&v)[8], float c1, float c2) {
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        if (FK == 1 && (e & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(e + PH) & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(e + PH) & 7]) : "v"(c1), "v"(c2));
    }
}

template <int ORD, int NV, int FK>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
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
            int m, p;
            if (ORD == 0) { m = ev / 3; p = ev % 3; }
            else if (ORD == 1) { p = ev / 4; m = ev % 4; }
            else { const int g = ev / 6, r = ev % 6; p = r / 2; m = 2 * g + (r & 1); }
            const h16x8 a = p == 0 ? alo : ahi, b = p == 1 ? blo : bhi;
            __builtin_amdgcn_sched_barrier(0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ev % 3 == 0) fill<NV, FK, 0>(v, c1, c2);
            else if (ev % 3 == 1) fill<NV, FK, 3>(v, c1, c2);
            else fill<NV, FK, 5>(v, c1, c2);
        }
    }
    float s = 0.f;
    for (int e = 0; e < 8; ++e) s += v[e];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ORD, int NV, int FK>
float run(float* out, int threads, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ORD, NV, FK>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<ORD, NV, FK>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int NV, int FK>
void row(float* out) {
    const int iters = 4000;     // per wave 48000 MFMAs; at 32 cycles and 2.4 GHz: 0.64 ms (1 wave / SIMD), 1.28 ms (2 waves / SIMD)
    printf("NV %2d FK %d | 1 wave/SIMD: same-acc %.3f  part-major %.3f  pairs %.3f | 2 waves/SIMD: same-acc %.3f  part-major %.3f  pairs %.3f ms\n", NV, FK,
           run<0, NV, FK>(out, 256, iters), run<1, NV, FK>(out, 256, iters), run<2, NV, FK>(out, 256, iters),
           run<0, NV, FK>(out, 512, iters), run<1, NV, FK>(out, 512, iters), run<2, NV, FK>(out, 512, iters));
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    printf("48000 MFMAs per wave; matrix-pipe floor at 2.4 GHz: 0.64 ms (1 wave/SIMD), 1.28 ms (2 waves/SIMD)\n");
    row<0, 0>(out); row<2, 0>(out); row<4, 0>(out); row<5, 0>(out); row<6, 0>(out); row<7, 0>(out); row<8, 0>(out); row<10, 0>(out); row<12, 0>(out); row<16, 0>(out);
    row<4, 1>(out); row<8, 1>(out); row<12, 1>(out);
    return 0;
}
