// One wave per SIMD with TWO 32-sample tiles against two waves per SIMD with one tile each: the event stream of a chunk GEMM of the
// spline coupling kernel in miniature.  Per tile-step: one A fragment pair (hi + lo, 2 KiB per wave, streamed from a 270 KiB weight
// array through a 4-deep register ring), three dependent-free MFMAs per sample tile (4 accumulators per tile, round robin), NV
// independent v_fma per MFMA as the threaded VALU filler.  NB = sample tiles per wave; the launch uses 2 / NB waves per SIMD (the NB = 2
// run asks for 100 KiB of LDS so that only one 4-wave workgroup fits a CU).   build: hipcc --offload-arch=gfx950 -O3 -o tile_pair tile_pair.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int STEPS = 135;            // tile-steps of a layer (32 per GEMM x 4 + layer 0), 2 KiB each = 270 KiB

template <int NB, int NV>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ W, float* out, int tiles_per_wave) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[NB][4];
    for (int n = 0; n < NB; ++n) for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;
    h16x8 bhi[NB], blo[NB];
    for (int n = 0; n < NB; ++n) for (int e = 0; e < 8; ++e) { bhi[n][e] = (_Float16)(0.5f + 0.01f * e + n); blo[n][e] = (_Float16)(0.001f * e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.1f * e + lane * 1e-3f;
    const float c1 = 0.999f, c2 = 1e-3f;
    for (int t = 0; t < tiles_per_wave; ++t) {
        uint4 ring[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ring[u][0] = W[(u * 2 + 0) * 64 + lane]; ring[u][1] = W[(u * 2 + 1) * 64 + lane]; }
        for (int s0 = 0; s0 < STEPS - 3; s0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u;
                const h16x8 ahi = __builtin_bit_cast(h16x8, ring[u][0]), alo = __builtin_bit_cast(h16x8, ring[u][1]);
                const int sn = (s + 4) % STEPS;
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[n][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[n], acc[n][u], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(e + 0) & 7]) : "v"(c1), "v"(c2));
                    acc[n][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[n], acc[n][u], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(e + 3) & 7]) : "v"(c1), "v"(c2));
                    acc[n][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[n], acc[n][u], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(e + 5) & 7]) : "v"(c1), "v"(c2));
                }
                ring[u][0] = W[(sn * 2 + 0) * 64 + lane]; ring[u][1] = W[(sn * 2 + 1) * 64 + lane];
            }
        }
    }
    float s = smem[0] * 0.f;
    for (int e = 0; e < 8; ++e) s += v[e];
    for (int n = 0; n < NB; ++n) for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[n][m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NB, int NV>
float run(const uint4* W, float* out, int tiles_per_wave) {
    const size_t shmem = NB == 2 ? 100 * 1024 : 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NB, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int blocks = 256 * (2 / NB);        // 2 / NB workgroups of 4 waves per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NB, NV>), dim3(blocks), dim3(256), shmem, 0, W, out, tiles_per_wave);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NB, NV>), dim3(blocks), dim3(256), shmem, 0, W, out, tiles_per_wave);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    uint4* W; float* out;
    (void)hipMalloc(&W, STEPS * 2 * 1024); (void)hipMemset(W, 0x3c, STEPS * 2 * 1024);
    (void)hipMalloc(&out, 512 * 256 * 4);
    // same total work in both shapes: 2048 wave-slots x 16 tiles x 1 sample tile, or 1024 wave-slots x 16 tiles x 2 sample tiles
    printf("sample tiles processed per launch: %d, MFMAs per sample tile %d\n", 2048 * 16, 3 * (STEPS - 3));
    printf("NV (v_fma per MFMA) |  2 waves/SIMD x 1 tile  |  1 wave/SIMD x 2 tiles   [ms]\n");
    printf("   0                |  %8.3f               |  %8.3f\n", run<1, 0>(W, out, 16), run<2, 0>(W, out, 16));
    printf("   4                |  %8.3f               |  %8.3f\n", run<1, 4>(W, out, 16), run<2, 4>(W, out, 16));
    printf("   6                |  %8.3f               |  %8.3f\n", run<1, 6>(W, out, 16), run<2, 6>(W, out, 16));
    printf("   8                |  %8.3f               |  %8.3f\n", run<1, 8>(W, out, 16), run<2, 8>(W, out, 16));
    return 0;
}
