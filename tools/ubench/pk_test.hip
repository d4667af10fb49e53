// micro-benchmark: scalar vs packed (v_pk_*) f32 VALU for the deterministic exp + division chains
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../bgflow_amd/csrc/bgk_detmath.h"
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 exp2v(f2 x) {
    x = __builtin_elementwise_max(x, (f2)(-87.0f));
    x = __builtin_elementwise_min(x, (f2)(88.0f));
    const f2 magic = (f2)(12582912.0f);
    f2 t = fma2(x, (f2)(1.44269504088896341f), magic);
    f2 n = t - magic;
    int nx = (int)(__builtin_bit_cast(uint32_t, t.x) - 0x4B400000u), ny = (int)(__builtin_bit_cast(uint32_t, t.y) - 0x4B400000u);
    f2 r = fma2(n, (f2)(-0.693359375f), x);
    r = fma2(n, (f2)(2.12194440e-4f), r);
    f2 p = (f2)(1.9875691500e-4f);
    p = fma2(p, r, (f2)(1.3981999507e-3f));
    p = fma2(p, r, (f2)(8.3334519073e-3f));
    p = fma2(p, r, (f2)(4.1665795894e-2f));
    p = fma2(p, r, (f2)(1.6666665459e-1f));
    p = fma2(p, r, (f2)(5.0000001201e-1f));
    f2 r2 = r * r;
    p = fma2(p, r2, r);
    p = p + (f2)(1.0f);
    f2 sc; sc.x = __builtin_bit_cast(float, (uint32_t)(nx + 127) << 23); sc.y = __builtin_bit_cast(float, (uint32_t)(ny + 127) << 23);
    return p * sc;
}
// correctly rounded n/d for safe-range operands: hardware rcp seed + fma refinement (no div_scale / div_fixup)
__device__ __forceinline__ f2 div2(f2 n, f2 d) {
    f2 r; r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
    f2 e = fma2(-d, r, (f2)(1.0f));
    r = fma2(e, r, r);
    f2 q = n * r;
    f2 rem = fma2(-d, q, n);
    q = fma2(rem, r, q);
    rem = fma2(-d, q, n);
    q = fma2(rem, r, q);
    return q;
}
template <int MODE>
__global__ void k(const float* in, float* out, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[tid * 16 + i];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { float x = v[i]; v[i] = x / (1.0f + bgk_expf(-x)) + 0.01f; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 x = {v[i], v[i + 1]};
                f2 e = exp2v(-x);
                f2 q = div2(x, (f2)(1.0f) + e);
                v[i] = q.x + 0.01f; v[i + 1] = q.y + 0.01f;
            }
        }
    }
    for (int i = 0; i < 16; ++i) acc += v[i];
    out[tid] = acc;
}
int main() {
    const int N = 256 * 8 * 256;   // threads
    float *in, *out, *out2;
    hipMalloc(&in, N * 16 * 4); hipMalloc(&out, N * 4); hipMalloc(&out2, N * 4);
    float* h = (float*)malloc(N * 16 * 4);
    for (int i = 0; i < N * 16; ++i) h[i] = ((i * 7919) % 2001 - 1000) / 250.0f;
    hipMemcpy(in, h, N * 16 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<N / 256, 256>>>(in, out, 50); else k<1><<<N / 256, 256>>>(in, out2, 50);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.3f ms  (%.2f G silu/s)\n", mode, ms, (double)N * 16 * 50 / ms / 1e6);
        }
    }
    float* a = (float*)malloc(N * 4); float* b = (float*)malloc(N * 4);
    hipMemcpy(a, out, N * 4, hipMemcpyDeviceToHost); hipMemcpy(b, out2, N * 4, hipMemcpyDeviceToHost);
    long diff = 0; for (int i = 0; i < N; ++i) diff += (a[i] != b[i]);
    printf("bitwise mismatches scalar-IEEE vs packed-rcp: %ld of %d\n", diff, N);
    return 0;
}
