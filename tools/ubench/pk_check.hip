#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../bgflow_amd/csrc/bgk_detmath.h"
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float div_nr(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = n * r;
    float rem = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(rem, r, q);
    return q;
}
__global__ void k(const float* n, const float* d, float* q0, float* q1, float* e0, float* e1, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    q0[i] = n[i] / d[i];
    q1[i] = div_nr(n[i], d[i]);
    e0[i] = bgk_expf(n[i]);
    f2 x = {n[i], n[i]};
    x = __builtin_elementwise_max(x, (f2)(-87.0f)); x = __builtin_elementwise_min(x, (f2)(88.0f));
    const f2 magic = (f2)(12582912.0f);
    f2 t = fma2(x, (f2)(1.44269504088896341f), magic);
    f2 nn = t - magic;
    int nx = (int)(__builtin_bit_cast(uint32_t, t.x) - 0x4B400000u);
    f2 r = fma2(nn, (f2)(-0.693359375f), x); r = fma2(nn, (f2)(2.12194440e-4f), r);
    f2 p = (f2)(1.9875691500e-4f);
    p = fma2(p, r, (f2)(1.3981999507e-3f)); p = fma2(p, r, (f2)(8.3334519073e-3f)); p = fma2(p, r, (f2)(4.1665795894e-2f));
    p = fma2(p, r, (f2)(1.6666665459e-1f)); p = fma2(p, r, (f2)(5.0000001201e-1f));
    f2 r2 = r * r; p = fma2(p, r2, r); p = p + (f2)(1.0f);
    e1[i] = p.x * __builtin_bit_cast(float, (uint32_t)(nx + 127) << 23);
}
int main() {
    const int N = 1 << 24;
    float *n, *d, *q0, *q1, *e0, *e1;
    hipMalloc(&n, N * 4); hipMalloc(&d, N * 4); hipMalloc(&q0, N * 4); hipMalloc(&q1, N * 4); hipMalloc(&e0, N * 4); hipMalloc(&e1, N * 4);
    float* hn = (float*)malloc(N * 4); float* hd = (float*)malloc(N * 4);
    srand(1);
    for (int i = 0; i < N; ++i) { float x = (i & 1) ? (rand() / (float)RAND_MAX) * 0.05f : (rand() / (float)RAND_MAX - 0.5f) * 8.0f; hn[i] = x; hd[i] = 1.0f + bgk_expf(-x); }
    hipMemcpy(n, hn, N * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd, N * 4, hipMemcpyHostToDevice);
    k<<<N / 256, 256>>>(n, d, q0, q1, e0, e1, N);
    float* a = (float*)malloc(N * 4); float* b = (float*)malloc(N * 4);
    hipMemcpy(a, q0, N * 4, hipMemcpyDeviceToHost); hipMemcpy(b, q1, N * 4, hipMemcpyDeviceToHost);
    long dq = 0; for (int i = 0; i < N; ++i) dq += (a[i] != b[i]);
    long dcpu = 0; for (int i = 0; i < N; ++i) dcpu += (a[i] != hn[i] / hd[i]);
    hipMemcpy(a, e0, N * 4, hipMemcpyDeviceToHost); hipMemcpy(b, e1, N * 4, hipMemcpyDeviceToHost);
    long de = 0; for (int i = 0; i < N; ++i) de += (a[i] != b[i]);
    printf("div: rcp-refined vs IEEE mismatches %ld / %d ; IEEE(gpu) vs cpu %ld ; exp packed vs scalar %ld\n", dq, N, dcpu, de);
    return 0;
}
