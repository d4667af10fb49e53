// Accuracy of f32 GEMM emulation on the bf16 / f16 matrix cores of gfx950 versus the exact-f32 MFMA and an f64 reference.
//   D[32x32] = A[32x128] * B[128x32],  A ~ weights (N(0, 0.1)), B ~ activations (SiLU-like, incl. tiny values)
// build: hipcc --offload-arch=gfx950 -O3 -o split_gemm split_gemm.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ inline float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }

// mode 0: f32 mfma 32x32x2 ; 1: bf16x3 (6 products) ; 2: f16x2 RNE (3 products) ; 3: bf16x3 with 5 products (no mid*mid)
__global__ void k(const float* A, const float* B, float* D, int mode) {
    const int lane = threadIdx.x, i = lane & 31, kb = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k0 = 0; k0 < 128; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 128 + k0 + kb], B[(k0 + kb) * 32 + i], acc, 0, 0, 0);
    } else if (mode == 1 || mode == 3) {
        for (int k0 = 0; k0 < 128; k0 += 16) {
            s16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                float va = A[i * 128 + k0 + 8 * kb + e], vb = B[(k0 + 8 * kb + e) * 32 + i];
                float h = u2f(f2u(va) & 0xffff0000u), r1 = va - h, m = u2f(f2u(r1) & 0xffff0000u), l = r1 - m;
                a[0][e] = (short)(f2u(h) >> 16); a[1][e] = (short)(f2u(m) >> 16); a[2][e] = (short)(f2u(l) >> 16);
                h = u2f(f2u(vb) & 0xffff0000u); r1 = vb - h; m = u2f(f2u(r1) & 0xffff0000u); l = r1 - m;
                b[0][e] = (short)(f2u(h) >> 16); b[1][e] = (short)(f2u(m) >> 16); b[2][e] = (short)(f2u(l) >> 16);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
            if (mode == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    } else {
        for (int k0 = 0; k0 < 128; k0 += 16) {
            h16x8 a[2], b[2];
            for (int e = 0; e < 8; ++e) {
                float va = A[i * 128 + k0 + 8 * kb + e] * 256.0f, vb = B[(k0 + 8 * kb + e) * 32 + i];
                _Float16 h = (_Float16)va; a[0][e] = h; a[1][e] = (_Float16)(va - (float)h);
                h = (_Float16)vb; b[0][e] = h; b[1][e] = (_Float16)(vb - (float)h);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) acc[r] *= (1.0f / 256.0f);
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = acc[r];
}

int main() {
    const int NT = 64;   // independent random problems
    float *A = (float*)malloc(32 * 128 * 4), *B = (float*)malloc(128 * 32 * 4), *D = (float*)malloc(32 * 32 * 4);
    float *dA, *dB, *dD; hipMalloc(&dA, 32 * 128 * 4); hipMalloc(&dB, 128 * 32 * 4); hipMalloc(&dD, 32 * 32 * 4);
    double err[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0}, scale = 0; long n = 0;
    srand(1);
    for (int t = 0; t < NT; ++t) {
        for (int q = 0; q < 32 * 128; ++q) {
            double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
            A[q] = (float)(0.1 * sqrt(-2 * log(u1)) * cos(6.283185307 * u2));
            double x = 3.0 * sqrt(-2 * log(u2)) * cos(6.283185307 * u1);
            B[q] = (float)(x / (1 + exp(-x)));                      // SiLU-like: many small negatives, some tiny
            if (t % 4 == 3 && q % 7 == 0) B[q] *= 1e-6f;              // sprinkle tiny magnitudes (f16 denormal range)
        }
        hipMemcpy(dA, A, 32 * 128 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B, 128 * 32 * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode);
            hipMemcpy(D, dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double ref = 0, sab = 0; for (int kk = 0; kk < 128; ++kk) { ref += (double)A[i * 128 + kk] * B[kk * 32 + j]; sab += fabs((double)A[i * 128 + kk] * B[kk * 32 + j]); }
                double e = fabs(D[i * 32 + j] - ref) / sab;   // error relative to sum |a||b|
                err[mode] += e * e; if (e > mx[mode]) mx[mode] = e;
                if (mode == 0) { n++; }
            }
        }
    }
    const char* nm[4] = {"f32 mfma (fmaf chain)", "bf16x3, 6 products", "f16x2 RNE, 3 products", "bf16x3, 5 products"};
    for (int m = 0; m < 4; ++m) printf("%-24s  rms err / sum|ab| = %.3e (%.2f ulp32)   max = %.3e (%.2f ulp32)\n", nm[m], sqrt(err[m] / n), sqrt(err[m] / n) / 5.96e-8, mx[m], mx[m] / 5.96e-8);
    return 0;
}
