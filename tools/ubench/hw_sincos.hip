// accuracy of v_sin_f32 / v_cos_f32 (argument in revolutions: sin(2 pi x)) on gfx950 against double precision, x in [0, 1) and [-2, 2]
//   build: hipcc --offload-arch=gfx950 -O3 -o hw_sincos hw_sincos.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = __builtin_amdgcn_sinf(x[i]); c[i] = __builtin_amdgcn_cosf(x[i]); }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n);
    for (int i = 0; i < n; ++i) x[i] = (i < n / 2) ? (float)((double)i / (n / 2)) : (float)(-2.0 + 4.0 * (double)(i - n / 2) / (n / 2));
    float *dx, *ds, *dc;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dc, n * 4);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
    (void)hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es[2] = {0, 0}, ec[2] = {0, 0};
    for (int i = 0; i < n; ++i) {
        const double t = 6.283185307179586476925 * (double)x[i];
        const int h = i >= n / 2;
        es[h] = fmax(es[h], fabs(s[i] - sin(t))); ec[h] = fmax(ec[h], fabs(c[i] - cos(t)));
    }
    printf("max abs error  [0,1): sin %.3e cos %.3e   [-2,2]: sin %.3e cos %.3e\n", es[0], ec[0], es[1], ec[1]);
    return 0;
}
