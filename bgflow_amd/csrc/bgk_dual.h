/* bgk_dual.h -- forward-mode dual numbers (value + N directional derivatives) for the backward kernels of the coordinate
 * transforms (device only).  The reference differentiates its EXPLICIT Jacobian-determinant formulas with torch autograd
 * (crd_transform/ic.py:386-433, ic_helper.py:148-293, 480-680); evaluating the same formulas on Dual<N> reproduces that
 * derivative exactly -- eps clamps included (a clamped quantity has zero derivative, like torch.clamp) -- without a hand-derived
 * adjoint.  A VJP of a k-input function takes ceil(k / N) passes; the functions here have 9 or 12 inputs and ~100 flops. */
#ifndef BGK_DUAL_H
#define BGK_DUAL_H

#include <hip/hip_runtime.h>

template <int N>
struct Dual {
    float v;
    float d[N];
};

template <int N> __device__ __forceinline__ Dual<N> dconst(float c) { Dual<N> r; r.v = c; for (int i = 0; i < N; ++i) r.d[i] = 0.0f; return r; }
template <int N> __device__ __forceinline__ Dual<N> dseed(float c, int k) { Dual<N> r; r.v = c; for (int i = 0; i < N; ++i) r.d[i] = (i == k) ? 1.0f : 0.0f; return r; }   /* (selects: k may be a run-time value) */

template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(Dual<N> a, Dual<N> b) {
    Dual<N> r; const float inv = 1.0f / b.v; r.v = a.v * inv;
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, float b) { a.v += b; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator+(float b, Dual<N> a) { a.v += b; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a, float b) { a.v -= b; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator-(float b, Dual<N> a) { Dual<N> r; r.v = b - a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, float b) { a.v *= b; for (int i = 0; i < N; ++i) a.d[i] *= b; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator*(float b, Dual<N> a) { return a * b; }
template <int N> __device__ __forceinline__ Dual<N> operator/(Dual<N> a, float b) { return a * (1.0f / b); }
template <int N> __device__ __forceinline__ Dual<N> operator/(float b, Dual<N> a) { return dconst<N>(b) / a; }

template <int N> __device__ __forceinline__ Dual<N> dsqrt(Dual<N> a) {
    Dual<N> r; r.v = __builtin_sqrtf(a.v); const float s = 0.5f / r.v;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> dsin(Dual<N> a) { Dual<N> r; const float c = cosf(a.v); r.v = sinf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
template <int N> __device__ __forceinline__ Dual<N> dcos(Dual<N> a) { Dual<N> r; const float s = -sinf(a.v); r.v = cosf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> __device__ __forceinline__ Dual<N> dacos(Dual<N> a) {
    Dual<N> r; r.v = acosf(a.v); const float s = -1.0f / __builtin_sqrtf(1.0f - a.v * a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> datan2(Dual<N> y, Dual<N> x) {
    Dual<N> r; r.v = atan2f(y.v, x.v); const float q = 1.0f / (x.v * x.v + y.v * y.v);
    for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * q;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> dlog(Dual<N> a) { Dual<N> r; r.v = logf(a.v); const float s = 1.0f / a.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> __device__ __forceinline__ Dual<N> dabs(Dual<N> a) { return a.v < 0.0f ? -a : a; }
/* torch.clamp semantics: the derivative passes only where the value is not clamped */
template <int N> __device__ __forceinline__ Dual<N> dclamp_min(Dual<N> a, float lo) { return a.v < lo ? dconst<N>(lo) : a; }
template <int N> __device__ __forceinline__ Dual<N> dclamp(Dual<N> a, float lo, float hi) { return a.v < lo ? dconst<N>(lo) : (a.v > hi ? dconst<N>(hi) : a); }

template <int N> struct DV3 { Dual<N> x, y, z; };
template <int N> __device__ __forceinline__ DV3<N> dsub(DV3<N> a, DV3<N> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <int N> __device__ __forceinline__ Dual<N> ddot(DV3<N> a, DV3<N> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <int N> __device__ __forceinline__ DV3<N> dcross(DV3<N> a, DV3<N> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <int N> __device__ __forceinline__ Dual<N> dnorm(DV3<N> a) { return dsqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
template <int N> __device__ __forceinline__ DV3<N> ddivs(DV3<N> a, Dual<N> s) { return {a.x / s, a.y / s, a.z / s}; }
template <int N> __device__ __forceinline__ DV3<N> dscale(DV3<N> a, Dual<N> s) { return {a.x * s, a.y * s, a.z * s}; }
template <int N> __device__ __forceinline__ DV3<N> dadd(DV3<N> a, DV3<N> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

#endif /* BGK_DUAL_H */
