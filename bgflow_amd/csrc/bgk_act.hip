/* bgk_act.hip -- the hidden activations of a DenseNet as stand-alone elementwise kernels and their VJP (round 6): what the
 * layer-by-layer TRAINING path of a conditioner outside the one-launch envelopes runs between its bgk_dense_layer calls (nn/dense.py:30-48:
 * `activation` after every hidden Linear; autograd of it in loss.backward(), nn/training/trainers.py:156-163).  Before round 6 these
 * were torch's aten kernels (silu / tanh / threshold and their backward forms).  Same reproducible SiLU / Tanh forms as the epilogue of
 * bgk_dense_layer (bgk_detmath_pk.h), so a forward that fuses the activation into the layer kernel and one that runs it here agree
 * bit for bit.  Roofline: HBM, 8 B (forward) / 12 B (backward) per element.
 */
#include "bgk_common.h"
#include "bgk_detmath_pk.h"

namespace {

struct ActArgs { const float* z; int64_t ldz; const float* g; int64_t ldg; float* out; int64_t ldo; int64_t B; int n; int act; };

/* d act(z) / dz: SiLU s (1 + z (1 - s)) with s = 1 / (1 + exp(-z)); ReLU [z > 0]; Tanh 1 - tanh(z)^2 */
__device__ __forceinline__ float act_deriv(float z, int act) {
    if (act == 1) {
        const float s = bgk_div_safe(1.0f, 1.0f + bgk_expf(-z));
        return s * (1.0f + z * (1.0f - s));
    }
    if (act == 2) return z > 0.0f ? 1.0f : 0.0f;
    const float t = bgk_tanhf(z);
    return 1.0f - t * t;
}

template <bool BWD>
__global__ __launch_bounds__(256) void act_kernel(ActArgs a) {
    /* pairs of a row (rows of even length at 8-byte aligned addresses: the launcher checks), walked with (row, column) counters */
    const int hn = a.n >> 1;
    const int64_t stride = (int64_t)gridDim.x * 256, i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = i0 / hn;
    int c = (int)(i0 - r * hn);
    const int64_t dr = stride / hn;
    const int dc = (int)(stride - dr * hn);
    for (; r < a.B;) {
        const float2 z = *reinterpret_cast<const float2*>(a.z + r * a.ldz + 2 * c);
        float2 o;
        if (BWD) {
            const float2 g = *reinterpret_cast<const float2*>(a.g + r * a.ldg + 2 * c);
            o = make_float2(g.x * act_deriv(z.x, a.act), g.y * act_deriv(z.y, a.act));
        } else if (a.act == 1) {
            const bgk_f2 u = bgk_siluf2((bgk_f2){z.x, z.y});
            o = make_float2(u.x, u.y);
        } else if (a.act == 3) {
            const bgk_f2 u = bgk_tanhf2((bgk_f2){z.x, z.y});
            o = make_float2(u.x, u.y);
        } else {
            o = make_float2(z.x > 0.0f ? z.x : 0.0f, z.y > 0.0f ? z.y : 0.0f);
        }
        *reinterpret_cast<float2*>(a.out + r * a.ldo + 2 * c) = o;
        c += dc; r += dr;
        if (c >= hn) { c -= hn; ++r; }
    }
}

/* any row length / alignment: one element per step */
template <bool BWD>
__global__ __launch_bounds__(256) void act_scalar_kernel(ActArgs a) {
    const int64_t n = a.B * (int64_t)a.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.n;
        const int c = (int)(i - r * a.n);
        const float z = a.z[r * a.ldz + c];
        float o;
        if (BWD) o = a.g[r * a.ldg + c] * act_deriv(z, a.act);
        else if (a.act == 1) o = bgk_siluf2((bgk_f2){z, z}).x;
        else if (a.act == 3) o = bgk_tanhf2((bgk_f2){z, z}).x;
        else o = z > 0.0f ? z : 0.0f;
        a.out[r * a.ldo + c] = o;
    }
}

int act_launch(const ActArgs& a, bool bwd, hipStream_t st) {
    const auto even = [](const void* p, int64_t ld) { return p == nullptr || (ld % 2 == 0 && ((uintptr_t)p & 7) == 0); };
    const bool pairs = a.n % 2 == 0 && even(a.z, a.ldz) && even(a.g, a.ldg) && even(a.out, a.ldo);
    const int64_t items = pairs ? a.B * (a.n / 2) : a.B * (int64_t)a.n;
    const int64_t want = (items + 255) / 256;
    const int grid = (int)(want < 256 * 16 ? want : 256 * 16);
    if (pairs) {
        if (bwd) hipLaunchKernelGGL(act_kernel<true>, dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(act_kernel<false>, dim3(grid), dim3(256), 0, st, a);
    } else {
        if (bwd) hipLaunchKernelGGL(act_scalar_kernel<true>, dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(act_scalar_kernel<false>, dim3(grid), dim3(256), 0, st, a);
    }
    return 0;
}

}  // namespace

extern "C" int bgk_activation(const float* z, int64_t ldz, int64_t B, int32_t n, int32_t act, float* out, int64_t ldo, void* stream) {
    if (B == 0) return 0;
    BGK_CHECK_ARG(z && out && B > 0 && n > 0 && ldz >= n && ldo >= n && act >= 1 && act <= 3, "bgk_activation: bad arguments (act: 1 SiLU, 2 ReLU, 3 Tanh)");
    act_launch(ActArgs{z, ldz, nullptr, 0, out, ldo, B, n, act}, false, (hipStream_t)stream);
    return bgk_launch_status("bgk_activation");
}

extern "C" int bgk_activation_backward(const float* z, int64_t ldz, const float* g, int64_t ldg, int64_t B, int32_t n, int32_t act,
                                       float* g_z, int64_t ldgz, void* stream) {
    if (B == 0) return 0;
    BGK_CHECK_ARG(z && g && g_z && B > 0 && n > 0 && ldz >= n && ldg >= n && ldgz >= n && act >= 1 && act <= 3,
                  "bgk_activation_backward: bad arguments (act: 1 SiLU, 2 ReLU, 3 Tanh)");
    act_launch(ActArgs{z, ldz, g, ldg, g_z, ldgz, B, n, act}, true, (hipStream_t)stream);
    return bgk_launch_status("bgk_activation_backward");
}
