/* bgk_capi.hip -- library-level entry points (version, error string, deterministic-math probe). */
#include <stdarg.h>

#include "bgk_common.h"
#include "bgk_fused2.h"

extern int bgk_affine_variant;      /* bgk_fused_affine.hip */

static thread_local char g_err[512] = "";

void bgk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int bgk_abi_version(void) { return 1; }
extern "C" const char* bgk_last_error(void) { return g_err; }

extern "C" int bgk_set_option(int32_t option, int32_t value) {
    if (option == 1 && (value == 1 || value == 2)) {
        const int prev = bgk_h2_variant;
        bgk_h2_variant = value;
        return prev;
    }
    if (option == 2 && (value == 1 || value == 2)) {
        const int prev = bgk_affine_variant;
        bgk_affine_variant = value;
        return prev;
    }
    if (option == 3 && (value == 1 || value == 2)) {
        const int prev = bgk_rc_vjp_variant;
        bgk_rc_vjp_variant = value;
        return prev;
    }
    bgk_set_error("bgk_set_option: unknown option %d / value %d", option, value);
    return BGK_EINVAL;
}

namespace {
__global__ void detmath_probe_kernel(const float* x, int64_t n, int which, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i], r;
    switch (which) {
        case 0: r = bgk_expf(v); break;
        case 1: r = bgk_logf(v); break;
        case 2: r = bgk_softplusf(v, 0.69384102162f); break;
        case 3: r = bgk_siluf(v); break;
        case 4: r = bgk_tanhf(v); break;
        default: r = v;
    }
    out[i] = r;
}
}  // namespace

extern "C" int bgk_detmath_probe(const float* x, int64_t n, int32_t which, float* out, void* stream) {
    if (n <= 0) return 0;
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(detmath_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, which, out);
    return bgk_launch_status("bgk_detmath_probe");
}
