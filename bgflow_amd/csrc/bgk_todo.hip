/* bgk_todo.hip -- entry points declared in include/bgflow_amd.h whose kernels are not written yet.
 * They fail loudly (BGK_EUNSUPPORTED + message); nothing falls back to a CPU path. */
#include "bgk_common.h"

extern "C" int bgk_rqs_backward(const float*, int64_t, const float*, int64_t, int32_t, const int32_t*, int64_t,
                                int32_t, int32_t, int32_t, double, double, double, double, double, double,
                                double, int32_t, const float*, int64_t, const float*, float*, int64_t, float*,
                                int64_t, void*) {
    bgk_set_error("bgk_rqs_backward: not implemented yet");
    return BGK_EUNSUPPORTED;
}

extern "C" int bgk_affine_backward(const float*, int64_t, const float*, int64_t, const float*, int64_t,
                                   const float*, int32_t, int32_t, int32_t, int64_t, int32_t, const float*,
                                   int64_t, const float*, float*, int64_t, float*, int64_t, float*, int64_t,
                                   float*, void*) {
    bgk_set_error("bgk_affine_backward: not implemented yet");
    return BGK_EUNSUPPORTED;
}

