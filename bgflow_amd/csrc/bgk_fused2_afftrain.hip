/* bgk_fused2_afftrain.hip -- training forward of the one-launch AFFINE coupling layer (round 6): coupling_affine_dense_v2_kernel
 * (bgk_fused2.hip compiled with BGK_V2_AFFTRAIN = 1) -- same arithmetic and MFMA event threading as the inference kernel, and in
 * addition it writes what the hand-written backward reads: per conditioner network the scaled pre-activations z0, z1 [B, 64 | 128] of
 * its two hidden layers (for bgk_dense_backward_dx / bgk_mlp_weight_grad) and its output rows -- the shift values mu and the scale
 * values before tanh (for bgk_affine_backward).  Replaces, under autograd, CouplingFlow._forward (nn/flow/coupling.py:162-182) around
 * AffineTransformer (nn/flow/transformer/affine.py:35-70) with DenseNet conditioners (nn/dense.py:30-48): before round 6 every
 * affine coupling that needed a gradient ran its networks layer by layer and their backward on library GEMMs. */
#define BGK_V2_AFFTRAIN 1
#include "bgk_fused2.hip"

extern "C" int bgk_coupling_affine_dense_h2_train(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                                  const void* sA0, const void* sA1, const void* sA2, const float* s_cs, int32_t s_act,
                                                  const void* tA0, const void* tA1, const void* tA2, const float* t_cs, int32_t t_act,
                                                  const float* log_alpha, int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                                                  const float* y, int64_t ldy, int64_t B, int32_t d, float* out, int64_t ldo,
                                                  float* dlogp, int32_t accumulate,
                                                  float* s_z0, float* s_z1, float* t_z0, float* t_z1, int64_t ldz,
                                                  float* mu, float* s_raw, int64_t ldms, void* stream) {
    if (B == 0) return 0;
    const char* what = "bgk_coupling_affine_dense_h2_train";
    BGK_CHECK_ARG(cond && ldc && width && n_cond >= 1 && n_cond <= BGK_MAX_COND, "%s: 1 .. %d conditioning tensors", what, BGK_MAX_COND);
    BGK_CHECK_ARG(y && out && dlogp && B > 0 && d > 0 && d <= 96, "%s: bad sizes", what);
    BGK_CHECK_ARG((sA0 || tA0) && (!sA0 || (sA1 && sA2 && s_cs)) && (!tA0 || (tA1 && tA2 && t_cs && log_alpha)), "%s: null operand", what);
    BgkCondSegs segs;
    int d_c = 0;
    for (int i = 0; i < BGK_MAX_COND; ++i) { segs.ptr[i] = nullptr; segs.ld[i] = 0; segs.w[i] = 0; }
    for (int i = 0; i < n_cond; ++i) { segs.ptr[i] = cond[i]; segs.ld[i] = ldc[i]; segs.w[i] = width[i]; d_c += width[i]; }
    segs.n = n_cond;
    const int n_in = periodic ? 2 * d_c : d_c;
    if (n_in > 127) { bgk_set_error("%s: %d input features > 127", what, n_in); return BGK_EUNSUPPORTED; }
    const BgkAffTrainSave save{s_cs, s_z0, s_z1, t_cs, t_z0, t_z1, mu, s_raw, ldms, ldz};
    return bgk_launch_affine_dense_v2_train(&save, cond[0], ldc[0], d_c, periodic,
                                            sA0, sA1, nullptr, sA2, 1.0f, 1.0f, 1.0f, 1.0f, s_act,
                                            tA0, tA1, nullptr, tA2, 1.0f, 1.0f, 1.0f, 1.0f, t_act,
                                            log_alpha, preserve_volume, is_circular, inverse, y, ldy, B, d, out, ldo, dlogp, accumulate, stream, &segs);
}
