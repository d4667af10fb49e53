/* bgk_fused2.h -- internal: launcher of the second-generation split-f16 coupling kernel (bgk_fused2.hip) */
#ifndef BGK_FUSED2_H
#define BGK_FUSED2_H
#include <stdint.h>

#define BGK_MAX_COND 3
/* several conditioning tensors [B, w_i] standing for their concatenation (host-side table of device pointers; NULL / n <= 1: the
 * single (cond, ldc, d_c) tensor of the launcher's own arguments) */
struct BgkCondSegs { const float* ptr[BGK_MAX_COND]; int64_t ld[BGK_MAX_COND]; int32_t w[BGK_MAX_COND]; int32_t n; };

int bgk_launch_rqs_dense_h2v2(const char* what, const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                              const void* A0p, const void* A1p, const void* A2p, float c0, float c1, float c2, const float* cs_dev,
                              int32_t act, const float* y, int64_t ldy, int64_t B, int32_t d, uint64_t circ_mask, int32_t inverse,
                              double left, double right, double bottom, double top,
                              double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
                              float* out, int64_t ldo, float* dlogp, int32_t accumulate, int32_t* bin_idx, int32_t* oob_count,
                              void* stream, const BgkCondSegs* segs = nullptr);

/* the training forward (bgk_fused2_train.hip): same kernel + z0, z1 [B, 128] and params [B, P] written for the backward */
int bgk_launch_rqs_dense_h2v2_train(const char* what, float* z0, float* z1, float* params, int64_t ldp, const int32_t* src_col,
                                    const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                    const void* A0p, const void* A1p, const void* A2p, float c0, float c1, float c2, const float* cs_dev,
                                    int32_t act, const float* y, int64_t ldy, int64_t B, int32_t d, uint64_t circ_mask, int32_t inverse,
                                    double left, double right, double bottom, double top,
                                    double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
                                    float* out, int64_t ldo, float* dlogp, int32_t accumulate, int32_t* bin_idx, int32_t* oob_count,
                                    void* stream, const BgkCondSegs* segs = nullptr);

/* spline backward of a layer the training forward ran WITHOUT writing its parameters (params == NULL there): the output layer of
 * the conditioner redone from z1 on the matrix cores, bgk_rqs_vjp_element on every element (bgk_fused2_train.hip) */
int bgk_launch_rqs_bwd_recompute(const char* what, const float* z1, const void* A2p, float c2, const float* cs_dev, int32_t act,
                                 const float* y, int64_t ldy, int64_t B, int32_t d, uint64_t circ_mask, int32_t inverse,
                                 double left, double right, double bottom, double top,
                                 double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
                                 const float* g_out, int64_t ldgo, const float* g_dlogp, float* g_y, int64_t ldgy,
                                 float* g_params, int64_t ldgp, float* g_absmax, void* stream);

/* reduced-precision mode "bf16" (bgk_fused2_bf16.hip): same kernel, one bf16 MFMA per product */
int bgk_launch_rqs_dense_h2v2_bf16(const char* what, const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                   const void* A0p, const void* A1p, const void* A2p, float c0, float c1, float c2, const float* cs_dev,
                                   int32_t act, const float* y, int64_t ldy, int64_t B, int32_t d, uint64_t circ_mask, int32_t inverse,
                                   double left, double right, double bottom, double top,
                                   double min_bin_width, double min_bin_height, double min_derivative, int32_t identity_init,
                                   float* out, int64_t ldo, float* dlogp, int32_t accumulate, int32_t* bin_idx, int32_t* oob_count,
                                   void* stream, const BgkCondSegs* segs = nullptr);

/* affine coupling layer with conditioners of width 128 (two or three hidden layers) on the same event-threaded GEMM stream
 * (bgk_fused2.hip); BGK_EUNSUPPORTED for activation pairs it has no instance for */
int bgk_launch_affine_dense_v2(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                               const void* sA0, const void* sA1, const void* sA1b, const void* sA2, float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                               const void* tA0, const void* tA1, const void* tA1b, const void* tA2, float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                               const float* log_alpha, int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                               const float* y, int64_t ldy, int64_t B, int32_t d,
                               float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream, const BgkCondSegs* segs = nullptr);

/* the training forward of the affine layer (bgk_fused2_afftrain.hip): the same kernel (two hidden layers) + what the backward reads --
 * per network the scaled pre-activations z0, z1 [B, ldz] and its output rows (mu; the scale values before tanh) [B, ldms];
 * s_cs / t_cs: device scale tables of the packed operands (NULL: the c values of the call) */
struct BgkAffTrainSave { const float* s_cs; float* s_z0; float* s_z1; const float* t_cs; float* t_z0; float* t_z1; float* mu; float* s_raw; int64_t ldms;
                         int64_t ldz; };      /* row pitch of the z arrays: 128, or 64 when every hidden layer has <= 64 units */
int bgk_launch_affine_dense_v2_train(const BgkAffTrainSave* save, const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                               const void* sA0, const void* sA1, const void* sA1b, const void* sA2, float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                               const void* tA0, const void* tA1, const void* tA1b, const void* tA2, float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                               const float* log_alpha, int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                               const float* y, int64_t ldy, int64_t B, int32_t d,
                               float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream, const BgkCondSegs* segs = nullptr);

/* 2 (default): coupling_rqs_dense_h2v2_kernel for the split-f16 path (inference and training forward); 1: the first-generation kernel */
extern int bgk_h2_variant;
/* 2 (default): bgk_coupling_rqs_dense_h2_backward evaluates the element VJP's softmax / knots on the hardware exp2 / rcp forms (like the
 * fused forward it belongs to); 1: on the deterministic forms of bgk_rqs_backward (bit-identical gradients with the saved-parameter path) */
extern int bgk_rc_vjp_variant;

#endif
