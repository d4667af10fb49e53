/* bgflow_amd.h -- C ABI of libbgflow_amd.so: MI355X (gfx950) kernels for the bgflow coupling-flow
 * hot path.  Plain pointers and sizes only (no torch types); every pointer is a DEVICE pointer
 * unless stated otherwise; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * No entry point allocates, synchronises or mutates its inputs.  Return value: 0 on success,
 * otherwise a hipError_t code (launch failure) or a negative BGK_E* code (bad arguments); the
 * message is available from bgk_last_error().
 *
 * bgflow itself has no FFI: its "operator interface" for this path is the Python Flow /
 * Transformer protocol.  Each entry point below replaces the aten-op chain of ONE reference
 * method (file:line relative to /root/reference/bgflow/); bgflow_amd/ (python) binds them with ctypes
 * behind classes that keep the reference's names and signatures.
 *
 * Layout conventions: 2-d operands are row-major with an explicit row stride in ELEMENTS
 * (ld*), unit column stride; `dlogp` vectors are [B] contiguous (the [B,1] tensors of bgflow).
 * `accumulate` != 0 adds the layer's log|det J| to dlogp (SequentialFlow's `dlogp += ddlogp`,
 * nn/flow/sequential.py:58) instead of overwriting it.
 *
 * Empty batches: with B == 0 every entry point that takes a batch size returns 0 before it looks at a pointer (the tensors of an
 * empty batch have no storage); reductions over the batch write their neutral element (bgk_column_sum: zeros; the loss sums of
 * bgk_energy_fields: {0, 0}).  BGK_EUNSUPPORTED means "outside this kernel's envelope" (a width, bin count or molecule size it has no
 * instance for): the caller is expected to take its next path down, not to report an error.
 */
#ifndef BGFLOW_AMD_H
#define BGFLOW_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden and an export list generated from this header (bgflow_amd/build.py): the prototypes
 * below are its complete dynamic symbol table (tests/test_host_logic.py::test_exported_symbols_are_the_headers_prototypes). */
#pragma GCC visibility push(default)

#define BGK_EINVAL (-1)      /* bad argument (shape / unsupported size)          */
#define BGK_EUNSUPPORTED (-2) /* valid request the kernels do not cover (yet)    */

/* library identification / diagnostics */
int bgk_abi_version(void);
const char* bgk_last_error(void);            /* thread-local, host string */

/* Library-wide switches (diagnostics / A-B measurements; defaults are the shipped configuration).
 *   option 1: generation of the split-f16 inference coupling kernel behind bgk_coupling_rqs_dense_h2:
 *             2 (default) = coupling_rqs_dense_h2v2_kernel (MFMA stream threaded through the VALU work, bgk_fused2.hip),
 *             1 = the first-generation kernel (bgk_fused.hip).
 *   option 2: kernel behind bgk_coupling_affine_dense_h2: 2 (default) = for hidden (64, 64) the weight-resident kernel (both
 *             conditioners' packed operands staged once per workgroup in LDS), for hidden (128, 128) the event-threaded kernel
 *             of bgk_fused2.hip; 1 = the streaming kernel (operands from L2, GEMM and activation phases alternate).
 *   option 3: element VJP inside bgk_coupling_rqs_dense_h2_backward: 2 (default) = softmax / knots on the hardware exp2 / rcp forms
 *             (what the fused forward it belongs to evaluates the spline with), 1 = the deterministic forms of bgk_rqs_backward
 *             (gradients bit-identical to the saved-parameter path; ~40 % more VALU instructions per element).
 * Returns the previous value, BGK_EINVAL for an unknown option / value. */
int bgk_set_option(int32_t option, int32_t value);

/* Deterministic-math probe: out[i] = f(x[i]) on the device with the same primitives the spline
 * kernels use (which: 0 exp, 1 log, 2 softplus(beta=ln2/(1-1e-3)), 3 silu, 4 tanh).  Test hook. */
int bgk_detmath_probe(const float* x, int64_t n, int32_t which, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Rational-quadratic spline transformer, conditioner output given.
 * Replaces ConditionalSplineTransformer._compute_params (reshape/cat/index_put,
 * nn/flow/transformer/spline.py:109-126) + nflows rational_quadratic_spline (spline.py:133-144,
 * 164-175) + the clamp-and-retry on InputOutsideDomain (spline.py:145-155) + sum(-1)
 * (spline.py:157,188).
 *   y       [B, d]   inputs in [left, right]           (ldy)
 *   params  [B, P]   P = 3*K*d + n_nc, packed [w: d*K | h: d*K | s: d*K | s_nc: n_nc]   (ldp >= P)
 *   nc_slot [d]      int32: index of dim j's extra slope inside the s_nc block, -1 if circular
 *   inverse          0 = bgflow _forward (nflows inverse=True, root solve), 1 = bgflow _inverse
 *   out     [B, d]   (ldo);  dlogp [B];  bin_idx [B, d] int32 or NULL (parity output)
 *   oob_count        int32[1] or NULL: += number of inputs that were outside [left,right]
 *                    (they are clamped exactly as the reference does; the caller raises the
 *                    UserWarning lazily)
 * --------------------------------------------------------------------------------------------- */
int bgk_rqs_transform(const float* y, int64_t ldy, const float* params, int64_t ldp, int32_t P,
                      const int32_t* nc_slot, int64_t B, int32_t d, int32_t K, int32_t inverse,
                      double left, double right, double bottom, double top,
                      double min_bin_width, double min_bin_height, double min_derivative,
                      int32_t identity_init,
                      float* out, int64_t ldo, float* dlogp, int32_t accumulate,
                      int32_t* bin_idx, int32_t* oob_count, void* stream);

/* Backward of bgk_rqs_transform for first-order losses (KL / NLL):
 *   given g_out [B,d] (ldgo) and g_dlogp [B], produces g_y [B,d] (ldgy) and g_params [B,P] (ldgp).
 * Replaces torch autograd through the op chain above.  Any bin count: 4 / 8 / 12 / 16 / 32 bins on a register-resident streaming
 * kernel, any other on a variant that walks the element's parameters in memory (round 5; like the forward, more than 64 bins use
 * compensated knot sums).  g_absmax (device, may be NULL): g_absmax[0] is raised to the largest |g_params| written -- zero it first;
 * bgk_dense_backward_dx / bgk_dense_weight_grad derive the power-of-two scale of their f16 operand split from it. */
int bgk_rqs_backward(const float* y, int64_t ldy, const float* params, int64_t ldp, int32_t P,
                     const int32_t* nc_slot, int64_t B, int32_t d, int32_t K, int32_t inverse,
                     double left, double right, double bottom, double top,
                     double min_bin_width, double min_bin_height, double min_derivative,
                     int32_t identity_init,
                     const float* g_out, int64_t ldgo, const float* g_dlogp,
                     float* g_y, int64_t ldgy, float* g_params, int64_t ldgp, float* g_absmax, int32_t params_layout, void* stream);
/* params_layout: 0 = `params` in the reference's column order [w | h | s | s_nc] (ldp >= P); 1 = element-major [B][d][3 K + 1]
 * (an element's widths | heights | slopes | slot in one run; ldp >= (3 K + 1) d; 8 bins only) -- what
 * bgk_coupling_rqs_dense_h2_train writes with its params_layout = 1.  g_params is in the reference's order either way. */

/* ---------------------------------------------------------------------------------------------
 * Affine (RealNVP / NICE) transformer, conditioner outputs given.
 * Replaces AffineTransformer._get_mu_and_log_sigma's elementwise tail + _forward/_inverse
 * (nn/flow/transformer/affine.py:41-45, 50-70): log_sigma = tanh(s_raw)*exp(log_alpha)
 * [- mean], y' = exp(+-log_sigma)*(...), dlogp = +-sum log_sigma, optional `% 1.0`.
 *   mu / s_raw may be NULL (no shift / no scale network); log_alpha is a DEVICE scalar
 *   (the learnable parameter, affine.py:31).
 * --------------------------------------------------------------------------------------------- */
int bgk_affine_transform(const float* y, int64_t ldy, const float* mu, int64_t ldmu,
                         const float* s_raw, int64_t lds, const float* log_alpha,
                         int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                         int64_t B, int32_t d, float* out, int64_t ldo,
                         float* dlogp, int32_t accumulate, void* stream);

/* backward: g_y, g_mu, g_s_raw [B,d] (any may be NULL), g_log_alpha float[1] (+=, may be NULL).
 * g_mu_absmax / g_s_absmax (round 6; NULL or one device float each, zeroed by the caller): raised to max |g_mu| / max |g_s_raw| -- the
 * power-of-two scale source of the backward GEMMs that consume them (bgk_dense_backward_dx / bgk_mlp_weight_grad of the shift and
 * the scale network) */
int bgk_affine_backward(const float* y, int64_t ldy, const float* mu, int64_t ldmu,
                        const float* s_raw, int64_t lds, const float* log_alpha,
                        int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                        int64_t B, int32_t d,
                        const float* g_out, int64_t ldgo, const float* g_dlogp,
                        float* g_y, int64_t ldgy, float* g_mu, int64_t ldgmu,
                        float* g_s, int64_t ldgs, float* g_log_alpha, float* g_mu_absmax, float* g_s_absmax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Internal coordinates (Z-matrix <-> Cartesian), relative to fixed atoms, optional PCA
 * whitening of the fixed block.
 * bgk_ic_xyz2ic replaces RelativeInternalCoordinateTransformation._forward
 *   (nn/flow/crd_transform/ic.py:386-433; dist/angle/torsion_deriv ic_helper.py:148-293) and, with
 *   Twhiten != NULL, MixedCoordinateTransformation._forward (ic.py:838-860, pca.py:74-83).
 * bgk_ic_ic2xyz replaces RelativeInternalCoordinateTransformation._inverse
 *   (ic.py:435-513; ic2xyz_deriv ic_helper.py:372-452) and, with Tblacken != NULL,
 *   MixedCoordinateTransformation._inverse (ic.py:862-884, pca.py:85-93).
 *   x        [B, 3*n_atoms] (ldx)
 *   zmat     [n, 4] int32 rows (a, b, c, d)                        (xyz2ic)
 *   place    [n, 5] int32 rows (atom, p1, p2, p3, zrow) in placement order (ic2xyz)
 *   fixed    [n_fixed] int32 atom ids
 *   bonds/angles/torsions [B, n] (ld 'ldic'), xfix [B, keep] (ldf) with keep = 3*n_fixed when
 *   no whitening;  wh_mean [3*n_fixed], Twhiten [3*n_fixed, keep], Tblacken [keep, 3*n_fixed]
 *   warn_count int32[1] or NULL: += number of eps-clamps that fired (reference: warnings.warn)
 * --------------------------------------------------------------------------------------------- */
int bgk_ic_xyz2ic(const float* x, int64_t ldx, const int32_t* zmat, int32_t n,
                  const int32_t* fixed, int32_t n_fixed, int32_t normalize_angles, float eps,
                  int32_t enforce_boundaries,
                  const float* wh_mean, const float* Twhiten, int32_t keep, float jac_xz,
                  int64_t B, float* bonds, float* angles, float* torsions, int64_t ldic,
                  float* xfix, int64_t ldf, float* dlogp, int32_t accumulate,
                  int32_t* warn_count, void* stream);

int bgk_ic_ic2xyz(const float* bonds, const float* angles, const float* torsions, int64_t ldic,
                  const float* xfix, int64_t ldf, const int32_t* place, int32_t n,
                  const int32_t* fixed, int32_t n_fixed, int32_t normalize_angles, float eps,
                  int32_t enforce_boundaries,
                  const float* wh_mean, const float* Tblacken, int32_t keep, float jac_xz,
                  int64_t B, float* x, int64_t ldx, float* dlogp, int32_t accumulate,
                  int32_t* warn_count, void* stream);

/* Global reference frame of the first three atoms: replaces ReferenceSystemTransformation._forward /
 * _inverse (crd_transform/ic.py:162-265; init_xyz2ics / init_ics2xyz with their batched autograd
 * 9x9 Jacobians, ic_helper.py:480-680).  Packed 9-vectors per sample:
 *   inverse = 0:  (x0, x1, x2) -> (x0[3], d01, d12, a012, alpha, beta, gamma);  inverse = 1: the reverse.
 * log|det J| in closed form: -/+ (2 ln d01 + 2 ln d12 + ln sin a012) (+ angle-normalisation constants). */
int bgk_ic_refsys(const float* in, int64_t B, int32_t inverse, int32_t normalize_angles, float eps,
                  int32_t enforce_boundaries, float* out, float* dlogp, int32_t accumulate, void* stream);

/* Generation path: bgk_ic_ic2xyz with the icdf domain maps of the builder fused into its prologue (replaces the chain
 * CDFTransform._inverse x 4 -> RelativeInternalCoordinateTransformation._inverse: nn/flow/cdf.py:36-45,
 * generator_builder.py:443-459, crd_transform/ic.py:435-513): bonds / angles / torsions / xfix arrive as values in [0, 1],
 * desc_* = per-channel descriptors like bgk_cdf_transform's ([n, 6] x 3, [keep, 6]; NULL = that field is used as is).
 * dlogp receives the log-dets of the maps AND of the coordinate transform.  Placement log-det in closed form
 * (2 ln d + ln|sin a|), normalisations on reciprocal square roots. */
int bgk_icdf_ic2xyz(const float* bonds, const float* angles, const float* torsions, int64_t ldic,
                    const float* xfix, int64_t ldf,
                    const float* desc_bonds, const float* desc_angles, const float* desc_torsions, const float* desc_fixed,
                    int32_t use_eps, float cdf_eps,
                    const int32_t* place, int32_t n, const int32_t* fixed, int32_t n_fixed,
                    int32_t normalize_angles, float eps, int32_t enforce_boundaries,
                    const float* wh_mean, const float* Tblacken, int32_t keep, float jac_xz, int64_t B,
                    float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream);

/* The same tail on the register-resident kernel (csrc/bgk_tail.hip; the shipped path of builder flows up to 32 atoms): the growing
 * position table lives in registers (wave-uniform atom indices -> uniform register indexing), all uniform tables come through scalar
 * loads, field tiles are contiguous [B, n] / [B, keep] blocks (row stride = width), angles are normalised.
 *   place8 [n, 8] int32: one record per placement (atom, p1, p2, p3, zrow, 0, 0, 0)
 *   desc20 [3 n + keep, 20]: per-channel descriptors, the bonds | angles | torsions rows in PLACEMENT order (row f n + i = the channel
 *          of field f that placement i consumes), then the keep fixed rows; built on the host in f64
 *          (bgflow_amd/cdf.py::_tail_descriptor; layout documented in bgk_tail.hip): kind as int32 bits, the affine maps around erfinv
 *          folded into one fma each, and for truncated-normal marginals the reverted cdf series around each finite bound, with which
 *          the kernel evaluates the distance to the bound directly (no cancellation mu + sigma z for near-degenerate samples)
 *   const_ld: n (ln pi + ln 2 pi) - jac_xz (the constant part of the log-det)
 * Returns BGK_EUNSUPPORTED for n + n_fixed > 32 or keep > 16: the caller uses bgk_icdf_ic2xyz. */
int bgk_icdf_ic2xyz_reg(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                        const float* desc20, int32_t use_eps, float cdf_eps,
                        const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                        float eps, int32_t enforce_boundaries,
                        const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                        float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream);

/* ... and for FIELD-UNIFORM marginals (every channel of a field shares one descriptor: what the builder installs) the variant whose
 * icdf maps run elementwise over the field tiles in their memory layout: tiles reach LDS by DMA (global_load_lds), descriptors stay in
 * SGPRs, finished rows leave through LDS as 16-byte coalesced stores.  desc4 [4, 20] = the bonds / angles / torsions / fixed descriptor;
 * x contiguous (ldx = 3 (n + n_fixed)), all tensors 16-byte aligned; BGK_EUNSUPPORTED otherwise (use bgk_icdf_ic2xyz_reg). */
int bgk_icdf_ic2xyz_uni(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                        const float* desc4, int32_t use_eps, float cdf_eps,
                        const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                        float eps, int32_t enforce_boundaries,
                        const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                        float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream);
/* ... for a TRAINING forward (KLTrainer, trainers.py:158-170 through generator_builder.py:443-459's tail): the same launch also writes
 * the four mapped fields y = icdf(z) (contiguous [B, n] x 3, [B, keep]) -- the inputs bgk_ic_ic2xyz_backward and bgk_cdf_backward
 * need -- so the backward pass runs on those kernels while the forward is one launch instead of 4 x bgk_cdf_transform + bgk_ic_ic2xyz */
int bgk_icdf_ic2xyz_uni_train(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                              const float* desc4, int32_t use_eps, float cdf_eps,
                              const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                              float eps, int32_t enforce_boundaries,
                              const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                              float* x, int64_t ldx, float* dlogp, int32_t accumulate, int32_t* warn_count,
                              float* y_bonds, float* y_angles, float* y_torsions, float* y_fixed, void* stream);
/* ... and with the KL integrand of BoltzmannGenerator.kldiv (bg.py:140-147) formed in the same launch, for a target that is a normal
 * distribution about t_mean [3 (n + n_fixed)] (NULL: 0) -- distribution/normal.py:61-72: u = (|x - t_mean|^2 / 2 + c_in) / temperature
 * + c_out; every lane holds its sample's coordinates in registers when the placements are done, so the target energy costs no second
 * pass over x and the loss no per-sample tensor.  dlogp_in [B] (NULL: 0): log-det of the flow in front of the tail, read only.
 * Written besides x and the mapped fields: u [B], dlogp_total [B], partial [ceil(B / 64)][2] (per 64-sample tile: the sum of
 * u - dlogp_total over the samples kept and their number; drop_nonfinite: a sample whose integrand is not finite is not kept) and
 * loss_sums [2] (f64: the partials added in a fixed order by a second small launch) -- what bgk_energy_fields' loss form delivers. */
int bgk_icdf_ic2xyz_uni_train_kl(const float* bonds, const float* angles, const float* torsions, const float* xfix,
                                 const float* desc4, int32_t use_eps, float cdf_eps,
                                 const int32_t* place8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                                 float eps, int32_t enforce_boundaries,
                                 const float* wh_mean, const float* Tblacken, int32_t keep, double const_ld, int64_t B,
                                 float* x, int64_t ldx, const float* dlogp_in, int32_t* warn_count,
                                 float* y_bonds, float* y_angles, float* y_torsions, float* y_fixed,
                                 const float* t_mean, double temperature, double c_in, double c_out, int32_t drop_nonfinite,
                                 float* u, float* dlogp_total, float* partial, double* loss_sums, void* stream);

/* The INVERSE (NLL) direction of the builder tail in one launch: x [B, 3 (n + n_fixed)] -> cdf-mapped bonds / angles / torsions [B, n]
 * and (whitened) fixed coordinates [B, keep], all contiguous and 16-byte aligned, + log|det J|.  Replaces bgk_ic_xyz2ic + 4 x
 * bgk_cdf_transform, i.e. RelativeInternalCoordinateTransformation._forward / MixedCoordinateTransformation._forward
 * (crd_transform/ic.py:386-433, 838-860; ic_helper.py:148-293; pca.py:74-83) followed by CDFTransform._forward x 4 (nn/flow/cdf.py:28-35).
 * zmat8 [n, 8] int32 rows (a, b, c, d, 0, 0, 0, 0); desc4 [4, 20] as in bgk_icdf_ic2xyz_uni (field-uniform marginals); Twhiten
 * [3 n_fixed, keep]; const_ld = -n (ln pi + ln 2 pi) + jac_xz.  BGK_EUNSUPPORTED outside the envelope (n + n_fixed <= 32, keep <= 16,
 * 3 n + keep <= 3 (n + n_fixed)): the caller runs the separate kernels. */
int bgk_xyz2ic_cdf_uni(const float* x, const float* desc4, int32_t use_eps, float cdf_eps,
                       const int32_t* zmat8, int32_t n, const int32_t* fixed, int32_t n_fixed,
                       float eps, int32_t enforce_boundaries,
                       const float* wh_mean, const float* Twhiten, int32_t keep, double const_ld, int64_t B,
                       float* bonds, float* angles, float* torsions, float* xfix,
                       float* dlogp, int32_t accumulate, int32_t* warn_count, void* stream);

/* Backward (VJP) of bgk_ic_ic2xyz for first-order losses (replaces torch autograd through
 * ic2xyz_deriv / det3x3, ic.py:435-513): x is the forward OUTPUT; g_x [B, 3*n_atoms], g_dlogp [B]
 * -> g_bonds / g_angles / g_torsions [B, n] (ldgic), g_xfix [B, keep] (ldgf).  The log-det term uses
 * log|det J| = 2 ln d + ln|sin a|, exact away from the eps clamps; a placement whose norms the forward clamped (eps,
 * enforce_boundaries: the forward's, ic_helper.py:372-452) is differentiated the way the reference's autograd does it -- the
 * explicit Jacobian determinant with torch.clamp's derivative -- on forward-mode dual numbers.
 * fix_ws: device workspace of 1 + B int32 (contents irrelevant on entry), or NULL.  With it the sweep over all samples evaluates the
 * closed form only and lists the samples with a clamped norm, a second small launch redoes the listed samples with the dual numbers
 * (a few hundred of 2^18 at cfg 3's uniform prior); without it the dual numbers are evaluated inside a slower generic sweep.
 * Contiguous 16-byte aligned tensors (ldx = ldgx = 3 n_atoms, ldic = ldgic = n, ldgf = keep) are staged by DMA. */
int bgk_ic_ic2xyz_backward(const float* bonds, const float* angles, const float* torsions, int64_t ldic,
                           const float* x, int64_t ldx, const int32_t* place, int32_t n,
                           const int32_t* fixed, int32_t n_fixed, int32_t normalize_angles,
                           float eps, int32_t enforce_boundaries,
                           const float* Tblacken, int32_t keep, int64_t B,
                           const float* g_x, int64_t ldgx, const float* g_dlogp,
                           float* g_bonds, float* g_angles, float* g_torsions, int64_t ldgic,
                           float* g_xfix, int64_t ldgf, int32_t* fix_ws, void* stream);

/* Backward (VJP) of bgk_ic_xyz2ic (replaces torch autograd through RelativeInternalCoordinateTransformation._forward,
 * crd_transform/ic.py:386-433, the row Jacobians dist_deriv / angle_deriv / torsion_deriv of ic_helper.py:148-293 and the
 * whitening pca.py:74-82): upstream g_bonds / g_angles / g_torsions [B, n] (ldgic), g_xfix [B, keep] (ldgf), g_dlogp [B]
 * -> g_x [B, 3 n_atoms] (ldgx).  The same explicit-Jacobian formulas as the forward kernel, evaluated on dual numbers. */
int bgk_ic_xyz2ic_backward(const float* x, int64_t ldx, const int32_t* zmat, int32_t n,
                           const int32_t* fixed, int32_t n_fixed, int32_t normalize_angles, float eps,
                           int32_t enforce_boundaries, const float* Twhiten, int32_t keep, int64_t B,
                           const float* g_bonds, const float* g_angles, const float* g_torsions, int64_t ldgic,
                           const float* g_xfix, int64_t ldgf, const float* g_dlogp,
                           float* g_x, int64_t ldgx, void* stream);

/* Backward (VJP) of bgk_ic_refsys, both directions (replaces torch autograd through ReferenceSystemTransformation,
 * crd_transform/ic.py:162-265, ic_helper.py:480-680 incl. its autograd-Jacobian log-det): in [B, 9] = the forward INPUT,
 * g_out [B, 9], g_dlogp [B] -> g_in [B, 9]. */
int bgk_ic_refsys_backward(const float* in, const float* g_out, const float* g_dlogp, int64_t B, int32_t inverse,
                           int32_t normalize_angles, float eps, int32_t enforce_boundaries, float* g_in, void* stream);

/* Domain-mapping layer: replaces CDFTransform._forward/_inverse (nn/flow/cdf.py:28-46) for the
 * marginals of factory/icmarginals.py:41-77.  desc [d,6] floats per column: (kind, p0..p4) with
 * kind 0 uniform (low, high, tol), 1 normal (loc, scale), 2 truncated normal (mu, sigma, cdf_lower, Z).
 * inverse = 1: x in [0,1] -> icdf(x), logdet = -log_prob; inverse = 0: cdf, logdet = log_prob.
 * use_eps: clamp cdf values to [eps, 1-eps] and log-dets to >= -1/eps like the reference. */
int bgk_cdf_transform(const float* x, int64_t ldx, const float* desc, int64_t B, int32_t d,
                      int32_t inverse, int32_t use_eps, float eps, float* out, int64_t ldo,
                      float* dlogp, int32_t accumulate, void* stream);

/* Backward (VJP) of bgk_cdf_transform (replaces torch autograd through CDFTransform, nn/flow/cdf.py:28-46, and the
 * marginals' cdf / icdf / log_prob, distribution/normal.py:215-227): x = forward input, y = forward output (saved),
 * g_y [B, d], g_dlogp [B] -> g_x [B, d].  Values the forward pass clamped at eps receive no gradient (torch.clamp). */
int bgk_cdf_backward(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* desc, int64_t B, int32_t d,
                     int32_t inverse, int32_t use_eps, float eps, const float* g_y, int64_t ldgy,
                     const float* g_dlogp, float* g_x, int64_t ldgx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused spline coupling layer with a DenseNet conditioner (fast path of
 * CouplingFlow(ConditionalSplineTransformer(DenseNet | WrapPeriodic(DenseNet)))).
 * One launch replaces CouplingFlow._forward/_inverse (nn/flow/coupling.py:162-182) including the
 * conditioner MLP (nn/dense.py:47-48: Linear+act, Linear+act, Linear), the optional cos/sin
 * featuriser (nn/periodic.py:30-37) and everything bgk_rqs_transform does.  The three GEMMs run
 * on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ascending fma chain, bias added
 * last) and the spline parameters never leave the CU.
 *   cond  [B, d_c] (ldc): conditioner input; if `periodic` != 0 it is expanded to
 *         [cos(2 pi c), sin(2 pi c)] (all conditioner inputs circular, [0,1]) before layer 0
 *   W0p, W1p, W2p: the three Linear layers in the kernel's MFMA operand packing (built once per
 *         weight update by bgflow_amd/dense.py::pack_dense_for_fused; layout documented there and in
 *         DESIGN.md): per k-step one [64 lanes][4 tiles] float4 block, bias as the final k-step;
 *         W2p column-packed per transformed dim in the order given by bgk_pack_rqs_columns.
 *   circ_mask: bit j set = transformed dim j is circular (its slope at knot K is the slope at knot 0)
 *   act: 1 SiLU (builder default), 2 ReLU, 3 Tanh
 * Returns BGK_EUNSUPPORTED for shapes outside the fused envelope (hidden != (128,128), n_bins != 8,
 * d > 64): the caller then runs conditioner + bgk_rqs_transform.
 * --------------------------------------------------------------------------------------------- */
int bgk_coupling_rqs_dense(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                           const float* W0p, const float* W1p, const float* W2p,
                           int32_t H0, int32_t H1, int32_t act,
                           const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K,
                           uint64_t circ_mask, int32_t inverse,
                           double left, double right, double bottom, double top,
                           double min_bin_width, double min_bin_height, double min_derivative,
                           int32_t identity_init,
                           float* out, int64_t ldo, float* dlogp, int32_t accumulate,
                           int32_t* bin_idx, int32_t* oob_count, void* stream);

/* Same layer, same arguments, with the conditioner GEMMs on the f16 matrix cores in split-f16 form
 * (every f32 operand = hi + lo f16 pair, three v_mfma_f32_32x32x16_f16 per product, f32 accumulate): f32-class
 * accuracy (measured 0.47 ulp32 rms of sum|a||b| vs 0.65 for the f32 fma chain) at 16/3 times the MFMA rate;
 * not bit-identical to the f32 chain.  A0p/A1p/A2p: f16 operand blocks and c0..c2: per-layer power-of-two
 * unscale factors from bgflow_amd/dense.py::pack_dense_for_fused_h2 (layout in bgk_fused.hip / DESIGN.md); or, when the
 * operands were packed on the device by bgk_pack_dense_h2, cs_dev = its scale table (c0..c2 are then ignored).
 * operand_dtype 0: split-f16 (above); 1: single bf16 operands (bf16 parameter storage + bf16 GEMM inputs, f32
 * accumulate; knots, bin search and log-det stay f32) -- the reduced-precision variant of BASELINE config 5.
 * H0 = H1 = 128: the envelope of every variant.  H0 = H1 = 256 (conditioner_factory.py:76-80 takes any `hidden`; 129 .. 255 units
 * zero-padded by the packer): split-f16 inference (operand_dtype 0) with operands from bgflow_amd/dense.py::pack_dense_for_fused_w256
 * -- one wave per SIMD on the unified 512-register file (bgk_fused.hip::coupling_rqs_dense_w256_kernel); other widths: BGK_EUNSUPPORTED. */
int bgk_coupling_rqs_dense_h2(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                              const void* A0p, const void* A1p, const void* A2p,
                              float c0, float c1, float c2, const float* cs_dev, int32_t operand_dtype,
                              int32_t H0, int32_t H1, int32_t act,
                              const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K,
                              uint64_t circ_mask, int32_t inverse,
                              double left, double right, double bottom, double top,
                              double min_bin_width, double min_bin_height, double min_derivative,
                              int32_t identity_init,
                              float* out, int64_t ldo, float* dlogp, int32_t accumulate,
                              int32_t* bin_idx, int32_t* oob_count, void* stream);

/* The same layer for a conditioner with ANY number of hidden layers, n_hidden = 1 .. 8 (conditioner_factory.py:76-80 takes any `hidden`
 * tuple; nn/dense.py:30-48) of up to 128 units (narrower ones zero-padded by the packer): split-f16 inference, both directions,
 * K in {4, 8, 12, 16, 32}.  Operands from bgflow_amd/dense.py::pack_dense_for_fused_deep: A0p / A2p / c0 / c2 as above, A1p the
 * n_hidden - 1 hidden -> hidden layers back to back (68 blocks of 1 KiB each; NULL for one hidden layer), c1s their unscale factors
 * (HOST array of n_hidden - 1 floats).  Remaining arguments as bgk_coupling_rqs_dense_h2. */
int bgk_coupling_rqs_dense_deep(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                const void* A0p, const void* A1p, const void* A2p, float c0, const float* c1s, float c2,
                                int32_t n_hidden, int32_t act,
                                const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K,
                                uint64_t circ_mask, int32_t inverse,
                                double left, double right, double bottom, double top,
                                double min_bin_width, double min_bin_height, double min_derivative,
                                int32_t identity_init,
                                float* out, int64_t ldo, float* dlogp, int32_t accumulate,
                                int32_t* bin_idx, int32_t* oob_count, void* stream);

/* Training forward of the same layer: additionally writes what the backward pass needs -- the hidden layers'
 * pre-activations z0, z1 [B, 128] and the spline parameters [B, P] in the reference layout [w | h | s | s_nc]
 * (transformer/spline.py:113-126) -- so that autograd (KLTrainer.train, nn/training/trainers.py:158-170) can run
 * bgk_rqs_backward and the MLP backward on them.  src_col_dev: device copy of bgk_pack_rqs_columns' table.
 * params_layout = 1 (round 5): the parameters are written element-major, [B][d][3 K + 1] with ldp >= (3 K + 1) d + 3 -- the order
 * the kernel holds them in, so they leave as 16-byte pieces of contiguous runs (16 store instructions per 32 x 125 chunk instead of
 * 64; src_col_dev may be NULL); BGK_EUNSUPPORTED where the second-generation kernel does not run (K != 8, option 1 = 1): call again
 * with params_layout = 0.  bgk_rqs_backward takes the same flag. */
int bgk_coupling_rqs_dense_h2_train(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                    const void* A0p, const void* A1p, const void* A2p,
                                    float c0, float c1, float c2, const float* cs_dev,
                                    int32_t H0, int32_t H1, int32_t act,
                                    const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K,
                                    uint64_t circ_mask, int32_t inverse,
                                    double left, double right, double bottom, double top,
                                    double min_bin_width, double min_bin_height, double min_derivative,
                                    int32_t identity_init,
                                    float* out, int64_t ldo, float* dlogp, int32_t accumulate,
                                    int32_t* oob_count, float* z0, float* z1, float* params, int64_t ldp,
                                    const int32_t* src_col_dev, int32_t params_layout, void* stream);
/* params_layout = 2 (round 5): the parameters are NOT written (params may be NULL; 446 MB per layer at 2^18 samples x 17 dims,
 * 0.11 of the launch's 0.27 ms) -- the backward recomputes them from z1:
 *
 * bgk_coupling_rqs_dense_h2_backward: backward of the layer's spline transformer (autograd of transformer/spline.py:109-188 + nflows
 * through the parameters the conditioner's last Linear produced, nn/dense.py:47-48) without saved parameters: the output layer is
 * redone from z1 [B, 128] (contiguous, 16-byte aligned) on the matrix cores with the forward's operand A2p / scale c2 / device scale
 * table cs_dev (same blocks, f16 split and MFMA order: bit-identical parameters), then the VJP of bgk_rqs_backward per element.
 * circ_mask as in the forward call (P = 3 K d + the number of non-circular dims; a non-circular dim's slot is its rank among them).
 * Outputs as bgk_rqs_backward's: g_y [B, d], g_params [B, P] in the reference's column order, g_absmax.  BGK_EUNSUPPORTED outside
 * the fused envelope (hidden width 128, 8 bins, d <= 64) or with option 1 = 1. */
int bgk_coupling_rqs_dense_h2_backward(const float* z1, const void* A2p, float c2, const float* cs_dev, int32_t H1, int32_t act,
                                       const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K, int32_t P,
                                       uint64_t circ_mask, int32_t inverse,
                                       double left, double right, double bottom, double top,
                                       double min_bin_width, double min_bin_height, double min_derivative,
                                       int32_t identity_init, const float* g_out, int64_t ldgo, const float* g_dlogp,
                                       float* g_y, int64_t ldgy, float* g_params, int64_t ldgp, float* g_absmax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused affine coupling layer: replaces CouplingFlow._forward/_inverse (nn/flow/coupling.py:162-182) around
 * AffineTransformer (nn/flow/transformer/affine.py:41-70) when shift / scale conditioners are DenseNets
 * [n_in, H, H, d] (nn/dense.py:47-48; n_in = d_c, or 2 d_c behind the WrapPeriodic cos/sin featuriser when
 * `periodic` != 0) with H = 64 | 128, d <= 96, n_in <= 127: both MLPs on the f16 matrix cores
 * (split-f16, see bgk_coupling_rqs_dense_h2) + everything bgk_affine_transform does, one launch.
 *   s* / t*: shift / scale network operands from bgflow_amd/dense.py::pack_dense_for_affine_h2 (A0 == NULL: network
 *            absent), c0..c2 power-of-two unscale factors, act: 1 SiLU, 2 ReLU, 3 Tanh (hidden activations)
 * Returns BGK_EUNSUPPORTED outside the envelope: the caller runs the networks + bgk_affine_transform.
 * --------------------------------------------------------------------------------------------- */
int bgk_coupling_affine_dense_h2(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                 const void* sA0, const void* sA1, const void* sA2,
                                 float sc0, float sc1, float sc2, int32_t s_act,
                                 const void* tA0, const void* tA1, const void* tA2,
                                 float tc0, float tc1, float tc2, int32_t t_act,
                                 int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                 int32_t is_circular, int32_t inverse,
                                 const float* y, int64_t ldy, int64_t B, int32_t d,
                                 float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream);

/* Same layer with THREE hidden layers per conditioner (DenseNet [n_in, H, H, H, d], e.g. the ala2 RealNVP conditioners
 * [30, 128, 128, 128, 30] of the reference's examples; conditioner_factory.py:76-80 lets users choose the depth):
 * s/tA1b, s/tc1b = operands / unscale factor of the second H x H layer (packed like A1). */
int bgk_coupling_affine_dense_h3(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                 const void* sA0, const void* sA1, const void* sA1b, const void* sA2,
                                 float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                                 const void* tA0, const void* tA1, const void* tA1b, const void* tA2,
                                 float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                                 int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                 int32_t is_circular, int32_t inverse,
                                 const float* y, int64_t ldy, int64_t B, int32_t d,
                                 float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream);

/* The same layer for conditioners with ANY number of hidden layers, n_hidden = 1 .. 8 (both networks alike; README.md:72-79's
 * DenseNet([dim // 2, 4, dim // 2]) has one; conditioner_factory.py:76-80 takes any `hidden` tuple) of width 64 | 128 (narrower ones
 * zero-padded by the packer).  Operands from bgflow_amd/dense.py::pack_dense_for_affine_deep: sA0 / sA2 / sc0 / sc2 as above, sA1 the
 * n_hidden - 1 hidden -> hidden layers back to back (NULL for one hidden layer), sc1s their unscale factors (HOST array); likewise t*. */
int bgk_coupling_affine_dense_deep(const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                                   const void* sA0, const void* sA1, const void* sA2, float sc0, const float* sc1s, float sc2, int32_t s_act,
                                   const void* tA0, const void* tA1, const void* tA2, float tc0, const float* tc1s, float tc2, int32_t t_act,
                                   int32_t n_hidden, int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                   int32_t is_circular, int32_t inverse,
                                   const float* y, int64_t ldy, int64_t B, int32_t d,
                                   float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training forward of the fused affine coupling layer (round 6): what autograd needs of CouplingFlow._forward / _inverse
 * (nn/flow/coupling.py:162-182) around AffineTransformer (nn/flow/transformer/affine.py:35-70) with DenseNet conditioners
 * [n_in, H0, H1, d] (nn/dense.py:30-48), H0, H1 <= 128, in ONE launch: the arithmetic of bgk_coupling_affine_dense_h2 (width-128 kernel:
 * narrower hidden layers run zero-padded inside the operands) plus, per network, the scaled pre-activations z0, z1 [B, ldz] of its
 * two hidden layers (contiguous; ldz = 128, or 64 when no hidden layer has more than 64 units; padded units hold 0) and its output rows -- mu and the scale network's values before tanh,
 * [B, ldms] with ldms a multiple of 4 and >= 32 ceil(d / 32) (columns past d hold 0).  Their consumers: bgk_affine_backward
 * (mu, s_raw), bgk_dense_backward_dx and bgk_mlp_weight_grad (z0, z1) -- the backward of KLTrainer's loss.backward()
 * (nn/training/trainers.py:156-163) for an affine coupling without a library GEMM or an elementwise torch kernel.
 *   cond / ldc / width / n_cond: 1 .. 3 conditioning tensors standing for their concatenation (as the *_mc entry points)
 *   s* / t*: shift / scale network operands from bgk_pack_mlp_h2 (HT = 4, NT2 = ceil(d / 32), one group, identity row map) and its
 *            device scale table cs; A0 == NULL: network absent; act: 1 SiLU, 2 ReLU, 3 Tanh (both equal, or ReLU / Tanh)
 * Returns BGK_EUNSUPPORTED outside the envelope (activation pairs, > 127 input features): the caller runs the networks layer by layer.
 * --------------------------------------------------------------------------------------------- */
int bgk_coupling_affine_dense_h2_train(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                       const void* sA0, const void* sA1, const void* sA2, const float* s_cs, int32_t s_act,
                                       const void* tA0, const void* tA1, const void* tA2, const float* t_cs, int32_t t_act,
                                       const float* log_alpha, int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                                       const float* y, int64_t ldy, int64_t B, int32_t d, float* out, int64_t ldo,
                                       float* dlogp, int32_t accumulate,
                                       float* s_z0, float* s_z1, float* t_z0, float* t_z1, int64_t ldz,
                                       float* mu, float* s_raw, int64_t ldms, void* stream);

/* The fused coupling layers with SEVERAL conditioning tensors: CouplingFlow concatenates the tensors at cond_indices before it
 * calls the transformer (torch.cat, nn/flow/coupling.py:162-165; e.g. cfg 5's AUGMENTED | (FIXED, BONDS, ANGLES) layers).  Here
 * cond / ldc / width are HOST arrays of n_cond (1..3) device pointers [B, width_i], row strides and widths; the kernels stage
 * every tensor from its own rows, the concatenation is never materialised.  Everything else as in the single-tensor entry points
 * (d_c = sum of the widths).  BGK_EUNSUPPORTED for n_cond > 1 on kernels without the segment table (first-generation spline
 * kernel / K != 8, hidden width 64, exact-f32 mode): the caller concatenates and uses the single-tensor entry point. */
int bgk_coupling_rqs_dense_h2_mc(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                 const void* A0p, const void* A1p, const void* A2p,
                                 float c0, float c1, float c2, const float* cs_dev, int32_t operand_dtype,
                                 int32_t H0, int32_t H1, int32_t act,
                                 const float* y, int64_t ldy, int64_t B, int32_t d, int32_t K,
                                 uint64_t circ_mask, int32_t inverse,
                                 double left, double right, double bottom, double top,
                                 double min_bin_width, double min_bin_height, double min_derivative,
                                 int32_t identity_init,
                                 float* out, int64_t ldo, float* dlogp, int32_t accumulate,
                                 int32_t* bin_idx, int32_t* oob_count, void* stream);
int bgk_coupling_affine_dense_h2_mc(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                    const void* sA0, const void* sA1, const void* sA2,
                                    float sc0, float sc1, float sc2, int32_t s_act,
                                    const void* tA0, const void* tA1, const void* tA2,
                                    float tc0, float tc1, float tc2, int32_t t_act,
                                    int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                    int32_t is_circular, int32_t inverse,
                                    const float* y, int64_t ldy, int64_t B, int32_t d,
                                    float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream);
int bgk_coupling_affine_dense_h3_mc(const float* const* cond, const int64_t* ldc, const int32_t* width, int32_t n_cond, int32_t periodic,
                                    const void* sA0, const void* sA1, const void* sA1b, const void* sA2,
                                    float sc0, float sc1, float sc1b, float sc2, int32_t s_act,
                                    const void* tA0, const void* tA1, const void* tA1b, const void* tA2,
                                    float tc0, float tc1, float tc1b, float tc2, int32_t t_act,
                                    int32_t hidden, const float* log_alpha, int32_t preserve_volume,
                                    int32_t is_circular, int32_t inverse,
                                    const float* y, int64_t ldy, int64_t B, int32_t d,
                                    float* out, int64_t ldo, float* dlogp, int32_t accumulate, void* stream);

/* Device-side packing of a DenseNet [n_in, H, H, rows2] into split-f16 MFMA operands (no host synchronisation; used
 * whenever the weights change, i.e. every training step).  Layer 2's packed rows are row_map2_dev[packed row] (source
 * row or -1; NULL = identity), laid out as n_groups2 groups of NT2 32-row tiles, each group followed by its bias
 * blocks -- the spline kernel uses (row_map = bgk_pack_rqs_columns, n_groups = chunks, NT2 = 4), the affine kernel
 * (NULL, 1, ceil(d / 32)).  cs[6] receives {2^s, 2^-s} per layer.  Replaces bgflow_amd/dense.py::_pack_h2. */
int bgk_pack_dense_h2(const float* W0, const float* b0, int32_t n_in, int32_t H,
                      const float* W1, const float* b1,
                      const float* W2, const float* b2, int32_t rows2,
                      const int32_t* row_map2_dev, int32_t n_groups2, int32_t NT2, int32_t operand_dtype,
                      void* A0, void* A1, void* A2, float* cs, void* stream);

/* bgk_pack_dense_h2 (H = 128, NT2 = 4, split-f16 operands) for n conditioners in two launches per 16 of them: after an optimizer
 * step every coupling layer of a flow is re-packed; one by one that is a memset + three ~5 us launches per layer (0.4 ms of a
 * 17 ms KL step of the 16-layer flow).  Arrays of n host-side entries (device pointers inside); same results as n single calls. */
int bgk_pack_dense_h2_many(int32_t n, const float* const* W0, const float* const* b0, const int32_t* n_in,
                           const float* const* W1, const float* const* b1, const float* const* W2, const float* const* b2,
                           const int32_t* rows2, const int32_t* const* row_map2_dev, const int32_t* n_groups2,
                           void* const* A0, void* const* A1, void* const* A2, float* const* cs, void* stream);

/* bgk_pack_dense_h2 / _many for conditioners whose hidden layers have H0 / H1 <= 32 HT units (DenseNet([n_in, H0, H1, rows2]),
 * nn/dense.py:9-28; factory/conditioner_factory.py:81-85 takes any `hidden` tuple): the operands are laid out for a kernel that runs
 * 32 HT hidden rows, the rows / k-slots past H0 / H1 are zero (act(0) = 0 feeds zero columns: the same function), no padded copy of
 * the weights exists.  Affine networks: row_map2_dev = NULL, n_groups2 = 1, NT2 = ceil(d / 32).  _many: H0 / H1 / NT2 / HT may be NULL
 * (128 / 128 / 4 / 4 for every conditioner). */
int bgk_pack_mlp_h2(const float* W0, const float* b0, int32_t n_in, int32_t H0,
                    const float* W1, const float* b1, int32_t H1,
                    const float* W2, const float* b2, int32_t rows2,
                    const int32_t* row_map2_dev, int32_t n_groups2, int32_t NT2, int32_t HT,
                    void* A0, void* A1, void* A2, float* cs, void* stream);
int bgk_pack_mlp_h2_many(int32_t n, const float* const* W0, const float* const* b0, const int32_t* n_in, const int32_t* H0,
                         const float* const* W1, const float* const* b1, const int32_t* H1,
                         const float* const* W2, const float* const* b2,
                         const int32_t* rows2, const int32_t* const* row_map2_dev, const int32_t* n_groups2, const int32_t* NT2,
                         const int32_t* HT,
                         void* const* A0, void* const* A1, void* const* A2, float* const* cs, void* stream);

/* Input-gradient chain of the conditioner MLP [n_in, 128, 128, P] in one launch (autograd of nn/dense.py:47-48 in the
 * training step): from g [B, P] (gradient w.r.t. the MLP output, e.g. bgk_rqs_backward's g_params) and the saved
 * pre-activations z1, z0 it writes g_z1, g_z0 (gradients w.r.t. the pre-activations), h1, h0 (the activations; both NULL: not
 * written -- bgk_dense_weight_grad can recompute them from z1 / z0) -- the operands of the weight / bias gradient GEMMs -- and g_cond [B, d_c] (NULL to skip; periodic != 0: through the cos / sin
 * featuriser of nn/periodic.py:30-37, needs cond).  T0..T2: transposed-weight operands from bgk_pack_dense_h2_t with the
 * scale table cs of bgk_pack_dense_h2 for the same weights; sizes in 1 KiB blocks: T0 17 ceil(n_in / 32), T1 68,
 * T2 8 S2 + 4 with S2 = ceil(P / 16) rounded up to a multiple of 4 (zero blocks behind the last column).
 * Arithmetic (round 5): every GEMM multiplies f16 hi + lo operand pairs (22 significant bits per product, f32 accumulate) like the
 * forward; the gradient operands are split under a power-of-two scale -- g under 2^s with max |g| 2^s in [2^14, 2^15), max |g| read from
 * g_absmax[0] (device; what bgk_rqs_backward / bgk_absmax wrote; NULL: unscaled, values below 6e-5 lose bits), the tiles g_z1 / g_z0
 * that feed the next GEMM under the scale of the tile's own maximum.  gz_absmax (device, may be NULL): [0] / [1] are raised to the
 * largest |g_z1| / |g_z0| written (zero them first) -- the scales bgk_dense_weight_grad needs.
 * g_cond_add (may be NULL; row stride ldga): a [B, d_c] tensor added to the conditioner-input gradient on its way out,
 * g_cond = g_cond_add + (chain result) -- the sum autograd forms with an elementwise launch when the conditioning tensor has other
 * consumers (a later coupling transforms it, another one is conditioned on it too); g_cond_add may BE g_cond (every element is read
 * and written by the same lane). */
int bgk_pack_dense_h2_t(const float* W0, int32_t n_in, const float* W1, const float* W2, int32_t P,
                        const float* cs, void* T0, void* T1, void* T2, void* stream);
/* bgk_pack_dense_h2_t / _many for hidden layers of H0 / H1 <= 128 units (W0 [H0, n_in], W1 [H1, H0], W2 [P, H1]): the transposed operands
 * of bgk_dense_backward_dx, which runs 128 hidden units -- the others are zero in the operands (round 6: the shift / scale networks
 * of an affine coupling, P = d).  _many: H0 / H1 may be NULL (128 everywhere). */
int bgk_pack_mlp_h2_t(const float* W0, int32_t n_in, int32_t H0, const float* W1, int32_t H1, const float* W2, int32_t P,
                      const float* cs, void* T0, void* T1, void* T2, void* stream);
int bgk_pack_mlp_h2_t_many(int32_t n, const float* const* W0, const int32_t* n_in, const int32_t* H0, const float* const* W1,
                           const int32_t* H1, const float* const* W2, const int32_t* P, const float* const* cs,
                           void* const* T0, void* const* T1, void* const* T2, void* stream);

/* bgk_pack_dense_h2_t for n conditioners in one launch per 16 of them (same results as n single calls) */
int bgk_pack_dense_h2_t_many(int32_t n, const float* const* W0, const int32_t* n_in, const float* const* W1,
                             const float* const* W2, const int32_t* P, const float* const* cs,
                             void* const* T0, void* const* T1, void* const* T2, void* stream);
/* bgk_mlp_backward_dx (round 6): bgk_dense_backward_dx (below) with the row pitch ldz of z1 / z0 / g_z1 / g_z0 / h1 / h0 as an argument
 * -- 128, or 64 for networks whose hidden layers have <= 64 units (an affine coupling's [32, 64, 64, 32] networks: half the bytes of the
 * zero-padded form; the operands of bgk_pack_mlp_h2_t hold zeros for the units that do not exist) */
int bgk_mlp_backward_dx(const float* g, int64_t ldg, int32_t P, const float* z1, const float* z0, int64_t ldz,
                        const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                        const void* T0, const void* T1, const void* T2, const float* cs, int32_t act,
                        int64_t B, float* g_z1, float* g_z0, float* h1, float* h0,
                        float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga,
                        const float* g_absmax, float* gz_absmax, void* stream);
int bgk_dense_backward_dx(const float* g, int64_t ldg, int32_t P, const float* z1, const float* z0,
                          const float* cond, int64_t ldc, int32_t d_c, int32_t periodic,
                          const void* T0, const void* T1, const void* T2, const float* cs, int32_t act,
                          int64_t B, float* g_z1, float* g_z0, float* h1, float* h0,
                          float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga,
                          const float* g_absmax, float* gz_absmax, void* stream);

/* Optimizer step on the flat parameter bucket (f-2: the optimizer of KLTrainer.train, nn/training/trainers.py:148-201).
 * bgk_grad_nan_flag sets flag[0] = any(isnan(g)) on the device (the reference's "found nan in grad; skipping optimization
 * step" check, trainers.py:198-201, without a host round trip); bgk_adam_step is torch.optim.Adam's update (step = 1, 2, ...:
 * `step` counts the calls; the time step of the bias corrections 1 - beta^t is t = step - skipped_count[0], read on the device, so a
 * skipped call does not advance it -- like the reference, which does not call optim.step() then) over [n] contiguous floats, a
 * no-op that increments skipped_count[0] when skip_flag[0] != 0 (either pointer may be NULL). */
int bgk_grad_nan_flag(const float* g, int64_t n, int32_t* flag, void* stream);
int bgk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, const int32_t* skip_flag, int32_t* skipped_count, void* stream);

/* Energy of the isotropic, optionally shifted normal distribution (bgflow/distribution/normal.py:61-72, NormalDistribution._energy
 * without `cov`; the target end of the KL integrand u(F(z)) - log|det J|, bgflow/bg.py:13-17, for the synthetic Gaussian targets,
 * and the prior energy of cfg 1 / cfg 2):  u[b] = 0.5 sum_j (x[b,j] - mean[j])^2 / temperature + log_z   (mean may be NULL;
 * log_z = d / 2 log(2 pi T) from the caller), and its VJP  g_x[b,j] = g_u[b] (x[b,j] - mean[j]) / temperature. */
int bgk_normal_energy(const float* x, int64_t ldx, const float* mean, int32_t d, int64_t B,
                      double temperature, double log_z, float* u, void* stream);
int bgk_normal_energy_backward(const float* x, int64_t ldx, const float* mean, int32_t d, int64_t B,
                               double temperature, const float* g_u, float* g_x, int64_t ldg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Priors and targets around the flow, one launch over all tensors of a sample.
 * bgk_energy_fields: u[b] = (sum_f e_f(x_f[b]) + c_in) / temperature + c_out over n_fields <= BGK_MAX_ENERGY_FIELDS tensors
 *   (HOST arrays x / ldx / d / kind / param / coef[3 n_fields] of device pointers, row strides, widths, ...):
 *   kind 0  0.5 sum_j (x_j - param_j)^2            NormalDistribution without cov (distribution/normal.py:61-72; param = mean or NULL)
 *   kind 1  a x_0 + b x_0^2 + c x_0^4 + 0.5 sum_{j>=1} x_j^2   DoubleWellEnergy (distribution/energy/double_well.py:17-22; coef = a, b, c)
 *   kind 2  the constant coef[0]                    UniformDistribution (distribution/distributions.py:100-117); x not read
 *   ProductEnergy / ProductDistribution (distribution/product.py:13-117) = the list of its components' fields.
 *   loss_sums != NULL (needs dlogp [B], workspace partial [nblk, 2] floats): additionally the KL integrand u - dlogp (bg.py:13-17) is
 *   summed over the batch, loss_sums[0] = sum, loss_sums[1] = number of samples kept (non-finite ones dropped if drop_nonfinite),
 *   both double, fixed summation order -- what dp.global_mean all-reduces.
 * bgk_energy_fields_backward: g_x_f = g_row (de_f / dx) / temperature for every field with g_x[f] != NULL; g_row = g_u[b], or, in the
 *   loss-sum form (g_u == NULL), g_scalar[0] for the kept samples, with g_dlogp[b] = -g_row.
 * bgk_normal_energy(_backward): the single-field forms (ABI of round 2). */
#define BGK_MAX_ENERGY_FIELDS 8
int bgk_energy_fields(const float* const* x, const int64_t* ldx, const int32_t* d, const int32_t* kind,
                      const float* const* param, const float* coef, int32_t n_fields, int64_t B,
                      double temperature, double c_in, double c_out, float* u,
                      const float* dlogp, int32_t drop_nonfinite, float* partial, int32_t nblk, double* loss_sums, void* stream);
int bgk_energy_fields_backward(const float* const* x, const int64_t* ldx, const int32_t* d, const int32_t* kind,
                               const float* const* param, const float* coef, int32_t n_fields, int64_t B,
                               double temperature, const float* g_u,
                               const float* g_scalar, const float* u, const float* dlogp, int32_t drop_nonfinite, float* g_dlogp,
                               float* const* g_x, const int64_t* ldg, void* stream);

/* Prior sampling in one launch from a counter-based generator (Philox4x32-10; counter = (global row, field, 4-column block, offset),
 * key = seed: independent of launch geometry and of the sharding of a batch, row0 = first global row of this launch), replacing
 * torch.randn / Uniform.sample + the shift / scale ops of NormalDistribution._sample_with_temperature (distribution/normal.py:74-92),
 * UniformDistribution._sample (distributions.py:116-117) and ProductDistribution.sample (product.py:84-117); an opt-in path of the
 * python priors (sample_fused=True).  kind 0: low + u (high - low) with p0 = low, p1 = high (NULL: 0 / 1); kind 1: p0 (mean or NULL) +
 * scale * n.  energy != NULL: energy[b] = sum_f (0.5 sum_j n_j^2 [kind 1] + e_const[f]) + c_out = the prior energy of the sample. */
int bgk_philox_fields(uint64_t seed, uint32_t offset, int64_t row0, int32_t n_fields, float* const* out, const int64_t* ldo,
                      const int32_t* d, const int32_t* kind, const float* const* p0, const float* const* p1,
                      const float* scale, const float* e_const, double c_out, int64_t B, float* energy, void* stream);

/* Weight and bias gradients of the conditioner MLP [n_in, 128, 128, P] of one coupling layer (autograd of nn/dense.py:47-48 in
 * the training step: dW = g^T h, db = sum over the batch of g) from the tensors bgk_rqs_backward / bgk_dense_backward_dx wrote:
 *   (g_params [B, P], h1) -> gW2 [P, 128], gb2 [P];  (g_z1, h0) -> gW1 [128, 128], gb1 [128];
 *   (g_z0, featurised cond) -> gW0 [128, n_in], gb0 [128]   (periodic != 0: cond [B, d_c] is featurised on the fly as
 *   [cos 2 pi x | sin 2 pi x], nn/periodic.py:30-37; n_in = 2 d_c).  Any gW pointer may be NULL (skipped with its gb);
 *   accumulate != 0: results are ADDED to the destinations (gradient buckets of the optimizer).
 *   h_act = 0: h1 / h0 hold the hidden activations; 1 SiLU | 2 ReLU | 3 Tanh: they hold the PRE-activations z1 / z0 saved by the
 *   training forward and the kernel applies the activation while loading (bgk_dense_backward_dx then need not write h1 / h0:
 *   a fifth less HBM traffic in that kernel).
 * Split over the batch into slabs whose partials are summed in fixed order (deterministic); workspace size in floats from
 * bgk_dense_weight_grad_workspace.  Replaces 3 split-K hipBLASLt GEMMs + 3 reductions + 6 column-sum launches per layer.
 * Arithmetic (round 5): products of f16 hi + lo operand pairs (22 significant bits, f32 accumulate; rounds 1 - 4: bf16 pairs, 16 bits);
 * g_absmax (device, may be NULL) = {max |g_params|, max |g_z1|, max |g_z0|} as published by bgk_rqs_backward (g_absmax) and
 * bgk_dense_backward_dx (gz_absmax) or computed by bgk_absmax: each gradient tensor is split under the power of two that puts its
 * maximum into [2^14, 2^15) (NULL: unscaled -- gradient values below 6e-5 lose bits). */
/* accumulate = 2: only the partial sums are formed (deterministic slabs in the workspace); bgk_dense_weight_grad_reduce_many then
 * reduces the partial sets of n layers -- same B, P, n_in, workspace and destinations as their bgk_dense_weight_grad calls, all
 * six destinations of a layer given -- in ONE launch per 16 layers (accumulate there: 0 overwrite, 1 add to the destinations): the
 * backward pass of a 16-layer flow ends with one reduction instead of sixteen 25 us launches. */
int64_t bgk_dense_weight_grad_workspace(int64_t B, int32_t P, int32_t n_in);
int bgk_dense_weight_grad(const float* g_params, int64_t ldg, int32_t P, const float* g_z1, const float* g_z0,
                          const float* h1, const float* h0, int32_t h_act, const float* cond, int64_t ldc, int32_t d_c,
                          int32_t periodic, int64_t B, float* workspace, int64_t workspace_floats,
                          float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0, int32_t accumulate,
                          const float* g_absmax, void* stream);
int bgk_dense_weight_grad_reduce_many(int32_t n, const int64_t* B, const int32_t* P, const int32_t* n_in,
                                      float* const* workspace, float* const* gW2, float* const* gb2, float* const* gW1,
                                      float* const* gb1, float* const* gW0, float* const* gb0, int32_t accumulate, void* stream);

/* The general form of bgk_dense_weight_grad (round 6): hidden layers of H1 / H0 <= 128 units whose pre-activations / gradients sit in
 * arrays of row pitch ldz (the [B, 128] arrays of the fused training forwards); the gradients come out in the parameters' own shapes
 * gW2 [P, H1], gW1 [H1, H0], gW0 [H0, n_in] -- autograd of nn/dense.py:47-48 for the shift / scale networks of an affine coupling
 * (nn/flow/transformer/affine.py:35-43), P = d.  Arguments otherwise as bgk_dense_weight_grad; _reduce_many: H1 / H0 may be NULL (128). */
int64_t bgk_mlp_weight_grad_workspace(int64_t B, int32_t P, int32_t H1, int32_t H0, int32_t n_in);
int bgk_mlp_weight_grad(const float* g_out, int64_t ldg, int32_t P, const float* g_z1, const float* g_z0,
                        const float* h1, const float* h0, int64_t ldz, int32_t H1, int32_t H0, int32_t h_act,
                        const float* cond, int64_t ldc, int32_t d_c, int32_t periodic, int64_t B,
                        float* workspace, int64_t workspace_floats,
                        float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0, int32_t accumulate,
                        const float* g_absmax, void* stream);
int bgk_mlp_weight_grad_reduce_many(int32_t n, const int64_t* B, const int32_t* P, const int32_t* H1, const int32_t* H0,
                                    const int32_t* n_in, float* const* workspace, float* const* gW2, float* const* gb2,
                                    float* const* gW1, float* const* gb1, float* const* gW0, float* const* gb0,
                                    int32_t accumulate, void* stream);

/* Column sums of a row-major [B, P] matrix: out[c] = sum_r x[r, c] -- the bias gradient of a Linear layer
 * (autograd of the conditioner MLP, nn/dense.py:47-48, inside KLTrainer.train, nn/training/trainers.py:158-170).
 * Deterministic two-stage reduction; `partial` is a caller-provided [nblk, P] workspace. */
/* bgk_coupling_affine_dense_fwd64_train (round 6): bgk_coupling_affine_dense_h2_train for networks with hidden layers of <= 64 units,
 * d <= 32 transformed dims and ONE non-periodic conditioning tensor of n_in <= 32 columns (BASELINE cfg 2's couplings,
 * nn/flow/transformer/affine.py:35-70 with DenseNet([32, 64, 64, 32]) networks) on a kernel sized for them: operands packed for 64
 * hidden rows (bgk_pack_mlp_h2 with HT = 2, NT2 = 1) resident in LDS, three waves per SIMD; z0 / z1 [B, 64] contiguous, mu / s_raw
 * [B, ldms].  Same outputs, bit for bit (same products in the same order).  BGK_EUNSUPPORTED outside the envelope.
 * mu NULL: the shift values are not saved (bgk_affine_coupling_backward64 does not read them); all six save pointers NULL: nothing is
 * written for the backward (bgk_affine_coupling_backward64 recomputes the networks). */
int bgk_coupling_affine_dense_fwd64_train(const float* cond, int64_t ldc, int32_t n_in,
                                          const void* sA0, const void* sA1, const void* sA2, const float* s_cs, int32_t s_act,
                                          const void* tA0, const void* tA1, const void* tA2, const float* t_cs, int32_t t_act,
                                          const float* log_alpha, int32_t preserve_volume, int32_t is_circular, int32_t inverse,
                                          const float* y, int64_t ldy, int64_t B, int32_t d, float* out, int64_t ldo,
                                          float* dlogp, int32_t accumulate,
                                          float* s_z0, float* s_z1, float* t_z0, float* t_z1, float* mu, float* s_raw, int64_t ldms,
                                          void* stream);

/* bgk_affine_net_backward64 (round 6): backward of ONE conditioner network [n_in, H0, H1, d] of an affine coupling with hidden layers
 * of <= 64 units, d <= 32 transformed dims and n_in <= 32 (non-periodic) inputs -- BASELINE cfg 2's DenseNet([32, 64, 64, 32]) shift /
 * scale networks -- in ONE launch + one reduction: the input-gradient chain of bgk_mlp_backward_dx AND the weight / bias gradients of
 * bgk_mlp_weight_grad, with g_z1 / g_z0 never written to memory: the weight gradients contract over the batch, a wave forms its tiles'
 * share g^T h on chip (operands transposed through LDS) and keeps the network's three gradients in 128 accumulator registers across all
 * its tiles; per-wave partials in `workspace` (bgk_affine_net_backward64_workspace floats), summed in fixed order.  Autograd of
 * nn/dense.py:47-48 behind nn/flow/transformer/affine.py:35-43 in loss.backward() (nn/training/trainers.py:156-163).
 *   g [B, d] (row stride ldg): gradient w.r.t. the network's output; z1, z0 [B, 64] contiguous (bgk_coupling_affine_dense_h2_train, ldz = 64)
 *   T0, T1, T2, cs: the transposed operands / scale table of bgk_pack_mlp_h2_t; g_absmax: device float[1] = max |g| (or NULL)
 *   g_cond (may be NULL) = W0^T g_z0 + g_cond_add; gW2 [d, H1], gW1 [H1, H0], gW0 [H0, n_in] + biases: written, or (accumulate = 1) added to
 * BGK_EUNSUPPORTED outside the envelope: the caller runs bgk_mlp_backward_dx + bgk_mlp_weight_grad. */
int64_t bgk_affine_net_backward64_workspace(int64_t B, int32_t d, int32_t H1, int32_t H0, int32_t n_in);
int bgk_affine_net_backward64(const float* g, int64_t ldg, int32_t d, const float* z1, const float* z0,
                              const float* cond, int64_t ldc, int32_t n_in, int32_t H1, int32_t H0,
                              const void* T0, const void* T1, const void* T2, const float* cs, int32_t act, int64_t B,
                              float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga, const float* g_absmax,
                              float* workspace, int64_t workspace_floats,
                              float* gW2, float* gb2, float* gW1, float* gb1, float* gW0, float* gb0, int32_t accumulate,
                              void* stream);

/* bgk_affine_coupling_backward64 (round 6): the WHOLE backward of an affine coupling (either direction, no volume preservation) with two
 * networks inside bgk_affine_net_backward64's envelope in two launches (+ their reductions).  The scale network's launch forms
 * g_y = g_out e^s and g_s_raw = (g_out e^s y + g_dlogp) alpha (1 - tanh^2 s_raw) on chip (bgk_affine_backward's arithmetic: that launch,
 * its g_mu / g_s arrays and its atomics do not exist) and runs the network's backward; the shift network's takes g_mu = g_out as it is.
 * inverse = 1 (the layer ran out = (y - mu) e^-s, dlogp = - sum s): `y` is the layer's OUTPUT (g_s = - g_out out - g_dlogp needs nothing
 * else), g_y = g_out e^-s, and g_mu = - g_y goes through the [B, d] buffer g_mu (pitch ldgy) to the shift network's launch.
 *   s_z0, s_z1, t_z0, t_z1 [B, 64], s_raw [B, lds]: what bgk_coupling_affine_dense_fwd64_train saved (mu is not needed) -- or ALL FIVE
 *   NULL: nothing was saved (that entry point with NULL save pointers), every wave recomputes the networks' forward on its tile (the
 *   forward's products in the forward's order).  HBM per sample and layer for BASELINE cfg 2, forward included: 4.4 KB with
 *   bgk_affine_backward + 2 x bgk_affine_net_backward64, 3.3 KB here with the saved arrays, 1.6 KB with recomputation.  Autograd of
 * nn/flow/transformer/affine.py:35-70 + nn/dense.py:30-48 + nn/flow/coupling.py:152-182 under loss.backward().
 *   sA0, sA1 / tA0, tA1, tA2: forward operands of bgk_pack_mlp_h2 (HT = 2); sT* / tT*: transposed operands of bgk_pack_mlp_h2_t; *_cs: scale tables
 *   g_y [B, d] written; g_cond (may be NULL) = both networks' conditioner-input gradients + g_cond_add; g_log_alpha: float[1]
 *   s_grads / t_grads: HOST arrays of six device pointers (gW2, gb2, gW1, gb1, gW0, gb0; NULL: not wanted); accumulate = 1: added to
 *   (the weight gradients and g_log_alpha alike).  Deterministic (fixed-order partial sums, no atomics). */
int64_t bgk_affine_coupling_backward64_workspace(int64_t B, int32_t d, int32_t n_in, int32_t sH1, int32_t sH0, int32_t tH1, int32_t tH0);
int bgk_affine_coupling_backward64(const float* cond, int64_t ldc, int32_t n_in, const float* y, int64_t ldy, int32_t d,
                                   const float* g_out, int64_t ldgo, const float* g_dlogp,
                                   const float* s_z0, const float* s_z1, const float* t_z0, const float* t_z1, const float* s_raw, int64_t lds,
                                   const void* sA0, const void* sA1, const void* sT0, const void* sT1, const void* sT2,
                                   const float* s_cs, int32_t s_act, int32_t sH1, int32_t sH0,
                                   const void* tA0, const void* tA1, const void* tA2, const void* tT0, const void* tT1, const void* tT2,
                                   const float* t_cs, int32_t t_act, int32_t tH1, int32_t tH0,
                                   const float* log_alpha, int32_t inverse, int64_t B,
                                   float* g_y, int64_t ldgy, float* g_mu, float* g_cond, int64_t ldgc, const float* g_cond_add, int64_t ldga,
                                   float* g_log_alpha, float* workspace, int64_t workspace_floats,
                                   float* const* s_grads, float* const* t_grads, int32_t accumulate, void* stream);

/* bgk_linear_weight_grad (round 6): ONE Linear layer of any width -- gW [n, k] = g^T h (row-major, contiguous), gb [n] = sum_rows g
 * (may be NULL) for g [B, n], h [B, k]: autograd of nn/dense.py:47-48 for a Linear outside the fused training envelopes (conditioners
 * of other depths / widths, factory/conditioner_factory.py:81-85; a stand-alone DenseNet).  Split-f16 products under the power-of-two
 * scale of g_absmax (device float[1] = max |g|, or NULL), deterministic two-stage reduction; workspace from
 * bgk_linear_weight_grad_workspace; accumulate = 1: added to gW / gb.  Before round 6: torch.bmm + sum on hipBLASLt. */
int64_t bgk_linear_weight_grad_workspace(int64_t B, int32_t n, int32_t k);
int bgk_linear_weight_grad(const float* g, int64_t ldg, int32_t n, const float* h, int64_t ldh, int32_t k, int64_t B,
                           float* workspace, int64_t workspace_floats, float* gW, float* gb, int32_t accumulate,
                           const float* g_absmax, void* stream);

int bgk_column_sum(const float* x, int64_t ldx, int64_t B, int32_t P, float* partial, int32_t nblk,
                   float* out, void* stream);

/* out[0] = max(out[0], max |x[r, c]|) over a row-major [B, P] matrix (out[0] >= 0 on entry, e.g. zeroed): the per-tensor power-of-two
 * scale of the backward GEMMs (bgk_dense_backward_dx, bgk_dense_weight_grad) for a gradient tensor that did not come out of a
 * kernel of this library (those publish their own maximum: the g_absmax / gz_absmax arguments below). */
int bgk_absmax(const float* x, int64_t ldx, int64_t B, int32_t P, float* out, void* stream);

/* One Linear layer (+ bias + activation) of a conditioner network on its own: y = act(x W^T + b).
 * Replaces one `Linear (, activation)` pair of DenseNet._layers (nn/dense.py:30-48) for the networks the one-launch coupling kernels do
 * not take (conditioner_factory.py:76-80 allows any `hidden` tuple: other depths, layers wider than 256; README.md:72-79's [1, 4, 1]
 * nets; couplings of more than 64 transformed dims): the layer-by-layer path is bgk_dense_layer per layer + bgk_rqs_transform /
 * bgk_affine_transform.  Split-f16 GEMM on the f16 matrix cores (f32-class: hi + lo f16 operand pairs, f32 accumulate), the input tile
 * under a per-32-sample power-of-two scale (inputs of any range), f32 bias and activation on the accumulators.
 *   x [B, n_in] (ldx): this pass' input columns, n_in <= 256 -- a wider input runs as passes over column blocks of 256 with
 *         accumulate = 1 from the second pass on (bias / act given to the last pass only)
 *   Ap: the weights W[:, block] packed by bgk_pack_linear_layer (below) or bgflow_amd/dense.py::pack_linear_layer (ceil(n_out / 128)
 *         groups x S k16-steps x 4 tiles x {hi, lo} blocks of 1 KiB, natural k order, zero-padded), S = bgk_dense_layer_steps(n_in);
 *         c = 2^-s its unscale factor, or c_dev = the packer's device scale pair (c_dev[1] is read instead of c)
 *   bias [n_out] or NULL; act: 0 none, 1 SiLU, 2 ReLU, 3 Tanh;  y [B, n_out] (ldy); accumulate: y = act(y + x W^T + b). */
int bgk_dense_layer(const float* x, int64_t ldx, int64_t B, int32_t n_in, const void* Ap, int32_t S, float c, const float* c_dev,
                    const float* bias, int32_t n_out, int32_t act, float* y, int64_t ldy, int32_t accumulate, void* stream);

/* Operands of bgk_dense_layer for one column block W[:, k0 : k0 + n_in] (pass W + k0, ldw = the weight's row stride; n_in <= 256) of a
 * Linear layer's weight [n_out, *] (nn/dense.py:30-48), packed on the device without a host synchronisation: Ap receives
 * ceil(n_out / 128) * bgk_dense_layer_steps(n_in) * 8 blocks of 1 KiB, cs[0] the block's largest magnitude, cs[1] the unscale factor. */
int bgk_pack_linear_layer(const float* W, int64_t ldw, int32_t n_out, int32_t n_in, void* Ap, float* cs, void* stream);

/* k16-steps the kernel instance for n_in input columns runs (1, 2, 4, 8, 12 or 16; the packer pads to it); -1 beyond 256 columns. */
int bgk_dense_layer_steps(int32_t n_in);
/* bgk_refresh_linear_layer (round 6): the operands of bgk_pack_linear_layer kept in step with the weights ON THE DEVICE -- one small
 * launch that fingerprints the column block (64-bit, position-mixed sum of the bit patterns) and re-packs Ap / cs only when the
 * fingerprint differs from the one in `state` (two device uint64, zero-initialised by the caller: {fingerprint, valid}).  Launched in
 * front of every bgk_dense_layer call by the host mirror: a DenseNet's forward then follows ANY update of its Linear weights
 * (nn/dense.py:30-48 reads the live parameter) -- also those torch's version counter does not see (`p.data` edits, kernels writing
 * through a view). */
int bgk_refresh_linear_layer(const float* W, int64_t ldw, int32_t n_out, int32_t n_in, int32_t transposed, void* Ap, float* cs, void* state,
                             void* stream);
/* transposed != 0: the operands of W^T -- element (row, k) = W[k * ldw + row], n_out rows, n_in <= 256 columns -- so that the input
 * gradient of the layer, dX = g W (autograd of nn/dense.py:47-48), is one more bgk_dense_layer call on g.
 *
 * bgk_activation / bgk_activation_backward (round 6): the hidden activation of nn/dense.py:30-48 (1 SiLU, 2 ReLU, 3 Tanh; the forms of
 * bgk_dense_layer's epilogue) and its VJP g_z = g * act'(z) as elementwise kernels: what the layer-by-layer TRAINING path of a
 * conditioner outside the one-launch envelopes runs between its Linear layers (before: aten silu / tanh / threshold kernels). */
int bgk_activation(const float* z, int64_t ldz, int64_t B, int32_t n, int32_t act, float* out, int64_t ldo, void* stream);
int bgk_activation_backward(const float* z, int64_t ldz, const float* g, int64_t ldg, int64_t B, int32_t n, int32_t act,
                            float* g_z, int64_t ldgz, void* stream);

/* Static PCA whitening / blackening of a coordinate block on its own: out = (x - pre) T + post.
 * Replaces WhitenFlow._whiten / _blacken (nn/flow/pca.py:74-93: torch.matmul(x - X0mean, Twhiten), torch.matmul(z, Tblacken) + X0mean);
 * the constant log-det -+ sum log std is formed by the caller.  The VJP w.r.t. x is the same call on the transposed matrix.
 *   x [B, n_in] (ldx), T [n_in, n_out] row-major, pre [n_in] or NULL, post [n_out] or NULL, out [B, n_out] (ldo).
 * n_in, n_out <= 128; BGK_EUNSUPPORTED beyond (a plain library GEMM then). */
int bgk_whiten(const float* x, int64_t ldx, const float* T, const float* pre, const float* post,
               int32_t n_in, int32_t n_out, int64_t B, float* out, int64_t ldo, void* stream);

/* number of packed columns NCp for (d, K) and the source column (in the reference's params
 * layout, P = 3*K*d + n_nc) of every packed column, -1 for padding.  HOST function:
 * src_col is a host int32[NCp] buffer (pass NULL to query NCp only). */
int32_t bgk_pack_rqs_columns(int32_t d, int32_t K, const int32_t* nc_slot_host, int32_t* src_col);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* BGFLOW_AMD_H */
