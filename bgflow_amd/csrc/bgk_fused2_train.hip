/* bgk_fused2_train.hip -- training forward of the one-launch spline coupling layer on the second-generation kernel
 * (bgk_fused2.hip compiled with BGK_V2_SAVE = 1): same arithmetic and threading of the MFMA stream, and in addition the kernel
 * writes what the analytic backward kernels read -- the scaled pre-activations z0, z1 [B, 128] (as full rows through the LDS
 * chunk buffer, which is free while layers 0 / 1 run) and the spline parameters [B, P] (row-wise out of the LDS chunk after
 * its spline, chunk row stride 33).  Replaces coupling_rqs_dense_h2_kernel<.., SAVE = true> (bgk_fused.hip) behind
 * bgk_coupling_rqs_dense_h2_train for K = 8. */
#define BGK_V2_SAVE 1
#define BGK_V2_KARG 1       /* spline constants by scalar loads at their uses: this variant's extra pointers leave no SGPRs for them */
#include "bgk_fused2.hip"
