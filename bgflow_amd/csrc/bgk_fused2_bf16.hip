/* bgk_fused2_bf16.hip -- the second-generation spline coupling kernel in REDUCED-PRECISION mode gemm_mode = "bf16"
 * (bgk_fused2.hip compiled with BGK_V2_BF16 = 1): weights and GEMM inputs are bf16 (operands packed by bgk_pack_dense_h2 with
 * operand_dtype = 1), ONE v_mfma_f32_32x32x16_bf16 per product instead of the three split-f16 ones, f32 accumulation; knots,
 * bin search and log-det stay f32.  The "bf16" leg of BASELINE config 5 -- never the headline (log-det error 1.2e-4). */
#define BGK_V2_BF16 1
#include "bgk_fused2.hip"
