// bf16 GEMM kernels, operand layout A RC x B RC (see gemm_kernels.h)
#define MEBT_GEMM_AK false
#define MEBT_GEMM_BK false
#define MEBT_GEMM_TAG rr
#define MEBT_GEMM_GROUPED
#include "gemm_layout.inc"
