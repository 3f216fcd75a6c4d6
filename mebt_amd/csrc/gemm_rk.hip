// bf16 GEMM kernels, operand layout A RC x B KC (see gemm_kernels.h)
#define MEBT_GEMM_AK false
#define MEBT_GEMM_BK true
#define MEBT_GEMM_TAG rk
#include "gemm_layout.inc"
