// bf16 GEMM kernels, operand layout A KC x B RC (see gemm_kernels.h)
#define MEBT_GEMM_AK true
#define MEBT_GEMM_BK false
#define MEBT_GEMM_TAG kr
#define MEBT_GEMM_PAIR
#include "gemm_layout.inc"
