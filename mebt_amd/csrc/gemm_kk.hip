// bf16 GEMM kernels, operand layout A KC x B KC (see gemm_kernels.h)
#define MEBT_GEMM_AK true
#define MEBT_GEMM_BK true
#define MEBT_GEMM_TAG kk
#define MEBT_GEMM_PAIR
#include "gemm_layout.inc"
