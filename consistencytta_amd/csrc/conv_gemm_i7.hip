// Tile variants of conv_gemm_kernel, group 7 of 7 (see conv_gemm_kernel.h: one translation unit per group so that the
// variants compile in parallel).
#include "conv_gemm_kernel.h"

CTTA_CONV_VARIANTS_7(CTTA_CONV_INSTANTIATE)
