// Tile variants of conv_gemm_kernel, group 6 of 6 (see conv_gemm_kernel.h: one translation unit per group so that the
// variants compile in parallel).
#include "conv_gemm_kernel.h"

CTTA_CONV_VARIANTS_6(CTTA_CONV_INSTANTIATE)
