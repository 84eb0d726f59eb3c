// Tile variants of conv_gemm_kernel, group 8 of 8 (see conv_gemm_kernel.h: one translation unit per group so that the
// variants compile in parallel): the register-prefetch (MODE 3) tiles.
#include "conv_gemm_kernel.h"

CTTA_CONV_VARIANTS_8(CTTA_CONV_INSTANTIATE)
