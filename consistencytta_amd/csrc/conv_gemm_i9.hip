// Tile variants of conv_gemm_kernel, group 9 (see conv_gemm_kernel.h: one translation unit per group so that the
// variants compile in parallel): the slab mode of the big tile (stride-1 1-D convolutions).
#include "conv_gemm_kernel.h"

CTTA_CONV_VARIANTS_9(CTTA_CONV_INSTANTIATE)
