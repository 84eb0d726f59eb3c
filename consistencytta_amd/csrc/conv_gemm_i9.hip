// Tile variants of conv_gemm_kernel.h, group 9: the stream-K kernels (one translation unit per group so that the variants
// compile in parallel).
#include "conv_gemm_kernel.h"

CTTA_CONV_VARIANTS_9(CTTA_CONV_INSTANTIATE_K)
