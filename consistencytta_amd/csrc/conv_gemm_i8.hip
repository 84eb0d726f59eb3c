// Tile variants of conv_gemm_kernel, group 8 (see conv_gemm_kernel.h: one translation unit per group so that the
// variants compile in parallel): deeper LDS rings for the thin K-heavy launches.
#include "conv_gemm_kernel.h"

CTTA_CONV_VARIANTS_8(CTTA_CONV_INSTANTIATE)
