// The slab fold of a weight-gradient layer -- grad_w[row_off[n] + col(k)] += sum_s slab[s][n][k], grad_b[bias_idx[n]] +=
// sum_s slab[s][n][bias_col] (wgrad_scatter_rows_kernel, backward.hip) -- as a device function that the NEXT weight-gradient
// launch runs in its prologue ("launch-boundary reduce": the previous launch's slabs are complete at the kernel boundary, so
// no inter-workgroup hand-off is needed; what disappears is the scatter LAUNCH -- ~230 per distillation step at 12-15 us plus
// a ~10 us dependent-launch gap each on the weight-gradient stream).  Same per-element arithmetic and slab order as the kernel:
// the same bits.
#pragma once
#include "common.h"

struct WgradFold {
  const float* slabs;     // NULL: nothing to fold
  int S;
  long long slab_stride;
  int ldk, k_cols, n_rows;
  const int* row_off;
  const int* col_off;     // NULL = identity
  float* grad;
  int vec4;
  int bias_col, n_bias;
  const int* bias_idx;
  float* grad_bias;
};

// workgroup `wg` of `nwg` (256 threads each) takes rows wg, wg + nwg, ...
__device__ __forceinline__ void wgrad_fold_rows(const WgradFold& f, int wg, int nwg) {
  const int tid = threadIdx.x;
  for (int n = wg; n < f.n_rows; n += nwg) {
    if (f.bias_col >= 0 && tid < 64 && n < f.n_bias) {
      const int j = f.bias_idx ? f.bias_idx[n] : n;
      if (j >= 0) {
        float v = 0.f;
        for (int s = tid; s < f.S; s += 64) v += f.slabs[(size_t)s * f.slab_stride + (size_t)n * f.ldk + f.bias_col];
        v = wave_sum(v);
        if (tid == 0) f.grad_bias[j] = f.grad_bias[j] + v;
      }
    }
    const int ro = f.row_off[n];
    if (ro < 0) continue;
    const float* src = f.slabs + (size_t)n * f.ldk;
    float* dst = f.grad + (size_t)ro;
    if (f.vec4 && (ro & 3) == 0) {
      const int k4 = f.k_cols >> 2;
      for (int k = tid; k < k4; k += 256) {
        float4 v = reinterpret_cast<const float4*>(src)[k];
        int s = 1;
        for (; s + 8 <= f.S; s += 8) {
          float4 w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) w[u] = reinterpret_cast<const float4*>(src + (size_t)(s + u) * f.slab_stride)[k];
#pragma unroll
          for (int u = 0; u < 8; ++u) { v.x += w[u].x; v.y += w[u].y; v.z += w[u].z; v.w += w[u].w; }
        }
        for (; s < f.S; ++s) {
          const float4 w = reinterpret_cast<const float4*>(src + (size_t)s * f.slab_stride)[k];
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        float4* d = reinterpret_cast<float4*>(dst) + k;
        const float4 o = *d;
        v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w;
        *d = v;
      }
      continue;
    }
    for (int k = tid; k < f.k_cols; k += 256) {
      const int co = f.col_off ? f.col_off[k] : k;
      if (co < 0) continue;
      float v = 0.f;
      int s = 0;
      for (; s + 8 <= f.S; s += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = src[k + (size_t)(s + u) * f.slab_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += w[u];
      }
      for (; s < f.S; ++s) v += src[k + (size_t)s * f.slab_stride];
      dst[co] = dst[co] + v;
    }
  }
}
