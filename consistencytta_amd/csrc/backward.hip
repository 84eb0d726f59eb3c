// Backward-pass kernels of the consistency-distillation step (student U-Net only; the teacher,
// target and EMA networks are inference-only).  Every contraction of the backward pass runs on
// conv_gemm: data gradients with re-packed (flipped / transposed) weights, weight gradients as
//     dW[k][n] = sum_m  Q[k][m] * P[n][m],   Q = im2col(X)^T (+ indicator rows),  P = dY^T
// where both operands are made m-contiguous by the transpose kernels below (the extra rows of Q
// give the bias gradient and the per-sample time-embedding gradient from the same GEMM).
// The kernels here are the HBM-bound glue: transposes, GroupNorm / LayerNorm / GEGLU / softmax
// backward, gradient scatter into the reference parameter layout, AdamW.
// Reference semantics: torch autograd of resnet.py:549-597, attention.py:276-334,
// attention_processor.py:1068-1147; optimizer tools/train_utils.py:59-63 (torch.optim.AdamW).
#include "common.h"

#include <map>
#include <vector>
#include <mutex>
#include <utility>

#include <math.h>

static int grid1d(long long total, int per = 256, int cap = 16384) {
  long long b = (total + per - 1) / per;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------ transposes
// dst[g][c][r] = src[g][r][col0 + c] for r < rows, 0 for rows <= r < dst_ld.   64x64 tiles via LDS.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, long long sgs, int rows,
                                                        int cols, int src_ld, int col0, bf16_t* __restrict__ dst,
                                                        long long dgs, int dst_ld) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  const int g = blockIdx.z;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const bf16_t* s = src + (size_t)g * sgs;
  bf16_t* d = dst + (size_t)g * dgs;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = tid + i * 256;
    const int r = ch >> 3, cc = (ch & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < rows && c0 + cc < cols) v = *reinterpret_cast<const uint4*>(s + (size_t)(r0 + r) * src_ld + col0 + c0 + cc);
    *reinterpret_cast<uint4*>(&tile[r][cc]) = v;
  }
  __syncthreads();
  // columns out through the LDS transpose read (ds_read_b64_tr_b16: lane t of a 16-lane group names row k0 + t / 4, columns
  // 4 (t % 4) .. + 3 and receives column t % 16, rows k0 .. k0 + 3): wave w owns columns 16 w .. + 15, group kg of pass i the
  // eight rows (4 i + kg) * 8 .. + 7 -- two reads and one 16-byte store per lane, where the scalar form took eight 2-byte
  // LDS reads and their packing
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
  const int lane = tid & 63, wave = tid >> 6, t = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = (i * 4 + kg) * 8, c = wave * 16 + t;
    const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)&tile[rr + (t >> 2)][wave * 16 + (t & 3) * 4]);
    const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)&tile[rr + 4 + (t >> 2)][wave * 16 + (t & 3) * 4]);
    if (c0 + c < cols && r0 + rr < dst_ld) {
      const uint2 u0 = __builtin_bit_cast(uint2, a0), u1 = __builtin_bit_cast(uint2, a1);
      *reinterpret_cast<uint4*>(d + (size_t)(c0 + c) * dst_ld + r0 + rr) = make_uint4(u0.x, u0.y, u1.x, u1.y);
    }
  }
}

extern "C" ctta_status ctta_transpose_bf16(const void* src, int64_t src_group_stride, int rows, int cols, int src_ld,
                                           int col0, void* dst, int64_t dst_group_stride, int dst_ld, int groups,
                                           void* stream) {
  CTTA_REQUIRE(src && dst && cols % 8 == 0 && src_ld % 8 == 0 && col0 % 8 == 0 && dst_ld % 8 == 0 && dst_ld >= rows,
               "transpose: bad arguments (cols=%d src_ld=%d col0=%d dst_ld=%d rows=%d)", cols, src_ld, col0, dst_ld, rows);
  dim3 grid((dst_ld + 63) / 64, (cols + 63) / 64, groups);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     (long long)src_group_stride, rows, cols, src_ld, col0, (bf16_t*)dst, (long long)dst_group_stride,
                     dst_ld);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// many small transposes in one launch (job table on the device, one 64x64 tile per block)
__global__ __launch_bounds__(256) void transpose_multi_kernel(const ctta_tpose_job* __restrict__ jobs, int n_jobs) {
  __shared__ bf16_t tile[64][72];
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ctta_tpose_job j = jobs[lo];
  const int local = (int)blockIdx.x - j.block0;
  const int r0 = (local % j.tiles_r) * 64, c0 = (local / j.tiles_r) * 64;
  const bf16_t* s = (const bf16_t*)j.src;
  bf16_t* d = (bf16_t*)j.dst;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = tid + i * 256;
    const int r = ch >> 3, cc = (ch & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < j.rows && c0 + cc < j.cols) v = *reinterpret_cast<const uint4*>(s + (size_t)(r0 + r) * j.src_ld + c0 + cc);
    *reinterpret_cast<uint4*>(&tile[r][cc]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = tid + i * 256;
    const int c = ch >> 3, rr = (ch & 7) * 8;
    if (c0 + c < j.cols && r0 + rr < j.wcols) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[rr + 2 * e][c] | ((uint32_t)tile[rr + 2 * e + 1][c] << 16);
      *reinterpret_cast<uint4*>(d + (size_t)(c0 + c) * j.dst_ld + r0 + rr) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}
extern "C" ctta_status ctta_transpose_multi(const ctta_tpose_job* jobs, int n_jobs, int total_blocks, void* stream) {
  CTTA_REQUIRE(jobs && n_jobs >= 1 && total_blocks >= 1, "transpose_multi: bad arguments");
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// Q[(c*T + t)][m] = X[pixel(m, tap t)][c] (0 outside the image), m = (b, oh, ow), T = kh*kw taps; rows padded
// to m_pad.  Channel-major rows = the (cin, kh, kw) order of a conv weight row, so the weight-gradient slab
// scatters back with an identity column map.
struct Im2colParams {
  const bf16_t* x; int C, B, hi, wi, hs, ws, ups, ho, wo, kh, kw, sh, sw, ph, pw, dh, dw, M, m_pad;
  bf16_t* dst;
  int ind_rows;   // > 0: indicator rows behind the kh*kw*C im2col rows ([0] = 1 for m < M, [1 + b] = pixels of sample b)
};
__global__ __launch_bounds__(256) void im2col_t_kernel(Im2colParams p) {
  __shared__ bf16_t tile[64][72];
  const int t = blockIdx.z;
  const int kh = t / p.kw, kw = t - kh * p.kw;
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = tid + i * 256;
    const int r = ch >> 3, cc = (ch & 7) * 8;
    const int m = m0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < p.M && c0 + cc < p.C) {
      const int howo = p.ho * p.wo;
      const int b = m / howo;
      const int rem = m - b * howo;
      const int oh = rem / p.wo, ow = rem - oh * p.wo;
      int ih = oh * p.sh - p.ph + kh * p.dh, iw = ow * p.sw - p.pw + kw * p.dw;
      if ((unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi) {
        if (p.ups) { ih >>= 1; iw >>= 1; }
        v = *reinterpret_cast<const uint4*>(p.x + ((size_t)(b * p.hs + ih) * p.ws + iw) * p.C + c0 + cc);
      }
    }
    *reinterpret_cast<uint4*>(&tile[r][cc]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = tid + i * 256;
    const int c = ch >> 3, rr = (ch & 7) * 8;
    if (c0 + c < p.C && m0 + rr < p.m_pad) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[rr + 2 * e][c] | ((uint32_t)tile[rr + 2 * e + 1][c] << 16);
      *reinterpret_cast<uint4*>(p.dst + ((size_t)(c0 + c) * (p.kh * p.kw) + t) * p.m_pad + m0 + rr) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  // the indicator rows ride in the same launch: the (channel tile 0, tap 0) block of every 64-column slice writes them
  if (p.ind_rows > 0 && blockIdx.y == 0 && t == 0) {
    bf16_t* rows = p.dst + (size_t)p.kh * p.kw * p.C * p.m_pad;
    const int per_batch = p.ho * p.wo;
    for (int i = tid; i < p.ind_rows * 64; i += 256) {
      const int r = i >> 6, m = m0 + (i & 63);
      if (m < p.m_pad) {
        const bool one = m < p.M && (r == 0 || m / per_batch == r - 1);
        rows[(size_t)r * p.m_pad + m] = one ? (bf16_t)0x3F80 : (bf16_t)0;
      }
    }
  }
}
extern "C" ctta_status ctta_im2col_t(const void* x, int c, int batch, int hi, int wi, int upsample, int ho, int wo,
                                     int kh, int kw, int stride, int pad_h, int pad_w, int dil_w, void* dst, int m_pad,
                                     int indicator_batches, void* stream) {
  CTTA_REQUIRE(x && dst && c % 8 == 0 && m_pad % 8 == 0, "im2col_t: bad arguments");
  Im2colParams p;
  p.x = (const bf16_t*)x; p.C = c; p.B = batch; p.hi = hi; p.wi = wi; p.ups = upsample ? 1 : 0;
  p.hs = p.ups ? hi / 2 : hi; p.ws = p.ups ? wi / 2 : wi; p.ho = ho; p.wo = wo; p.kh = kh; p.kw = kw;
  p.sh = p.sw = stride; p.ph = pad_h; p.pw = pad_w; p.dh = 1; p.dw = dil_w;
  p.M = batch * ho * wo; p.m_pad = m_pad; p.dst = (bf16_t*)dst;
  CTTA_REQUIRE(m_pad >= p.M, "im2col_t: m_pad < M");
  p.ind_rows = indicator_batches >= 0 ? 1 + indicator_batches : 0;
  dim3 grid((m_pad + 63) / 64, (c + 63) / 64, kh * kw);
  hipLaunchKernelGGL(im2col_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ gradient scatter
// sum over the S split slabs of one float, in slab order, eight loads in flight: the plain loop compiles to
// `global_load; s_waitcnt vmcnt(0); v_add` per slab -- S serial memory round trips per lane (S = 64 on the level-0 linears).
__device__ __forceinline__ float sum_slabs(const float* __restrict__ p, int S, long long stride) {
  float v = 0.f;
  int s = 0;
  for (; s + 8 <= S; s += 8) {
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = p[(size_t)(s + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += w[u];
  }
  for (; s < S; ++s) v += p[(size_t)s * stride];
  return v;
}
// slabs [S][R][ldn] fp32 (R >= k_rows (+ extra rows)); weight grad:
//   grad_w[row_off[n] + col_off[k]] (+)= sum_s slab[s][k][n]     (k < k_rows, n < n_cols, offsets >= 0, aux ok)
__global__ void wgrad_scatter_kernel(const float* __restrict__ slabs, int S, long long slab_stride, int ldn, int k_rows,
                                     int n_cols, const int* __restrict__ row_off, const int* __restrict__ col_off,
                                     const int* __restrict__ row_aux, const int* __restrict__ col_aux, int aux_limit,
                                     float* __restrict__ grad, int accumulate) {
  const long long total = (long long)k_rows * n_cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % n_cols);
    const int k = (int)(i / n_cols);
    const int ro = row_off[n], co = col_off[k];
    if (ro < 0 || co < 0) continue;
    if (aux_limit > 0 && row_aux[n] + col_aux[k] >= aux_limit) continue;
    const float v = sum_slabs(slabs + (size_t)k * ldn + n, S, slab_stride);
    float* g = grad + (size_t)ro + (size_t)co;
    *g = accumulate ? *g + v : v;
  }
}
// Same contraction for the common case (no aux maps), tiled through LDS so that BOTH sides are coalesced:
// the slab is read along n, the gradient is written along k (a weight row is k-contiguous in the state dict).
// col_off == NULL means the identity map.
__global__ __launch_bounds__(256) void wgrad_scatter_tiled_kernel(const float* __restrict__ slabs, int S,
                                                                  long long slab_stride, int ldn, int k_rows, int n_cols,
                                                                  const int* __restrict__ row_off,
                                                                  const int* __restrict__ col_off,
                                                                  float* __restrict__ grad, int accumulate) {
  __shared__ float tile[64][65];
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int idx = tid + i * 256;
    const int kk = idx >> 6, nn = idx & 63;
    float v = 0.f;
    if (k0 + kk < k_rows && n0 + nn < n_cols) {
      const float* p = slabs + (size_t)(k0 + kk) * ldn + n0 + nn;
      v = sum_slabs(p, S, slab_stride);
    }
    tile[kk][nn] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int idx = tid + i * 256;
    const int nn = idx >> 6, kk = idx & 63;
    if (k0 + kk >= k_rows || n0 + nn >= n_cols) continue;
    const int ro = row_off[n0 + nn];
    const int co = col_off ? col_off[k0 + kk] : k0 + kk;
    if (ro < 0 || co < 0) continue;
    float* g = grad + (size_t)ro + (size_t)co;
    *g = accumulate ? *g + tile[kk][nn] : tile[kk][nn];
  }
}
// Row-major slabs [S][n_rows][ldk] (one slab row per weight row, k contiguous): a streaming add with a row remap,
// coalesced on both sides with no staging.  grad_w[row_off[n] + col_off[k]] (+)= sum_s slab[s][n][k].
__global__ __launch_bounds__(256) void wgrad_scatter_rows_kernel(const float* __restrict__ slabs, int S,
                                                                 long long slab_stride, int ldk, int k_cols, int n_rows,
                                                                 const int* __restrict__ row_off,
                                                                 const int* __restrict__ col_off,
                                                                 float* __restrict__ grad, int accumulate, int vec4,
                                                                 int bias_col, int n_bias, const int* __restrict__ bias_idx,
                                                                 float* __restrict__ grad_bias) {
  const int n = blockIdx.y;
  // the layer's bias gradient rides along (one launch per layer instead of two): the first wave of the row's first block
  // folds column `bias_col` of the S slabs -- lanes stride the slabs, fixed butterfly
  if (bias_col >= 0 && blockIdx.x == 0 && threadIdx.x < 64 && n < n_bias) {
    const int j = bias_idx ? bias_idx[n] : n;
    if (j >= 0) {
      float v = 0.f;
      for (int s = threadIdx.x; s < S; s += 64) v += slabs[(size_t)s * slab_stride + (size_t)n * ldk + bias_col];
      v = wave_sum(v);
      if (threadIdx.x == 0) grad_bias[j] = accumulate ? grad_bias[j] + v : v;
    }
  }
  const int ro = row_off[n];
  if (ro < 0) return;
  const float* src = slabs + (size_t)n * ldk;
  float* dst = grad + (size_t)ro;
  if (vec4 && (ro & 3) == 0) {   // identity columns, 16-byte aligned rows on both sides: four columns per lane
    const int k4 = k_cols >> 2;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < k4; k += gridDim.x * 256) {
      float4 v = reinterpret_cast<const float4*>(src)[k];
      // eight slabs requested together, added in slab order (the same sum as one at a time: the compiler's own loop was
      // `load; s_waitcnt vmcnt(0); add` -- 63 serial memory round trips per lane at the 64 splits of the level-0 linears)
      int s = 1;
      for (; s + 8 <= S; s += 8) {
        float4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = reinterpret_cast<const float4*>(src + (size_t)(s + u) * slab_stride)[k];
#pragma unroll
        for (int u = 0; u < 8; ++u) { v.x += w[u].x; v.y += w[u].y; v.z += w[u].z; v.w += w[u].w; }
      }
      for (; s < S; ++s) {
        const float4 w = reinterpret_cast<const float4*>(src + (size_t)s * slab_stride)[k];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      float4* d = reinterpret_cast<float4*>(dst) + k;
      if (accumulate) { const float4 o = *d; v.x = o.x + v.x; v.y = o.y + v.y; v.z = o.z + v.z; v.w = o.w + v.w; }
      *d = v;
    }
    return;
  }
  for (int k = blockIdx.x * 256 + threadIdx.x; k < k_cols; k += gridDim.x * 256) {
    const int co = col_off ? col_off[k] : k;
    if (co < 0) continue;
    const float v = sum_slabs(src + k, S, slab_stride);
    dst[co] = accumulate ? dst[co] + v : v;
  }
}
// vector grad (bias / per-sample rows): dst[idx[n]] (+)= sum_s slab[s][row][n]  (idx[n] < 0 skipped; idx NULL = identity)
__global__ void row_scatter_kernel(const float* __restrict__ slabs, int S, long long slab_stride, int ldn, int row,
                                   int n_cols, const int* __restrict__ idx, float* __restrict__ dst, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_cols) return;
  const int j = idx ? idx[n] : n;
  if (j < 0) return;
  const float v = sum_slabs(slabs + (size_t)row * ldn + n, S, slab_stride);
  dst[j] = accumulate ? dst[j] + v : v;
}

extern "C" ctta_status ctta_wgrad_scatter(const float* slabs, int n_slabs, int64_t slab_stride, int ldn, int k_rows,
                                          int n_cols, const int32_t* row_off, const int32_t* col_off,
                                          const int32_t* row_aux, const int32_t* col_aux, int aux_limit, float* grad,
                                          int accumulate, void* stream) {
  CTTA_REQUIRE(slabs && row_off && grad && n_slabs >= 1 && (col_off || aux_limit <= 0), "wgrad_scatter: bad arguments");
  if (aux_limit <= 0) {
    hipLaunchKernelGGL(wgrad_scatter_tiled_kernel, dim3((k_rows + 63) / 64, (n_cols + 63) / 64), dim3(256), 0,
                       (hipStream_t)stream, slabs, n_slabs, (long long)slab_stride, ldn, k_rows, n_cols, row_off, col_off,
                       grad, accumulate);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  }
  const long long total = (long long)k_rows * n_cols;
  hipLaunchKernelGGL(wgrad_scatter_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                     (long long)slab_stride, ldn, k_rows, n_cols, row_off, col_off, row_aux, col_aux, aux_limit, grad,
                     accumulate);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
// dst[j * dst_stride + idx[n]] (+)= sum_s slab[s][n][col + j]   (row-major slabs; bias column, per-sample columns)
__global__ void col_scatter_kernel(const float* __restrict__ slabs, int S, long long slab_stride, int ldk, int col,
                                   int n_rows, const int* __restrict__ idx, float* __restrict__ dst, long long dst_stride,
                                   int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_rows) return;
  const int j = blockIdx.y;
  const int t = idx ? idx[n] : n;
  if (t < 0) return;
  const float v = sum_slabs(slabs + (size_t)n * ldk + col + j, S, slab_stride);
  float* d = dst + (size_t)j * dst_stride + t;
  *d = accumulate ? *d + v : v;
}
extern "C" ctta_status ctta_col_scatter(const float* slabs, int n_slabs, int64_t slab_stride, int ldk, int col, int n_cols,
                                        int n_rows, const int32_t* idx, float* dst, int64_t dst_stride, int accumulate,
                                        void* stream) {
  CTTA_REQUIRE(slabs && dst && n_slabs >= 1 && n_cols >= 1 && n_rows >= 1, "col_scatter: bad arguments");
  hipLaunchKernelGGL(col_scatter_kernel, dim3((n_rows + 255) / 256, n_cols), dim3(256), 0, (hipStream_t)stream, slabs,
                     n_slabs, (long long)slab_stride, ldk, col, n_rows, idx, dst, (long long)dst_stride, accumulate);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_wgrad_scatter_rows(const float* slabs, int n_slabs, int64_t slab_stride, int ldk, int k_cols,
                                               int n_rows, const int32_t* row_off, const int32_t* col_off, float* grad,
                                               int accumulate, void* stream) {
  return ctta_wgrad_scatter_rows_bias(slabs, n_slabs, slab_stride, ldk, k_cols, n_rows, row_off, col_off, grad, -1, 0, nullptr,
                                      nullptr, accumulate, stream);
}
extern "C" ctta_status ctta_wgrad_scatter_rows_bias(const float* slabs, int n_slabs, int64_t slab_stride, int ldk, int k_cols,
                                                    int n_rows, const int32_t* row_off, const int32_t* col_off, float* grad,
                                                    int bias_col, int n_bias, const int32_t* bias_idx, float* grad_bias,
                                                    int accumulate, void* stream) {
  CTTA_REQUIRE(slabs && row_off && grad && n_slabs >= 1 && n_rows >= 1 && k_cols >= 1, "wgrad_scatter_rows: bad arguments");
  CTTA_REQUIRE(bias_col < 0 || (grad_bias && bias_col < ldk && n_bias <= n_rows), "wgrad_scatter_rows: bad bias arguments");
  // float4 lanes when the column map is the identity and a row starts on a 16-byte boundary on both sides (the kernel
  // looks at its own row offset: every conv / linear weight whose row length is a multiple of 4 qualifies).  The scalar
  // form moved 4 bytes per lane and ran the 1024 x 9216 layers at a quarter of the HBM rate (89 us, round-3 profile).
  const int vec4 = !col_off && (k_cols % 4) == 0 && (ldk % 4) == 0 && (slab_stride % 4) == 0 &&
                   (((uintptr_t)slabs | (uintptr_t)grad) & 15) == 0;
  int gx = ((vec4 ? k_cols / 4 : k_cols) + 255) / 256;
  if (gx > 8) gx = 8;
  hipLaunchKernelGGL(wgrad_scatter_rows_kernel, dim3(gx, n_rows), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                     (long long)slab_stride, ldk, k_cols, n_rows, row_off, col_off, grad, accumulate, vec4, bias_col, n_bias,
                     bias_idx, grad_bias);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_row_scatter(const float* slabs, int n_slabs, int64_t slab_stride, int ldn, int row,
                                        int n_cols, const int32_t* idx, float* dst, int accumulate, void* stream) {
  CTTA_REQUIRE(slabs && dst && n_slabs >= 1, "row_scatter: bad arguments");
  hipLaunchKernelGGL(row_scatter_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                     (long long)slab_stride, ldn, row, n_cols, idx, dst, accumulate);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ GroupNorm backward
// y = silu?(xhat*gamma + beta), xhat = (x - mean)*rstd.  stats: [B][G][2] = (mean, rstd).
__device__ __forceinline__ float silu_grad(float z) {
  const float s = sigmoid_f(z);
  return s * (1.0f + z * (1.0f - s));
}
// pass 1: per (b, chunk) per-channel sums  A = sum dz, Bc = sum dz*xhat     part: [B][nchunk][2][C]
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             int HW, int C, int G, int ppc, int nchunk,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int silu,
                                                             float* __restrict__ part) {
  extern __shared__ float sm[];   // [PL][C] A, [PL][C] Bc
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int VC = C / 8, cpg = C / G;
  int PL, vstride, tv, tp;
  if (VC <= 256) { PL = 256 / VC; vstride = VC; tv = tid % VC; tp = tid / VC; }
  else { PL = 1; vstride = 256; tv = tid; tp = 0; }
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  float* sA = sm;
  float* sB = sm + PL * C;
  if (tp < PL) {
    for (int v = tv; v < VC; v += vstride) {
      float a[8], bb[8], mean[8], rstd[8], gm[8], bt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        const float* st = stats + ((size_t)b * G + c / cpg) * 2;
        mean[e] = st[0]; rstd[e] = st[1]; gm[e] = gamma[c]; bt[e] = beta[c];
        a[e] = 0.f; bb[e] = 0.f;
      }
      for (int pix = p0 + tp; pix < p1; pix += PL) {
        float fx[8], fd[8];
        const size_t off = ((size_t)b * HW + pix) * C + v * 8;
        unpack8(*reinterpret_cast<const uint4*>(x + off), fx);
        unpack8(*reinterpret_cast<const uint4*>(dy + off), fd);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (fx[e] - mean[e]) * rstd[e];
          float dz = fd[e];
          if (silu) dz *= silu_grad(xh * gm[e] + bt[e]);
          a[e] += dz; bb[e] += dz * xh;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { sA[tp * C + v * 8 + e] = a[e]; sB[tp * C + v * 8 + e] = bb[e]; }
    }
  }
  __syncthreads();
  float* o = part + ((size_t)b * nchunk + chunk) * 2 * C;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, bb = 0.f;
    for (int l = 0; l < PL; ++l) { a += sA[l * C + c]; bb += sB[l * C + c]; }
    o[c] = a; o[C + c] = bb;
  }
}
// pass 2a: one block per sample folds the chunks; coef[b][g] = (S1/n, S2/n) with S1 = sum_c gamma*A,
// S2 = sum_c gamma*Bc; red[b][2][C] keeps the folded per-channel sums for pass 2b.
__global__ __launch_bounds__(1024) void gn_bwd_fold_kernel(const float* __restrict__ part, int nchunk, int G, int C, int HW,
                                                           const float* __restrict__ gamma, float* __restrict__ coef,
                                                           float* __restrict__ red) {
  const int b = blockIdx.x;
  const int cpg = C / G;
  float* rb = red + (size_t)b * 2 * C;
  // C <= 512: the 1024 threads split the chunk walk `parts` ways per channel and meet in LDS in a fixed order (one thread
  // per channel walking up to 256 chunks serially, on as many CUs as there are samples, was 35 us per call and 2 ms per
  // distillation step)
  __shared__ double sa[1024], sb[1024];
  const int parts = C <= 512 ? (int)blockDim.x / C : 1;
  if (parts > 1) {
    const int c = threadIdx.x % C, pt = threadIdx.x / C;
    double a = 0.0, bb = 0.0;
    if (pt < parts)
      for (int ch = pt; ch < nchunk; ch += parts) {
        const float* o = part + ((size_t)b * nchunk + ch) * 2 * C;
        a += (double)o[c]; bb += (double)o[C + c];
      }
    sa[threadIdx.x] = a; sb[threadIdx.x] = bb;
    __syncthreads();
    if (pt == 0) {
      for (int q = 1; q < parts; ++q) { a += sa[q * C + c]; bb += sb[q * C + c]; }
      rb[c] = (float)a;
      rb[C + c] = (float)bb;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      double a = 0.0, bb = 0.0;
      for (int ch = 0; ch < nchunk; ++ch) {
        const float* o = part + ((size_t)b * nchunk + ch) * 2 * C;
        a += (double)o[c]; bb += (double)o[C + c];
      }
      rb[c] = (float)a;
      rb[C + c] = (float)bb;
    }
  }
  __syncthreads();
  const double n = (double)HW * cpg;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    double s1 = 0.0, s2 = 0.0;
    for (int cc = 0; cc < cpg; ++cc) {
      const int c = g * cpg + cc;
      s1 += (double)gamma[c] * rb[c];
      s2 += (double)gamma[c] * rb[C + c];
    }
    coef[((size_t)b * G + g) * 2] = (float)(s1 / n);
    coef[((size_t)b * G + g) * 2 + 1] = (float)(s2 / n);
  }
}
// pass 2a on a (group slices, batch) grid: a block owns the channels of `gpb` consecutive groups (64 channels for the U-Net's
// group widths), its 1024 threads split the chunk walk 1024 / channels ways per channel and meet in LDS in a fixed order,
// then the block's first threads turn the per-channel sums into the groups' coefficients.  (One block per SAMPLE -- nine
// blocks at batch 9 -- walked up to 64 chunks per thread: 28-36 us per call, 1.7-2.2 ms of the distillation step.)
__global__ __launch_bounds__(1024) void gn_bwd_fold_slices_kernel(const float* __restrict__ part, int nchunk, int G, int C, int HW,
                                                                  const float* __restrict__ gamma, float* __restrict__ coef,
                                                                  float* __restrict__ red, int gpb) {
  __shared__ double sa[1024], sb[1024];
  __shared__ float fa[1024], fb[1024];
  const int b = blockIdx.y, g0 = blockIdx.x * gpb;
  const int cpg = C / G, ng = min(gpb, G - g0), c_lo = g0 * cpg, nc = ng * cpg;      // nc <= 1024 (host)
  const int parts = 1024 / nc;
  const int c = threadIdx.x % nc, pt = threadIdx.x / nc;
  double a = 0.0, bb = 0.0;
  if (pt < parts) {
    const float* pc = part + (size_t)b * nchunk * 2 * C + c_lo + c;
    for (int ch = pt; ch < nchunk; ch += parts) { a += (double)pc[(size_t)ch * 2 * C]; bb += (double)pc[(size_t)ch * 2 * C + C]; }
  }
  sa[threadIdx.x] = a; sb[threadIdx.x] = bb;
  __syncthreads();
  if (pt == 0) {
    for (int q = 1; q < parts; ++q) { a += sa[q * nc + c]; bb += sb[q * nc + c]; }
    float* rb = red + (size_t)b * 2 * C;
    rb[c_lo + c] = fa[c] = (float)a;
    rb[C + c_lo + c] = fb[c] = (float)bb;
  }
  __syncthreads();
  const double n = (double)HW * cpg;
  if ((int)threadIdx.x < ng) {
    const int g = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int cc = 0; cc < cpg; ++cc) {
      const double gm = (double)gamma[c_lo + g * cpg + cc];
      s1 += gm * fa[g * cpg + cc];
      s2 += gm * fb[g * cpg + cc];
    }
    coef[((size_t)b * G + g0 + g) * 2] = (float)(s1 / n);
    coef[((size_t)b * G + g0 + g) * 2 + 1] = (float)(s2 / n);
  }
}
// pass 2b: dgamma[c] (+)= sum_b Bc, dbeta[c] (+)= sum_b A   (fixed summation order: deterministic)
__global__ void gn_bwd_param_kernel(const float* __restrict__ red, int B, int C, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, bb = 0.0;
  for (int b = 0; b < B; ++b) { a += red[((size_t)b * 2) * C + c]; bb += red[((size_t)b * 2 + 1) * C + c]; }
  dbeta[c] = accumulate ? dbeta[c] + (float)a : (float)a;
  dgamma[c] = accumulate ? dgamma[c] + (float)bb : (float)bb;
}
// pass 3: dx = rstd * (dz*gamma - S1/n - xhat*S2/n)   (+ existing dx when acc_dx)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ dx, int HW, int C, int G,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int silu, int acc_dx,
                                                           long long total_vec) {
  const int VC = C / 8, cpg = C / G;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total_vec;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % VC);
    const int b = (int)((idx / VC) / HW);
    float fx[8], fd[8], fo[8];
    unpack8(*reinterpret_cast<const uint4*>(x + (size_t)idx * 8), fx);
    unpack8(*reinterpret_cast<const uint4*>(dy + (size_t)idx * 8), fd);
    if (acc_dx) unpack8(*reinterpret_cast<const uint4*>(dx + (size_t)idx * 8), fo);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = v * 8 + e;
      const int g = c / cpg;
      const float* st = stats + ((size_t)b * G + g) * 2;
      const float* cf = coef + ((size_t)b * G + g) * 2;
      const float xh = (fx[e] - st[0]) * st[1];
      float dz = fd[e];
      if (silu) dz *= silu_grad(xh * gamma[c] + beta[c]);
      const float r = st[1] * (dz * gamma[c] - cf[0] - xh * cf[1]);
      fo[e] = acc_dx ? fo[e] + r : r;
    }
    *reinterpret_cast<uint4*>(dx + (size_t)idx * 8) = pack8(fo);
  }
}

// The same pass with a block of (256 / VC) * VC threads, VC = C / 8 <= 256: grid (blocks per sample, batch); a thread keeps ONE
// 8-channel vector column for the whole launch, so (mean, rstd, S1/n, S2/n) of its channels' groups, gamma and beta sit in
// registers -- the kernel above re-loads six scalars per ELEMENT and divides per element -- and four vectors are in
// flight per thread (round 3: 36-44 us per call at batch 9 against an HBM time of 11).
__global__ __launch_bounds__(256) void gn_bwd_apply_cols_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                bf16_t* __restrict__ dx, int HW, int C, int G,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ coef,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int silu, int acc_dx) {
  const int VC = C / 8, cpg = C / G, b = blockIdx.y;
  const long long total = (long long)HW * VC;
  const long long stride = (long long)gridDim.x * blockDim.x;     // blockDim.x is a multiple of VC
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int v = (int)(idx % VC);
  float mu[8], rs[8], c0[8], c1[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = v * 8 + e, g = c / cpg;
    mu[e] = stats[((size_t)b * G + g) * 2]; rs[e] = stats[((size_t)b * G + g) * 2 + 1];
    c0[e] = coef[((size_t)b * G + g) * 2]; c1[e] = coef[((size_t)b * G + g) * 2 + 1];
    gm[e] = gamma[c]; bt[e] = beta[c];
  }
  const uint4* xb = reinterpret_cast<const uint4*>(x + (size_t)b * HW * C);
  const uint4* db = reinterpret_cast<const uint4*>(dy + (size_t)b * HW * C);
  uint4* ob = reinterpret_cast<uint4*>(dx + (size_t)b * HW * C);
  auto one = [&](const uint4 rx, const uint4 rd, const uint4 ro) {
    float fx[8], fd[8], fo[8];
    unpack8(rx, fx); unpack8(rd, fd); unpack8(ro, fo);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (fx[e] - mu[e]) * rs[e];
      float dz = fd[e];
      if (silu) dz *= silu_grad(xh * gm[e] + bt[e]);
      const float r = rs[e] * (dz * gm[e] - c0[e] - xh * c1[e]);
      fo[e] = acc_dx ? fo[e] + r : r;
    }
    return pack8(fo);
  };
  for (; idx + 3 * stride < total; idx += 4 * stride) {
    uint4 rx[4], rd[4], ro[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { rx[u] = xb[idx + u * stride]; rd[u] = db[idx + u * stride]; ro[u] = acc_dx ? ob[idx + u * stride] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < 4; ++u) ob[idx + u * stride] = one(rx[u], rd[u], ro[u]);
  }
  for (; idx < total; idx += stride) ob[idx] = one(xb[idx], db[idx], acc_dx ? ob[idx] : make_uint4(0, 0, 0, 0));
}

// The whole backward of one (sample, group) slab in ONE launch when the slab is small (HW * C / G <= 16384 elements: levels 1-3
// of the batch-9 U-Net, 45 of its 61 GroupNorms): a block keeps x and dy of its slab in registers (<= 8 + 8 vectors per
// thread), reduces the per-channel sums (A = sum dz, Bc = sum dz * xhat) through LDS in a fixed order, turns them into the
// group's two coefficients and writes dx -- x and dy are read once instead of twice, and the partial / fold / apply
// launches (5-20 us each plus their dependent-launch gaps on the step's main stream) become one.  red[b][2][C] receives
// the per-channel sums for gn_bwd_param_kernel exactly like the fold kernels leave them.
template <bool SILU, bool ACC>
__global__ __launch_bounds__(256) void gn_bwd_small_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ dx, int HW, int C, int G,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ red) {
  constexpr int MAXV = 8;
  __shared__ float sA[256][9], sB[256][9];          // [thread][8 channels], padded
  __shared__ float chA[128], chB[128], coef[2];     // per-channel sums of the group (cpg <= 128), the two coefficients
  __shared__ double wsum[4];                        // per-wave fp64 sums of gamma * A, gamma * Bc (waves 0, 1)
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G, vpr = cpg / 8;              // vectors per pixel row of this group; vpr divides 256 (host)
  const int nvec = HW * vpr;
  const int cv = tid % vpr;                          // this thread's channel vector inside the group: fixed (256 % vpr == 0)
  const size_t base = (size_t)b * HW * C + (size_t)g * cpg;
  const float mean = stats[((size_t)b * G + g) * 2], rstd = stats[((size_t)b * G + g) * 2 + 1];
  float gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gm[e] = gamma[g * cpg + cv * 8 + e]; bt[e] = beta[g * cpg + cv * 8 + e]; }
  // ALL of the thread's vectors are requested before the first one is used (round 4 loaded, waited and consumed one pair
  // per iteration: eight serial memory round trips per thread, and eight more for the old dx in the accumulate form --
  // 28-32 us per launch for slabs that move in 3-6 us; SILU / ACC are template flags: no per-element branch); the raw bf16
  // vectors stay in registers (64 VGPRs) and xhat / dz are recomputed in the apply phase instead of being kept as 128 floats
  uint4 xr[MAXV], dr[MAXV], orr[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = tid + i * 256;
    const size_t off = base + (size_t)(idx < nvec ? idx / vpr : 0) * C + cv * 8;
    xr[i] = make_uint4(0, 0, 0, 0); dr[i] = make_uint4(0, 0, 0, 0); orr[i] = make_uint4(0, 0, 0, 0);
    if (idx < nvec) {
      xr[i] = *reinterpret_cast<const uint4*>(x + off);
      dr[i] = *reinterpret_cast<const uint4*>(dy + off);
      if (ACC) orr[i] = *reinterpret_cast<const uint4*>(dx + off);
    }
  }
  float a[8], bb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = 0.f; bb[e] = 0.f; }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = tid + i * 256;
    if (idx < nvec) {
      float fx[8], fd[8];
      unpack8(xr[i], fx);
      unpack8(dr[i], fd);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float h = (fx[e] - mean) * rstd;
        float z = fd[e];
        if (SILU) z *= silu_grad(h * gm[e] + bt[e]);
        a[e] += z; bb[e] += z * h;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { sA[tid][e] = a[e]; sB[tid][e] = bb[e]; }
  __syncthreads();
  if (tid < cpg) {                                   // channel tid of the group: threads cv, cv + vpr, ... hold its sums
    const int v = tid / 8, e = tid % 8;
    float ta = 0.f, tb = 0.f;
    for (int t = v; t < 256; t += vpr) { ta += sA[t][e]; tb += sB[t][e]; }
    chA[tid] = ta; chB[tid] = tb;
    red[(size_t)b * 2 * C + g * cpg + tid] = ta;
    red[(size_t)b * 2 * C + C + g * cpg + tid] = tb;
  }
  __syncthreads();
  // the group's two coefficients: sum over its <= 128 channels of gamma * A / gamma * Bc in fp64, as a fixed tree (one term per
  // thread, xor butterfly inside each wave, the two waves' sums added in wave order) -- round 4 had thread 0 walk the channels
  // alone: up to 128 dependent global loads of gamma, ~12 us of a kernel whose data moves in ~1 us (31.6 us per launch, 41
  // launches per distillation step on the backward's main stream)
  {
    double s1 = 0.0, s2 = 0.0;
    if (tid < cpg) { const double gmc = (double)gamma[g * cpg + tid]; s1 = gmc * (double)chA[tid]; s2 = gmc * (double)chB[tid]; }
    if (tid < 128) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
      if ((tid & 63) == 0) { wsum[tid >> 6] = s1; wsum[2 + (tid >> 6)] = s2; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    const double t1 = wsum[0] + wsum[1];
    const double t2 = wsum[2] + wsum[3];
    const double n = (double)HW * cpg;
    coef[0] = (float)(t1 / n); coef[1] = (float)(t2 / n);
  }
  __syncthreads();
  const float c0 = coef[0], c1 = coef[1];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = tid + i * 256;
    if (idx < nvec) {
      bf16_t* o = dx + base + (size_t)(idx / vpr) * C + cv * 8;
      float fx[8], fd[8], fo[8];
      unpack8(xr[i], fx);
      unpack8(dr[i], fd);
      if (ACC) unpack8(orr[i], fo);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float h = (fx[e] - mean) * rstd;         // the same expressions as in the first phase: the same bits
        float z = fd[e];
        if (SILU) z *= silu_grad(h * gm[e] + bt[e]);
        const float r = rstd * (z * gm[e] - c0 - h * c1);
        fo[e] = ACC ? fo[e] + r : r;
      }
      *reinterpret_cast<uint4*>(o) = pack8(fo);
    }
  }
}
static bool gn_bwd_small_ok(int hw, int c, int groups) {
  if (c % groups) return false;
  const int cpg = c / groups;
  if (cpg % 8 || cpg > 128) return false;
  const int vpr = cpg / 8;
  return 256 % vpr == 0 && (long long)hw * vpr <= 256 * 8;
}

static void gnb_geometry(int hw, int c, int* ppc, int* nchunk) {
  int p = 16384 / c;
  if (p < 16) p = 16;
  if (p > 1024) p = 1024;
  if (p > hw) p = hw;
  *ppc = p;
  *nchunk = (hw + p - 1) / p;
}
extern "C" size_t ctta_groupnorm_bwd_scratch_floats(int batch, int hw, int c, int groups) {
  int ppc, nchunk;
  gnb_geometry(hw, c, &ppc, &nchunk);
  return (size_t)batch * nchunk * 2 * c + (size_t)batch * 2 * c + (size_t)batch * groups * 2 + 64;
}
extern "C" ctta_status ctta_groupnorm_bwd(const void* x, const void* dy, void* dx, int batch, int hw, int c, int groups,
                                          const float* stats, const float* gamma, const float* beta, int silu,
                                          int accumulate_dx, float* dgamma, float* dbeta, int accumulate_param,
                                          float* scratch, void* stream) {
  CTTA_REQUIRE(x && dy && dx && stats && gamma && beta && scratch, "groupnorm_bwd: null pointer");
  CTTA_REQUIRE(c % 8 == 0 && c % groups == 0, "groupnorm_bwd: C=%d groups=%d", c, groups);
  hipStream_t s = (hipStream_t)stream;
  int ppc, nchunk;
  gnb_geometry(hw, c, &ppc, &nchunk);
  float* part = scratch;
  float* red = part + (size_t)batch * nchunk * 2 * c;
  float* coef = red + (size_t)batch * 2 * c;
  if (gn_bwd_small_ok(hw, c, groups)) {
#define GNBS(SI, AC) hipLaunchKernelGGL((gn_bwd_small_kernel<SI, AC>), dim3(groups, batch), dim3(256), 0, s, (const bf16_t*)x, \
                                        (const bf16_t*)dy, (bf16_t*)dx, hw, c, groups, stats, gamma, beta, red)
    if (silu) { if (accumulate_dx) GNBS(true, true); else GNBS(true, false); }
    else { if (accumulate_dx) GNBS(false, true); else GNBS(false, false); }
#undef GNBS
    CTTA_LAUNCH_CHECK();
    if (dgamma) {
      CTTA_REQUIRE(dbeta, "groupnorm_bwd: dgamma without dbeta");
      hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, s, red, batch, c, dgamma, dbeta,
                         accumulate_param);
      CTTA_LAUNCH_CHECK();
    }
    return CTTA_OK;
  }
  const int VC = c / 8, PL = VC <= 256 ? 256 / VC : 1;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, batch), dim3(256), (size_t)2 * PL * c * sizeof(float), s,
                     (const bf16_t*)x, (const bf16_t*)dy, hw, c, groups, ppc, nchunk, stats, gamma, beta, silu, part);
  CTTA_LAUNCH_CHECK();
  {
    const int cpg = c / groups;
    const int gpb = cpg >= 64 ? 1 : 64 / cpg;
    if (cpg * gpb <= 1024)
      hipLaunchKernelGGL(gn_bwd_fold_slices_kernel, dim3((groups + gpb - 1) / gpb, batch), dim3(1024), 0, s, part, nchunk, groups, c,
                         hw, gamma, coef, red, gpb);
    else
      hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(batch), dim3(1024), 0, s, part, nchunk, groups, c, hw, gamma, coef, red);
  }
  CTTA_LAUNCH_CHECK();
  if (dgamma) {
    CTTA_REQUIRE(dbeta, "groupnorm_bwd: dgamma without dbeta");
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, s, red, batch, c, dgamma, dbeta,
                       accumulate_param);
    CTTA_LAUNCH_CHECK();
  }
  const long long total_vec = (long long)batch * hw * VC;
  if (VC <= 256) {
    const int bd = 256 / VC * VC;
    const long long per_sample = (long long)hw * VC;
    long long bps = (per_sample + bd * 8 - 1) / (bd * 8);               // >= 8 vectors per thread
    const long long cap = (8192 + batch - 1) / batch;
    if (bps > cap) bps = cap;
    if (bps < 1) bps = 1;
    hipLaunchKernelGGL(gn_bwd_apply_cols_kernel, dim3((unsigned)bps, (unsigned)batch), dim3(bd), 0, s, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, hw, c, groups, stats, coef, gamma, beta, silu, accumulate_dx);
  } else {
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid1d(total_vec, 256, 8192)), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, hw, c, groups, stats, coef, gamma, beta, silu, accumulate_dx,
                       total_vec);
  }
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// GroupNorm statistics (mean, rstd) per (batch, group) for the training forward: reuses the forward
// partial sums layout; one small kernel that recomputes them from x (saved-tensor friendly).
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, int HW, int C, int G, float eps,
                                                       float* __restrict__ stats) {
  __shared__ double red[2][256];
  const int b = blockIdx.y, g = blockIdx.x, cpg = C / G;
  double s = 0.0, q = 0.0;
  const long long n = (long long)HW * cpg;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const int pix = (int)(i / cpg), cc = (int)(i - (long long)pix * cpg);
    const float v = bf2f(x[((size_t)b * HW + pix) * C + g * cpg + cc]);
    s += v; q += (double)v * v;
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = red[0][0] / (double)n;
    double var = red[1][0] / (double)n - mean * mean;
    if (var < 0) var = 0;
    stats[((size_t)b * G + g) * 2] = (float)mean;
    stats[((size_t)b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
extern "C" ctta_status ctta_groupnorm_stats(const void* x, int batch, int hw, int c, int groups, float eps, float* stats,
                                            void* stream) {
  CTTA_REQUIRE(x && stats && c % groups == 0, "groupnorm_stats: bad arguments");
  hipLaunchKernelGGL(gn_stats_kernel, dim3(groups, batch), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, hw, c,
                     groups, eps, stats);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ LayerNorm backward
// One wave per row (HALF: two rows per wave, 32 lanes each, for rows of <= 256 channels -- otherwise half the wave idles);
// the next row's x / dy are requested before the current row's five reductions run (a row is otherwise a chain of two
// exposed load latencies and five shuffle reductions: 0.65 TB/s measured on the 36 864 x 256 matrices of the distillation
// step); dgamma/dbeta via LDS + one atomicAdd per block/column.
template <bool HALF>
__device__ __forceinline__ float ln_row_sum(float v) {
  if (!HALF) v += __shfl_xor(v, 32, 64);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int MAXV, bool HALF>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                     bf16_t* dx, long long rows, int d, int ld,
                                                     const float* __restrict__ gamma, float eps, const bf16_t* dx_add,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int rows_per_block, float* __restrict__ part) {
  extern __shared__ float sm[];   // [row groups][2][ld]: one (d gamma, d beta) copy per row group of the block
  const int wave = threadIdx.x >> 6;
  const int lane = HALF ? (threadIdx.x & 31) : (threadIdx.x & 63);      // lane within the row's group
  constexpr int LW = HALF ? 32 : 64;                                      // lanes per row
  const int VC = ld / 8;
  float gacc[MAXV][8], bacc[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { gacc[i][e] = 0.f; bacc[i][e] = 0.f; }
  float gam[MAXV][8];             // this lane's columns of gamma (they do not change from row to row)
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (lane + i * LW) * 8 + e;
      gam[i][e] = c < d ? gamma[c] : 0.f;
    }
  const long long r_begin = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r_begin + rows_per_block < rows ? r_begin + rows_per_block : rows;
  const int step = HALF ? 8 : 4;
  long long row = r_begin + (HALF ? wave * 2 + ((threadIdx.x >> 5) & 1) : wave);
  uint4 nx[MAXV], nd[MAXV];
  auto request = [&](long long r) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + i * LW;
      if (v < VC) {
        nx[i] = *reinterpret_cast<const uint4*>(x + (size_t)r * ld + v * 8);
        nd[i] = *reinterpret_cast<const uint4*>(dy + (size_t)r * ld + v * 8);
      }
    }
  };
  if (row < r_end) request(row);
  for (; row < r_end; row += step) {
    float fx[MAXV][8], fd[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + i * LW;
      if (v < VC) { unpack8(nx[i], fx[i]); unpack8(nd[i], fd[i]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!(v < VC && v * 8 + e < d)) { fx[i][e] = 0.f; fd[i][e] = 0.f; }
        s += fx[i][e];
      }
    }
    if (row + step < r_end) request(row + step);
    const float mean = ln_row_sum<HALF>(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = (lane + i * LW) * 8 + e;
        if (c < d) { const float t = fx[i][e] - mean; q += t * t; }
      }
    const float rstd = rsqrtf(ln_row_sum<HALF>(q) / (float)d + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = (lane + i * LW) * 8 + e;
        if (c < d) {
          const float xh = (fx[i][e] - mean) * rstd;
          const float dg = fd[i][e] * gam[i][e];
          s1 += dg; s2 += dg * xh;
          gacc[i][e] += fd[i][e] * xh; bacc[i][e] += fd[i][e];
          fx[i][e] = xh; fd[i][e] = dg;
        }
      }
    s1 = ln_row_sum<HALF>(s1) / (float)d;
    s2 = ln_row_sum<HALF>(s2) / (float)d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + i * LW;
      if (v < VC) {
        float o[8];
        if (dx_add) unpack8(*reinterpret_cast<const uint4*>(dx_add + (size_t)row * ld + v * 8), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = v * 8 + e;
          const float r = c < d ? rstd * (fd[i][e] - s1 - fx[i][e] * s2) : 0.f;
          o[e] = dx_add ? o[e] + r : r;
        }
        *reinterpret_cast<uint4*>(dx + (size_t)row * ld + v * 8) = pack8(o);
      }
    }
  }
  // every row group parks its column sums in its own LDS copy (plain stores: a lane owns its columns), then the block
  // folds the NG copies per column in a fixed order.  (The copies replace LDS float atomics on one shared copy, which cost
  // ~25 us of a 58 us launch at 36864 x 256: tools/ln_bwd_bench.py with the epilogue ablated.)
  constexpr int NG = HALF ? 8 : 4;
  {
    float* mine = sm + (size_t)(HALF ? threadIdx.x >> 5 : wave) * 2 * ld;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + i * LW;
      if (v < VC) {
        *reinterpret_cast<float4*>(mine + v * 8) = make_float4(gacc[i][0], gacc[i][1], gacc[i][2], gacc[i][3]);
        *reinterpret_cast<float4*>(mine + v * 8 + 4) = make_float4(gacc[i][4], gacc[i][5], gacc[i][6], gacc[i][7]);
        *reinterpret_cast<float4*>(mine + ld + v * 8) = make_float4(bacc[i][0], bacc[i][1], bacc[i][2], bacc[i][3]);
        *reinterpret_cast<float4*>(mine + ld + v * 8 + 4) = make_float4(bacc[i][4], bacc[i][5], bacc[i][6], bacc[i][7]);
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * ld; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) t += sm[(size_t)g * 2 * ld + c];
    sm[c] = t;                    // copy 0 now holds the block's sums (this thread is the only reader of column c)
  }
  __syncthreads();
  if (part) {   // this block's (d gamma, d beta) row of the partial table; ln_bwd_reduce_kernel folds the table in a fixed order
    float* o = part + (size_t)blockIdx.x * 2 * ld;
    for (int c = threadIdx.x; c < 2 * ld; c += 256) o[c] = sm[c];
    return;
  }
  for (int c = threadIdx.x; c < d; c += 256) {
    atomicAdd(&dgamma[c], sm[c]);
    atomicAdd(&dbeta[c], sm[ld + c]);
  }
}
// dgamma[c] += sum_b part[b][0][c], dbeta[c] += sum_b part[b][1][c].  grid (column blocks of 64, row slices): a block folds
// its slice of the table with 4 row lanes per column (8 loads in flight each), then adds ONE value per column and slice.
// (Round 3: the kernel above used to add every block's 2 d partials straight into dgamma / dbeta -- 1536 blocks x 2 x 320
// global atomics on 640 addresses per call, which the L2 serialises per address: 65-70 us per launch on a tensor whose
// HBM time is 14, tools/ln_bwd_bench.py.  Now <= 32 atomics per address.)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int ld, int d,
                                                            int rows_per_slice, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  __shared__ float red[2][4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r_lo = blockIdx.y * rows_per_slice, r_hi = min(nblocks, r_lo + rows_per_slice);
  float g = 0.f, b = 0.f;
  if (col < d) {
    constexpr int U = 8;
    for (int r = r_lo + rl; r < r_hi; r += 4 * U) {
      float pg[U], pb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = r + 4 * u;
        pg[u] = rr < r_hi ? part[(size_t)rr * 2 * ld + col] : 0.f;
        pb[u] = rr < r_hi ? part[(size_t)rr * 2 * ld + ld + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { g += pg[u]; b += pb[u]; }
    }
  }
  red[0][rl][threadIdx.x & 63] = g;
  red[1][rl][threadIdx.x & 63] = b;
  __syncthreads();
  if (rl == 0 && col < d) {
    const int t = threadIdx.x & 63;
    atomicAdd(&dgamma[col], (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]));
    atomicAdd(&dbeta[col], (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]));
  }
}
// Launch geometry of ctta_layernorm_bwd*: rows per block.  64 amortises the per-block dgamma / dbeta atomics on long
// matrices, but the distillation step's token matrices are short (9 216 .. 36 864 rows): aim for >= ~2000 blocks so that
// every CU holds several.
static int ln_bwd_rows_per_block(int64_t rows) {
  int rpb = (int)(rows / 2048);
  return rpb < 8 ? 8 : (rpb > 64 ? 64 : (rpb + 7) / 8 * 8);
}
static constexpr bool ln_bwd_two_pass() { return true; }
// The per-block partial table of d gamma / d beta belongs to the CALLER (an engine handle's arena, a torch tensor): the
// library keeps no table of its own (round 4 kept a process-global one keyed by (device, stream) that could only ever
// grow -- SURVEY 8b: no global state, one handle per (device, model)).  0 = this shape takes the atomics-only path.
extern "C" size_t ctta_layernorm_bwd_scratch_floats(int64_t rows, int ld) {
  if (rows <= 0 || ld <= 0 || !ln_bwd_two_pass()) return 0;
  const int64_t blocks = cdiv64(rows, ln_bwd_rows_per_block(rows));
  const size_t floats = (size_t)blocks * 2 * (size_t)ld;
  return (blocks >= 16 && floats * sizeof(float) <= ((size_t)64 << 20)) ? floats : 0;
}
extern "C" ctta_status ctta_layernorm_bwd(const void* x, const void* dy, void* dx, int64_t rows, int d, int ld,
                                          const float* gamma, float eps, int accumulate_dx, float* dgamma, float* dbeta,
                                          void* stream) {
  return ctta_layernorm_bwd_ws(x, dy, accumulate_dx ? dx : nullptr, dx, rows, d, ld, gamma, eps, dgamma, dbeta, nullptr, 0, stream);
}
// dx = dx_add + dL/dx  (dx_add NULL: plain; dx_add == dx: the in-place accumulate of ctta_layernorm_bwd).  A separate
// output lets the backward keep the PREVIOUS running gradient untouched for a weight-gradient job that reads it later.
extern "C" ctta_status ctta_layernorm_bwd_add(const void* x, const void* dy, const void* dx_add, void* dx, int64_t rows, int d,
                                              int ld, const float* gamma, float eps, float* dgamma, float* dbeta, void* stream) {
  return ctta_layernorm_bwd_ws(x, dy, dx_add, dx, rows, d, ld, gamma, eps, dgamma, dbeta, nullptr, 0, stream);
}
// `scratch` (>= ctta_layernorm_bwd_scratch_floats(rows, ld) floats, caller-owned, free again once the call's kernels have
// run on `stream`): per-block partial sums of d gamma / d beta + a sliced fold (<= 32 atomics per address).  NULL / too
// small: one atomic per block and column straight into dgamma / dbeta (same result up to the order of the fp32 atomics).
extern "C" ctta_status ctta_layernorm_bwd_ws(const void* x, const void* dy, const void* dx_add, void* dx, int64_t rows, int d,
                                             int ld, const float* gamma, float eps, float* dgamma, float* dbeta,
                                             float* scratch, size_t scratch_floats, void* stream) {
  CTTA_REQUIRE(x && dy && dx && gamma && dgamma && dbeta, "layernorm_bwd: null pointer (dgamma/dbeta must be zeroed or hold the running sum)");
  CTTA_REQUIRE(ld % 8 == 0 && d <= ld && ld <= 2048, "layernorm_bwd: d=%d ld=%d", d, ld);
  const int rpb = ln_bwd_rows_per_block(rows);
  const dim3 grid((unsigned)cdiv64(rows, rpb));
  const size_t smem = (size_t)(ld <= 256 ? 8 : 4) * 2 * ld * sizeof(float);      // one copy per row group (<= 64 KB at ld = 2048)
  hipStream_t s = (hipStream_t)stream;
  const size_t need = ctta_layernorm_bwd_scratch_floats(rows, ld);
  float* part = (scratch && need && scratch_floats >= need) ? scratch : nullptr;
#define LNB(MV, HF) hipLaunchKernelGGL((ln_bwd_kernel<MV, HF>), grid, dim3(256), smem, s, (const bf16_t*)x, (const bf16_t*)dy, \
                                       (bf16_t*)dx, (long long)rows, d, ld, gamma, eps, (const bf16_t*)dx_add, dgamma, dbeta, rpb, part)
  if (ld <= 256) LNB(1, true); else if (ld <= 512) LNB(1, false); else if (ld <= 1024) LNB(2, false); else LNB(4, false);
#undef LNB
  CTTA_LAUNCH_CHECK();
  if (part) {
    const int slices = (int)grid.x >= 32 * 8 ? 32 : ((int)grid.x + 7) / 8;          // >= 8 table rows per slice
    const int rps = ((int)grid.x + slices - 1) / slices;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((d + 63) / 64, ((int)grid.x + rps - 1) / rps), dim3(256), 0, s, part,
                       (int)grid.x, ld, d, rps, dgamma, dbeta);
    CTTA_LAUNCH_CHECK();
  }
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ GEGLU / SiLU / adds
// f = [value | gate] (rows x 2hp), dout (rows x hp) -> df (rows x 2hp)
__global__ void geglu_bwd_kernel(const bf16_t* __restrict__ f, const bf16_t* __restrict__ dout, bf16_t* __restrict__ df,
                                 long long rows, int hp, int interleaved) {
  const int vc = hp / 8;
  const long long total = rows * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    const long long r = idx / vc;
    const int va = interleaved ? (v >> 1) * 32 + (v & 1) * 8 : v * 8;   // layouts: see geglu_kernel
    const int vg = interleaved ? va + 16 : hp + v * 8;
    float a[8], g[8], dd[8], da[8], dg[8];
    unpack8(*reinterpret_cast<const uint4*>(f + (size_t)r * 2 * hp + va), a);
    unpack8(*reinterpret_cast<const uint4*>(f + (size_t)r * 2 * hp + vg), g);
    unpack8(*reinterpret_cast<const uint4*>(dout + (size_t)r * hp + v * 8), dd);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float cdf, pdf;
      gelu_cdf_pdf(g[e], cdf, pdf);
      da[e] = dd[e] * g[e] * cdf;
      dg[e] = dd[e] * a[e] * (cdf + g[e] * pdf);
    }
    *reinterpret_cast<uint4*>(df + (size_t)r * 2 * hp + va) = pack8(da);
    *reinterpret_cast<uint4*>(df + (size_t)r * 2 * hp + vg) = pack8(dg);
  }
}
extern "C" ctta_status ctta_geglu_bwd(const void* f, const void* dout, void* df, int64_t rows, int hp, int interleaved,
                                      void* stream) {
  CTTA_REQUIRE(f && dout && df && hp % 8 == 0 && (!interleaved || hp % 16 == 0), "geglu_bwd: bad arguments");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid1d(rows * (hp / 8), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)f, (const bf16_t*)dout, (bf16_t*)df, (long long)rows, hp, interleaved);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// out = a + b (bf16, 8-wide); b may be a column slice (ldb, colb) of a wider matrix, a/out likewise
__global__ void add_slices_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b, int ldb,
                                  bf16_t* __restrict__ out, int ldo, long long rows, int cols) {
  const int vc = cols / 8;
  const long long total = rows * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    const long long r = idx / vc;
    float fa[8], fb[8];
    unpack8(*reinterpret_cast<const uint4*>(a + (size_t)r * lda + v * 8), fa);
    if (b) {
      unpack8(*reinterpret_cast<const uint4*>(b + (size_t)r * ldb + v * 8), fb);
#pragma unroll
      for (int e = 0; e < 8; ++e) fa[e] += fb[e];
    }
    *reinterpret_cast<uint4*>(out + (size_t)r * ldo + v * 8) = pack8(fa);
  }
}
extern "C" ctta_status ctta_add_slices(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows,
                                       int cols, void* stream) {
  CTTA_REQUIRE(a && out && cols % 8 == 0 && lda % 8 == 0 && ldo % 8 == 0 && (!b || ldb % 8 == 0), "add_slices: bad arguments");
  hipLaunchKernelGGL(add_slices_kernel, dim3(grid1d(rows * (cols / 8), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, (long long)rows, cols);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// stride-2 conv data gradient helper: dyz[b][2oh][2ow][c] = dy[b][oh][ow][c], zeros elsewhere (extent hz x wz)
__global__ void zero_insert_kernel(const uint4* __restrict__ dy, uint4* __restrict__ dz, int B, int ho, int wo, int hz,
                                   int wz, int vc) {
  const long long total = (long long)B * hz * wz * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    long long p = idx / vc;
    const int w = (int)(p % wz); p /= wz;
    const int h = (int)(p % hz);
    const int b = (int)(p / hz);
    uint4 val = make_uint4(0, 0, 0, 0);
    if (!(h & 1) && !(w & 1) && (h >> 1) < ho && (w >> 1) < wo)
      val = dy[(((size_t)b * ho + (h >> 1)) * wo + (w >> 1)) * vc + v];
    dz[idx] = val;
  }
}
extern "C" ctta_status ctta_zero_insert2(const void* dy, void* dz, int batch, int ho, int wo, int hz, int wz, int c,
                                         void* stream) {
  CTTA_REQUIRE(dy && dz && c % 8 == 0, "zero_insert2: bad arguments");
  const long long total = (long long)batch * hz * wz * (c / 8);
  hipLaunchKernelGGL(zero_insert_kernel, dim3(grid1d(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)dy, (uint4*)dz, batch, ho, wo, hz, wz, c / 8);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
// nearest x2 upsample backward: dx[b][h][w] (+)= sum of the 2x2 block of dup
__global__ void pool2_sum_kernel(const bf16_t* __restrict__ du, bf16_t* __restrict__ dx, int B, int h, int w, int C,
                                 int acc) {
  const int vc = C / 8;
  const long long total = (long long)B * h * w * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    long long p = idx / vc;
    const int x = (int)(p % w); p /= w;
    const int y = (int)(p % h);
    const int b = (int)(p / h);
    float s[8];
    if (acc) unpack8(*reinterpret_cast<const uint4*>(dx + (size_t)idx * 8), s);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = 0.f;
    }
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(du + ((((size_t)b * 2 * h + 2 * y + dyy) * 2 * w) + 2 * x + dxx) * C + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    *reinterpret_cast<uint4*>(dx + (size_t)idx * 8) = pack8(s);
  }
}
extern "C" ctta_status ctta_pool2_sum(const void* dup, void* dx, int batch, int h, int w, int c, int accumulate,
                                      void* stream) {
  CTTA_REQUIRE(dup && dx && c % 8 == 0, "pool2_sum: bad arguments");
  const long long total = (long long)batch * h * w * (c / 8);
  hipLaunchKernelGGL(pool2_sum_kernel, dim3(grid1d(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dup, (bf16_t*)dx, batch, h, w, c, accumulate);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ softmax (training) + backward
// scores fp32 [rows][cols]; optional additive per-(batch,col) bias (rows_per_batch rows share a bias row)
__global__ __launch_bounds__(256) void softmax_bias_kernel(const float* __restrict__ s, int lds,
                                                           const float* __restrict__ bias, int rows_per_bias,
                                                           bf16_t* __restrict__ p, int cols, int ldp, float scale) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* sr = s + (size_t)row * lds;
  const float* br = bias ? bias + (size_t)(row / rows_per_bias) * cols : nullptr;
  bf16_t* pr = p + (size_t)row * ldp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, sr[c] * scale + (br ? br[c] : 0.f));
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) sum += __expf(sr[c] * scale + (br ? br[c] : 0.f) - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid; c < ldp; c += 256)
    pr[c] = c < cols ? f2bf(__expf(sr[c] * scale + (br ? br[c] : 0.f) - mx) * inv) : (bf16_t)0;
}
extern "C" ctta_status ctta_softmax_bias_rows(const float* s, int lds, const float* bias, int rows_per_bias, void* p,
                                              int64_t rows, int cols, int ldp, float scale, void* stream) {
  CTTA_REQUIRE(s && p && cols >= 1 && ldp >= cols && lds >= cols && rows_per_bias >= 1, "softmax_bias_rows: bad arguments");
  hipLaunchKernelGGL(softmax_bias_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, lds, bias,
                     rows_per_bias, (bf16_t*)p, cols, ldp, scale);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
// dS = P * (dP - sum_j P_j dP_j) * scale      P bf16 [rows][ldp], dP fp32 [rows][cols] -> dS bf16 [rows][ldp]
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ p, const float* __restrict__ dp,
                                                          int lddp, bf16_t* __restrict__ ds, int cols, int ldp,
                                                          float scale) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const bf16_t* pr = p + (size_t)row * ldp;
  const float* dr = dp + (size_t)row * lddp;
  bf16_t* o = ds + (size_t)row * ldp;
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int c = tid; c < cols; c += 256) acc += bf2f(pr[c]) * dr[c];
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  const float dot = red[0] + red[1] + red[2] + red[3];
  for (int c = tid; c < ldp; c += 256) o[c] = c < cols ? f2bf(bf2f(pr[c]) * (dr[c] - dot) * scale) : (bf16_t)0;
}
extern "C" ctta_status ctta_softmax_bwd_rows(const void* p, const float* dp, int lddp, void* ds, int64_t rows, int cols,
                                             int ldp, float scale, void* stream) {
  CTTA_REQUIRE(p && dp && ds && ldp >= cols && lddp >= cols, "softmax_bwd_rows: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)p, dp,
                     lddp, (bf16_t*)ds, cols, ldp, scale);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ fp32 embedding-MLP backward
// y = x W^T + b (W [N][K]);  given dy [M][N]:  dx[m][k] = sum_n dy[m][n] W[n][k] (* silu'(xpre) when xpre given),
// dW[n][k] (+)= sum_m dy[m][n] x[m][k], db[n] (+)= sum_m dy[m][n]
// grid (K/64, M, n-chunks): 64 k-columns x 4 n-lanes per block, partial sums land with one atomicAdd per (m, k)
__global__ __launch_bounds__(256) void linear_f32_bwd_dx_kernel(const float* __restrict__ dy, int ldy,
                                                                const float* __restrict__ w, float* __restrict__ dx, int N,
                                                                int K, int n_chunk) {
  __shared__ float red[4][64];
  const int kk = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kk, m = blockIdx.y;
  const int n0 = blockIdx.z * n_chunk, n1 = min(N, n0 + n_chunk);
  float acc = 0.f;
  if (k < K)
    for (int n = n0 + part; n < n1; n += 4) acc += dy[(size_t)m * ldy + n] * w[(size_t)n * K + k];
  red[part][kk] = acc;
  __syncthreads();
  if (part == 0 && k < K) atomicAdd(&dx[(size_t)m * K + k], red[0][kk] + red[1][kk] + red[2][kk] + red[3][kk]);
}
__global__ void silu_grad_scale_kernel(float* __restrict__ dx, const float* __restrict__ xpre, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float z = xpre[i];
  const float s = 1.0f / (1.0f + expf(-z));
  dx[i] *= s * (1.0f + z * (1.0f - s));
}
__global__ void linear_f32_bwd_dw_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                         float* __restrict__ dw, float* __restrict__ db, int M, int N, int K,
                                         int accumulate) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)N * K) return;
  const int n = (int)(i / K), k = (int)(i - (long long)n * K);
  float acc = 0.f, bs = 0.f;
  for (int m = 0; m < M; ++m) {
    const float d = dy[(size_t)m * ldy + n];
    acc += d * x[(size_t)m * K + k];
    bs += d;
  }
  dw[i] = accumulate ? dw[i] + acc : acc;
  if (db && k == 0) db[n] = accumulate ? db[n] + bs : bs;
}
extern "C" ctta_status ctta_linear_f32_bwd(const float* x, const float* w, const float* dy, int dy_ld,
                                           const float* xpre_silu, float* dx, float* dw, float* db, int m, int n, int k,
                                           int accumulate_dx, int accumulate_param, void* stream) {
  CTTA_REQUIRE(x && w && dy && dy_ld >= n, "linear_f32_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    CTTA_REQUIRE(!(accumulate_dx && xpre_silu), "linear_f32_bwd: accumulate_dx with a SiLU pre-activation is not supported");
    if (!accumulate_dx) CTTA_CHECK_HIP(ctta_zero_async(dx, (size_t)m * k * sizeof(float), s));
    int chunks = (n + 255) / 256;
    if (chunks > 64) chunks = 64;
    const int n_chunk = (n + chunks - 1) / chunks;
    hipLaunchKernelGGL(linear_f32_bwd_dx_kernel, dim3((k + 63) / 64, m, chunks), dim3(256), 0, s, dy, dy_ld, w, dx, n, k,
                       n_chunk);
    CTTA_LAUNCH_CHECK();
    if (xpre_silu) {
      hipLaunchKernelGGL(silu_grad_scale_kernel, dim3((m * k + 255) / 256), dim3(256), 0, s, dx, xpre_silu, m * k);
      CTTA_LAUNCH_CHECK();
    }
  }
  if (dw) {
    const long long total = (long long)n * k;
    hipLaunchKernelGGL(linear_f32_bwd_dw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dy, dy_ld, x, dw,
                       db, m, n, k, accumulate_param);
    CTTA_LAUNCH_CHECK();
  }
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ loss gradient
// loss = mean_b( w_b * mean((pred-target)^2) ), w_b = min(sigma_b^-2, gamma) (1 when gamma <= 0)
// d pred (NCHW fp32) -> NHWC bf16 [B][HW][Cpad] gradient, scaled by loss_scale
__global__ void snr_mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                    const float* __restrict__ sigma, float gamma, float loss_scale, int B, int C, int HW,
                                    int cpad, bf16_t* __restrict__ out) {
  const long long total = (long long)B * HW * cpad;
  const float n = (float)C * (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad);
    const long long pix = i / cpad;
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (long long)b * HW);
    float g = 0.f;
    if (c < C) {
      float w = 1.0f;
      if (gamma > 0.f) { const float sg = sigma[b]; w = fminf(1.0f / (sg * sg), gamma); }
      const size_t j = ((size_t)b * C + c) * HW + hw;
      g = 2.0f * w / (n * (float)B) * (pred[j] - target[j]) * loss_scale;
    }
    out[i] = f2bf(g);
  }
}
extern "C" ctta_status ctta_snr_mse_grad(const float* pred, const float* target, const float* sigma, float gamma,
                                         float loss_scale, int batch, int c, int hw, int c_pad, void* dpred_nhwc,
                                         void* stream) {
  CTTA_REQUIRE(pred && target && dpred_nhwc && (gamma <= 0.f || sigma) && c_pad >= c, "snr_mse_grad: bad arguments");
  const long long total = (long long)batch * hw * c_pad;
  hipLaunchKernelGGL(snr_mse_grad_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, pred, target, sigma,
                     gamma, loss_scale, batch, c, hw, c_pad, (bf16_t*)dpred_nhwc);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// loss = mean_b( w_b * mean((pred-target)^2) ) with explicit per-instance weights (NULL = 1): stage-1 guided distillation
// (models/audio_guided_model.py:92-117, min-SNR weights computed by the caller)
__global__ void weighted_mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                         const float* __restrict__ w, float loss_scale, int B, int C, int HW, int cpad,
                                         bf16_t* __restrict__ out) {
  const long long total = (long long)B * HW * cpad;
  const float n = (float)C * (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cpad);
    const long long pix = i / cpad;
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (long long)b * HW);
    float g = 0.f;
    if (c < C) {
      const size_t j = ((size_t)b * C + c) * HW + hw;
      g = 2.0f * (w ? w[b] : 1.0f) / (n * (float)B) * (pred[j] - target[j]) * loss_scale;
    }
    out[i] = f2bf(g);
  }
}
extern "C" ctta_status ctta_weighted_mse_grad(const float* pred, const float* target, const float* weights,
                                              float loss_scale, int batch, int c, int hw, int c_pad, void* dpred_nhwc,
                                              void* stream) {
  CTTA_REQUIRE(pred && target && dpred_nhwc && c_pad >= c, "weighted_mse_grad: bad arguments");
  const long long total = (long long)batch * hw * c_pad;
  hipLaunchKernelGGL(weighted_mse_grad_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, pred, target,
                     weights, loss_scale, batch, c, hw, c_pad, (bf16_t*)dpred_nhwc);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ vocoder / decoder input gradients
// LeakyReLU backward with the residual stream folded in (ResBlock.forward, hifigan/models.py:56-63, in reverse):
//   out (+)= res + alpha * g * (act > 0 ? 1 : slope)
// `act` is the SAVED leaky_relu output (same sign as its argument since slope > 0).
__global__ void lrelu_bwd_kernel(const uint4* __restrict__ g, const uint4* __restrict__ act, float slope, float alpha,
                                 const uint4* __restrict__ res, uint4* __restrict__ out, long long nvec, int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float fg[8], fa[8], fo[8];
    unpack8(g[i], fg);
    unpack8(act[i], fa);
#pragma unroll
    for (int e = 0; e < 8; ++e) fo[e] = alpha * fg[e] * (fa[e] > 0.f ? 1.0f : slope);
    if (res) {
      float fr[8];
      unpack8(res[i], fr);
#pragma unroll
      for (int e = 0; e < 8; ++e) fo[e] += fr[e];
    }
    if (accumulate) {
      float fp[8];
      unpack8(out[i], fp);
#pragma unroll
      for (int e = 0; e < 8; ++e) fo[e] += fp[e];
    }
    out[i] = pack8(fo);
  }
}
extern "C" ctta_status ctta_lrelu_bwd(const void* g, const void* act, float slope, float alpha, const void* res, void* out,
                                      int64_t n, int accumulate, void* stream) {
  CTTA_REQUIRE(g && act && out && n % 8 == 0, "lrelu_bwd: bad arguments");
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid1d(n / 8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const uint4*)g,
                     (const uint4*)act, slope, alpha, (const uint4*)res, (uint4*)out, (long long)(n / 8), accumulate);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// Data gradient of a single-output-channel convolution (the decoder's conv_out and the vocoder's conv_post):
//   dx[b][y][x][c] = mask * sum_taps w[tap][c] * gy'[b][y - (a - pad_h)][x - (bq - pad_w)],
// gy' = gy * (1 - yt^2) when yt (the tanh output) is given; mask = leaky_relu'(act) when act is given.
__global__ void conv_cout1_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ yt,
                                        const float* __restrict__ w, int B, int H, int W, int kh, int kw, int pad_h,
                                        int pad_w, int C, const bf16_t* __restrict__ act, float slope,
                                        bf16_t* __restrict__ dx) {
  const int vc = C / 8;
  const long long total = (long long)B * H * W * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    long long p = idx / vc;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int a = 0; a < kh; ++a) {
      const int yy = y - (a - pad_h);
      if (yy < 0 || yy >= H) continue;
      for (int q = 0; q < kw; ++q) {
        const int xx = x - (q - pad_w);
        if (xx < 0 || xx >= W) continue;
        const size_t o = ((size_t)b * H + yy) * W + xx;
        float gv = gy[o];
        if (yt) { const float t = yt[o]; gv *= 1.0f - t * t; }
        const float* wr = w + (size_t)(a * kw + q) * C + v * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wr[e] * gv;
      }
    }
    if (act) {
      float fa[8];
      unpack8(*reinterpret_cast<const uint4*>(act + (size_t)idx * 8), fa);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= fa[e] > 0.f ? 1.0f : slope;
    }
    *reinterpret_cast<uint4*>(dx + (size_t)idx * 8) = pack8(acc);
  }
}
extern "C" ctta_status ctta_conv_cout1_dgrad(const float* gy, const float* y_tanh, const float* w, int batch, int h, int wd,
                                             int kh, int kw, int pad_h, int pad_w, int c, const void* act, float slope,
                                             void* dx, void* stream) {
  CTTA_REQUIRE(gy && w && dx && c % 8 == 0 && kh >= 1 && kw >= 1, "conv_cout1_dgrad: bad arguments");
  const long long total = (long long)batch * h * wd * (c / 8);
  hipLaunchKernelGGL(conv_cout1_dgrad_kernel, dim3(grid1d(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, gy,
                     y_tanh, w, batch, h, wd, kh, kw, pad_h, pad_w, c, (const bf16_t*)act, slope, (bf16_t*)dx);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ AdamW (torch.optim.AdamW, no amsgrad)
// One element's update with the contractions written out: left to the compiler (-ffp-contract=fast), `beta2 * v + omb2 * g * g`
// became fma(beta2, v, (omb2 g) g) in one kernel and fma(omb2 g, g, beta2 v) in another -- two valid roundings, and the fused
// optimizer tail below was not bit-identical to the three launches it replaces.  Every AdamW kernel of the library goes
// through this function.
struct AdamWConst { float decay, step, omb1, omb2, beta2, bc2_sqrt, eps, grad_scale; };
__device__ __forceinline__ void adamw_one(float& pp, float gq, float& mm, float& vv, const AdamWConst& c) {
#pragma clang fp contract(off)
  const float gr = gq * c.grad_scale;
  const float q = pp * c.decay;
  mm = fmaf(c.omb1, gr - mm, mm);                       // exp_avg.lerp_(grad, 1 - beta1)
  vv = fmaf(c.omb2 * gr, gr, c.beta2 * vv);
  const float denom = sqrtf(vv) / c.bc2_sqrt + c.eps;
  pp = fmaf(-c.step, mm / denom, q);
}
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, float lr, float beta1, float beta2, float eps, float wd,
                             float bc1, float bc2_sqrt, float grad_scale) {
  const AdamWConst c = {1.0f - lr * wd, lr / bc1, 1.0f - beta1, 1.0f - beta2, beta2, bc2_sqrt, eps, grad_scale};
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float pp = p[i], mm = m[i], vv = v[i];
    adamw_one(pp, g[i], mm, vv, c);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}
// the same arithmetic on 16-byte vectors, two vectors in flight per thread (seven fp32 streams: the scalar kernel above
// moved 4.8 TB/s; round 4)
__global__ __launch_bounds__(256) void adamw4_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                     float4* __restrict__ m, float4* __restrict__ v, long long n4, float lr,
                                                     float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                     float grad_scale) {
  const AdamWConst c = {1.0f - lr * wd, lr / bc1, 1.0f - beta1, 1.0f - beta2, beta2, bc2_sqrt, eps, grad_scale};
  auto one = [&](float& pp, float gq, float& mm, float& vv) { adamw_one(pp, gq, mm, vv, c); };
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
    const long long j = i + stride;
    const bool two = j < n4;
    float4 p0 = p[i], g0 = g[i], m0 = m[i], v0 = v[i];
    float4 p1, g1, m1, v1;
    if (two) { p1 = p[j]; g1 = g[j]; m1 = m[j]; v1 = v[j]; }
    one(p0.x, g0.x, m0.x, v0.x); one(p0.y, g0.y, m0.y, v0.y); one(p0.z, g0.z, m0.z, v0.z); one(p0.w, g0.w, m0.w, v0.w);
    p[i] = p0; m[i] = m0; v[i] = v0;
    if (two) {
      one(p1.x, g1.x, m1.x, v1.x); one(p1.y, g1.y, m1.y, v1.y); one(p1.z, g1.z, m1.z, v1.z); one(p1.w, g1.w, m1.w, v1.w);
      p[j] = p1; m[j] = m1; v[j] = v1;
    }
  }
}
// ------------------------------------------------------------------------------ the optimizer tail as ONE pass
// tools/train_utils.py:177-183 runs optimizer.step() -> optimizer.zero_grad() -> update_ema() (255-282) back to back: three
// passes over the training state, 7 + 1 + 5 = 13 fp32 streams of 559 M elements = 29.1 GB per step.  One pass reads p, g, m, v
// and the two shadows and writes p, m, v, both shadows and g = 0: 12 streams; every element sees the same fp32 operations in
// the same order as in the three launches (AdamW contracted like adamw4_kernel in this translation unit, the EMA line
// uncontracted like elementwise.hip's ema2_kernel), so the results are bit-identical (tests/test_bwd_ops_gpu.py).
// Elements [0, n_train) take the AdamW update (unless do_step == 0: the reference skips it on a NaN loss but still zeroes
// the gradients and moves the shadows); [n_train, n_all) are the frozen suffix of the flat buffers.
__device__ __forceinline__ float ema_lerp(float s, float p, float k) {
#pragma clang fp contract(off)
  return s + k * (p - s);
}
template <bool TWO>
__global__ __launch_bounds__(256) void adamw_ema2_zero_kernel(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                                              float4* __restrict__ v, float4* __restrict__ sa, float ka,
                                                              float4* __restrict__ sb, float kb, long long nt4, long long n4,
                                                              int do_step, float lr, float beta1, float beta2, float eps, float wd,
                                                              float bc1, float bc2_sqrt, float grad_scale) {
  const AdamWConst c = {1.0f - lr * wd, lr / bc1, 1.0f - beta1, 1.0f - beta2, beta2, bc2_sqrt, eps, grad_scale};
  auto one = [&](float& pp, float gq, float& mm, float& vv) { adamw_one(pp, gq, mm, vv, c); };
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
    const long long j = i + stride;
    const bool two = j < n4;
    const bool ti = do_step && i < nt4, tj = do_step && two && j < nt4;
    float4 p0 = p[i], a0 = sa[i], b0, p1, a1, b1, g0, m0, v0, g1, m1, v1;
    if (TWO) b0 = sb[i];
    if (ti) { g0 = g[i]; m0 = m[i]; v0 = v[i]; }
    if (two) { p1 = p[j]; a1 = sa[j]; if (TWO) b1 = sb[j]; }
    if (tj) { g1 = g[j]; m1 = m[j]; v1 = v[j]; }
    if (ti) {
      one(p0.x, g0.x, m0.x, v0.x); one(p0.y, g0.y, m0.y, v0.y); one(p0.z, g0.z, m0.z, v0.z); one(p0.w, g0.w, m0.w, v0.w);
      p[i] = p0; m[i] = m0; v[i] = v0;
    }
    g[i] = zero;
    a0.x = ema_lerp(a0.x, p0.x, ka); a0.y = ema_lerp(a0.y, p0.y, ka); a0.z = ema_lerp(a0.z, p0.z, ka); a0.w = ema_lerp(a0.w, p0.w, ka);
    sa[i] = a0;
    if (TWO) {
      b0.x = ema_lerp(b0.x, p0.x, kb); b0.y = ema_lerp(b0.y, p0.y, kb); b0.z = ema_lerp(b0.z, p0.z, kb); b0.w = ema_lerp(b0.w, p0.w, kb);
      sb[i] = b0;
    }
    if (two) {
      if (tj) {
        one(p1.x, g1.x, m1.x, v1.x); one(p1.y, g1.y, m1.y, v1.y); one(p1.z, g1.z, m1.z, v1.z); one(p1.w, g1.w, m1.w, v1.w);
        p[j] = p1; m[j] = m1; v[j] = v1;
      }
      g[j] = zero;
      a1.x = ema_lerp(a1.x, p1.x, ka); a1.y = ema_lerp(a1.y, p1.y, ka); a1.z = ema_lerp(a1.z, p1.z, ka); a1.w = ema_lerp(a1.w, p1.w, ka);
      sa[j] = a1;
      if (TWO) {
        b1.x = ema_lerp(b1.x, p1.x, kb); b1.y = ema_lerp(b1.y, p1.y, kb); b1.z = ema_lerp(b1.z, p1.z, kb); b1.w = ema_lerp(b1.w, p1.w, kb);
        sb[j] = b1;
      }
    }
  }
}
extern "C" ctta_status ctta_adamw_ema2_zero(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n_train,
                                            int64_t n_all, float* shadow_a, double decay_a, float* shadow_b, double decay_b,
                                            int do_step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                            int step, float grad_scale, void* stream) {
  CTTA_REQUIRE(param && grad && exp_avg && exp_avg_sq && shadow_a && step >= 1, "adamw_ema2_zero: bad arguments");
  CTTA_REQUIRE(n_train >= 0 && n_train <= n_all && n_train % 4 == 0 && n_all % 4 == 0,
               "adamw_ema2_zero: n_train=%lld / n_all=%lld must be multiples of 4 with n_train <= n_all", (long long)n_train, (long long)n_all);
  CTTA_REQUIRE(decay_a >= 0.0 && decay_a <= 1.0 && decay_b >= 0.0 && decay_b <= 1.0, "adamw_ema2_zero: decay outside [0,1]");
  CTTA_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)shadow_a | (uintptr_t)shadow_b) & 15) == 0,
               "adamw_ema2_zero: buffers must be 16-byte aligned");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const float ka = (float)(1.0 - decay_a), kb = (float)(1.0 - decay_b);
  const long long n4 = n_all / 4, nt4 = n_train / 4;
  if (n4 == 0) return CTTA_OK;
  const dim3 grid(grid1d(n4, 256, 2048)), block(256);
  if (shadow_b)
    hipLaunchKernelGGL(adamw_ema2_zero_kernel<true>, grid, block, 0, (hipStream_t)stream, (float4*)param, (float4*)grad, (float4*)exp_avg,
                       (float4*)exp_avg_sq, (float4*)shadow_a, ka, (float4*)shadow_b, kb, nt4, n4, do_step, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, grad_scale);
  else
    hipLaunchKernelGGL(adamw_ema2_zero_kernel<false>, grid, block, 0, (hipStream_t)stream, (float4*)param, (float4*)grad, (float4*)exp_avg,
                       (float4*)exp_avg_sq, (float4*)shadow_a, ka, (float4*)nullptr, 0.f, nt4, n4, do_step, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, grad_scale);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                       float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                       float grad_scale, void* stream) {
  CTTA_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1, "adamw_step: bad arguments");
  // bias corrections in double on the host, like torch's Python-float arithmetic (optim/adamw.py)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  const bool aligned = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
  const long long n4 = aligned ? n / 4 : 0;
  if (n4 > 0) {
    hipLaunchKernelGGL(adamw4_kernel, dim3(grid1d(n4, 256, 2048)), dim3(256), 0, (hipStream_t)stream, (float4*)param,
                       (const float4*)grad, (float4*)exp_avg, (float4*)exp_avg_sq, n4, lr, beta1, beta2, eps, weight_decay, bc1,
                       bc2s, grad_scale);
    CTTA_LAUNCH_CHECK();
  }
  const long long done = n4 * 4;
  if (done < n) {
    hipLaunchKernelGGL(adamw_kernel, dim3(grid1d(n - done, 256, 4096)), dim3(256), 0, (hipStream_t)stream, param + done,
                       grad + done, exp_avg + done, exp_avg_sq + done, (long long)(n - done), lr, beta1, beta2, eps, weight_decay,
                       bc1, bc2s, grad_scale);
    CTTA_LAUNCH_CHECK();
  }
  return CTTA_OK;
}
