// HBM-bound kernels of the path: layout packs, GroupNorm(+SiLU), LayerNorm, GEGLU, row
// softmax, concat, weight packing, small fp32 linears and the time/guidance features.
// All bf16 traffic is 16 bytes per lane (8 channels), NHWC, fp32 statistics.
#include "common.h"

#include <math.h>
#include <stdlib.h>

// ------------------------------------------------------------------------------ layout packs
// (B,C,H,W) f32 -> (B,H,W,Cp) bf16, channels >= C zero.  One thread per (pixel, 8-ch group).
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B,
                                    int C, int HW, int Cp, float scale) {
  const int vc = Cp / 8;
  const long long total = (long long)B * HW * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    const long long pix = idx / vc;
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (long long)b * HW);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = v * 8 + e;
      f[e] = c < C ? src[((size_t)b * C + c) * HW + hw] * scale : 0.f;
    }
    *reinterpret_cast<uint4*>(dst + (size_t)pix * Cp + v * 8) = pack8(f);
  }
}

// (B,H,W,Cs) bf16 -> (B,C,H,W) f32.  Threads run along hw for coalesced fp32 stores.
__global__ void nhwc_to_nchw_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int B,
                                    int C, int HW, int Cs) {
  const long long total = (long long)B * C * HW;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int hw = (int)(idx % HW);
    const long long bc = idx / HW;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    dst[idx] = bf2f(src[((size_t)b * HW + hw) * Cs + c]);
  }
}

// (B,HW,Cs) f32 -> (B,C,HW) f32 (the U-Net's conv_out result, written [pixel][8] by conv_gemm, to the caller's NCHW)
__global__ void nhwc_f32_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int HW, int Cs) {
  const long long total = (long long)B * C * HW;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int hw = (int)(idx % HW);
    const long long bc = idx / HW;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    dst[idx] = src[((size_t)b * HW + hw) * Cs + c];
  }
}
extern "C" ctta_status ctta_nhwc_f32_to_nchw_f32(const float* src, float* dst, int batch, int c, int hw, int c_stride, void* stream) {
  CTTA_REQUIRE(src && dst && batch > 0 && c > 0 && hw > 0 && c_stride >= c, "nhwc_f32_to_nchw_f32: bad arguments");
  const long long total = (long long)batch * c * hw;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(nhwc_f32_to_nchw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, batch, c, hw, c_stride);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

__global__ void rows_f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                        long long rows, int cols, int cols_pad) {
  const int vc = cols_pad / 8;
  const long long total = rows * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    const long long r = idx / vc;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = v * 8 + e;
      f[e] = c < cols ? src[(size_t)r * cols + c] : 0.f;
    }
    *reinterpret_cast<uint4*>(dst + (size_t)r * cols_pad + v * 8) = pack8(f);
  }
}

__global__ void concat_kernel(const uint4* __restrict__ a, int va, const uint4* __restrict__ b,
                              int vb, uint4* __restrict__ dst, long long pixels) {
  const int vt = va + vb;
  const long long total = pixels * vt;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vt);
    const long long pix = idx / vt;
    dst[idx] = v < va ? a[pix * va + v] : b[pix * vb + (v - va)];
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ src, const int* __restrict__ row_off,
                                   const int* __restrict__ col_off, const int* __restrict__ row_aux,
                                   const int* __restrict__ col_aux, int aux_limit, int n_rows,
                                   int k_pad, bf16_t* __restrict__ dst) {
  const long long total = (long long)n_rows * k_pad;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % k_pad);
    const int r = (int)(idx / k_pad);
    const int ro = row_off[r], co = col_off[k];
    bool ok = ro >= 0 && co >= 0;
    if (ok && aux_limit > 0) ok = row_aux[r] + col_aux[k] < aux_limit;
    dst[idx] = ok ? f2bf(src[(size_t)ro + (size_t)co]) : (bf16_t)0;
  }
}

// All pack jobs of a state dict in one launch: a block finds its job by bisection over block0 and packs
// CTTA_PACK_ELEMS_PER_BLOCK consecutive destination elements (consecutive blocks walk the same destination row, so
// the strided source reads of a conv row -- (cin, kh, kw) -> (kh, kw, cin) -- are served by L2 after the first tap;
// one-row-per-block variants measured slower: too few blocks in flight for the 1024 x 9216 layers).
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const ctta_pack_job* __restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ctta_pack_job j = jobs[lo];
  const long long total = (long long)j.n_rows * j.k_pad;
  const long long base = (long long)(blockIdx.x - j.block0) * CTTA_PACK_ELEMS_PER_BLOCK;
  bf16_t* dst = (bf16_t*)j.dst;
  static_assert(CTTA_PACK_ELEMS_PER_BLOCK == 256 * 8, "one 8-element group per thread");
  // a thread packs 8 consecutive destination elements (k_pad is a multiple of 64: they share the row), so the
  // row / column split is one division per 16-byte store instead of two 64-bit divisions per element
  const long long idx = base + (long long)threadIdx.x * 8;
  if (idx >= total) return;
  const int r = (int)(idx / j.k_pad);
  const int k = (int)(idx - (long long)r * j.k_pad);
  const int ro = j.row_off[r];
  const int4 c0 = *reinterpret_cast<const int4*>(j.col_off + k);
  const int4 c1 = *reinterpret_cast<const int4*>(j.col_off + k + 4);
  const int co[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bool ok = ro >= 0 && co[e] >= 0;
    if (ok && j.aux_limit > 0) ok = j.row_aux[r] + j.col_aux[k + e] < j.aux_limit;
    f[e] = ok ? j.src[(size_t)ro + (size_t)co[e]] : 0.f;
  }
  *reinterpret_cast<uint4*>(dst + idx) = pack8(f);
}
// Row-staged pack: a block owns rows_per_block whole destination rows of one job.  Their source rows (contiguous
// fp32 ranges) are read once, coalesced, into LDS; the (cin, kh, kw) -> (kh, kw, cin) permutation is then an LDS
// gather with a destination pair per lane (lane stride 2 * kh * kw floats: at most 2-way bank conflicts for 3x3
// taps) and a coalesced 4-byte store.  HBM traffic is the algorithmic 4 + 2 bytes per element; the column table
// (4 bytes per element, shared by all rows of a job) is served by the L2.
#define CTTA_PACK_ROWS_MAX_RPB 64
__global__ __launch_bounds__(1024) void pack_weight_rows_multi_kernel(const ctta_pack_job* __restrict__ jobs, int n_jobs) {
  extern __shared__ float srow[];
  __shared__ int s_ro[CTTA_PACK_ROWS_MAX_RPB];
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ctta_pack_job j = jobs[lo];
  const int r0 = ((int)blockIdx.x - j.block0) * j.rows_per_block;
  const int nr = min(j.rows_per_block, j.n_rows - r0);
  const int L = j.src_row_len, kp = j.k_pad;
  const int nt = blockDim.x;
  if ((int)threadIdx.x < nr) s_ro[threadIdx.x] = j.row_off[r0 + threadIdx.x];
  __syncthreads();
  if ((L & 3) == 0 && (reinterpret_cast<uintptr_t>(j.src) & 15) == 0) {   // row_off is a multiple of L: 16-byte rows
    const int L4 = L >> 2, n4 = nr * L4;
    float4* s4 = reinterpret_cast<float4*>(srow);
#pragma unroll 4
    for (int i = threadIdx.x; i < n4; i += nt) {
      const int r = i / L4, c = i - r * L4;
      const int ro = s_ro[r];
      if (ro >= 0) s4[i] = *reinterpret_cast<const float4*>(j.src + ro + 4 * c);
    }
  } else {
    const int n = nr * L;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += nt) {
      const int r = i / L, c = i - r * L;
      const int ro = s_ro[r];
      if (ro >= 0) srow[i] = j.src[(size_t)ro + c];
    }
  }
  __syncthreads();
  bf16_t* dst = (bf16_t*)j.dst + (size_t)r0 * kp;
  const int total = nr * kp;   // k_pad is a multiple of 64: a pair never straddles rows
  for (int base = threadIdx.x * 2; base < total; base += nt * 8) {
    int2 c[4];
    int rr[4], kk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * nt * 2;
      const int r = idx < total ? idx / kp : 0;
      rr[u] = r; kk[u] = idx < total ? idx - r * kp : 0;
      c[u] = *reinterpret_cast<const int2*>(j.col_off + kk[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * nt * 2;
      if (idx >= total) break;
      const int ro = s_ro[rr[u]];
      bool ok0 = ro >= 0 && c[u].x >= 0, ok1 = ro >= 0 && c[u].y >= 0;
      if (j.aux_limit > 0 && ro >= 0) {
        const int ra = j.row_aux[r0 + rr[u]];
        ok0 = ok0 && ra + j.col_aux[kk[u]] < j.aux_limit;
        ok1 = ok1 && ra + j.col_aux[kk[u] + 1] < j.aux_limit;
      }
      const float* sr = srow + rr[u] * L;
      const float f0 = ok0 ? sr[c[u].x] : 0.f;
      const float f1 = ok1 ? sr[c[u].y] : 0.f;
      *reinterpret_cast<uint32_t*>(dst + idx) = pack2bf(f0, f1);
    }
  }
}
extern "C" ctta_status ctta_pack_weight_rows_multi(const ctta_pack_job* jobs, int n_jobs, int total_blocks,
                                                   int lds_floats, int threads, void* stream) {
  CTTA_REQUIRE(jobs && n_jobs >= 1 && total_blocks >= 1 && lds_floats >= 1, "pack_weight_rows_multi: bad arguments");
  CTTA_REQUIRE(threads == 256 || threads == 512 || threads == 1024, "pack_weight_rows_multi: threads must be 256, 512 or 1024");
  const size_t smem = (size_t)lds_floats * sizeof(float);
  CTTA_REQUIRE(smem + sizeof(int) * CTTA_PACK_ROWS_MAX_RPB <= 160 * 1024,
               "pack_weight_rows_multi: %d staged floats do not fit the LDS", lds_floats);
  static size_t configured = 0;
  if (smem > configured) {
    CTTA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_weight_rows_multi_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    configured = smem;
  }
  hipLaunchKernelGGL(pack_weight_rows_multi_kernel, dim3(total_blocks), dim3(threads), smem, (hipStream_t)stream, jobs, n_jobs);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
__global__ __launch_bounds__(256) void copy_segments_multi_kernel(const ctta_copy_seg* __restrict__ segs) {
  const ctta_copy_seg g = segs[blockIdx.x];
  for (int i = threadIdx.x; i < g.count; i += 256) g.dst[i] = g.src[i];
}
extern "C" ctta_status ctta_pack_weight_multi(const ctta_pack_job* jobs, int n_jobs, int total_blocks, void* stream) {
  CTTA_REQUIRE(jobs && n_jobs >= 1 && total_blocks >= 1, "pack_weight_multi: bad arguments");
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_copy_segments_multi(const ctta_copy_seg* segs, int n_segs, void* stream) {
  CTTA_REQUIRE(segs && n_segs >= 1, "copy_segments_multi: bad arguments");
  hipLaunchKernelGGL(copy_segments_multi_kernel, dim3(n_segs), dim3(256), 0, (hipStream_t)stream, segs);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ GroupNorm
// Pass 1: per (batch, pixel-chunk) partial sums per group.  A thread owns one 8-channel
// vector column (tid % VC) and strides over the chunk's pixels; per-channel partials meet in
// LDS and are folded per group -- deterministic (no atomics).
// part layout: [B][nchunk][G][2] floats (sum, sumsq).
// CAT: x is the channel concatenation [x (C1 channels) | x2 (C - C1 channels)] (torch.cat([h, skip], 1) of the up blocks),
// which this pass also WRITES to y while it sums it: the separate concat launch and its re-read disappear.
template <bool CAT>
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, int HW, int C,
                                                         int G, int pix_per_chunk, int nchunk,
                                                         float* __restrict__ part, const bf16_t* __restrict__ x2, int C1,
                                                         bf16_t* __restrict__ y) {
  extern __shared__ float sm[];  // [PL][C] sum, [PL][C] sumsq
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int VC = C / 8;
  const int tid = threadIdx.x;
  int PL, vstride, tv, tp;
  if (VC <= 256) { PL = 256 / VC; vstride = VC; tv = tid % VC; tp = tid / VC; }
  else { PL = 1; vstride = 256; tv = tid; tp = 0; }
  const int p_begin = chunk * pix_per_chunk;
  const int p_end = min(HW, p_begin + pix_per_chunk);
  float* ssum = sm;
  float* ssq = sm + PL * C;
  if (tp < PL) {
    for (int v = tv; v < VC; v += vstride) {
      float s[8], q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
      // four pixels requested together, summed in pixel order (the rolled loop was one memory round trip per pixel and
      // lane; a pixel past the chunk contributes + 0.0: the same sums)
      for (int pix0 = p_begin + tp; pix0 < p_end; pix0 += 4 * PL) {
        uint4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pix = pix0 + u * PL;
          raw[u] = make_uint4(0, 0, 0, 0);
          if (pix < p_end) {
            if (CAT)
              raw[u] = v * 8 < C1 ? *reinterpret_cast<const uint4*>(x + ((size_t)b * HW + pix) * C1 + v * 8)
                                  : *reinterpret_cast<const uint4*>(x2 + ((size_t)b * HW + pix) * (C - C1) + v * 8 - C1);
            else
              raw[u] = *reinterpret_cast<const uint4*>(x + ((size_t)b * HW + pix) * C + v * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pix = pix0 + u * PL;
          if (CAT && pix < p_end) *reinterpret_cast<uint4*>(y + ((size_t)b * HW + pix) * C + v * 8) = raw[u];
          float f[8];
          unpack8(raw[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ssum[tp * C + v * 8 + e] = s[e];
        ssq[tp * C + v * 8 + e] = q[e];
      }
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int g = tid; g < G; g += 256) {
    float s = 0.f, q = 0.f;
    for (int l = 0; l < PL; ++l)
      for (int cc = 0; cc < cpg; ++cc) {
        s += ssum[l * C + g * cpg + cc];
        q += ssq[l * C + g * cpg + cc];
      }
    float* o = part + (((size_t)b * nchunk + chunk) * G + g) * 2;
    o[0] = s;
    o[1] = q;
  }
}

// One lane's share of a group's partials -- chunks l, l + LPG, l + 2 LPG, ... of sample b -- summed in that order in
// double.  Eight loads are requested before the first is added (the rolled loop paid one L2 round trip per chunk: 19 us
// per finalize launch on average, 1.3 ms of a B = 32 generation step); a chunk past the end contributes + 0.0, which
// leaves the sums as they are, so the result is the rolled loop's bit for bit.
__device__ __forceinline__ void gn_fold_chunks(const float* __restrict__ pg, int nchunk, int G, int l, int LPG, double& s,
                                               double& q) {
  constexpr int U = 8;
  for (int ch = l; ch < nchunk; ch += LPG * U) {
    float2 pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = ch + u * LPG;
      pp[u] = c < nchunk ? *reinterpret_cast<const float2*>(pg + (size_t)c * G * 2) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s += (double)pp[u].x;
      q += (double)pp[u].y;
    }
  }
}

// Pass 2: fold chunks in double, emit per-(batch, channel) scale/shift:
//   y = x*scale + shift,  scale = rstd*gamma, shift = beta - mean*rstd*gamma.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, int nchunk, int G, int C,
                                                          int HW, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale_shift, float* __restrict__ stats) {
  const int b = blockIdx.x;
  const int cpg = C / G;
  // LPG lanes share a group: each folds every LPG-th chunk, then a butterfly (fixed order, so the
  // result does not depend on the launch) leaves the total in all of them.
  int LPG = 1;
  if (G <= 128 && 256 % G == 0) LPG = min(64, 256 / G);
  const int per_pass = 256 / LPG;
  for (int g0 = 0; g0 < G; g0 += per_pass) {
    const int g = g0 + (int)threadIdx.x / LPG, l = (int)threadIdx.x % LPG;
    double s = 0.0, q = 0.0;
    if (g < G) gn_fold_chunks(part + ((size_t)b * nchunk * G + g) * 2, nchunk, G, l, LPG, s, q);
    for (int off = 1; off < LPG; off <<= 1) {
      s += __shfl_xor(s, off);
      q += __shfl_xor(q, off);
    }
    if (g >= G) continue;
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    if (stats && l == 0) {   // training forward: (mean, rstd) per (sample, group) for the backward pass
      stats[((size_t)b * G + g) * 2] = meanf;
      stats[((size_t)b * G + g) * 2 + 1] = rstd;
    }
    for (int cc = l; cc < cpg; cc += LPG) {
      const int c = g * cpg + cc;
      const float sc = rstd * gamma[c];
      scale_shift[((size_t)b * 2 + 0) * C + c] = sc;
      scale_shift[((size_t)b * 2 + 1) * C + c] = beta[c] - meanf * sc;
    }
  }
}

// Pass 2 for long chunk lists (nchunk * G > 2048: the VAE's top levels, 256 row tiles per sample): grid (G / 4, batch), 64 lanes
// per group, so a lane folds nchunk / 64 chunks instead of nchunk / 8 (the one-block-per-sample form above spent 14 us per
// call there, 0.8 ms of a batch-32 generation step).  The fused finalize + apply never sees these lists, so its "same order
// as gn_finalize_kernel" contract is untouched.
__global__ __launch_bounds__(256) void gn_finalize_wide_kernel(const float* __restrict__ part, int nchunk, int G, int C, int HW,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               float* __restrict__ scale_shift, float* __restrict__ stats) {
  const int b = blockIdx.y, g = blockIdx.x * 4 + (int)threadIdx.x / 64, l = (int)threadIdx.x % 64;
  const int cpg = C / G;
  double s = 0.0, q = 0.0;
  if (g < G) gn_fold_chunks(part + ((size_t)b * nchunk * G + g) * 2, nchunk, G, l, 64, s, q);
  for (int off = 1; off < 64; off <<= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  if (g >= G) return;
  const double n = (double)HW * cpg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  if (stats && l == 0) {
    stats[((size_t)b * G + g) * 2] = meanf;
    stats[((size_t)b * G + g) * 2 + 1] = rstd;
  }
  for (int cc = l; cc < cpg; cc += 64) {
    const int c = g * cpg + cc;
    const float sc = rstd * gamma[c];
    scale_shift[((size_t)b * 2 + 0) * C + c] = sc;
    scale_shift[((size_t)b * 2 + 1) * C + c] = beta[c] - meanf * sc;
  }
}

static constexpr bool gn_finalize_wide_on() { return true; }

// Pass 3: apply (+SiLU).  grid = (blocks per sample, batch): a block stays inside one sample, and because its stride
// (gridDim.x * 256 vectors) is a multiple of the C / 8 vector columns whenever C / 8 divides 256, a thread keeps ONE
// channel vector for its whole walk -- its 8 scales and 8 shifts are loaded once into registers instead of 64 bytes of
// (cached) table per 16 bytes of x (round 2's kernel: 4.0 TB/s on the 537 MB tensors of the VAE decoder).  Four vectors
// are requested before the first is used.
// FUSED: passes 2 + 3 in one launch for partials that are cheap to fold (the conv epilogue's per-tile partials: <= 2048
// (chunk, group) pairs per sample): every block of sample b folds that sample's partials itself (fp64, the same fixed
// order as gn_finalize_kernel -> the same scale / shift in every block and in the two-launch path), parks them in LDS.
template <bool FUSED>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int HW, int C,
                                                       const float* __restrict__ scale_shift, int silu, int G,
                                                       const float* __restrict__ part, int nchunk,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps) {
  extern __shared__ float ss[];   // FUSED: [2][C] scale, shift
  const int b = blockIdx.y;
  const float* tab = scale_shift + (size_t)b * 2 * C;
  const int VC = C / 8;
  const uint4* xb = reinterpret_cast<const uint4*>(x + (size_t)b * HW * C);
  uint4* yb = reinterpret_cast<uint4*>(y + (size_t)b * HW * C);
  const long long total_end = (long long)HW * VC;
  const long long stride = (long long)gridDim.x * 256;
  long long idx = blockIdx.x * 256LL + threadIdx.x;
  // FUSED: the first four vectors are requested BEFORE the fold of the partials (they do not depend on it), so the fold's
  // two or three L2 round trips pass behind the first HBM round trip instead of in front of it
  uint4 r0[4];
  const bool pre = FUSED && 256 % VC == 0 && idx + 3 * stride < total_end;
  if (pre) {
#pragma unroll
    for (int u = 0; u < 4; ++u) r0[u] = xb[idx + u * stride];
  }
  if constexpr (FUSED) {
    const int cpg = C / G;
    int LPG = 1;
    if (G <= 128 && 256 % G == 0) LPG = min(64, 256 / G);
    const int per_pass = 256 / LPG;
    for (int g0 = 0; g0 < G; g0 += per_pass) {
      const int g = g0 + (int)threadIdx.x / LPG, l = (int)threadIdx.x % LPG;
      double s = 0.0, q = 0.0;
      if (g < G) gn_fold_chunks(part + ((size_t)b * nchunk * G + g) * 2, nchunk, G, l, LPG, s, q);
      for (int off = 1; off < LPG; off <<= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
      }
      if (g >= G) continue;
      const double n = (double)HW * cpg;
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      const float meanf = (float)mean;
      for (int cc = l; cc < cpg; cc += LPG) {
        const int c = g * cpg + cc;
        const float sc = rstd * gamma[c];
        ss[c] = sc;
        ss[C + c] = beta[c] - meanf * sc;
      }
    }
    __syncthreads();
    tab = ss;
  }
  auto apply8 = [&](const uint4 raw, const float* scl, const float* shf) {
    float f[8];
    unpack8(raw, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = f[e] * scl[e] + shf[e];
      f[e] = silu ? silu_f(t) : t;
    }
    return pack8(f);
  };
  if (256 % VC == 0) {   // the thread's channel vector never changes
    const int v = (int)(idx % VC);
    float scl[8], shf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { scl[e] = tab[v * 8 + e]; shf[e] = tab[C + v * 8 + e]; }
    if (pre) {
#pragma unroll
      for (int u = 0; u < 4; ++u) yb[idx + u * stride] = apply8(r0[u], scl, shf);
      idx += 4 * stride;
    }
    for (; idx + 3 * stride < total_end; idx += 4 * stride) {
      uint4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = xb[idx + u * stride];
#pragma unroll
      for (int u = 0; u < 4; ++u) yb[idx + u * stride] = apply8(r[u], scl, shf);
    }
    for (; idx < total_end; idx += stride) yb[idx] = apply8(xb[idx], scl, shf);
    return;
  }
  for (; idx < total_end; idx += stride) {
    const int v = (int)(idx % VC);
    float scl[8], shf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { scl[e] = tab[v * 8 + e]; shf[e] = tab[C + v * 8 + e]; }
    yb[idx] = apply8(xb[idx], scl, shf);
  }
}
static void gn_launch_apply(const void* x, void* y, int batch, int hw, int c, const float* scale_shift, int silu, bool fused,
                            int groups, const float* part, int nchunk, const float* gamma, const float* beta, float eps,
                            hipStream_t s) {
  const long long per_sample = (long long)hw * (c / 8);
  long long bps = cdiv64(per_sample, 256 * 8);          // >= 8 vectors per thread
  const long long cap = cdiv64(8192, batch);
  if (bps > cap) bps = cap;
  if (bps < 1) bps = 1;
  const dim3 grid((unsigned)bps, (unsigned)batch);
  if (fused)
    hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(256), (size_t)2 * c * sizeof(float), s, (const bf16_t*)x, (bf16_t*)y,
                       hw, c, scale_shift, silu, groups, part, nchunk, gamma, beta, eps);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, hw, c, scale_shift, silu,
                       groups, part, nchunk, gamma, beta, eps);
}

// The whole GroupNorm(+SiLU) in ONE launch when a (sample, group) slab is small (HW * C/G <= 16384 elements: the deep
// U-Net levels, whose split-K convolutions emit no statistics): one block per (group, sample) keeps its slab in registers
// (<= 8 vectors of 8 channels per thread), reduces (sum, sum of squares) through LDS in a fixed order, normalises from the
// registers and stores -- one read, one write, no scratch, instead of three launches of 5..16 us each on a 4 MB tensor.
__global__ __launch_bounds__(256) void gn_small_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int HW, int C,
                                                       int G, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu,
                                                       float* __restrict__ stats) {
  constexpr int MAXV = 8;
  __shared__ double red[2][4];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cpg = C / G, vpr = cpg / 8;            // vectors per pixel row of this group
  const int nvec = HW * vpr;
  const bf16_t* xb = x + (size_t)b * HW * C + (size_t)g * cpg;
  bf16_t* yb = y + (size_t)b * HW * C + (size_t)g * cpg;
  uint4 v[MAXV];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = threadIdx.x + i * 256;
    v[i] = make_uint4(0, 0, 0, 0);
    if (idx < nvec) {
      const int pix = idx / vpr, cv = idx - pix * vpr;
      v[i] = *reinterpret_cast<const uint4*>(xb + (size_t)pix * C + cv * 8);
      float f[8];
      unpack8(v[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += f[e]; q += f[e] * f[e]; }
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = (double)s; red[1][threadIdx.x >> 6] = (double)q; }
  __syncthreads();
  const double ts = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  const double tq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double n = (double)HW * cpg;
  const double mean = ts / n;
  double var = tq / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  if (stats && threadIdx.x == 0) {
    stats[((size_t)b * G + g) * 2] = meanf;
    stats[((size_t)b * G + g) * 2 + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = threadIdx.x + i * 256;
    if (idx < nvec) {
      const int pix = idx / vpr, cv = idx - pix * vpr;
      const int c0 = g * cpg + cv * 8;
      float f[8];
      unpack8(v[i], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sc = rstd * gamma[c0 + e];
        const float t = f[e] * sc + (beta[c0 + e] - meanf * sc);
        f[e] = silu ? silu_f(t) : t;
      }
      *reinterpret_cast<uint4*>(yb + (size_t)pix * C + cv * 8) = pack8(f);
    }
  }
}
static bool gn_small_ok(int hw, int c, int groups) {
  const int cpg = c / groups;
  return cpg % 8 == 0 && (long long)hw * cpg <= 16384 && groups <= 65535;
}

static void gn_geometry(int hw, int c, int* pix_per_chunk, int* nchunk) {
  int ppc = 32768 / c;
  if (ppc < 16) ppc = 16;
  if (ppc > 1024) ppc = 1024;
  if (ppc > hw) ppc = hw;
  *pix_per_chunk = ppc;
  *nchunk = (hw + ppc - 1) / ppc;
}

extern "C" size_t ctta_groupnorm_scratch_floats(int batch, int hw, int c, int groups) {
  int ppc, nchunk;
  gn_geometry(hw, c, &ppc, &nchunk);
  return (size_t)batch * nchunk * groups * 2 + (size_t)batch * 2 * c;
}

extern "C" ctta_status ctta_groupnorm(const void* x, void* y, int batch, int hw, int c, int groups,
                                      const float* gamma, const float* beta, float eps, int silu,
                                      float* scratch, void* stream) {
  return ctta_groupnorm_stats_out(x, y, batch, hw, c, groups, gamma, beta, eps, silu, scratch, nullptr, stream);
}

extern "C" ctta_status ctta_groupnorm_stats_out(const void* x, void* y, int batch, int hw, int c, int groups,
                                                const float* gamma, const float* beta, float eps, int silu,
                                                float* scratch, float* stats, void* stream) {
  CTTA_REQUIRE(x && y && gamma && beta && scratch, "groupnorm: null pointer");
  CTTA_REQUIRE(c % 8 == 0 && groups > 0 && c % groups == 0, "groupnorm: C=%d groups=%d unsupported", c, groups);
  CTTA_REQUIRE(groups <= 256 * 64, "groupnorm: too many groups");
  hipStream_t s = (hipStream_t)stream;
  if (gn_small_ok(hw, c, groups)) {
    hipLaunchKernelGGL(gn_small_kernel, dim3(groups, batch), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, hw, c, groups, gamma,
                       beta, eps, silu, stats);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  }
  int ppc, nchunk;
  gn_geometry(hw, c, &ppc, &nchunk);
  float* part = scratch;
  float* ss = scratch + (size_t)batch * nchunk * groups * 2;
  const int VC = c / 8;
  const int PL = VC <= 256 ? 256 / VC : 1;
  const size_t smem = (size_t)2 * PL * c * sizeof(float);
  hipLaunchKernelGGL(gn_partial_kernel<false>, dim3(nchunk, batch), dim3(256), smem, s, (const bf16_t*)x, hw,
                     c, groups, ppc, nchunk, part, (const bf16_t*)nullptr, 0, (bf16_t*)nullptr);
  CTTA_LAUNCH_CHECK();
  if (gn_finalize_wide_on() && (long long)nchunk * groups > 2048)
    hipLaunchKernelGGL(gn_finalize_wide_kernel, dim3((groups + 3) / 4, batch), dim3(256), 0, s, part, nchunk, groups, c, hw,
                       gamma, beta, eps, ss, stats);
  else
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(batch), dim3(256), 0, s, part, nchunk, groups, c, hw,
                       gamma, beta, eps, ss, stats);
  CTTA_LAUNCH_CHECK();
  gn_launch_apply(x, y, batch, hw, c, ss, silu, false, groups, nullptr, 0, gamma, beta, eps, s);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// torch.cat([a, b], 1) of two NHWC tensors that also leaves the GroupNorm partial sums of the result in `partials`
// ([batch][*nchunk][groups][2], the layout ctta_groupnorm_from_partials reads): same chunking and summation order as the
// first pass of ctta_groupnorm, so the statistics -- and the normalised tensor -- are bit-identical to concat + GroupNorm.
extern "C" ctta_status ctta_concat_channels_gn(const void* a, int ca, const void* b, int cb, void* dst, int batch, int hw,
                                               int groups, float* partials, int64_t partials_floats, int* nchunk_out,
                                               void* stream) {
  CTTA_REQUIRE(a && b && dst && partials && nchunk_out && ca % 8 == 0 && cb % 8 == 0 && ca > 0 && cb > 0 && batch >= 1 && hw >= 1,
               "concat_channels_gn: bad arguments");
  const int c = ca + cb;
  CTTA_REQUIRE(groups > 0 && c % groups == 0, "concat_channels_gn: C=%d groups=%d unsupported", c, groups);
  int ppc = 0, nchunk = 0;
  gn_geometry(hw, c, &ppc, &nchunk);
  CTTA_REQUIRE((int64_t)batch * nchunk * groups * 2 <= partials_floats, "concat_channels_gn: partials buffer too small");
  const int VC = c / 8;
  const int PL = VC <= 256 ? 256 / VC : 1;
  const size_t smem = (size_t)2 * PL * c * sizeof(float);
  hipLaunchKernelGGL(gn_partial_kernel<true>, dim3(nchunk, batch), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)a, hw, c,
                     groups, ppc, nchunk, partials, (const bf16_t*)b, ca, (bf16_t*)dst);
  CTTA_LAUNCH_CHECK();
  *nchunk_out = nchunk;
  return CTTA_OK;
}

// GroupNorm(+SiLU) whose (sum, sum of squares) partials already exist -- written by the producing convolution's epilogue
// (ctta_conv_desc.gn_part) in the [batch][nchunk][groups][2] layout: finalize + apply only, one read of x instead of two.
// scratch: >= batch * 2 * c floats (scale / shift).
extern "C" ctta_status ctta_groupnorm_from_partials(const void* x, void* y, int batch, int hw, int c, int groups,
                                                    const float* gamma, const float* beta, float eps, int silu,
                                                    const float* partials, int nchunk, float* scratch, float* stats,
                                                    void* stream) {
  CTTA_REQUIRE(x && y && gamma && beta && partials && scratch && nchunk >= 1, "groupnorm_from_partials: null pointer");
  CTTA_REQUIRE(c % 8 == 0 && groups > 0 && c % groups == 0, "groupnorm_from_partials: C=%d groups=%d unsupported", c, groups);
  hipStream_t s = (hipStream_t)stream;
  const int VC = c / 8;
  if (!stats && (long long)nchunk * groups <= 2048 && c <= 4096) {
    gn_launch_apply(x, y, batch, hw, c, scratch, silu, true, groups, partials, nchunk, gamma, beta, eps, s);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  }
  if (gn_finalize_wide_on() && (long long)nchunk * groups > 2048)
    hipLaunchKernelGGL(gn_finalize_wide_kernel, dim3((groups + 3) / 4, batch), dim3(256), 0, s, partials, nchunk, groups, c, hw,
                       gamma, beta, eps, scratch, stats);
  else
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(batch), dim3(256), 0, s, partials, nchunk, groups, c, hw, gamma, beta, eps,
                       scratch, stats);
  CTTA_LAUNCH_CHECK();
  gn_launch_apply(x, y, batch, hw, c, scratch, silu, false, groups, nullptr, 0, gamma, beta, eps, s);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ LayerNorm
// One wave per row; the row (<= 2048 padded columns) lives in registers: two-pass variance.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x,
                                                        bf16_t* __restrict__ y, long long rows, int d,
                                                        int ld, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int VC = ld / 8;
  float f[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < VC) {
      unpack8(*reinterpret_cast<const uint4*>(x + (size_t)row * ld + v * 8), f[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (v * 8 + e >= d) f[i][e] = 0.f;
        s += f[i][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (v < VC && v * 8 + e < d) { const float t = f[i][e] - mean; q += t * t; }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < VC) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        o[e] = c < d ? (f[i][e] - mean) * rstd * gamma[c] + beta[c] : 0.f;
      }
      *reinterpret_cast<uint4*>(y + (size_t)row * ld + v * 8) = pack8(o);
    }
  }
}

// Narrow rows (ld <= 512): GL lanes per row (32 or 64), 4 rows per lane group in flight, affine parameters
// of the lane's 8 columns held in registers.  The one-wave-per-row kernel above leaves half the lanes idle and a
// single load in flight at ld = 256 (the level-0 transformer width).
template <int GL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                             long long rows, int d, int ld,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps) {
  constexpr int RPB = 256 / GL, PASSES = 4;
  const int sub = threadIdx.x % GL, slot = threadIdx.x / GL;
  const bool act = sub < ld / 8;
  float g[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = sub * 8 + e;
    g[e] = (act && c < d) ? gamma[c] : 0.f;
    bt[e] = (act && c < d) ? beta[c] : 0.f;
  }
  const long long row0 = (long long)blockIdx.x * (RPB * PASSES) + slot;
  uint4 raw[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const long long row = row0 + p * RPB;
    raw[p] = make_uint4(0, 0, 0, 0);
    if (act && row < rows) raw[p] = *reinterpret_cast<const uint4*>(x + (size_t)row * ld + sub * 8);
  }
  const float inv_d = 1.0f / (float)d;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const long long row = row0 + p * RPB;
    float f[8];
    unpack8(raw[p], f);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (sub * 8 + e >= d) f[e] = 0.f;
      s += f[e];
    }
#pragma unroll
    for (int o = 1; o < GL; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * inv_d;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (sub * 8 + e < d) { const float t = f[e] - mean; q += t * t; }
#pragma unroll
    for (int o = 1; o < GL; o <<= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * inv_d + eps);
    if (act && row < rows) {
      float o8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o8[e] = sub * 8 + e < d ? (f[e] - mean) * rstd * g[e] + bt[e] : 0.f;
      *reinterpret_cast<uint4*>(y + (size_t)row * ld + sub * 8) = pack8(o8);
    }
  }
}

// Round 3: the transformer widths of the U-Net (ld = 256 / 512 / 1024) in one kernel.  GL lanes share a row (32 / 64 / 64),
// VPL vectors of 8 channels per lane (1 / 1 / 2); a lane group walks RPG consecutive rows with the loads of ALL of them
// issued before the first reduction (round 2's kernels had 4 rows, or ONE row per wave at ld = 1024: 1.1 TB/s on the
// 17 MB level-2 tensor, a quarter of the launches' time was the launch itself); gamma / beta of the lane's columns sit in
// registers; the row sums fold on the VALU -- DPP row rotations inside 16 lanes, v_permlane16/32_swap across them --
// instead of ten ds_bpermute round trips per row.  Same two-pass variance as before.
// (dpp_row_sum / ln_group_sum: common.h -- shared with the LayerNorm-on-load of ffn_fused.hip)
template <int GL, int VPL, int RPG>
__global__ __launch_bounds__(256) void layernorm_fast_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                             long long rows, int d, int ld,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps) {
  constexpr int GPB = 256 / GL;                       // lane groups per block
  const int sub = threadIdx.x % GL, grp = threadIdx.x / GL;
  float g[VPL][8], bt[VPL][8];
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (sub + i * GL) * 8 + e;
      g[i][e] = c < d ? gamma[c] : 0.f;
      bt[i][e] = c < d ? beta[c] : 0.f;
    }
  const long long row0 = ((long long)blockIdx.x * GPB + grp) * RPG;
  uint4 raw[RPG][VPL];
#pragma unroll
  for (int r = 0; r < RPG; ++r)
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      raw[r][i] = make_uint4(0, 0, 0, 0);
      if (row0 + r < rows) raw[r][i] = *reinterpret_cast<const uint4*>(x + (size_t)(row0 + r) * ld + (sub + i * GL) * 8);
    }
  const float inv_d = 1.0f / (float)d;
#pragma unroll
  for (int r = 0; r < RPG; ++r) {
    float f[VPL][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      unpack8(raw[r][i], f[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if ((sub + i * GL) * 8 + e >= d) f[i][e] = 0.f;
        s += f[i][e];
      }
    }
    const float mean = ln_group_sum<GL>(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((sub + i * GL) * 8 + e < d) { const float t = f[i][e] - mean; q += t * t; }
    const float rstd = rsqrtf(ln_group_sum<GL>(q) * inv_d + eps);
    if (row0 + r < rows) {
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = (sub + i * GL) * 8 + e < d ? (f[i][e] - mean) * rstd * g[i][e] + bt[i][e] : 0.f;
        *reinterpret_cast<uint4*>(y + (size_t)(row0 + r) * ld + (sub + i * GL) * 8) = pack8(o8);
      }
    }
  }
}

extern "C" ctta_status ctta_layernorm(const void* x, void* y, int64_t rows, int d, int ld,
                                      const float* gamma, const float* beta, float eps, void* stream) {
  CTTA_REQUIRE(x && y && gamma && beta, "layernorm: null pointer");
  CTTA_REQUIRE(ld % 8 == 0 && d <= ld && d > 0 && ld <= 2048, "layernorm: d=%d ld=%d unsupported", d, ld);
  const dim3 grid((unsigned)cdiv64(rows, 4));
  hipStream_t s = (hipStream_t)stream;
  if ((ld == 256 || ld == 512 || ld == 1024)) {
    if (ld == 256)
      hipLaunchKernelGGL((layernorm_fast_kernel<32, 1, 8>), dim3((unsigned)cdiv64(rows, 8 * 8)), dim3(256), 0, s, (const bf16_t*)x,
                         (bf16_t*)y, (long long)rows, d, ld, gamma, beta, eps);
    else if (ld == 512)
      hipLaunchKernelGGL((layernorm_fast_kernel<64, 1, 8>), dim3((unsigned)cdiv64(rows, 4 * 8)), dim3(256), 0, s, (const bf16_t*)x,
                         (bf16_t*)y, (long long)rows, d, ld, gamma, beta, eps);
    else
      hipLaunchKernelGGL((layernorm_fast_kernel<64, 2, 4>), dim3((unsigned)cdiv64(rows, 4 * 4)), dim3(256), 0, s, (const bf16_t*)x,
                         (bf16_t*)y, (long long)rows, d, ld, gamma, beta, eps);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  }
  if (ld <= 256) {
    hipLaunchKernelGGL(layernorm_rows_kernel<32>, dim3((unsigned)cdiv64(rows, 32)), dim3(256), 0, s, (const bf16_t*)x,
                       (bf16_t*)y, (long long)rows, d, ld, gamma, beta, eps);
  } else if (ld <= 512) {
    hipLaunchKernelGGL(layernorm_rows_kernel<64>, dim3((unsigned)cdiv64(rows, 16)), dim3(256), 0, s, (const bf16_t*)x,
                       (bf16_t*)y, (long long)rows, d, ld, gamma, beta, eps);
  } else if (ld <= 1024)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (long long)rows, d, ld, gamma, beta, eps);
  else
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, (long long)rows, d, ld, gamma, beta, eps);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ GEGLU
// layout 0: [value | gate] halves (torch chunk(2), attention.py:430-432); layout 1: 16-column blocks
// [v0..15][g0..15][v16..31][g16..31]... (what the fused GEMM epilogue and the U-Net engine use)
__global__ void geglu_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, long long rows,
                             int hp, int interleaved) {
  const int vc = hp / 8;
  const long long total = rows * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    const long long r = idx / vc;
    const int va = interleaved ? (v >> 1) * 32 + (v & 1) * 8 : v * 8;
    const int vg = interleaved ? va + 16 : hp + v * 8;
    float a[8], g[8];
    unpack8(*reinterpret_cast<const uint4*>(in + (size_t)r * 2 * hp + va), a);
    unpack8(*reinterpret_cast<const uint4*>(in + (size_t)r * 2 * hp + vg), g);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      a[e] = a[e] * gelu_erf_f(g[e]);
    *reinterpret_cast<uint4*>(out + (size_t)r * hp + v * 8) = pack8(a);
  }
}

extern "C" ctta_status ctta_geglu(const void* in, void* out, int64_t rows, int hp, int interleaved, void* stream) {
  CTTA_REQUIRE(in && out && hp % 8 == 0 && (!interleaved || hp % 16 == 0), "geglu: bad arguments");
  const long long total = rows * (hp / 8);
  const int blocks = (int)fmin((double)cdiv64(total, 256), 8192.0);
  hipLaunchKernelGGL(geglu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                     (bf16_t*)out, (long long)rows, hp, interleaved);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ row softmax
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s,
                                                           bf16_t* __restrict__ p, int cols,
                                                           float scale) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* sr = s + (size_t)row * cols;
  bf16_t* pr = p + (size_t)row * cols;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(sr + c);
    mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
  float sum = 0.f;
  for (int c = tid * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(sr + c);
    sum += __expf(v.x * scale - mx) + __expf(v.y * scale - mx) + __expf(v.z * scale - mx) +
           __expf(v.w * scale - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(sr + c);
    uint2 o;
    o.x = pack2bf(__expf(v.x * scale - mx) * inv, __expf(v.y * scale - mx) * inv);
    o.y = pack2bf(__expf(v.z * scale - mx) * inv, __expf(v.w * scale - mx) * inv);
    *reinterpret_cast<uint2*>(pr + c) = o;
  }
}

// The same softmax with the row held in registers: NV float4 per thread (cols = 1024 * NV), ONE read of the fp32 scores
// (the three-pass form above re-reads the 16 KB row from L2 twice and evaluates every exponential twice).  Same operation
// order per element (max -> exp(s * scale - max * scale) -> sum -> * 1 / sum), same reduction trees: bit-identical output.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                               float scale) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  constexpr int cols = NV * 1024;
  const float* sr = s + (size_t)row * cols;
  bf16_t* pr = p + (size_t)row * cols;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(sr + tid * 4 + i * 1024);
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) mx = fmaxf(fmaxf(mx, v[i].x), fmaxf(v[i].y, fmaxf(v[i].z, v[i].w)));
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x = __expf(v[i].x * scale - mx); v[i].y = __expf(v[i].y * scale - mx);
    v[i].z = __expf(v[i].z * scale - mx); v[i].w = __expf(v[i].w * scale - mx);
    sum += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    uint2 o;
    o.x = pack2bf(v[i].x * inv, v[i].y * inv);
    o.y = pack2bf(v[i].z * inv, v[i].w * inv);
    *reinterpret_cast<uint2*>(pr + tid * 4 + i * 1024) = o;
  }
}

extern "C" ctta_status ctta_softmax_rows(const float* s, void* p, int64_t rows, int cols, float scale,
                                         void* stream) {
  CTTA_REQUIRE(s && p && cols % 4 == 0 && scale > 0.f, "softmax_rows: bad arguments");
  if ((cols == 4096 || cols == 2048 || cols == 1024)) {
    if (cols == 4096) hipLaunchKernelGGL((softmax_rows_reg_kernel<4>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, scale);
    else if (cols == 2048) hipLaunchKernelGGL((softmax_rows_reg_kernel<2>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, scale);
    else hipLaunchKernelGGL((softmax_rows_reg_kernel<1>), dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, scale);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  }
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s,
                     (bf16_t*)p, cols, scale);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ small fp32 linear
// y[m][n] = act_out(sum_k act_in(x[m][k]) w[n][k] + b[n]) for a handful of rows (the batch) and many output features
// (the concatenated time_emb_proj table: 14 336 x 1024).  One wave owns NF output features, lanes split K in float4
// chunks, rows go in chunks of RM: every x chunk loaded is used for NF features and every weight chunk for RM rows, so
// the L2 traffic per wave is (M + NF * M / RM) * K floats instead of the (8 + 1) * M / 8 * K per SINGLE feature of the
// one-feature-per-wave walk this replaces (measured there: 414 us for the table at batch 32).
template <int NF, int RM>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ b,
                                                         float* __restrict__ y, int M, int N, int K,
                                                         int act_in, int act_out) {
  const int lane = threadIdx.x & 63;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NF;
  if (n0 >= N) return;
  const bool vec = (K & 3) == 0;
  for (int m0 = 0; m0 < M; m0 += RM) {
    float acc[NF][RM];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < RM; ++r) acc[f][r] = 0.f;
    if (vec) {
      for (int k = lane * 4; k < K; k += 256) {
        float4 xv[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          xv[r] = m0 + r < M ? *reinterpret_cast<const float4*>(x + (size_t)(m0 + r) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
          if (act_in == 1) {
            xv[r].x = xv[r].x / (1.0f + expf(-xv[r].x)); xv[r].y = xv[r].y / (1.0f + expf(-xv[r].y));
            xv[r].z = xv[r].z / (1.0f + expf(-xv[r].z)); xv[r].w = xv[r].w / (1.0f + expf(-xv[r].w));
          }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const float4 wv = n0 + f < N ? *reinterpret_cast<const float4*>(w + (size_t)(n0 + f) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int r = 0; r < RM; ++r) {   // explicit fma chain: every (row, feature) sum rounds the same way whatever its
            float a = acc[f][r];            // slot in the row chunk, so a sample's result never depends on its batch mates
            a = __fmaf_rn(xv[r].x, wv.x, a); a = __fmaf_rn(xv[r].y, wv.y, a);
            a = __fmaf_rn(xv[r].z, wv.z, a); a = __fmaf_rn(xv[r].w, wv.w, a);
            acc[f][r] = a;
          }
        }
      }
    } else {
      for (int k = lane; k < K; k += 64) {
        float xv[RM];
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          xv[r] = m0 + r < M ? x[(size_t)(m0 + r) * K + k] : 0.f;
          if (act_in == 1) xv[r] = xv[r] / (1.0f + expf(-xv[r]));
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const float wv = n0 + f < N ? w[(size_t)(n0 + f) * K + k] : 0.f;
#pragma unroll
          for (int r = 0; r < RM; ++r) acc[f][r] = __fmaf_rn(xv[r], wv, acc[f][r]);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const float t = wave_sum(acc[f][r]);
        if (lane == 0 && m0 + r < M && n0 + f < N) {
          float o = t + (b ? b[n0 + f] : 0.f);
          if (act_out == 1) o = o / (1.0f + expf(-o));
          y[(size_t)(m0 + r) * N + n0 + f] = o;
        }
      }
  }
}

extern "C" ctta_status ctta_linear_f32(const float* x, const float* w, const float* b, float* y, int m,
                                       int n, int k, int act_in, int act_out, void* stream) {
  CTTA_REQUIRE(x && w && y && m > 0 && m <= 1024 && n > 0 && k > 0, "linear_f32: bad arguments");
  if (n >= 2048) {   // wide tables: 4 features per wave
    hipLaunchKernelGGL((linear_f32_kernel<4, 8>), dim3((n + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, w, b, y, m, n, k,
                       act_in, act_out);
  } else {           // narrow layers (the 1024-wide embedding MLPs): one feature per wave keeps the chip busy
    hipLaunchKernelGGL((linear_f32_kernel<1, 8>), dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, b, y, m, n, k,
                       act_in, act_out);
  }
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ embeddings
// get_timestep_embedding: emb = t * freq (fp32), [sin | cos], optionally flipped to [cos | sin].
__global__ void time_features_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                     int dim, int flip, float* __restrict__ out, int B) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx - b * half;
  const float arg = t[b] * freqs[i];
  const float sv = sinf(arg), cv = cosf(arg);
  float* o = out + (size_t)b * dim;
  if (flip) { o[i] = cv; o[half + i] = sv; }
  else { o[i] = sv; o[half + i] = cv; }
}

// GaussianFourierProjection: x = w * W * 2*pi evaluated in double (the reference uses fp64 when
// guidance is a Python float and fp32 when it is a tensor; fp64 is within 1e-5 of both).
__global__ void fourier_features_kernel(const double* __restrict__ wv, const float* __restrict__ weight,
                                        int half, int flip, float* __restrict__ out, int B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, i = idx - b * half;
  const double arg = wv[b] * (double)weight[i] * 2.0 * 3.14159265358979323846;
  const float sv = (float)sin(arg), cv = (float)cos(arg);
  float* o = out + (size_t)b * 2 * half;
  if (flip) { o[i] = cv; o[half + i] = sv; }
  else { o[i] = sv; o[half + i] = cv; }
}

extern "C" ctta_status ctta_time_features(const float* t, const float* freqs, int dim, int flip,
                                          float* out, int batch, void* stream) {
  CTTA_REQUIRE(t && freqs && out && dim % 2 == 0, "time_features: bad arguments");
  const int total = batch * (dim / 2);
  hipLaunchKernelGGL(time_features_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     t, freqs, dim, flip, out, batch);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_fourier_features(const double* w, const float* weight, int half, int flip,
                                             float* out, int batch, void* stream) {
  CTTA_REQUIRE(w && weight && out && half > 0, "fourier_features: bad arguments");
  const int total = batch * half;
  hipLaunchKernelGGL(fourier_features_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, w, weight, half, flip, out, batch);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ C wrappers
static int grid_for(long long total) { return (int)fmin((double)cdiv64(total, 256), 16384.0); }

extern "C" ctta_status ctta_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int batch, int c, int h,
                                                  int w, int c_pad, float scale, void* stream) {
  CTTA_REQUIRE(src && dst && c_pad % 8 == 0 && c_pad >= c, "nchw->nhwc: bad arguments");
  const long long total = (long long)batch * h * w * (c_pad / 8);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, batch, c, h * w, c_pad, scale);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int batch, int c, int h,
                                                  int w, int c_stride, void* stream) {
  CTTA_REQUIRE(src && dst && c_stride >= c, "nhwc->nchw: bad arguments");
  const long long total = (long long)batch * c * h * w;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, dst, batch, c, h * w, c_stride);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_rows_f32_to_bf16(const float* src, void* dst, int64_t rows, int cols,
                                             int cols_pad, void* stream) {
  CTTA_REQUIRE(src && dst && cols_pad % 8 == 0 && cols_pad >= cols, "rows f32->bf16: bad arguments");
  const long long total = rows * (cols_pad / 8);
  hipLaunchKernelGGL(rows_f32_to_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     src, (bf16_t*)dst, (long long)rows, cols, cols_pad);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_concat_channels(const void* a, int ca, const void* b, int cb, void* dst,
                                            int64_t pixels, void* stream) {
  CTTA_REQUIRE(a && b && dst && ca % 8 == 0 && cb % 8 == 0, "concat: bad arguments");
  const long long total = pixels * ((ca + cb) / 8);
  hipLaunchKernelGGL(concat_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)a, ca / 8, (const uint4*)b, cb / 8, (uint4*)dst, (long long)pixels);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_pack_weight(const float* src, const int32_t* row_off, const int32_t* col_off,
                                        const int32_t* row_aux, const int32_t* col_aux, int aux_limit,
                                        int n_rows, int k_pad, void* dst, void* stream) {
  CTTA_REQUIRE(src && row_off && col_off && dst, "pack_weight: null pointer");
  CTTA_REQUIRE(aux_limit <= 0 || (row_aux && col_aux), "pack_weight: aux arrays missing");
  const long long total = (long long)n_rows * k_pad;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src,
                     row_off, col_off, row_aux, col_aux, aux_limit, n_rows, k_pad, (bf16_t*)dst);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
