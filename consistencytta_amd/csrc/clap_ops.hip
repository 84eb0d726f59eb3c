// Glue kernels of the CLAP audio tower (CLAP fine-tuning stage, tools/losses.py:259-316): everything around the
// GEMMs / attention / LayerNorm of the HTSAT Swin transformer (laion_clap/clap_module/htsat.py) that is not a
// contraction.  All HBM-bound elementwise / gather work; the contractions run on ctta_conv_gemm and the flash
// attention kernels.
//   ctta_resample_poly / _bwd   torchaudio.functional.resample (sinc_interp_kaiser) as a polyphase FIR, fp32
//   ctta_gather_rows            row permutation of a token matrix: window partition + cyclic shift + their inverses,
//                               the 2x2 patch-merging concat (htsat.py:259-287,471-492,517-537)
//   ctta_gelu / _bwd            nn.GELU (erf) of the Swin MLP (htsat.py:156-174)
//   ctta_mean_tokens / _bwd     AdaptiveAvgPool1d over the final token grid (htsat.py:818-819)
//   ctta_htsat_image / _bwd     bn0 (eval) + bicubic stretch of the frame axis + fold into the square "image"
//                               (htsat.py:913-925,856-878), written as the NHWC bf16 input of the patch-embedding conv
#include "common.h"

// ------------------------------------------------------------------------------------------ polyphase resampler
// y[b][n*up + p] = sum_j kern[p][j] * x[b][n*down + j - width]   (zero outside [0, len)); out_len outputs per row
__global__ __launch_bounds__(256) void resample_fwd_kernel(const float* __restrict__ x, int len, const float* __restrict__ kern,
                                                           int up, int down, int width, int taps, float* __restrict__ y,
                                                           long long out_len) {
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * len;
  for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < out_len; o += (long long)gridDim.x * blockDim.x) {
    const long long n = o / up;
    const int p = (int)(o - n * up);
    const float* kp = kern + (size_t)p * taps;
    const long long base = n * down - width;
    float acc = 0.f;
    for (int j = 0; j < taps; ++j) {
      const long long i = base + j;
      if (i >= 0 && i < len) acc += kp[j] * xb[i];
    }
    y[(size_t)b * out_len + o] = acc;
  }
}
// dx[b][i] = sum_{n, p, j : n*down + j - width == i} kern[p][j] * dy[b][n*up + p]
__global__ __launch_bounds__(256) void resample_bwd_kernel(const float* __restrict__ dy, long long out_len,
                                                           const float* __restrict__ kern, int up, int down, int width,
                                                           int taps, float* __restrict__ dx, int len) {
  const int b = blockIdx.y;
  const float* db = dy + (size_t)b * out_len;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
    // j = i + width - n*down must lie in [0, taps)
    const long long hi = ((long long)i + width) / down;                   // largest n with j >= 0
    long long lo = ((long long)i + width - taps + down) / down;           // smallest n with j <= taps - 1 (ceil)
    if (lo < 0) lo = 0;
    float acc = 0.f;
    for (long long n = lo; n <= hi; ++n) {
      const int j = (int)((long long)i + width - n * down);
      if (j < 0 || j >= taps) continue;
      for (int p = 0; p < up; ++p) {
        const long long o = n * up + p;
        if (o < out_len) acc += kern[(size_t)p * taps + j] * db[o];
      }
    }
    dx[(size_t)b * len + i] = acc;
  }
}
extern "C" ctta_status ctta_resample_poly(const float* x, int batch, int len, const float* kernels, int up, int down,
                                          int width, int taps, float* y, int64_t out_len, void* stream) {
  CTTA_REQUIRE(x && kernels && y && batch >= 1 && len >= 1 && up >= 1 && down >= 1 && taps >= 1 && out_len >= 1,
               "resample_poly: bad arguments");
  int blocks = (int)((out_len + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(resample_fwd_kernel, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, x, len, kernels, up, down,
                     width, taps, y, (long long)out_len);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_resample_poly_bwd(const float* dy, int batch, int64_t out_len, const float* kernels, int up,
                                              int down, int width, int taps, float* dx, int len, void* stream) {
  CTTA_REQUIRE(dy && kernels && dx && batch >= 1 && len >= 1 && up >= 1 && down >= 1 && taps >= 1 && out_len >= 1,
               "resample_poly_bwd: bad arguments");
  int blocks = (len + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(resample_bwd_kernel, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, dy, (long long)out_len,
                     kernels, up, down, width, taps, dx, len);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------------------ row gather
// dst[r][:] = src[idx[r]][:] (idx < 0: zeros); rows of row_elems bf16 (multiple of 8), 16-byte chunks
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int32_t* __restrict__ idx,
                                                          uint4* __restrict__ dst, long long n_rows, int chunks,
                                                          long long src_ld_chunks, long long dst_ld_chunks) {
  const long long total = n_rows * chunks;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / chunks;
    const int c = (int)(i - r * chunks);
    const int s = idx[r];
    dst[r * dst_ld_chunks + c] = s >= 0 ? src[(long long)s * src_ld_chunks + c] : make_uint4(0, 0, 0, 0);
  }
}
extern "C" ctta_status ctta_gather_rows(const void* src, int src_ld, const int32_t* idx, void* dst, int dst_ld,
                                        int64_t n_rows, int row_elems, void* stream) {
  CTTA_REQUIRE(src && idx && dst && n_rows >= 1, "gather_rows: null pointer");
  CTTA_REQUIRE(row_elems >= 8 && row_elems % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0 && src_ld >= row_elems &&
                   dst_ld >= row_elems, "gather_rows: row_elems / strides must be multiples of 8");
  const long long total = n_rows * (row_elems / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, idx, (uint4*)dst,
                     (long long)n_rows, row_elems / 8, (long long)(src_ld / 8), (long long)(dst_ld / 8));
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------------------ GELU (erf)
__global__ __launch_bounds__(256) void gelu_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long long n8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(x[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = gelu_erf_f(f[e]);
    y[i] = pack8(f);
  }
}
// dx = dy * (Phi(x) + x * phi(x)),  Phi = standard normal cdf, phi = pdf
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy,
                                                       uint4* __restrict__ dx, long long n8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8(x[i], f);
    unpack8(dy[i], g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float cdf, pdf;
      gelu_cdf_pdf(f[e], cdf, pdf);
      g[e] *= cdf + f[e] * pdf;
    }
    dx[i] = pack8(g);
  }
}
extern "C" ctta_status ctta_gelu(const void* x, void* y, int64_t n, void* stream) {
  CTTA_REQUIRE(x && y && n >= 8 && n % 8 == 0, "gelu: n=%lld must be a positive multiple of 8", (long long)n);
  int blocks = (int)((n / 8 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)y, (long long)(n / 8));
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  CTTA_REQUIRE(x && dy && dx && n >= 8 && n % 8 == 0, "gelu_bwd: n=%lld must be a positive multiple of 8", (long long)n);
  int blocks = (int)((n / 8 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)dy,
                     (uint4*)dx, (long long)(n / 8));
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------------------ token mean
// x bf16 [B][tokens][C] -> y fp32 [B][C] ; backward: dx[b][t][c] = dy[b][c] / tokens
__global__ __launch_bounds__(256) void mean_tokens_kernel(const bf16_t* __restrict__ x, int tokens, int C, int ld,
                                                          float* __restrict__ y) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int t = 0; t < tokens; ++t) acc += bf2f(x[((size_t)b * tokens + t) * ld + c]);
  y[(size_t)b * C + c] = acc / (float)tokens;
}
__global__ __launch_bounds__(256) void mean_tokens_bwd_kernel(const float* __restrict__ dy, int tokens, int C, int ld,
                                                              bf16_t* __restrict__ dx) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const bf16_t v = f2bf(dy[(size_t)b * C + c] / (float)tokens);
  for (int t = 0; t < tokens; ++t) dx[((size_t)b * tokens + t) * ld + c] = v;
}
extern "C" ctta_status ctta_mean_tokens(const void* x, int batch, int tokens, int channels, int ld, float* y, void* stream) {
  CTTA_REQUIRE(x && y && batch >= 1 && tokens >= 1 && channels >= 1 && ld >= channels, "mean_tokens: bad arguments");
  hipLaunchKernelGGL(mean_tokens_kernel, dim3((channels + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, tokens, channels, ld, y);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_mean_tokens_bwd(const float* dy, int batch, int tokens, int channels, int ld, void* dx,
                                            void* stream) {
  CTTA_REQUIRE(dy && dx && batch >= 1 && tokens >= 1 && channels >= 1 && ld >= channels, "mean_tokens_bwd: bad arguments");
  hipLaunchKernelGGL(mean_tokens_bwd_kernel, dim3((channels + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, dy,
                     tokens, channels, ld, (bf16_t*)dx);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------------------ log-mel -> "image"
// logmel fp32 [B][T][F]; per-mel-bin affine a[f] * v + c[f] (bn0 in eval mode); bicubic stretch T -> TT frames with
// host-computed taps (tap_idx / tap_w [TT][4], PyTorch's align_corners=True cubic convolution, A = -0.75, border
// indices clamped); fold: image row = (tt / S) * F + f, column = tt % S with S = TT / ratio (= spec_size).
// Output NHWC bf16 [B][S][S][cpad], channel 0 = the value, the pad channels zero.
__global__ __launch_bounds__(256) void htsat_image_kernel(const float* __restrict__ lm, int T, int F, const float* __restrict__ a,
                                                          const float* __restrict__ c, const int32_t* __restrict__ tap_idx,
                                                          const float* __restrict__ tap_w, int TT, int S, int cpad,
                                                          bf16_t* __restrict__ img) {
  const int b = blockIdx.y;
  const long long total = (long long)TT * F;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int tt = (int)(i / F), f = (int)(i - (long long)tt * F);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += tap_w[tt * 4 + k] * (a[f] * lm[((size_t)b * T + tap_idx[tt * 4 + k]) * F + f] + c[f]);
    const int row = (tt / S) * F + f, col = tt % S;
    bf16_t* o = img + (((size_t)b * S + row) * S + col) * cpad;
    o[0] = f2bf(v);
    for (int e = 1; e < cpad; ++e) o[e] = 0;
  }
}
// dlm[b][t][f] = a[f] * sum over (tt, k) with tap_idx[tt][k] == t of tap_w[tt][k] * dimg[b][row(tt, f)][col(tt)]
// (host passes the transposed tap list in CSR form: for frame t the entries rt_ptr[t] .. rt_ptr[t+1]).
// dimg is the patch-embedding conv's data gradient in TOKEN layout: [B][(S/ps)^2 patches][ld >= ps*ps] fp32, pixel
// (row, col) of the image at patch (row/ps, col/ps), entry (row%ps)*ps + col%ps.
__global__ __launch_bounds__(256) void htsat_image_bwd_kernel(const float* __restrict__ dtok, int ld, int ps, int T, int F,
                                                              const float* __restrict__ a, const int32_t* __restrict__ rt_ptr,
                                                              const int32_t* __restrict__ rt_tt, const float* __restrict__ rt_w,
                                                              int S, float* __restrict__ dlm) {
  const int b = blockIdx.y;
  const int G = S / ps;
  const long long total = (long long)T * F;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i / F), f = (int)(i - (long long)t * F);
    float acc = 0.f;
    for (int e = rt_ptr[t]; e < rt_ptr[t + 1]; ++e) {
      const int tt = rt_tt[e];
      const int row = (tt / S) * F + f, col = tt % S;
      acc += rt_w[e] * dtok[(((size_t)b * G + row / ps) * G + col / ps) * ld + (row % ps) * ps + col % ps];
    }
    dlm[((size_t)b * T + t) * F + f] = a[f] * acc;
  }
}
extern "C" ctta_status ctta_htsat_image(const float* logmel, int batch, int frames, int mel_bins, const float* bn_scale,
                                        const float* bn_shift, const int32_t* tap_idx, const float* tap_w, int out_frames,
                                        int spec_size, int cpad, void* image_nhwc, void* stream) {
  CTTA_REQUIRE(logmel && bn_scale && bn_shift && tap_idx && tap_w && image_nhwc, "htsat_image: null pointer");
  CTTA_REQUIRE(out_frames % spec_size == 0 && (out_frames / spec_size) * mel_bins == spec_size && cpad >= 1,
               "htsat_image: out_frames=%d, mel_bins=%d do not fold into a %d x %d image", out_frames, mel_bins, spec_size,
               spec_size);
  const long long total = (long long)out_frames * mel_bins;
  hipLaunchKernelGGL(htsat_image_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream, logmel,
                     frames, mel_bins, bn_scale, bn_shift, tap_idx, tap_w, out_frames, spec_size, cpad, (bf16_t*)image_nhwc);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_htsat_image_bwd(const float* dtokens, int dtokens_ld, int patch, int batch, int frames,
                                            int mel_bins, const float* bn_scale, const int32_t* rt_ptr, const int32_t* rt_tt,
                                            const float* rt_w, int spec_size, float* dlogmel, void* stream) {
  CTTA_REQUIRE(dtokens && bn_scale && rt_ptr && rt_tt && rt_w && dlogmel, "htsat_image_bwd: null pointer");
  CTTA_REQUIRE(patch >= 1 && spec_size % patch == 0 && dtokens_ld >= patch * patch, "htsat_image_bwd: bad patch geometry");
  const long long total = (long long)frames * mel_bins;
  hipLaunchKernelGGL(htsat_image_bwd_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream,
                     dtokens, dtokens_ld, patch, frames, mel_bins, bn_scale, rt_ptr, rt_tt, rt_w, spec_size, dlogmel);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
