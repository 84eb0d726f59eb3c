// Glue kernels of the evaluation suite's classifier (audioldm_eval/feature_extractors/panns/models.py:168-323, PANNs
// Cnn14): everything around its twelve 3x3 convolutions that is not a contraction.  The convolutions (BatchNorm folded into
// weights and bias, ReLU in the epilogue) run on ctta_conv_gemm, the power-STFT / log-mel front end on ctta_wav_to_logmel_db,
// the two fully connected layers on ctta_linear_f32.  All HBM-bound elementwise / reduction work on tensors of a few MB.
//   ctta_logmel_to_image   bn0 over the mel axis (models.py:276-278, eval mode) + cast: fp32 log-mel -> the NHWC bf16 input
//                          of conv_block1 with its single channel padded to 8
//   ctta_avgpool2          F.avg_pool2d(kernel 2, stride 2; a trailing odd row / column is dropped) on NHWC bf16
//   ctta_cnn14_head        torch.mean(x, dim=3) over frequency, then max + mean over time (models.py:305-309) -> fp32
#include "common.h"

__global__ __launch_bounds__(256) void logmel_to_image_kernel(const float* __restrict__ lm, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int F, long long total,
                                                              uint4* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    float v[8] = {lm[i] * scale[f] + shift[f], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    out[i] = pack8(v);
  }
}

extern "C" ctta_status ctta_logmel_to_image(const float* logmel, int batch, int frames, int mel_bins, const float* scale,
                                            const float* shift, void* image, void* stream) {
  CTTA_REQUIRE(logmel && scale && shift && image && batch >= 1 && frames >= 1 && mel_bins >= 1, "logmel_to_image: bad arguments");
  const long long total = (long long)batch * frames * mel_bins;
  long long blocks = cdiv64(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(logmel_to_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logmel, scale, shift,
                     mel_bins, total, (uint4*)image);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

__global__ __launch_bounds__(256) void avgpool2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int hi,
                                                       int wi, int C) {
  const int ho = hi / 2, wo = wi / 2, vc = C / 8;
  const long long total = (long long)B * ho * wo * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    long long p = idx / vc;
    const int ox = (int)(p % wo); p /= wo;
    const int oy = (int)(p % ho);
    const int b = (int)(p / ho);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + ((((size_t)b * hi + 2 * oy + dy) * wi) + 2 * ox + dx) * C + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] *= 0.25f;
    *reinterpret_cast<uint4*>(y + (size_t)idx * 8) = pack8(s);
  }
}

extern "C" ctta_status ctta_avgpool2(const void* x, void* y, int batch, int hi, int wi, int c, void* stream) {
  CTTA_REQUIRE(x && y && batch >= 1 && hi >= 2 && wi >= 2 && c >= 8 && c % 8 == 0,
               "avgpool2: bad arguments (hi=%d wi=%d c=%d: the channel count must be a multiple of 8)", hi, wi, c);
  const long long total = (long long)batch * (hi / 2) * (wi / 2) * (c / 8);
  long long blocks = cdiv64(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y,
                     batch, hi, wi, c);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// one thread per (sample, channel): T x F values (31 x 2 for a 10 s clip), channel-contiguous reads across the wave
__global__ __launch_bounds__(256) void cnn14_head_kernel(const bf16_t* __restrict__ x, int T, int F, int C,
                                                         float* __restrict__ y) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  const bf16_t* xb = x + (size_t)b * T * F * C + c;
  float mx = -INFINITY, sum = 0.f;
  for (int t = 0; t < T; ++t) {
    float m = 0.f;
    for (int f = 0; f < F; ++f) m += __uint_as_float((unsigned)xb[((size_t)t * F + f) * C] << 16);
    m /= (float)F;
    mx = fmaxf(mx, m);
    sum += m;
  }
  y[(size_t)b * C + c] = mx + sum / (float)T;
}

extern "C" ctta_status ctta_cnn14_head(const void* x, int batch, int frames, int freq, int c, float* y, void* stream) {
  CTTA_REQUIRE(x && y && batch >= 1 && frames >= 1 && freq >= 1 && c >= 1, "cnn14_head: bad arguments");
  hipLaunchKernelGGL(cnn14_head_kernel, dim3((c + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     frames, freq, c, y);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
