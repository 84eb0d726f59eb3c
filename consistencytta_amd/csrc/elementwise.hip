// Streaming fp32 kernels: Heun solver halves, CFG combine, SNR-weighted instance MSE, fused
// two-shadow EMA, vocoder post-processing, and the direct convolution used for the three
// output layers whose Cout is 1 or 8 (an MFMA tile would be >87 % padding there).
// All are HBM-bound: float4 per lane, grid-stride.
#include "common.h"

#include <math.h>

static int grid_for(long long total_vec) {
  long long b = (total_vec + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

#define VEC_LOOP(total)                                                                   \
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (total);       \
       i += (long long)gridDim.x * blockDim.x)

// x / sqrt(sigma^2 + 1)        scheduling_heun_discrete.py:170-171
__global__ void heun_scale_kernel(const float4* __restrict__ x, const float* __restrict__ sigma,
                                  float4* __restrict__ out, long long nvec_per, long long total) {
  VEC_LOOP(total) {
    const int b = (int)(i / nvec_per);
    const float s = sigma[b];
    const float d = sqrtf(s * s + 1.0f);
    float4 v = x[i];
    v.x /= d; v.y /= d; v.z /= d; v.w /= d;
    out[i] = v;
  }
}

// x0 + noise * sigma            :384
__global__ void heun_add_noise_kernel(const float4* __restrict__ x0, const float4* __restrict__ nz,
                                      const float* __restrict__ sigma, float4* __restrict__ out,
                                      long long nvec_per, long long total) {
  VEC_LOOP(total) {
    const float s = sigma[(int)(i / nvec_per)];
    const float4 a = x0[i], n = nz[i];
    out[i] = make_float4(a.x + n.x * s, a.y + n.y * s, a.z + n.z * s, a.w + n.w * s);
  }
}

__device__ __forceinline__ float pred_x0(float x, float v, float alpha_prod, float c2) {
  return x * alpha_prod - v * c2;   // :319-323 (v_prediction)
}

// 1st-order half   :332-341,357
__global__ void heun_first_kernel(const float4* __restrict__ v, const float4* __restrict__ x,
                                  const float* __restrict__ sigma, const float* __restrict__ sigma_next,
                                  float4* __restrict__ prev, float4* __restrict__ deriv,
                                  long long nvec_per, long long total) {
  VEC_LOOP(total) {
    const int b = (int)(i / nvec_per);
    const float s = sigma[b], sn = sigma_next[b];
    const float ap = 1.0f / (s * s + 1.0f);
    const float c2 = s * sqrtf(ap);
    const float dt = sn - s;
    const float4 vv = v[i], xx = x[i];
    float4 d, p;
    d.x = (xx.x - pred_x0(xx.x, vv.x, ap, c2)) / s; p.x = xx.x + d.x * dt;
    d.y = (xx.y - pred_x0(xx.y, vv.y, ap, c2)) / s; p.y = xx.y + d.y * dt;
    d.z = (xx.z - pred_x0(xx.z, vv.z, ap, c2)) / s; p.z = xx.z + d.z * dt;
    d.w = (xx.w - pred_x0(xx.w, vv.w, ap, c2)) / s; p.w = xx.w + d.w * dt;
    deriv[i] = d;
    prev[i] = p;
  }
}

// 2nd-order half   :343-357 -- derivative at the predicted point, averaged with the stored one,
// applied to the STORED sample with the stored dt.
__global__ void heun_second_kernel(const float4* __restrict__ v, const float4* __restrict__ xh,
                                   const float4* __restrict__ xs, const float4* __restrict__ dprev,
                                   const float* __restrict__ sigma, const float* __restrict__ sigma_next,
                                   float4* __restrict__ prev, long long nvec_per, long long total) {
  VEC_LOOP(total) {
    const int b = (int)(i / nvec_per);
    const float s = sigma[b], sn = sigma_next[b];
    const float ap = 1.0f / (sn * sn + 1.0f);
    const float c2 = sn * sqrtf(ap);
    const float dt = sn - s;
    const float4 vv = v[i], xx = xh[i], x0 = xs[i], dp = dprev[i];
    float4 p;
    p.x = x0.x + ((dp.x + (xx.x - pred_x0(xx.x, vv.x, ap, c2)) / sn) / 2.0f) * dt;
    p.y = x0.y + ((dp.y + (xx.y - pred_x0(xx.y, vv.y, ap, c2)) / sn) / 2.0f) * dt;
    p.z = x0.z + ((dp.z + (xx.z - pred_x0(xx.z, vv.z, ap, c2)) / sn) / 2.0f) * dt;
    p.w = x0.w + ((dp.w + (xx.w - pred_x0(xx.w, vv.w, ap, c2)) / sn) / 2.0f) * dt;
    prev[i] = p;
  }
}

__global__ void cfg_combine_kernel(const float4* __restrict__ u, const float4* __restrict__ c,
                                   const float* __restrict__ w, float4* __restrict__ out,
                                   long long nvec_per, long long total) {
  VEC_LOOP(total) {
    const float ww = w[(int)(i / nvec_per)];
    const float a = 1.0f - ww;
    const float4 uu = u[i], cc = c[i];
    out[i] = make_float4(a * uu.x + ww * cc.x, a * uu.y + ww * cc.y, a * uu.z + ww * cc.z,
                         a * uu.w + ww * cc.w);
  }
}

// out = a[b]*x + b_[b]*y per sample (optionally clamped): DDPM/DDIM add_noise and the v-prediction step pieces
__global__ void lincomb2_kernel(const float4* __restrict__ x, const float4* __restrict__ y, const float* __restrict__ a,
                                const float* __restrict__ b_, float4* __restrict__ out, long long nvec_per, long long total,
                                float clamp) {
  VEC_LOOP(total) {
    const int s = (int)(i / nvec_per);
    const float ca = a[s], cb = b_[s];
    const float4 xx = x[i], yy = y[i];
    float4 o = make_float4(ca * xx.x + cb * yy.x, ca * xx.y + cb * yy.y, ca * xx.z + cb * yy.z, ca * xx.w + cb * yy.w);
    if (clamp > 0.f) {
      o.x = fminf(fmaxf(o.x, -clamp), clamp); o.y = fminf(fmaxf(o.y, -clamp), clamp);
      o.z = fminf(fmaxf(o.z, -clamp), clamp); o.w = fminf(fmaxf(o.w, -clamp), clamp);
    }
    out[i] = o;
  }
}
// loss = mean_b(w[b] * inst[b])  (w NULL = 1)
__global__ void weighted_mean_kernel(const float* __restrict__ inst, const float* __restrict__ w, int B,
                                     float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += inst[b] * (w ? w[b] : 1.0f);
  loss[0] = s / (float)B;
}

// per-instance mean squared error (one workgroup per instance), then the SNR-clamped mean
__global__ __launch_bounds__(256) void inst_mse_kernel(const float4* __restrict__ a,
                                                       const float4* __restrict__ t,
                                                       long long nvec_per, float* __restrict__ inst) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (long long i = threadIdx.x; i < nvec_per; i += 256) {
    const float4 x = a[b * nvec_per + i], y = t[b * nvec_per + i];
    const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
    acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) inst[b] = (red[0] + red[1] + red[2] + red[3]) / (float)(nvec_per * 4);
}

__global__ void snr_mean_kernel(const float* __restrict__ inst, const float* __restrict__ sigma,
                                float gamma, int B, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    float w = 1.0f;
    if (gamma > 0.f) {
      const float sg = sigma[b];
      w = fminf(1.0f / (sg * sg), gamma);   // sigma = 0 -> inf -> clamp
    }
    s += inst[b] * w;
  }
  loss[0] = s / (float)B;
}

// shadow += (1-decay)*(param-shadow) for two shadows in one pass over param.  Two vectors in flight per thread on a grid of
// <= 2048 workgroups, the second shadow a template flag: the form adamw4_kernel reaches 0.70 of the HBM peak with (the
// one-vector loop with a per-element `if (sb)` sat at 0.58 for five rounds).  Same arithmetic per element.
template <bool TWO>
__global__ __launch_bounds__(256) void ema2_kernel(const float4* __restrict__ p, float4* __restrict__ sa, float ka,
                                                   float4* __restrict__ sb, float kb, long long nvec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += 2 * stride) {
    const long long j = i + stride;
    const bool two = j < nvec;
    const float4 p0 = p[i];
    float4 a0 = sa[i], c0, p1, a1, c1;
    if (TWO) c0 = sb[i];
    if (two) { p1 = p[j]; a1 = sa[j]; if (TWO) c1 = sb[j]; }
    a0.x += ka * (p0.x - a0.x); a0.y += ka * (p0.y - a0.y); a0.z += ka * (p0.z - a0.z); a0.w += ka * (p0.w - a0.w);
    sa[i] = a0;
    if (TWO) {
      c0.x += kb * (p0.x - c0.x); c0.y += kb * (p0.y - c0.y); c0.z += kb * (p0.z - c0.z); c0.w += kb * (p0.w - c0.w);
      sb[i] = c0;
    }
    if (two) {
      a1.x += ka * (p1.x - a1.x); a1.y += ka * (p1.y - a1.y); a1.z += ka * (p1.z - a1.z); a1.w += ka * (p1.w - a1.w);
      sa[j] = a1;
      if (TWO) {
        c1.x += kb * (p1.x - c1.x); c1.y += kb * (p1.y - c1.y); c1.z += kb * (p1.z - c1.z); c1.w += kb * (p1.w - c1.w);
        sb[j] = c1;
      }
    }
  }
}
__global__ void ema2_tail_kernel(const float* __restrict__ p, float* __restrict__ sa, float ka,
                                 float* __restrict__ sb, float kb, long long start, long long n) {
  const long long i = start + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  sa[i] += ka * (p[i] - sa[i]);
  if (sb) sb[i] += kb * (p[i] - sb[i]);
}

// vocoder_infer centring: batch-global max / min  (utilities.py:85)
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, long long n,
                                                     unsigned int* __restrict__ mm) {
  // one atomic pair per BLOCK of a capped grid (every wave of 5 121 blocks hammering the same two words cost 0.47 ms at
  // batch 32: 40 000 serialised L2 atomics); float4 loads over the 16-byte-aligned body, scalars for the tail
  __shared__ float smx[4], smn[4];
  float mx = -INFINITY, mn = INFINITY;
  const long long n4 = (reinterpret_cast<size_t>(x) & 15) == 0 ? n >> 2 : 0;   // unaligned base: everything is "tail"
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    mn = fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w));
  }
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  mx = wave_max(mx);
  mn = -wave_max(-mn);
  if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    // order-preserving float <-> uint encoding so integer atomics give float max/min
    auto enc = [](float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    atomicMax(&mm[0], enc(mx));
    atomicMin(&mm[1], enc(mn));
  }
}
__global__ void minmax_init_kernel(unsigned int* mm) { mm[0] = 0u; mm[1] = 0xffffffffu; }

__global__ void wav_finalize_kernel(const float* __restrict__ x, long long n,
                                    const unsigned int* __restrict__ mm, float* __restrict__ centred,
                                    int16_t* __restrict__ pcm) {
  auto dec = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
  const float shift = (dec(mm[0]) + dec(mm[1])) / 2.0f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i] - shift;
    if (centred) centred[i] = v;
    if (pcm) {
      // numpy float32 -> int16 astype: truncate toward zero (values stay inside int32 range)
      const int q = (int)(v * 32768.0f);
      pcm[i] = (int16_t)q;
    }
  }
}

// ------------------------------------------------------------------------------ direct conv
// Cout <= 8, stride 1, dilation 1; weights fp32 [n][kh][kw][c].  LPP lanes share one output pixel: lane `sub` takes the
// 8-channel chunks sub, sub + LPP, ... of every tap (neighbouring lanes read neighbouring 16-byte chunks) and the partial
// sums meet in a shuffle reduction.  LPP = 1 is the plain one-thread-per-pixel walk (millions of pixels: enough threads
// anyway); LPP = 8 serves the U-Net's conv_out, whose 36 864 .. 131 072 pixels x 2304-long dot products would otherwise
// run as a few hundred long serial loops per CU.
template <int N, int LPP>
__global__ __launch_bounds__(256) void conv_small_n_kernel(
    const bf16_t* __restrict__ x, int C, int B, int H, int W, int KH, int KW, int ph, int pw,
    const float* __restrict__ w, const float* __restrict__ bias, int in_act, float in_slope,
    int out_act, float* __restrict__ out, bf16_t* __restrict__ out_bf) {
  const long long total = (long long)B * H * W;
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long m = gid / LPP;
  const int sub = (int)(gid % LPP);
  const bool live = m < total;
  const long long mm = live ? m : total - 1;      // idle lanes of the last block still take part in the shuffles
  const int b = (int)(mm / ((long long)H * W));
  const int rem = (int)(mm - (long long)b * H * W);
  const int oh = rem / W, ow = rem - oh * W;
  float acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = (bias && sub == 0) ? bias[n] : 0.f;
  const int K = KH * KW * C;
  for (int kh = 0; kh < KH; ++kh) {
    const int ih = oh - ph + kh;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int kw = 0; kw < KW; ++kw) {
      const int iw = ow - pw + kw;
      if ((unsigned)iw >= (unsigned)W) continue;
      const bf16_t* xp = x + ((size_t)((size_t)b * H + ih) * W + iw) * C;
      const float* wp = w + (size_t)(kh * KW + kw) * C;
      for (int c = sub * 8; c < C; c += 8 * LPP) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(xp + c), f);
        if (in_act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * in_slope;
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const float4 w0 = *reinterpret_cast<const float4*>(wp + (size_t)n * K + c);
          const float4 w1 = *reinterpret_cast<const float4*>(wp + (size_t)n * K + c + 4);
          acc[n] += f[0] * w0.x + f[1] * w0.y + f[2] * w0.z + f[3] * w0.w + f[4] * w1.x +
                    f[5] * w1.y + f[6] * w1.z + f[7] * w1.w;
        }
      }
    }
  }
  if (LPP > 1) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) acc[n] += __shfl_xor(acc[n], o, 64);
  }
  if (!live || sub != 0) return;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    float v = acc[n];
    if (out_act == 2) v = tanhf(v);
    if (out) out[((size_t)b * N + n) * H * W + rem] = v;
    if (out_bf) out_bf[(size_t)m * N + n] = f2bf(v);
  }
}

extern "C" ctta_status ctta_conv_small_n(const void* x, int c, int batch, int hi, int wi, int kh, int kw,
                                         int pad_h, int pad_w, const float* w, const float* bias, int n,
                                         int in_act, float in_slope, int out_act, float* out_nchw,
                                         void* out_bf16_nhwc, void* stream) {
  CTTA_REQUIRE(x && w && (out_nchw || out_bf16_nhwc), "conv_small_n: null pointer");
  CTTA_REQUIRE(c % 8 == 0, "conv_small_n: C=%d must be a multiple of 8", c);
  const long long total = (long long)batch * hi * wi;
  hipStream_t s = (hipStream_t)stream;
  // few pixels with long dot products: 8 lanes per pixel (see the kernel)
  const int lpp = (total <= (1 << 18) && c >= 64) ? 8 : 1;
  const dim3 grid((unsigned)((total * lpp + 255) / 256));
#define LAUNCH_NL(NN, LL)                                                                                 \
  hipLaunchKernelGGL((conv_small_n_kernel<NN, LL>), grid, dim3(256), 0, s, (const bf16_t*)x, c, batch, hi, \
                     wi, kh, kw, pad_h, pad_w, w, bias, in_act, in_slope, out_act, out_nchw,              \
                     (bf16_t*)out_bf16_nhwc)
#define LAUNCH_N(NN) do { if (lpp == 8) LAUNCH_NL(NN, 8); else LAUNCH_NL(NN, 1); } while (0)
  switch (n) {
    case 1: LAUNCH_N(1); break;
    case 2: LAUNCH_N(2); break;
    case 4: LAUNCH_N(4); break;
    case 8: LAUNCH_N(8); break;
    default: CTTA_REQUIRE(false, "conv_small_n: n=%d unsupported (1,2,4,8)", n);
  }
#undef LAUNCH_NL
#undef LAUNCH_N
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ C wrappers
#define REQ_VEC(n) CTTA_REQUIRE((n) % 4 == 0, "n_per_sample must be a multiple of 4")

extern "C" ctta_status ctta_heun_scale_model_input(const float* x, const float* sigma, float* out,
                                                   int batch, int64_t n_per_sample, void* stream) {
  CTTA_REQUIRE(x && sigma && out, "heun_scale: null pointer"); REQ_VEC(n_per_sample);
  const long long nv = n_per_sample / 4, total = nv * batch;
  hipLaunchKernelGGL(heun_scale_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x, sigma, (float4*)out, nv, total);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_heun_add_noise(const float* x0, const float* noise, const float* sigma,
                                           float* out, int batch, int64_t n_per_sample, void* stream) {
  CTTA_REQUIRE(x0 && noise && sigma && out, "heun_add_noise: null pointer"); REQ_VEC(n_per_sample);
  const long long nv = n_per_sample / 4, total = nv * batch;
  hipLaunchKernelGGL(heun_add_noise_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x0, (const float4*)noise, sigma, (float4*)out, nv, total);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_heun_step_first(const float* v, const float* x, const float* sigma,
                                            const float* sigma_next, float* prev, float* deriv,
                                            int batch, int64_t n_per_sample, void* stream) {
  CTTA_REQUIRE(v && x && sigma && sigma_next && prev && deriv, "heun_step_first: null pointer");
  REQ_VEC(n_per_sample);
  const long long nv = n_per_sample / 4, total = nv * batch;
  hipLaunchKernelGGL(heun_first_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)v, (const float4*)x, sigma, sigma_next, (float4*)prev,
                     (float4*)deriv, nv, total);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_heun_step_second(const float* v, const float* x_hat, const float* x_stored,
                                             const float* deriv_prev, const float* sigma,
                                             const float* sigma_next, float* prev, int batch,
                                             int64_t n_per_sample, void* stream) {
  CTTA_REQUIRE(v && x_hat && x_stored && deriv_prev && sigma && sigma_next && prev,
               "heun_step_second: null pointer");
  REQ_VEC(n_per_sample);
  const long long nv = n_per_sample / 4, total = nv * batch;
  hipLaunchKernelGGL(heun_second_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)v, (const float4*)x_hat, (const float4*)x_stored,
                     (const float4*)deriv_prev, sigma, sigma_next, (float4*)prev, nv, total);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_cfg_combine(const float* uncond, const float* cond, const float* w,
                                        float* out, int batch, int64_t n_per_sample, void* stream) {
  CTTA_REQUIRE(uncond && cond && w && out, "cfg_combine: null pointer"); REQ_VEC(n_per_sample);
  const long long nv = n_per_sample / 4, total = nv * batch;
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)uncond, (const float4*)cond, w, (float4*)out, nv, total);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_snr_mse_loss(const float* pred, const float* target, const float* sigma,
                                         float gamma, float* per_instance, float* loss, int batch,
                                         int64_t n_per_sample, void* stream) {
  CTTA_REQUIRE(pred && target && per_instance && loss && (gamma <= 0.f || sigma),
               "snr_mse_loss: null pointer");
  REQ_VEC(n_per_sample);
  hipLaunchKernelGGL(inst_mse_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)pred, (const float4*)target, (long long)(n_per_sample / 4),
                     per_instance);
  CTTA_LAUNCH_CHECK();
  hipLaunchKernelGGL(snr_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, per_instance, sigma,
                     gamma, batch, loss);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_lincomb2_rows(const float* x, const float* y, const float* a, const float* b, float* out,
                                          int batch, int64_t n_per_sample, float clamp, void* stream) {
  CTTA_REQUIRE(x && y && a && b && out, "lincomb2_rows: null pointer");
  REQ_VEC(n_per_sample);
  const long long nv = n_per_sample / 4, total = nv * batch;
  hipLaunchKernelGGL(lincomb2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (const float4*)y, a, b, (float4*)out, nv, total, clamp);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_weighted_mse_loss(const float* pred, const float* target, const float* weights,
                                              float* per_instance, float* loss, int batch, int64_t n_per_sample,
                                              void* stream) {
  CTTA_REQUIRE(pred && target && per_instance && loss, "weighted_mse_loss: null pointer");
  REQ_VEC(n_per_sample);
  hipLaunchKernelGGL(inst_mse_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const float4*)pred,
                     (const float4*)target, (long long)(n_per_sample / 4), per_instance);
  CTTA_LAUNCH_CHECK();
  hipLaunchKernelGGL(weighted_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, per_instance, weights, batch, loss);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_ema_update2(const float* param, float* shadow_a, double decay_a,
                                        float* shadow_b, double decay_b, int64_t n, void* stream) {
  CTTA_REQUIRE(param && shadow_a && n >= 0, "ema_update2: null pointer");
  CTTA_REQUIRE(decay_a >= 0.0 && decay_a <= 1.0 && decay_b >= 0.0 && decay_b <= 1.0,
               "ema_update2: decay outside [0,1]");   // train_utils.py:272
  const bool aligned = (((uintptr_t)param | (uintptr_t)shadow_a | (uintptr_t)shadow_b) & 15) == 0;
  const long long nv = aligned ? n / 4 : 0;
  // (1. - ema_decay) is a Python double scalar, rounded to fp32 when it meets the fp32 tensor
  const float ka = (float)(1.0 - decay_a), kb = (float)(1.0 - decay_b);
  if (nv > 0) {
    long long blocks = (nv + 511) / 512;       // two vectors per thread and trip
    if (blocks > 2048) blocks = 2048;
    if (shadow_b)
      hipLaunchKernelGGL(ema2_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                         (const float4*)param, (float4*)shadow_a, ka, (float4*)shadow_b, kb, nv);
    else
      hipLaunchKernelGGL(ema2_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                         (const float4*)param, (float4*)shadow_a, ka, (float4*)nullptr, 0.f, nv);
    CTTA_LAUNCH_CHECK();
  }
  const long long start = nv * 4;
  if (start < n) {
    const long long rem = n - start;
    hipLaunchKernelGGL(ema2_tail_kernel, dim3((unsigned)((rem + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, param, shadow_a, ka, shadow_b, kb, start, (long long)n);
    CTTA_LAUNCH_CHECK();
  }
  return CTTA_OK;
}

// The two halves of ctta_wav_finalize for a batch that is sharded over ranks (vocoder_infer centres with the extrema of
// the WHOLE batch, hifigan/utilities.py:85): extrema -> (max, min) as floats on the device, which the caller MAX-reduces
// over ranks, then centre with the reduced pair.  Same arithmetic as the one-call form: shift = (max + min) / 2.
__global__ void minmax_decode_kernel(const unsigned int* __restrict__ mm, float* __restrict__ out) {
  auto dec = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
  out[0] = dec(mm[0]);
  out[1] = dec(mm[1]);
}
__global__ void minmax_encode_kernel(const float* __restrict__ in, unsigned int* __restrict__ mm) {
  auto enc = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
  mm[0] = enc(in[0]);
  mm[1] = enc(in[1]);
}
extern "C" ctta_status ctta_wav_extrema(const float* wav, int64_t n, float* scratch, float* max_min, void* stream) {
  CTTA_REQUIRE(wav && scratch && max_min && n > 0, "wav_extrema: null pointer");
  hipStream_t s = (hipStream_t)stream;
  unsigned int* mm = reinterpret_cast<unsigned int*>(scratch);
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, mm);
  long long mb = (n / 4 + 255) / 256;
  if (mb > 1024) mb = 1024;
  if (mb < 1) mb = 1;
  hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)mb), dim3(256), 0, s, wav, (long long)n, mm);
  hipLaunchKernelGGL(minmax_decode_kernel, dim3(1), dim3(1), 0, s, mm, max_min);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
extern "C" ctta_status ctta_wav_center(const float* wav, int64_t n, const float* max_min, float* scratch, float* centred,
                                       int16_t* pcm, void* stream) {
  CTTA_REQUIRE(wav && max_min && scratch && n > 0, "wav_center: null pointer");
  hipStream_t s = (hipStream_t)stream;
  unsigned int* mm = reinterpret_cast<unsigned int*>(scratch);
  hipLaunchKernelGGL(minmax_encode_kernel, dim3(1), dim3(1), 0, s, max_min, mm);
  hipLaunchKernelGGL(wav_finalize_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, wav, (long long)n, mm, centred, pcm);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_wav_finalize(const float* wav, int64_t n, float* scratch, float* centred,
                                         int16_t* pcm, void* stream) {
  CTTA_REQUIRE(wav && scratch && n > 0, "wav_finalize: null pointer");
  hipStream_t s = (hipStream_t)stream;
  unsigned int* mm = reinterpret_cast<unsigned int*>(scratch);
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, mm);
  {
    long long mb = (n / 4 + 255) / 256;
    if (mb > 1024) mb = 1024;
    if (mb < 1) mb = 1;
    hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)mb), dim3(256), 0, s, wav, (long long)n, mm);
  }
  CTTA_LAUNCH_CHECK();
  hipLaunchKernelGGL(wav_finalize_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, wav, (long long)n,
                     mm, centred, pcm);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
