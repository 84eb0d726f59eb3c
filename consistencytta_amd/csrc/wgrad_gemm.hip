// Implicit weight-gradient GEMM of the distillation step's backward (train.py:332-346 -> autograd of F.conv2d / F.linear):
//
//   slab[s][n][c * T + t] = sum_{m in split s}  dY[m][n] * X[pixel(m, tap t)][c]          T = 9 (3x3, stride 1, pad 1) or 1
//
// Until round 3 the operand Q[(c, t)][m] = im2col(X)^T was MATERIALISED (ctta_im2col_t: nine shifted, transposed copies of
// the layer input per convolution, 4.3 ms of pure data movement per B = 9 step and the largest single item of its HBM
// traffic) and multiplied by the plain GEMM.  Here the layer input is read as it lies in HBM (NHWC, channels contiguous):
//   * a workgroup stages, per 64 output positions, the pixels those positions' taps touch ONCE -- a (64 / W + 2) x (W + 2)
//     patch with its zero border (a linear layer: the 64 rows) -- and the 64 x 64 tile of dY^T;
//   * the MFMA operand whose contraction index is the POSITION is read out of that channel-contiguous image with
//     ds_read_b64_tr_b16 (the LDS transpose read: lane t of a 16-lane group names the address of row k0 + t / 4, columns
//     4 (t % 4) .. + 3, and receives column t % 16, rows k0 .. k0 + 3; tools/tr_probe.hip).  A tap is nothing but another
//     row address per lane: no shifted copy, no alignment constraint, the zero border is part of the patch;
//   * positions inside a 32-chunk are visited in the order {4 g + j, 16 + 4 g + j} (g = lane / 16, j = 0..3) on BOTH
//     operands -- a sum does not care -- which puts the two 16-lane groups a pass of the LDS serves on disjoint banks
//     for row strides = 32 (mod 64) bytes;
//   * nine taps x 16 channels x 64 output channels per wave stay in 144 accumulator registers, so a lane ends with the
//     36 CONSECUTIVE floats (4 channels x 9 taps) of one weight row and stores them as nine float4.
// dY^T (m contiguous) is still the transposed copy the main stream makes into the job's slot: that copy is also what lets
// the gradient job outlive dY on the side stream.
// Bias and per-sample (time-embedding) columns -- the all-ones / indicator rows of the old Q -- are plain row sums of
// dY^T: wgrad_rowsum_kernel.
#include "common.h"

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

struct WgradParams {
  const bf16_t* dyt;     // [N][mp]   dY^T, positions contiguous
  const bf16_t* x;       // NHWC pixels, xld elements apart, C channels used
  float* slabs;          // [S][N][ld]
  int N, C, xld, mp, ld;
  int H, W, HW;          // image geometry (T = 9)
  int M;                 // valid positions
  int seg;               // positions per split, a multiple of 64
  long long slab_stride;
};

template <int T>
struct WgTile {
  static constexpr int CB = T == 9 ? 1 : 4;          // 16-channel blocks per wave (and tap)
  static constexpr int TC = 64 * CB;                 // channels per workgroup
  static constexpr int RSX = TC * 2 + 32;            // LDS bytes per staged pixel: = 32 (mod 64)
  static constexpr int RSB = 64 * 2 + 16;            // LDS bytes per dY^T row
  static constexpr int MAXPIX = T == 9 ? 136 : 64;   // (64 / W + 2) (W + 2) <= 136 for W in {2 .. 32}
  static constexpr int XV = (MAXPIX * (TC / 8) + 255) / 256;   // 16-byte vectors of the patch per thread
};

template <int T>
__global__ __launch_bounds__(256, 2) void wgrad_implicit_kernel(const WgradParams p) {
  using K = WgTile<T>;
  constexpr int CB = K::CB, TC = K::TC, RSX = K::RSX, RSB = K::RSB, XV = K::XV, NA = T * CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                                   // [npix][RSX]
  unsigned char* bs = smem + (size_t)K::MAXPIX * RSX;         // [64 n][RSB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, kg = lane >> 4;
  const int c0 = blockIdx.x * TC, n0 = blockIdx.y * 64, split = blockIdx.z;
  const int W = p.W, Wp = W + 2;
  const int npix = T == 9 ? (64 / W + 2) * Wp : 64;
  const int m_lo = split * p.seg, m_hi = min(m_lo + p.seg, p.mp);

  // ---- per-lane LDS addresses.  A operand (X^T, channels x positions): k slot (r, g = kg, j) <-> position 32 ks + 16 r + 4 kg + j;
  // this lane NAMES the row of j = lq / 4 and the 4 channels (lq % 4) * 4 .. + 3 of its wave's channel block
  int abase[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = 32 * ks + 16 * r + 4 * kg + (lq >> 2);
      const int pix = T == 9 ? (k / W) * Wp + (k % W) : k;    // tap (0, 0) of position k inside the bordered patch
      abase[ks][r] = pix * RSX + (wave * 16 + (lq & 3) * 4) * 2;
    }
  // B operand (dY^T, output channels x positions): row lq of block nb, positions 32 ks + 4 kg .. + 3 and 32 ks + 16 + 4 kg .. + 3
  const int bbase = lq * RSB + kg * 8;

  f32x4_t acc[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[a][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- staging plan: this thread's vectors of the patch and of the dY^T tile
  uint4 xr[XV], br[2];
  auto fetch = [&](int m0) {
    if (T == 9) {
      const int img = m0 / p.HW, oh0 = (m0 - img * p.HW) / W;       // a 64-chunk never straddles an image (HW % 64 == 0)
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int q = tid + i * 256;
        const int pix = q / (TC / 8), ch = q - pix * (TC / 8);
        const int pr = pix / Wp, pc = pix - pr * Wp;
        const int ih = oh0 - 1 + pr, iw = pc - 1;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (pix < npix && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)W && c0 + ch * 8 < p.C && m0 < p.M)
          v = *reinterpret_cast<const uint4*>(p.x + ((size_t)(img * p.H + ih) * W + iw) * p.xld + c0 + ch * 8);
        xr[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int q = tid + i * 256;
        const int row = q / (TC / 8), ch = q - row * (TC / 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m0 + row < p.M && c0 + ch * 8 < p.C) v = *reinterpret_cast<const uint4*>(p.x + (size_t)(m0 + row) * p.xld + c0 + ch * 8);
        xr[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + i * 256;
      const int row = q >> 3, ch = q & 7;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n0 + row < p.N) v = *reinterpret_cast<const uint4*>(p.dyt + (size_t)(n0 + row) * p.mp + m0 + ch * 8);   // the pad of dY^T is zero
      br[i] = v;
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int q = tid + i * 256;
      const int pix = q / (TC / 8), ch = q - pix * (TC / 8);
      if (pix < npix) *reinterpret_cast<uint4*>(xs + (size_t)pix * RSX + ch * 16) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + i * 256;
      *reinterpret_cast<uint4*>(bs + (size_t)(q >> 3) * RSB + (q & 7) * 16) = br[i];
    }
  };

  if (m_lo < m_hi) fetch(m_lo);
  for (int m0 = m_lo; m0 < m_hi; m0 += 64) {
    __syncthreads();                       // the previous chunk's fragments have been read
    park();
    __syncthreads();
    if (m0 + 64 < m_hi) fetch(m0 + 64);    // in flight behind this chunk's MFMAs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t bf[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const unsigned char* q = bs + bbase + nb * 16 * RSB + ks * 64;
        const uint2 lo = *reinterpret_cast<const uint2*>(q), hi = *reinterpret_cast<const uint2*>(q + 32);
        bf[nb] = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int toff = T == 9 ? ((t / 3) * Wp + (t % 3)) * RSX : 0;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xs + abase[ks][0] + toff + cb * 128));
          const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xs + abase[ks][1] + toff + cb * 128));
          const uint2 u0 = __builtin_bit_cast(uint2, a0), u1 = __builtin_bit_cast(uint2, a1);
          const bf16x8_t af = __builtin_bit_cast(bf16x8_t, make_uint4(u0.x, u0.y, u1.x, u1.y));
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
            acc[t * CB + cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[nb], acc[t * CB + cb][nb], 0, 0, 0);
        }
      }
    }
  }

  // ---- D[i][j]: i = channel (4 kg + e inside the wave's block), j = output channel lq.  Row n of the slab receives, from
  // this lane, the 4 T consecutive floats of channels cc .. cc + 3: [e][t] = acc[t][..][e]
  float* slab = p.slabs + (size_t)split * p.slab_stride;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = n0 + nb * 16 + lq;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int cc = c0 + (cb * 4 + wave) * 16 + kg * 4;
      if (n < p.N && cc < p.C) {
        float* dst = slab + (size_t)n * p.ld + (size_t)cc * T;
        float v[4 * T];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < T; ++t) v[e * T + t] = acc[t * CB + cb][nb][e];
#pragma unroll
        for (int q = 0; q < T; ++q) *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      }
    }
  }
}

// slab[s][n][col0]         = sum over split s's positions of dY^T[n][m]                         (bias gradient)
// slab[s][n][col0 + 1 + b] = the same restricted to sample b = m / hw, b < nb                  (d temb of the resnets' conv1)
// One wave per (output channel, split); 8 positions per lane and load (the row is contiguous and 16-byte aligned: mp % 8 == 0).
// Sample boundaries are multiples of hw, which the host requires to be a multiple of 8 whenever nb > 0.
__global__ __launch_bounds__(64) void wgrad_rowsum_kernel(const bf16_t* __restrict__ dyt, int mp, int M, int seg, int hw,
                                                          int nb, float* __restrict__ slabs, long long slab_stride, int ld,
                                                          int col0) {
  const int n = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
  const bf16_t* row = dyt + (size_t)n * mp;
  float* out = slabs + (size_t)s * slab_stride + (size_t)n * ld + col0;
  const int lo = s * seg, hi = min(min(lo + seg, mp), M);
  float total = 0.f;
  const int nparts = nb > 0 ? nb : 1;
  for (int b = 0; b < nparts; ++b) {
    const int a = nb > 0 ? max(lo, b * hw) : lo, e = nb > 0 ? min(hi, (b + 1) * hw) : hi;
    float sum = 0.f;
    int m = a + lane * 8;
    for (; m + 8 <= e; m += 512) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(row + m), f);
      sum += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    }
    for (int t = m; t < e && t < m + 8; ++t) sum += __uint_as_float((unsigned)row[t] << 16);     // the ragged tail of M
    sum = wave_sum(sum);
    total += sum;
    if (nb > 0 && lane == 0) out[1 + b] = sum;
  }
  if (lane == 0) out[0] = total;
}

extern "C" int ctta_wgrad_implicit_supported(int taps, int c, int h, int w, int x_ld, int n) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("CTTA_WGRAD_IMPLICIT"); env = (e && e[0] == '0') ? 0 : 1; }
  if (!env) return 0;
  if (c < 32 || c % 8 != 0 || x_ld % 8 != 0 || n < 8) return 0;
  if (taps == 1) return 1;
  if (taps != 9) return 0;
  if (w < 2 || w > 32 || (w & (w - 1)) != 0) return 0;                 // 64 / w rows per chunk, patch <= 136 pixels
  return ((long long)h * w) % 64 == 0 ? 1 : 0;
}

extern "C" ctta_status ctta_wgrad_implicit(const void* dyt, int n, int mp, const void* x, int x_ld, int c, int batch, int h, int w,
                                           int taps, int m_valid, int splits, int bias_col, int sample_cols, float* slabs,
                                           int64_t slab_stride, int ld, void* stream) {
  CTTA_REQUIRE(dyt && x && slabs && n >= 1 && c >= 1 && splits >= 1 && mp >= 64 && m_valid >= 1 && m_valid <= mp,
               "wgrad_implicit: bad arguments");
  CTTA_REQUIRE(ctta_wgrad_implicit_supported(taps, c, h, w, x_ld, n),
               "wgrad_implicit: taps=%d c=%d h=%d w=%d x_ld=%d is outside the kernel's range (1x1, or 3x3 stride 1 pad 1 with a "
               "power-of-two width <= 32 and h*w a multiple of 64; c >= 32, multiples of 8)", taps, c, h, w, x_ld);
  CTTA_REQUIRE(mp % (64 * splits) == 0 && mp % 8 == 0, "wgrad_implicit: mp=%d must be a multiple of 64 * splits=%d", mp, 64 * splits);
  CTTA_REQUIRE(ld % 4 == 0 && ld >= c * taps + (bias_col >= 0 ? 1 + sample_cols : 0) && slab_stride % 4 == 0 &&
               ((uintptr_t)slabs & 15) == 0, "wgrad_implicit: slab rows must be 16-byte aligned and hold c * taps (+ bias) columns");
  CTTA_REQUIRE((long long)batch * h * w == m_valid, "wgrad_implicit: m_valid=%d is not batch * h * w", m_valid);
  CTTA_REQUIRE(bias_col < 0 || sample_cols == 0 || ((h * w) % 8 == 0 && sample_cols <= batch),
               "wgrad_implicit: per-sample columns need h*w %% 8 == 0 and sample_cols <= batch");
  WgradParams p;
  p.dyt = (const bf16_t*)dyt; p.x = (const bf16_t*)x; p.slabs = slabs;
  p.N = n; p.C = c; p.xld = x_ld; p.mp = mp; p.ld = ld;
  p.H = h; p.W = w; p.HW = h * w; p.M = m_valid; p.seg = mp / splits; p.slab_stride = slab_stride;
  hipStream_t s = (hipStream_t)stream;
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(0, 130 + (taps == 9 ? 1 : 0), n, (long long)c * taps, mp / splits, splits, s);
  if (taps == 9) {
    using K = WgTile<9>;
    const size_t smem = (size_t)K::MAXPIX * K::RSX + 64 * K::RSB;
    dim3 grid((unsigned)((c + K::TC - 1) / K::TC), (unsigned)((n + 63) / 64), (unsigned)splits);
    hipLaunchKernelGGL(wgrad_implicit_kernel<9>, grid, dim3(256), smem, s, p);
  } else {
    using K = WgTile<1>;
    const size_t smem = (size_t)K::MAXPIX * K::RSX + 64 * K::RSB;
    dim3 grid((unsigned)((c + K::TC - 1) / K::TC), (unsigned)((n + 63) / 64), (unsigned)splits);
    hipLaunchKernelGGL(wgrad_implicit_kernel<1>, grid, dim3(256), smem, s, p);
  }
  if (prof) ctta_prof_end(s);
  CTTA_LAUNCH_CHECK();
  if (bias_col >= 0) {
    hipLaunchKernelGGL(wgrad_rowsum_kernel, dim3((unsigned)n, (unsigned)splits), dim3(64), 0, s, (const bf16_t*)dyt, mp, m_valid,
                       mp / splits, h * w, sample_cols, slabs, (long long)slab_stride, ld, bias_col);
    CTTA_LAUNCH_CHECK();
  }
  return CTTA_OK;
}

// The row sums alone (bias gradient, per-sample d temb columns), for the layers whose product runs on conv_gemm (linears,
// stride-2 / upsampling samplers, conv_in): the all-ones / indicator ROWS that carried them through the GEMM cost a whole
// extra tile column (N = K + 1: 257 -> 3 tiles of 128 instead of 2) for one useful row.
extern "C" ctta_status ctta_wgrad_rowsum(const void* dyt, int n, int mp, int m_valid, int splits, int hw, int sample_cols,
                                         float* slabs, int64_t slab_stride, int ld, int bias_col, void* stream) {
  CTTA_REQUIRE(dyt && slabs && n >= 1 && splits >= 1 && mp % (8 * splits) == 0 && m_valid >= 1 && m_valid <= mp && bias_col >= 0 &&
                   bias_col + 1 + sample_cols <= ld, "wgrad_rowsum: bad arguments");
  CTTA_REQUIRE(sample_cols == 0 || hw % 8 == 0, "wgrad_rowsum: per-sample columns need hw %% 8 == 0");
  hipLaunchKernelGGL(wgrad_rowsum_kernel, dim3((unsigned)n, (unsigned)splits), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)dyt,
                     mp, m_valid, mp / splits, hw, sample_cols, slabs, (long long)slab_stride, ld, bias_col);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
