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
#include <string.h>

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// "Direct" epilogue (round 5): a launch with ONE split adds its tile straight into the layer's gradient tensors through the
// pack map -- grad_w[row_off[n] + col(k)] += acc, grad_b[bias_idx[n]] += column sum -- instead of writing an fp32 slab that a
// scatter launch reads back (the slab round trip is 8 bytes per weight, the launch ~17 us + its dependent-launch gap on the
// weight-gradient stream).  Every gradient element belongs to exactly one workgroup, and old + acc is the sum the scatter
// forms for one slab: the same bits.
struct WgradDirect {
  float* grad_w;          // NULL: slab output
  const int* row_off;     // [n_rows] element offset of packed row n in grad_w, < 0: padding row
  const int* col_off;     // [k_cols] or NULL = identity
  int n_rows, k_cols;
  float* grad_b;          // NULL: no bias
  const int* bias_idx;    // [n_bias] or NULL = identity
  int n_bias;
};
__device__ __forceinline__ void direct_add4(const WgradDirect& d, int ro, int k, const float v[4]) {
  // four consecutive slab columns k .. k + 3 of one row
  // (the ABSOLUTE address decides: the flat gradient buffer packs the tensors back to back, and 155 of the light U-Net's 691
  // start at element offsets that are not multiples of 4 -- the 255-wide transformer layers -- so `(ro + k) & 3` alone let
  // misaligned dwordx4 accesses through, which only the GPU's unaligned-access mode made work)
  if (!d.col_off && k + 3 < d.k_cols && (reinterpret_cast<uintptr_t>(d.grad_w + (size_t)ro + k) & 15) == 0) {
    float4* dst = reinterpret_cast<float4*>(d.grad_w + (size_t)ro + k);
    float4 o = *dst;
    o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
    *dst = o;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (k + e >= d.k_cols) continue;
    const int co = d.col_off ? d.col_off[k + e] : k + e;
    if (co >= 0) d.grad_w[(size_t)ro + co] += v[e];
  }
}

struct WgradParams {
  const bf16_t* dy;      // NAT: [M][ldy] dY where it lies (output channels contiguous)
  int ldy, bias_col, sample_cols;   // NAT: column sums of dY (bias gradient, per-sample d temb) written by channel tile 0
  const bf16_t* dyt;     // [N][mp]   dY^T, positions contiguous
  const bf16_t* x;       // NHWC pixels, xld elements apart, C channels used
  float* slabs;          // [S][N][ld]
  int N, C, xld, mp, ld;
  int H, W, HW;          // image geometry (T = 9)
  int M;                 // valid positions
  int seg;               // positions per split, a multiple of 64
  long long slab_stride;
  WgradDirect direct;
};

template <int T>
struct WgTile {
  static constexpr int CB = T == 9 ? 1 : 4;          // 16-channel blocks per wave (and tap)
  static constexpr int TC = 64 * CB;                 // channels per workgroup
  static constexpr int RSX = TC * 2 + 32;            // LDS bytes per staged pixel: = 32 (mod 64)
  static constexpr int RSB = 64 * 2 + 16;            // LDS bytes per dY^T row
  static constexpr int RSN = 64 * 2 + 32;            // NAT: LDS bytes per natural dY row (64 output channels): = 32 (mod 64)
  static constexpr int MAXPIX = T == 9 ? 136 : 64;   // (64 / W + 2) (W + 2) <= 136 for W in {2 .. 32}
  static constexpr int XV = (MAXPIX * (TC / 8) + 255) / 256;   // 16-byte vectors of the patch per thread
};

// NAT (round 4): dY is read where it lies -- natural 64-position x 64-channel tiles, the B operand through the same transposing
// LDS read as X -- and the bias / per-sample column sums are added up by the workgroups of channel tile 0 from the vectors
// they stage (no dY^T copy, no row-sum launch).
template <int T, bool NAT>
__global__ __launch_bounds__(256, 2) void wgrad_implicit_kernel(const WgradParams p) {
  using K = WgTile<T>;
  constexpr int CB = K::CB, TC = K::TC, RSX = K::RSX, RSB = NAT ? K::RSN : K::RSB, XV = K::XV, NA = T * CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                                   // [npix][RSX]
  unsigned char* bs = smem + (size_t)K::MAXPIX * RSX;         // [64 n][RSB]   (NAT: [64 positions][RSN])
  float* red = reinterpret_cast<float*>(bs + 64 * RSB);       // NAT: [32 row phases][64 channels] for the column sums
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
  int nbase[2][2];       // NAT: the transposing read names position row 32 ks + 16 r + 4 kg + lq / 4, channels (lq % 4) * 4 .. + 3
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 2; ++r) nbase[ks][r] = (32 * ks + 16 * r + 4 * kg + (lq >> 2)) * RSB + (lq & 3) * 8;
  const bool do_sums = NAT && p.bias_col >= 0 && blockIdx.x == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float tot = 0.f;       // threads 0..63: running column sum of channel n0 + tid over the whole split
  int acc_sample = -1;   // the sample whose rows bsum holds (per-sample columns only)
  if (do_sums && !p.direct.grad_w && tid < 64 && n0 + tid < p.N)     // samples this split does not touch contribute zero (the scatter adds all splits)
    for (int b = 0; b < p.sample_cols; ++b) p.slabs[(size_t)split * p.slab_stride + (size_t)(n0 + tid) * p.ld + p.bias_col + 1 + b] = 0.f;

  f32x4_t acc[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[a][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- staging plan: this thread's vectors of the patch and of the dY^T tile
  uint4 xr[XV], br[2];
  bool xok[XV], bok[2];        // what must read as zero is selected when the vectors are parked (see wgrad_tn_kernel)
#pragma unroll
  for (int i = 0; i < XV; ++i) xok[i] = true;
  bok[0] = bok[1] = true;
  auto fetch = [&](int m0) {
    if (T == 9) {
      // (branch-free like wgrad_tn_kernel's fetch: clamped addresses, zeros selected afterwards -- counted vmcnt waits)
      const int m0c = m0 < p.M ? m0 : 0;
      const int img = m0c / p.HW, oh0 = (m0c - img * p.HW) / W;       // a 64-chunk never straddles an image (HW % 64 == 0)
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int q = tid + i * 256;
        const int pix = q / (TC / 8), ch = q - pix * (TC / 8);
        const int pr = pix / Wp, pc = pix - pr * Wp;
        const int ih = oh0 - 1 + pr, iw = pc - 1;
        const bool ok = pix < npix && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)W && c0 + ch * 8 < p.C && m0 < p.M;
        const int ihc = ih < 0 ? 0 : (ih >= p.H ? p.H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
        const int cc = c0 + ch * 8 < p.C ? c0 + ch * 8 : 0;
        xr[i] = *reinterpret_cast<const uint4*>(p.x + ((size_t)(img * p.H + ihc) * W + iwc) * p.xld + cc);
        xok[i] = ok;
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
      if constexpr (NAT) {
        const bool ok = m0 + row < p.M && n0 + ch * 8 < p.N;
        const int rc = m0 + row < p.M ? m0 + row : p.M - 1, nc = n0 + ch * 8 < p.N ? n0 + ch * 8 : 0;
        v = *reinterpret_cast<const uint4*>(p.dy + (size_t)rc * p.ldy + nc);
        bok[i] = ok;
      } else {
        if (n0 + row < p.N) v = *reinterpret_cast<const uint4*>(p.dyt + (size_t)(n0 + row) * p.mp + m0 + ch * 8);   // the pad of dY^T is zero
      }
      br[i] = v;
    }
  };
  // NAT column sums: fold the 32 row phases of the chunk range summed so far in a fixed order; threads 0..63 own one channel
  auto fold_sums = [&](int sample) {     // called by every thread of a do_sums workgroup (uniform)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[(tid >> 3) * 64 + (tid & 7) * 8 + e] = bsum[e]; bsum[e] = 0.f; }
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
      for (int r = 0; r < 32; ++r) t += red[r * 64 + tid];
      tot += t;
      if (!p.direct.grad_w && sample >= 0 && sample < p.sample_cols && n0 + tid < p.N)
        p.slabs[(size_t)split * p.slab_stride + (size_t)(n0 + tid) * p.ld + p.bias_col + 1 + sample] = t;
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int q = tid + i * 256;
      const int pix = q / (TC / 8), ch = q - pix * (TC / 8);
      if (pix < npix) *reinterpret_cast<uint4*>(xs + (size_t)pix * RSX + ch * 16) = xok[i] ? xr[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + i * 256;
      if (!bok[i]) br[i] = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(bs + (size_t)(q >> 3) * RSB + (q & 7) * 16) = br[i];
      if (do_sums) {
        float f[8];
        unpack8(br[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += f[e];
      }
    }
  };

  if (m_lo < m_hi) fetch(m_lo);
  for (int m0 = m_lo; m0 < m_hi; m0 += 64) {
    // NAT, per-sample columns: a 64-position chunk lies inside one sample; at a sample boundary what was summed so far
    // belongs to the previous sample
    if (do_sums && p.sample_cols > 0 && m0 < p.M) {
      const int b = m0 / p.HW;
      if (acc_sample >= 0 && b != acc_sample) fold_sums(acc_sample);
      acc_sample = b;
    }
    __syncthreads();                       // the previous chunk's fragments have been read
    park();
    __syncthreads();
    if (m0 + 64 < m_hi) fetch(m0 + 64);    // in flight behind this chunk's MFMAs
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t bf[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        if constexpr (NAT) {
          const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bs + nbase[ks][0] + nb * 32));
          const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(bs + nbase[ks][1] + nb * 32));
          const uint2 u0 = __builtin_bit_cast(uint2, b0), u1 = __builtin_bit_cast(uint2, b1);
          bf[nb] = __builtin_bit_cast(bf16x8_t, make_uint4(u0.x, u0.y, u1.x, u1.y));
        } else {
          const unsigned char* q = bs + bbase + nb * 16 * RSB + ks * 64;
          const uint2 lo = *reinterpret_cast<const uint2*>(q), hi = *reinterpret_cast<const uint2*>(q + 32);
          bf[nb] = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
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

  if (do_sums) {       // the last sample's share, then the split's total = the bias column
    fold_sums(acc_sample);
    if (tid < 64 && n0 + tid < p.N) {
      if (p.direct.grad_w) {     // direct: one split, no per-sample columns (host)
        const int n = n0 + tid;
        if (p.direct.grad_b && n < p.direct.n_bias) {
          const int j = p.direct.bias_idx ? p.direct.bias_idx[n] : n;
          if (j >= 0) p.direct.grad_b[j] += tot;
        }
      } else {
        p.slabs[(size_t)split * p.slab_stride + (size_t)(n0 + tid) * p.ld + p.bias_col] = tot;
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
        float v[4 * T];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < T; ++t) v[e * T + t] = acc[t * CB + cb][nb][e];
        if (p.direct.grad_w) {
          const int ro = n < p.direct.n_rows ? p.direct.row_off[n] : -1;
          if (ro >= 0) {
#pragma unroll
            for (int q = 0; q < T; ++q) direct_add4(p.direct, ro, cc * T + 4 * q, v + 4 * q);
          }
        } else {
          float* dst = slab + (size_t)n * p.ld + (size_t)cc * T;
#pragma unroll
          for (int q = 0; q < T; ++q) *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
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
  if (c < 32 || c % 8 != 0 || x_ld % 8 != 0 || n < 8) return 0;
  if (taps == 1) return 1;
  if (taps != 9) return 0;
  if (w < 2 || w > 32 || (w & (w - 1)) != 0) return 0;                 // 64 / w rows per chunk, patch <= 136 pixels
  return ((long long)h * w) % 64 == 0 ? 1 : 0;
}

static ctta_status wgrad_implicit_launch(const void* dyt, const void* dy_nat, int ldy, int n, int mp, const void* x, int x_ld,
                                         int c, int batch, int h, int w, int taps, int m_valid, int splits, int bias_col,
                                         int sample_cols, float* slabs, int64_t slab_stride, int ld, void* stream,
                                         const WgradDirect* direct = nullptr);
extern "C" ctta_status ctta_wgrad_implicit(const void* dyt, int n, int mp, const void* x, int x_ld, int c, int batch, int h, int w,
                                           int taps, int m_valid, int splits, int bias_col, int sample_cols, float* slabs,
                                           int64_t slab_stride, int ld, void* stream) {
  CTTA_REQUIRE(dyt, "wgrad_implicit: null pointer");
  return wgrad_implicit_launch(dyt, nullptr, 0, n, mp, x, x_ld, c, batch, h, w, taps, m_valid, splits, bias_col, sample_cols, slabs,
                               slab_stride, ld, stream);
}
// the same product with dY [m_valid][ldy] read where it lies (3x3 convolutions; the linears have ctta_wgrad_tn)
extern "C" ctta_status ctta_wgrad_implicit_inplace(const void* dy, int ldy, int n, int mp, const void* x, int x_ld, int c, int batch,
                                                   int h, int w, int taps, int m_valid, int splits, int bias_col, int sample_cols,
                                                   float* slabs, int64_t slab_stride, int ld, void* stream) {
  CTTA_REQUIRE(dy && ldy % 8 == 0 && ldy >= n && n % 8 == 0 && taps == 9, "wgrad_implicit_inplace: dy [m][ldy] with ldy, n multiples of 8, 3x3 only");
  return wgrad_implicit_launch(nullptr, dy, ldy, n, mp, x, x_ld, c, batch, h, w, taps, m_valid, splits, bias_col, sample_cols, slabs,
                               slab_stride, ld, stream);
}
// One split, the tile added straight into the gradient tensors through the pack map (identity columns: a 3x3 weight row is
// (cin, kh, kw)-contiguous like the kernel's own row): no slab, no scatter launch.
extern "C" ctta_status ctta_wgrad_implicit_direct(const void* dy, int ldy, int n, int mp, const void* x, int x_ld, int c, int batch,
                                                  int h, int w, int m_valid, int k_cols, int n_rows, const int32_t* row_off,
                                                  float* grad_w, int n_bias, const int32_t* bias_idx, float* grad_b, void* stream) {
  CTTA_REQUIRE(dy && ldy % 8 == 0 && ldy >= n && n % 8 == 0 && grad_w && row_off && k_cols >= 1 && n_rows >= 1,
               "wgrad_implicit_direct: bad arguments");
  WgradDirect d;
  d.grad_w = grad_w; d.row_off = row_off; d.col_off = nullptr; d.n_rows = n_rows; d.k_cols = k_cols;
  d.grad_b = grad_b; d.bias_idx = bias_idx; d.n_bias = grad_b ? n_bias : 0;
  return wgrad_implicit_launch(nullptr, dy, ldy, n, mp, x, x_ld, c, batch, h, w, 9, m_valid, 1, grad_b ? c * 9 : -1, 0, nullptr, 0, 0,
                               stream, &d);
}
static ctta_status wgrad_implicit_launch(const void* dyt, const void* dy_nat, int ldy, int n, int mp, const void* x, int x_ld,
                                         int c, int batch, int h, int w, int taps, int m_valid, int splits, int bias_col,
                                         int sample_cols, float* slabs, int64_t slab_stride, int ld, void* stream,
                                         const WgradDirect* direct) {
  const bool nat = dy_nat != nullptr;
  if (direct) {     // the tile goes straight into the gradient tensors: no slab (the alignment checks below look at a dummy)
    CTTA_REQUIRE(nat && taps == 9 && splits == 1 && sample_cols == 0 && direct->grad_w && direct->row_off,
                 "wgrad_implicit_direct: 3x3, dY in place, one split, no per-sample columns");
    slabs = reinterpret_cast<float*>((uintptr_t)16);
    ld = (c * taps + 1 + 3) / 4 * 4; slab_stride = (int64_t)n * ld;
  }
  CTTA_REQUIRE((dyt || dy_nat) && x && slabs && n >= 1 && c >= 1 && splits >= 1 && mp >= 64 && m_valid >= 1 && m_valid <= mp,
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
  p.dy = (const bf16_t*)dy_nat; p.ldy = ldy; p.bias_col = bias_col; p.sample_cols = sample_cols;
  p.dyt = (const bf16_t*)dyt; p.x = (const bf16_t*)x; p.slabs = slabs;
  p.N = n; p.C = c; p.xld = x_ld; p.mp = mp; p.ld = ld;
  p.H = h; p.W = w; p.HW = h * w; p.M = m_valid; p.seg = mp / splits; p.slab_stride = slab_stride;
  if (direct) p.direct = *direct; else memset(&p.direct, 0, sizeof(p.direct));
  hipStream_t s = (hipStream_t)stream;
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(0, 130 + (taps == 9 ? 1 : 0), n, (long long)c * taps, mp / splits, splits, s);
  if (taps == 9 && nat) {
    using K = WgTile<9>;
    const size_t smem = (size_t)K::MAXPIX * K::RSX + 64 * K::RSN + 32 * 64 * sizeof(float);
    dim3 grid((unsigned)((c + K::TC - 1) / K::TC), (unsigned)((n + 63) / 64), (unsigned)splits);
    hipLaunchKernelGGL((wgrad_implicit_kernel<9, true>), grid, dim3(256), smem, s, p);
  } else if (taps == 9) {
    using K = WgTile<9>;
    const size_t smem = (size_t)K::MAXPIX * K::RSX + 64 * K::RSB;
    dim3 grid((unsigned)((c + K::TC - 1) / K::TC), (unsigned)((n + 63) / 64), (unsigned)splits);
    hipLaunchKernelGGL((wgrad_implicit_kernel<9, false>), grid, dim3(256), smem, s, p);
  } else {
    using K = WgTile<1>;
    const size_t smem = (size_t)K::MAXPIX * K::RSX + 64 * K::RSB;
    dim3 grid((unsigned)((c + K::TC - 1) / K::TC), (unsigned)((n + 63) / 64), (unsigned)splits);
    hipLaunchKernelGGL((wgrad_implicit_kernel<1, false>), grid, dim3(256), smem, s, p);
  }
  if (prof) ctta_prof_end(s);
  CTTA_LAUNCH_CHECK();
  if (bias_col >= 0 && !nat) {
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

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a LINEAR layer with BOTH operands read where they lie (round 4):
//     slab[s][n][c] = sum over split s's rows m of dY[m][n] * X[m][c]          dY [M][ldy], X [M][ldx], row-major bf16
// The conv_gemm route needs both operands m-contiguous and therefore a transposed copy of each (ctta_transpose_bf16 on the
// main stream + ctta_im2col_t on the side stream: 10.4 + 5.5 GB of the 259 GB a distillation step moves).  Here a workgroup
// stages natural 64-row tiles of dY (128 columns) and X (128 columns) in LDS -- rows of 256 bytes, coalesced -- and BOTH
// MFMA operands come out of them through ds_read_b64_tr_b16, the transposing LDS read wgrad_implicit_kernel already uses
// for X (same k-slot order on both operands: position 32 ks + 16 r + 4 kg + j).  Four waves of 64 (c) x 64 (n); two 64-row
// chunks ride in registers behind a double-buffered LDS pair, one barrier per chunk.
struct WgradTnParams {
  const bf16_t* dy;
  const bf16_t* x;
  float* slabs;
  int N, C, ldy, ldx, ld;
  int M, mp, seg;
  long long slab_stride;
  int bias_col;          // >= C: the workgroups of the first column tile also write the column sums of dY there; < 0: none
  WgradDirect direct;
};

__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(const WgradTnParams p) {
  constexpr int RS = 128 * 2 + 32;                 // LDS bytes per staged row: = 32 (mod 64), conflict-free transposing reads
  constexpr int TILE = 64 * RS;                    // one operand tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [dY tile | X tile], ONE buffer (36 KB: a second one
  // made the kernel 73 KB per workgroup -- two of them fill a CU's LDS and lock the main and teacher streams' 48 KB
  // conv_gemm workgroups out of it: measured 92 vs 86 ms per pipelined step)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, kg = lane >> 4;
  const int wc = wave & 1, wn = wave >> 1;
  const int c0 = blockIdx.x * 128, n0 = blockIdx.y * 128, split = blockIdx.z;
  const int m_lo = split * p.seg, m_hi = min(m_lo + p.seg, p.mp);

  int abase[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r = 0; r < 2; ++r) abase[ks][r] = (32 * ks + 16 * r + 4 * kg + (lq >> 2)) * RS + (lq & 3) * 8;

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // staging plan: thread -> (row = q / 16, 16-byte column chunk = q % 16), four rows apart per vector
  const int srow = tid >> 4, sch = tid & 15;
  const bool n_ok = n0 + sch * 8 < p.N, c_ok = c0 + sch * 8 < p.C;
  uint4 st[2][8];                                  // [register stage][4 dY vectors | 4 X vectors]
  // bias gradient = column sums of dY: a thread's 8 dY columns never change (sch), so the workgroups of column tile 0 add up
  // what they stage anyway (a separate column-sum pass over dY cost more than the product: 50 us per launch)
  const bool do_bias = p.bias_col >= 0 && blockIdx.x == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // BRANCH-FREE: every load is issued (row clamped into the matrix, column chunk clamped to chunk 0 where the tile hangs over
  // the edge) and what must read as zero is selected afterwards.  With the loads under `if (m < M)` the compiler cannot count
  // vmcnt across the divergent branches and puts s_waitcnt vmcnt(0) in front of every park: the "two chunks in flight" of the
  // register stages were one (ISA of round 4's kernel: vmcnt(0) twice per chunk).
  const int ncol = n_ok ? n0 + sch * 8 : 0, ccol = c_ok ? c0 + sch * 8 : 0;
  auto fetch = [&](uint4 (&v)[8], int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + srow + i * 16;
      const int mc = m < p.M ? m : p.M - 1;
      v[i] = *reinterpret_cast<const uint4*>(p.dy + (size_t)mc * p.ldy + ncol);
      v[4 + i] = *reinterpret_cast<const uint4*>(p.x + (size_t)mc * p.ldx + ccol);
    }
  };
  // ... the zeros are selected when the stage is PARKED (its loads have landed by then; a select next to the load would make
  // the compiler wait for it on the spot)
  auto mask = [&](uint4 (&v)[8], int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool row_ok = m0 + srow + i * 16 < p.M;
      if (!(row_ok && n_ok)) v[i] = make_uint4(0, 0, 0, 0);
      if (!(row_ok && c_ok)) v[4 + i] = make_uint4(0, 0, 0, 0);
    }
  };
  auto add_bias = [&](const uint4 (&v)[8]) {       // when the stage is parked: its loads have landed
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack8(v[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += f[e];
      }
    }
  };
  auto park = [&](uint4 (&v)[8], int m0) {
    mask(v, m0);
    add_bias(v);
    unsigned char* d = smem + (size_t)srow * RS + sch * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(d + (size_t)i * 16 * RS) = v[i];
      *reinterpret_cast<uint4*>(d + TILE + (size_t)i * 16 * RS) = v[4 + i];
    }
  };
  auto compute = [&](int buf) {
    const unsigned char* ys = smem + (wn * 64) * 2;
    const unsigned char* xs = smem + TILE + (wc * 64) * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t bf[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ys + abase[ks][0] + nb * 32));
        const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ys + abase[ks][1] + nb * 32));
        const uint2 u0 = __builtin_bit_cast(uint2, b0), u1 = __builtin_bit_cast(uint2, b1);
        bf[nb] = __builtin_bit_cast(bf16x8_t, make_uint4(u0.x, u0.y, u1.x, u1.y));
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xs + abase[ks][0] + cb * 32));
        const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xs + abase[ks][1] + cb * 32));
        const uint2 u0 = __builtin_bit_cast(uint2, a0), u1 = __builtin_bit_cast(uint2, a1);
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, make_uint4(u0.x, u0.y, u1.x, u1.y));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[nb], acc[cb][nb], 0, 0, 0);
      }
    }
  };

  // chunk j (rows m_lo + 64 j ..) rides in register stage j % 2 (two chunks in flight) until it is parked into the LDS tiles
  const int nch = (m_hi - m_lo + 63) / 64;
  if (nch > 0) fetch(st[0], m_lo);
  if (nch > 1) fetch(st[1], m_lo + 64);
  for (int j0 = 0; j0 < nch; j0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = j0 + u;
      if (j < nch) {
        __syncthreads();                                   // chunk j - 1's fragments have been read
        park(st[u], m_lo + j * 64);
        if (j + 2 < nch) fetch(st[u], m_lo + (j + 2) * 64);
        __syncthreads();
        compute(0);
      }
    }
  }
  __syncthreads();

  // D[i][j]: i = X column 4 kg + e of the fragment, j = dY column lq  ->  row n of the slab gets 4 consecutive floats
  float* slab = p.slabs + (size_t)split * p.slab_stride;
  if (do_bias) {       // fold the 16 row phases of every column in a fixed order (the tiles are dead: the loop ended on a barrier)
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(srow * 16 + sch) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < p.N) {
      float t = 0.f;
      for (int r = 0; r < 16; ++r) t += red[(r * 16 + (tid >> 3)) * 8 + (tid & 7)];
      if (p.direct.grad_w) {
        const int n = n0 + tid;
        if (p.direct.grad_b && n < p.direct.n_bias) {
          const int j = p.direct.bias_idx ? p.direct.bias_idx[n] : n;
          if (j >= 0) p.direct.grad_b[j] += t;
        }
      } else {
        slab[(size_t)(n0 + tid) * p.ld + p.bias_col] = t;
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = n0 + wn * 64 + nb * 16 + lq;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int cc = c0 + wc * 64 + cb * 16 + kg * 4;
      if (n < p.N && cc < p.C) {
        const f32x4_t a = acc[cb][nb];
        if (p.direct.grad_w) {
          const int ro = n < p.direct.n_rows ? p.direct.row_off[n] : -1;
          const float v[4] = {a[0], a[1], a[2], a[3]};
          if (ro >= 0) direct_add4(p.direct, ro, cc, v);
        } else {
          *reinterpret_cast<float4*>(slab + (size_t)n * p.ld + cc) = make_float4(a[0], a[1], a[2], a[3]);
        }
      }
    }
  }
}

static ctta_status wgrad_tn_launch(const void* dy, int ldy, int n, const void* x, int ldx, int c, int m_valid, int mp, int splits,
                                   int bias_col, float* slabs, int64_t slab_stride, int ld, void* stream, const WgradDirect* direct);
extern "C" ctta_status ctta_wgrad_tn(const void* dy, int ldy, int n, const void* x, int ldx, int c, int m_valid, int mp, int splits,
                                     int bias_col, float* slabs, int64_t slab_stride, int ld, void* stream) {
  return wgrad_tn_launch(dy, ldy, n, x, ldx, c, m_valid, mp, splits, bias_col, slabs, slab_stride, ld, stream, nullptr);
}
// One split, the tile added straight into the gradient tensors through the pack map (col_off NULL = identity columns).
extern "C" ctta_status ctta_wgrad_tn_direct(const void* dy, int ldy, int n, const void* x, int ldx, int c, int m_valid, int mp,
                                            int k_cols, int n_rows, const int32_t* row_off, const int32_t* col_off, float* grad_w,
                                            int n_bias, const int32_t* bias_idx, float* grad_b, void* stream) {
  CTTA_REQUIRE(grad_w && row_off && k_cols >= 1 && k_cols <= c && n_rows >= 1, "wgrad_tn_direct: bad arguments");
  WgradDirect d;
  d.grad_w = grad_w; d.row_off = row_off; d.col_off = col_off; d.n_rows = n_rows; d.k_cols = k_cols;
  d.grad_b = grad_b; d.bias_idx = bias_idx; d.n_bias = grad_b ? n_bias : 0;
  const int ld = (c + 1 + 3) / 4 * 4;
  return wgrad_tn_launch(dy, ldy, n, x, ldx, c, m_valid, mp, 1, grad_b ? c : -1, reinterpret_cast<float*>((uintptr_t)16), (int64_t)n * ld, ld,
                         stream, &d);
}
static ctta_status wgrad_tn_launch(const void* dy, int ldy, int n, const void* x, int ldx, int c, int m_valid, int mp, int splits,
                                   int bias_col, float* slabs, int64_t slab_stride, int ld, void* stream, const WgradDirect* direct) {
  CTTA_REQUIRE(dy && x && slabs && n >= 1 && c >= 1 && splits >= 1 && m_valid >= 1 && m_valid <= mp, "wgrad_tn: bad arguments");
  CTTA_REQUIRE(ldy % 8 == 0 && ldx % 8 == 0 && n % 8 == 0 && c % 8 == 0 && ldy >= n && ldx >= c,
               "wgrad_tn: n=%d c=%d ldy=%d ldx=%d must be multiples of 8", n, c, ldy, ldx);
  CTTA_REQUIRE(mp % (64 * splits) == 0 && ld % 4 == 0 && ld >= c + (bias_col >= 0 ? 1 : 0) && (bias_col < 0 || bias_col >= c),
               "wgrad_tn: mp=%d splits=%d ld=%d", mp, splits, ld);
  static bool attr = false;
  constexpr int smem = 2 * 64 * (128 * 2 + 32);
  if (!attr) {
    CTTA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr = true;
  }
  WgradTnParams p;
  p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.slabs = slabs; p.N = n; p.C = c; p.ldy = ldy; p.ldx = ldx; p.ld = ld;
  p.M = m_valid; p.mp = mp; p.seg = mp / splits; p.slab_stride = slab_stride; p.bias_col = bias_col;
  if (direct) p.direct = *direct; else memset(&p.direct, 0, sizeof(p.direct));
  hipStream_t s = (hipStream_t)stream;
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(0, 142, c, n, p.seg, splits, s);
  hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)((c + 127) / 128), (unsigned)((n + 127) / 128), (unsigned)splits), dim3(256), smem, s, p);
  if (prof) ctta_prof_end(s);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
