// ConvParams and the fused output epilogues shared by the implicit-GEMM kernels (conv_gemm.hip) and the fused
// HiFi-GAN ResBlock unit (resunit.hip).
#pragma once
#include "common.h"

// Which tile variants carry the fp32 wide-store epilogue (`ConvParams::wide_f32`); see the epilogue in conv_gemm_kernel.h.
constexpr bool conv_wide_f32_ok(int bm, int bn, int bk, int wm, int wn, int mode, int stages) {
  return !(mode != 0 && bk == 32 && stages == 2 && bm / wm == 64 && bn / wn == 64);
}

struct ConvParams {
  const bf16_t* x0; const bf16_t* x1; int c0, c1, ct;
  int M, hi, wi, hs, ws, ups, ho, wo, howo;
  int kh, kw, taps, sh, sw, ph, pw, dh, dw;
  const bf16_t* w; int k_pad, n, nk;
  const float* bias; const float* bias_m; const float* rowvec; int rowvec_ld;
  const bf16_t* res; int res_ld;
  int in_act; float in_slope; int out_act; float out_slope; float alpha; int accumulate;
  void* out; int ldc; int out_f32; bf16_t* out2; float out2_slope; int scalar_store;
  long long obs, out_offset, out_limit;
  long long xgs, wgs, ogs;
  const bf16_t* zero;   // >= 16 bytes of zeros: source of out-of-range chunks in the direct-to-LDS path
  unsigned x_bytes, w_bytes;   // buffer extents for the descriptor (MODE 2) path
  int xs0;                     // row stride (elements) of source 0: c0 unless a wider matrix is sliced
  // split-K (groups == 1 only): blockIdx.z = split index, each split walks nk_split K-tiles and writes raw fp32
  // partial sums to slab `split` of a workspace (out / ldc / ogs are redirected by the host); the fused epilogue
  // runs afterwards in splitk_finish_kernel
  int ksplit, nk_split;
  // XCD-aware tile order (1-D grid): block id b runs on XCD b % 8; XCD x owns the contiguous M-tile range
  // [x * xcd_per, (x + 1) * xcd_per) and walks it with the N tiles innermost, so the tiles that share input rows
  // (neighbouring image rows, all N tiles of one M tile) meet in the same 4 MB L2 close in time.  0 = plain 2-D grid.
  int xcd_per, m_tiles, n_tiles, n_inner;   // n_inner = 0: M tiles innermost (many N tiles: keep the weight slice hot)
  // Weight-slab affinity (1-D grid; weight-dominated launches: every split-K launch, and launches with few row tiles).  A
  // "slab" is one (K-split, N-tile) pair = one slice of the weight matrix; work item i = slab * m_tiles + mt with
  // slab = zs * n_tiles + nt.  XCD x owns the contiguous item range [x * slab_per, (x + 1) * slab_per): all row tiles of a
  // slab run on ONE XCD, close in time, so a weight byte crosses the fabric into one L2 once (the plain 3-D grid spreads
  // the row tiles of a slab over all eight XCDs and every L2 pulls the whole weight matrix: 157.6 MB read for an
  // 18.9 MB matrix, profiles/pmc_by_shape_r01.txt).  The K-split index stays the summation slot: results are unchanged.
  int slab_total, slab_per;                 // slab_total = ksplit * n_tiles * m_tiles; 0 = off
  // first output row of this launch (a multiple of 4; normally 0).  The host splits the ragged last row tile of a
  // 256-row-tile launch off into a second launch with small tiles when that tile alone would open another round of the
  // 256 CUs (HiFi-GAN: M = 32 * 5121 = 640 * 256 + 32 rows x 2 column tiles = 5.008 rounds).
  int m_off;
  // 1: the tile is transposed through LDS and leaves as whole 64..256-byte row segments (see the kernel's epilogue)
  int wide_store;
  int wide_f32;         // fp32 output, bias-only epilogue: rows leave as whole TN*4-byte segments through the LDS transpose
  unsigned long long* stamps;   // diagnostic (ctta_conv_debug_stamps): per workgroup {hw id, t_begin, t_first_tile, t_main_done, t_epilogue_done}
  unsigned howo_inv, wo_inv;   // floor(2^32 / howo), floor(2^32 / wo) for fast_div
  int epi_fast;      // straight-line wide-store epilogue (bias / rowvec / residual / LeakyReLU / second output)
  int epi_act;       // ... with alpha != 1 or a LeakyReLU (the ACT instantiations)
  int epi_fast_geglu;
  int plain_out;   // destination element (m, n) sits at m*ldc + n (no per-batch stride, offset or limit)
  // GroupNorm statistics of the OUTPUT tensor from the epilogue (wide-store path only): every workgroup writes
  // (sum, sum of squares) of its BM x BN tile per channel group to gn_part[((b * gn_nchunk + chunk) * gn_G + g) * 2],
  // chunk = row-tile index inside sample b (gn_hw rows per sample, a multiple of BM) -- the layout gn_finalize_kernel folds.
  float* gn_part;
  int gn_cpg, gn_G, gn_hw, gn_nchunk;
  // Stream-K (conv_gemm_sk_kernel: ONE persistent launch of <= one workgroup per CU slot, groups == 1): the (output tile,
  // K step) items -- tile-major inside each of sk_chunks XCD chunks of whole tiles -- are cut into equal contiguous ranges, one
  // per workgroup.  A tile whose K walk is split over several workgroups leaves as fp32 partial tiles ([BM][BN] rows): the part
  // with the first K step in slot gridDim.x + id of sk_slots, every later part in slot id of its writer.  Each written part is
  // counted in sk_hdr[SK_FLAGS + tile]; the workgroups that wrote a tile's parts then fold it together -- fold blocks of a few
  // rows claimed from sk_hdr[SK_FLAGS + sk_tiles + tile], every block = the sum of the parts in K order + the fused epilogue,
  // so the result does not depend on who folds or when.  The writer of the LAST part folds at once; the others wait a bounded
  // time at the end of their own K walk and help: no unbounded wait anywhere, forward progress whatever is resident.
  // sk_hdr[0..2] = {start ticket, finished count, launches}; the last workgroup to finish zeroes tickets and counters (no
  // memset node per launch; the header is zeroed once where the workspace is allocated).
  unsigned* sk_hdr;
  float* sk_slots;
  int sk_tiles, sk_m_inner;   // sk_m_inner != 0: tile = nt * m_tiles + mt (row tiles innermost), else mt * n_tiles + nt
  int sk_chunks;              // 8, 4, 2 or 1 XCD chunks (conv_gemm_sk_kernel); gridDim.x is a multiple of it
};
constexpr int SK_FLAGS = 64;            // first flag word of the stream-K header
constexpr int SK_HDR_WORDS = 2048;      // header words in front of the partial slots (8 KB)
constexpr int SK_MAX_GRID = SK_HDR_WORDS - SK_FLAGS;

// The fused epilogue on 4 consecutive channels of one output row, for the wide-store paths (plain row-major bf16
// destination: element (m, n) at m*ldc + n): same operation order as epilogue_store.
__device__ __forceinline__ void epilogue_wide4(const ConvParams& p, const float4 q, const float4 bias4, int m, int n,
                                               size_t gofs, float2* gn_acc = nullptr) {
  // one order for every epilogue of the library: acc + (bias + rowvec), then the per-row bias, then the residual
  float4 c4 = bias4;
  if (p.rowvec) {
    const float4 rv = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m / p.howo) * p.rowvec_ld + n);
    c4.x += rv.x; c4.y += rv.y; c4.z += rv.z; c4.w += rv.w;
  }
  float v[4] = {q.x + c4.x, q.y + c4.y, q.z + c4.z, q.w + c4.w};
  if (p.bias_m) {
    const float bm = p.bias_m[m];
    v[0] += bm; v[1] += bm; v[2] += bm; v[3] += bm;
  }
  if (p.res) {
    const uint2 rr = *reinterpret_cast<const uint2*>(p.res + (size_t)m * p.res_ld + n);
    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
  }
  size_t oidx = gofs + (size_t)m * p.ldc + n;
  if (!p.plain_out) {   // ConvTranspose phases: per-batch stride, shifted and clipped rows (same rule as epilogue_store)
    const int b = m / p.howo;
    const long long inb = (long long)(m - b * p.howo) * p.ldc + n + p.out_offset;
    if (p.out_limit > 0 && (inb < 0 || inb >= p.out_limit)) return;
    oidx = gofs + (size_t)((long long)b * p.obs + inb);
  }
  bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + oidx;
  if (p.accumulate) {
    const uint2 old = *reinterpret_cast<const uint2*>(o);
    v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
    v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[e] *= p.alpha;
    if (p.out_act == 1) v[e] = silu_f(v[e]);
    else if (p.out_act == 2) v[e] = tanhf(v[e]);
    else if (p.out_act == 3) v[e] = v[e] > 0.f ? v[e] : v[e] * p.out_slope;
  }
  uint2 pk;
  pk.x = pack2bf(v[0], v[1]);
  pk.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(o) = pk;
  if (gn_acc) {   // statistics of the values as stored (bf16-rounded), like a separate pass over the tensor would see them
    const float r0 = __uint_as_float(pk.x << 16), r1 = __uint_as_float(pk.x & 0xffff0000u);
    const float r2 = __uint_as_float(pk.y << 16), r3 = __uint_as_float(pk.y & 0xffff0000u);
    gn_acc->x += (r0 + r1) + (r2 + r3);
    gn_acc->y += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
  }
  if (p.out2) {   // leaky_relu of the SAME (bf16-rounded) values
    float w2[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                   __uint_as_float(pk.y & 0xffff0000u)};
#pragma unroll
    for (int e = 0; e < 4; ++e) w2[e] = w2[e] > 0.f ? w2[e] : w2[e] * p.out2_slope;
    uint2 pk2;
    pk2.x = pack2bf(w2[0], w2[1]);
    pk2.y = pack2bf(w2[2], w2[3]);
    *reinterpret_cast<uint2*>(p.out2 + oidx) = pk2;
  }
}

// GLDS = true: tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no
// ds_write): the LDS image is lane-linear per wave instruction (8 rows x 128 B for BK = 64), so the
// XOR swizzle is applied to the per-lane SOURCE chunk instead of the destination; padding / tail
// chunks read a zero page.  GLDS = false stages through registers (needed for in_act).
// STAGES > 2 (GLDS only): an S-slot LDS ring with S-1 tiles in flight; the wait for tile kt is a
// COUNTED s_waitcnt vmcnt((S-2) * loads_per_tile) followed by a raw s_barrier, so younger tiles stay
// in flight across the barrier (a __syncthreads() would drain them: its release carries vmcnt(0)).
// One accumulator fragment (4 consecutive output channels of one pixel) through the fused epilogue.
__device__ __forceinline__ void epilogue_store(const ConvParams& p, const f32x4_t a, int m, int n, int b,
                                               long long mrem, int g, const uint2* res_pre = nullptr) {
  const long long inb = mrem * p.ldc + n + p.out_offset;
  if (p.out_limit > 0 && (inb < 0 || inb >= p.out_limit)) return;
  const size_t oidx = (size_t)((long long)g * p.ogs + (long long)b * p.obs + inb);
  float v[4] = {a[0], a[1], a[2], a[3]};
  {   // acc + (bias + rowvec): the order of epilogue_wide4 / wide_epilogue_fast
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      if (p.scalar_store) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.n) c[r] = p.bias[n + r];
      } else {
        const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
        c[0] = bb.x; c[1] = bb.y; c[2] = bb.z; c[3] = bb.w;
      }
    }
    if (p.rowvec) {
      const float4 rv = *reinterpret_cast<const float4*>(p.rowvec + (size_t)b * p.rowvec_ld + n);
      c[0] += rv.x; c[1] += rv.y; c[2] += rv.z; c[3] += rv.w;
    }
    v[0] += c[0]; v[1] += c[1]; v[2] += c[2]; v[3] += c[3];
  }
  if (p.bias_m) {
    const float bm = p.bias_m[m];
    v[0] += bm; v[1] += bm; v[2] += bm; v[3] += bm;
  }
  if (p.res) {   // res_pre: the caller already fetched the residual (rolled epilogues prefetch a whole chunk)
    const uint2 rr = res_pre ? *res_pre : *reinterpret_cast<const uint2*>(p.res + (size_t)m * p.res_ld + n);
    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
  }
  if (p.scalar_store) {
    // tiny Cout (ldc not a multiple of 4): element-wise fp32 / bf16 stores of the valid channels
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < p.n) {
        float t = v[r] * p.alpha;
        if (p.out_act == 1) t = silu_f(t);
        else if (p.out_act == 2) t = tanhf(t);
        else if (p.out_act == 3) t = t > 0.f ? t : t * p.out_slope;
        if (p.out_f32) reinterpret_cast<float*>(p.out)[oidx + r] = t;
        else reinterpret_cast<bf16_t*>(p.out)[oidx + r] = f2bf(t);
      }
    }
    return;
  }
  if (p.out_f32) {
    float* o = reinterpret_cast<float*>(p.out) + oidx;
    if (p.accumulate) {
      const float4 old = *reinterpret_cast<const float4*>(o);
      v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] *= p.alpha;
      if (p.out_act == 1) v[r] = silu_f(v[r]);
      else if (p.out_act == 2) v[r] = tanhf(v[r]);
      else if (p.out_act == 3) v[r] = v[r] > 0.f ? v[r] : v[r] * p.out_slope;
    }
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    return;
  }
  bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + oidx;
  if (p.accumulate) {
    const uint2 old = *reinterpret_cast<const uint2*>(o);
    v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
    v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] *= p.alpha;
    if (p.out_act == 1) v[r] = silu_f(v[r]);
    else if (p.out_act == 2) v[r] = tanhf(v[r]);
    else if (p.out_act == 3) v[r] = v[r] > 0.f ? v[r] : v[r] * p.out_slope;
  }
  uint2 pk;
  pk.x = pack2bf(v[0], v[1]);
  pk.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(o) = pk;
  if (p.out2) {   // second output: leaky_relu of the SAME (bf16-rounded) values, for the next conv's input
    float w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float q = bf2f(f2bf(v[r]));
      w[r] = q > 0.f ? q : q * p.out2_slope;
    }
    uint2 pk2;
    pk2.x = pack2bf(w[0], w[1]);
    pk2.y = pack2bf(w[2], w[3]);
    *reinterpret_cast<uint2*>(p.out2 + oidx) = pk2;
  }
}

