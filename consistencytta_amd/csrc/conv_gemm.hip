// Implicit-GEMM convolution / linear / batched matmul on the gfx950 matrix cores.
//
//   out[m][n] = alpha * ( sum_k X[m][k] * W[n][k] + bias[n] + rowvec[b(m)][n] + res[m][n] )
//
// m runs over output pixels (b, oh, ow) of an NHWC bf16 tensor, k over (kh, kw, c) of the
// receptive field (gathered on the fly: zero padding, stride, dilation, optional x2 nearest
// upsample, optional two-source channel concat), n over output channels.  W is pre-packed
// bf16 [n][k_pad] (K contiguous), so both MFMA operands are K-contiguous 16-byte fragments.
//
// This one kernel family is every conv2d / conv1d / ConvTranspose1d / Linear / QK^T / PV of
// the reference path:
//   F.conv2d   resnet.py:549-597, modules.py:155-175,  Upsample2D resnet.py:126-161 (fused),
//   F.conv1d / conv_transpose1d  hifigan/models.py:56-63,101-117 (ConvTranspose1d is run as
//   `stride` phase-convolutions written through an output remap),
//   F.linear   attention.py:276-334, attention_processor.py:1107-1136, torch.bmm modules.py:204-230.
//
// CDNA4 mapping: 256-thread workgroups (4 wave64), v_mfma_f32_16x16x32_bf16 with the WEIGHT
// tile as the A operand and the PIXEL tile as the B operand, so each lane ends up holding 4
// consecutive output channels of one pixel (8-byte packed bf16 stores along NHWC's fastest
// axis).  Global -> register -> LDS staging with a 2-deep LDS ring and the next tile's
// global loads issued before the current tile's MFMAs (one barrier per K-step).  LDS rows are
// padded by 16 B to spread ds_read_b128 over banks.
#include "common.h"

#include <stdlib.h>

#include "conv_epilogue.h"

// MODE 0: register-staged tiles (supports in_act).  MODE 1: direct-to-LDS, generic gather (per-lane
// global pointers, zero page).  MODE 2: direct-to-LDS through BUFFER descriptors with the address
// work hoisted out of the K loop: requires ct % BK == 0 (a K-tile never straddles a tap, so tap /
// channel base are wave-uniform scalars), <= 32 taps and one source; per row only a pixel base and
// a tap-validity bitmask are kept, padding chunks are sent out of range (hardware returns zeros),
// weight rows >= n fall outside the descriptor, and the per-step K advance rides in soffset.
// n / d for 0 <= n < 2^31 with inv = floor(2^32 / d) (0xFFFFFFFF for d == 1): the estimate is q or q - 1
__device__ __forceinline__ int fast_div(int n, int d, unsigned inv) {
  unsigned q = __umulhi((unsigned)n, inv);
  if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
  return (int)q;
}
// ---- wide-store epilogue, fast path -------------------------------------------------------------------------
struct WideCtx {
  unsigned char* stg;      // this wave's staging rows in the (dead) LDS ring
  int rsf;                 // staging row stride in bytes
  int frow, nsub;          // MFMA accumulator coordinates of the lane (row within a 16-row fragment, first of its 4 channels)
  int prow, col4;          // read-back coordinates: row within a pass, 4-channel column
  int m_first;             // output row of (chunk 0, pass 0) for this lane
  size_t gofs;             // group offset (elements) into out / out2
  int n_lane;
  float4 bias4, rv4;
};
#define WAVE_LDS_FENCE_() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// out = act((acc + bias + rowvec + res) * alpha) for one wave tile of FM x FN fragments, CJ fragments (CHR rows) per
// staging chunk, RPW rows per read-back pass.  Everything is unrolled and free of divergent control flow, so the waits
// the compiler inserts are exact counts.  Stores and residual loads go through buffer descriptors sized to the M valid
// rows: one VGPR offset per lane, the row advance rides in the scalar offset, rows past M are dropped / read as zero by
// the bounds check.  The residual row of (chunk c + 1, pass i) is requested right after that of (chunk c, pass i) was
// consumed (same registers), i.e. a chunk ahead of its use and before the younger half of chunk c's stores.
// Same operation order as epilogue_wide4.
template <int FM, int FN, int CJ, int CHR, int RPW, bool RES, bool OUT2, bool ACC>
__device__ __forceinline__ void wide_epilogue_fast(const ConvParams& p, f32x4_t (&acc)[FN][FM], const WideCtx& w) {
  constexpr int IT = CHR / RPW;          // read-back passes per chunk
  constexpr int NCH = FM / CJ;           // chunks
  constexpr int QB = (RES || ACC) ? 2 : 4;        // passes read back from LDS at a time
  const float slope = p.out_act == 3 ? p.out_slope : 1.0f;      // max(v, v * 1) == v
  const float alpha = p.alpha, slope2 = p.out2_slope;
  const unsigned rows_bytes = (unsigned)(((long long)(p.M - 1) * p.ldc + p.n) * 2);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<bf16_t*>(p.out) + w.gofs), 0, rows_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs2 = rso, rsr = rso;
  if constexpr (OUT2) rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out2 + w.gofs), 0, rows_bytes, 0x00020000);
  if constexpr (RES)
    rsr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (unsigned)(((long long)(p.M - 1) * p.res_ld + p.n) * 2), 0x00020000);
  const int voff = (w.m_first * p.ldc + w.n_lane) * 2;          // byte offset of the lane's (chunk 0, pass 0) element
  const int roff = RES ? (w.m_first * p.res_ld + w.n_lane) * 2 : 0;
  const int ostep = RPW * p.ldc * 2, rstep = RPW * p.res_ld * 2;   // bytes per read-back pass (wave-uniform)
  u32x2_t rr[RES ? IT : 1], oo[ACC ? IT : 1];      // residual / old-output rows, requested a chunk ahead
  if constexpr (RES) {
#pragma unroll
    for (int it = 0; it < IT; ++it) rr[it] = __builtin_amdgcn_raw_buffer_load_b64(rsr, roff, it * rstep, 0);
  }
  if constexpr (ACC) {
#pragma unroll
    for (int it = 0; it < IT; ++it) oo[it] = __builtin_amdgcn_raw_buffer_load_b64(rso, voff, it * ostep, 0);
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
    for (int jj = 0; jj < CJ; ++jj)
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const f32x4_t a = acc[i][ch * CJ + jj];
        *reinterpret_cast<float4*>(w.stg + (jj * 16 + w.frow) * w.rsf + (i * 16 + w.nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
      }
    WAVE_LDS_FENCE_();
#pragma unroll
    for (int h = 0; h < IT; h += QB) {
      float4 q[QB];
#pragma unroll
      for (int e = 0; e < QB; ++e)
        if (h + e < IT) q[e] = *reinterpret_cast<const float4*>(w.stg + (w.prow + (h + e) * RPW) * w.rsf + w.col4 * 16);
#pragma unroll
      for (int e = 0; e < QB; ++e) {
        if (h + e < IT) {
          const int it = h + e;
          const int pass = ch * IT + it;       // rows advance by RPW per pass: CHR == IT * RPW
          float v[4] = {q[e].x + w.bias4.x, q[e].y + w.bias4.y, q[e].z + w.bias4.z, q[e].w + w.bias4.w};
          v[0] += w.rv4.x; v[1] += w.rv4.y; v[2] += w.rv4.z; v[3] += w.rv4.w;
          if constexpr (RES) {
            const u32x2_t r2 = rr[it];
            if (ch + 1 < NCH) rr[it] = __builtin_amdgcn_raw_buffer_load_b64(rsr, roff, (pass + IT) * rstep, 0);
            v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
            v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
          }
          if constexpr (ACC) {
            const u32x2_t o2 = oo[it];
            if (ch + 1 < NCH) oo[it] = __builtin_amdgcn_raw_buffer_load_b64(rso, voff, (pass + IT) * ostep, 0);
            v[0] += __uint_as_float(o2.x << 16); v[1] += __uint_as_float(o2.x & 0xffff0000u);
            v[2] += __uint_as_float(o2.y << 16); v[3] += __uint_as_float(o2.y & 0xffff0000u);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) { v[c] *= alpha; v[c] = fmaxf(v[c], v[c] * slope); }
          u32x2_t pk;
          pk.x = pack2bf(v[0], v[1]);
          pk.y = pack2bf(v[2], v[3]);
          __builtin_amdgcn_raw_buffer_store_b64(pk, rso, voff, pass * ostep, 0);
          if constexpr (OUT2) {
            float w2[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                           __uint_as_float(pk.y & 0xffff0000u)};
#pragma unroll
            for (int c = 0; c < 4; ++c) w2[c] = fmaxf(w2[c], w2[c] * slope2);
            u32x2_t pk2;
            pk2.x = pack2bf(w2[0], w2[1]);
            pk2.y = pack2bf(w2[2], w2[3]);
            __builtin_amdgcn_raw_buffer_store_b64(pk2, rs2, voff, pass * ostep, 0);
          }
        }
      }
    }
    if (ch + 1 < NCH) WAVE_LDS_FENCE_();     // the staging rows are rewritten by the next chunk
  }
}

// orders a wave's own LDS writes before its LDS reads (other lanes of the SAME wave) without a workgroup barrier
#define WAVE_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <int BM, int BN, int BK, int WM, int WN, int MODE, int STAGES>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_gemm_kernel(ConvParams p) {
  constexpr bool GLDS = MODE != 0;
  static_assert(STAGES == 2 || GLDS, "multi-stage ring needs the direct-to-LDS path");
  constexpr int NT = 64 * WM * WN;
  constexpr int LDK = BK;              // bf16 elements per LDS row: no padding, XOR-swizzled chunks
  // 16-byte chunk c of row r lives at chunk position c ^ swz(r): conflict-free for the 16-lane
  // groups of ds_read_b128 (rows r..r+15 at one logical chunk) and for the row-contiguous writes.
  constexpr int SWZ_SHIFT = (BK == 64) ? 0 : 1;
  constexpr int SWZ_MASK = BK / 8 - 1;
  constexpr int CPR = BK / 8;          // 16-byte chunks per row
  constexpr int RPP = NT / CPR;        // rows staged per pass
  constexpr int XP = BM / RPP;
  constexpr int WP = BN / RPP;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/pass mismatch");
  static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem_raw);               // [STAGES][BM][LDK]
  bf16_t* Ws = Xs + STAGES * BM * LDK;                             // [STAGES][BN][LDK]
  unsigned long long* stamp = nullptr;
  if (p.stamps) {
    stamp = p.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 6;
    if (threadIdx.x == 0) { stamp[0] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32); stamp[1] = __builtin_amdgcn_s_memtime(); }
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int mt = blockIdx.x, nt = blockIdx.y;
  if (p.xcd_per > 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    if (p.n_inner) { nt = local % p.n_tiles; mt = xcd * p.xcd_per + local / p.n_tiles; }
    else { nt = local / p.xcd_per; mt = xcd * p.xcd_per + local % p.xcd_per; }
    if (mt >= p.m_tiles) return;   // padding blocks of the last XCD range (whole workgroup, before any barrier)
  }
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  // a wave whose TM rows all lie past M (the last row tile of M = k * BM + a few rows) skips its MFMAs: its SIMD partner
  // then runs at full matrix-pipe rate and the tail tile of a one-workgroup-per-CU launch takes about half a tile time
  const bool wave_live = m0 + (wave / WN) * TM < p.M;
  const int zs = blockIdx.z;
  const int g = p.ksplit > 1 ? 0 : zs;                    // group index (pointer offsets)
  const int kt_begin = p.ksplit > 1 ? zs * p.nk_split : 0;
  const int nk = p.ksplit > 1 ? min(p.nk - kt_begin, p.nk_split) : p.nk;

  const bf16_t* x0 = p.x0 + (size_t)g * p.xgs;
  const bf16_t* x1 = p.x1;
  const bf16_t* wbase = p.w + (size_t)g * p.wgs;

  const int r0 = tid / CPR;
  // logical 16-byte K chunk this thread fetches: register path -> position tid % CPR (swizzled on
  // store); direct-to-LDS path -> the chunk whose swizzled home is position tid % CPR
  const int kc = GLDS ? ((tid % CPR) ^ ((r0 >> SWZ_SHIFT) & SWZ_MASK)) : (tid % CPR);
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int ROWS_PER_INSTR = 64 / CPR;   // rows one wave-wide 1 KiB LDS-DMA covers

  // per-thread K state (shared by all of this thread's rows)
  int c, tap, kh, kw;
  {
    int kk = kc * 8 + kt_begin * BK;
    tap = kk / p.ct;
    c = kk - tap * p.ct;
    kh = tap / p.kw;
    kw = tap - kh * p.kw;
  }
  // per-row pixel state
  int rb[XP], rih[XP], riw[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    int m = m0 + r0 + i * RPP;
    if (m < p.M) {
      int b = m / p.howo;
      int rem = m - b * p.howo;
      int oh = rem / p.wo;
      int ow = rem - oh * p.wo;
      rb[i] = b;
      rih[i] = oh * p.sh - p.ph;
      riw[i] = ow * p.sw - p.pw;
    } else {
      rb[i] = 0;
      rih[i] = -(1 << 28);
      riw[i] = 0;
    }
  }
  // per-row weight pointers
  const bf16_t* wrow[WP];
  bool wok[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    int n = n0 + r0 + j * RPP;
    wok[j] = n < p.n;
    wrow[j] = wbase + (size_t)(wok[j] ? n : 0) * p.k_pad + kc * 8;
  }

  uint4 xr[XP], wr[WP];

  auto load_tile = [&](int kt) {
    const int ihk = kh * p.dh, iwk = kw * p.dw;
    const bool tap_ok = tap < p.taps;
    const bool second = c >= p.c0;
    const bf16_t* src = second ? x1 : x0;
    const int cs = second ? p.c1 : p.xs0;
    const int cc = second ? c - p.c0 : c;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int ih = rih[i] + ihk, iw = riw[i] + iwk;
      bool ok = tap_ok && (unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi;
      if (p.ups) { ih >>= 1; iw >>= 1; }
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        size_t pix = (size_t)(rb[i] * p.hs + ih) * p.ws + iw;
        v = *reinterpret_cast<const uint4*>(src + pix * cs + cc);
      }
      xr[i] = v;
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (wok[j]) v = *reinterpret_cast<const uint4*>(wrow[j] + (size_t)kt * BK);
      wr[j] = v;
    }
    // advance K state
    c += BK;
    while (c >= p.ct) {
      c -= p.ct;
      ++tap;
      if (++kw == p.kw) { kw = 0; ++kh; }
    }
  };

  auto store_tile = [&](int buf) {
    bf16_t* xs = Xs + buf * BM * LDK;
    bf16_t* ws = Ws + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      uint4 v = xr[i];
      if (p.in_act == 1) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * p.in_slope;
        v = pack8(f);
      }
      const int r = r0 + i * RPP;
      *reinterpret_cast<uint4*>(xs + r * LDK + ((kc ^ ((r >> SWZ_SHIFT) & SWZ_MASK)) * 8)) = v;
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int r = r0 + j * RPP;
      *reinterpret_cast<uint4*>(ws + r * LDK + ((kc ^ ((r >> SWZ_SHIFT) & SWZ_MASK)) * 8)) = wr[j];
    }
  };

  // direct global -> LDS issue of K-tile kt into ring slot buf (same row/chunk ownership as above)
  auto issue_tile = [&](int kt, int buf) {
    const int ihk = kh * p.dh, iwk = kw * p.dw;
    const bool tap_ok = tap < p.taps;
    const bool second = c >= p.c0;
    const bf16_t* src = second ? x1 : x0;
    const int cs = second ? p.c1 : p.xs0;
    const int cc = second ? c - p.c0 : c;
    bf16_t* xs = Xs + buf * BM * LDK + wave_u * ROWS_PER_INSTR * LDK;
    bf16_t* ws = Ws + buf * BN * LDK + wave_u * ROWS_PER_INSTR * LDK;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      int ih = rih[i] + ihk, iw = riw[i] + iwk;
      const bool ok = tap_ok && (unsigned)ih < (unsigned)p.hi && (unsigned)iw < (unsigned)p.wi;
      if (p.ups) { ih >>= 1; iw >>= 1; }
      const bf16_t* g = p.zero;
      if (ok) g = src + ((size_t)(rb[i] * p.hs + ih) * p.ws + iw) * cs + cc;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(xs + i * RPP * LDK), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const bf16_t* g = wok[j] ? wrow[j] + (size_t)kt * BK : p.zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(ws + j * RPP * LDK), 16, 0, 0);
    }
    c += BK;
    while (c >= p.ct) {
      c -= p.ct;
      ++tap;
      if (++kw == p.kw) { kw = 0; ++kh; }
    }
  };

  // ---------------- MODE 2: descriptor path, K-loop-invariant address work hoisted
  unsigned fvoff[XP], fmask[XP], fph[XP], fpw[XP], fwoff[WP];
  int fih0[XP], fiw0[XP];
  int ftap = 0, fkh = 0, fkw = 0, fcb = 0;   // wave-uniform K state (tap index, its (kh,kw), channel base)
  __amdgpu_buffer_rsrc_t rsx, rsw;
  if constexpr (MODE == 2) {
    // K order on this path: taps INNERMOST -- K-step s = (channel chunk s / taps, tap s % taps).  All taps of one
    // 64-channel chunk re-read the same (BM + halo) x BK footprint (tens of KB: L1/L2 resident); with channels
    // innermost the whole (BM + halo) x C footprint of every resident workgroup has to survive between taps, which
    // overflows the 4 MB L2 of an XCD (measured with FETCH_SIZE: 10-26x the input tensor per launch on the
    // 256/512-channel convs, i.e. fabric-bound at 3.5 TB/s).
    if (kt_begin > 0) {   // split-K: start inside the K range
      const int chunk = kt_begin / p.taps;
      ftap = kt_begin - chunk * p.taps;
      fcb = chunk * BK;
      fkh = ftap / p.kw;
      fkw = ftap - fkh * p.kw;
    }
    rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x0, 0, p.x_bytes, 0x00020000);
    rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, p.w_bytes, 0x00020000);
    // Only what the FIRST K tile needs is computed here (pixel offsets and the validity of its one tap); the tap-validity
    // masks of all taps follow in full_masks(), called right after that tile's loads are in flight, so that this ALU
    // work (1.85 us per tile, tools/tile_timeline.py) runs inside the first load's latency instead of in front of it.
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int m = m0 + r0 + i * RPP;
      unsigned mask = 0;
      int pix = 0, ph = 0, pw = 0, ih0 = -(1 << 28), iw0 = 0;
      if (m < p.M) {
        const int b = fast_div(m, p.howo, p.howo_inv);
        const int rem = m - b * p.howo;
        const int oh = fast_div(rem, p.wo, p.wo_inv);
        const int ow = rem - oh * p.wo;
        ih0 = oh * p.sh - p.ph; iw0 = ow * p.sw - p.pw;
        if ((unsigned)(ih0 + fkh * p.dh) < (unsigned)p.hi && (unsigned)(iw0 + fkw * p.dw) < (unsigned)p.wi) mask = 1u << ftap;
        if (p.ups) {   // 3x3 / pad 1 / stride 1 on a x2 nearest-upsampled source: base = (oh>>1, ow>>1)
          pix = (b * p.hs + (oh >> 1)) * p.ws + (ow >> 1);
          ph = oh & 1; pw = ow & 1;
        } else {
          pix = (b * p.hs + ih0) * p.ws + iw0;
        }
      }
      fih0[i] = ih0; fiw0[i] = iw0;
      fmask[i] = mask;
      fvoff[i] = (unsigned)(pix * p.xs0 + kc * 8) * 2u;      // bytes; wraps correctly for border rows
      fph[i] = ph ? (unsigned)(p.ws * p.xs0) * 2u : 0u;
      fpw[i] = pw ? (unsigned)p.xs0 * 2u : 0u;
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) fwoff[j] = (unsigned)((n0 + r0 + j * RPP) * p.k_pad + kc * 8) * 2u;
  }
  // valid taps are a rectangle: rows a with 0 <= ih0 + a*dh < hi  x  columns bq with 0 <= iw0 + bq*dw < wi.  Undilated
  // axes get their bit range by arithmetic, dilated ones by a loop over that axis only (rows past M carry ih0 = -2^28:
  // empty ranges).
  auto full_masks = [&]() {
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int ih0 = fih0[i], iw0 = fiw0[i];
      unsigned hm = 0, wmk = 0, mask = 0;
      if (p.dh == 1) {
        const int lo = max(0, -ih0), hi_ = min(p.kh - 1, p.hi - 1 - ih0);
        if (hi_ >= lo) hm = (2u << hi_) - (1u << lo);
      } else {
        for (int a = 0; a < p.kh; ++a) if ((unsigned)(ih0 + a * p.dh) < (unsigned)p.hi) hm |= 1u << a;
      }
      if (p.dw == 1) {
        const int lo = max(0, -iw0), hi_ = min(p.kw - 1, p.wi - 1 - iw0);
        if (hi_ >= lo) wmk = (2u << hi_) - (1u << lo);
      } else {
        for (int bq = 0; bq < p.kw; ++bq) if ((unsigned)(iw0 + bq * p.dw) < (unsigned)p.wi) wmk |= 1u << bq;
      }
      for (int a = 0; a < p.kh; ++a) if ((hm >> a) & 1u) mask |= wmk << (a * p.kw);
      fmask[i] = mask;
    }
  };
  auto issue_fast = [&](int kt, int buf) {
    bf16_t* xs = Xs + buf * BM * LDK + wave_u * ROWS_PER_INSTR * LDK;
    bf16_t* ws = Ws + buf * BN * LDK + wave_u * ROWS_PER_INSTR * LDK;
    unsigned s_u;          // uniform byte offset of (tap, channel base)
    bool use_h = false, use_w = false;
    if (p.ups) {           // dy = (kh==0 ? ph-1 : kh==1 ? 0 : ph), same for dx
      const int bh = fkh == 0 ? -1 : 0, bw = fkw == 0 ? -1 : 0;
      use_h = fkh != 1; use_w = fkw != 1;
      s_u = (unsigned)((bh * p.ws + bw) * p.xs0 + fcb) * 2u;
    } else {
      s_u = (unsigned)((fkh * p.dh * p.ws + fkw * p.dw) * p.xs0 + fcb) * 2u;
    }
    const unsigned tbit = 1u << ftap;
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      unsigned v = fvoff[i] + s_u;
      if (use_h) v += fph[i];
      if (use_w) v += fpw[i];
      v = (fmask[i] & tbit) ? v : 0xFFFFFFF0u;   // out of range -> the buffer unit returns zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(xs + i * RPP * LDK), 16,
                                               (int)v, 0, 0, 0);
    }
    const int soff = (ftap * p.ct + fcb) * 2;   // weight columns stay (tap, channel)-ordered; only the walk changes
#pragma unroll
    for (int j = 0; j < WP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(ws + j * RPP * LDK), 16,
                                               (int)fwoff[j], soff, 0, 0);
    ++ftap;
    if (++fkw == p.kw) { fkw = 0; ++fkh; }
    if (ftap == p.taps) { ftap = 0; fkh = 0; fkw = 0; fcb += BK; }
  };

  f32x4_t acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fchunk = lane >> 4;                       // logical 16-byte chunk within a 32-wide k-slab
  const int fswz = (frow >> SWZ_SHIFT) & SWZ_MASK;    // tile/frag row offsets are multiples of 16

  auto compute_tile = [&](int buf) {
    const bf16_t* xs = Xs + buf * BM * LDK + (wm * TM + frow) * LDK;
    const bf16_t* ws = Ws + buf * BN * LDK + (wn * TN + frow) * LDK;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t af[FN], bfr[FM];
      const int koff = (((ks * 4 + fchunk) ^ fswz) & SWZ_MASK) * 8;
#pragma unroll
      for (int i = 0; i < FN; ++i)
        af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ws + i * 16 * LDK + koff));
#pragma unroll
      for (int j = 0; j < FM; ++j)
        bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xs + j * 16 * LDK + koff));
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  if constexpr (STAGES > 2) {
    constexpr int LPT = XP + WP;                 // LDS-DMA instructions per wave per tile
    constexpr int INFLIGHT = (STAGES - 2) * LPT; // what may stay outstanding while tile kt is consumed
    int issued = 0;
    for (; issued < STAGES - 1 && issued < nk; ++issued) {
      if constexpr (MODE == 2) { issue_fast(kt_begin + issued, issued); if (issued == 0) full_masks(); }
      else issue_tile(kt_begin + issued, issued);
    }
    int slot = 0, fill = issued % STAGES;
    for (int kt = 0; kt < nk; ++kt) {
      if (issued - kt - 1 >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // tile kt landed for every wave; slot `fill` is free
      if (issued < nk) {
        if constexpr (MODE == 2) issue_fast(kt_begin + issued, fill); else issue_tile(kt_begin + issued, fill);
        ++issued;
        fill = (fill + 1 == STAGES) ? 0 : fill + 1;
      }
      if (wave_live) compute_tile(slot);
      slot = (slot + 1 == STAGES) ? 0 : slot + 1;
    }
  } else {
    if (stamp && threadIdx.x == 0) stamp[5] = __builtin_amdgcn_s_memtime();
    if constexpr (MODE == 2) {
      issue_fast(kt_begin, 0);
      full_masks();
    } else if constexpr (MODE == 1) {
      issue_tile(kt_begin, 0);
    } else {
      load_tile(kt_begin);
      store_tile(0);
    }
    __syncthreads();
    if (stamp && threadIdx.x == 0) stamp[2] = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) {
        if constexpr (MODE == 2) issue_fast(kt_begin + kt + 1, buf ^ 1);
        else if constexpr (MODE == 1) issue_tile(kt_begin + kt + 1, buf ^ 1);
        else load_tile(kt_begin + kt + 1);
      }
      if (wave_live) compute_tile(buf);
      if constexpr (!GLDS) {
        if (kt + 1 < nk) store_tile(buf ^ 1);
      }
      __syncthreads();
    }
  }

  if (stamp && threadIdx.x == 0) stamp[3] = __builtin_amdgcn_s_memtime();
  // ---- epilogue: lane holds n = nb + (lane>>4)*4 + {0..3} (rows of D), m = mb + (lane&15)
  const int nsub = (lane >> 4) * 4;
  // Wide-store epilogue (bf16 output, plain row-major destination).  In the MFMA layout a store instruction writes
  // 16 rows x 32 bytes; measured, a launch then pays ~0.5 us per MB of output on top of its main loop (335 MB outputs:
  // 180 of 300 us of K-independent time) because workgroups cannot retire before their scattered stores drain.  Here
  // every wave transposes its tile through the (dead) LDS ring in fp32, CHR rows at a time, and reads it back with
  // TN/4 consecutive lanes per output row: stores -- and the residual reads -- are whole TN*2-byte row segments, the
  // bias sits in registers because a lane keeps its 4 channels, and the arithmetic runs in one rolled loop.
  if (p.wide_store) {
    constexpr int NW = WM * WN;
    constexpr size_t RING = (size_t)STAGES * (BM + BN) * BK * 2;
    constexpr int RSF = TN * 4 + 16;                       // staging row stride (bytes): +16 keeps 16-byte accesses conflict-free
    constexpr int CHR = (FM % 2 == 0 && (size_t)NW * 32 * RSF <= RING) ? 32 : 16;   // rows per chunk
    static_assert((size_t)NW * CHR * RSF <= RING, "wide-store staging does not fit the ring");
    constexpr int CJ = CHR / 16;
    constexpr int LPR = TN / 4;                            // lanes per output row (4 channels each)
    constexpr int RPW = 64 / LPR;                          // rows per pass of the wave
    unsigned char* stg = smem_raw + (size_t)wave * CHR * RSF;
    const int col4 = lane % LPR, prow = lane / LPR;
    const int n_lane = n0 + wn * TN + col4 * 4;
    const bool n_ok = n_lane < p.n;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) bias4 = *reinterpret_cast<const float4*>(p.bias + n_lane);
    __syncthreads();   // every wave is done with the ring
    if constexpr (FM * FN <= 16 && FN % 2 == 0) {
      if (p.out_act == 4) {   // fused GEGLU: hidden unit hl of the wave's row = value column (hl/16)*32 + hl%16, gate 16 further
        constexpr int LPG = TN / 8, RPG = 64 / LPG;          // lanes per output row (4 hidden units each), rows per pass
        const int h4 = lane % LPG, grow = lane / LPG;
        const int vcol = (h4 / 4) * 32 + (h4 % 4) * 4;       // value columns of this lane within the wave tile
        const int n_val = n0 + wn * TN + vcol;
        const bool g_ok = n_val < p.n;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
        if (p.bias && g_ok) {
          bv = *reinterpret_cast<const float4*>(p.bias + n_val);
          bg = *reinterpret_cast<const float4*>(p.bias + n_val + 16);
        }
        bf16_t* og = reinterpret_cast<bf16_t*>(p.out) + (size_t)g * p.ogs + ((n0 + wn * TN) >> 1) + h4 * 4;
        if (p.epi_fast_geglu && n0 + wn * TN + TN <= p.n) {
          // straight-line variant (see wide_epilogue_fast): buffer stores bounded to the M valid rows, row advance in the
          // scalar offset, nothing predicated, wave-local LDS ordering only
          constexpr int ITG = CHR / RPG;
          const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(reinterpret_cast<bf16_t*>(p.out) + (size_t)g * p.ogs), 0,
              (unsigned)(((long long)(p.M - 1) * p.ldc + (p.n >> 1)) * 2), 0x00020000);
          const int voff = ((m0 + wm * TM + grow) * p.ldc + ((n0 + wn * TN) >> 1) + h4 * 4) * 2;
          const int ostep = RPG * p.ldc * 2;
#pragma unroll
          for (int j0 = 0; j0 < FM; j0 += CJ) {
#pragma unroll
            for (int jj = 0; jj < CJ; ++jj)
#pragma unroll
              for (int i = 0; i < FN; ++i) {
                const f32x4_t a = acc[i][j0 + jj];
                *reinterpret_cast<float4*>(stg + (jj * 16 + frow) * RSF + (i * 16 + nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
              }
            WAVE_LDS_FENCE();
#pragma unroll
            for (int it = 0; it < ITG; ++it) {
              const int r = grow + it * RPG;
              const float4 qv = *reinterpret_cast<const float4*>(stg + r * RSF + vcol * 4);
              const float4 qg = *reinterpret_cast<const float4*>(stg + r * RSF + (vcol + 16) * 4);
              const float vv[4] = {qv.x + bv.x, qv.y + bv.y, qv.z + bv.z, qv.w + bv.w};
              const float gg[4] = {qg.x + bg.x, qg.y + bg.y, qg.z + bg.z, qg.w + bg.w};
              float o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = vv[e] * (0.5f * gg[e] * (1.0f + erff(gg[e] * 0.70710678118654752f)));
              u32x2_t pk;
              pk.x = pack2bf(o[0], o[1]);
              pk.y = pack2bf(o[2], o[3]);
              __builtin_amdgcn_raw_buffer_store_b64(pk, rso, voff, ((j0 / CJ) * ITG + it) * ostep, 0);
            }
            if (j0 + CJ < FM) WAVE_LDS_FENCE();
          }
          return;
        }
#pragma unroll
        for (int j0 = 0; j0 < FM; j0 += CJ) {
#pragma unroll
          for (int jj = 0; jj < CJ; ++jj)
#pragma unroll
            for (int i = 0; i < FN; ++i) {
              const f32x4_t a = acc[i][j0 + jj];
              *reinterpret_cast<float4*>(stg + (jj * 16 + frow) * RSF + (i * 16 + nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
            }
          if (p.epi_barrier) __syncthreads(); else WAVE_LDS_FENCE();   // wave-private staging rows
#pragma unroll 2
          for (int r = grow; r < CHR; r += RPG) {
            const int m = m0 + wm * TM + j0 * 16 + r;
            if (m < p.M && g_ok) {
              const float4 qv = *reinterpret_cast<const float4*>(stg + r * RSF + vcol * 4);
              const float4 qg = *reinterpret_cast<const float4*>(stg + r * RSF + (vcol + 16) * 4);
              const float vv[4] = {qv.x + bv.x, qv.y + bv.y, qv.z + bv.z, qv.w + bv.w};
              const float gg[4] = {qg.x + bg.x, qg.y + bg.y, qg.z + bg.z, qg.w + bg.w};
              float o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = vv[e] * (0.5f * gg[e] * (1.0f + erff(gg[e] * 0.70710678118654752f)));
              uint2 pk;
              pk.x = pack2bf(o[0], o[1]);
              pk.y = pack2bf(o[2], o[3]);
              *reinterpret_cast<uint2*>(og + (size_t)m * p.ldc) = pk;
            }
          }
          if (j0 + CJ < FM) { if (p.epi_barrier) __syncthreads(); else WAVE_LDS_FENCE(); }
        }
        return;
      }
    }
    // Fast path (bias [+ per-sample row vector] [+ residual] [+ LeakyReLU] [+ second LeakyReLU output], plain destination):
    // straight-line code so that the waits the compiler inserts are exact -- inside the rolled generic loop below every
    // iteration waits for all but one outstanding store (vmcnt counts stores too, and the back edge makes the count
    // conservative), which serialises the epilogue on the store round trip: 16.7 us per 256x256 tile measured with
    // s_memtime stamps (tools/tile_timeline.py) against 76 us of main loop at K = 2816.
    // (per wave: its TN columns must lie inside the matrix -- no predicates, no divergent control flow, which would also
    // make the compiler's wait counts conservative; rows past M fall to the buffer bounds check; edge waves take the
    // generic loop, there is no barrier)
    if (p.epi_fast && n0 + wn * TN + TN <= p.n) {
      const bool one_sample = m0 / p.howo == (min(m0 + BM, p.M) - 1) / p.howo;
      if (!p.rowvec || one_sample) {
        float4 rv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.rowvec) rv4 = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m0 / p.howo) * p.rowvec_ld + n_lane);
        const WideCtx wc = {stg, RSF, frow, nsub, prow, col4, m0 + wm * TM + prow, (size_t)g * p.ogs, n_lane, bias4, rv4};
        if (p.accumulate) {
          if (!p.res) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, true>(p, acc, wc);
          else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, false, true>(p, acc, wc);
        } else if (!p.res) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, false>(p, acc, wc);
        else if (!p.out2) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, false, false>(p, acc, wc);
        else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, true, false>(p, acc, wc);
        if (stamp && threadIdx.x == 0) stamp[4] = __builtin_amdgcn_s_memtime();
        return;
      }
    }
    float2 gacc = make_float2(0.f, 0.f);   // always passed (a conditional pointer would force it into scratch)
#pragma unroll
    for (int j0 = 0; j0 < FM; j0 += CJ) {
#pragma unroll
      for (int jj = 0; jj < CJ; ++jj)
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const f32x4_t a = acc[i][j0 + jj];
          *reinterpret_cast<float4*>(stg + (jj * 16 + frow) * RSF + (i * 16 + nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
      // the staging rows are private to the wave and a wave's LDS operations execute in order: no workgroup barrier
      if (p.epi_barrier) __syncthreads(); else WAVE_LDS_FENCE();
#pragma unroll 2
      for (int r = prow; r < CHR; r += RPW) {
        const int m = m0 + wm * TM + j0 * 16 + r;
        if (m < p.M && n_ok)
          epilogue_wide4(p, *reinterpret_cast<const float4*>(stg + r * RSF + col4 * 16), bias4, m, n_lane, (size_t)g * p.ogs,
                         &gacc);
      }
      if (j0 + CJ < FM) { if (p.epi_barrier) __syncthreads(); else WAVE_LDS_FENCE(); }
    }
    if (stamp && threadIdx.x == 0) stamp[4] = __builtin_amdgcn_s_memtime();
    if (p.gn_part) {
      // GroupNorm statistics.  Inside a wave the lanes of one channel group (cw / 4 neighbouring column lanes x all row
      // lanes) fold their (sum, sum of squares) with a fixed butterfly, and the group's leader lane writes the WAVE's
      // partial straight to global memory: one "chunk" per (row tile, wave row, column slice of a group wider than the
      // wave), no LDS, no barrier, no atomics (bit-reproducible).  gn_finalize_kernel adds the chunks.
      const int cpg = p.gn_cpg;
      const int cw = cpg < TN ? cpg : TN;        // channels of one group inside a wave's TN columns
      const int LG = cw / 4;                     // column lanes per group (power of two)
      const int SUB = cpg / cw;                  // wave columns a group spans (1 unless cpg > TN)
      float s1 = gacc.x, s2 = gacc.y;
      for (int o = 1; o < LG; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      if (prow == 0 && col4 % LG == 0 && n_ok) {
        const int b = m0 / p.gn_hw;
        const int chunk = (((m0 - b * p.gn_hw) / BM) * WM + wm) * SUB + (wn % SUB);
        const int grp = n_lane / cpg;
        float* dst = p.gn_part + (((size_t)b * p.gn_nchunk + chunk) * p.gn_G + grp) * 2;
        *reinterpret_cast<float2*>(dst) = make_float2(s1, s2);
      }
    }
    return;
  }
  // fused GEGLU: only compiled into the small-fragment tiles (a longer epilogue on the 16-fragment tiles
  // pushes their accumulators into scratch)
  if constexpr (FM * FN <= 8 && FN % 2 == 0) {
    if (p.out_act == 4) {
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int m = m0 + wm * TM + j * 16 + frow;
#pragma unroll
        for (int i = 0; i < FN; i += 2) {
          const int n = n0 + wn * TN + i * 16 + nsub;
          if (m < p.M && n < p.n) epilogue_geglu(p, acc[i][j], acc[i + 1][j], m, n);
        }
      }
      return;
    }
  }
  if constexpr (FM * FN <= 16) {
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + wm * TM + j * 16 + frow;
      const bool m_ok = m < p.M;
      const int b = m_ok ? m / p.howo : 0;
      const long long mrem = m - (long long)b * p.howo;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int n = n0 + wn * TN + i * 16 + nsub;
        if (m_ok && n < p.n) epilogue_store(p, acc[i][j], m, n, b, mrem, zs);
      }
    }
  } else {
    // Large wave tiles: unrolling the generic epilogue once per fragment would blow the code size (and
    // a rolled loop cannot index registers), so fragments bounce through thread-private LDS slots in
    // chunks of 8 and a ROLLED loop runs the epilogue on them.  The ring is dead by now.
    constexpr size_t RING = (size_t)STAGES * (BM + BN) * BK * 2;
    constexpr int CH = (size_t)8 * NT * 24 <= RING ? 8 : 4;
    static_assert((FM * FN) % CH == 0 && (size_t)CH * NT * 24 <= RING, "stage size");
    __syncthreads();
    float4* stage = reinterpret_cast<float4*>(smem_raw);
    uint2* rstage = reinterpret_cast<uint2*>(stage + CH * NT);
#pragma unroll
    for (int c0_ = 0; c0_ < FM * FN; c0_ += CH) {
      // unrolled: park the fragments AND issue all residual loads of the chunk back to back -- inside the rolled
      // loop below each load would expose its full latency (measured: -40 % on the 256-wide tiles with a residual)
#pragma unroll
      for (int f = 0; f < CH; ++f) {
        const int idx = c0_ + f;
        const f32x4_t a = acc[idx / FM][idx % FM];
        stage[f * NT + tid] = make_float4(a[0], a[1], a[2], a[3]);
        if (p.res) {
          const int m = m0 + wm * TM + (idx % FM) * 16 + frow;
          const int n = n0 + wn * TN + (idx / FM) * 16 + nsub;
          uint2 rr = make_uint2(0, 0);
          if (m < p.M && n < p.n) rr = *reinterpret_cast<const uint2*>(p.res + (size_t)m * p.res_ld + n);
          rstage[f * NT + tid] = rr;
        }
      }
#pragma unroll 1
      for (int f = 0; f < CH; ++f) {
        const int idx = c0_ + f;
        const int i = idx / FM, j = idx % FM;
        const float4 q = stage[f * NT + tid];
        const uint2 rr = p.res ? rstage[f * NT + tid] : make_uint2(0, 0);
        const int m = m0 + wm * TM + j * 16 + frow;
        const int n = n0 + wn * TN + i * 16 + nsub;
        if (m < p.M && n < p.n) {
          const int b = m / p.howo;
          epilogue_store(p, (f32x4_t){q.x, q.y, q.z, q.w}, m, n, b, m - (long long)b * p.howo, zs, &rr);
        }
      }
    }
  }
}

// Fused GEGLU epilogue (out_act 4): the weight rows are packed in 16-row blocks [16 value][16 gate], so fragment
// i (even) holds the values and fragment i+1 the gates of the same 16 hidden units for the same lane positions:
// out[m][h] = (val + bias_v) * gelu(gate + bias_g), h = (n / 32) * 16 + n % 16, written to a matrix of HALF the
// GEMM width.  Replaces attention.py:430-432 (proj -> chunk(2) -> value * gelu(gate)) without the round trip.
__device__ __forceinline__ void epilogue_geglu(const ConvParams& p, const f32x4_t av, const f32x4_t ag, int m, int n) {
  float v[4] = {av[0], av[1], av[2], av[3]}, g[4] = {ag[0], ag[1], ag[2], ag[3]};
  if (p.bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
    const float4 bg = *reinterpret_cast<const float4*>(p.bias + n + 16);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    g[0] += bg.x; g[1] += bg.y; g[2] += bg.z; g[3] += bg.w;
  }
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = v[r] * (0.5f * g[r] * (1.0f + erff(g[r] * 0.70710678118654752f)));
  const int h = (n >> 5) * 16 + (n & 15);
  uint2 pk;
  pk.x = pack2bf(o[0], o[1]);
  pk.y = pack2bf(o[2], o[3]);
  *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + h) = pk;
}

// ------------------------------------------------------------------------------------------
// 1-D "halo" convolution for the narrowest layers (C = Cin = Cout = 32, stride 1): the HiFi-GAN ResBlock
// convolutions of the last upsampling stage (hifigan/models.py:56-63), 5.2 M positions x 32 channels at B=32.  As an implicit
// GEMM these layers re-gather every input row once per tap from L2 (k = 3/7/11 times) while producing
// only C output channels per row -- the generic kernel is L2-bandwidth bound there.  Here a workgroup
// stages its BL output positions plus the (k-1)*dilation halo ONCE in LDS; the taps are row offsets into
// that tile, the weight fragments (A operands, [n][k_pad] packing shared with conv_gemm) stream straight
// from L1/L2 into registers one (tap, 32-channel chunk) ahead of the MFMAs.  Same MFMA roles and the same
// epilogue as conv_gemm_kernel: weights = A (rows = Cout), positions = B (cols), lane owns 4 consecutive Cout.
template <int C, int BL>
__global__ __launch_bounds__(256) void conv1d_halo_kernel(const ConvParams p) {
  constexpr int RS = C + 8;      // LDS row stride (bf16): C*2 + 16 bytes -> conflict-free ds_read_b128 over 16 rows
  constexpr int NCB = C / 16;    // Cout blocks
  constexpr int NCH = C / 32;    // 32-channel chunks per tap
  constexpr int PB = BL / 64;    // position blocks per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y, l0 = blockIdx.x * BL;
  const int L = p.wo;
  const int nrows = BL + (p.taps - 1) * p.dw;
  const bf16_t* xb = p.x0 + (size_t)b * L * p.xs0;
  for (int idx = tid; idx < nrows * (C / 8); idx += 256) {
    const int r = idx / (C / 8), cc = idx - r * (C / 8);
    const int pos = l0 - p.pw + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((unsigned)pos < (unsigned)L) v = *reinterpret_cast<const uint4*>(xb + (size_t)pos * p.xs0 + cc * 8);
    *reinterpret_cast<uint4*>(xs + r * RS + cc * 8) = v;
  }
  f32x4_t acc[NCB][PB];
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bf16_t* wl = p.w + (size_t)lq * p.k_pad + lg * 8;
  bf16x8_t a_cur[NCB], a_nxt[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
    a_cur[cb] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wl + (size_t)cb * 16 * p.k_pad));
  __syncthreads();
  const bf16_t* xw = xs + (wave * (BL / 4) + lq) * RS + lg * 8;
  const int nsteps = p.taps * NCH;
  for (int st = 0; st < nsteps; ++st) {
    const int tap = st / NCH, ch = st - tap * NCH;
    if (st + 1 < nsteps) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        a_nxt[cb] = __builtin_bit_cast(
            bf16x8_t, *reinterpret_cast<const uint4*>(wl + (size_t)cb * 16 * p.k_pad + (size_t)(st + 1) * 32));
    }
    const bf16_t* xr = xw + (tap * p.dw) * RS + ch * 32;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xr + pb * 16 * RS));
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur[cb], bf, acc[cb][pb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) a_cur[cb] = a_nxt[cb];
  }
  if (p.wide_store) {   // same transposed store as conv_gemm_kernel: 16 positions x C channels per wave at a time
    constexpr int RSF = C * 4 + 16;
    constexpr int LPR = C / 4, RPW = 64 / LPR;
    unsigned char* stg = smem_raw + (size_t)wave * 16 * RSF;
    const int col4 = lane % LPR, prow = lane / LPR;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + col4 * 4);
    __syncthreads();   // the input tile is dead
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const f32x4_t a = acc[cb][pb];
        *reinterpret_cast<float4*>(stg + lq * RSF + (cb * 16 + lg * 4) * 4) = make_float4(a[0], a[1], a[2], a[3]);
      }
      __syncthreads();
#pragma unroll
      for (int r = prow; r < 16; r += RPW) {
        const int l = l0 + wave * (BL / 4) + pb * 16 + r;
        if (l < L) epilogue_wide4(p, *reinterpret_cast<const float4*>(stg + r * RSF + col4 * 16), bias4, b * L + l, col4 * 4, 0);
      }
      if (pb + 1 < PB) __syncthreads();
    }
    return;
  }
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int l = l0 + wave * (BL / 4) + pb * 16 + lq;
    if (l < L) {
      const int m = b * L + l;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) epilogue_store(p, acc[cb][pb], m, cb * 16 + lg * 4, b, l, 0);
    }
  }
}

template <int C, int BL>
static void launch_halo(const ConvParams& p, int batch, hipStream_t s) {
  const size_t smem = (size_t)(BL + (p.taps - 1) * p.dw) * (C + 8) * 2;
  dim3 grid((unsigned)((p.wo + BL - 1) / BL), (unsigned)batch);
  conv1d_halo_kernel<C, BL><<<grid, dim3(256), smem, s>>>(p);
}
// stride-1 1-D conv with Cin == Cout == 32: weight fragments straight from cache (k = tap*C + c).  Measured on MI355X
// (profiles/): 1.7x over the generic kernel at C=32; at C=64 an LDS weight ring only ties and at C=128 the
// activation tile limits the CU to one workgroup and loses 2x, so those widths stay on conv_gemm_kernel.
static bool halo_eligible(const ctta_conv_desc* d, const ConvParams& p, int groups) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("CTTA_HALO"); env = (e && e[0] == '0') ? 0 : 1; }
  if (!env) return false;
  const int C = p.c0;
  if (C != 32 || p.c1 != 0 || d->n != C || groups != 1) return false;
  if (d->kh != 1 || d->hi != 1 || d->ho != 1 || d->stride_w != 1 || d->upsample || d->in_act) return false;
  if (d->wo != d->wi || p.xs0 != C || p.taps < 2) return false;
  if (d->ldc % 4 != 0 || d->out_limit != 0 || d->out_offset != 0) return false;
  return (size_t)(256 + (p.taps - 1) * p.dw) * (C + 8) * 2 <= 64 * 1024;
}

// ------------------------------------------------------------------------------------------
// split-K second pass: sums the fp32 partial slabs [S][M][ld] and runs the fused epilogue of the original launch
__global__ __launch_bounds__(256) void splitk_finish_kernel(const ConvParams p, const float* __restrict__ slabs, int S,
                                                            long long slab_stride, int ld) {
  const int n4 = (p.n + 3) / 4;
  const long long total = (long long)p.M * n4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / n4), n = (int)(i - (long long)m * n4) * 4;
    const float* src = slabs + (size_t)m * ld + n;
    float4 a = *reinterpret_cast<const float4*>(src);
    for (int s2 = 1; s2 < S; ++s2) {
      const float4 q = *reinterpret_cast<const float4*>(src + (size_t)s2 * slab_stride);
      a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
    }
    const int b = m / p.howo;
    epilogue_store(p, (f32x4_t){a.x, a.y, a.z, a.w}, m, n, b, m - (long long)b * p.howo, 0);
  }
}
// Split-K workspace.  Every engine handle owns one (SplitWs in engine_common.h) and binds it to the calling host
// thread for the duration of each entry point (ctta_conv_bind_workspace), so handles running on different streams
// or host threads never share partial-sum slabs.  Raw ctta_conv_gemm callers that bound nothing get a lazily
// allocated workspace PER DEVICE (mutex-guarded); launches that share it must be ordered on one stream.
static const size_t kSplitWsBytes = (size_t)192 << 20;
static thread_local float* t_ws = nullptr;
static thread_local size_t t_ws_bytes = 0;
extern "C" void ctta_conv_bind_workspace(void* ws, size_t bytes) { t_ws = (float*)ws; t_ws_bytes = ws ? bytes : 0; }
extern "C" size_t ctta_conv_workspace_bytes(void) { return kSplitWsBytes; }
#include <mutex>
static float* splitk_workspace(size_t* bytes) {
  if (t_ws) { *bytes = t_ws_bytes; return t_ws; }
  static std::mutex mu;
  static float* per_dev[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!per_dev[dev] && hipMalloc((void**)&per_dev[dev], kSplitWsBytes) != hipSuccess) per_dev[dev] = nullptr;
  *bytes = kSplitWsBytes;
  return per_dev[dev];
}

// ------------------------------------------------------------------------------------------
struct Variant {
  const char* name;
  int bm, bn, bk;
  int wm, wn;
  int mode;
  void (*launch)(const ConvParams&, dim3, hipStream_t);
  ctta_status (*prepare)();
};

template <int BM, int BN, int BK, int STAGES>
static constexpr size_t smem_bytes() { return (size_t)STAGES * (BM + BN) * BK * 2; }

template <int BM, int BN, int BK, int WM, int WN, int GLDS, int STAGES>
static void launch_variant(const ConvParams& p, dim3 grid, hipStream_t s) {
  const size_t smem = smem_bytes<BM, BN, BK, STAGES>();
  conv_gemm_kernel<BM, BN, BK, WM, WN, GLDS, STAGES><<<grid, dim3(64 * WM * WN), smem, s>>>(p);
}

template <int BM, int BN, int BK, int WM, int WN, int GLDS, int STAGES>
static ctta_status prepare_variant() {
  static bool done = false;
  if (done) return CTTA_OK;
  CTTA_CHECK_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, BK, WM, WN, GLDS, STAGES>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes<BM, BN, BK, STAGES>()));
  done = true;
  return CTTA_OK;
}

#define VARIANT(BM, BN, BK, WM, WN, G, S) \
  {#BM "x" #BN "x" #BK "_w" #WM "x" #WN "_m" #G "_s" #S, BM, BN, BK, WM, WN, G, \
   launch_variant<BM, BN, BK, WM, WN, G, S>, prepare_variant<BM, BN, BK, WM, WN, G, S>}

static const Variant kVariants[] = {
    VARIANT(128, 128, 64, 2, 2, 0, 2),  // 1   register-staged (support in_act)
    VARIANT(128, 128, 32, 2, 2, 0, 2),  // 2
    VARIANT(256, 64, 64, 4, 1, 0, 2),   // 3
    VARIANT(256, 32, 64, 4, 1, 0, 2),   // 4
    VARIANT(64, 64, 64, 2, 2, 0, 2),    // 5
    VARIANT(64, 128, 64, 2, 2, 0, 2),   // 6
    VARIANT(256, 128, 64, 4, 2, 0, 2),  // 7
    VARIANT(128, 64, 64, 2, 2, 0, 2),   // 8
    VARIANT(128, 128, 64, 2, 2, 1, 2),  // 9   direct-to-LDS, generic gather: twins of 1..8
    VARIANT(128, 128, 32, 2, 2, 1, 2),  // 10
    VARIANT(256, 64, 64, 4, 1, 1, 2),   // 11
    VARIANT(256, 32, 64, 4, 1, 1, 2),   // 12
    VARIANT(64, 64, 64, 2, 2, 1, 2),    // 13
    VARIANT(64, 128, 64, 2, 2, 1, 2),   // 14
    VARIANT(256, 128, 64, 4, 2, 1, 2),  // 15
    VARIANT(128, 64, 64, 2, 2, 1, 2),   // 16
    VARIANT(128, 128, 64, 2, 2, 2, 2),  // 17  direct-to-LDS, descriptor fast path: twins of 1..8
    VARIANT(128, 128, 32, 2, 2, 2, 2),  // 18
    VARIANT(256, 64, 64, 4, 1, 2, 2),   // 19
    VARIANT(256, 32, 64, 4, 1, 2, 2),   // 20
    VARIANT(64, 64, 64, 2, 2, 2, 2),    // 21
    VARIANT(64, 128, 64, 2, 2, 2, 2),   // 22
    VARIANT(256, 128, 64, 4, 2, 2, 2),  // 23
    VARIANT(128, 64, 64, 2, 2, 2, 2),   // 24
    VARIANT(128, 128, 32, 2, 2, 1, 4),  // 25  multi-stage rings (counted vmcnt)
    VARIANT(128, 128, 32, 2, 2, 2, 3),  // 26
    VARIANT(64, 128, 64, 2, 2, 2, 3),   // 27
    VARIANT(256, 128, 32, 4, 2, 2, 2),  // 28
    VARIANT(256, 256, 64, 2, 4, 2, 2),  // 29  8 waves, 128x64 per wave
    VARIANT(256, 256, 32, 2, 4, 2, 2),  // 30
    VARIANT(256, 128, 64, 2, 2, 2, 2),  // 31  4 waves, 128x64 per wave
    VARIANT(256, 128, 32, 2, 2, 2, 2),  // 32
    VARIANT(256, 256, 32, 2, 4, 2, 3),  // 33  deeper rings for the big tile (96 / 128 KB)
    VARIANT(256, 256, 32, 2, 4, 2, 4),  // 34
};
static const int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

extern "C" int ctta_conv_gemm_num_variants(void) { return kNumVariants; }
extern "C" const char* ctta_conv_gemm_variant_name(int id) {
  return (id >= 1 && id <= kNumVariants) ? kVariants[id - 1].name : "auto";
}

static const bf16_t* zero_page() {   // 256 zero bytes per device (source of out-of-range chunks in the LDS-direct paths)
  static std::mutex mu;
  static bf16_t* per_dev[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!per_dev[dev]) {
    bf16_t* z = nullptr;
    if (hipMalloc((void**)&z, 256) != hipSuccess) return nullptr;
    if (hipMemset(z, 0, 256) != hipSuccess) { (void)hipFree(z); return nullptr; }
    per_dev[dev] = z;
  }
  return per_dev[dev];
}

static bool xcd_default() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CTTA_XCD"); v = (e && e[0] == '0') ? 0 : 1; }
  return v != 0;
}
static bool wide_store_default() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CTTA_WIDE_STORE"); v = (e && e[0] == '0') ? 0 : 1; }
  return v != 0;
}
static thread_local unsigned long long* t_stamps = nullptr;
extern "C" void ctta_conv_debug_stamps(void* buf) { t_stamps = (unsigned long long*)buf; }
static thread_local int t_no_splitk = 0;
extern "C" void ctta_conv_suppress_splitk(int on) { t_no_splitk = on ? 1 : 0; }
static bool splitk_default() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CTTA_SPLITK"); v = (e && e[0] == '0') ? 0 : 1; }
  return v != 0 && !t_no_splitk;
}
static bool epi_fast_default() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CTTA_EPI_FAST"); v = (e && e[0] == '0') ? 0 : 1; }
  return v != 0;
}
static bool epi_barrier_default() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("CTTA_EPI_BARRIER"); v = (e && e[0] == '1') ? 1 : 0; }
  return v != 0;
}
static bool glds_default() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("CTTA_GLDS");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}

// Tile choice from the on-device sweep (tools/sweep_conv.py, profiles/sweep_r01*.json); ids index
// kVariants (1-based).  Returns a register-staged id (1..8); the caller adds +8 / +16 for the
// direct-to-LDS twins.  kBigTile (256x256x64, 8 waves of 128x64) is chosen separately: it halves the
// L1->LDS bytes per FLOP, which is what bounds the 128-wide tiles (64 B/clk/CU vs 512 MFMA-cycles).
static const int kBigTile = 29;
static bool want_big_tile(long long M, int N, long long K, int groups) {
  const long long t256 = ((M + 255) / 256) * ((N + 255) / 256) * groups;
  return N >= 256 && (N % 256 == 0 || N >= 1024) && K >= 512 && t256 >= 192;   // (K >= 512 since the straight-line epilogue, profiles/sweep_r02*.json)
}
static int pick_variant(long long M, int N, long long K, int groups) {
  if (N <= 32) {                                           // 256x32; few row tiles (the per-sample cross-attention
    const long long t256 = ((M + 255) / 256) * groups;     // K / V^T projections: 18..180 workgroups walking K = 1024
    return t256 >= 512 ? 4 : 5;                            // one latency-bound tile at a time): 64x64 quadruples them
  }
  if (N <= 64) return K >= 512 ? 8 : 5;                    // 128x64 / 64x64
  const long long t128 = ((M + 127) / 128) * ((N + 127) / 128) * groups;
  if (t128 < 200) return 5;                                // too few 128x128 tiles to fill 256 CUs
  if (t128 < 1024 && (K < 4096 || t128 < 400 || N <= 512)) return 6;                    // thin grids (distillation micro-batch): 64x128x64 doubles the workgroups
  if (K >= 4096) return N >= 256 ? 1 : 6;                  // 128x128x64 / 64x128x64
  if (K > 1536) return M >= 400000 ? 6 : 2;                // 64x128x64 / 128x128x32
  return N >= 256 ? 2 : 6;                                 // 128x128x32 / 64x128x64 (re-swept with the wide-store epilogue)
}

static thread_local int t_last_gn_chunks = 0;
extern "C" int ctta_conv_last_gn_chunks(void) { return t_last_gn_chunks; }

extern "C" ctta_status ctta_conv_gemm(const ctta_conv_desc* d, void* stream) {
  t_last_gn_chunks = 0;
  CTTA_REQUIRE(d && d->x0 && d->w && d->out, "conv_gemm: null pointer");
  CTTA_REQUIRE(d->c0 > 0 && d->c0 % 8 == 0 && d->c1 % 8 == 0 && d->c1 >= 0,
               "conv_gemm: channel counts must be multiples of 8 (c0=%d c1=%d)", d->c0, d->c1);
  CTTA_REQUIRE(d->n > 0, "conv_gemm: n=%d must be positive", d->n);
  const bool scalar_store = d->ldc % 4 != 0;
  CTTA_REQUIRE(scalar_store || d->n % 4 == 0 ||
                   (!d->bias && !d->rowvec && !d->res && !d->accumulate && !d->out_limit && (d->n + 3) / 4 * 4 <= d->ldc),
               "conv_gemm: n=%d must be a multiple of 4 for this epilogue", d->n);
  CTTA_REQUIRE(!scalar_store || (!d->rowvec && !d->res && !d->accumulate && !d->out2 && !d->out_limit && d->out_offset == 0),
               "conv_gemm: scalar-store mode (ldc %% 4 != 0) supports only bias/bias_m epilogues");
  CTTA_REQUIRE(!d->out2 || !d->out_f32, "conv_gemm: out2 needs a bf16 primary output");
  CTTA_REQUIRE(d->k_pad % 64 == 0, "conv_gemm: k_pad=%d must be a multiple of 64", d->k_pad);
  CTTA_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride_h >= 1 && d->stride_w >= 1 && d->dil_h >= 1 &&
                   d->dil_w >= 1, "conv_gemm: bad kernel geometry");
  CTTA_REQUIRE(!d->upsample || (d->hi % 2 == 0 && d->wi % 2 == 0), "conv_gemm: odd upsample extent");
  CTTA_REQUIRE(d->out_offset % 4 == 0, "conv_gemm: out_offset must be a multiple of 4");
  CTTA_REQUIRE(!d->res || d->res_ld % 4 == 0, "conv_gemm: res_ld must be a multiple of 4");
  CTTA_REQUIRE(!d->rowvec || d->rowvec_ld % 4 == 0, "conv_gemm: rowvec_ld must be a multiple of 4");
  const bool geglu = d->out_act == 4;
  CTTA_REQUIRE(!geglu || (d->n % 32 == 0 && !d->rowvec && !d->res && !d->accumulate && !d->out2 && !d->out_f32 &&
                          d->groups <= 1 && d->out_limit == 0 && d->out_offset == 0 && d->bias_m == nullptr &&
                          d->alpha == 1.0f && d->ldc % 4 == 0 && d->ldc >= d->n / 2),
               "conv_gemm: the fused GEGLU epilogue takes bias only, n %% 32 == 0 and an output of half the width");
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x0 = (const bf16_t*)d->x0; p.x1 = (const bf16_t*)d->x1;
  p.c0 = d->c0; p.c1 = d->x1 ? d->c1 : 0; p.ct = p.c0 + p.c1;
  p.xs0 = d->x_stride > 0 ? d->x_stride : d->c0;
  CTTA_REQUIRE(p.xs0 >= d->c0 && p.xs0 % 8 == 0, "conv_gemm: x_stride=%d must be >= c0 and a multiple of 8", p.xs0);
  const long long M = (long long)d->batch * d->ho * d->wo;
  CTTA_REQUIRE(M > 0 && M < (1LL << 31), "conv_gemm: M out of range");
  p.M = (int)M; p.hi = d->hi; p.wi = d->wi; p.ups = d->upsample ? 1 : 0;
  p.hs = p.ups ? d->hi / 2 : d->hi; p.ws = p.ups ? d->wi / 2 : d->wi;
  p.ho = d->ho; p.wo = d->wo; p.howo = d->ho * d->wo;
  p.howo_inv = p.howo == 1 ? 0xFFFFFFFFu : (unsigned)((1ULL << 32) / (unsigned)p.howo);
  p.wo_inv = p.wo == 1 ? 0xFFFFFFFFu : (unsigned)((1ULL << 32) / (unsigned)p.wo);
  p.kh = d->kh; p.kw = d->kw; p.taps = d->kh * d->kw;
  p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
  p.dh = d->dil_h; p.dw = d->dil_w;
  p.w = (const bf16_t*)d->w; p.k_pad = d->k_pad; p.n = d->n;
  const long long K = (long long)p.taps * p.ct;
  CTTA_REQUIRE(K <= d->k_pad, "conv_gemm: K=%lld exceeds k_pad=%d", K, d->k_pad);
  p.bias = d->bias; p.bias_m = d->bias_m; p.rowvec = d->rowvec; p.rowvec_ld = d->rowvec_ld;
  p.res = (const bf16_t*)d->res; p.res_ld = d->res_ld;
  p.in_act = d->in_act; p.in_slope = d->in_slope; p.out_act = d->out_act; p.out_slope = d->out_slope;
  p.out2 = (bf16_t*)d->out2; p.out2_slope = d->out2_slope; p.scalar_store = scalar_store ? 1 : 0;
  p.alpha = d->alpha; p.accumulate = d->accumulate;
  p.out = d->out; p.ldc = d->ldc; p.out_f32 = d->out_f32;
  p.obs = d->out_batch_stride ? d->out_batch_stride : (long long)p.howo * d->ldc;
  p.out_offset = d->out_offset; p.out_limit = d->out_limit;
  const int groups = d->groups > 0 ? d->groups : 1;
  p.xgs = d->x_group_stride; p.wgs = d->w_group_stride; p.ogs = d->out_group_stride;

  const long long x_bytes = (((long long)d->batch * p.hs * p.ws - 1) * p.xs0 + p.c0) * 2;
  const long long w_bytes = (long long)d->n * d->k_pad * 2;
  auto fast_ok = [&](int bk) {
    return p.ct % bk == 0 && p.taps <= 32 && p.c1 == 0 && !d->in_act && x_bytes < 0xFFFFFF00LL && w_bytes < 0xFFFFFF00LL &&
           (!p.ups || (d->kh == 3 && d->kw == 3 && d->pad_h == 1 && d->pad_w == 1 && d->stride_h == 1 &&
                       d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1));
  };
  p.plain_out = (d->out_limit == 0 && d->out_offset == 0 && p.obs == (long long)p.howo * d->ldc) ? 1 : 0;
  p.wide_store = (wide_store_default() && !d->out_f32 && !scalar_store && d->ldc % 4 == 0 && d->n % 4 == 0 &&
                  (p.plain_out || (!geglu && p.obs % 4 == 0 && d->out_offset % 4 == 0 && d->out_limit % 4 == 0)) &&
                  (!d->res || d->res_ld % 4 == 0) && (!d->rowvec || d->rowvec_ld % 4 == 0))
                     ? 1 : 0;
  p.epi_barrier = epi_barrier_default() ? 1 : 0;
  p.epi_fast_geglu = (epi_fast_default() && !p.epi_barrier && geglu && p.wide_store && p.plain_out &&
                      M * (long long)d->ldc * 2 < 0x7FFFFF00LL) ? 1 : 0;
  p.epi_fast = (epi_fast_default() && !p.epi_barrier && p.wide_store && M * (long long)d->ldc * 2 < 0x7FFFFF00LL &&
                (!d->res || M * (long long)d->res_ld * 2 < 0x7FFFFF00LL) && p.plain_out && !geglu && !d->bias_m && !d->gn_part && !(d->accumulate && d->out2) &&
                (d->out_act == 0 || (d->out_act == 3 && d->out_slope >= 0.f && d->out_slope <= 1.f)) &&
                (!d->out2 || (d->res && d->out2_slope >= 0.f && d->out2_slope <= 1.f))) ? 1 : 0;
  p.stamps = t_stamps;
  int vid = d->tile;
  if (vid <= 0 && halo_eligible(d, p, groups)) {
    const bool prof = ctta_prof_active();
    if (prof) ctta_prof_begin(0, 39, M, d->n, K, groups, (hipStream_t)stream);
    launch_halo<32, 256>(p, d->batch, (hipStream_t)stream);
    if (prof) ctta_prof_end((hipStream_t)stream);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  }
  if (vid <= 0 || vid > kNumVariants) {
    if (!d->in_act && !geglu && glds_default() && want_big_tile(M, d->n, K, groups) && fast_ok(kVariants[kBigTile - 1].bk)) {
      vid = kBigTile;
      // (round 1 sent short-K launches with a residual / second output / accumulate to the 256x128x32 tile because the
      // big tile's rolled epilogue could not hide behind another workgroup; with the straight-line epilogue the big tile
      // wins there too: profiles/sweep_r02_epi.json, 670 vs 626 TFLOP/s at K = 768)
    } else {
      vid = pick_variant(M, d->n, K, groups);
      // fused GEGLU: 128x128x32 through the wide-store epilogue (its read-back loop is rolled, so the 16-fragment tile
      // keeps its accumulators in registers); the direct epilogue only exists in the <= 8-fragment tiles (64x128x64)
      if (geglu) vid = p.wide_store ? 2 : 6;
      // deep and narrow (few 128x128 tiles, long K): the 128x128 tile with split-K beats small tiles that only
      // exist to create workgroups (measured: M=1152, N=1024, K=9216 at 176 TFLOP/s on 64x64 tiles)
      const long long t128 = ((M + 127) / 128) * ((d->n + 127) / 128);
      if (splitk_default() && groups == 1 && K >= 4096 && d->n >= 256 && t128 < 192 && !scalar_store && !geglu &&
          d->out_limit == 0 && d->out_offset == 0)
        vid = 1;
      if (!d->in_act && glds_default()) vid += fast_ok(kVariants[vid - 1].bk) ? 16 : 8;
      // 64 < N <= 128 with enough rows: the 256x128x32 tile (8 waves of 64x64) stages 25 % fewer bytes per FLOP than
      // 128x128 / 64x128 and, with the wide-store epilogue, wins from K = 384 up (sweep: +12..22 %)
      const long long t28 = ((M + 255) / 256) * groups;
      if (!geglu && !d->in_act && glds_default() && d->n > 64 && d->n <= 128 && K >= 384 && t28 >= 512 && fast_ok(32) &&
          vid != 1 + 16)
        vid = 28;
    }
  }
  CTTA_REQUIRE(!(kVariants[vid - 1].mode != 0 && d->in_act), "conv_gemm: in_act needs a register-staged variant (tile 1..8)");
  if (geglu) {   // direct epilogue: <= 8-fragment tiles (64x64, 64x128, 128x64, 256x32); wide-store: also the 128x128 tiles
    const int base = (vid - 1) % 8 + 1;
    CTTA_REQUIRE(vid <= 24 && (base == 4 || base == 5 || base == 6 || base == 8 || (p.wide_store && (base == 1 || base == 2))),
                 "conv_gemm: the fused GEGLU epilogue needs a 64x64 / 64x128 / 128x64 / 256x32 (or, wide-store, 128x128) tile (got %s)",
                 kVariants[vid - 1].name);
  }
  CTTA_REQUIRE(kVariants[vid - 1].mode != 2 || fast_ok(kVariants[vid - 1].bk),
               "conv_gemm: variant %s needs (c0+c1) %% BK == 0, one source and <= 32 taps", kVariants[vid - 1].name);
  const Variant& v = kVariants[vid - 1];
  p.x_bytes = (unsigned)x_bytes; p.w_bytes = (unsigned)w_bytes;
  p.zero = zero_page();
  CTTA_REQUIRE(p.zero, "conv_gemm: could not allocate the zero page");
  p.nk = (int)((K + v.bk - 1) / v.bk);
  CTTA_REQUIRE((long long)p.nk * v.bk <= d->k_pad, "conv_gemm: k_pad too small for BK");
  CTTA_TRY(v.prepare());
  dim3 grid((unsigned)((M + v.bm - 1) / v.bm), (unsigned)((d->n + v.bn - 1) / v.bn), (unsigned)groups);
  p.ksplit = 1; p.nk_split = p.nk;
  // split-K: deep, narrow problems (the 1024-channel levels at small batch: M <= 2304, K = 9216 / 18432) launch
  // too few workgroups to fill 256 CUs; split the K walk over blockIdx.z and reduce in a second pass
  const long long tiles = (long long)grid.x * grid.y;
  int splits = 1;
  float* ws = nullptr;
  size_t ws_bytes = 0;
  if (splitk_default() && groups == 1 && !scalar_store && !geglu && d->out_limit == 0 && d->out_offset == 0 &&
      tiles < 192 && p.nk >= 32 && (ws = splitk_workspace(&ws_bytes)) != nullptr) {
    static int target = -1, cap = -1;   // tuning knobs: workgroups aimed for / most splits
    if (target < 0) { const char* e = getenv("CTTA_SPLITK_TARGET"); target = e ? atoi(e) : 512; }
    if (cap < 0) { const char* e = getenv("CTTA_SPLITK_MAX"); cap = e ? atoi(e) : 8; }
    splits = (int)(target / tiles);
    if (splits > cap) splits = cap;
    if (splits > p.nk / 8) splits = p.nk / 8;
    const int ld = (d->n + 3) / 4 * 4;
    if ((long long)splits * M * ld * 4 > (long long)ws_bytes) splits = 1;
  }
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(0, vid + ((p.epi_fast || p.epi_fast_geglu) ? 0 : 100), M, d->n, K, groups, (hipStream_t)stream);   // +100: generic epilogue
  if (splits == 1 && groups == 1 && xcd_default() && grid.x >= 64) {
    p.m_tiles = (int)grid.x; p.n_tiles = (int)grid.y;
    p.xcd_per = (p.m_tiles + 7) / 8;
    // few N tiles: visit them back to back per M tile (the input tile is read once per XCD); many N tiles (wide
    // linears: the weight matrix is far larger than L2): keep one weight slice hot and walk the XCD's M range
    p.n_inner = (p.n_tiles <= 4 || w_bytes <= (2LL << 20)) ? 1 : 0;   // a <= 2 MB weight matrix stays L2-resident anyway
    grid = dim3((unsigned)(8 * p.xcd_per * p.n_tiles), 1, 1);
  }
  if (d->gn_part && d->gn_groups > 0 && d->gn_hw > 0 && splits == 1 && groups == 1 && p.wide_store && !geglu) {
    // GroupNorm partials from the epilogue: whole tiles per sample, whole groups per tile, a lane's 4 channels in one group
    const int cpg = d->n % d->gn_groups == 0 ? d->n / d->gn_groups : 0;
    const int tn = v.bn / v.wn;
    if (cpg >= 4 && (cpg & (cpg - 1)) == 0 && v.bn % cpg == 0 && d->gn_hw % v.bm == 0 && M % d->gn_hw == 0) {
      const int sub = cpg > tn ? cpg / tn : 1;
      const int nchunk = d->gn_hw / v.bm * v.wm * sub;
      if ((long long)(M / d->gn_hw) * nchunk * d->gn_groups * 2 <= (long long)d->gn_part_floats) {
        p.gn_part = (float*)d->gn_part; p.gn_cpg = cpg; p.gn_G = d->gn_groups; p.gn_hw = d->gn_hw;
        p.gn_nchunk = nchunk;
        t_last_gn_chunks = nchunk;
      }
    }
  }
  if (splits > 1) {
    const int ld = (d->n + 3) / 4 * 4;
    ConvParams q = p;   // first pass: raw partial sums
    q.nk_split = (p.nk + splits - 1) / splits;
    splits = (p.nk + q.nk_split - 1) / q.nk_split;   // every split owns at least one K-tile
    q.ksplit = splits;
    q.bias = nullptr; q.bias_m = nullptr; q.rowvec = nullptr; q.res = nullptr; q.out_act = 0; q.alpha = 1.0f;
    q.accumulate = 0; q.out2 = nullptr; q.out = ws; q.ldc = ld; q.out_f32 = 1; q.obs = (long long)p.howo * ld;
    q.wide_store = 0;
    q.ogs = (long long)M * ld;
    grid.z = (unsigned)splits;
    v.launch(q, grid, (hipStream_t)stream);
    const long long total = M * (ld / 4);
    int fb = (int)((total + 255) / 256);
    if (fb > 4096) fb = 4096;
    splitk_finish_kernel<<<dim3(fb), dim3(256), 0, (hipStream_t)stream>>>(p, ws, splits, M * ld, ld);
  } else {
    v.launch(p, grid, (hipStream_t)stream);
  }
  if (prof) ctta_prof_end((hipStream_t)stream);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
