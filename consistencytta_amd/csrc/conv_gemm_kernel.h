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
#pragma once
#include "common.h"

#include <stdlib.h>

#include "conv_epilogue.h"
#ifndef CTTA_XBAR
#define CTTA_XBAR 1
#endif
#ifndef CTTA_XBAR_MIN
#define CTTA_XBAR_MIN 32
#endif

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
  float4 c4;               // bias + the sample's row vector for the lane's 4 channels (loop invariant on this path)
  int m0, wm, wn;          // tile origin row and the wave's coordinates in the workgroup (GroupNorm partials)
  // destination of the lane's (chunk 0, pass 0) element: byte offset inside a descriptor of out_bytes bytes at out + gofs.
  // Plain destination: the whole matrix, offset (m_first * ldc + n_lane) * 2.  ConvTranspose phases (one GEMM row = `stride`
  // output positions x Cout, per-sample stride, shifted by the padding and clipped at both ends of the sample): the wave's
  // rows lie in ONE sample b, the descriptor covers that sample's out_limit elements at out + b * obs, the offset is
  // ((m_first - b * howo) * ldc + n_lane + out_offset) * 2 -- negative (as unsigned: huge) for the elements the padding cuts
  // off in front, past the extent for those behind: the bounds check is the clipping.
  int voff0; unsigned out_bytes;
};
#define WAVE_LDS_FENCE_() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// out = act((acc + (bias + rowvec) + res [+ old]) * alpha) for one wave tile of FM x FN fragments, CJ fragments (CHR rows)
// per staging chunk, RPW rows per read-back pass.  (bias + rowvec) is one per-lane constant here (a tile on this path
// lies inside one sample); every epilogue of the library adds in this order -- acc + (bias + rowvec), then the residual --
// so that a sample's numbers do not depend on which path its tile took.  ACT = false (alpha == 1 and no LeakyReLU, the
// U-Net / VAE case) drops the scale and the max: per 4 outputs the arithmetic is 4 additions + 2 conversions (+ 8
// operations with a residual) instead of 22 (30) -- round 2's epilogue spent about half of its 4.1 us per 256x256 tile
// issuing VALU work.  Everything is unrolled and free of divergent control flow, so the waits
// the compiler inserts are exact counts.  Stores and residual loads go through buffer descriptors sized to the M valid
// rows: one VGPR offset per lane, the row advance rides in the scalar offset, rows past M are dropped / read as zero by
// the bounds check.  The residual row of (chunk c + 1, pass i) is requested right after that of (chunk c, pass i) was
// consumed (same registers), i.e. a chunk ahead of its use and before the younger half of chunk c's stores.
// GroupNorm statistics of the OUTPUT tile.  (s1, s2) = the lane's (sum, sum of squares) over its 4 channels of every row
// it stored.  Inside a wave the lanes of one channel group (cw / 4 neighbouring column lanes x all row lanes) fold with a
// fixed butterfly, and the group's leader lane writes the WAVE's partial straight to global memory: one "chunk" per (row
// tile, wave row, column slice of a group wider than the wave), no LDS, no barrier, no atomics (bit-reproducible).
// gn_finalize_kernel / gn_apply_kernel<true> add the chunks.
template <int BM, int WM, int TN>
__device__ __forceinline__ void gn_partial_store(const ConvParams& p, float s1, float s2, int m0, int wm, int wn, int prow,
                                                 int col4, int n_lane, bool n_ok) {
  constexpr int LPR = TN / 4;
  const int cpg = p.gn_cpg;
  const int cw = cpg < TN ? cpg : TN;        // channels of one group inside a wave's TN columns
  const int LG = cw / 4;                     // column lanes per group (power of two)
  const int SUB = cpg / cw;                  // wave columns a group spans (1 unless cpg > TN)
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
// GN: additionally accumulates the GroupNorm statistics of the fp32 values (before the bf16 rounding: what the reference's
// fp32 GroupNorm sees; two packed adds and two packed FMAs per 4 outputs) and writes the wave's partials
// (gn_partial_store); BM / WM only matter then.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int FM, int FN, int CJ, int CHR, int RPW, bool RES, bool OUT2, bool ACC, bool ACT, bool GN = false, int BM = 0, int WM = 0,
          int RB = 16, int CB = 16,     // RB x CB: pixel rows x channels of one accumulator fragment
          bool STR = false>             // strided destination (WideCtx::voff0): the row advance rides in the VECTOR offset
__device__ __forceinline__ void wide_epilogue_fast(const ConvParams& p, f32x4_t (&acc)[FN][FM], const WideCtx& w) {
  constexpr int IT = CHR / RPW;          // read-back passes per chunk
  constexpr int NCH = FM / CJ;           // chunks
  constexpr int QB = (RES || ACC) ? 2 : 4;        // passes read back from LDS at a time
  const float slope = p.out_act == 3 ? p.out_slope : 1.0f;      // max(v, v * 1) == v
  const float alpha = p.alpha, slope2 = p.out2_slope;
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<bf16_t*>(p.out) + w.gofs), 0, w.out_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs2 = rso, rsr = rso;
  if constexpr (OUT2) rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out2 + w.gofs), 0, w.out_bytes, 0x00020000);
  if constexpr (RES)
    rsr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (unsigned)(((long long)(p.M - 1) * p.res_ld + p.n) * 2), 0x00020000);
  const int voff = w.voff0;                                     // byte offset of the lane's (chunk 0, pass 0) element
  const int roff = RES ? (w.m_first * p.res_ld + w.n_lane) * 2 : 0;
  const int ostep = RPW * p.ldc * 2, rstep = RPW * p.res_ld * 2;   // bytes per read-back pass (wave-uniform)
  u32x2_t rr[RES ? IT : 1], oo[ACC ? IT : 1];      // residual / old-output rows, requested a chunk ahead
  f32x2_t gs1 = {0.f, 0.f}, gs2 = {0.f, 0.f};
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
        *reinterpret_cast<float4*>(w.stg + (jj * RB + w.frow) * w.rsf + (i * CB + w.nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
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
          float v[4] = {q[e].x + w.c4.x, q[e].y + w.c4.y, q[e].z + w.c4.z, q[e].w + w.c4.w};
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
          if constexpr (ACT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[c] *= alpha; v[c] = fmaxf(v[c], v[c] * slope); }
          }
          u32x2_t pk;
          pk.x = pack2bf(v[0], v[1]);
          pk.y = pack2bf(v[2], v[3]);
          // (STR: a lane whose first element the padding cuts off has a NEGATIVE voff; the hardware adds the scalar offset
          // without wrap-around, so with the row advance there all its later rows would be dropped too)
          if constexpr (STR) __builtin_amdgcn_raw_buffer_store_b64(pk, rso, voff + pass * ostep, 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b64(pk, rso, voff, pass * ostep, 0);
          if constexpr (GN) {
            const f32x2_t a = {v[0], v[1]}, b2 = {v[2], v[3]};
            gs1 += a; gs1 += b2;
            gs2 = a * a + gs2; gs2 = b2 * b2 + gs2;
          }
          if constexpr (OUT2) {
            float w2[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                           __uint_as_float(pk.y & 0xffff0000u)};
#pragma unroll
            for (int c = 0; c < 4; ++c) w2[c] = fmaxf(w2[c], w2[c] * slope2);
            u32x2_t pk2;
            pk2.x = pack2bf(w2[0], w2[1]);
            pk2.y = pack2bf(w2[2], w2[3]);
            if constexpr (STR) __builtin_amdgcn_raw_buffer_store_b64(pk2, rs2, voff + pass * ostep, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(pk2, rs2, voff, pass * ostep, 0);
          }
        }
      }
    }
    if (ch + 1 < NCH) WAVE_LDS_FENCE_();     // the staging rows are rewritten by the next chunk
  }
  if constexpr (GN) gn_partial_store<BM, WM, FN * CB>(p, gs1[0] + gs1[1], gs2[0] + gs2[1], w.m0, w.wm, w.wn, w.prow, w.col4, w.n_lane, true);
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
  for (int r = 0; r < 4; ++r) o[r] = v[r] * gelu_erf_f(g[r]);
  const int h = (n >> 5) * 16 + (n & 15);
  uint2 pk;
  pk.x = pack2bf(o[0], o[1]);
  pk.y = pack2bf(o[2], o[3]);
  *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + h) = pk;
}

// orders a wave's own LDS writes before its LDS reads (other lanes of the SAME wave) without a workgroup barrier
#define WAVE_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// One (output tile, K range) segment of a stream-K workgroup (ConvParams::sk_hdr).  mode 0: the whole K walk (plain
// epilogue); 1: a part of a split tile's K walk -> the accumulators go, as fp32 rows, to partial-tile slot `slot`.
struct SkSeg { int kt_begin, nk, mode, id, slot; };
// One output tile: K walk + fused epilogue on v_mfma_f32_16x16x32_bf16 (a fragment = RB x CB = 16 pixels x 16 channels, a lane
// holds 4 channels of one pixel).  SK: see SkSeg.
// (Round 6 built the 128x64-per-wave tiles on v_mfma_f32_32x32x16_bf16 as well -- 32 cycles per instruction for twice the FLOPs
// of a 16-17 cycle 16x16x32, half the MFMA issue slots, same fragment bytes and accumulator registers; 16 accumulator registers
// of a 32 x 32 block = four fragments of 32 pixels x 8 channels, LDS key (row >> 1) & 7 for conflict-free 32-row fragment
// reads; parity-green, no lost wave -- and measured -3..+2.5 % on the 256x256x64 tile, -2..-7 % on 512x128x64, +5 % on the
// 4-wave 256x128x64 that no rule picks (profiles/sweep_r06_mf32.txt): the loop is not bound by MFMA issue slots.  Removed
// again; commit d0e0819 holds it.)
template <int BM, int BN, int BK, int WM, int WN, int MODE, int STAGES, bool SK>
__device__ __forceinline__ void conv_tile(const ConvParams& p, unsigned char* smem_raw, const int mt, const int nt, const int zs_,
                                          const SkSeg& sk) {
  constexpr bool GLDS = MODE != 0;
  // (Round 5 built and removed a MODE 4 -- the pixel operand through the LDS-DMA ring, the WEIGHT operand straight from L2 into
  // the MFMA fragment registers: a packed weight row is K-contiguous, a lane's 16 bytes are its fragment; inline-asm
  // global_load_dwordx4 with hand-counted vmcnt, parity-green.  It takes two thirds of a 64x128 tile's bytes off the global ->
  // LDS path that bounds the thin launches -- and runs at 0.3-0.6x the LDS-DMA tiles on every shape
  // (profiles/sweep_r05_mode4.txt): a fragment load touches 16 rows x 64 bytes, a quarter of the rate of full-line DMA.)
  constexpr bool FAST = MODE == 2;
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
  constexpr int RB = 16, CB = 16;     // pixel rows x channels of one accumulator fragment
  constexpr int FM = TM / RB, FN = TN / CB;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/pass mismatch");
  static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile");

  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem_raw);               // [STAGES][BM][LDK]
  bf16_t* Ws = Xs + STAGES * BM * LDK;                             // [STAGES][BN][LDK]
  unsigned long long* stamp = nullptr;
  unsigned long long* skst = nullptr;     // stream-K stamps (tools/sk_timeline.py): 8 words per workgroup id
  if constexpr (SK) { if (p.stamps) skst = p.stamps + (size_t)sk.id * 8; }
  if (!SK && p.stamps) {
    stamp = p.stamps + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 6;
    if (threadIdx.x == 0) { stamp[0] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32); stamp[1] = __builtin_amdgcn_s_memtime(); }
  }

  int tid_ = threadIdx.x;
  // (stream-K calls this body in a loop: everything derived from the thread index alone is loop-invariant, and hoisted in front of
  // the loop -- the epilogue's lane constants, the staging offsets -- it stays live across the main loop: 242 spilled registers.
  // An opaque copy per call keeps every value where the one-tile kernel computes it.)
  if constexpr (SK) asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = p.m_off + mt * BM;
  const int n0 = nt * BN;
  // a wave whose TM rows all lie past M (the last row tile of M = k * BM + a few rows) skips its MFMAs: its SIMD partner
  // then runs at full matrix-pipe rate and the tail tile of a one-workgroup-per-CU launch takes about half a tile time
  const bool wave_live = m0 + (wave / WN) * TM < p.M;
  const int zs = zs_;
  const int g = p.ksplit > 1 ? 0 : zs;                    // group index (pointer offsets)
  const int kt_begin = SK ? sk.kt_begin : (p.ksplit > 1 ? zs * p.nk_split : 0);
  const int nk = SK ? sk.nk : (p.ksplit > 1 ? min(p.nk - kt_begin, p.nk_split) : p.nk);

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
  if constexpr (FAST) {
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

  const int frow = lane & (RB - 1);
  const int fchunk = lane / RB;                       // logical 16-byte chunk within a 32-wide k-slab
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
      if constexpr (FAST) { issue_fast(kt_begin + issued, issued); if (issued == 0) full_masks(); }
      else issue_tile(kt_begin + issued, issued);
    }
    int slot = 0, fill = issued % STAGES;
    for (int kt = 0; kt < nk; ++kt) {
      if (issued - kt - 1 >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // tile kt landed for every wave; slot `fill` is free
      if (issued < nk) {
        if constexpr (FAST) issue_fast(kt_begin + issued, fill); else issue_tile(kt_begin + issued, fill);
        ++issued;
        fill = (fill + 1 == STAGES) ? 0 : fill + 1;
      }
      if (wave_live) compute_tile(slot);
      slot = (slot + 1 == STAGES) ? 0 : slot + 1;
    }
  } else {
    if (stamp && threadIdx.x == 0) stamp[5] = __builtin_amdgcn_s_memtime();
    if constexpr (FAST) {
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
    // XBAR (round 5): the tiles whose waves own 32 fragments (128x64 per wave: the big tile, 512x128x64, 256x128x64 with
    // four waves) keep the SECOND half of a K tile's fragments in registers across the barrier and issue its 32 MFMAs right
    // behind the next barrier, while the first fragments of the new tile are still on their way from LDS -- otherwise all
    // eight waves start a K step with reads and the matrix pipe idles for their latency.  Same MFMA order per accumulator
    // (bit-identical results), +14 registers, same occupancy.  Same-box A/B, three alternating rounds of the generation
    // leg: 392.8 -> 398.5 clips/s (VAE decoder -0.45 ms, HiFi-GAN -0.6 ms, U-Net -0.15 ms).  For the tiles with fewer
    // fragments per wave (three to five workgroups per CU fill each other's gaps) it measures nothing: 398.7 vs 399.4
    // clips/s, distillation 77.2 vs 77.4 ms (-DCTTA_XBAR_MIN=8); -DCTTA_XBAR=0 compiles the round-4 loop.
    // (Measured on top of it and not kept, profiles/ab_r05_xbar2_midstep_issue.txt: a second barrier in the middle of the
    // step -- the tile's LDS reads are over by then -- behind which tile t + 2 is issued into the freed buffer, so that every
    // LDS-DMA has 1.5 steps to land with the same two buffers: parity-green, 256 registers, 2.4 % SLOWER.  Like the deeper
    // rings at BK = 32, it says the step is not waiting for the LDS-DMA's latency.  And a hand-pinned two-phase schedule --
    // sched_group_barrier: per position fragment four MFMAs, then the read that takes over its registers, so that no read
    // is waited for behind the barrier at all (in the loop below the compiler sinks the first-half reads under the held
    // MFMAs) -- runs at exactly the same speed with 14 more registers: profiles/ab_r05_xbar3_streaming_schedule.txt.)
    constexpr bool XBAR = FAST && BK == 64 && FM * FN >= CTTA_XBAR_MIN && CTTA_XBAR;
    if constexpr (XBAR) {
      bf16x8_t ha[FN], hb[FM];
      const bf16_t* xs0 = Xs + (wm * TM + frow) * LDK;
      const bf16_t* ws0 = Ws + (wn * TN + frow) * LDK;
      const int koff0 = ((fchunk ^ fswz) & SWZ_MASK) * 8, koff1 = (((4 + fchunk) ^ fswz) & SWZ_MASK) * 8;
      for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue_fast(kt_begin + kt + 1, buf ^ 1);
        if (wave_live) {
          const bf16_t* xs = xs0 + buf * BM * LDK;
          const bf16_t* ws = ws0 + buf * BN * LDK;
          bf16x8_t fa[FN], fb[FM];
#pragma unroll
          for (int i = 0; i < FN; ++i) fa[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ws + i * 16 * LDK + koff0));
#pragma unroll
          for (int j = 0; j < FM; ++j) fb[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xs + j * 16 * LDK + koff0));
          if (kt > 0) {
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
              for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < FN; ++i) ha[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ws + i * 16 * LDK + koff1));
#pragma unroll
          for (int j = 0; j < FM; ++j) hb[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xs + j * 16 * LDK + koff1));
#pragma unroll
          for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
      }
      if (wave_live && nk > 0) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);
      }
    } else
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) {
        if constexpr (FAST) issue_fast(kt_begin + kt + 1, buf ^ 1);
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
  // ---- stream-K (ConvParams::sk_hdr).  A split tile's partial sums travel as fp32 ROWS ([BM][BN] per slot, written through
  // the LDS transpose of the wide-store epilogues), the part that holds the tile's first K step included; sk_fold_tile (called
  // by the kernel below) sums them in K order and runs the fused epilogue.
  // (Not built the cheaper-looking way -- partners' partials added to the owner's accumulators, or to its staged rows inside the
  // straight-line epilogues: any VALU write to the 32 MFMA tuples between main loop and epilogue, even `+= 1.0f`, and any
  // runtime loop inside the unrolled epilogue variants costs this compiler 50-240 spilled registers INSIDE the main loop.)
  const int nsub = (lane / RB) * 4;
  if constexpr (SK) {
    if (skst && threadIdx.x == 0) skst[sk.mode == 1 && sk.kt_begin > 0 ? 2 : 4] = __builtin_amdgcn_s_memtime();
    if (sk.mode != 0) {
      // the wave's TM x TN block through its staging rows, out as whole TN * 4-byte row segments, write-through (sc1) so that
      // no release fence has anything to write back; every wave drains its stores before the caller counts the part as arrived
      // (cdna_hip_programming.md Guideline 16, form R1)
      constexpr int NW = WM * WN;
      constexpr size_t RING = (size_t)STAGES * (BM + BN) * BK * 2;
      constexpr int RSF = TN * 4 + 16;
      constexpr int CHR = (FM % 2 == 0 && (size_t)NW * 32 * RSF <= RING) ? 32 : 16;
      static_assert((size_t)NW * CHR * RSF <= RING && CHR % RB == 0, "stream-K staging does not fit the ring");
      constexpr int CJ = CHR / RB, LPR = TN / 4, RPW = 64 / LPR;
      unsigned char* stg = smem_raw + (size_t)wave * CHR * RSF;
      const int col4 = lane % LPR, prow = lane / LPR;
      const int m_w = m0 + (wave_u / WN) * TM;
      const int rows_in = min(TM, p.M - m_w);            // rows past M are never read back
      float* obase = p.sk_slots + (size_t)sk.slot * (BM * BN) + (size_t)((wave_u / WN) * TM) * BN;
      const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
          (void*)obase, 0, rows_in > 0 ? (unsigned)rows_in * (unsigned)BN * 4u : 0u, 0x00020000);
      const unsigned ovoff = (unsigned)(prow * BN + wn * TN + col4 * 4) * 4u;
      constexpr unsigned ostep = (unsigned)(RPW * BN) * 4u;
      __syncthreads();   // every wave is done with the ring
#pragma unroll
      for (int j0 = 0; j0 < FM; j0 += CJ) {
#pragma unroll
        for (int jj = 0; jj < CJ; ++jj)
#pragma unroll
          for (int i = 0; i < FN; ++i) {
            const f32x4_t a = acc[i][j0 + jj];
            *reinterpret_cast<float4*>(stg + (jj * RB + frow) * RSF + (i * CB + nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
          }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int it = 0; it < CHR / RPW; ++it) {
          const float4 q = *reinterpret_cast<const float4*>(stg + (prow + it * RPW) * RSF + col4 * 16);
          typedef float st4_t __attribute__((ext_vector_type(4)));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (st4_t){q.x, q.y, q.z, q.w}), rso, ovoff,
                                                 (j0 * RB / RPW + it) * ostep, 16);
        }
        if (j0 + CJ < FM) WAVE_LDS_FENCE();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (skst && threadIdx.x == 0) skst[sk.kt_begin > 0 ? 3 : 5] = __builtin_amdgcn_s_memtime();
      return;
    }
  }
  // ---- epilogue: lane holds n = nb + (lane / RB) * 4 + {0..3} (rows of D), m = mb + (lane & (RB - 1))
  // Wide-store epilogue (bf16 output, plain row-major destination).  In the MFMA layout a store instruction writes
  // 16 rows x 32 bytes; measured, a launch then pays ~0.5 us per MB of output on top of its main loop (335 MB outputs:
  // 180 of 300 us of K-independent time) because workgroups cannot retire before their scattered stores drain.  Here
  // every wave transposes its tile through the (dead) LDS ring in fp32, CHR rows at a time, and reads it back with
  // TN/4 consecutive lanes per output row: stores -- and the residual reads -- are whole TN*2-byte row segments, the
  // bias sits in registers because a lane keeps its 4 channels, and the arithmetic runs in one rolled loop.
  // (Not in the LDS-DMA tiles whose waves own 64 x 64 at BK = 32 with two stages -- 128x128x32 and the 8-wave 256x128x32: they
  // sit at exactly 128 VGPRs, and with this block compiled in they need 129 = one workgroup less per SIMD quarter; the 8-wave
  // tile then runs ONE workgroup per CU and its fused-GEGLU launches drop from 796 to 557 TFLOP/s.  conv_wide_f32_ok mirrors
  // the condition for the host.)
  if constexpr (conv_wide_f32_ok(BM, BN, BK, WM, WN, MODE, STAGES))
  if (p.wide_f32) {
    // fp32 output (split-K partial slabs, the VAE attention's score matrix, weight-gradient slabs of the conv_gemm route):
    // the same LDS transpose as the bf16 wide store, rows written as float4 per lane = TN * 4 contiguous bytes per row and
    // wave.  In the MFMA layout a store instruction wrote 16 rows x 64 bytes: half cache lines, and the 2 GiB score slab of
    // the VAE mid-block attention left at 1.9 TB/s (round 4).
    constexpr int NW = WM * WN;
    constexpr size_t RING = (size_t)STAGES * (BM + BN) * BK * 2;
    constexpr int RSF = TN * 4 + 16;
    constexpr int CHR = (FM % 2 == 0 && (size_t)NW * 32 * RSF <= RING) ? 32 : 16;
    static_assert((size_t)NW * CHR * RSF <= RING && CHR % RB == 0, "wide-store staging does not fit the ring");
    constexpr int CJ = CHR / RB;
    constexpr int LPR = TN / 4, RPW = 64 / LPR;
    unsigned char* stg = smem_raw + (size_t)wave * CHR * RSF;
    const int col4 = lane % LPR, prow = lane / LPR;
    const int n_lane = n0 + wn * TN + col4 * 4;
    const bool n_ok = n_lane < p.n;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) bias4 = *reinterpret_cast<const float4*>(p.bias + n_lane);
    // Stores through a buffer descriptor over THIS wave's rows (base = its first row, extent = its rows inside M): rows past
    // M and lanes past n fall outside and are dropped by the bounds check, so there is no divergent `if (m < M)` around a
    // store -- behind one the compiler waits vmcnt(0) in front of every read-back, i.e. each of the 16-32 row sweeps waited
    // for the previous store to be acknowledged (the bf16 epilogue got the same treatment in round 2).
    const int m_w = m0 + (wave_u / WN) * TM;                           // wave-uniform, and known to be (scalar descriptor: no waterfall loop)
    const int rows_in = min(TM, p.M - m_w);                            // <= 0: nothing of this wave's rows is inside M
    float* obase = reinterpret_cast<float*>(p.out) + (size_t)zs * p.ogs + (size_t)(rows_in > 0 ? m_w : 0) * p.ldc;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
        (void*)obase, 0, rows_in > 0 ? (unsigned)rows_in * (unsigned)p.ldc * 4u : 0u, 0x00020000);
    const unsigned ovoff = n_ok ? (unsigned)(prow * p.ldc + n_lane) * 4u : 0xFFFFFFF0u;
    const unsigned ostep = (unsigned)(RPW * p.ldc) * 4u;              // bytes per row sweep
    __syncthreads();   // every wave is done with the ring
#pragma unroll
    for (int j0 = 0; j0 < FM; j0 += CJ) {
#pragma unroll
      for (int jj = 0; jj < CJ; ++jj)
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const f32x4_t a = acc[i][j0 + jj];
          *reinterpret_cast<float4*>(stg + (jj * RB + frow) * RSF + (i * CB + nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
      WAVE_LDS_FENCE();
#pragma unroll
      for (int it = 0; it < CHR / RPW; ++it) {
        const int r = prow + it * RPW;
        float4 q = *reinterpret_cast<const float4*>(stg + r * RSF + col4 * 16);
        q.x += bias4.x; q.y += bias4.y; q.z += bias4.z; q.w += bias4.w;
        typedef float st4_t __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (st4_t){q.x, q.y, q.z, q.w}), rso, ovoff,
                                               (j0 * RB / RPW + it) * ostep, 0);
      }
      if (j0 + CJ < FM) WAVE_LDS_FENCE();
    }
    return;
  }
  if (p.wide_store) {
    constexpr int NW = WM * WN;
    constexpr size_t RING = (size_t)STAGES * (BM + BN) * BK * 2;
    constexpr int RSF = TN * 4 + 16;                       // staging row stride (bytes): +16 keeps 16-byte accesses conflict-free
    constexpr int CHR = (FM % 2 == 0 && (size_t)NW * 32 * RSF <= RING) ? 32 : 16;   // rows per chunk
    static_assert((size_t)NW * CHR * RSF <= RING && CHR % RB == 0, "wide-store staging does not fit the ring");
    constexpr int CJ = CHR / RB;
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
              for (int e = 0; e < 4; ++e) o[e] = vv[e] * gelu_erf_f(gg[e]);
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
          WAVE_LDS_FENCE();   // wave-private staging rows
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
              for (int e = 0; e < 4; ++e) o[e] = vv[e] * gelu_erf_f(gg[e]);
              uint2 pk;
              pk.x = pack2bf(o[0], o[1]);
              pk.y = pack2bf(o[2], o[3]);
              *reinterpret_cast<uint2*>(og + (size_t)m * p.ldc) = pk;
            }
          }
          if (j0 + CJ < FM) { WAVE_LDS_FENCE(); }
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
    bool fast_wave = false;
    const int m_wv = m0 + (wave_u / WN) * TM;                 // this wave's first row
    if (p.epi_fast && n0 + wn * TN + TN <= p.n) {
      const bool one_sample = m0 / p.howo == (min(m0 + BM, p.M) - 1) / p.howo;
      fast_wave = !p.rowvec || one_sample;
      // a destination with a per-sample stride: the wave's rows (below M) in one sample
      if (!p.plain_out) fast_wave = fast_wave && (m_wv >= p.M || m_wv / p.howo == (min(m_wv + TM, p.M) - 1) / p.howo);
    }
    if (fast_wave) {
      float4 c4 = bias4;
      if (p.rowvec) {
        const float4 rv4 = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m0 / p.howo) * p.rowvec_ld + n_lane);
        c4.x += rv4.x; c4.y += rv4.y; c4.z += rv4.z; c4.w += rv4.w;
      }
      WideCtx wc = {stg, RSF, frow, nsub, prow, col4, m0 + wm * TM + prow, (size_t)g * p.ogs, n_lane, c4, m0, wm, wn,
                    ((m0 + wm * TM + prow) * p.ldc + n_lane) * 2, (unsigned)(((long long)(p.M - 1) * p.ldc + p.n) * 2)};
      if (!p.plain_out) {
        const int b = min(m_wv, p.M - 1) / p.howo;
        wc.gofs += (size_t)((long long)b * p.obs);
        wc.voff0 = (int)(((long long)(m0 + wm * TM + prow - b * p.howo) * p.ldc + n_lane + p.out_offset) * 2);
        wc.out_bytes = (unsigned)(p.out_limit * 2);
      }
      if (!p.plain_out) {    // host: no residual, no accumulate, no statistics, no row vector (the ConvTranspose upsamplers)
        if (p.epi_act) {
          if (p.out2) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, true, false, true, false, 0, 0, RB, CB, true>(p, acc, wc);
          else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, false, true, false, 0, 0, RB, CB, true>(p, acc, wc);
        } else {
          if (p.out2) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, true, false, false, false, 0, 0, RB, CB, true>(p, acc, wc);
          else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, false, false, false, 0, 0, RB, CB, true>(p, acc, wc);
        }
      } else if (p.gn_part) {       // host: no accumulate, no second output, alpha == 1, no activation on this path
        if (!p.res) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, false, false, true, BM, WM, RB, CB>(p, acc, wc);
        else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, false, false, false, true, BM, WM, RB, CB>(p, acc, wc);
      } else if (p.epi_act) {
        if (p.accumulate) {
          if (!p.res) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, true, true, false, 0, 0, RB, CB>(p, acc, wc);
          else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, false, true, true, false, 0, 0, RB, CB>(p, acc, wc);
        } else if (!p.res && p.out2) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, true, false, true, false, 0, 0, RB, CB>(p, acc, wc);
        else if (!p.res) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, false, true, false, 0, 0, RB, CB>(p, acc, wc);
        else if (!p.out2) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, false, false, true, false, 0, 0, RB, CB>(p, acc, wc);
        else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, true, false, true, false, 0, 0, RB, CB>(p, acc, wc);
      } else {
        if (p.accumulate) {
          if (!p.res) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, true, false, false, 0, 0, RB, CB>(p, acc, wc);
          else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, false, true, false, false, 0, 0, RB, CB>(p, acc, wc);
        } else if (!p.res && p.out2) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, true, false, false, false, 0, 0, RB, CB>(p, acc, wc);
        else if (!p.res) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, false, false, false, false, false, 0, 0, RB, CB>(p, acc, wc);
        else if (!p.out2) wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, false, false, false, false, 0, 0, RB, CB>(p, acc, wc);
        else wide_epilogue_fast<FM, FN, CJ, CHR, RPW, true, true, false, false, false, 0, 0, RB, CB>(p, acc, wc);
      }
      if (stamp && threadIdx.x == 0) stamp[4] = __builtin_amdgcn_s_memtime();
      return;
    }
    float2 gacc = make_float2(0.f, 0.f);   // always passed (a conditional pointer would force it into scratch)
#pragma unroll
    for (int j0 = 0; j0 < FM; j0 += CJ) {
#pragma unroll
      for (int jj = 0; jj < CJ; ++jj)
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const f32x4_t a = acc[i][j0 + jj];
          *reinterpret_cast<float4*>(stg + (jj * RB + frow) * RSF + (i * CB + nsub) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
      // the staging rows are private to the wave and a wave's LDS operations execute in order: no workgroup barrier
      WAVE_LDS_FENCE();
#pragma unroll 2
      for (int r = prow; r < CHR; r += RPW) {
        const int m = m0 + wm * TM + j0 * RB + r;
        if (m < p.M && n_ok)
          epilogue_wide4(p, *reinterpret_cast<const float4*>(stg + r * RSF + col4 * 16), bias4, m, n_lane, (size_t)g * p.ogs,
                         &gacc);
      }
      if (j0 + CJ < FM) { WAVE_LDS_FENCE(); }
    }
    if (stamp && threadIdx.x == 0) stamp[4] = __builtin_amdgcn_s_memtime();
    if (p.gn_part) gn_partial_store<BM, WM, TN>(p, gacc.x, gacc.y, m0, wm, wn, prow, col4, n_lane, n_ok);
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
      const int m = m0 + wm * TM + j * RB + frow;
      const bool m_ok = m < p.M;
      const int b = m_ok ? m / p.howo : 0;
      const long long mrem = m - (long long)b * p.howo;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int n = n0 + wn * TN + i * CB + nsub;
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
          const int m = m0 + wm * TM + (idx % FM) * RB + frow;
          const int n = n0 + wn * TN + (idx / FM) * CB + nsub;
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
        const int m = m0 + wm * TM + j * RB + frow;
        const int n = n0 + wn * TN + i * CB + nsub;
        if (m < p.M && n < p.n) {
          const int b = m / p.howo;
          epilogue_store(p, (f32x4_t){q.x, q.y, q.z, q.w}, m, n, b, m - (long long)b * p.howo, zs, &rr);
        }
      }
    }
  }
}


// blockIdx -> (row tile, column tile, group / K split) of the one-tile-per-workgroup launches; false: a padding block
__device__ __forceinline__ bool conv_block_tile(const ConvParams& p, int& mt, int& nt, int& zs_) {
  mt = blockIdx.x; nt = blockIdx.y; zs_ = blockIdx.z;
  if (p.slab_total > 0) {
    const int item = (int)(blockIdx.x & 7) * p.slab_per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.slab_per || item >= p.slab_total) return false;   // padding blocks (whole workgroup, before any barrier)
    const int slab = item / p.m_tiles;
    mt = item - slab * p.m_tiles;
    zs_ = slab / p.n_tiles;
    nt = slab - zs_ * p.n_tiles;
  } else if (p.xcd_per > 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    if (p.n_inner) { nt = local % p.n_tiles; mt = xcd * p.xcd_per + local / p.n_tiles; }
    else { nt = local / p.xcd_per; mt = xcd * p.xcd_per + local % p.xcd_per; }
    if (mt >= p.m_tiles) return false;   // padding blocks of the last XCD range (whole workgroup, before any barrier)
  }
  return true;
}
// Waves per SIMD the register allocation must leave room for.  The LDS-DMA tiles whose waves own 64 x 64 at BK = 32 with two
// stages (128x128x32, the 8-wave 256x128x32) run FOUR per SIMD and sat at exactly 128 VGPRs by luck of the allocator: round 4
// lost 1.3 % of a generation step when one epilogue line made it 129, and the move of the body into conv_tile did it again.
// Declared, the bound is the compiler's problem.
constexpr int conv_min_waves(int bm, int bn, int bk, int wm, int wn, int mode, int stages) {
  return (mode != 0 && bk == 32 && stages == 2 && bm / wm == 64 && bn / wn == 64) ? 4 : 2;
}
template <int BM, int BN, int BK, int WM, int WN, int MODE, int STAGES>
__global__ __launch_bounds__(64 * WM * WN, conv_min_waves(BM, BN, BK, WM, WN, MODE, STAGES)) void conv_gemm_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int mt, nt, zs_;
  if (!conv_block_tile(p, mt, nt, zs_)) return;
  conv_tile<BM, BN, BK, WM, WN, MODE, STAGES, false>(p, smem_raw, mt, nt, zs_, SkSeg{});
}
// Stream-K fold: rows [blk * FB, (blk + 1) * FB) of one split tile = the sum of its `parts` partial tiles (slot of part 0, then
// slot1, slot1 + 1, ...: K order) through the fused epilogue.  Streams: per pass a lane holds U float4 of three slots in flight
// (part c + 1 is requested before part c is added; parts that do not exist are read through an empty descriptor: zeros, no
// traffic, no branch).  Whoever runs it gets the same bits: the order is the decomposition's.
template <int BM, int BN, int NT, int FB>
__device__ __forceinline__ void sk_fold_block(const ConvParams& p, int mt, int nt, int blk, int slot0, int slot1, int parts) {
  constexpr int C4 = BN / 4;
  constexpr int PER = FB * C4 / NT;                       // float4s per lane and block
  static_assert(PER >= 1 && (FB * C4) % NT == 0 && NT % C4 == 0, "fold block");
  const int tid = threadIdx.x;
  const int m0 = p.m_off + mt * BM, n0 = nt * BN;
  const int rows = min(BM, p.M - m0);
  const int r_lo = blk * FB;
  if (r_lo >= rows) return;
  const unsigned tile_bytes = (unsigned)rows * BN * 4u;   // loads past the valid rows return zeros (and are dropped below)
  auto rsrc = [&](int c) {
    const bool ok = c < parts;
    const int slot = c == 0 ? slot0 : slot1 + c - 1;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(p.sk_slots + (size_t)(ok ? slot : slot0) * (BM * BN)), 0, ok ? tile_bytes : 0u, 0x00020000);
  };
  const int vo = (r_lo * C4 + tid) * 16;
  u32x4_t a[PER], t0[PER], t1[PER];
  __amdgpu_buffer_rsrc_t r = rsrc(0);
#pragma unroll
  for (int u = 0; u < PER; ++u) a[u] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, u * NT * 16, 0);
  r = rsrc(1);
#pragma unroll
  for (int u = 0; u < PER; ++u) t0[u] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, u * NT * 16, 0);
  for (int c = 1; c < parts; c += 2) {
    r = rsrc(c + 1);
#pragma unroll
    for (int u = 0; u < PER; ++u) t1[u] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, u * NT * 16, 0);
#pragma unroll
    for (int u = 0; u < PER; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[u][e] = __float_as_uint(__uint_as_float(a[u][e]) + __uint_as_float(t0[u][e]));
    r = rsrc(c + 2);
#pragma unroll
    for (int u = 0; u < PER; ++u) t0[u] = __builtin_amdgcn_raw_buffer_load_b128(r, vo, u * NT * 16, 0);
#pragma unroll
    for (int u = 0; u < PER; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[u][e] = __float_as_uint(__uint_as_float(a[u][e]) + __uint_as_float(t1[u][e]));
  }
  // the lane's column never changes (NT % C4 == 0): its bias vector is loaded once
  const int n = n0 + (tid % C4) * 4;
  const bool n_ok = n < p.n;
  const bool wide = p.wide_store != 0;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wide && p.bias && n_ok) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int row = r_lo + (u * NT + tid) / C4, m = m0 + row;
    if (row < rows && n_ok) {
      const float4 q = make_float4(__uint_as_float(a[u][0]), __uint_as_float(a[u][1]), __uint_as_float(a[u][2]), __uint_as_float(a[u][3]));
      if (wide) epilogue_wide4(p, q, bias4, m, n, 0);
      else { const int b = m / p.howo; epilogue_store(p, (f32x4_t){q.x, q.y, q.z, q.w}, m, n, b, m - (long long)b * p.howo, 0); }
    }
  }
}
// Stream-K: one persistent launch (ConvParams::sk_hdr explains the protocol; conv_gemm.hip decides which launches take it).
template <int BM, int BN, int BK, int WM, int WN, int MODE, int STAGES>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_gemm_sk_kernel(ConvParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int FB = NT * 4 / (BN / 4) < BM ? NT * 4 / (BN / 4) : BM;      // rows per fold block: 4 float4 per lane
  constexpr int NB = BM / FB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned* bc = reinterpret_cast<unsigned*>(smem_raw);
  if (threadIdx.x == 0) bc[0] = __hip_atomic_fetch_add(p.sk_hdr + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const int G = (int)gridDim.x;
  const int ticket = (int)__builtin_amdgcn_readfirstlane(bc[0]);
  // XCD chunks.  Workgroups are dispatched round-robin over the 8 XCDs (block b on XCD b % 8: observed, not promised -- only
  // speed depends on it) and tickets follow dispatch, so the tickets t with (t % 8) / r equal -- r = 8 / sk_chunks XCDs -- form
  // one CHUNK that owns a contiguous range of whole tiles and walks it stream-K fashion with its G / sk_chunks workgroups: the
  // tiles of one weight slab / of neighbouring rows meet in the same L2(s), as in the one-tile-per-workgroup launches.  No
  // tile spans two chunks.
  const int nch = p.sk_chunks, r = 8 / nch, per = G / nch;
  const int chunk = (ticket & 7) / r;
  const int lid = (ticket >> 3) * r + (ticket & 7) % r;
  const int id = chunk * per + lid;                                  // slot index
  const int tile_lo = (int)((long long)chunk * p.sk_tiles / nch), tile_hi = (int)((long long)(chunk + 1) * p.sk_tiles / nch);
  const long long items = (long long)(tile_hi - tile_lo) * p.nk;     // of this chunk
  long long a = (long long)lid * items / per;
  const long long b = (long long)(lid + 1) * items / per;
  unsigned steps_a = 0, steps_b = 0, tiles_b = 0;
  // Fold duty.  A split tile is folded by the workgroups that wrote its parts: each part's arrival is counted in
  // sk_hdr[SK_FLAGS + tile]; whoever brings the count to `parts` folds at once, the others come back when their own K walk is
  // over and wait a BOUNDED time for it; every one of them then claims fold blocks (FB rows) from sk_hdr[SK_FLAGS + sk_tiles +
  // tile] until none is left.  Nobody waits without a bound and the last arriver never waits at all: the launch completes
  // whatever is resident, whatever the dispatch order (a helper whose patience runs out simply leaves the blocks to the others).
  int duty0 = -1, duty1 = -1;        // (two scalars: an array indexed at run time would live in scratch)
  auto tile_parts = [&](int tl, int& first_lid, int& parts) {     // the workgroups (local ids) whose ranges cover chunk tile tl
    first_lid = (int)((((long long)tl * p.nk + 1) * per - 1) / items);
    const int last = (int)((((long long)(tl + 1) * p.nk) * per - 1) / items);
    parts = last - first_lid + 1;
  };
  auto tile_mn = [&](int tile, int& mt, int& nt) {
    if (p.sk_m_inner) { nt = tile / p.m_tiles; mt = tile - nt * p.m_tiles; }
    else { mt = tile / p.n_tiles; nt = tile - mt * p.n_tiles; }
  };
  auto fold = [&](int tl) {       // claim and fold blocks of chunk tile tl (its parts have all arrived)
    int first_lid, parts, mt, nt;
    tile_parts(tl, first_lid, parts);
    tile_mn(tile_lo + tl, mt, nt);
    const int slot0 = G + chunk * per + first_lid, slot1 = chunk * per + first_lid + 1;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) bc[2] = __hip_atomic_fetch_add(p.sk_hdr + SK_FLAGS + p.sk_tiles + tile_lo + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int blk = (int)__builtin_amdgcn_readfirstlane(bc[2]);
      if (blk >= NB) break;
      sk_fold_block<BM, BN, NT, FB>(p, mt, nt, blk, slot0, slot1, parts);
    }
  };
  int n_seg = 0;
  while (a < b) {
    const int tl = (int)(a / p.nk);                                  // tile index inside the chunk
    const int tile = tile_lo + tl;
    const int k0 = (int)(a - (long long)tl * p.nk);
    const long long tile_end = (long long)(tl + 1) * p.nk;
    const int k1 = (int)((b < tile_end ? b : tile_end) - (long long)tl * p.nk);
    SkSeg sk;
    sk.kt_begin = k0; sk.nk = k1 - k0; sk.id = id;
    sk.mode = (k0 > 0 || k1 < p.nk) ? 1 : 0;
    sk.slot = k0 > 0 ? id : G + id;                                  // the part with the tile's first K step is part 0
    int mt, nt;
    tile_mn(tile, mt, nt);
    if (k0 > 0) steps_a += k1 - k0; else { steps_b += k1 - k0; ++tiles_b; }
    __syncthreads();      // the previous segment's epilogue / fold (and the broadcast words) are done with the ring
    conv_tile<BM, BN, BK, WM, WN, MODE, STAGES, true>(p, smem_raw, mt, nt, 0, sk);
    if (sk.mode == 1) {   // the part is in memory (every wave drained its write-through stores): count it
      int first_lid, parts;
      tile_parts(tl, first_lid, parts);
      __syncthreads();
      if (threadIdx.x == 0) bc[1] = __hip_atomic_fetch_add(p.sk_hdr + SK_FLAGS + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if ((int)__builtin_amdgcn_readfirstlane(bc[1]) == parts - 1) fold(tl);      // the last to arrive: fold now, no waiting
      else if (n_seg == 0) duty0 = tl;                                            // come back at the end
      else duty1 = tl;
      ++n_seg;
    }
    a += k1 - k0;
  }
  if (p.stamps && threadIdx.x == 0) (p.stamps + (size_t)id * 8)[6] = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int tl = d == 0 ? duty0 : duty1;
    if (tl < 0) continue;
    int first_lid, parts;
    tile_parts(tl, first_lid, parts);
    __syncthreads();
    if (threadIdx.x == 0) {     // bounded patience: 48 polls, a few tens of microseconds
      unsigned spins = 0, ok = 1;
      while (__hip_atomic_load(p.sk_hdr + SK_FLAGS + tile_lo + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)parts) {
        if (++spins > 48u) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      bc[3] = ok;
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(bc[3])) fold(tl);
  }
  __syncthreads();
  if (p.stamps && threadIdx.x == 0) {      // {hw id | xcc << 32 | K steps in later parts << 36 | in first parts / whole tiles << 48 | those tiles << 60, begin, ..., end}
    unsigned long long* st = p.stamps + (size_t)id * 8;
    st[0] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
            ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32) | ((unsigned long long)(steps_a & 0xfff) << 36) |
            ((unsigned long long)(steps_b & 0xfff) << 48) | ((unsigned long long)(tiles_b & 0xf) << 60);
    st[1] = t_begin;
    st[7] = __builtin_amdgcn_s_memtime();
  }
  if (threadIdx.x == 0) bc[1] = __hip_atomic_fetch_add(p.sk_hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(bc[1]) == (unsigned)G - 1u) {
    // the last workgroup to finish: every ticket was drawn, every part counted, every block claimed long ago -- the header goes
    // back to zeros for the next launch on this workspace
    for (int i = threadIdx.x; i < 2 * p.sk_tiles; i += NT) __hip_atomic_store(p.sk_hdr + SK_FLAGS + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
      __hip_atomic_store(p.sk_hdr + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.sk_hdr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(p.sk_hdr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // launches seen (tests)
    }
  }
}

// ---- launchers (explicitly instantiated in conv_gemm_i1..9.hip so that the tile variants compile in parallel)
// KIND 0: conv_gemm_kernel, 2: conv_gemm_sk_kernel
template <int BM, int BN, int BK, int STAGES>
static constexpr size_t smem_bytes() { return (size_t)STAGES * (BM + BN) * BK * 2; }

template <int BM, int BN, int BK, int WM, int WN, int GLDS, int STAGES, int KIND>
static const void* variant_symbol() {
  if constexpr (KIND == 2) return reinterpret_cast<const void*>(&conv_gemm_sk_kernel<BM, BN, BK, WM, WN, GLDS, STAGES>);
  else return reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, BK, WM, WN, GLDS, STAGES>);
}
template <int BM, int BN, int BK, int WM, int WN, int GLDS, int STAGES, int KIND = 0>
void launch_variant(const ConvParams& p, dim3 grid, hipStream_t s) {
  const size_t smem = smem_bytes<BM, BN, BK, STAGES>();
  if constexpr (KIND == 2) conv_gemm_sk_kernel<BM, BN, BK, WM, WN, GLDS, STAGES><<<grid, dim3(64 * WM * WN), smem, s>>>(p);
  else conv_gemm_kernel<BM, BN, BK, WM, WN, GLDS, STAGES><<<grid, dim3(64 * WM * WN), smem, s>>>(p);
}

template <int BM, int BN, int BK, int WM, int WN, int GLDS, int STAGES, int KIND = 0>
ctta_status prepare_variant() {
  static bool done = false;
  if (done) return CTTA_OK;
  CTTA_CHECK_HIP(hipFuncSetAttribute(variant_symbol<BM, BN, BK, WM, WN, GLDS, STAGES, KIND>(),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes<BM, BN, BK, STAGES>()));
  done = true;
  return CTTA_OK;
}


#define CTTA_CONV_VARIANTS_1(X) \
  X(256, 256, 64, 2, 4, 2, 2) \
  X(256, 128, 64, 2, 2, 2, 2) \
  X(128, 128, 64, 2, 2, 0, 2) \
  X(64, 64, 64, 2, 2, 0, 2) \
  X(256, 32, 64, 4, 1, 0, 2)
#define CTTA_CONV_VARIANTS_2(X) \
  X(256, 256, 32, 2, 4, 2, 2) \
  X(256, 128, 32, 2, 2, 2, 2) \
  X(128, 128, 32, 2, 2, 0, 2) \
  X(64, 128, 64, 2, 2, 0, 2)
#define CTTA_CONV_VARIANTS_3(X) \
  X(256, 256, 32, 2, 4, 2, 3) \
  X(256, 256, 32, 2, 4, 2, 4) \
  X(256, 64, 64, 4, 1, 0, 2) \
  X(128, 64, 64, 2, 2, 0, 2)
#define CTTA_CONV_VARIANTS_4(X) \
  X(256, 128, 64, 4, 2, 0, 2) \
  X(128, 128, 64, 2, 2, 1, 2) \
  X(128, 128, 32, 2, 2, 1, 2) \
  X(256, 64, 64, 4, 1, 1, 2) \
  X(256, 128, 64, 4, 2, 1, 2) \
  X(256, 32, 64, 4, 1, 1, 2)
#define CTTA_CONV_VARIANTS_5(X) \
  X(64, 64, 64, 2, 2, 1, 2) \
  X(64, 128, 64, 2, 2, 1, 2) \
  X(128, 64, 64, 2, 2, 1, 2) \
  X(128, 128, 64, 2, 2, 2, 2) \
  X(128, 128, 32, 2, 2, 2, 2) \
  X(256, 64, 64, 4, 1, 2, 2) \
  X(256, 128, 64, 4, 2, 2, 2)
#define CTTA_CONV_VARIANTS_7(X) \
  X(512, 128, 32, 4, 2, 2, 2) \
  X(512, 128, 64, 4, 2, 2, 2)
#define CTTA_CONV_VARIANTS_6(X) \
  X(256, 32, 64, 4, 1, 2, 2) \
  X(64, 64, 64, 2, 2, 2, 2) \
  X(64, 128, 64, 2, 2, 2, 2) \
  X(128, 64, 64, 2, 2, 2, 2) \
  X(128, 128, 32, 2, 2, 1, 4) \
  X(128, 128, 32, 2, 2, 2, 3) \
  X(64, 128, 64, 2, 2, 2, 3) \
  X(256, 128, 32, 4, 2, 2, 2)
#define CTTA_CONV_VARIANTS_8(X) \
  X(64, 128, 64, 2, 2, 2, 4) \
  X(128, 128, 64, 2, 2, 2, 3) \
  X(128, 64, 64, 2, 2, 2, 3) \
  X(128, 128, 64, 2, 2, 2, 4)
// (BM, BN, BK, WM, WN, MODE, STAGES, KIND) of the stream-K kernels
#define CTTA_CONV_VARIANTS_9(X) \
  X(256, 256, 64, 2, 4, 2, 2, 2) \
  X(256, 128, 64, 2, 2, 2, 2, 2) \
  X(128, 128, 64, 2, 2, 2, 2, 2)
#define CTTA_CONV_VARIANTS_KIND(X) CTTA_CONV_VARIANTS_9(X)
#define CTTA_CONV_INSTANTIATE_K(BM, BN, BK, WM, WN, G, S, KIND)                                      \
  template void launch_variant<BM, BN, BK, WM, WN, G, S, KIND>(const ConvParams&, dim3, hipStream_t); \
  template ctta_status prepare_variant<BM, BN, BK, WM, WN, G, S, KIND>();
#define CTTA_CONV_DECLARE_K(BM, BN, BK, WM, WN, G, S, KIND)                                                     \
  extern template void launch_variant<BM, BN, BK, WM, WN, G, S, KIND>(const ConvParams&, dim3, hipStream_t); \
  extern template ctta_status prepare_variant<BM, BN, BK, WM, WN, G, S, KIND>();
#define CTTA_CONV_VARIANTS_ALL(X) CTTA_CONV_VARIANTS_8(X) CTTA_CONV_VARIANTS_1(X) CTTA_CONV_VARIANTS_2(X) CTTA_CONV_VARIANTS_3(X) CTTA_CONV_VARIANTS_4(X) CTTA_CONV_VARIANTS_5(X) CTTA_CONV_VARIANTS_6(X) CTTA_CONV_VARIANTS_7(X)
#define CTTA_CONV_INSTANTIATE(BM, BN, BK, WM, WN, G, S)                                      \
  template void launch_variant<BM, BN, BK, WM, WN, G, S, 0>(const ConvParams&, dim3, hipStream_t); \
  template ctta_status prepare_variant<BM, BN, BK, WM, WN, G, S, 0>();
#define CTTA_CONV_DECLARE(BM, BN, BK, WM, WN, G, S)                                                     \
  extern template void launch_variant<BM, BN, BK, WM, WN, G, S, 0>(const ConvParams&, dim3, hipStream_t); \
  extern template ctta_status prepare_variant<BM, BN, BK, WM, WN, G, S, 0>();
