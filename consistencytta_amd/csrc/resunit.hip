// Fused HiFi-GAN ResBlock unit (C = 32 / 64 / 128 channels: four waves, three workgroups per CU; round 6: C = 256 and, for
// k = 3, C = 512 with eight waves and one workgroup per CU):
//
//   out = x + conv2( leaky_relu( conv1( leaky_relu(x, 0.1) ) + b1, 0.1 ) ) + b2        hifigan/models.py:56-63
//   (conv1: k taps, dilation d; conv2: k taps, dilation 1; both "same"-padded, stride 1)
//
// As two implicit-GEMM launches each unit moves x, lrelu(x), the intermediate and the output through HBM and
// re-gathers every input row k times from L2 while producing only C channels per row (the generic kernel runs these
// stages at 107-725 TFLOP/s).  Here one workgroup owns T output positions of one sample:
//   1. stages leaky_relu(x) for its T positions plus the (k-1)(d+1)/2 halo on each side ONCE in LDS,
//   2. conv1 over T + (k-1) positions, taps = row offsets into that tile; the result (bias, leaky_relu, bf16) is
//      written back into the same LDS buffer (the input tile is dead by then),
//   3. conv2 over the T positions from that intermediate, then the fused epilogue (bias, residual x re-read from
//      L2, optional accumulate / scale / leaky_relu of the stage fold) through an LDS transpose, so that whole
//      C-channel rows leave as 64..256-byte segments.
// HBM sees x once and the output once.
//
// MFMA mapping (v_mfma_f32_16x16x32_bf16, weights = A operand, positions = B operand, as in conv_gemm): the
// workgroup's 4 waves are split WC x WP; wave (wc, wp) owns a slice of C / WC output channels (NCB = 2 blocks of 16)
// for the positions of part wp.  WEIGHT-STATIONARY: a wave streams its own rows of the weight matrix straight from
// L2 into registers (fragment-major packing: one contiguous 1 KB load per 16x32 fragment, two K-steps ahead of the
// MFMAs) and reads every position block of the LDS tile; no barrier inside a convolution's K loop.  The weight
// traffic per workgroup is C*k*C*2 bytes per conv = 1 byte per T flops: T = 128 (C = 128) keeps L2 -> CU below
// 32 B/clk/CU at the MFMA peak; a position-split design with every wave reading all weights needs 4x that and is
// L2-bound (measured in round 1: 2x slower than the generic kernel at C = 128).
#include "conv_epilogue.h"

#include <stdlib.h>

struct ResUnitParams {
  const bf16_t* x;       // [B][L][C] raw residual stream
  const bf16_t* w1f;     // conv1 weights, fragment-major [C/16][k*C/32][64][8]
  const bf16_t* w2f;
  const float* b1;
  int L, k, dil;
  float slope;
  ConvParams epi;        // conv2's epilogue: bias = b2, res = x, out, accumulate, alpha, out_act / out_slope
};

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// workgroup barrier that orders LDS traffic only (leaves global loads / stores in flight)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// KT: the tap count when it is one of the vocoder's (3 / 7 / 11: both K loops fully unrolled, so the weight prefetch keeps
// its distance without a back edge), 0 = read it from the parameters.
template <int C, int WC, int WP, int T, int KT>
__global__ __launch_bounds__(WC * WP * 64, WC * WP == 4 ? 3 : 2) void resunit_kernel(const ResUnitParams p) {
  static_assert(WC * WP == 4 || WC * WP == 8, "four waves per workgroup (C <= 128) or eight (C = 256: one workgroup per CU)");
  constexpr int NT = WC * WP * 64;           // threads per workgroup
  // Round 6: without the scheduling barrier behind each weight prefetch the compiler sinks the loads to the MFMAs that consume
  // them (s_waitcnt vmcnt(0) per K step, the same thing ffn_fused.hip's first version showed): pinned, C = 128 k = 11 / 7 / 3
  // 0.885 / 0.603 / 0.328 -> 0.749 / 0.518 / 0.304 ms per unit at B = 32 (1 262 / 1 162 / 849 TFLOP/s), C = 64 k = 11 0.503 -> 0.428
  constexpr bool RU_PIN = true;
  constexpr int RS = C + 8;                  // LDS row stride (bf16): 16 B of padding spreads ds_read_b128 over banks
  constexpr int NCB = C / 16 / WC;           // cout blocks per wave
  constexpr int NCH = C / 32;                // 32-channel chunks per tap
  constexpr int OB = T / 16 / WP;            // conv2 position blocks per wave
  constexpr int MBW = OB + 1;                // conv1 position blocks per wave (T + k - 1 <= T + 16 rows in total)
  static_assert(NCB >= 1 && T % (16 * WP) == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int wc = wave % WC, wp = wave / WC;
  const int b = blockIdx.y, l0 = blockIdx.x * T;
  const int L = p.L, k = KT ? KT : p.k, dil = p.dil;
  const float slope = p.slope;
  // diagnostic (ctta_conv_debug_stamps, tools/resunit_timeline.py): 8 words per workgroup {hw id, entry, tile staged, conv1
  // done, intermediate written, conv2 done, last store issued}
  unsigned long long* stamp = nullptr;
  if (p.epi.stamps) {
    stamp = p.epi.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    if (tid == 0) {
      stamp[0] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
                 ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);
      stamp[1] = __builtin_amdgcn_s_memtime();
    }
  }
  const int H1 = dil * (k - 1) / 2, H2 = (k - 1) / 2;
  const int MT = T + 2 * H2;                 // rows of the intermediate (conv1 outputs) this tile needs
  // conv1 runs WP * MBW blocks of 16 rows -- T/16 + WP >= ceil(MT / 16) -- whatever k is: a part that would own fewer live
  // blocks computes dead ones beside waves that are busy anyway, and the K loop carries no per-block branch (with the
  // branches the compiler drained every weight prefetch at each join: s_waitcnt vmcnt(0) per step, round-3 ISA reading)
  const int rows_in = WP * MBW * 16 + 2 * H1;      // staged input rows
  const bf16_t* xb = p.x + (size_t)b * L * C;
  const unsigned sample_bytes = (unsigned)L * C * 2;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, sample_bytes, 0x00020000);

  // ---- 1. leaky_relu(x) tile -> LDS (zero outside the sequence: F.conv1d pads the ACTIVATED signal with zeros; the
  //         descriptor's bounds check returns those zeros, so the loop is branch-free and SU loads are in flight per thread)
  {
    constexpr int SU = 4, CV = C / 8, RPS = NT / CV;           // uint4 per row, rows per sweep of the workgroup
    const int cc = tid % CV, rr = tid / CV;
    // the whole byte offset in the VGPR: the bounds check looks at it alone, and a negative one (left halo of the first
    // tile) must stay out of range whatever the row advance is
    const int voff = ((l0 - H1 - H2 + rr) * C + cc * 8) * 2;
    bf16_t* trow = tile + rr * RS + cc * 8;
    auto put = [&](const u32x4_t v, int r) {
      float f[8];
      unpack8(make_uint4(v[0], v[1], v[2], v[3]), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], f[e] * slope);       // leaky_relu for 0 <= slope <= 1 (checked by the host)
      *reinterpret_cast<uint4*>(trow + r * RS) = pack8(f);
    };
    int r0 = 0;
    for (; r0 + SU * RPS <= rows_in; r0 += SU * RPS) {          // whole batches: no guard anywhere
      u32x4_t v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, voff + (r0 + u * RPS) * C * 2, 0, 0);
#pragma unroll
      for (int u = 0; u < SU; ++u) put(v[u], r0 + u * RPS);
    }
    {                                                           // the last, partial batch: all loads first, stores guarded
      u32x4_t v[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, voff + (r0 + u * RPS) * C * 2, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < SU; ++u)
        if (r0 + u * RPS + rr < rows_in) put(v[u], r0 + u * RPS);
    }
  }

  if (stamp && tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[2] = __builtin_amdgcn_s_memtime(); }

  // ---- 2. conv1: rows [mb0*16, (mb0 + MBW)*16) of the intermediate, couts [cout0, cout0 + 16*NCB)
  const int cb0 = wc * NCB;
  const int mb0 = wp * MBW;
  f32x4_t acc[NCB][MBW];
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < MBW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float4 bias1[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) bias1[cb] = *reinterpret_cast<const float4*>(p.b1 + (cb0 + cb) * 16 + lg * 4);
  const int nsteps = k * NCH;
  {
    const uint4* wf = reinterpret_cast<const uint4*>(p.w1f) + (size_t)cb0 * nsteps * 64 + lane;
    bf16x8_t a0[NCB], a1[NCB], a2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      a0[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps) * 64]);
      a1[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + (nsteps > 1 ? 1 : 0)) * 64]);
    }
    __syncthreads();
    const bf16_t* xw = tile + (mb0 * 16 + lq) * RS + lg * 8;
    int st = 0;
#pragma unroll
    for (int tap = 0; tap < k; ++tap) {
      const bf16_t* xt = xw + tap * dil * RS;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch, ++st) {
        const int nx = st + 2 < nsteps ? st + 2 : nsteps - 1;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) a2[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + nx) * 64]);
        if (RU_PIN) __builtin_amdgcn_sched_barrier(0);     // the prefetch stays two steps ahead of its use (see ffn_fused.hip)
#pragma unroll
        for (int pb = 0; pb < MBW; ++pb) {
          const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xt + ch * 32 + pb * 16 * RS));
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[cb], bf, acc[cb][pb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { a0[cb] = a1[cb]; a1[cb] = a2[cb]; }
      }
    }
  }
  if (stamp && tid == 0) stamp[3] = __builtin_amdgcn_s_memtime();
  __syncthreads();   // every wave is done with the input tile: the intermediate takes its place
  {
#pragma unroll
    for (int pb = 0; pb < MBW; ++pb) {
      {
        const int m = (mb0 + pb) * 16 + lq;          // intermediate row <-> sequence position l0 - H2 + m
        const int pos = l0 - H2 + m;
        const bool live = m < MT && (unsigned)pos < (unsigned)L;   // conv2 zero-pads the activated intermediate
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int n = (cb0 + cb) * 16 + lg * 4;
          const float4 bb = bias1[cb];
          float v[4] = {acc[cb][pb][0] + bb.x, acc[cb][pb][1] + bb.y, acc[cb][pb][2] + bb.z, acc[cb][pb][3] + bb.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = fmaxf(v[e], v[e] * slope);
            v[e] = live ? t : 0.f;
          }
          uint2 pk;
          pk.x = pack2bf(v[0], v[1]);
          pk.y = pack2bf(v[2], v[3]);
          *reinterpret_cast<uint2*>(tile + m * RS + n) = pk;
        }
      }
    }
  }

  if (stamp && tid == 0) stamp[4] = __builtin_amdgcn_s_memtime();

  // ---- 3. conv2 over the T output positions: rows [ob0*16, (ob0 + OB)*16), taps = consecutive intermediate rows
  const int ob0 = wp * OB;
  f32x4_t acc2[NCB][OB];
#pragma unroll
  for (int i = 0; i < NCB; ++i)
#pragma unroll
    for (int j = 0; j < OB; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  {
    const uint4* wf = reinterpret_cast<const uint4*>(p.w2f) + (size_t)cb0 * nsteps * 64 + lane;
    bf16x8_t a0[NCB], a1[NCB], a2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      a0[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps) * 64]);
      a1[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + (nsteps > 1 ? 1 : 0)) * 64]);
    }
    __syncthreads();
    const bf16_t* xw = tile + (ob0 * 16 + lq) * RS + lg * 8;
    int st = 0;
#pragma unroll
    for (int tap = 0; tap < k; ++tap) {
      const bf16_t* xt = xw + tap * RS;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch, ++st) {
        const int nx = st + 2 < nsteps ? st + 2 : nsteps - 1;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) a2[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + nx) * 64]);
        if (RU_PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pb = 0; pb < OB; ++pb) {
          const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xt + ch * 32 + pb * 16 * RS));
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            acc2[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[cb], bf, acc2[cb][pb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { a0[cb] = a1[cb]; a1[cb] = a2[cb]; }
      }
    }
  }

  if (stamp && tid == 0) stamp[5] = __builtin_amdgcn_s_memtime();

  // (Round 3 also tried the epilogue straight from the accumulators -- 8-byte stores, 64-byte runs of 16 consecutive rows,
  // no LDS pass, no barriers, the residual requested before conv2: within +-3 % of this one at C = 32 and 4-13 % slower at
  // C = 64, where the extra live registers spill.  With every load and store disabled a C = 32, k = 3 launch still takes
  // 125 of its 232 us: the kernel's own VALU work -- unpack / leaky_relu / pack of the staged tile, the two fused
  // epilogues -- and the HBM time (89 us at 8 TB/s) add up rather than overlap at three workgroups per CU.)
  // ---- epilogue: two position blocks per part at a time through an fp32 LDS transpose; a row of C channels is then
  //      finished (bias, residual, accumulate, scale, LeakyReLU) and stored by C/4 consecutive lanes.
  // Straight-line code (4 passes x 4 row sweeps), buffer descriptors bounded to this sample's L rows (rows past the
  // sequence end are dropped / read as zero by the bounds check, so nothing is predicated), the row advance in the
  // scalar offset, raw LDS barriers (a __syncthreads() would also drain the stores and the prefetched residual rows):
  // the residual -- and, when accumulating, the old output -- of pass p + 1 is requested while pass p is finished.
  // A rolled loop over epilogue_wide4 made every sweep wait for its own residual load and for all but one of the
  // stores issued before it (the compiler's wait counts are conservative across a back edge).
  constexpr int RSF = C * 4 + 16;            // staging row stride (bytes)
  constexpr int CH_ROWS = 32;                // rows per part and pass
  constexpr int LPR = C / 4;                 // lanes per output row
  constexpr int RPP = NT / LPR;              // rows finished per sweep of the workgroup
  constexpr int NSW = WP * CH_ROWS / RPP;    // sweeps per pass
  constexpr int NP = (OB + 1) / 2;           // passes (an odd OB -- the 80-position tile of C = 512 -- leaves half a pass)
  const int col4 = tid % LPR, prow = tid / LPR;
  const ConvParams& e = p.epi;
  const float4 bias4 = e.bias ? *reinterpret_cast<const float4*>(e.bias + col4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float oslope = e.out_act == 3 ? e.out_slope : 1.0f, alpha = e.alpha;
  const bool acc_old = e.accumulate != 0;
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<bf16_t*>(e.out) + (size_t)b * L * C), 0, sample_bytes, 0x00020000);
  const int voff = ((l0 + prow) * C + col4 * 4) * 2;
  // sequence position of (pass, sweep) relative to l0 + prow: part = sweep * RPP / 32 rows of 32, OB blocks per part
  auto rel = [](int pass, int sw) { return (((sw * RPP) / CH_ROWS) * OB + pass * 2) * 16 + (sw * RPP) % CH_ROWS; };
  // the second 16-row block of the last pass does not exist when OB is odd: its sweeps are skipped (compile-time, the loops
  // are unrolled) -- their positions belong to the NEXT tile
  auto live = [](int pass, int sw) { return !((OB & 1) && pass == NP - 1 && (sw * RPP) % CH_ROWS >= 16); };
  u32x2_t rx[NSW], ro[NSW];
#pragma unroll
  for (int sw = 0; sw < NSW; ++sw) {
    rx[sw] = (u32x2_t){0u, 0u};
    if (live(0, sw)) rx[sw] = __builtin_amdgcn_raw_buffer_load_b64(rsx, voff, rel(0, sw) * C * 2, 0);
    ro[sw] = (u32x2_t){0u, 0u};
  }
  if (acc_old) {
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw)
      if (live(0, sw)) ro[sw] = __builtin_amdgcn_raw_buffer_load_b64(rso, voff, rel(0, sw) * C * 2, 0);
  }
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    LDS_BARRIER();     // the intermediate (first pass) / the previous pass's staging rows are dead
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pb = pass * 2 + h;
      if (pb < OB) {
        unsigned char* row = smem_raw + (size_t)(wp * CH_ROWS + h * 16 + lq) * RSF;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const f32x4_t a = acc2[cb][pb < OB ? pb : 0];
          *reinterpret_cast<float4*>(row + ((cb0 + cb) * 16 + lg * 4) * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
      }
    }
    LDS_BARRIER();
    float4 q[NSW];
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw)
      if (live(pass, sw)) q[sw] = *reinterpret_cast<const float4*>(smem_raw + (size_t)(prow + sw * RPP) * RSF + col4 * 16);
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) {
      if (!live(pass, sw)) continue;
      float v[4] = {q[sw].x + bias4.x, q[sw].y + bias4.y, q[sw].z + bias4.z, q[sw].w + bias4.w};
      const u32x2_t r2 = rx[sw], o2 = ro[sw];
      if (pass + 1 < NP && live(pass + 1, sw)) {
        rx[sw] = __builtin_amdgcn_raw_buffer_load_b64(rsx, voff, rel(pass + 1, sw) * C * 2, 0);
        if (acc_old) ro[sw] = __builtin_amdgcn_raw_buffer_load_b64(rso, voff, rel(pass + 1, sw) * C * 2, 0);
      }
      v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
      v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
      v[0] += __uint_as_float(o2.x << 16); v[1] += __uint_as_float(o2.x & 0xffff0000u);      // + 0 unless accumulating
      v[2] += __uint_as_float(o2.y << 16); v[3] += __uint_as_float(o2.y & 0xffff0000u);
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] *= alpha; v[c] = fmaxf(v[c], v[c] * oslope); }
      u32x2_t pk;
      pk.x = pack2bf(v[0], v[1]);
      pk.y = pack2bf(v[2], v[3]);
      __builtin_amdgcn_raw_buffer_store_b64(pk, rso, voff, rel(pass, sw) * C * 2, 0);
    }
  }
  if (stamp && tid == 0) stamp[6] = __builtin_amdgcn_s_memtime();
}

// ------------------------------------------------------------------------------------------
// A whole HiFi-GAN ResBlock -- three units  x <- x + conv2(lrelu(conv1_d(lrelu(x))))  with dilations d0, d1, d2
// (hifigan/models.py:56-63; resblock "1": convs1 dilated, convs2 plain) -- per launch, for the stages where a single unit is
// bound by its own fixed phases and its HBM round trip rather than by the matrix pipe (k = 3 and k = 7 at C = 32 / 64:
// 205-737 TFLOP/s per unit).  A workgroup owns T output positions and stages them ONCE with the halo of all
// three units, HT = sum_u (d_u + 1)(k - 1)/2 rows per side (12 for k = 3, 36 for k = 7); two LDS tiles:
//   xr : the raw residual stream x_u (bf16, exactly the values the unfused path writes to HBM between units),
//   xa : leaky_relu(x_u), later overwritten by the unit's intermediate (bias + leaky_relu of conv1), zero outside [0, L).
// Unit u computes on the rows the later units still need (the valid range shrinks by (d_u + 1)(k - 1)/2 per side and
// unit), so the halo is recomputed, not exchanged: +8 % / +12..23 % MFMA work for k = 3 / 7 at T = 256 / 128.  The result of
// units 0 and 1 goes back into xr / xa straight from the accumulators (the lane that owns (row, 4 channels) adds its
// residual and writes both tiles: no transpose); unit 2 leaves through the wide-store epilogue of resunit_kernel with the
// residual read from xr.  Same MFMA mapping, weight streaming and operation order as resunit_kernel: bit-identical to
// three resunit launches.
struct ResChainParams {
  const bf16_t* x;
  const bf16_t* w1f[3];
  const bf16_t* w2f[3];
  const float* b1[3];
  const float* b2[3];
  int L, k, dil[3];
  float slope;
  ConvParams epi;        // unit 2's epilogue: out, accumulate, alpha, out_act / out_slope
};

template <int C, int WC, int WP, int T, int MBW>
__global__ __launch_bounds__(256, 2) void reschain_kernel(const ResChainParams p) {
  static_assert(WC * WP == 4, "four waves per workgroup");
  constexpr int RS = C + 8;
  constexpr int NCB = C / 16 / WC;
  constexpr int NCH = C / 32;
  constexpr int OB = T / 16 / WP;            // position blocks per wave in the final epilogue (T % (32 * WP) == 0)
  static_assert(NCB >= 1 && T % (32 * WP) == 0 && OB <= MBW, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int wc = wave % WC, wp = wave / WC;
  const int b = blockIdx.y, l0 = blockIdx.x * T;
  const int L = p.L, k = p.k;
  const int H2 = (k - 1) / 2;
  int HT = 0;
#pragma unroll
  for (int u = 0; u < 3; ++u) HT += p.dil[u] * H2 + H2;
  const int R0 = T + 2 * HT;                 // staged rows: row r <-> sequence position l0 - HT + r
  const int rows_alloc = R0 + 32;            // slack: the last 16-row blocks of a unit may run past its range
  bf16_t* xa = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* xr = xa + (size_t)rows_alloc * RS;
  const bf16_t* xb = p.x + (size_t)b * L * C;

  // ---- stage x (raw) and leaky_relu(x) once; zero outside the sequence and in the slack rows
  for (int idx = tid; idx < rows_alloc * (C / 8); idx += 256) {
    const int r = idx / (C / 8), cc = idx - r * (C / 8);
    const int pos = l0 - HT + r;
    uint4 v = make_uint4(0, 0, 0, 0), va = v;
    if (r < R0 && (unsigned)pos < (unsigned)L) {
      v = *reinterpret_cast<const uint4*>(xb + (size_t)pos * C + cc * 8);
      float f[8];
      unpack8(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * p.slope;
      va = pack8(f);
    }
    *reinterpret_cast<uint4*>(xr + r * RS + cc * 8) = v;
    *reinterpret_cast<uint4*>(xa + r * RS + cc * 8) = va;
  }

  const int cb0 = wc * NCB;
  const int nsteps = k * NCH;
  int in_lo = 0;                             // unit u reads rows [in_lo, R0 - in_lo)
  f32x4_t acc[NCB][MBW];
#pragma unroll 1
  for (int u = 0; u < 3; ++u) {
    const int dil = p.dil[u];
    const int H1 = dil * H2;
    // ---- conv1: intermediate rows [lo1, hi1), blocks of 16 split over the WP position parts
    const int lo1 = in_lo + H1, hi1 = R0 - in_lo - H1;
    const int MB = (hi1 - lo1 + 15) / 16;
    const int mb_per = (MB + WP - 1) / WP;
    const int mb0 = wp * mb_per;
    int nmb = MB - mb0;
    if (nmb > mb_per) nmb = mb_per;
    if (nmb < 0) nmb = 0;
#pragma unroll
    for (int i = 0; i < NCB; ++i)
#pragma unroll
      for (int j = 0; j < MBW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
      const uint4* wf = reinterpret_cast<const uint4*>(p.w1f[u]) + (size_t)cb0 * nsteps * 64 + lane;
      bf16x8_t a0[NCB], a1[NCB], a2[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        a0[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps) * 64]);
        a1[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + (nsteps > 1 ? 1 : 0)) * 64]);
      }
      __syncthreads();                       // xa holds leaky_relu(x_u) for every row of this unit
      const bf16_t* xw = xa + (lo1 - H1 + mb0 * 16 + lq) * RS + lg * 8;    // tap 0 of row lo1 reads row lo1 - H1
      int st = 0;
      for (int tap = 0; tap < k; ++tap) {
        const bf16_t* xt = xw + tap * dil * RS;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch, ++st) {
          const int nx = st + 2 < nsteps ? st + 2 : nsteps - 1;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) a2[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + nx) * 64]);
          __builtin_amdgcn_sched_barrier(0);     // (as in resunit_kernel: the prefetch stays two steps ahead)
#pragma unroll
          for (int pb = 0; pb < MBW; ++pb) {
            if (pb < nmb) {
              const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xt + ch * 32 + pb * 16 * RS));
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb)
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[cb], bf, acc[cb][pb], 0, 0, 0);
            }
          }
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) { a0[cb] = a1[cb]; a1[cb] = a2[cb]; }
        }
      }
    }
    __syncthreads();   // every wave is done reading leaky_relu(x_u): the intermediate takes its place
#pragma unroll
    for (int pb = 0; pb < MBW; ++pb) {
      if (pb < nmb) {
        const int r = lo1 + (mb0 + pb) * 16 + lq;
        const int pos = l0 - HT + r;
        const bool live = r < hi1 && (unsigned)pos < (unsigned)L;      // conv2 zero-pads the activated intermediate
        if (r < rows_alloc) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            const int n = (cb0 + cb) * 16 + lg * 4;
            const float4 bb = *reinterpret_cast<const float4*>(p.b1[u] + n);
            float v[4] = {acc[cb][pb][0] + bb.x, acc[cb][pb][1] + bb.y, acc[cb][pb][2] + bb.z, acc[cb][pb][3] + bb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = live ? (v[e] > 0.f ? v[e] : v[e] * p.slope) : 0.f;
            uint2 pk;
            pk.x = pack2bf(v[0], v[1]);
            pk.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(xa + r * RS + n) = pk;
          }
        }
      }
    }
    // ---- conv2: output rows [lo2, hi2) = the next unit's input range
    const int lo2 = lo1 + H2, hi2 = hi1 - H2;
    const int MB2 = (hi2 - lo2 + 15) / 16;
    const int mb2_per = u == 2 ? OB : (MB2 + WP - 1) / WP;            // the last unit: exactly T rows, OB blocks per part
    const int mb20 = wp * mb2_per;
    int nmb2 = MB2 - mb20;
    if (nmb2 > mb2_per) nmb2 = mb2_per;
    if (nmb2 < 0) nmb2 = 0;
#pragma unroll
    for (int i = 0; i < NCB; ++i)
#pragma unroll
      for (int j = 0; j < MBW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
      const uint4* wf = reinterpret_cast<const uint4*>(p.w2f[u]) + (size_t)cb0 * nsteps * 64 + lane;
      bf16x8_t a0[NCB], a1[NCB], a2[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        a0[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps) * 64]);
        a1[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + (nsteps > 1 ? 1 : 0)) * 64]);
      }
      __syncthreads();                       // the intermediate is complete
      const bf16_t* xw = xa + (lo2 - H2 + mb20 * 16 + lq) * RS + lg * 8;
      int st = 0;
      for (int tap = 0; tap < k; ++tap) {
        const bf16_t* xt = xw + tap * RS;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch, ++st) {
          const int nx = st + 2 < nsteps ? st + 2 : nsteps - 1;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) a2[cb] = __builtin_bit_cast(bf16x8_t, wf[(size_t)(cb * nsteps + nx) * 64]);
          __builtin_amdgcn_sched_barrier(0);     // (as in resunit_kernel: the prefetch stays two steps ahead)
#pragma unroll
          for (int pb = 0; pb < MBW; ++pb) {
            if (pb < nmb2) {
              const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xt + ch * 32 + pb * 16 * RS));
#pragma unroll
              for (int cb = 0; cb < NCB; ++cb)
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[cb], bf, acc[cb][pb], 0, 0, 0);
            }
          }
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) { a0[cb] = a1[cb]; a1[cb] = a2[cb]; }
        }
      }
    }
    if (u == 2) break;                       // the last unit leaves through the wide-store epilogue below
    __syncthreads();   // every wave is done reading the intermediate: xa receives leaky_relu(x_{u+1})
#pragma unroll
    for (int pb = 0; pb < MBW; ++pb) {
      if (pb < nmb2) {
        const int r = lo2 + (mb20 + pb) * 16 + lq;
        const int pos = l0 - HT + r;
        const bool inside = (unsigned)pos < (unsigned)L;
        if (r < rows_alloc) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            const int n = (cb0 + cb) * 16 + lg * 4;
            const float4 bb = *reinterpret_cast<const float4*>(p.b2[u] + n);
            const uint2 old = *reinterpret_cast<const uint2*>(xr + r * RS + n);
            float v[4] = {acc[cb][pb][0] + bb.x, acc[cb][pb][1] + bb.y, acc[cb][pb][2] + bb.z, acc[cb][pb][3] + bb.w};
            v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
            v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
            uint2 pk;                        // x_{u+1}, bf16 like the tensor the unfused path writes between units
            pk.x = pack2bf(v[0], v[1]);
            pk.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(xr + r * RS + n) = pk;
            float w[4] = {__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u), __uint_as_float(pk.y << 16),
                          __uint_as_float(pk.y & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = inside ? (w[e] > 0.f ? w[e] : w[e] * p.slope) : 0.f;
            uint2 pa;
            pa.x = pack2bf(w[0], w[1]);
            pa.y = pack2bf(w[2], w[3]);
            *reinterpret_cast<uint2*>(xa + r * RS + n) = pa;
          }
        }
      }
    }
    in_lo += H1 + H2;
  }

  // ---- unit 2's epilogue: acc[.][0 .. OB) are the T output rows of part wp (rows HT + (wp*OB + pb)*16 + lq); fp32 LDS
  //      transpose over the (dead) xa tile, residual from xr, whole C-channel rows to HBM (resunit_kernel's epilogue)
  constexpr int RSF = C * 4 + 16;
  constexpr int CH_ROWS = 32;
  constexpr int LPR = C / 4;
  constexpr int RPP = 256 / LPR;
  constexpr int NSW = WP * CH_ROWS / RPP;
  constexpr int NP = OB / 2;
  const int col4 = tid % LPR, prow = tid / LPR;
  const ConvParams& e = p.epi;
  const float4 bias4 = *reinterpret_cast<const float4*>(p.b2[2] + col4 * 4);
  const float oslope = e.out_act == 3 ? e.out_slope : 1.0f, alpha = e.alpha;
  const bool acc_old = e.accumulate != 0;
  const unsigned sample_bytes = (unsigned)L * C * 2;
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<bf16_t*>(e.out) + (size_t)b * L * C), 0, sample_bytes, 0x00020000);
  const int voff = ((l0 + prow) * C + col4 * 4) * 2;
  auto rel = [](int pass, int sw) { return (((sw * RPP) / CH_ROWS) * OB + pass * 2) * 16 + (sw * RPP) % CH_ROWS; };
  u32x2_t ro[NSW];
#pragma unroll
  for (int sw = 0; sw < NSW; ++sw) ro[sw] = (u32x2_t){0u, 0u};
  if (acc_old) {
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) ro[sw] = __builtin_amdgcn_raw_buffer_load_b64(rso, voff, rel(0, sw) * C * 2, 0);
  }
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    LDS_BARRIER();     // the intermediate (first pass) / the previous pass's staging rows are dead
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pb = pass * 2 + h;
      unsigned char* row = smem_raw + (size_t)(wp * CH_ROWS + h * 16 + lq) * RSF;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const f32x4_t a = acc[cb][pb];
        *reinterpret_cast<float4*>(row + ((cb0 + cb) * 16 + lg * 4) * 4) = make_float4(a[0], a[1], a[2], a[3]);
      }
    }
    LDS_BARRIER();
#pragma unroll
    for (int sw = 0; sw < NSW; ++sw) {
      const float4 q = *reinterpret_cast<const float4*>(smem_raw + (size_t)(prow + sw * RPP) * RSF + col4 * 16);
      const int rr = HT + rel(pass, sw) + prow;                        // tile row of this output position
      const uint2 r2 = *reinterpret_cast<const uint2*>(xr + rr * RS + col4 * 4);
      const u32x2_t o2 = ro[sw];
      if (pass + 1 < NP && acc_old) ro[sw] = __builtin_amdgcn_raw_buffer_load_b64(rso, voff, rel(pass + 1, sw) * C * 2, 0);
      float v[4] = {q.x + bias4.x, q.y + bias4.y, q.z + bias4.z, q.w + bias4.w};
      v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
      v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
      v[0] += __uint_as_float(o2.x << 16); v[1] += __uint_as_float(o2.x & 0xffff0000u);      // + 0 unless accumulating
      v[2] += __uint_as_float(o2.y << 16); v[3] += __uint_as_float(o2.y & 0xffff0000u);
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] *= alpha; v[c] = fmaxf(v[c], v[c] * oslope); }
      u32x2_t pk;
      pk.x = pack2bf(v[0], v[1]);
      pk.y = pack2bf(v[2], v[3]);
      __builtin_amdgcn_raw_buffer_store_b64(pk, rso, voff, rel(pass, sw) * C * 2, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// packed [n][k_pad] (K contiguous, conv_gemm's layout) -> fragment-major [n/16][nsteps][64 lanes][8]: lane (lq, lg) of
// fragment (cb, st) holds W[cb*16 + lq][st*32 + lg*8 .. +7], so an A operand is one contiguous 1 KB wave load
__global__ void frag_pack_kernel(const bf16_t* __restrict__ src, int k_pad, int n_cb, int nsteps, bf16_t* __restrict__ dst) {
  const long long total = (long long)n_cb * nsteps * 64;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long f = i >> 6;
    const int st = (int)(f % nsteps), cb = (int)(f / nsteps);
    const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(cb * 16 + (lane & 15)) * k_pad + st * 32 + (lane >> 4) * 8);
    reinterpret_cast<uint4*>(dst)[i] = v;
  }
}

extern "C" ctta_status ctta_frag_pack(const void* packed, int n, int k_pad, int k_valid, void* dst, void* stream) {
  CTTA_REQUIRE(packed && dst, "frag_pack: null pointer");
  CTTA_REQUIRE(n > 0 && n % 16 == 0 && k_valid > 0 && k_valid % 32 == 0 && k_valid <= k_pad && k_pad % 8 == 0,
               "frag_pack: n=%d must be a multiple of 16 and k_valid=%d a multiple of 32 within k_pad=%d", n, k_valid, k_pad);
  const int n_cb = n / 16, nsteps = k_valid / 32;
  const long long total = (long long)n_cb * nsteps * 64;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(frag_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)packed, k_pad, n_cb,
                     nsteps, (bf16_t*)dst);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

static size_t resunit_smem(int C, int WP, int T, int k, int dil) {
  const int H1 = dil * (k - 1) / 2;
  size_t smem = (size_t)(T + 16 * WP + 2 * H1) * (C + 8) * 2;      // WP * (T/16/WP + 1) blocks of 16 rows + the conv1 halo
  const size_t stage = (size_t)WP * 32 * (C * 4 + 16);
  return stage > smem ? stage : smem;
}

template <int C, int WC, int WP, int T, int KT>
static ctta_status launch_resunit_k(const ResUnitParams& p, int batch, hipStream_t s) {
  const size_t smem = resunit_smem(C, WP, T, p.k, p.dil);
  static size_t configured = 0;
  if (smem > configured) {
    CTTA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resunit_kernel<C, WC, WP, T, KT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    configured = smem;
  }
  dim3 grid((unsigned)((p.L + T - 1) / T), (unsigned)batch);
  resunit_kernel<C, WC, WP, T, KT><<<grid, dim3(WC * WP * 64), smem, s>>>(p);
  return CTTA_OK;
}

template <int C, int WC, int WP, int T>
static ctta_status launch_resunit(const ResUnitParams& p, int batch, hipStream_t s) {
  switch (p.k) {
    case 3: return launch_resunit_k<C, WC, WP, T, 3>(p, batch, s);
    case 7: return launch_resunit_k<C, WC, WP, T, 7>(p, batch, s);
    case 11: return launch_resunit_k<C, WC, WP, T, 11>(p, batch, s);
    default: return launch_resunit_k<C, WC, WP, T, 0>(p, batch, s);
  }
}

// C = 512: positions per workgroup = the longest tile whose rows (1 040 B each, halo included) fit 160 KB AND that cuts the
// stage's 5 121 positions into rounds of 256 workgroups without a near-empty last one: 96 (54 tiles x 32 samples = 6.75 rounds;
// 128 -> 5.125 rounds measured 0.500 ms per k = 3 unit, 64 -> 0.509, 96 -> 0.449), 80 for k = 11 at dilation 5 (LDS).
static int resunit_tile512(int k, int dil) {
  if (k == 3) return resunit_smem(512, 1, 96, k, dil) <= (size_t)160 * 1024 ? 96 : resunit_smem(512, 1, 64, k, dil) <= (size_t)160 * 1024 ? 64 : 0;
  if (k != 7 && k != 11) return 0;
  if (resunit_smem(512, 1, 96, k, dil) <= (size_t)160 * 1024) return 96;
  if (resunit_smem(512, 1, 80, k, dil) <= (size_t)160 * 1024) return 80;
  return 0;
}

extern "C" int ctta_resunit_supported(int channels, int k, int dil) {
  if (!ctta_opt(CTTA_OPT_FUSED_RES)) return 0;
  if (k < 1 || k > 11 || (k & 1) == 0 || dil < 1) return 0;
  // C = 256 / 512: eight waves, one workgroup per CU; every weight byte is streamed once per T positions, so T is as long as the
  // tile's rows fit the LDS.  Measured against the two conv_gemm launches (B = 32, ms per unit): C = 256: k = 3 (T = 192) 0.52 vs
  // 0.69, k = 7 (T = 224) 0.96 vs 1.17, k = 11 (T = 224) 1.42 vs 1.67 (T = 128: 0.53 / 1.01 / 1.52); C = 512: k = 3 (T = 96) 0.45 vs
  // 0.57, k = 7 (T = 96) 0.93 vs 1.05, k = 11 (T = 96; 80 at dilation 5) 1.42 (1.58) vs 1.57 -- at T = 64 k = 7 / 11 lost (1.08 / 1.66)
  if (channels == 256) return resunit_smem(256, 1, k <= 3 ? 192 : 224, k, dil) <= (size_t)160 * 1024 ? 1 : 0;
  if (channels == 512) return resunit_tile512(k, dil) > 0 ? 1 : 0;
  if (channels != 32 && channels != 64 && channels != 128) return 0;
  const int T = channels == 128 ? 128 : channels == 64 ? 256 : 512;
  const int WP = channels == 128 ? 1 : channels == 64 ? 2 : 4;
  return resunit_smem(channels, WP, T, k, dil) <= (size_t)64 * 1024 ? 1 : 0;
}

extern "C" ctta_status ctta_resunit_conv1d(const void* x, int batch, int len, int channels, int k, int dil,
                                           const void* w1_frag, const float* b1, const void* w2_frag, const float* b2,
                                           float slope, void* out, int accumulate, float alpha, float out_slope,
                                           void* stream) {
  CTTA_REQUIRE(x && w1_frag && w2_frag && b1 && b2 && out, "resunit_conv1d: null pointer");
  CTTA_REQUIRE(batch >= 1 && len >= 1 && (long long)batch * len < (1LL << 31), "resunit_conv1d: bad extent");
  CTTA_REQUIRE(out_slope >= 0.f && out_slope <= 1.f, "resunit_conv1d: out_slope=%g must lie in [0, 1]", (double)out_slope);
  CTTA_REQUIRE(slope >= 0.f && slope <= 1.f, "resunit_conv1d: slope=%g must lie in [0, 1]", (double)slope);
  CTTA_REQUIRE(ctta_resunit_supported(channels, k, dil),
               "resunit_conv1d: channels=%d k=%d dilation=%d is outside the fused kernel's range (C in {32,64,128,256}, odd k <= 11, "
               "tile <= 64 KB of LDS (C <= 128); C = 512: k in {3, 7, 11})", channels, k, dil);
  ResUnitParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x; p.w1f = (const bf16_t*)w1_frag; p.w2f = (const bf16_t*)w2_frag; p.b1 = b1;
  p.L = len; p.k = k; p.dil = dil; p.slope = slope;
  ConvParams& e = p.epi;
  e.bias = b2; e.res = (const bf16_t*)x; e.res_ld = channels;
  e.out = out; e.ldc = channels; e.plain_out = 1; e.wide_store = 1;
  e.accumulate = accumulate ? 1 : 0; e.alpha = alpha;
  if (out_slope > 0.f) { e.out_act = 3; e.out_slope = out_slope; }
  e.howo = len; e.n = channels; e.M = batch * len;
  e.stamps = ctta_debug_stamps_current();
  hipStream_t s = (hipStream_t)stream;
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(0, 40, (long long)batch * len, channels, 2LL * k * channels, 1, s);
  ctta_status st;
  if (channels == 512) {
    const int t512 = resunit_tile512(k, dil);
    st = k == 3 ? (t512 == 96 ? launch_resunit_k<512, 8, 1, 96, 3>(p, batch, s) : launch_resunit_k<512, 8, 1, 64, 3>(p, batch, s))
       : k == 7 ? launch_resunit_k<512, 8, 1, 96, 7>(p, batch, s)
       : t512 == 96 ? launch_resunit_k<512, 8, 1, 96, 11>(p, batch, s) : launch_resunit_k<512, 8, 1, 80, 11>(p, batch, s);
  }
  else if (channels == 256) st = k <= 3 ? launch_resunit<256, 8, 1, 192>(p, batch, s) : launch_resunit<256, 8, 1, 224>(p, batch, s);
  else if (channels == 128) st = launch_resunit<128, 4, 1, 128>(p, batch, s);
  else if (channels == 64) st = launch_resunit<64, 2, 2, 256>(p, batch, s);
  else st = launch_resunit<32, 1, 4, 512>(p, batch, s);
  if (prof) ctta_prof_end(s);
  CTTA_TRY(st);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ------------------------------------------------------------------------------------------
// chained form: the three units of one ResBlock per launch (reschain_kernel)
static int reschain_tile(int channels) { return channels == 64 ? 128 : 256; }
static int reschain_halo(int k, const int* dils) {
  int ht = 0;
  for (int u = 0; u < 3; ++u) ht += (dils[u] + 1) * ((k - 1) / 2);
  return ht;
}
static size_t reschain_smem(int channels, int k, const int* dils) {
  const int T = reschain_tile(channels);
  return (size_t)(T + 2 * reschain_halo(k, dils) + 32) * (channels + 8) * 2 * 2;
}

extern "C" int ctta_reschain_supported(int channels, int k, const int* dils) {
  if (!dils) return 0;
  if (channels != 32 && channels != 64) return 0;
  if (k < 3 || k > 7 || (k & 1) == 0) return 0;
  for (int u = 0; u < 3; ++u)
    if (dils[u] < 1 || !ctta_resunit_supported(channels, k, dils[u])) return 0;
  // position blocks of the widest convolution (unit 0's conv1) per wave part must fit the kernel's accumulator file
  const int T = reschain_tile(channels), WP = channels == 64 ? 2 : 4, MBW = channels == 64 ? 7 : 6;
  const int rows = T + 2 * reschain_halo(k, dils) - (k - 1) * dils[0];
  if (((rows + 15) / 16 + WP - 1) / WP > MBW) return 0;
  return reschain_smem(channels, k, dils) <= (size_t)72 * 1024 ? 1 : 0;     // two workgroups per CU
}

template <int C, int WC, int WP, int T, int MBW>
static ctta_status launch_reschain(const ResChainParams& p, int batch, size_t smem, hipStream_t s) {
  static size_t configured = 0;
  if (smem > configured) {
    CTTA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&reschain_kernel<C, WC, WP, T, MBW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    configured = smem;
  }
  dim3 grid((unsigned)((p.L + T - 1) / T), (unsigned)batch);
  reschain_kernel<C, WC, WP, T, MBW><<<grid, dim3(256), smem, s>>>(p);
  return CTTA_OK;
}

extern "C" ctta_status ctta_reschain_conv1d(const void* x, int batch, int len, int channels, int k, const int* dils,
                                            const void* const* w1_frag, const float* const* b1,
                                            const void* const* w2_frag, const float* const* b2, float slope, void* out,
                                            int accumulate, float alpha, float out_slope, void* stream) {
  CTTA_REQUIRE(x && dils && w1_frag && w2_frag && b1 && b2 && out, "reschain_conv1d: null pointer");
  CTTA_REQUIRE(batch >= 1 && len >= 1 && (long long)batch * len < (1LL << 31), "reschain_conv1d: bad extent");
  CTTA_REQUIRE(x != out, "reschain_conv1d: the output may not alias the input (neighbouring tiles read its halo)");
  CTTA_REQUIRE(out_slope >= 0.f && out_slope <= 1.f, "reschain_conv1d: out_slope=%g must lie in [0, 1]", (double)out_slope);
  CTTA_REQUIRE(ctta_reschain_supported(channels, k, dils),
               "reschain_conv1d: channels=%d k=%d dilations=(%d,%d,%d) is outside the chained kernel's range (C in {32,64}, "
               "k in {3,5,7}, tile <= 72 KB of LDS)", channels, k, dils[0], dils[1], dils[2]);
  ResChainParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const bf16_t*)x;
  for (int u = 0; u < 3; ++u) {
    CTTA_REQUIRE(w1_frag[u] && w2_frag[u] && b1[u] && b2[u], "reschain_conv1d: null pointer in unit %d", u);
    p.w1f[u] = (const bf16_t*)w1_frag[u]; p.w2f[u] = (const bf16_t*)w2_frag[u];
    p.b1[u] = b1[u]; p.b2[u] = b2[u]; p.dil[u] = dils[u];
  }
  p.L = len; p.k = k; p.slope = slope;
  ConvParams& e = p.epi;
  e.out = out; e.ldc = channels; e.plain_out = 1; e.wide_store = 1;
  e.accumulate = accumulate ? 1 : 0; e.alpha = alpha;
  if (out_slope > 0.f) { e.out_act = 3; e.out_slope = out_slope; }
  e.howo = len; e.n = channels; e.M = batch * len;
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = reschain_smem(channels, k, dils);
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(0, 41, (long long)batch * len, channels, 6LL * k * channels, 1, s);
  ctta_status st;
  if (channels == 64) st = launch_reschain<64, 2, 2, 128, 7>(p, batch, smem, s);
  else st = launch_reschain<32, 1, 4, 256, 6>(p, batch, smem, s);
  if (prof) ctta_prof_end(s);
  CTTA_TRY(st);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
