// Fused GEGLU feed-forward of a transformer block for the narrow U-Net levels (inner width 256 after padding):
//
//   out = x_res + ff2( value * gelu(gate) ) + b2,   [value | gate] = ff1(n) + b1        diffusers/models/attention.py:276-334, 430-432
//
// As two conv_gemm launches (ff1 with the GEGLU epilogue, ff2 with the residual epilogue) the hidden activations -- M x 1024
// bf16 -- make a round trip through HBM, both launches run short K walks (K = 256: four steps of the big tile; 652 / 692
// TFLOP/s at M = 131 072) and each pays its own prologue and epilogue.  Here ONE workgroup owns BM = 128 token rows for the
// whole feed-forward:
//   1. the 128 x CP LayerNorm output tile is staged ONCE in LDS (A tile, 66 KB);
//   2. the hidden width is walked in chunks of 128 units.  Per chunk, GEMM1: wave w of 8 computes value and gate of ITS 16
//      hidden units for all 128 rows (2 weight fragments x 8 row blocks, K = CP) from the A tile; bias + GEGLU in registers,
//      the bf16 result goes to an LDS tile H[128 rows][128 hidden] (double-buffered, one barrier per chunk);
//   3. GEMM2: wave w accumulates ITS CP / 8 output channels for all 128 rows over the chunk's 128 hidden units from H; the
//      accumulators (16 fragments) live across the chunks;
//   4. epilogue from the accumulators: + b2 + residual, bf16, 8-byte stores (64 contiguous bytes per row and wave).
// HBM sees the A tile once, the residual once and the output once; the hidden activations never leave the CU.
//
// WEIGHT-STATIONARY like resunit.hip: every weight element is used by exactly one wave of a workgroup, so the waves stream
// their own fragments straight from L2 into registers (ctta_ffn_pack lays the two matrices out as ONE contiguous stream per
// wave: no address arithmetic, no LDS space, no barrier inside a GEMM); L2 -> CU traffic is 1 byte per 128 flops (the big
// conv_gemm tile: 131).  MFMA mapping as everywhere in this library (v_mfma_f32_16x16x32_bf16, weights = A operand, token rows
// = B operand, a lane owns 4 consecutive channels of one row).  Same products, same K order, same rounding points (bf16 after
// GEGLU, bf16 after the residual) as the two-launch form: bit-identical to it (tests/test_ops_gpu.py).
#include "conv_epilogue.h"

#include <string.h>

struct FfnParams {
  const bf16_t* x; int ld_x;      // [M][ld_x], CP columns used: the LayerNorm output -- or, with ln_gamma, its INPUT
  const bf16_t* w1s;              // per-wave streams, see ffn_pack_kernel
  const bf16_t* w2s;
  const float* b1;                // [2 * ffp] in ff1's packed row order ([16 value][16 gate] blocks)
  const float* b2;                // [CP]
  const bf16_t* res; int res_ld;
  bf16_t* out; int ldc;
  int M, nchunk, n_valid;         // n_valid: output columns stored (<= CP, multiple of 4)
  const float* ln_gamma; const float* ln_beta; int ln_d; float ln_eps;   // LayerNorm over the first ln_d columns while staging
  const bf16_t* w3s; const float* b3; const bf16_t* res3; int res3_ld;   // tail projection [CP][CP] (stream as w2s with S1 steps)
  // front projection: x := att[M][att_ld] (k0 columns) . W0^T + b0 + res0 -> s2 (stored to s2out, then the LayerNorm's input)
  const bf16_t* w0s; const float* b0; const bf16_t* att; int att_ld, k0; const bf16_t* res0; int res0_ld; bf16_t* s2out; int s2_ld;
};

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define FFN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int FFN_NW = 8, FFN_HC = 128;     // waves per workgroup, hidden units per chunk (16 per wave)
constexpr int FFN_PF = 2;                   // weight prefetch distance (K steps)

template <int CP, int BM>
__global__ __launch_bounds__(512, 2) void ffn_geglu_kernel(const FfnParams p) {
  constexpr int RS = CP + 8;                 // A tile row stride (bf16): 16 B of padding spreads ds_read_b128 over the banks
  constexpr int HS = FFN_HC + 8;
  constexpr int S1 = CP / 32;                // GEMM1 K steps per chunk
  constexpr int S2 = FFN_HC / 32;            // GEMM2 K steps per chunk
  constexpr int NC2 = CP / 16 / FFN_NW;      // output-channel blocks per wave
  constexpr int PB = BM / 16;                // row blocks
  constexpr int R = FFN_PF + 1;              // prefetch ring size
  static_assert(BM % 16 == 0 && PB >= S2 - 1, "row tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* atile = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* htile = atile + BM * RS;           // [2][BM][HS]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const int nchunk = p.nchunk;

  // weight streams of this wave: [chunk][S1][2][64][8] and [chunk][S2][NC2][64][8] (+ FFN_PF steps of padding at the end)
  const uint4* w1p = reinterpret_cast<const uint4*>(p.w1s) + (size_t)w * ((size_t)nchunk * S1 + FFN_PF) * 2 * 64 + lane;
  const uint4* w2p = reinterpret_cast<const uint4*>(p.w2s) + (size_t)w * ((size_t)nchunk * S2 + FFN_PF) * NC2 * 64 + lane;
  bf16x8_t r1[R][2], r2[R][NC2];
#pragma unroll
  for (int s = 0; s < FFN_PF; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) r1[s][i] = __builtin_bit_cast(bf16x8_t, w1p[(s * 2 + i) * 64]);
#pragma unroll
    for (int i = 0; i < NC2; ++i) r2[s][i] = __builtin_bit_cast(bf16x8_t, w2p[(s * NC2 + i) * 64]);
  }
  w1p += FFN_PF * 2 * 64;
  w2p += FFN_PF * NC2 * 64;

  f32x4_t acc2[NC2][PB];
#pragma unroll
  for (int i = 0; i < NC2; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t acc1[2][PB];
  const bf16_t* aw = atile + lq * RS + lg * 8;

  // ---- epilogue of a GEMM whose accumulators are acc2: acc + bias + residual -> bf16, to HBM and / or to the A tile region
  //      ([BM][RS] bf16: the operand of the next GEMM).  Descriptors bounded to M rows (rows past M read zeros / are dropped),
  //      a lane whose columns lie past n_valid gets an offset outside them: no branch, all residual loads in flight before
  //      the first store.
  const int n0 = w * NC2 * 16 + lg * 4;
  auto finish = [&](const float* bias, const bf16_t* res, const int res_ld, bf16_t* out, const int ldc, const int n_valid,
                    const bool to_lds, const bool to_hbm) {
    const __amdgpu_buffer_rsrc_t rsr =
        __builtin_amdgcn_make_buffer_rsrc((void*)res, 0, (unsigned)((size_t)p.M * res_ld * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rso =
        __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, to_hbm ? (unsigned)((size_t)p.M * ldc * 2) : 0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < NC2; ++i) {
      const int n = n0 + i * 16;
      const float4 bb = *reinterpret_cast<const float4*>(bias + n);
      const bool live = n < n_valid;
      const int vr = live ? ((m0 + lq) * res_ld + n) * 2 : (int)0x80000000, vo = live ? ((m0 + lq) * ldc + n) * 2 : (int)0x80000000;   // extents < 2^31 - 2^20 (host check)
      u32x2_t r[PB];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) r[pb] = __builtin_amdgcn_raw_buffer_load_b64(rsr, vr + pb * 16 * res_ld * 2, 0, 0);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const f32x4_t a = acc2[i][pb];
        const float v0 = a[0] + bb.x + __uint_as_float(r[pb][0] << 16);
        const float v1 = a[1] + bb.y + __uint_as_float(r[pb][0] & 0xffff0000u);
        const float v2 = a[2] + bb.z + __uint_as_float(r[pb][1] << 16);
        const float v3 = a[3] + bb.w + __uint_as_float(r[pb][1] & 0xffff0000u);
        u32x2_t pk;
        pk[0] = pack2bf(v0, v1);
        pk[1] = pack2bf(v2, v3);
        if (to_lds) *reinterpret_cast<u32x2_t*>(atile + (pb * 16 + lq) * RS + n) = pk;
        if (to_hbm) __builtin_amdgcn_raw_buffer_store_b64(pk, rso, vo + pb * 16 * ldc * 2, 0, 0);
      }
    }
  };
  // a cp-wide GEMM of this wave's NC2 output-channel blocks over `steps` K steps: B operand = LDS rows of stride `rs` (bf16)
  // at `base`, A operand = the wave's stream `wp` ([step][NC2][64][8], FFN_PF steps of padding behind it)
  auto gemm_rows = [&](const uint4* wp, const bf16_t* base, const int rs, const int steps) {
    bf16x8_t r0[R][NC2];
#pragma unroll
    for (int s = 0; s < FFN_PF; ++s)
#pragma unroll
      for (int i = 0; i < NC2; ++i) r0[s][i] = __builtin_bit_cast(bf16x8_t, wp[(s * NC2 + i) * 64]);
    wp += FFN_PF * NC2 * 64;
    const bf16_t* bw = base + lq * rs + lg * 8;
    for (int st = 0; st < steps; ++st) {
#pragma unroll
      for (int i = 0; i < NC2; ++i) r0[FFN_PF][i] = __builtin_bit_cast(bf16x8_t, wp[(st * NC2 + i) * 64]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(bw + pb * 16 * rs + st * 32));
#pragma unroll
        for (int i = 0; i < NC2; ++i)
          acc2[i][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r0[0][i], bf, acc2[i][pb], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < FFN_PF; ++s)
#pragma unroll
        for (int i = 0; i < NC2; ++i) r0[s][i] = r0[s + 1][i];
    }
  };

  // ---- 1. A tile -> LDS.  CP / 8 lanes share a row, 8 consecutive channels each; with ln_gamma the rows are layer-normalised
  //         on the way -- the lane mapping, the summation order (ln_group_sum) and the arithmetic of ctta_layernorm's kernel
  //         for this width, bit for bit.  Source: global rows (rows past M read as zeros through the descriptor's bounds
  //         check), or -- FRONT -- the rows a projection GEMM of this workgroup has just left in the tile region.
  {
    constexpr int CV = CP / 8, RPS = 512 / CV, NSW = BM / RPS;
    const int cc = tid % CV, rr = tid / CV;
    u32x4_t v[NSW];
    if (p.w0s) {
      // front projection (attn2.to_out + residual, attention.py:318-327): s2 = att . W0^T + b0 + s1, K = k0 (the head-padded
      // attention width).  The att tile [BM][k0 + 8] starts at the tile region's base and may reach into H (unused so far).
      const int k0 = p.k0, rs0 = k0 + 8, cv0 = k0 / 8, total = BM * cv0;
      const __amdgpu_buffer_rsrc_t rsa =
          __builtin_amdgcn_make_buffer_rsrc((void*)p.att, 0, (unsigned)((size_t)p.M * p.att_ld * 2), 0x00020000);
      for (int base = 0; base < total; base += 4 * 512) {
        u32x4_t t4[4];
        int row[4], col[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 512 + tid;
          row[u] = idx / cv0;
          col[u] = idx - row[u] * cv0;
          t4[u] = __builtin_amdgcn_raw_buffer_load_b128(rsa, idx < total ? ((m0 + row[u]) * p.att_ld + col[u] * 8) * 2 : (int)0x80000000, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (base + u * 512 + tid < total)
            *reinterpret_cast<uint4*>(atile + row[u] * rs0 + col[u] * 8) = make_uint4(t4[u][0], t4[u][1], t4[u][2], t4[u][3]);
      }
      __syncthreads();
      gemm_rows(reinterpret_cast<const uint4*>(p.w0s) + (size_t)w * (k0 / 32 + FFN_PF) * NC2 * 64 + lane, atile, rs0, k0 / 32);
      __syncthreads();                 // every wave is done with the att tile: s2 takes its place
      finish(p.b0, p.res0, p.res0_ld, p.s2out, p.s2_ld, CP, true, true);
      // the feed-forward's residual is read back from s2out by the lanes that wrote it: the stores are acknowledged first
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NC2; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < NSW; ++u) {
        const uint4 t = *reinterpret_cast<const uint4*>(atile + (rr + u * RPS) * RS + cc * 8);
        v[u] = (u32x4_t){t.x, t.y, t.z, t.w};
      }
    } else {
      const __amdgpu_buffer_rsrc_t rsx =
          __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((size_t)p.M * p.ld_x * 2), 0x00020000);
#pragma unroll
      for (int u = 0; u < NSW; ++u)
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ((m0 + rr + u * RPS) * p.ld_x + cc * 8) * 2, 0, 0);
    }
    if (p.ln_gamma) {
      const int d = p.ln_d;
      const float inv_d = 1.0f / (float)d, eps = p.ln_eps;
      float g[8], bt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = cc * 8 + e;
        g[e] = c < d ? p.ln_gamma[c] : 0.f;
        bt[e] = c < d ? p.ln_beta[c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < NSW; ++u) {
        float f[8];
        unpack8(make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]), f);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (cc * 8 + e >= d) f[e] = 0.f;
          sum += f[e];
        }
        const float mean = ln_group_sum<CV>(sum) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (cc * 8 + e < d) { const float t = f[e] - mean; q += t * t; }
        const float rstd = rsqrtf(ln_group_sum<CV>(q) * inv_d + eps);
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = cc * 8 + e < d ? (f[e] - mean) * rstd * g[e] + bt[e] : 0.f;
        *reinterpret_cast<uint4*>(atile + (rr + u * RPS) * RS + cc * 8) = pack8(o8);
      }
    } else {
#pragma unroll
      for (int u = 0; u < NSW; ++u)
        *reinterpret_cast<uint4*>(atile + (rr + u * RPS) * RS + cc * 8) = make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]);
    }
  }
  __syncthreads();

  const float* b1p = p.b1 + w * 32 + lg * 4;

  // GEMM1 of the next chunk: value / gate of this wave's 16 hidden units, all rows, from the A tile
  auto gemm1 = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < PB; ++j) acc1[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < S1; ++st) {
#pragma unroll
      for (int i = 0; i < 2; ++i) r1[FFN_PF][i] = __builtin_bit_cast(bf16x8_t, w1p[(st * 2 + i) * 64]);
      __builtin_amdgcn_sched_barrier(0);     // the prefetch stays HERE, two steps ahead of its use (left alone the scheduler
                                             // sinks every load to the MFMA that consumes it: s_waitcnt vmcnt(0) per step)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(aw + pb * 16 * RS + st * 32));
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc1[i][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r1[0][i], bf, acc1[i][pb], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < FFN_PF; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) r1[s][i] = r1[s + 1][i];
    }
    w1p += S1 * 2 * 64;
  };
  // bias + GEGLU of row blocks [pb0, pb1) from the GEMM1 accumulators -> H
  auto geglu_rows = [&](const float4 bv, const float4 bg, bf16_t* hw, const int pb0, const int pb1) {
#pragma unroll
    for (int pb = pb0; pb < pb1; ++pb) {
      const f32x4_t av = acc1[0][pb], ag = acc1[1][pb];
      const float o0 = (av[0] + bv.x) * gelu_erf_f(ag[0] + bg.x);
      const float o1 = (av[1] + bv.y) * gelu_erf_f(ag[1] + bg.y);
      const float o2 = (av[2] + bv.z) * gelu_erf_f(ag[2] + bg.z);
      const float o3 = (av[3] + bv.w) * gelu_erf_f(ag[3] + bg.w);
      uint2 pk;
      pk.x = pack2bf(o0, o1);
      pk.y = pack2bf(o2, o3);
      *reinterpret_cast<uint2*>(hw + pb * 16 * HS) = pk;
    }
  };
  // One K step of GEMM2(c): this wave's output channels += H[c & 1] x W2 chunk
  auto gemm2_step = [&](const bf16_t* hr, const int ks) {
#pragma unroll
    for (int i = 0; i < NC2; ++i) r2[FFN_PF][i] = __builtin_bit_cast(bf16x8_t, w2p[(ks * NC2 + i) * 64]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(hr + pb * 16 * HS + ks * 32));
#pragma unroll
      for (int i = 0; i < NC2; ++i)
        acc2[i][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r2[0][i], bf, acc2[i][pb], 0, 0, 0);
    }
  };
  auto ring2 = [&]() {
#pragma unroll
    for (int s = 0; s < FFN_PF; ++s)
#pragma unroll
      for (int i = 0; i < NC2; ++i) r2[s][i] = r2[s + 1][i];
  };
  // chunk a: GEGLU(a) -> H[a & 1], interleaved (when WITH2) with GEMM2(a - 1) from H[(a - 1) & 1]: per K step of GEMM2 the
  // GELUs of a quarter of the row blocks sit in the same scheduling region, so the VALU work runs under the matrix pipe
  auto fused = [&](const int a, const bool with2) {
    const float4 bv = *reinterpret_cast<const float4*>(b1p + (size_t)a * (FFN_NW * 32));
    const float4 bg = *reinterpret_cast<const float4*>(b1p + (size_t)a * (FFN_NW * 32) + 16);
    bf16_t* hw = htile + (a & 1) * (BM * HS) + lq * HS + w * 16 + lg * 4;
    const bf16_t* hr = htile + ((a - 1) & 1) * (BM * HS) + lq * HS + lg * 8;
#pragma unroll
    for (int ks = 0; ks < S2; ++ks) {
      if (with2) gemm2_step(hr, ks);
      geglu_rows(bv, bg, hw, ks * PB / S2, (ks + 1) * PB / S2);
      if (with2) ring2();
    }
    if (with2) w2p += S2 * NC2 * 64;
  };

  // Barrier a closes H[a & 1] (every wave has written its 16 hidden units of chunk a); GEMM2(a - 1) of the next block reads
  // H[(a - 1) & 1] while GEGLU(a) writes the other buffer, which the barrier before it has freed (every GEMM2(a - 2) done).
  gemm1();
  fused(0, false);
  FFN_LDS_BARRIER();
#pragma unroll 1
  for (int a = 1; a < nchunk; ++a) {
    gemm1();
    fused(a, true);
    FFN_LDS_BARRIER();
  }
  {
    const bf16_t* hr = htile + ((nchunk - 1) & 1) * (BM * HS) + lq * HS + lg * 8;
#pragma unroll
    for (int ks = 0; ks < S2; ++ks) {
      gemm2_step(hr, ks);
      ring2();
    }
  }

  // ---- epilogue
  const bf16_t* res = p.w0s ? p.s2out : p.res;
  const int res_ld = p.w0s ? p.s2_ld : p.res_ld;
  if (!p.w3s) {
    finish(p.b2, res, res_ld, p.out, p.ldc, p.n_valid, false, true);
    return;
  }
  // ---- tail projection (proj_out of the Transformer2DModel, transformer_2d.py:  out = proj(s3) + bias + block input): the
  //      feed-forward result s3 (bf16, exactly what the two-launch form stores) becomes the A tile of one more GEMM with K = CP
  //      -- every GEMM1 is behind the loop's last barrier, so the LayerNorm tile is dead; s3 never reaches HBM.
  finish(p.b2, res, res_ld, nullptr, 0, CP, true, false);
#pragma unroll
  for (int i = 0; i < NC2; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc2[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  FFN_LDS_BARRIER();
  gemm_rows(reinterpret_cast<const uint4*>(p.w3s) + (size_t)w * (S1 + FFN_PF) * NC2 * 64 + lane, atile, RS, S1);
  finish(p.b3, p.res3, p.res3_ld, p.out, p.ldc, p.n_valid, false, true);
}

// ------------------------------------------------------------------------------------------
// packed ff1 [2*ffp][k_pad1] and ff2 [cp][k_pad2] (conv_gemm's layout, K contiguous) -> one stream per wave:
//   w1s[w][c][st][i][lane][8] = W1[(c*8 + w)*32 + i*16 + lq][st*32 + lg*8 ..]          i = 0 value rows, 1 gate rows
//   w2s[w][c][ks][i][lane][8] = W2[(w*NC2 + i)*16 + lq][(c*4 + ks)*32 + lg*8 ..]
// each stream followed by FFN_PF steps of zeros (the prefetch runs that far past the last chunk).
__global__ void ffn_pack_kernel(const bf16_t* __restrict__ w1, int k_pad1, const bf16_t* __restrict__ w2, int k_pad2, int cp,
                                int nchunk, bf16_t* __restrict__ w1s, bf16_t* __restrict__ w2s) {
  const int S1 = cp / 32, S2 = FFN_HC / 32, NC2 = cp / 16 / FFN_NW;
  const long long per1 = ((long long)nchunk * S1 + FFN_PF) * 2 * 64, per2 = ((long long)nchunk * S2 + FFN_PF) * NC2 * 64;
  const long long total = FFN_NW * (per1 + per2);
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < FFN_NW * per1) {
      const int wv = (int)(t / per1);
      const long long r = t - wv * per1;
      const int lane = (int)(r & 63), i = (int)((r >> 6) & 1);
      const long long s = r >> 7;
      const int c = (int)(s / S1), st = (int)(s % S1);
      if (c < nchunk)
        v = *reinterpret_cast<const uint4*>(w1 + (size_t)((c * FFN_NW + wv) * 32 + i * 16 + (lane & 15)) * k_pad1 + st * 32 + (lane >> 4) * 8);
      reinterpret_cast<uint4*>(w1s)[t] = v;
    } else {
      const long long t2 = t - FFN_NW * per1;
      const int wv = (int)(t2 / per2);
      const long long r = t2 - wv * per2;
      const int lane = (int)(r & 63);
      const long long f = r >> 6;
      const int i = (int)(f % NC2);
      const long long s = f / NC2;
      const int c = (int)(s / S2), ks = (int)(s % S2);
      if (c < nchunk)
        v = *reinterpret_cast<const uint4*>(w2 + (size_t)((wv * NC2 + i) * 16 + (lane & 15)) * k_pad2 + (c * S2 + ks) * 32 + (lane >> 4) * 8);
      reinterpret_cast<uint4*>(w2s)[t2] = v;
    }
  }
}

// a projection [cp][k_pad] with k used columns -> stream[w][st][i][lane][8] = W[(w*NC2 + i)*16 + lq][st*32 + lg*8 ..], st < k / 32,
// + FFN_PF steps of zeros (gemm_rows' operand: the tail projection with k = cp, the front projection with k = the attention width)
__global__ void ffn_pack_proj_kernel(const bf16_t* __restrict__ w3, int k_pad, int k, int cp, bf16_t* __restrict__ w3s) {
  const int S = k / 32, NC2 = cp / 16 / FFN_NW;
  const long long per = (long long)(S + FFN_PF) * NC2 * 64, total = FFN_NW * per;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int wv = (int)(t / per);
    const long long r = t - wv * per;
    const int lane = (int)(r & 63);
    const long long f = r >> 6;
    const int i = (int)(f % NC2), st = (int)(f / NC2);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (st < S) v = *reinterpret_cast<const uint4*>(w3 + (size_t)((wv * NC2 + i) * 16 + (lane & 15)) * k_pad + st * 32 + (lane >> 4) * 8);
    reinterpret_cast<uint4*>(w3s)[t] = v;
  }
}

static size_t ffn_smem(int cp, int bm, int k0) {
  const size_t main_ = (size_t)bm * (cp + 8) * 2 + (size_t)2 * bm * (FFN_HC + 8) * 2, front = (size_t)bm * (k0 + 8) * 2;
  return front > main_ ? front : main_;
}
static bool ffn_shape_ok(int cp, int ffp) { return (cp == 256 || cp == 512) && ffp >= 2 * FFN_HC && ffp % FFN_HC == 0; }

extern "C" int ctta_ffn_geglu_supported(int cp, int ffp) {
  if (!ctta_opt(CTTA_OPT_FFN_FUSE)) return 0;
  return ffn_shape_ok(cp, ffp) ? 1 : 0;
}

extern "C" size_t ctta_ffn_pack_bytes(int cp, int ffp) {
  if (!ffn_shape_ok(cp, ffp)) return 0;
  const size_t nchunk = ffp / FFN_HC, S1 = cp / 32, S2 = FFN_HC / 32, NC2 = cp / 16 / FFN_NW;
  return (size_t)FFN_NW * ((nchunk * S1 + FFN_PF) * 2 + (nchunk * S2 + FFN_PF) * NC2) * 64 * 16;
}

extern "C" ctta_status ctta_ffn_pack(const void* w1_packed, int k_pad1, const void* w2_packed, int k_pad2, int cp, int ffp,
                                     void* dst, void* stream) {
  CTTA_REQUIRE(w1_packed && w2_packed && dst, "ffn_pack: null pointer");
  CTTA_REQUIRE(ffn_shape_ok(cp, ffp) && k_pad1 >= cp && k_pad2 >= ffp && k_pad1 % 8 == 0 && k_pad2 % 8 == 0,
               "ffn_pack: cp=%d (256 or 512), ffp=%d (a multiple of %d, >= %d), k_pad1=%d >= cp, k_pad2=%d >= ffp", cp, ffp, FFN_HC,
               2 * FFN_HC, k_pad1, k_pad2);
  const int nchunk = ffp / FFN_HC;
  const size_t per1 = ((size_t)nchunk * (cp / 32) + FFN_PF) * 2 * 64;
  bf16_t* w1s = (bf16_t*)dst;
  bf16_t* w2s = w1s + (size_t)FFN_NW * per1 * 8;
  hipLaunchKernelGGL(ffn_pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w1_packed, k_pad1,
                     (const bf16_t*)w2_packed, k_pad2, cp, nchunk, w1s, w2s);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

template <int CP, int BM>
static ctta_status launch_ffn(const FfnParams& p, hipStream_t s) {
  const size_t smem = ffn_smem(CP, BM, p.w0s ? p.k0 : 0);
  CTTA_REQUIRE(smem <= (size_t)160 * 1024, "ffn_block: the %d-row tile of the front projection (k = %d) does not fit the LDS", BM, p.k0);
  static size_t configured = 0;
  if (smem > configured) {
    CTTA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_geglu_kernel<CP, BM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    configured = smem;
  }
  ffn_geglu_kernel<CP, BM><<<dim3((unsigned)((p.M + BM - 1) / BM)), dim3(512), smem, s>>>(p);
  return CTTA_OK;
}

static int g_ffn_bm = 0;   // tools/ffn_bench.py: force the row tile (0 = the rule below)
extern "C" void ctta_ffn_debug_rows(int bm) { g_ffn_bm = bm; }

// Row tiles per width.  One workgroup per CU (LDS), so a launch takes ceil(tiles / CUs) rounds of about (BM + 16) row-times
// each: the tile is chosen to make that product smallest -- 144 = 9 x 16 rows turn the distillation's M = 9 x 4096 / 18 x 4096
// into exactly 1 / 2 rounds of 256 tiles where 128-row tiles need 2 / 3 (the last one 12.5 / 25 % full).
static const int kFfnRows256[] = {128, 144}, kFfnRows512[] = {64, 80, 48};
static int ffn_pick_rows(int cp, long long M, long long* rounds_out, long long* tiles_out) {
  const int* cand = cp == 256 ? kFfnRows256 : kFfnRows512;
  const int n = cp == 256 ? 2 : 3;
  const int cus = ctta_cu_count();
  int best = cand[0];
  long long best_cost = -1;
  for (int i = 0; i < n; ++i) {
    const long long tiles = (M + cand[i] - 1) / cand[i], rounds = (tiles + cus - 1) / cus, cost = rounds * (cand[i] + 16);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand[i]; *rounds_out = rounds; *tiles_out = tiles; }
  }
  return best;
}

// 1: M rows of this (cp, ffp) are better off in the fused kernel -- the shape fits and the best row tile fills its rounds to
// >= 60 % (or there is only one round): below that the two conv_gemm launches, whose small tiles fill the chip, win (measured:
// M = 36 864 on 128-row tiles, 288 tiles = 2 rounds at 56 %: 0.103 vs 0.092 ms).  A pure function of its arguments and the CU
// count -- the option is read by ctta_ffn_geglu_supported when a handle decides to keep the weight streams.
extern "C" int ctta_ffn_geglu_wanted(int cp, int ffp, int64_t M) {
  if (!ffn_shape_ok(cp, ffp) || M < 1) return 0;
  long long rounds = 0, tiles = 0;
  (void)ffn_pick_rows(cp, M, &rounds, &tiles);
  if (rounds == 1) return tiles * 4 >= ctta_cu_count() ? 1 : 0;      // a launch that leaves > 3/4 of the CUs idle: small tiles
  return tiles * 10 >= rounds * ctta_cu_count() * 6 ? 1 : 0;
}

static bool ffn_proj_ok(int cp, int k) { return (cp == 256 || cp == 512) && k >= 32 && k % 32 == 0 && k <= 1024; }
extern "C" size_t ctta_ffn_proj_pack_bytes(int cp, int k) {
  if (!ffn_proj_ok(cp, k)) return 0;
  return (size_t)FFN_NW * (k / 32 + FFN_PF) * (cp / 16 / FFN_NW) * 64 * 16;
}

extern "C" ctta_status ctta_ffn_proj_pack(const void* w_packed, int k_pad, int k, int cp, void* dst, void* stream) {
  CTTA_REQUIRE(w_packed && dst, "ffn_proj_pack: null pointer");
  CTTA_REQUIRE(ffn_proj_ok(cp, k) && k_pad >= k && k_pad % 8 == 0, "ffn_proj_pack: cp=%d (256 or 512), k=%d (a multiple of 32, <= 1024), k_pad=%d >= k", cp, k, k_pad);
  hipLaunchKernelGGL(ffn_pack_proj_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w_packed, k_pad, k, cp, (bf16_t*)dst);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" void ctta_ffn_desc_init(ctta_ffn_desc* d) { memset(d, 0, sizeof(*d)); }

extern "C" ctta_status ctta_ffn_block(const ctta_ffn_desc* d, void* stream) {
  CTTA_REQUIRE(d, "ffn_block: null descriptor");
  const int cp = d->cp, ffp = d->ffp, ld_x = d->ld_x, res_ld = d->res_ld, ldc = d->ldc, n_valid = d->n_valid;
  const int64_t M = d->M;
  const bool front = d->front_packed != nullptr;
  CTTA_REQUIRE((front || (d->x && d->res)) && d->packed && d->b1 && d->b2 && d->out, "ffn_block: null pointer");
  CTTA_REQUIRE(ffn_shape_ok(cp, ffp), "ffn_block: cp=%d ffp=%d is outside the fused kernel's range (cp 256 or 512, ffp a multiple of %d, >= %d)",
               cp, ffp, FFN_HC, 2 * FFN_HC);
  CTTA_REQUIRE(M >= 1 && (front || (ld_x >= cp && ld_x % 8 == 0 && (long long)(M + 144) * ld_x * 2 < 0x7FF00000LL &&
                                    (long long)(M + 144) * res_ld * 2 < 0x7FF00000LL && res_ld % 4 == 0 &&
                                    res_ld >= (d->proj_packed ? cp : n_valid))) &&
               (long long)(M + 144) * ldc * 2 < 0x7FF00000LL && ldc % 4 == 0 && n_valid > 0 && n_valid <= cp && n_valid % 4 == 0 && ldc >= n_valid,
               "ffn_block: bad extents (M=%lld ld_x=%d res_ld=%d ldc=%d n_valid=%d)", (long long)M, ld_x, res_ld, ldc, n_valid);
  CTTA_REQUIRE(!front || (d->front_bias && d->att && d->front_res && d->s2_out && ffn_proj_ok(cp, d->front_k) && d->att_ld >= d->front_k &&
                          d->att_ld % 8 == 0 && d->front_res_ld >= cp && d->front_res_ld % 4 == 0 && d->s2_ld >= cp && d->s2_ld % 8 == 0 &&
                          (long long)(M + 144) * d->att_ld * 2 < 0x7FF00000LL && (long long)(M + 144) * d->front_res_ld * 2 < 0x7FF00000LL &&
                          (long long)(M + 144) * d->s2_ld * 2 < 0x7FF00000LL),
               "ffn_block: the front projection needs its bias, operand (att_ld=%d >= front_k=%d, a multiple of 32), residual and the s2 "
               "destination (s2_ld=%d >= cp, a multiple of 8)", d->att_ld, d->front_k, d->s2_ld);
  CTTA_REQUIRE(!d->ln_gamma || (d->ln_beta && d->ln_d > 0 && d->ln_d <= cp), "ffn_block: LayerNorm on load needs gamma, beta and 0 < ln_d=%d <= cp", d->ln_d);
  CTTA_REQUIRE(!d->proj_packed || (d->proj_bias && d->proj_res && d->proj_res_ld >= n_valid && d->proj_res_ld % 4 == 0 &&
                                   (long long)(M + 144) * d->proj_res_ld * 2 < 0x7FF00000LL),
               "ffn_block: the tail projection needs its bias and its residual (proj_res_ld=%d >= n_valid)", d->proj_res_ld);
  FfnParams p;
  memset(&p, 0, sizeof(p));
  const int nchunk = ffp / FFN_HC;
  const size_t per1 = ((size_t)nchunk * (cp / 32) + FFN_PF) * 2 * 64;
  p.x = (const bf16_t*)d->x; p.ld_x = ld_x;
  p.w1s = (const bf16_t*)d->packed; p.w2s = p.w1s + (size_t)FFN_NW * per1 * 8;
  p.b1 = d->b1; p.b2 = d->b2; p.res = (const bf16_t*)d->res; p.res_ld = res_ld; p.out = (bf16_t*)d->out; p.ldc = ldc;
  p.M = (int)M; p.nchunk = nchunk; p.n_valid = n_valid;
  p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.ln_d = d->ln_d; p.ln_eps = d->ln_eps;
  p.w3s = (const bf16_t*)d->proj_packed; p.b3 = d->proj_bias; p.res3 = (const bf16_t*)d->proj_res; p.res3_ld = d->proj_res_ld;
  p.w0s = (const bf16_t*)d->front_packed; p.b0 = d->front_bias; p.att = (const bf16_t*)d->att; p.att_ld = d->att_ld; p.k0 = d->front_k;
  p.res0 = (const bf16_t*)d->front_res; p.res0_ld = d->front_res_ld; p.s2out = (bf16_t*)d->s2_out; p.s2_ld = d->s2_ld;
  long long rounds = 0, tiles = 0;
  int bm = ffn_pick_rows(cp, M, &rounds, &tiles);
  if (g_ffn_bm) bm = g_ffn_bm;
  hipStream_t s = (hipStream_t)stream;
  const bool prof = ctta_prof_active();
  // 2 M cp (2 ffp + ffp [+ cp]) flops as one "launch" of K = 3 ffp [+ cp]
  if (prof) ctta_prof_begin(0, 44, M, cp, 3LL * ffp + (d->proj_packed ? cp : 0) + (front ? d->front_k : 0), 1, s);
  ctta_status st;
  if (cp == 512) st = bm == 80 ? launch_ffn<512, 80>(p, s) : bm == 48 ? launch_ffn<512, 48>(p, s) : launch_ffn<512, 64>(p, s);
  else st = bm == 144 ? launch_ffn<256, 144>(p, s) : launch_ffn<256, 128>(p, s);
  if (prof) ctta_prof_end(s);
  CTTA_TRY(st);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_ffn_geglu(const void* x, int ld_x, int64_t M, int cp, int ffp, const void* packed, const float* b1,
                                      const float* b2, const void* res, int res_ld, void* out, int ldc, int n_valid,
                                      const float* ln_gamma, const float* ln_beta, int ln_d, float ln_eps, void* stream) {
  ctta_ffn_desc d;
  memset(&d, 0, sizeof(d));
  d.x = x; d.ld_x = ld_x; d.M = M; d.cp = cp; d.ffp = ffp; d.packed = packed; d.b1 = b1; d.b2 = b2; d.res = res; d.res_ld = res_ld;
  d.out = out; d.ldc = ldc; d.n_valid = n_valid; d.ln_gamma = ln_gamma; d.ln_beta = ln_beta; d.ln_d = ln_d; d.ln_eps = ln_eps;
  return ctta_ffn_block(&d, stream);
}
