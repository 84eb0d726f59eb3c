// Flash-style multi-head attention for the U-Net transformer blocks (head dim 51 in the
// light config, padded to 64): softmax(q k^T * scale + bias) v without materialising scores.
// Replaces F.scaled_dot_product_attention as called by AttnProcessor2_0
// (diffusers/models/attention_processor.py:1127-1129); `bias` is the additive (B, L) mask
// bias (0 / -10000) of unet_2d_condition_guided.py:793-795.
//
// CDNA4 mapping (wave64, v_mfma_f32_16x16x32_bf16): a 256-thread workgroup owns 128 queries of
// one (batch, head); each wave owns 32 queries (two 16-wide MFMA column blocks) and walks the
// keys in tiles of 64 staged in LDS.  Both products are computed TRANSPOSED so that the query
// index is always the MFMA column (lane & 15):
//     S^T[key][q] = K * Q^T      (A = K rows from LDS, B = Q^T held in registers)
//     O^T[d][q]  += V^T * P^T    (A = V^T rows from LDS, B = P^T = the S^T accumulators
//                                 re-used in place as bf16 -- no cross-lane movement)
// so the online-softmax statistics (one per query) live in the lane that also owns that
// query's accumulator columns.  V is supplied transposed ([head*64 + d][key]) by the
// projection GEMM (conv_gemm with swapped operands), which makes both LDS images K-contiguous.
#include "common.h"

#include <math.h>

#define ATT_KT 64          // keys per tile
#define ATT_LDK 80         // K tile row stride (bf16): 160 B rows are conflict-free for ds_read_b128
#define ATT_LDV 72         // V^T tile row stride: 144 B rows are conflict-free for the paired ds_read_b64

__global__ __launch_bounds__(256, 2) void attention_kernel(
    const bf16_t* __restrict__ q, int q_ld, const bf16_t* __restrict__ k, int k_ld, int k_rows,
    const bf16_t* __restrict__ vt, int vt_ld, const float* __restrict__ bias,
    bf16_t* __restrict__ out, int out_ld, int heads, int nq, int nk, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[ATT_KT * ATT_LDK];   // [key][d]
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * ATT_LDV];       // [d][key]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y;
  const int b = bh / heads, h = bh - b * heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int lq = lane & 15;        // query column within a 16-block
  const int lg = lane >> 4;        // lane group 0..3

  const bf16_t* qb = q + (size_t)b * nq * q_ld + h * 64;
  const bf16_t* kb = k + (size_t)b * k_rows * k_ld + h * 64;
  const bf16_t* vb = vt + ((size_t)b * heads + h) * 64 * vt_ld;
  const float* bb = bias ? bias + (size_t)b * nk : nullptr;

  // Q^T fragments: B operand, lane j = query, 8 consecutive d at (ds*32 + lg*8)
  bf16x8_t qf[2][2];
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    int qi = q0 + jq * 16 + lq;
    if (qi >= nq) qi = nq - 1;   // clamp (never stored)
#pragma unroll
    for (int ds = 0; ds < 2; ++ds)
      qf[jq][ds] = __builtin_bit_cast(
          bf16x8_t, *reinterpret_cast<const uint4*>(qb + (size_t)qi * q_ld + ds * 32 + lg * 8));
  }

  f32x4_t o[4][2];   // O^T accumulators: [d block][q block], rows d = jd*16 + lg*4 + r
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) o[jd][jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {-INFINITY, -INFINITY};
  float lrun[2] = {0.f, 0.f};   // per-lane partial row sums (reduced over lane groups at the end)

  const int ntiles = (nk + ATT_KT - 1) / ATT_KT;
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * ATT_KT;
    __syncthreads();   // previous tile fully consumed
    // stage K tile [64 keys][64 d] and V^T tile [64 d][64 keys]: 512 16-byte chunks each
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = tid + i * 256;
      const int r = chunk >> 3, cc = (chunk & 7) * 8;
      uint4 kv = make_uint4(0, 0, 0, 0);
      if (key0 + r < nk) kv = *reinterpret_cast<const uint4*>(kb + (size_t)(key0 + r) * k_ld + cc);
      *reinterpret_cast<uint4*>(Ks + r * ATT_LDK + cc) = kv;
      uint4 vv = make_uint4(0, 0, 0, 0);
      const int valid = nk - (key0 + cc);   // keys of this chunk that exist
      if (valid > 0) {
        vv = *reinterpret_cast<const uint4*>(vb + (size_t)r * vt_ld + key0 + cc);
        if (valid < 8) {   // zero the tail: P is exactly 0 there, but 0 * garbage must stay 0
          uint32_t wv[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e >= valid) wv[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
          vv = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
      }
      *reinterpret_cast<uint4*>(Vs + r * ATT_LDV + cc) = vv;
    }
    __syncthreads();

    // ---- S^T = K Q^T : 4 key blocks x 2 query blocks
    f32x4_t s[4][2];
#pragma unroll
    for (int ik = 0; ik < 4; ++ik) {
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) s[ik][jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const bf16x8_t kf = __builtin_bit_cast(
            bf16x8_t, *reinterpret_cast<const uint4*>(Ks + (ik * 16 + lq) * ATT_LDK + ds * 32 + lg * 8));
#pragma unroll
        for (int jq = 0; jq < 2; ++jq)
          s[ik][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[jq][ds], s[ik][jq], 0, 0, 0);
      }
    }
    // ---- scale, bias, key mask; lane holds keys key0 + ik*16 + lg*4 + r for query column lq
    float kbias[4][4];
#pragma unroll
    for (int ik = 0; ik < 4; ++ik)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + ik * 16 + lg * 4 + r;
        float add = 0.f;
        if (key >= nk) add = -INFINITY;
        else if (bb) add = bb[key] * 1.4426950408889634f;
        kbias[ik][r] = add;
      }
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
      float mx = -INFINITY;
#pragma unroll
      for (int ik = 0; ik < 4; ++ik)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = s[ik][jq][r] * scale_log2e + kbias[ik][r];
          s[ik][jq][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[jq], mx);
      const float alpha = exp2f(mrun[jq] - mnew);   // first tile: exp2(-inf) = 0
      mrun[jq] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int ik = 0; ik < 4; ++ik)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = exp2f(s[ik][jq][r] - mnew);
          s[ik][jq][r] = pv;
          ps += pv;
        }
      lrun[jq] = lrun[jq] * alpha + ps;
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        o[jd][jq][0] *= alpha; o[jd][jq][1] *= alpha;
        o[jd][jq][2] *= alpha; o[jd][jq][3] *= alpha;
      }
    }
    // ---- O^T += V^T P^T : k-slot (lg, e): e<4 -> key block 2kk, row lg*4+e ; e>=4 -> block 2kk+1
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t pf[2];
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        uint4 pk;
        pk.x = pack2bf(s[2 * kk][jq][0], s[2 * kk][jq][1]);
        pk.y = pack2bf(s[2 * kk][jq][2], s[2 * kk][jq][3]);
        pk.z = pack2bf(s[2 * kk + 1][jq][0], s[2 * kk + 1][jq][1]);
        pk.w = pack2bf(s[2 * kk + 1][jq][2], s[2 * kk + 1][jq][3]);
        pf[jq] = __builtin_bit_cast(bf16x8_t, pk);
      }
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        const bf16_t* vr = Vs + (jd * 16 + lq) * ATT_LDV + kk * 32 + lg * 4;
        const uint2 lo = *reinterpret_cast<const uint2*>(vr);        // keys (2kk)*16 + lg*4 ..+3
        const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);   // keys (2kk+1)*16 + lg*4 ..+3
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
        for (int jq = 0; jq < 2; ++jq)
          o[jd][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[jq], o[jd][jq], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane owns query (q0 + jq*16 + lq), d = jd*16 + lg*4 + {0..3}
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    float l = lrun[jq];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int qi = q0 + jq * 16 + lq;
    if (qi < nq) {
      bf16_t* orow = out + ((size_t)b * nq + qi) * out_ld + h * 64;
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        uint2 pk;
        pk.x = pack2bf(o[jd][jq][0] * inv, o[jd][jq][1] * inv);
        pk.y = pack2bf(o[jd][jq][2] * inv, o[jd][jq][3] * inv);
        *reinterpret_cast<uint2*>(orow + jd * 16 + lg * 4) = pk;
      }
    }
  }
}

extern "C" ctta_status ctta_attention(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                                      int vt_ld, const float* bias, void* out, int out_ld, int batch,
                                      int heads, int nq, int nk, float scale, void* stream) {
  CTTA_REQUIRE(q && k && vt && out, "attention: null pointer");
  CTTA_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vt_ld % 8 == 0 && out_ld % 4 == 0,
               "attention: row strides must be multiples of 8");
  CTTA_REQUIRE(nq > 0 && nk > 0 && k_rows >= nk && vt_ld >= ((nk + 7) / 8) * 8, "attention: bad lengths nq=%d nk=%d vt_ld=%d", nq, nk, vt_ld);
  dim3 grid((nq + 127) / 128, batch * heads);
  const bool prof = ctta_prof_active();
  // executed flops: QK^T and PV over the padded head dim (2 * 2*nq*nk*64 per head)
  if (prof) ctta_prof_begin(1, 0, nq, nk, 128, (long long)batch * heads, (hipStream_t)stream);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, q_ld,
                     (const bf16_t*)k, k_ld, k_rows, (const bf16_t*)vt, vt_ld, bias, (bf16_t*)out, out_ld, heads,
                     nq, nk, scale * 1.4426950408889634f);
  if (prof) ctta_prof_end((hipStream_t)stream);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
