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
#include <stdlib.h>

// v_exp_f32 directly: exp2f() wraps it in a denormal-range fix-up (compare, select, add, ldexp: 6 instructions per
// element in a VALU-bound loop); probabilities below 2^-126 flushing to zero is immaterial here
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max without the canonicalising v_max_f32 x, x that clang puts in front of fmaxf() on MFMA results (one extra VALU op per
// operand in a VALU-bound loop), and three-input max; NaNs cannot occur here (finite products of finite operands)
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
// butterfly steps across the four 16-lane rows of a wave on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap) instead
// of ds_bpermute_b32: no LDS round trip, and no s_waitcnt lgkmcnt(0) that would also wait for the K / V fragment reads
__device__ __forceinline__ float xrow16(float x, float& other) {   // rows (0,1) and (2,3) exchanged: returns own-side, other-side
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  other = __uint_as_float(r[1]);
  return __uint_as_float(r[0]);
}
__device__ __forceinline__ float xhalf32(float x, float& other) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  other = __uint_as_float(r[1]);
  return __uint_as_float(r[0]);
}
__device__ __forceinline__ float rows_max(float x) {   // max over lanes l, l^16, l^32, l^48
  float o;
  float a = xrow16(x, o);
  a = vmax2(a, o);
  float b = xhalf32(a, o);
  return vmax2(b, o);
}
__device__ __forceinline__ float rows_sum(float x) {
  float o;
  float a = xrow16(x, o);
  a += o;
  float b = xhalf32(a, o);
  return b + o;
}

static constexpr bool attn_plain_enabled() { return true; }   // the plain self-attention kernels wherever they apply
#define ATT_KT 64          // keys per tile
#define ATT_LDK 80         // K tile row stride (bf16): 160 B rows are conflict-free for ds_read_b128
#define ATT_LDV 72         // V^T tile row stride: 144 B rows are conflict-free for the paired ds_read_b64

// REL: additionally adds rel_bias[h][key - query + nq - 1] (already in the log2 domain) -- T5's relative position bias
// (one table per head over the nq + nk - 1 possible offsets); q / out batches are then q_rows rows apart.
// BMODE 3 (PLAIN): bias == nullptr and nk % 64 == 0 -- self-attention over whole key tiles, its own instantiation so that
// the two softmax formulations do not share (and inflate) one register allocation.
// BMODE 2 (FULL): adds full_bias[(b % full_nb)][h][query][key] (log2 domain) -- Swin window attention: relative-position
// bias per head plus the shifted-window mask, one table per window position (htsat.py:336-361).
template <int BMODE>
__global__ __launch_bounds__(256, 2) void attention_kernel(
    const bf16_t* __restrict__ q, int q_ld, const bf16_t* __restrict__ k, int k_ld, int k_rows,
    const bf16_t* __restrict__ vt, int vt_ld, const float* __restrict__ bias,
    bf16_t* __restrict__ out, int out_ld, int heads, int nq, int nk, float scale_log2e, float* __restrict__ lse,
    int q_rows, const float* __restrict__ rel_bias, int full_nb) {
  constexpr bool REL = BMODE == 1, FULL = BMODE == 2;
  constexpr bool PLAIN = BMODE == 3;   // no additive term at all and nk a multiple of the key tile (host-checked)
  __shared__ __attribute__((aligned(16))) bf16_t Ks[ATT_KT * ATT_LDK];   // [key][d]
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * ATT_LDV];       // [d][key]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y;
  const int b = bh / heads, h = bh - b * heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int lq = lane & 15;        // query column within a 16-block
  const int lg = lane >> 4;        // lane group 0..3

  const bf16_t* qb = q + (size_t)b * q_rows * q_ld + h * 64;
  const bf16_t* kb = k + (size_t)b * k_rows * k_ld + h * 64;
  const bf16_t* vb = vt + ((size_t)b * heads + h) * 64 * vt_ld;
  const float* bb = bias ? bias + (size_t)b * nk : nullptr;
  const float* rb = REL ? rel_bias + (size_t)h * (nq + nk - 1) + (nq - 1) : nullptr;
  const float* fb = FULL ? rel_bias + ((size_t)(b % full_nb) * heads + h) * nq * nk : nullptr;

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
  // K / V^T tiles travel global -> registers one tile ahead of the MFMAs, registers -> LDS at the top of the
  // iteration: the global latency of tile t+1 hides behind the compute of tile t.
  uint4 kreg[2], vreg[2];
  auto fetch = [&](int key0) {
    if (key0 + ATT_KT <= nk) {   // whole tile (wave-uniform): no per-lane predicates, no tail masking
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int chunk = tid + i * 256;
        const int r = chunk >> 3, cc = (chunk & 7) * 8;
        kreg[i] = *reinterpret_cast<const uint4*>(kb + (size_t)(key0 + r) * k_ld + cc);
        vreg[i] = *reinterpret_cast<const uint4*>(vb + (size_t)r * vt_ld + key0 + cc);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = tid + i * 256;
      const int r = chunk >> 3, cc = (chunk & 7) * 8;
      uint4 kv = make_uint4(0, 0, 0, 0);
      if (key0 + r < nk) kv = *reinterpret_cast<const uint4*>(kb + (size_t)(key0 + r) * k_ld + cc);
      kreg[i] = kv;
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
      vreg[i] = vv;
    }
  };
  fetch(0);
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * ATT_KT;
    __syncthreads();   // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = tid + i * 256;
      const int r = chunk >> 3, cc = (chunk & 7) * 8;
      *reinterpret_cast<uint4*>(Ks + r * ATT_LDK + cc) = kreg[i];
      *reinterpret_cast<uint4*>(Vs + r * ATT_LDV + cc) = vreg[i];
    }
    __syncthreads();
    if (t + 1 < ntiles) fetch(key0 + ATT_KT);

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
    // ---- scale, bias, key mask; lane holds keys key0 + ik*16 + lg*4 + r for query column lq.
    // Written on 4-vectors so that the scale/bias FMA, the max subtraction and the row sums compile to the packed
    // fp32 instructions (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per issue) -- the softmax VALU work, not the
    // MFMAs, bounds this kernel.
    if constexpr (PLAIN) {
      // No additive term (self-attention over a whole key tile: the 4096-token case, 97 % of this kernel's time): the row
      // maximum is taken on the raw products (scale > 0) and scale and -max fold into ONE packed FMA in front of the
      // exponential -- this loop is bound by VALU issue, not by the MFMAs (34 v_exp + ~125 other VALU ops against 32 MFMAs
      // per tile before; the separate subtraction was 32 of them).
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        float mx = -INFINITY;
#pragma unroll
        for (int ik = 0; ik < 4; ++ik)
          mx = fmaxf(mx, fmaxf(fmaxf(s[ik][jq][0], s[ik][jq][1]), fmaxf(s[ik][jq][2], s[ik][jq][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun[jq], mx * scale_log2e);
        const float alpha = fast_exp2(mrun[jq] - mnew);
        mrun[jq] = mnew;
        const f32x4_t neg_m = {-mnew, -mnew, -mnew, -mnew};
        f32x4_t ps4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ik = 0; ik < 4; ++ik) {
          const f32x4_t d = s[ik][jq] * scale_log2e + neg_m;
          f32x4_t pv;
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = fast_exp2(d[r]);
          s[ik][jq] = pv;
          ps4 += pv;
        }
        lrun[jq] = lrun[jq] * alpha + ((ps4[0] + ps4[1]) + (ps4[2] + ps4[3]));
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int jd = 0; jd < 4; ++jd) o[jd][jq] *= alpha;
        }
      }
    } else {
    f32x4_t kbias[4];
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
      const int qrel = min(q0 + jq * 16 + lq, nq - 1);
#pragma unroll
      for (int ik = 0; ik < 4; ++ik) {
        f32x4_t v = s[ik][jq] * scale_log2e + kbias[ik];
        if (REL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rb[min(key0 + ik * 16 + lg * 4 + r, nk - 1) - qrel];
        }
        if (FULL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += fb[(size_t)qrel * nk + min(key0 + ik * 16 + lg * 4 + r, nk - 1)];
        }
        s[ik][jq] = v;
        mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[jq], mx);
      const float alpha = fast_exp2(mrun[jq] - mnew);   // first tile: exp2(-inf) = 0
      mrun[jq] = mnew;
      f32x4_t ps4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4_t neg_m = {-mnew, -mnew, -mnew, -mnew};
#pragma unroll
      for (int ik = 0; ik < 4; ++ik) {
        const f32x4_t d = s[ik][jq] + neg_m;
        f32x4_t pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = fast_exp2(d[r]);
        s[ik][jq] = pv;
        ps4 += pv;
      }
      lrun[jq] = lrun[jq] * alpha + ((ps4[0] + ps4[1]) + (ps4[2] + ps4[3]));
      if (__any(alpha != 1.0f)) {   // wave-uniform: after the first tiles the running maxima rarely move
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) o[jd][jq] *= alpha;
      }
    }
    }   // !plain
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
    // training: log2-domain log-sum-exp per query, so the backward pass can rebuild P = exp2(s - lse)
    if (lse && qi < nq && lg == 0) lse[((size_t)b * heads + h) * nq + qi] = mrun[jq] + log2f(l);
    if (qi < nq) {
      bf16_t* orow = out + ((size_t)b * q_rows + qi) * out_ld + h * 64;
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

// ---- self-attention over whole key tiles, round 3 --------------------------------------------------------------------
// Same mapping as attention_kernel<3> (S^T = K Q^T, O^T += V^T P^T, a wave owns 32 queries, 64-key tiles), different
// schedule.  Round 2's loop ran QK^T -> softmax -> PV strictly in sequence inside a wave, with two workgroup barriers and
// a register -> LDS copy per tile: 31 % of the matrix-pipe peak with NEITHER the MFMA nor the VALU pipe saturated (32
// MFMAs = 512 cycles against ~520 cycles of softmax VALU per wave and tile).  Here
//   * the raw scores of tile t+1 (16 MFMAs, independent of everything the softmax touches) are issued BEFORE the softmax
//     of tile t, so the matrix pipe works while the wave's VALU does max / exp / sum (two score register sets);
//   * K / V^T tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), K two
//     tiles ahead and V^T one, into two-slot rings; the 16-byte chunks are XOR-swizzled on the SOURCE side
//     (chunk c of row r sits at c ^ ((r >> 1) & 7): conflict-free for the ds_read_b128 K fragments and the paired
//     ds_read_b64 V^T fragments, rows are 128 bytes, no padding);
//   * ONE workgroup barrier per tile: at the top of iteration t every wave has finished tile t-1, so K slot t & 1 (read by
//     the scores of tile t, computed during iteration t-1) and V slot (t+1) & 1 (read by PV of tile t-1) are free, and the
//     DMA issued an iteration ago has landed (s_waitcnt vmcnt(0) in front of the barrier).
// Round 5 (counter passes of tools/pmc_attn.sh: per wave and tile ~830 cycles of VALU issue + 512 of MFMA against 1 270
// elapsed per SIMD -- the SIMD's issue port is busy 84 % of the time, so what is left is instruction COUNT):
//   * the loop is unrolled by two with the two score register sets swapping roles: the 32 v_mov_b64 per tile that
//     copied the next tile's scores into place are gone (the scores of the tile behind the last one are computed from a
//     stale K slot and never used -- a conditional write would make the compiler carry the old register set across the
//     branch, which is the same 32 copies again);
//   * the workgroup's Q fragments live in LDS (4 waves x 4 fragments x 1 KB, lane order, read back by the wave that wrote
//     them: no barrier) instead of 16 registers for the whole loop; the second LDS-DMA instruction of a stream covers rows
//     + 32 (same swizzle key, so its source is the first one's plus a UNIFORM offset that rides in the scalar operand);
//     the V slot sits in the ds_read offset field (a literal per unrolled copy).  168 registers = 3 waves per SIMD with
//     nothing spilled inside the loop.
// Same-box A/B against the round-3 form (B 32, 5 heads, 4096 tokens): 0.887 -> 0.865 ms (775 -> 794 TFLOP/s executed).
// Measured and NOT kept: scale / -max / row sum as v_pk_fma_f32 / v_pk_add_f32 on whole accumulator registers (same time
// to the microsecond: a packed fp32 operation costs two issue slots beside MFMAs); 256 queries per 512-thread workgroup
// (one LDS-DMA instruction per wave, stream and tile instead of two: 0.97 ms, one workgroup per CU leaves nobody to run
// while its eight waves sit at the barrier); two waves per SIMD at 186 registers (same time as three).
// STAMP (tools/attn_timeline.py): every wave notes s_memtime at six points of every tile (first 64 tiles) in LDS and the
// workgroup dumps them -- plus the hardware id of each wave -- when it is done.
// ABL (timing only, WRONG results; CTTA_ATTN_ABLATE): 1 no LDS-DMA inside the loop, 2 plain FMAs instead of the
// exponentials, 4 no score MFMAs, 6 no cross-lane row maximum -- what each component costs with everything else in place
// (profiles/attn_timeline_r05.txt: 0.866 ms whole; 0.753 / 0.776 / 0.751 / 0.835 without: every part costs its 4-13 % and none
// hides behind another -- the loop is bound by the number of instructions a SIMD issues, not by one pipe).
template <bool STAMP, int ABL>
__global__ __launch_bounds__(256, 3) void attention_plain2_kernel(
    const bf16_t* __restrict__ q, int q_ld, const bf16_t* __restrict__ k, int k_ld, int k_rows,
    const bf16_t* __restrict__ vt, int vt_ld, bf16_t* __restrict__ out, int out_ld, int heads, int nq, int nk,
    float scale_log2e, float* __restrict__ lse, unsigned* __restrict__ stamps) {
  // ONE LDS object: with two, hipcc waits vmcnt(0) (all LDS-DMA landed) in front of every ds_read of the other array
  __shared__ __attribute__((aligned(16))) bf16_t smem[4 * ATT_KT * 64 + 4 * 2048 + (STAMP ? 4 * 64 * 6 * 2 : 0)];
  unsigned* const stl = reinterpret_cast<unsigned*>(smem + 4 * ATT_KT * 64 + 4 * 2048);   // [wave][tile][6]
  bf16_t (*Ks)[ATT_KT * 64] = reinterpret_cast<bf16_t (*)[ATT_KT * 64]>(smem);                    // [slot][key][d], swizzled chunks
  bf16_t (*Vs)[ATT_KT * 64] = reinterpret_cast<bf16_t (*)[ATT_KT * 64]>(smem + 2 * ATT_KT * 64);  // [slot][d][key], swizzled chunks
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y;
  const int b = bh / heads, h = bh - b * heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int lq = lane & 15, lg = lane >> 4;
  const bf16_t* qb = q + (size_t)b * nq * q_ld + h * 64;
  const bf16_t* kb = k + (size_t)b * k_rows * k_ld + h * 64;
  const bf16_t* vb = vt + ((size_t)b * heads + h) * 64 * vt_ld;

  uint4* const qs = reinterpret_cast<uint4*>(smem + 4 * ATT_KT * 64) + wave * 256 + lane;   // [fragment jq*2 + ds][lane]
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    int qi = q0 + jq * 16 + lq;
    if (qi >= nq) qi = nq - 1;   // clamp (never stored)
#pragma unroll
    for (int ds = 0; ds < 2; ++ds)
      qs[(jq * 2 + ds) * 64] = *reinterpret_cast<const uint4*>(qb + (size_t)qi * q_ld + ds * 32 + lg * 8);
  }
  // LDS-DMA: a wave instruction covers 8 rows x 128 bytes; lane -> (row = lane / 8, chunk position = lane % 8), which
  // receives the logical chunk (lane % 8) ^ ((row >> 1) & 7) of that row.  A tile is 8 instructions: 2 per wave, rows
  // wave*8 + lane/8 and + 32.  Through buffer descriptors (as conv_gemm does): the per-lane byte offset is loop invariant,
  // the tile advance rides in the scalar offset -- and, unlike __builtin_amdgcn_global_load_lds, hipcc does not put
  // s_waitcnt vmcnt(0) in front of later ds_reads of the OTHER ring slot.
  const int drow = wave * 8 + (lane >> 3);
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(
      (void*)kb, 0, (unsigned)(((long long)(nk - 1) * k_ld + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(
      (void*)vb, 0, (unsigned)(((long long)63 * vt_ld + nk) * 2), 0x00020000);
  const int dchunk = (lane & 7) ^ ((drow >> 1) & 7);
  const int koff = (drow * k_ld + dchunk * 8) * 2;
  const int voff = (drow * vt_ld + dchunk * 8) * 2;
  auto issue_k = [&](int key0, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, (__attribute__((address_space(3))) void*)(&Ks[slot][(wave * 8 + i * 32) * 64]), 16,
                                               koff, (key0 + i * 32) * k_ld * 2, 0, 0);
  };
  auto issue_v = [&](int key0, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (__attribute__((address_space(3))) void*)(&Vs[slot][(wave * 8 + i * 32) * 64]), 16,
                                               voff, key0 * 2 + i * 32 * vt_ld * 2, 0, 0);
  };
  const int fsw = (lq >> 1) & 7;                      // swizzle of the fragment rows this lane reads (row % 16 == lq)
  auto scores = [&](int slot, f32x4_t (&s)[4][2]) {
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      const bf16x8_t qa = __builtin_bit_cast(bf16x8_t, qs[ds * 64]), qc = __builtin_bit_cast(bf16x8_t, qs[(2 + ds) * 64]);
#pragma unroll
      for (int ik = 0; ik < 4; ++ik) {
        const bf16x8_t kf = __builtin_bit_cast(
            bf16x8_t, *reinterpret_cast<const uint4*>(&Ks[slot][(ik * 16 + lq) * 64 + (((ds * 4 + lg) ^ fsw) * 8)]));
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ABL == 4) {
          s[ik][0] = ds ? s[ik][0] + __builtin_bit_cast(f32x4_t, kf) : __builtin_bit_cast(f32x4_t, qa);
          s[ik][1] = ds ? s[ik][1] + __builtin_bit_cast(f32x4_t, kf) : __builtin_bit_cast(f32x4_t, qc);
        } else {
          s[ik][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qa, ds ? s[ik][0] : z, 0, 0, 0);
          s[ik][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qc, ds ? s[ik][1] : z, 0, 0, 0);
        }
      }
    }
  };

  // V^T fragment byte offsets inside a slot, loop invariant: row d = jd*16 + lq (+ jd * 2048 bytes), keys kk*32 + lg*4 .. +3
  // = chunk kk*4 + lg/2, half lg & 1; the +16 keys of the second half-fragment = chunk + 2.
  unsigned vbase[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      vbase[kk][hh] = (unsigned)(lq * 128 + (((kk * 4 + hh * 2 + (lg >> 1)) ^ fsw) * 16) + (lg & 1) * 8);
    }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  f32x4_t o[4][2];
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) o[jd][jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {-INFINITY, -INFINITY};
  float lrun[2] = {0.f, 0.f};

  const int ntiles = nk / ATT_KT;
  issue_k(0, 0);
  issue_v(0, 0);
  if (ntiles > 1) issue_k(ATT_KT, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4_t sa[4][2], sb[4][2];
  scores(0, sa);
  // one tile: `sc` holds the raw scores of tile t, `sn` receives those of tile t + 1; `par` = t & 1 as a literal
  auto tile_step = [&](int t, const int par, f32x4_t (&sc)[4][2], f32x4_t (&sn)[4][2]) __attribute__((always_inline)) {
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
    if constexpr (STAMP) ts0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of K(t+1) / V(t) has landed ...
    __builtin_amdgcn_s_barrier();                        // ... and so has everybody else's; tile t-1 is finished everywhere
    if constexpr (STAMP) ts1 = __builtin_amdgcn_s_memtime();
    if (ABL != 1 && t + 2 < ntiles) issue_k((t + 2) * ATT_KT, par);
    if (ABL != 1 && t + 1 < ntiles) issue_v((t + 1) * ATT_KT, par ^ 1);
    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); ts5 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    scores(par ^ 1, sn);
    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); ts2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    // ---- softmax of tile t on the raw products (scale > 0): max, then scale and -max folded into one FMA before exp2
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
      float mx = vmax2(sc[0][jq][0], sc[0][jq][1]);
      mx = vmax3(mx, sc[0][jq][2], sc[0][jq][3]);
#pragma unroll
      for (int ik = 1; ik < 4; ++ik) {
        mx = vmax3(mx, sc[ik][jq][0], sc[ik][jq][1]);
        mx = vmax3(mx, sc[ik][jq][2], sc[ik][jq][3]);
      }
      if constexpr (ABL != 6) mx = rows_max(mx); else mx = sc[0][jq][0];
      const float mnew = vmax2(mrun[jq], mx * scale_log2e);
      const float alpha = fast_exp2(mrun[jq] - mnew);
      mrun[jq] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int ik = 0; ik < 4; ++ik)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = ABL == 2 ? fmaf(sc[ik][jq][r], scale_log2e, -mnew) : fast_exp2(fmaf(sc[ik][jq][r], scale_log2e, -mnew));
          sc[ik][jq][r] = pv;
          ps += pv;
        }
      lrun[jq] = lrun[jq] * alpha + ps;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) o[jd][jq] *= alpha;
      }
    }
    if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); ts3 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    // ---- O^T += V^T P^T.  The V^T fragments are read with inline-asm ds_read_b64: hipcc merges the compiler-visible
    // form into ds_read2st64_b64 and then waits vmcnt(0) -- i.e. for the LDS-DMA of the NEXT tiles, issued a few hundred
    // cycles earlier -- in front of it (the K fragments' ds_read_b128 get no such wait).  LDS operations return in
    // order, so the compiler's own counted lgkmcnt waits stay correct (at worst they wait for more); the values read
    // here are consumed behind an explicit lgkmcnt(0).
    // (Measured alternatives, tools/attn_bench.py at B=32, 5 heads, 4096 tokens: all 16 reads up front + PV per query
    // block behind its half of the softmax: 192 VGPRs = 2 waves per SIMD, 727 TFLOP/s; this form 790; round 2's kernel 717.)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint2 lo[4], hi[4];
      const unsigned a0 = lds_base + vbase[kk][0], a1 = lds_base + vbase[kk][1];
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {   // "i": a constant once tile_step is inlined with a literal `par`
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lo[jd]) : "v"(a0), "i"(2 * ATT_KT * 64 * 2 + par * (ATT_KT * 64 * 2) + jd * 2048));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(hi[jd]) : "v"(a1), "i"(2 * ATT_KT * 64 * 2 + par * (ATT_KT * 64 * 2) + jd * 2048));
      }
      bf16x8_t pf[2];
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        uint4 pk;
        pk.x = pack2bf(sc[2 * kk][jq][0], sc[2 * kk][jq][1]);
        pk.y = pack2bf(sc[2 * kk][jq][2], sc[2 * kk][jq][3]);
        pk.z = pack2bf(sc[2 * kk + 1][jq][0], sc[2 * kk + 1][jq][1]);
        pk.w = pack2bf(sc[2 * kk + 1][jq][2], sc[2 * kk + 1][jq][3]);
        pf[jq] = __builtin_bit_cast(bf16x8_t, pk);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, make_uint4(lo[jd].x, lo[jd].y, hi[jd].x, hi[jd].y));
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) o[jd][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[jq], o[jd][jq], 0, 0, 0);
      }
    }
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      ts4 = __builtin_amdgcn_s_memtime();
      if (t < 64 && lane == 0) {
        unsigned* d = stl + (wave * 64 + t) * 6;
        d[0] = (unsigned)ts0; d[1] = (unsigned)ts1; d[2] = (unsigned)ts5; d[3] = (unsigned)ts2; d[4] = (unsigned)ts3; d[5] = (unsigned)ts4;
      }
    }
  };
  int t = 0;
  for (; t + 1 < ntiles; t += 2) {
    tile_step(t, 0, sa, sb);
    tile_step(t + 1, 1, sb, sa);
  }
  if (t < ntiles) tile_step(t, 0, sa, sb);
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    const float l = rows_sum(lrun[jq]);
    const float inv = 1.0f / l;
    const int qi = q0 + jq * 16 + lq;
    if (lse && qi < nq && lg == 0) lse[((size_t)b * heads + h) * nq + qi] = mrun[jq] + log2f(l);
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
  if constexpr (STAMP) {
    __syncthreads();
    unsigned* dst = stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (4 * 64 * 6 + 8);
    for (int i = tid; i < 4 * 64 * 6; i += 256) dst[i] = stl[i];
    if (lane == 0) {
      dst[4 * 64 * 6 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_ID
      dst[4 * 64 * 6 + 4 + wave] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // XCC_ID
    }
  }
}
// debugging aid, like ctta_conv_debug_stamps: buf = (4 * 64 * 6 + 8) unsigned per workgroup of the next self-attention launches
static thread_local unsigned* t_attn_stamps = nullptr;
extern "C" void ctta_attention_debug_stamps(void* buf) { t_attn_stamps = (unsigned*)buf; }
static constexpr int attn_v2_mode() { return 1; }   // round 3's self-attention kernel (round 2's stays for the shapes it does not take)

extern "C" ctta_status ctta_attention(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                                      int vt_ld, const float* bias, void* out, int out_ld, int batch,
                                      int heads, int nq, int nk, float scale, void* stream) {
  return ctta_attention_lse(q, q_ld, k, k_ld, k_rows, vt, vt_ld, bias, out, out_ld, batch, heads, nq, nk, scale, nullptr,
                            stream);
}

extern "C" ctta_status ctta_attention_lse(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                                          int vt_ld, const float* bias, void* out, int out_ld, int batch, int heads,
                                          int nq, int nk, float scale, float* lse, void* stream) {
  CTTA_REQUIRE(q && k && vt && out, "attention: null pointer");
  CTTA_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vt_ld % 8 == 0 && out_ld % 4 == 0,
               "attention: row strides must be multiples of 8");
  CTTA_REQUIRE(nq > 0 && nk > 0 && k_rows >= nk && vt_ld >= ((nk + 7) / 8) * 8, "attention: bad lengths nq=%d nk=%d vt_ld=%d", nq, nk, vt_ld);
  dim3 grid((nq + 127) / 128, batch * heads);
  const bool prof = ctta_prof_active();
  // executed flops: QK^T and PV over the padded head dim (2 * 2*nq*nk*64 per head)
  if (prof) ctta_prof_begin(1, 0, nq, nk, 128, (long long)batch * heads, (hipStream_t)stream);
  if (!bias && nk % ATT_KT == 0 && attn_plain_enabled() && attn_v2_mode() && nk >= 2 * ATT_KT)
  {
#define CTTA_ATTN_LAUNCH(ST, AB, SP)                                                                                          \
  hipLaunchKernelGGL((attention_plain2_kernel<ST, AB>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, q_ld,     \
                     (const bf16_t*)k, k_ld, k_rows, (const bf16_t*)vt, vt_ld, (bf16_t*)out, out_ld, heads, nq, nk,          \
                     scale * 1.4426950408889634f, lse, SP)
    // (round 5's timing-only ablations -- AB = 1 / 2 / 4 / 6 of the kernel template -- are no longer instantiated)
    if (t_attn_stamps) CTTA_ATTN_LAUNCH(true, 0, t_attn_stamps);
    else CTTA_ATTN_LAUNCH(false, 0, nullptr);
#undef CTTA_ATTN_LAUNCH
  }
  else if (!bias && nk % ATT_KT == 0 && attn_plain_enabled())   // self-attention over whole key tiles: no additive term
    hipLaunchKernelGGL(attention_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, q_ld,
                       (const bf16_t*)k, k_ld, k_rows, (const bf16_t*)vt, vt_ld, bias, (bf16_t*)out, out_ld, heads,
                       nq, nk, scale * 1.4426950408889634f, lse, nq, (const float*)nullptr, 1);
  else
    hipLaunchKernelGGL(attention_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, q_ld,
                       (const bf16_t*)k, k_ld, k_rows, (const bf16_t*)vt, vt_ld, bias, (bf16_t*)out, out_ld, heads,
                       nq, nk, scale * 1.4426950408889634f, lse, nq, (const float*)nullptr, 1);
  if (prof) ctta_prof_end((hipStream_t)stream);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_attention_rel(const void* q, int q_ld, int q_rows, const void* k, int k_ld, int k_rows,
                                          const void* vt, int vt_ld, const float* key_bias, const float* rel_bias_log2,
                                          void* out, int out_ld, int batch, int heads, int nq, int nk, float scale,
                                          void* stream) {
  CTTA_REQUIRE(q && k && vt && out && rel_bias_log2, "attention_rel: null pointer");
  CTTA_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vt_ld % 8 == 0 && out_ld % 4 == 0,
               "attention_rel: row strides must be multiples of 8");
  CTTA_REQUIRE(nq > 0 && nk > 0 && q_rows >= nq && k_rows >= nk && vt_ld >= ((nk + 7) / 8) * 8,
               "attention_rel: bad lengths nq=%d nk=%d vt_ld=%d", nq, nk, vt_ld);
  dim3 grid((nq + 127) / 128, batch * heads);
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(1, 0, nq, nk, 128, (long long)batch * heads, (hipStream_t)stream);
  hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, q_ld,
                     (const bf16_t*)k, k_ld, k_rows, (const bf16_t*)vt, vt_ld, key_bias, (bf16_t*)out, out_ld, heads,
                     nq, nk, scale * 1.4426950408889634f, (float*)nullptr, q_rows, rel_bias_log2, 1);
  if (prof) ctta_prof_end((hipStream_t)stream);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_attention_fullbias(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                                               int vt_ld, const float* full_bias_log2, int n_bias_batches, void* out,
                                               int out_ld, int batch, int heads, int nq, int nk, float scale, float* lse,
                                               void* stream) {
  CTTA_REQUIRE(q && k && vt && out && full_bias_log2 && n_bias_batches >= 1, "attention_fullbias: null pointer");
  CTTA_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vt_ld % 8 == 0 && out_ld % 4 == 0,
               "attention_fullbias: row strides must be multiples of 8");
  CTTA_REQUIRE(nq > 0 && nk > 0 && k_rows >= nk && vt_ld >= ((nk + 7) / 8) * 8,
               "attention_fullbias: bad lengths nq=%d nk=%d vt_ld=%d", nq, nk, vt_ld);
  dim3 grid((nq + 127) / 128, batch * heads);
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(1, 0, nq, nk, 128, (long long)batch * heads, (hipStream_t)stream);
  hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, q_ld,
                     (const bf16_t*)k, k_ld, k_rows, (const bf16_t*)vt, vt_ld, (const float*)nullptr, (bf16_t*)out, out_ld,
                     heads, nq, nk, scale * 1.4426950408889634f, lse, nq, full_bias_log2, n_bias_batches);
  if (prof) ctta_prof_end((hipStream_t)stream);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

// ====================================================================================== backward
// Flash-style attention backward (torch autograd of F.scaled_dot_product_attention,
// attention_processor.py:1127-1129).  Probabilities are rebuilt from the forward's log-sum-exp:
//     P = exp2(s * scale*log2e + bias*log2e - lse),   dS = P * (dP - D) * scale,   D[q] = sum_d dO[q][d] O[q][d]
// Two kernels, no atomics:
//   * dq kernel : a workgroup owns 128 queries (32 per wave) and walks the keys -- the forward's structure:
//        S^T = K Q^T,  dP^T = V dO^T  (queries = MFMA columns, Q / dO rows held in registers as B operands),
//        dQ^T += K^T dS^T  (A = K^T tile from LDS, B = dS^T re-used from the accumulators).
//   * dkv kernel: a workgroup owns 128 keys (32 per wave) and walks the queries:
//        S = Q K^T,  dP = dO V^T  (keys = MFMA columns, K / V rows in registers),
//        dV^T += dO^T P,  dK^T += Q^T dS   (A = dO^T / Q^T tiles from LDS, B = P / dS from the accumulators).
// Operands that are needed with the contraction index contiguous (K^T, Q^T, dO^T, and V in natural
// layout) are produced once per call by ctta_transpose_bf16 over all heads.
struct AttnBwdParams {
  const bf16_t *q, *k, *vn, *kt, *qt, *dot, *dO;
  int q_ld, k_ld, k_rows, vn_ld, vn_rows, kt_ld, qt_ld, do_ld;
  const float *bias, *lse, *dsum;
  bf16_t *dq, *dk, *dv;
  int dq_ld, dk_ld, dv_ld;
  int heads, nq, nk;
  float scale, scale_log2e;
  // dkv kernel with few keys (cross-attention): the query range is split over blockIdx.z and every split
  // writes fp32 partial sums part[z][dk|dv][B][k_rows][hp], folded by attn_bwd_fold_kernel
  float* part;
  int q_tiles_per_split, batch, hp;
  const float* full_bias;   // FULL kernels: [full_nb][heads][nq][nk], log2 domain (window attention)
  int full_nb;
  // TR kernels (round 5): Q, K, dO are read where they lie and V as the forward keeps it, transposed [B][heads*64][vt_ld];
  // kt / qt / dot / vn are unused
  const bf16_t* vt;
  int vt_ld;
  // D = rowsum(dO * O) computed by the dq kernel itself (it holds every query's dO row in registers) and written to dsum
  // for the dkv kernel behind it: no separate row-dot launch (76 launches per distillation step)
  const bf16_t* o;
  int o_ld;
  float* dsum_out;
};

typedef short att_s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) att_s16x4_t* att_lds_s16x4_ptr;
// The transposing LDS read (ds_read_b64_tr_b16): lane t of a 16-lane group NAMES the address of row k0 + t / 4, columns
// 4 (t % 4) .. + 3 of a row-major bf16 tile and RECEIVES column t % 16, rows k0 .. k0 + 3.  Two of them (rows k0 .. + 3 and
// k0 + 16 .. + 19) make one MFMA operand fragment whose k slots (g = lane / 16, e) are rows 4 g + e (e < 4) and 16 + 4 g + e - 4
// of a 32-row chunk: the order in which the S / dS accumulators of the kernels below already hold their 32 positions, so an
// operand that is needed "the other way round" is read out of the tile the kernel stages anyway -- no transposed copy of
// Q, K or dO in HBM (wgrad_tn_kernel uses the same read on both of its operands).  `row0` = first of the 32 rows, `col0` =
// first of the fragment's 16 columns, `ld` = row stride in elements (160-byte rows: = 32 mod 64, conflict-free).
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* tile, int ld, int row0, int col0, int lq, int lg) {
  const bf16_t* a = tile + (row0 + lg * 4 + (lq >> 2)) * ld + col0 + (lq & 3) * 4;
  const att_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4_ptr)a);
  const att_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4_ptr)(a + 16 * ld));
  const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
  return __builtin_bit_cast(bf16x8_t, make_uint4(u0.x, u0.y, u1.x, u1.y));
}
// 8 values of a row in the k-slot order of tr_frag: columns c0 + 4 g .. + 3 and c0 + 16 + 4 g .. + 3 (g = lane / 16)
__device__ __forceinline__ bf16x8_t load_perm8(const bf16_t* row, int c0, int lg) {
  const uint2 lo = *reinterpret_cast<const uint2*>(row + c0 + lg * 4);
  const uint2 hi = *reinterpret_cast<const uint2*>(row + c0 + 16 + lg * 4);
  return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// a [64 d][64 keys] tile of V^T (rows vt_ld apart) into registers; keys >= nk come back as zeros (vt_ld >= nk, both
// multiples of 8 apart from nk itself: a vector that straddles nk is masked)
__device__ __forceinline__ void fetch_vt_rows(uint4 (&reg)[2], const bf16_t* src, size_t src_ld, int key0, int nk, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int chunk = tid + i * 256;
    const int r = chunk >> 3, cc = (chunk & 7) * 8;
    const int nv = nk - (key0 + cc);          // valid keys in this vector
    uint4 v = make_uint4(0, 0, 0, 0);
    if (nv > 0) {
      v = *reinterpret_cast<const uint4*>(src + (size_t)r * src_ld + key0 + cc);
      if (nv < 8) {
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (2 * j >= nv) ? 0u : ((2 * j + 1 >= nv) ? (w[j] & 0xffffu) : w[j]);
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    reg[i] = v;
  }
}


// stages a [64 rows][64 cols] bf16 tile (row stride ld in LDS); rows >= rows_valid are zero
__device__ __forceinline__ void stage_rows(bf16_t* lds, int ld, const bf16_t* src, size_t src_ld, int rows_valid, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int chunk = tid + i * 256;
    const int r = chunk >> 3, cc = (chunk & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows_valid) v = *reinterpret_cast<const uint4*>(src + (size_t)r * src_ld + cc);
    *reinterpret_cast<uint4*>(lds + r * ld + cc) = v;
  }
}

// stage_rows in two halves, so a tile's global loads can be issued one tile ahead of the MFMAs that consume it
__device__ __forceinline__ void fetch_rows(uint4 (&reg)[2], const bf16_t* src, size_t src_ld, int rows_valid, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int chunk = tid + i * 256;
    const int r = chunk >> 3, cc = (chunk & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows_valid) v = *reinterpret_cast<const uint4*>(src + (size_t)r * src_ld + cc);
    reg[i] = v;
  }
}
__device__ __forceinline__ void put_rows(bf16_t* lds, int ld, const uint4 (&reg)[2], int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int chunk = tid + i * 256;
    const int r = chunk >> 3, cc = (chunk & 7) * 8;
    *reinterpret_cast<uint4*>(lds + r * ld + cc) = reg[i];
  }
}

// PLAIN: no additive term (bias == nullptr, no full table) and nk % 64 == 0 -- the softmax rebuild on 4-vectors: P =
// exp2(s * scale - lse) and dS = P * (dP * scale - D * scale) are one packed FMA + exp, one packed FMA and one packed
// multiply per pair of elements instead of seven scalar operations and a key-range test per element.
template <bool FULL, bool PLAIN, bool TR>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * ATT_LDK];    // [key][d]
  __shared__ __attribute__((aligned(16))) bf16_t Vn[64 * ATT_LDK];    // [key][d]   (TR: the V^T tile, [d][key])
  __shared__ __attribute__((aligned(16))) bf16_t KTs[TR ? 8 : 64 * ATT_LDV];   // [d][key]   (TR: K^T is read out of Ks)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int lq = lane & 15, lg = lane >> 4;
  const bf16_t* qb = p.q + (size_t)b * p.nq * p.q_ld + h * 64;
  const bf16_t* dob = p.dO + (size_t)b * p.nq * p.do_ld + h * 64;
  const bf16_t* kb = p.k + (size_t)b * p.k_rows * p.k_ld + h * 64;
  const bf16_t* vb = TR ? p.vt + ((size_t)b * p.heads + h) * 64 * p.vt_ld : p.vn + (size_t)b * p.vn_rows * p.vn_ld + h * 64;
  const bf16_t* ktb = TR ? nullptr : p.kt + ((size_t)b * p.heads + h) * 64 * p.kt_ld;
  const float* bb = p.bias ? p.bias + (size_t)b * p.nk : nullptr;
  const float* fb = FULL ? p.full_bias + ((size_t)(b % p.full_nb) * p.heads + h) * p.nq * p.nk : nullptr;

  bf16x8_t qf[2][2], dof[2][2];
  float lse_q[2], d_q[2];
  int q_row[2];
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    int qi = q0 + jq * 16 + lq;
    if (qi >= p.nq) qi = p.nq - 1;   // clamp (never stored)
    q_row[jq] = qi;
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      qf[jq][ds] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qb + (size_t)qi * p.q_ld + ds * 32 + lg * 8));
      // TR: dP = dO V^T contracts over d in the k-slot order of the transposing read that delivers V out of the V^T tile
      if constexpr (TR) dof[jq][ds] = load_perm8(dob + (size_t)qi * p.do_ld, ds * 32, lg);
      else dof[jq][ds] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(dob + (size_t)qi * p.do_ld + ds * 32 + lg * 8));
    }
    lse_q[jq] = p.lse[((size_t)b * p.heads + h) * p.nq + qi];
    if (p.dsum_out) {     // this lane's 16 of the query's 64 products, then the four lane rows (fixed butterfly)
      const bf16_t* orow = p.o + ((size_t)b * p.nq + qi) * p.o_ld + h * 64;
      float part = 0.f;
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        bf16x8_t of;
        if constexpr (TR) of = load_perm8(orow, ds * 32, lg);
        else of = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(orow + ds * 32 + lg * 8));
        float a[8], g[8];
        unpack8(__builtin_bit_cast(uint4, dof[jq][ds]), a);
        unpack8(__builtin_bit_cast(uint4, of), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) part += a[e] * g[e];
      }
      d_q[jq] = rows_sum(part);
      if (lg == 0 && q0 + jq * 16 + lq < p.nq) p.dsum_out[((size_t)b * p.heads + h) * p.nq + qi] = d_q[jq];
    } else {
      d_q[jq] = p.dsum[((size_t)b * p.heads + h) * p.nq + qi];
    }
  }
  f32x4_t o[4][2];   // dQ^T accumulators [d block][q block]
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) o[jd][jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int ntiles = (p.nk + 63) / 64;
  uint4 rk[2], rv[2], rkt[2];
  auto fetch = [&](int key0) {
    fetch_rows(rk, kb + (size_t)key0 * p.k_ld, p.k_ld, p.nk - key0, tid);
    if constexpr (TR) {
      fetch_vt_rows(rv, vb, p.vt_ld, key0, p.nk, tid);
    } else {
      fetch_rows(rv, vb + (size_t)key0 * p.vn_ld, p.vn_ld, p.nk - key0, tid);
      fetch_rows(rkt, ktb + key0, p.kt_ld, 64, tid);   // zero beyond nk by construction of K^T
    }
  };
  fetch(0);
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * 64;
    __syncthreads();
    put_rows(Ks, ATT_LDK, rk, tid);
    put_rows(Vn, ATT_LDK, rv, tid);
    if constexpr (!TR) put_rows(KTs, ATT_LDV, rkt, tid);
    __syncthreads();
    if (t + 1 < ntiles) fetch(key0 + 64);
    f32x4_t s[4][2], dp[4][2];
#pragma unroll
    for (int ik = 0; ik < 4; ++ik) {
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) { s[ik][jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[ik][jq] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Ks + (ik * 16 + lq) * ATT_LDK + ds * 32 + lg * 8));
        bf16x8_t vf;      // V rows of keys ik * 16 + lq: natural tile, or (TR) out of the [d][key] tile
        if constexpr (TR) vf = tr_frag(Vn, ATT_LDK, ds * 32, ik * 16, lq, lg);
        else vf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Vn + (ik * 16 + lq) * ATT_LDK + ds * 32 + lg * 8));
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
          s[ik][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[jq][ds], s[ik][jq], 0, 0, 0);
          dp[ik][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[jq][ds], dp[ik][jq], 0, 0, 0);
        }
      }
    }
    // dS^T = P^T * (dP^T - D) * scale, lane holds keys key0 + ik*16 + lg*4 + r for query column lq
    if constexpr (PLAIN) {
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        const float nl = -lse_q[jq], nd = -d_q[jq] * p.scale;
        const f32x4_t nl4 = {nl, nl, nl, nl}, nd4 = {nd, nd, nd, nd};
#pragma unroll
        for (int ik = 0; ik < 4; ++ik) {
          const f32x4_t e = s[ik][jq] * p.scale_log2e + nl4;
          f32x4_t pv;
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = fast_exp2(e[r]);
          s[ik][jq] = pv * (dp[ik][jq] * p.scale + nd4);
        }
      }
    } else
#pragma unroll
    for (int ik = 0; ik < 4; ++ik)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + ik * 16 + lg * 4 + r;
        float add = 0.f;
        if (key >= p.nk) add = -INFINITY;
        else if (bb) add = bb[key] * 1.4426950408889634f;
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
          float addq = add;
          if (FULL) addq += fb[(size_t)q_row[jq] * p.nk + min(key, p.nk - 1)];
          const float pv = fast_exp2(s[ik][jq][r] * p.scale_log2e + addq - lse_q[jq]);
          s[ik][jq][r] = pv * (dp[ik][jq][r] - d_q[jq]) * p.scale;
        }
      }
    // dQ^T += K^T dS^T
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t sf[2];
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        uint4 pk;
        pk.x = pack2bf(s[2 * kk][jq][0], s[2 * kk][jq][1]);
        pk.y = pack2bf(s[2 * kk][jq][2], s[2 * kk][jq][3]);
        pk.z = pack2bf(s[2 * kk + 1][jq][0], s[2 * kk + 1][jq][1]);
        pk.w = pack2bf(s[2 * kk + 1][jq][2], s[2 * kk + 1][jq][3]);
        sf[jq] = __builtin_bit_cast(bf16x8_t, pk);
      }
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        bf16x8_t af;      // K^T rows d = jd * 16 + lq over the 32 keys of chunk kk
        if constexpr (TR) {
          af = tr_frag(Ks, ATT_LDK, kk * 32, jd * 16, lq, lg);
        } else {
          const bf16_t* kr = KTs + (jd * 16 + lq) * ATT_LDV + kk * 32 + lg * 4;
          const uint2 lo = *reinterpret_cast<const uint2*>(kr);
          const uint2 hi = *reinterpret_cast<const uint2*>(kr + 16);
          af = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) o[jd][jq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, sf[jq], o[jd][jq], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int jq = 0; jq < 2; ++jq) {
    const int qi = q0 + jq * 16 + lq;
    if (qi < p.nq) {
      bf16_t* orow = p.dq + ((size_t)b * p.nq + qi) * p.dq_ld + h * 64;
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        uint2 pk;
        pk.x = pack2bf(o[jd][jq][0], o[jd][jq][1]);
        pk.y = pack2bf(o[jd][jq][2], o[jd][jq][3]);
        *reinterpret_cast<uint2*>(orow + jd * 16 + lg * 4) = pk;
      }
    }
  }
}

template <bool FULL, bool PLAIN, bool TR>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[64 * ATT_LDK];     // [query][d]
  __shared__ __attribute__((aligned(16))) bf16_t dOs[64 * ATT_LDK];    // [query][d]
  __shared__ __attribute__((aligned(16))) bf16_t QTs[TR ? 8 : 64 * ATT_LDV];    // [d][query]   (TR: read out of Qs / dOs)
  __shared__ __attribute__((aligned(16))) bf16_t dOTs[TR ? 8 : 64 * ATT_LDV];   // [d][query]
  __shared__ __attribute__((aligned(16))) float lse_s[64];
  __shared__ __attribute__((aligned(16))) float d_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const int lq = lane & 15, lg = lane >> 4;
  const bf16_t* qb = p.q + (size_t)b * p.nq * p.q_ld + h * 64;
  const bf16_t* dob = p.dO + (size_t)b * p.nq * p.do_ld + h * 64;
  const bf16_t* kb = p.k + (size_t)b * p.k_rows * p.k_ld + h * 64;
  const bf16_t* vb = TR ? p.vt + ((size_t)b * p.heads + h) * 64 * p.vt_ld : p.vn + (size_t)b * p.vn_rows * p.vn_ld + h * 64;
  const bf16_t* qtb = TR ? nullptr : p.qt + ((size_t)b * p.heads + h) * 64 * p.qt_ld;
  const bf16_t* dotb = TR ? nullptr : p.dot + ((size_t)b * p.heads + h) * 64 * p.qt_ld;
  const float* lseb = p.lse + ((size_t)b * p.heads + h) * p.nq;
  const float* dsb = p.dsum + ((size_t)b * p.heads + h) * p.nq;
  const float* fb = FULL ? p.full_bias + ((size_t)(b % p.full_nb) * p.heads + h) * p.nq * p.nk : nullptr;

  bf16x8_t kf[2][2], vf[2][2];   // B operands: lane = key column, 8 consecutive d
  float kb2[2];
#pragma unroll
  for (int jk = 0; jk < 2; ++jk) {
    const int key = k0 + jk * 16 + lq;
    const int kc = key < p.nk ? key : p.nk - 1;
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      kf[jk][ds] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(kb + (size_t)kc * p.k_ld + ds * 32 + lg * 8));
      if constexpr (TR) {   // this key's 8 consecutive d out of V^T [d][key]: 8 strided 2-byte loads, once per workgroup
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = vb[(size_t)(ds * 32 + lg * 8 + i) * p.vt_ld + kc];
        vf[jk][ds] = __builtin_bit_cast(bf16x8_t, make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16),
                                                               e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16)));
      } else {
        vf[jk][ds] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(vb + (size_t)kc * p.vn_ld + ds * 32 + lg * 8));
      }
    }
    kb2[jk] = key < p.nk ? (p.bias ? p.bias[(size_t)b * p.nk + key] * 1.4426950408889634f : 0.f) : -INFINITY;
  }
  f32x4_t dv[4][2], dk[4][2];   // dV^T / dK^T accumulators [d block][key block]
#pragma unroll
  for (int jd = 0; jd < 4; ++jd)
#pragma unroll
    for (int jk = 0; jk < 2; ++jk) { dv[jd][jk] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dk[jd][jk] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

  const int ntiles = (p.nq + 63) / 64;
  const int t_begin = blockIdx.z * p.q_tiles_per_split;
  const int t_end = min(ntiles, t_begin + p.q_tiles_per_split);
  uint4 rq[2], rdo[2], rqt[2];
  float r_lse = INFINITY, r_d = 0.f;
  auto fetch = [&](int q0) {   // dO^T stays a synchronous stage: a fourth register tile spills the accumulators
    fetch_rows(rq, qb + (size_t)q0 * p.q_ld, p.q_ld, p.nq - q0, tid);
    fetch_rows(rdo, dob + (size_t)q0 * p.do_ld, p.do_ld, p.nq - q0, tid);
    if constexpr (!TR) fetch_rows(rqt, qtb + q0, p.qt_ld, 64, tid);     // zero beyond nq by construction
    if (tid < 64) {
      const bool ok = q0 + tid < p.nq;
      r_lse = ok ? lseb[q0 + tid] : INFINITY;   // exp2(-inf) = 0 for padded queries
      r_d = ok ? dsb[q0 + tid] : 0.f;
    }
  };
  if (t_begin < t_end) fetch(t_begin * 64);
  for (int t = t_begin; t < t_end; ++t) {
    const int q0 = t * 64;
    __syncthreads();
    put_rows(Qs, ATT_LDK, rq, tid);
    put_rows(dOs, ATT_LDK, rdo, tid);
    if constexpr (!TR) {
      put_rows(QTs, ATT_LDV, rqt, tid);
      stage_rows(dOTs, ATT_LDV, dotb + q0, p.qt_ld, 64, tid);
    }
    if (tid < 64) {   // PLAIN keeps them negated (and D pre-scaled): operands of the packed FMAs below
      lse_s[tid] = PLAIN ? -r_lse : r_lse;
      d_s[tid] = PLAIN ? -r_d * p.scale : r_d;
    }
    __syncthreads();
    if (t + 1 < t_end) fetch(q0 + 64);
    f32x4_t s[4][2], dp[4][2];
#pragma unroll
    for (int iq = 0; iq < 4; ++iq) {
#pragma unroll
      for (int jk = 0; jk < 2; ++jk) { s[iq][jk] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[iq][jk] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const bf16x8_t qa = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Qs + (iq * 16 + lq) * ATT_LDK + ds * 32 + lg * 8));
        const bf16x8_t da = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(dOs + (iq * 16 + lq) * ATT_LDK + ds * 32 + lg * 8));
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
          s[iq][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[jk][ds], s[iq][jk], 0, 0, 0);
          dp[iq][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[jk][ds], dp[iq][jk], 0, 0, 0);
        }
      }
    }
    // lane holds queries q0 + iq*16 + lg*4 + r for key column lq:  s -> P,  dp -> dS
#pragma unroll
    for (int iq = 0; iq < 4; ++iq) {
      const float4 l4 = *reinterpret_cast<const float4*>(lse_s + iq * 16 + lg * 4);
      const float4 d4 = *reinterpret_cast<const float4*>(d_s + iq * 16 + lg * 4);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w};
      const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
      if constexpr (PLAIN) {
        const f32x4_t nl4 = {l4.x, l4.y, l4.z, l4.w}, nd4 = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
          const f32x4_t e = s[iq][jk] * p.scale_log2e + nl4;
          f32x4_t pv;
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[r] = fast_exp2(e[r]);
          s[iq][jk] = pv;
          dp[iq][jk] = pv * (dp[iq][jk] * p.scale + nd4);
        }
      } else
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
          float addk = kb2[jk];
          if (FULL) addk += fb[(size_t)min(q0 + iq * 16 + lg * 4 + r, p.nq - 1) * p.nk + min(k0 + jk * 16 + lq, p.nk - 1)];
          const float pv = fast_exp2(s[iq][jk][r] * p.scale_log2e + addk - lv[r]);
          s[iq][jk][r] = pv;
          dp[iq][jk][r] = pv * (dp[iq][jk][r] - dd[r]) * p.scale;
        }
    }
    // dV^T += dO^T P ,  dK^T += Q^T dS : k-slot (lg, e): e<4 -> query block 2kk, row lg*4+e ; e>=4 -> block 2kk+1
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t pf[2], sf[2];
#pragma unroll
      for (int jk = 0; jk < 2; ++jk) {
        uint4 a, c;
        a.x = pack2bf(s[2 * kk][jk][0], s[2 * kk][jk][1]);
        a.y = pack2bf(s[2 * kk][jk][2], s[2 * kk][jk][3]);
        a.z = pack2bf(s[2 * kk + 1][jk][0], s[2 * kk + 1][jk][1]);
        a.w = pack2bf(s[2 * kk + 1][jk][2], s[2 * kk + 1][jk][3]);
        c.x = pack2bf(dp[2 * kk][jk][0], dp[2 * kk][jk][1]);
        c.y = pack2bf(dp[2 * kk][jk][2], dp[2 * kk][jk][3]);
        c.z = pack2bf(dp[2 * kk + 1][jk][0], dp[2 * kk + 1][jk][1]);
        c.w = pack2bf(dp[2 * kk + 1][jk][2], dp[2 * kk + 1][jk][3]);
        pf[jk] = __builtin_bit_cast(bf16x8_t, a);
        sf[jk] = __builtin_bit_cast(bf16x8_t, c);
      }
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        bf16x8_t daf, qaf;    // dO^T / Q^T rows d = jd * 16 + lq over the 32 queries of chunk kk
        if constexpr (TR) {
          daf = tr_frag(dOs, ATT_LDK, kk * 32, jd * 16, lq, lg);
          qaf = tr_frag(Qs, ATT_LDK, kk * 32, jd * 16, lq, lg);
        } else {
          const bf16_t* dr = dOTs + (jd * 16 + lq) * ATT_LDV + kk * 32 + lg * 4;
          const bf16_t* qr = QTs + (jd * 16 + lq) * ATT_LDV + kk * 32 + lg * 4;
          const uint2 dlo = *reinterpret_cast<const uint2*>(dr), dhi = *reinterpret_cast<const uint2*>(dr + 16);
          const uint2 qlo = *reinterpret_cast<const uint2*>(qr), qhi = *reinterpret_cast<const uint2*>(qr + 16);
          daf = __builtin_bit_cast(bf16x8_t, make_uint4(dlo.x, dlo.y, dhi.x, dhi.y));
          qaf = __builtin_bit_cast(bf16x8_t, make_uint4(qlo.x, qlo.y, qhi.x, qhi.y));
        }
#pragma unroll
        for (int jk = 0; jk < 2; ++jk) {
          dv[jd][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(daf, pf[jk], dv[jd][jk], 0, 0, 0);
          dk[jd][jk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qaf, sf[jk], dk[jd][jk], 0, 0, 0);
        }
      }
    }
  }
  if (p.part) {   // split over queries: fp32 partials
    const size_t plane = (size_t)p.batch * p.k_rows * p.hp;
    float* pk = p.part + (size_t)blockIdx.z * 2 * plane;
#pragma unroll
    for (int jk = 0; jk < 2; ++jk) {
      const int key = k0 + jk * 16 + lq;
      if (key < p.nk) {
        float* krow = pk + ((size_t)b * p.k_rows + key) * p.hp + h * 64;
        float* vrow = krow + plane;
#pragma unroll
        for (int jd = 0; jd < 4; ++jd) {
          *reinterpret_cast<float4*>(krow + jd * 16 + lg * 4) = make_float4(dk[jd][jk][0], dk[jd][jk][1], dk[jd][jk][2], dk[jd][jk][3]);
          *reinterpret_cast<float4*>(vrow + jd * 16 + lg * 4) = make_float4(dv[jd][jk][0], dv[jd][jk][1], dv[jd][jk][2], dv[jd][jk][3]);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int jk = 0; jk < 2; ++jk) {
    const int key = k0 + jk * 16 + lq;
    if (key < p.nk) {
      bf16_t* vrow = p.dv + ((size_t)b * p.k_rows + key) * p.dv_ld + h * 64;
      bf16_t* krow = p.dk + ((size_t)b * p.k_rows + key) * p.dk_ld + h * 64;
#pragma unroll
      for (int jd = 0; jd < 4; ++jd) {
        uint2 a, c;
        a.x = pack2bf(dv[jd][jk][0], dv[jd][jk][1]);
        a.y = pack2bf(dv[jd][jk][2], dv[jd][jk][3]);
        c.x = pack2bf(dk[jd][jk][0], dk[jd][jk][1]);
        c.y = pack2bf(dk[jd][jk][2], dk[jd][jk][3]);
        *reinterpret_cast<uint2*>(vrow + jd * 16 + lg * 4) = a;
        *reinterpret_cast<uint2*>(krow + jd * 16 + lg * 4) = c;
      }
    }
  }
}

// folds the query splits of the dkv kernel: dk/dv[b][key][c] = sum_z part[z][.][b][key][c]  (keys < nk)
__global__ void attn_bwd_fold_kernel(const float* __restrict__ part, int nz, int batch, int k_rows, int nk, int hp,
                                     bf16_t* __restrict__ dk, int dk_ld, bf16_t* __restrict__ dv, int dv_ld) {
  const size_t plane = (size_t)batch * k_rows * hp;
  const long long total = (long long)batch * nk * hp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % hp);
    const long long r = i / hp;
    const int key = (int)(r % nk), b = (int)(r / nk);
    const size_t off = ((size_t)b * k_rows + key) * hp + c;
    float a = 0.f, v = 0.f;
    for (int z = 0; z < nz; ++z) { a += part[(size_t)z * 2 * plane + off]; v += part[(size_t)z * 2 * plane + plane + off]; }
    dk[((size_t)b * k_rows + key) * dk_ld + c] = f2bf(a);
    dv[((size_t)b * k_rows + key) * dv_ld + c] = f2bf(v);
  }
}

// D[b][h][q] = sum_d dO[q][h*64+d] * O[q][h*64+d]: one wave per query row, 8 lanes per head
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const bf16_t* __restrict__ dO, int do_ld,
                                                          const bf16_t* __restrict__ o, int o_ld, float* __restrict__ dsum,
                                                          int batch, int heads, int nq) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)batch * nq) return;
  const int b = (int)(row / nq), qi = (int)(row - (long long)b * nq);
  for (int c8 = lane; c8 < heads * 8; c8 += 64) {
    float a[8], g[8];
    unpack8(*reinterpret_cast<const uint4*>(dO + (size_t)row * do_ld + c8 * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(o + (size_t)row * o_ld + c8 * 8), g);
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += a[e] * g[e];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if ((lane & 7) == 0) dsum[((size_t)b * heads + (c8 >> 3)) * nq + qi] = acc;
  }
}

static ctta_status attention_bwd_impl(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vn,
                                      int vn_ld, int vn_rows, const void* kt, int kt_ld, const void* qt,
                                      const void* dot, int qt_ld, const float* bias, const void* out, int out_ld,
                                      const void* dout, int do_ld, const float* lse, float* dsum, void* dq, int dq_ld,
                                      void* dk, int dk_ld, void* dv, int dv_ld, int batch, int heads, int nq, int nk,
                                      float scale, float* partial, int64_t partial_floats, void* stream,
                                      const float* full_bias, int full_nb, const void* vt = nullptr, int vt_ld = 0) {
  const bool tr = vt != nullptr;      // operands read where they lie (ctta_attention_bwd_inplace)
  CTTA_REQUIRE(q && k && out && dout && lse && dsum && dq && dk && dv && (tr ? !full_bias : (vn && kt && qt && dot)),
               "attention_bwd: null pointer");
  CTTA_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && do_ld % 8 == 0 && out_ld % 8 == 0 && dq_ld % 4 == 0 && dk_ld % 4 == 0 && dv_ld % 4 == 0 &&
                   (tr ? (vt_ld % 8 == 0 && vt_ld >= nk) : (vn_ld % 8 == 0 && kt_ld % 64 == 0 && qt_ld % 64 == 0)),
               "attention_bwd: row strides (transposed operands need 64-multiples)");
  CTTA_REQUIRE(nq > 0 && nk > 0 && k_rows >= nk && (tr || (vn_rows >= nk && kt_ld >= nk && qt_ld >= nq)), "attention_bwd: bad lengths");
  hipStream_t s = (hipStream_t)stream;
  constexpr bool fuse_dsum = true;      // D = rowsum(dO * O) inside the dq kernel (round 5); the row-dot launch serves callers without it
  if (!fuse_dsum) {
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)(((long long)batch * nq + 3) / 4)), dim3(256), 0, s,
                       (const bf16_t*)dout, do_ld, (const bf16_t*)out, out_ld, dsum, batch, heads, nq);
    CTTA_LAUNCH_CHECK();
  }
  AttnBwdParams p;
  p.o = (const bf16_t*)out; p.o_ld = out_ld; p.dsum_out = fuse_dsum ? dsum : nullptr;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vn = (const bf16_t*)vn; p.kt = (const bf16_t*)kt;
  p.qt = (const bf16_t*)qt; p.dot = (const bf16_t*)dot; p.dO = (const bf16_t*)dout;
  p.q_ld = q_ld; p.k_ld = k_ld; p.k_rows = k_rows; p.vn_ld = vn_ld; p.vn_rows = vn_rows; p.kt_ld = kt_ld; p.qt_ld = qt_ld;
  p.do_ld = do_ld; p.bias = bias; p.lse = lse; p.dsum = dsum;
  p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.dq_ld = dq_ld; p.dk_ld = dk_ld; p.dv_ld = dv_ld;
  p.heads = heads; p.nq = nq; p.nk = nk; p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
  p.full_bias = full_bias; p.full_nb = full_nb > 0 ? full_nb : 1;
  p.vt = (const bf16_t*)vt; p.vt_ld = vt_ld;
  const bool prof = ctta_prof_active();
  // executed flops: 7 products of 2*nq*nk*64 per head (S and dP are computed by both kernels)
  if (prof) ctta_prof_begin(1, 1, nq, nk, 448, (long long)batch * heads, s);
  const bool plain = !full_bias && !p.bias && nk % 64 == 0 && nq % 64 == 0 && attn_plain_enabled();   // self-attention over whole tiles
  const dim3 gq((nq + 127) / 128, batch * heads);
  if (full_bias) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false, false>), gq, dim3(256), 0, s, p);
  else if (plain && tr) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true, true>), gq, dim3(256), 0, s, p);
  else if (plain) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true, false>), gq, dim3(256), 0, s, p);
  else if (tr) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false, true>), gq, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false, false>), gq, dim3(256), 0, s, p);
  // few keys (cross-attention): split the query walk so that the launch still fills the chip
  const int ntiles = (nq + 63) / 64, kblocks = (nk + 127) / 128, hp = heads * 64;
  int nz = 1;
  if (partial) {
    while (nz < 32 && (long long)kblocks * batch * heads * nz < 512 && ntiles / (nz * 2) >= 4) nz *= 2;
    while (nz > 1 && (int64_t)nz * 2 * batch * k_rows * hp > partial_floats) nz /= 2;
  }
  p.q_tiles_per_split = (ntiles + nz - 1) / nz;
  p.part = nz > 1 ? partial : nullptr;
  p.batch = batch; p.hp = hp;
  const dim3 gk(kblocks, batch * heads, nz);
  if (full_bias) hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, false, false>), gk, dim3(256), 0, s, p);
  else if (plain && tr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true, true>), gk, dim3(256), 0, s, p);
  else if (plain) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true, false>), gk, dim3(256), 0, s, p);
  else if (tr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, false, true>), gk, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, false, false>), gk, dim3(256), 0, s, p);
  if (nz > 1) {
    CTTA_REQUIRE(dk_ld >= hp && dv_ld >= hp, "attention_bwd: split path needs dk/dv rows of at least heads*64");
    const long long total = (long long)batch * nk * hp;
    hipLaunchKernelGGL(attn_bwd_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, partial, nz, batch,
                       k_rows, nk, hp, (bf16_t*)dk, dk_ld, (bf16_t*)dv, dv_ld);
  }
  if (prof) ctta_prof_end(s);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}

extern "C" ctta_status ctta_attention_bwd(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vn,
                                          int vn_ld, int vn_rows, const void* kt, int kt_ld, const void* qt,
                                          const void* dot, int qt_ld, const float* bias, const void* out, int out_ld,
                                          const void* dout, int do_ld, const float* lse, float* dsum, void* dq, int dq_ld,
                                          void* dk, int dk_ld, void* dv, int dv_ld, int batch, int heads, int nq, int nk,
                                          float scale, float* partial, int64_t partial_floats, void* stream) {
  return attention_bwd_impl(q, q_ld, k, k_ld, k_rows, vn, vn_ld, vn_rows, kt, kt_ld, qt, dot, qt_ld, bias, out, out_ld, dout,
                            do_ld, lse, dsum, dq, dq_ld, dk, dk_ld, dv, dv_ld, batch, heads, nq, nk, scale, partial,
                            partial_floats, stream, nullptr, 1);
}
// The same backward with every operand read WHERE IT LIES (round 5): q / k / dout natural, V as the forward keeps it
// (vt [B][heads*64][vt_ld], what ctta_attention consumes) -- no K^T, Q^T, dO^T or natural-V copies: the kernels take the
// operands that need the other orientation out of the tiles they stage anyway through transposing LDS reads.
extern "C" ctta_status ctta_attention_bwd_inplace(const void* q, int q_ld, const void* k, int k_ld, int k_rows, const void* vt,
                                                  int vt_ld, const float* bias, const void* out, int out_ld, const void* dout,
                                                  int do_ld, const float* lse, float* dsum, void* dq, int dq_ld, void* dk,
                                                  int dk_ld, void* dv, int dv_ld, int batch, int heads, int nq, int nk,
                                                  float scale, float* partial, int64_t partial_floats, void* stream) {
  CTTA_REQUIRE(vt, "attention_bwd_inplace: null pointer");
  return attention_bwd_impl(q, q_ld, k, k_ld, k_rows, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, 0, bias, out, out_ld, dout, do_ld,
                            lse, dsum, dq, dq_ld, dk, dk_ld, dv, dv_ld, batch, heads, nq, nk, scale, partial, partial_floats, stream,
                            nullptr, 1, vt, vt_ld);
}
// Backward of ctta_attention_fullbias (same operands as ctta_attention_bwd; no per-key bias, no split path)
extern "C" ctta_status ctta_attention_fullbias_bwd(const void* q, int q_ld, const void* k, int k_ld, int k_rows,
                                                   const void* vn, int vn_ld, int vn_rows, const void* kt, int kt_ld,
                                                   const void* qt, const void* dot, int qt_ld,
                                                   const float* full_bias_log2, int n_bias_batches, const void* out,
                                                   int out_ld, const void* dout, int do_ld, const float* lse, float* dsum,
                                                   void* dq, int dq_ld, void* dk, int dk_ld, void* dv, int dv_ld, int batch,
                                                   int heads, int nq, int nk, float scale, void* stream) {
  CTTA_REQUIRE(full_bias_log2 && n_bias_batches >= 1, "attention_fullbias_bwd: null bias table");
  return attention_bwd_impl(q, q_ld, k, k_ld, k_rows, vn, vn_ld, vn_rows, kt, kt_ld, qt, dot, qt_ld, nullptr, out, out_ld,
                            dout, do_ld, lse, dsum, dq, dq_ld, dk, dk_ld, dv, dv_ld, batch, heads, nq, nk, scale, nullptr, 0,
                            stream, full_bias_log2, n_bias_batches);
}
