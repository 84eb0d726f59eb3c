// FLAN-T5 encoder engine: the frozen text encoder the reference runs before every U-Net query,
//   self.text_encoder(input_ids=..., attention_mask=...)[0]        models/audio_distilled_model.py:208-214
// i.e. transformers' T5EncoderModel (T5Stack / T5Block / T5Attention / T5LayerNorm / T5DenseGatedActDense of
// models/t5/modeling_t5.py, transformers==4.29.2 in environment.yml:157 -- third-party, not vendored; the algorithm
// is restated in oracle/t5.py and pinned to the installed release's own module).
// Layout: tokens are rows, batches Lp = round_up(L, 8) rows apart (pad rows are zero and never attended);
// the residual stream stays fp32 ([M][d_model]; every residual add is a conv_gemm fp32-accumulate epilogue),
// normalised activations and projections are bf16.  Per block: RMSNorm -> fused [q|k] linear + V^T GEMM ->
// flash attention with the relative-position table (ctta_attention_rel, no 1/sqrt(d) scaling) -> o-projection
// accumulated into the stream -> RMSNorm -> fused [wi_0|wi_1] linear -> gelu_new(.)*(.) -> wo accumulated.
#include "engine_common.h"

#include <math.h>

__global__ void t5_embed_kernel(const int64_t* __restrict__ ids, int L, int Lp, const float* __restrict__ table, int d,
                                int vocab, float* __restrict__ h) {
  const int row = blockIdx.x;   // b * Lp + l
  const int b = row / Lp, l = row - b * Lp;
  float4* dst = reinterpret_cast<float4*>(h + (size_t)row * d);
  if (l >= L) {
    for (int i = threadIdx.x; i < d / 4; i += blockDim.x) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  long long id = ids[(size_t)b * L + l];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;   // torch would raise; ids come from the tokenizer, checked on the host
  const float4* src = reinterpret_cast<const float4*>(table + (size_t)id * d);
  for (int i = threadIdx.x; i < d / 4; i += blockDim.x) dst[i] = src[i];
}

// T5LayerNorm: x * rsqrt(mean(x^2) + eps) * weight, statistics in fp32.  One wave per row.
// out_bf16 [rows][d] and/or out_f32 (final norm): rows (b, l < L) written densely as [B][L][d].
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float eps, int d, long long rows, bf16_t* __restrict__ out_bf16,
                                                         float* __restrict__ out_f32, int L, int Lp) {
  const int lane = threadIdx.x & 63;
  const long long row = blockIdx.x * 4LL + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
  float ss = 0.f;
  for (int i = lane; i < d / 4; i += 64) {
    const float4 v = xr[i];
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  const float r = rsqrtf(ss / (float)d + eps);
  const float4* wr = reinterpret_cast<const float4*>(w);
  float* of = nullptr;
  if (out_f32) {
    const long long b = row / Lp;
    const int l = (int)(row - b * Lp);
    if (l < L) of = out_f32 + ((size_t)b * L + l) * d;
  }
  for (int i = lane; i < d / 4; i += 64) {
    const float4 v = xr[i], g = wr[i];
    const float4 o = make_float4(v.x * r * g.x, v.y * r * g.y, v.z * r * g.z, v.w * r * g.w);
    if (out_bf16) {
      uint2 pk;
      pk.x = pack2bf(o.x, o.y);
      pk.y = pack2bf(o.z, o.w);
      *reinterpret_cast<uint2*>(out_bf16 + (size_t)row * d + i * 4) = pk;
    }
    if (of) reinterpret_cast<float4*>(of)[i] = o;
  }
}

// T5DenseGatedActDense's product: out = gelu_new(x[:, :ffp]) * x[:, ffp:]   (NewGELUActivation, tanh form)
__global__ void t5_gated_gelu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, long long rows, int ffp) {
  const int vc = ffp / 8;
  const long long total = rows * vc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % vc);
    const long long r = idx / vc;
    float a[8], g[8];
    unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * 2 * ffp + v * 8), a);
    unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * 2 * ffp + ffp + v * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = a[e];
      const float u = 0.7978845608028654f * (t + 0.044715f * t * t * t);
      a[e] = 0.5f * t * (1.0f + tanhf(u)) * g[e];
    }
    *reinterpret_cast<uint4*>(out + (size_t)r * ffp + v * 8) = pack8(a);
  }
}

// rel[h][i] = emb[bucket[i - (L-1) + (max_len-1)]][h] * log2(e), i in [0, 2L-1): offset i - (L-1) = key - query
__global__ void t5_rel_bias_kernel(const float* __restrict__ emb, const int32_t* __restrict__ bucket, int heads, int L,
                                   int max_len, float* __restrict__ rel) {
  const int n = 2 * L - 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= heads * n) return;
  const int h = idx / n, i = idx - h * n;
  rel[idx] = emb[(size_t)bucket[i - (L - 1) + (max_len - 1)] * heads + h] * 1.4426950408889634f;
}

// extended attention mask: 0 for real tokens, a large negative number for padding (finfo.min in the reference; any
// value that underflows exp() gives the same probabilities as long as one key is real)
__global__ void t5_mask_bias_kernel(const uint8_t* __restrict__ mask, int n, float* __restrict__ bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) bias[i] = mask[i] ? 0.f : -1e30f;
}

struct T5Block {
  float *ln1 = nullptr, *ln2 = nullptr;
  PackedW qk, v, o, wi, wo;
};

struct ctta_t5 {
  ctta_t5_config cfg;
  WeightStore store;
  Arena arena;
  SplitWs splitws;
  float* embed = nullptr;        // fp32 [vocab][d_model]
  float* rel_emb = nullptr;      // fp32 [buckets][heads]
  float* ln_f = nullptr;
  int32_t* bucket = nullptr;     // device [2*max_len-1]: bucket(key - query), the reference's own float arithmetic
  std::vector<T5Block> blocks;
  int inner = 0, ffp = 0;
};

// modeling_t5.py _relative_position_bucket (bidirectional): half the buckets per sign, first half of each exact,
// the rest log-spaced up to max_distance -- in float32 like torch evaluates it.
static int t5_bucket(int rel, int num_buckets, int max_distance) {
  int nb = num_buckets / 2, ret = rel > 0 ? nb : 0;
  const int n = rel < 0 ? -rel : rel;
  const int max_exact = nb / 2;
  if (n < max_exact) return ret + n;
  const float v = logf((float)n / (float)max_exact) / (float)log((double)max_distance / max_exact) * (float)(nb - max_exact);
  int large = max_exact + (int)v;
  if (large > nb - 1) large = nb - 1;
  return ret + large;
}

static ctta_status t5_build(ctta_t5* T) {
  const ctta_t5_config& cfg = T->cfg;
  WeightStore& ws = T->store;
  const int d = cfg.d_model, inner = cfg.num_heads * cfg.d_kv, ff = cfg.d_ff, ffp = round_up(ff, 64);
  T->inner = inner; T->ffp = ffp;
  CTTA_TRY(ws.add_vector("shared.weight", cfg.vocab_size * d, &T->embed));
  CTTA_TRY(ws.add_vector("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
                         cfg.rel_buckets * cfg.num_heads, &T->rel_emb));
  CTTA_TRY(ws.add_vector("encoder.final_layer_norm.weight", d, &T->ln_f));
  T->blocks.resize(cfg.num_layers);
  const std::vector<int32_t> id_d = identity_map(d, d), id_inner = identity_map(inner, inner);
  for (int i = 0; i < cfg.num_layers; ++i) {
    T5Block& B = T->blocks[i];
    const std::string p = "encoder.block." + std::to_string(i) + ".layer.";
    CTTA_TRY(ws.add_vector(p + "0.layer_norm.weight", d, &B.ln1));
    CTTA_TRY(ws.add_vector(p + "1.layer_norm.weight", d, &B.ln2));
    bf16_t* qk = ws.arena.get<bf16_t>((size_t)2 * inner * d);
    bf16_t* wi = ws.arena.get<bf16_t>((size_t)2 * ffp * d);
    if (!qk || !wi) { ctta_set_error("weight store exhausted"); return CTTA_ERR_NOMEM; }
    PackedW tmp;
    CTTA_TRY(make_linear(ws, p + "0.SelfAttention.q.weight", "", inner, d, id_inner, id_d, &tmp, qk));
    CTTA_TRY(make_linear(ws, p + "0.SelfAttention.k.weight", "", inner, d, id_inner, id_d, &tmp, qk + (size_t)inner * d));
    B.qk.w = qk; B.qk.n = 2 * inner; B.qk.k_pad = d;
    CTTA_TRY(make_linear(ws, p + "0.SelfAttention.v.weight", "", inner, d, id_inner, id_d, &B.v));
    CTTA_TRY(make_linear(ws, p + "0.SelfAttention.o.weight", "", d, inner, id_d, id_inner, &B.o));
    const std::vector<int32_t> ff_rows = identity_map(ff, ffp);
    CTTA_TRY(make_linear(ws, p + "1.DenseReluDense.wi_0.weight", "", ff, d, ff_rows, id_d, &tmp, wi));
    CTTA_TRY(make_linear(ws, p + "1.DenseReluDense.wi_1.weight", "", ff, d, ff_rows, id_d, &tmp, wi + (size_t)ffp * d));
    B.wi.w = wi; B.wi.n = 2 * ffp; B.wi.k_pad = d;
    CTTA_TRY(make_linear(ws, p + "1.DenseReluDense.wo.weight", "", d, ff, id_d, identity_map(ff, ffp), &B.wo));
  }
  std::vector<int32_t> bucket(2 * cfg.max_len - 1);
  for (int i = 0; i < 2 * cfg.max_len - 1; ++i)
    bucket[i] = t5_bucket(i - (cfg.max_len - 1), cfg.rel_buckets, cfg.rel_max_distance);
  CTTA_TRY(ws.upload(bucket, &T->bucket));
  return CTTA_OK;
}

static ctta_status t5_linear(RunCtx& c, const PackedW& P, const bf16_t* x, int x_ld, int64_t rows, void* out, int ldc,
                             bool f32_accumulate) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = x; d.c0 = x_ld;
  d.batch = 1; d.hi = (int)rows; d.wi = 1; d.ho = (int)rows; d.wo = 1;
  d.w = P.w; d.k_pad = P.k_pad; d.n = P.n;
  d.out = out; d.ldc = ldc;
  if (f32_accumulate) { d.out_f32 = 1; d.accumulate = 1; }
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

static ctta_status t5_forward_impl(ctta_t5* T, bool dry, const int64_t* ids, const uint8_t* mask, int B, int L, float* out,
                                   hipStream_t stream) {
  const ctta_t5_config& cfg = T->cfg;
  RunCtx c;
  c.arena = &T->arena; c.stream = stream; c.dry = dry;
  c.taps = nullptr; c.gn_scratch = nullptr; c.gn_scratch_floats = 0;
  Arena& A = T->arena;
  A.reset();
  const int d = cfg.d_model, H = cfg.num_heads, inner = T->inner, ffp = T->ffp;
  const int Lp = round_up(L, 8);
  const int64_t M = (int64_t)B * Lp;
  float* h = A.get<float>((size_t)M * d); ALLOC_OR_FAIL(h);
  bf16_t* n = A.get<bf16_t>((size_t)M * d); ALLOC_OR_FAIL(n);
  bf16_t* qk = A.get<bf16_t>((size_t)M * 2 * inner); ALLOC_OR_FAIL(qk);
  bf16_t* vt = A.get<bf16_t>((size_t)B * inner * Lp); ALLOC_OR_FAIL(vt);
  bf16_t* att = A.get<bf16_t>((size_t)M * inner); ALLOC_OR_FAIL(att);
  bf16_t* f = A.get<bf16_t>((size_t)M * 2 * ffp); ALLOC_OR_FAIL(f);
  bf16_t* g = A.get<bf16_t>((size_t)M * ffp); ALLOC_OR_FAIL(g);
  float* rel = A.get<float>((size_t)H * (2 * L - 1)); ALLOC_OR_FAIL(rel);
  float* kbias = A.get<float>((size_t)B * L); ALLOC_OR_FAIL(kbias);
  const unsigned nrm_blocks = (unsigned)((M + 3) / 4);
  if (!dry) {
    hipLaunchKernelGGL(t5_embed_kernel, dim3((unsigned)M), dim3(256), 0, stream, ids, L, Lp, T->embed, d, cfg.vocab_size, h);
    CTTA_LAUNCH_CHECK();
    hipLaunchKernelGGL(t5_rel_bias_kernel, dim3((H * (2 * L - 1) + 255) / 256), dim3(256), 0, stream, T->rel_emb, T->bucket,
                       H, L, cfg.max_len, rel);
    CTTA_LAUNCH_CHECK();
    hipLaunchKernelGGL(t5_mask_bias_kernel, dim3((B * L + 255) / 256), dim3(256), 0, stream, mask, B * L, kbias);
    CTTA_LAUNCH_CHECK();
    // pad rows of att are never written by the attention kernel but are read (and ignored) by the o-projection
    CTTA_CHECK_HIP(ctta_zero_async(att, (size_t)M * inner * sizeof(bf16_t), stream));
  }
  for (int i = 0; i < cfg.num_layers; ++i) {
    const T5Block& Bk = T->blocks[i];
    if (!dry) {
      hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(nrm_blocks), dim3(256), 0, stream, h, Bk.ln1, cfg.eps, d, (long long)M, n,
                         (float*)nullptr, L, Lp);
      CTTA_LAUNCH_CHECK();
    }
    CTTA_TRY(t5_linear(c, Bk.qk, n, d, M, qk, 2 * inner, false));
    CTTA_TRY(run_vt(c, Bk.v, n, B, Lp, Lp, vt, Lp));
    RUN(c, ctta_attention_rel(qk, 2 * inner, Lp, qk + inner, 2 * inner, Lp, vt, Lp, kbias, rel, att, inner, B, H, L, L, 1.0f,
                              stream));
    CTTA_TRY(t5_linear(c, Bk.o, att, inner, M, h, d, true));
    if (!dry) {
      hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(nrm_blocks), dim3(256), 0, stream, h, Bk.ln2, cfg.eps, d, (long long)M, n,
                         (float*)nullptr, L, Lp);
      CTTA_LAUNCH_CHECK();
    }
    CTTA_TRY(t5_linear(c, Bk.wi, n, d, M, f, 2 * ffp, false));
    if (!dry) {
      const long long total = M * (ffp / 8);
      hipLaunchKernelGGL(t5_gated_gelu_kernel, dim3((unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256)),
                         dim3(256), 0, stream, f, g, (long long)M, ffp);
      CTTA_LAUNCH_CHECK();
    }
    CTTA_TRY(t5_linear(c, Bk.wo, g, ffp, M, h, d, true));
  }
  if (!dry) {
    hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3(nrm_blocks), dim3(256), 0, stream, h, T->ln_f, cfg.eps, d, (long long)M,
                       (bf16_t*)nullptr, out, L, Lp);
    CTTA_LAUNCH_CHECK();
  }
  return CTTA_OK;
}

extern "C" void ctta_t5_destroy(ctta_t5* T) {
  if (!T) return;
  T->store.destroy();
  if (T->arena.base) (void)hipFree(T->arena.base);
  T->splitws.destroy();
  delete T;
}

extern "C" ctta_status ctta_t5_create(const ctta_t5_config* cfg, const ctta_tensor* weights, int n_weights, void* stream,
                                      ctta_t5** out) {
  CTTA_REQUIRE(cfg && weights && out, "t5_create: null pointer");
  CTTA_REQUIRE(cfg->d_kv == 64, "t5_create: d_kv=%d (every T5 size uses 64; the attention kernel is built for it)", cfg->d_kv);
  CTTA_REQUIRE(cfg->d_model % 64 == 0 && cfg->d_ff % 8 == 0 && cfg->num_layers >= 1 && cfg->num_heads >= 1,
               "t5_create: d_model=%d must be a multiple of 64, d_ff=%d of 8", cfg->d_model, cfg->d_ff);
  CTTA_REQUIRE(cfg->max_batch >= 1 && cfg->max_len >= 1 && cfg->rel_buckets >= 4 && cfg->rel_max_distance > cfg->rel_buckets / 4,
               "t5_create: bad max_batch / max_len / relative-attention settings");
  hipStream_t s = (hipStream_t)stream;
  ctta_t5* T = new ctta_t5();
  T->cfg = *cfg;
  size_t store = 64 << 20;
  for (int i = 0; i < n_weights; ++i) store += (size_t)tensor_numel(&weights[i]) * 4 + 8192;   // bf16 packs + fp32 embedding + maps
  ctta_status st = T->store.init(store);
  if (st != CTTA_OK) { delete T; return st; }
  WeightTable wt;
  wt.build(weights, n_weights);
  st = t5_build(T);
  if (st == CTTA_OK) st = T->store.run_all(wt, s);
  if (st == CTTA_OK) {
    T->arena.dry = true;
    st = t5_forward_impl(T, true, nullptr, nullptr, cfg->max_batch, cfg->max_len, nullptr, s);
  }
  if (st == CTTA_OK) {
    const size_t bytes = T->arena.peak + (1 << 20);
    T->arena.dry = false;
    T->arena.cap = bytes;
    if (hipMalloc((void**)&T->arena.base, bytes) != hipSuccess) {
      ctta_set_error("t5_create: hipMalloc of %zu-byte activation arena failed", bytes);
      st = CTTA_ERR_NOMEM;
    } else {
      st = T->splitws.init();
    }
  }
  if (st == CTTA_OK && hipStreamSynchronize(s) != hipSuccess) { ctta_set_error("t5_create: stream sync failed"); st = CTTA_ERR_HIP; }
  if (st != CTTA_OK) { ctta_t5_destroy(T); return st; }
  *out = T;
  return CTTA_OK;
}

extern "C" ctta_status ctta_t5_encode(ctta_t5* T, const int64_t* input_ids, const uint8_t* attention_mask, int batch, int len,
                                      float* last_hidden_state, void* stream) {
  CTTA_REQUIRE(T && input_ids && attention_mask && last_hidden_state, "t5_encode: null pointer");
  CTTA_REQUIRE(batch >= 1 && batch <= T->cfg.max_batch && len >= 1 && len <= T->cfg.max_len,
               "t5_encode: batch %d / length %d outside the handle's limits (%d, %d)", batch, len, T->cfg.max_batch,
               T->cfg.max_len);
  WsBind bind(T->splitws);
  return t5_forward_impl(T, false, input_ids, attention_mask, batch, len, last_hidden_state, (hipStream_t)stream);
}

extern "C" size_t ctta_t5_arena_bytes(const ctta_t5* T) { return T ? T->arena.cap + T->store.arena.cap : 0; }
