// AudioLDM VAE decoder and HiFi-GAN vocoder engines.
//   VAE:      AutoencoderKL.decode_first_stage -> decode -> Decoder.forward
//             (audioldm/variational_autoencoder/autoencoder.py:91-106, modules.py:650-683),
//             ResnetBlock.forward modules.py:155-175, AttnBlock.forward :204-230, Upsample :53-57.
//   HiFi-GAN: Generator.forward hifigan/models.py:101-117, ResBlock.forward :56-63.
// Layout: NHWC bf16 for the decoder; (B, L, C) bf16 for the vocoder (the decoder's
// (B,1,T,F) mel IS the vocoder's (B, T, F=num_mels) input, autoencoder.py:109).
#include "engine_common.h"

#include <stdlib.h>

#include <math.h>

// z (B,zc,H,W) f32 -> post_quant_conv(z / scale) as NHWC bf16 with channels padded to cpad.
// (autoencoder.py:99,105: 1x1 conv embed_dim -> z_channels, tiny: done in fp32 per pixel)
__global__ void post_quant_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                  const float* __restrict__ b, float inv_scale, int B, int zin, int zout,
                                  int HW, int cpad, bf16_t* __restrict__ out) {
  const long long total = (long long)B * HW;
  const long long pix = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (pix >= total) return;
  const int bb = (int)(pix / HW);
  const int hw = (int)(pix - (long long)bb * HW);
  float in[16];
  for (int c = 0; c < zin; ++c) in[c] = z[((size_t)bb * zin + c) * HW + hw] * inv_scale;
  for (int o = 0; o < cpad; ++o) {
    float acc = 0.f;
    if (o < zout) {
      acc = b[o];
      for (int c = 0; c < zin; ++c) acc += w[o * zin + c] * in[c];
    }
    out[(size_t)pix * cpad + o] = f2bf(acc);
  }
}

__global__ void repack_small_w3_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int c,
                                       int khw) {
  const int total = n * c * khw;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cc = i % c;
  const int t = (i / c) % khw;
  const int nn = i / (c * khw);
  dst[i] = src[((size_t)nn * c + cc) * khw + t];
}

static ctta_status add_small_conv_w(WeightStore& ws, const std::string& key, int co, int ci, int khw, float** out) {
  float* dst = ws.arena.get<float>((size_t)co * ci * khw);
  if (!dst) { ctta_set_error("weight store exhausted"); return CTTA_ERR_NOMEM; }
  *out = dst;
  ws.jobs.push_back([=](const WeightTable& wt, hipStream_t s) -> ctta_status {
    const ctta_tensor* t = wt.find(key);
    if (!t) { ctta_set_error("missing state-dict key '%s'", key.c_str()); return CTTA_ERR_MISSING_KEY; }
    if (tensor_numel(t) != (int64_t)co * ci * khw) { ctta_set_error("size mismatch for '%s'", key.c_str()); return CTTA_ERR_INVALID; }
    const int total = co * ci * khw;
    hipLaunchKernelGGL(repack_small_w3_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t->data, dst, co, ci, khw);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  });
  return CTTA_OK;
}

// ====================================================================================== VAE
struct VaeRes {
  int cin = 0, cout = 0;
  GNLayer n1, n2;
  ConvLayer c1, c2, sc;
  bool has_sc = false;
  // enable_grad: data-gradient operands and what the differentiable forward keeps for the backward
  ConvLayer d1, d2, dsc;
  struct Saved { const bf16_t* x = nullptr; bf16_t* t1 = nullptr; float *st1 = nullptr, *st2 = nullptr; int H = 0, W = 0; } sv;
};

struct VaeAttn {   // AttnBlock (modules.py:178-230): single head over H*W tokens, d = C
  GNLayer norm;
  ConvLayer q, k, v, proj;
  int C = 0;
  ConvLayer dq, dk, dv, dproj;
  struct Saved {
    const bf16_t* x = nullptr;
    bf16_t *q = nullptr, *k = nullptr, *vt = nullptr, *p = nullptr;
    float* st = nullptr;
    int H = 0, W = 0;
  } sv;
};

struct ctta_vae {
  ctta_vae_config cfg;
  WeightStore store;
  Arena arena;
  std::vector<Tap> taps;
  float *pq_w = nullptr, *pq_b = nullptr;
  ConvLayer conv_in;
  VaeRes mid1, mid2;
  VaeAttn attn;
  std::vector<std::vector<VaeRes>> up;     // [level][block]
  std::vector<ConvLayer> upsample;         // [level] (level 0 unused)
  GNLayer norm_out;
  ConvLayer conv_out;
  int block_in = 0, c_last = 0;
  float* gn_scratch = nullptr;
  size_t gn_scratch_floats = 0;
  size_t gn_fpart_floats = 0;
  SplitWs splitws;
  // enable_grad (CLAPLoss's differentiable decode, tools/losses.py:294-296)
  ConvLayer d_conv_in;
  std::vector<ConvLayer> d_upsample;
  float* out_w = nullptr;                  // conv_out weight as fp32 [tap][C]
  struct Saved { const bf16_t* h_last = nullptr; float* st_out = nullptr; int B = 0; } sv;
};

struct VCtx : RunCtx {
  int B;
  size_t gn_need = 0;
  bool grad = false;   // differentiable forward: keep what vae_backward_impl reads
};

static const int kVaeGroups = 32;    // Normalize(): GroupNorm(32, eps=1e-6)  modules.py:38-41
static const float kVaeEps = 1e-6f;

static ctta_status vgn(VCtx& c, const GNLayer& g, const bf16_t* x, bf16_t* y, int hw, bool silu, float* stats = nullptr) {
  const size_t need = ctta_groupnorm_scratch_floats(c.B, hw, g.c, kVaeGroups);
  if (need > c.gn_need) c.gn_need = need;
  return run_gn(c, g, x, y, c.B, hw, kVaeGroups, kVaeEps, silu, stats);
}
// (mean, rstd) per (sample, group) kept for the backward
static float* stats_buf(VCtx& c) { return c.grad ? c.arena->get<float>((size_t)c.B * kVaeGroups * 2) : nullptr; }

static ctta_status make_vae_res(WeightStore& ws, const std::string& p, int cin, int cout, VaeRes* R, bool grad = false) {
  R->cin = cin; R->cout = cout;
  CTTA_TRY(make_gn(ws, p + "norm1.", cin, &R->n1));
  CTTA_TRY(make_conv(ws, p + "conv1.", cout, cin, cin, 3, 3, 1, 1, &R->c1));
  CTTA_TRY(make_gn(ws, p + "norm2.", cout, &R->n2));
  CTTA_TRY(make_conv(ws, p + "conv2.", cout, cout, cout, 3, 3, 1, 1, &R->c2));
  R->has_sc = cin != cout;
  if (R->has_sc) CTTA_TRY(make_conv(ws, p + "nin_shortcut.", cout, cin, cin, 1, 1, 1, 0, &R->sc));
  if (grad) {
    CTTA_TRY(make_conv_dgrad_from(ws, R->c1, cout, cin, &R->d1));
    CTTA_TRY(make_conv_dgrad_from(ws, R->c2, cout, cout, &R->d2));
    if (R->has_sc) CTTA_TRY(make_conv_dgrad_from(ws, R->sc, cout, cin, &R->dsc));
  }
  return CTTA_OK;
}

static ctta_status run_vae_res(VCtx& c, VaeRes& R, const bf16_t* x, int H, int W, bf16_t** out_p) {
  Arena& A = *c.arena;
  const size_t M = (size_t)c.B * H * W;
  bf16_t* out = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(out);
  bf16_t* t1 = nullptr;
  float *st1 = nullptr, *st2 = nullptr;
  if (c.grad) {   // conv1's output and both norms' statistics outlive the block
    t1 = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(t1);
    st1 = stats_buf(c); ALLOC_OR_FAIL(st1);
    st2 = stats_buf(c); ALLOC_OR_FAIL(st2);
    R.sv.x = x; R.sv.t1 = t1; R.sv.st1 = st1; R.sv.st2 = st2; R.sv.H = H; R.sv.W = W;
  }
  const size_t mk = A.mark();
  bf16_t* a = A.get<bf16_t>(M * R.cin); ALLOC_OR_FAIL(a);
  CTTA_TRY(vgn(c, R.n1, x, a, H * W, true, st1));
  if (!t1) { t1 = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(t1); }
  CTTA_TRY(run_conv2d(c, R.c1, a, c.B, H, W, false, t1, nullptr, 0, nullptr, 0));
  bf16_t* a2 = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(a2);
  CTTA_TRY(vgn(c, R.n2, t1, a2, H * W, true, st2));
  const bf16_t* res = x;
  if (R.has_sc) {
    bf16_t* r = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(r);
    CTTA_TRY(run_conv2d(c, R.sc, x, c.B, H, W, false, r, nullptr, 0, nullptr, 0));
    res = r;
  }
  CTTA_TRY(run_conv2d(c, R.c2, a2, c.B, H, W, false, out, nullptr, 0, res, R.cout));
  A.release(mk);
  *out_p = out;
  return CTTA_OK;
}

// AttnBlock: single head over N = H*W tokens, d = C.  Scores are materialised in fp32
// (B x N x N; 2 GiB at B=32, N=4096 -- 288 GB of HBM makes this the simple choice), softmaxed
// to bf16, and multiplied by V^T; all three products run on conv_gemm.
static ctta_status make_vae_attn(WeightStore& ws, const std::string& p, int C, VaeAttn* T, bool grad = false) {
  T->C = C;
  CTTA_TRY(make_gn(ws, p + "norm.", C, &T->norm));
  CTTA_TRY(make_conv(ws, p + "q.", C, C, C, 1, 1, 1, 0, &T->q));
  CTTA_TRY(make_conv(ws, p + "k.", C, C, C, 1, 1, 1, 0, &T->k));
  CTTA_TRY(make_conv(ws, p + "v.", C, C, C, 1, 1, 1, 0, &T->v));
  CTTA_TRY(make_conv(ws, p + "proj_out.", C, C, C, 1, 1, 1, 0, &T->proj));
  if (grad) {
    CTTA_TRY(make_conv_dgrad_from(ws, T->q, C, C, &T->dq));
    CTTA_TRY(make_conv_dgrad_from(ws, T->k, C, C, &T->dk));
    CTTA_TRY(make_conv_dgrad_from(ws, T->v, C, C, &T->dv));
    CTTA_TRY(make_conv_dgrad_from(ws, T->proj, C, C, &T->dproj));
  }
  return CTTA_OK;
}

static ctta_status run_vae_attn(VCtx& c, VaeAttn* V, const bf16_t* x, int H, int W, bf16_t** out_p) {
  Arena& A = *c.arena;
  const int N = H * W, C = V->C, B = c.B;
  CTTA_REQUIRE(N % 64 == 0 && C % 64 == 0, "vae attention: tokens=%d and channels=%d must be multiples of 64", N, C);
  const size_t M = (size_t)B * N;
  bf16_t* out = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(out);
  bf16_t *q = nullptr, *k = nullptr, *vt = nullptr, *p = nullptr;
  float* st = nullptr;
  if (c.grad) {   // q, k, V^T, the probabilities and the norm statistics outlive the block
    q = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(q);
    k = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(k);
    vt = A.get<bf16_t>((size_t)B * C * N); ALLOC_OR_FAIL(vt);
    p = A.get<bf16_t>((size_t)B * N * N); ALLOC_OR_FAIL(p);
    st = stats_buf(c); ALLOC_OR_FAIL(st);
    V->sv.x = x; V->sv.q = q; V->sv.k = k; V->sv.vt = vt; V->sv.p = p; V->sv.st = st; V->sv.H = H; V->sv.W = W;
  }
  const size_t mk = A.mark();
  bf16_t* g = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(g);
  CTTA_TRY(vgn(c, V->norm, x, g, N, false, st));
  if (!q) { q = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(q); }
  if (!k) { k = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(k); }
  CTTA_TRY(run_conv2d(c, V->q, g, B, H, W, false, q, nullptr, 0, nullptr, 0));
  CTTA_TRY(run_conv2d(c, V->k, g, B, H, W, false, k, nullptr, 0, nullptr, 0));
  if (!vt) { vt = A.get<bf16_t>((size_t)B * C * N); ALLOC_OR_FAIL(vt); }
  {
    ctta_conv_desc d;
    desc_init(&d);
    d.x0 = V->v.p.w; d.c0 = V->v.p.k_pad;
    d.batch = 1; d.hi = C; d.wi = 1; d.ho = C; d.wo = 1;
    d.w = g; d.k_pad = C; d.n = N;
    d.bias_m = V->v.p.bias;
    d.out = vt; d.ldc = N;
    d.groups = B; d.w_group_stride = (int64_t)N * C; d.out_group_stride = (int64_t)C * N;
    RUN(c, ctta_conv_gemm(&d, c.stream));
  }
  // scores / probabilities in sample chunks small enough to stay in the 256 MB Infinity Cache between the three
  // launches (QK^T -> softmax -> PV): the (N x N) fp32 score matrix of one sample is 64 MB at N = 4096, the whole batch
  // at B = 32 would be 2 GiB of arena and 6 GB of HBM traffic per decode (modules.py:204-230 materialises it too)
  const int chunk_mb = 1 << 20;   // the whole batch in one go (measured fastest; 288 GB of HBM hold the 2 GiB slab of batch 32)
  int gb = (int)(((size_t)chunk_mb << 20) / ((size_t)N * N * 6));
  if (gb < 1) gb = 1;
  if (gb > B) gb = B;
  float* s = A.get<float>((size_t)gb * N * N); ALLOC_OR_FAIL(s);
  if (!p) { p = A.get<bf16_t>((size_t)gb * N * N); ALLOC_OR_FAIL(p); }
  const bool p_full = c.grad;   // the differentiable forward keeps every sample's probabilities
  bf16_t* o = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(o);
  for (int b0 = 0; b0 < B; b0 += gb) {
    const int nb = B - b0 < gb ? B - b0 : gb;
    bf16_t* pc = p_full ? p + (size_t)b0 * N * N : p;
    {
      ctta_conv_desc d;
      desc_init(&d);
      d.x0 = q + (size_t)b0 * N * C; d.c0 = C;
      d.batch = 1; d.hi = N; d.wi = 1; d.ho = N; d.wo = 1;
      d.w = k + (size_t)b0 * N * C; d.k_pad = C; d.n = N;
      d.out = s; d.ldc = N; d.out_f32 = 1;
      d.groups = nb; d.x_group_stride = (int64_t)N * C; d.w_group_stride = (int64_t)N * C;
      d.out_group_stride = (int64_t)N * N;
      RUN(c, ctta_conv_gemm(&d, c.stream));
    }
    RUN(c, ctta_softmax_rows(s, pc, (int64_t)nb * N, N, 1.0f / sqrtf((float)C), c.stream));
    {
      ctta_conv_desc d;
      desc_init(&d);
      d.x0 = pc; d.c0 = N;
      d.batch = 1; d.hi = N; d.wi = 1; d.ho = N; d.wo = 1;
      d.w = vt + (size_t)b0 * C * N; d.k_pad = N; d.n = C;
      d.out = o + (size_t)b0 * N * C; d.ldc = C;
      d.groups = nb; d.x_group_stride = (int64_t)N * N; d.w_group_stride = (int64_t)C * N;
      d.out_group_stride = (int64_t)N * C;
      RUN(c, ctta_conv_gemm(&d, c.stream));
    }
  }
  CTTA_TRY(run_conv2d(c, V->proj, o, B, H, W, false, out, nullptr, 0, x, C));
  A.release(mk);
  *out_p = out;
  return CTTA_OK;
}

static ctta_status vae_forward_impl(ctta_vae* V, bool dry, const float* z, int B, float* mel, hipStream_t stream,
                                    size_t* gn_need, bool grad = false) {
  const ctta_vae_config& cfg = V->cfg;
  VCtx c;
  c.arena = &V->arena; c.stream = stream; c.dry = dry;
  c.taps = cfg.debug_taps ? &V->taps : nullptr;
  c.gn_scratch = V->gn_scratch; c.gn_scratch_floats = V->gn_scratch_floats;
  if (V->gn_scratch && gn_fuse_enabled()) { c.gn_fpart = V->gn_scratch + V->gn_scratch_floats; c.gn_fpart_floats = V->gn_fpart_floats; }
  c.gn_groups = 32;
  c.B = B;
  c.grad = grad;
  V->sv.B = grad ? B : 0;
  Arena& A = V->arena;
  A.reset();
  int H = cfg.latent_h, W = cfg.latent_w;
  const int cpad = 32;
  bf16_t* zin = A.get<bf16_t>((size_t)B * H * W * cpad); ALLOC_OR_FAIL(zin);
  if (!dry) {
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(post_quant_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, z, V->pq_w,
                       V->pq_b, 1.0f / cfg.scale_factor, B, cfg.embed_dim, cfg.z_channels, H * W, cpad, zin);
    CTTA_LAUNCH_CHECK();
  }
  int ch = V->block_in;
  bf16_t* h = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(h);
  CTTA_TRY(run_conv2d(c, V->conv_in, zin, B, H, W, false, h, nullptr, 0, nullptr, 0));
  add_tap(c, "conv_in", h, B, ch, H, W, ch);
  CTTA_TRY(run_vae_res(c, V->mid1, h, H, W, &h));
  add_tap(c, "mid.block_1", h, B, ch, H, W, ch);
  CTTA_TRY(run_vae_attn(c, &V->attn, h, H, W, &h));
  add_tap(c, "mid.attn_1", h, B, ch, H, W, ch);
  CTTA_TRY(run_vae_res(c, V->mid2, h, H, W, &h));
  add_tap(c, "mid.block_2", h, B, ch, H, W, ch);
  for (int lvl = cfg.n_levels - 1; lvl >= 0; --lvl) {
    for (size_t b = 0; b < V->up[lvl].size(); ++b) {
      CTTA_TRY(run_vae_res(c, V->up[lvl][b], h, H, W, &h));
      ch = V->up[lvl][b].cout;
      add_tap(c, "up." + std::to_string(lvl) + ".block." + std::to_string(b), h, B, ch, H, W, ch);
    }
    if (lvl != 0) {
      bf16_t* u = A.get<bf16_t>((size_t)B * 4 * H * W * ch); ALLOC_OR_FAIL(u);
      CTTA_TRY(run_conv2d(c, V->upsample[lvl], h, B, H, W, true, u, nullptr, 0, nullptr, 0));
      h = u; H *= 2; W *= 2;
      add_tap(c, "up." + std::to_string(lvl) + ".upsample", h, B, ch, H, W, ch);
    }
  }
  if (grad) { V->sv.h_last = h; V->sv.st_out = stats_buf(c); ALLOC_OR_FAIL(V->sv.st_out); }
  bf16_t* a = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(a);
  CTTA_TRY(vgn(c, V->norm_out, h, a, H * W, true, grad ? V->sv.st_out : nullptr));
  {  // conv_out (Cout = out_ch = 1): MFMA kernel with element-wise fp32 stores; NCHW with C=1 == [m]
    ctta_conv_desc d;
    desc_init(&d);
    d.x0 = a; d.c0 = ch;
    d.batch = B; d.hi = H; d.wi = W; d.ho = H; d.wo = W;
    d.kh = 3; d.kw = 3; d.pad_h = d.pad_w = 1;
    d.w = V->conv_out.p.w; d.k_pad = V->conv_out.p.k_pad; d.n = cfg.out_ch; d.bias = V->conv_out.p.bias;
    d.out = mel; d.ldc = cfg.out_ch; d.out_f32 = 1;
    RUN(c, ctta_conv_gemm(&d, c.stream));
  }
  if (gn_need) *gn_need = c.gn_need;
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ decoder input gradient
// d mel / d z for a frozen decoder (CLAPLoss back-propagates the waveform loss into the student's latent,
// tools/losses.py:294-298): the differentiable forward's saved tensors are replayed in reverse with data-gradient
// convolutions (conv_gemm against the rotated / transposed packs) and the GroupNorm / softmax backward kernels.
__global__ void post_quant_bwd_kernel(const bf16_t* __restrict__ dzin, int ld, const float* __restrict__ w,
                                      float inv_scale, int B, int zin, int zout, int HW, float* __restrict__ gz) {
  const long long total = (long long)B * HW;
  const long long pix = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (pix >= total) return;
  const int bb = (int)(pix / HW);
  const int hw = (int)(pix - (long long)bb * HW);
  float d[32];
  for (int o = 0; o < zout; ++o) d[o] = bf2f(dzin[(size_t)pix * ld + o]);
  for (int e = 0; e < zin; ++e) {
    float acc = 0.f;
    for (int o = 0; o < zout; ++o) acc += w[o * zin + e] * d[o];
    gz[((size_t)bb * zin + e) * HW + hw] = acc * inv_scale;
  }
}

static ctta_status vconv_dgrad(VCtx& c, const ConvLayer& D, const bf16_t* dy, int H, int W, bf16_t* dx, bool accumulate) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = dy; d.c0 = D.cin_pad;
  d.batch = c.B; d.hi = H; d.wi = W; d.ho = H; d.wo = W;
  d.kh = D.kh; d.kw = D.kw; d.pad_h = d.pad_w = D.pad;
  d.w = D.p.w; d.k_pad = D.p.k_pad; d.n = D.p.n;
  d.out = dx; d.ldc = D.cout; d.accumulate = accumulate ? 1 : 0;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

static ctta_status vgn_bwd(VCtx& c, const GNLayer& g, const bf16_t* x, const bf16_t* dy, bf16_t* dx, int hw,
                           const float* stats, bool silu, bool accumulate_dx) {
  const size_t need = ctta_groupnorm_bwd_scratch_floats(c.B, hw, g.c, kVaeGroups);
  if (need > c.gn_need) c.gn_need = need;
  if (!c.dry && need > c.gn_scratch_floats) { ctta_set_error("groupnorm backward scratch too small"); return CTTA_ERR_INVALID; }
  RUN(c, ctta_groupnorm_bwd(x, dy, dx, c.B, hw, g.c, kVaeGroups, stats, g.gamma, g.beta, silu ? 1 : 0,
                            accumulate_dx ? 1 : 0, nullptr, nullptr, 0, c.gn_scratch, c.stream));
  return CTTA_OK;
}

// out = conv2(silu(gn2(t1))) + shortcut(x), t1 = conv1(silu(gn1(x)))   (ResnetBlock.forward modules.py:155-175)
static ctta_status bwd_vae_res(VCtx& c, const VaeRes& R, const bf16_t* dout, bf16_t** dx_p) {
  Arena& A = *c.arena;
  const int H = R.sv.H, W = R.sv.W;
  const size_t M = (size_t)c.B * H * W;
  bf16_t* dx = A.get<bf16_t>(M * R.cin); ALLOC_OR_FAIL(dx);
  const size_t mk = A.mark();
  bf16_t* da2 = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(da2);
  CTTA_TRY(vconv_dgrad(c, R.d2, dout, H, W, da2, false));
  bf16_t* dt1 = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(dt1);
  CTTA_TRY(vgn_bwd(c, R.n2, R.sv.t1, da2, dt1, H * W, R.sv.st2, true, false));
  bf16_t* da = A.get<bf16_t>(M * R.cin); ALLOC_OR_FAIL(da);
  CTTA_TRY(vconv_dgrad(c, R.d1, dt1, H, W, da, false));
  if (R.has_sc) CTTA_TRY(vconv_dgrad(c, R.dsc, dout, H, W, dx, false));
  else RUN(c, ctta_add_slices(dout, R.cout, nullptr, 0, dx, R.cin, (int64_t)M, R.cin, c.stream));
  CTTA_TRY(vgn_bwd(c, R.n1, R.sv.x, da, dx, H * W, R.sv.st1, true, true));
  A.release(mk);
  *dx_p = dx;
  return CTTA_OK;
}

// batched GEMM helper: out[g][m][n] = sum_k X[g][m][k] * Wt[g][n][k]
static ctta_status bgemm(VCtx& c, const bf16_t* x, int64_t xs, const bf16_t* w, int64_t wst, int Mr, int Nn, int K,
                         void* out, int64_t os, int ldc, bool f32) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = x; d.c0 = K;
  d.batch = 1; d.hi = Mr; d.wi = 1; d.ho = Mr; d.wo = 1;
  d.w = w; d.k_pad = K; d.n = Nn;
  d.out = out; d.ldc = ldc; d.out_f32 = f32 ? 1 : 0;
  d.groups = c.B; d.x_group_stride = xs; d.w_group_stride = wst; d.out_group_stride = os;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

// AttnBlock.forward (modules.py:204-230) in reverse.  With P = softmax(q k^T / sqrt(C)) saved:
//   dO = proj^T dout;  dP = dO V^T;  dS = P (dP - rowsum(P dP)) / sqrt(C);  dq = dS k;  dk = dS^T q;  dV = P^T dO;
//   d norm(x) = Wq^T dq + Wk^T dk + Wv^T dV;  dx = dout + GroupNorm backward.
static ctta_status bwd_vae_attn(VCtx& c, const VaeAttn& T, const bf16_t* dout, bf16_t** dx_p) {
  Arena& A = *c.arena;
  const int H = T.sv.H, W = T.sv.W, N = H * W, C = T.C, B = c.B;
  const size_t M = (size_t)B * N;
  const int64_t NC = (int64_t)N * C, NN = (int64_t)N * N;
  bf16_t* dx = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(dx);
  const size_t mk = A.mark();
  bf16_t* dO = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(dO);
  CTTA_TRY(vconv_dgrad(c, T.dproj, dout, H, W, dO, false));
  bf16_t* vn = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(vn);     // V   [B][N][C]
  bf16_t* kt = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(kt);     // K^T [B][C][N]
  bf16_t* qt = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(qt);     // Q^T
  bf16_t* dOt = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(dOt);   // dO^T
  RUN(c, ctta_transpose_bf16(T.sv.vt, NC, C, N, N, 0, vn, NC, C, B, c.stream));
  RUN(c, ctta_transpose_bf16(T.sv.k, NC, N, C, C, 0, kt, NC, N, B, c.stream));
  RUN(c, ctta_transpose_bf16(T.sv.q, NC, N, C, C, 0, qt, NC, N, B, c.stream));
  RUN(c, ctta_transpose_bf16(dO, NC, N, C, C, 0, dOt, NC, N, B, c.stream));
  float* dP = A.get<float>((size_t)B * NN); ALLOC_OR_FAIL(dP);
  CTTA_TRY(bgemm(c, dO, NC, vn, NC, N, N, C, dP, NN, N, true));
  bf16_t* dS = A.get<bf16_t>((size_t)B * NN); ALLOC_OR_FAIL(dS);
  RUN(c, ctta_softmax_bwd_rows(T.sv.p, dP, N, dS, (int64_t)B * N, N, N, 1.0f / sqrtf((float)C), c.stream));
  bf16_t* dq = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(dq);
  bf16_t* dk = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(dk);
  bf16_t* dv = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(dv);
  CTTA_TRY(bgemm(c, dS, NN, kt, NC, N, C, N, dq, NC, C, false));
  bf16_t* tt = reinterpret_cast<bf16_t*>(dP);               // dP is dead: its space holds dS^T, then P^T
  RUN(c, ctta_transpose_bf16(dS, NN, N, N, N, 0, tt, NN, N, B, c.stream));
  CTTA_TRY(bgemm(c, tt, NN, qt, NC, N, C, N, dk, NC, C, false));
  RUN(c, ctta_transpose_bf16(T.sv.p, NN, N, N, N, 0, tt, NN, N, B, c.stream));
  CTTA_TRY(bgemm(c, tt, NN, dOt, NC, N, C, N, dv, NC, C, false));
  bf16_t* dg = A.get<bf16_t>(M * C); ALLOC_OR_FAIL(dg);
  CTTA_TRY(vconv_dgrad(c, T.dq, dq, H, W, dg, false));
  CTTA_TRY(vconv_dgrad(c, T.dk, dk, H, W, dg, true));
  CTTA_TRY(vconv_dgrad(c, T.dv, dv, H, W, dg, true));
  RUN(c, ctta_add_slices(dout, C, nullptr, 0, dx, C, (int64_t)M, C, c.stream));
  CTTA_TRY(vgn_bwd(c, T.norm, T.sv.x, dg, dx, N, T.sv.st, false, true));
  A.release(mk);
  *dx_p = dx;
  return CTTA_OK;
}

static ctta_status vae_backward_impl(ctta_vae* V, bool dry, const float* gmel, int B, float* gz, hipStream_t stream,
                                     size_t* gn_need) {
  const ctta_vae_config& cfg = V->cfg;
  VCtx c;
  c.arena = &V->arena; c.stream = stream; c.dry = dry;
  c.taps = nullptr;
  c.gn_scratch = V->gn_scratch; c.gn_scratch_floats = V->gn_scratch_floats;
  if (V->gn_scratch && gn_fuse_enabled()) { c.gn_fpart = V->gn_scratch + V->gn_scratch_floats; c.gn_fpart_floats = V->gn_fpart_floats; }
  c.gn_groups = 32;
  c.B = B;
  Arena& A = V->arena;   // continues above the differentiable forward's saved tensors
  const int up = 1 << (cfg.n_levels - 1);
  int H = cfg.latent_h * up, W = cfg.latent_w * up, ch = V->c_last;
  bf16_t* da = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(da);
  RUN(c, ctta_conv_cout1_dgrad(gmel, nullptr, V->out_w, B, H, W, 3, 3, 1, 1, ch, nullptr, 0.f, da, stream));
  bf16_t* dh = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(dh);
  CTTA_TRY(vgn_bwd(c, V->norm_out, V->sv.h_last, da, dh, H * W, V->sv.st_out, true, false));
  for (int lvl = 0; lvl < cfg.n_levels; ++lvl) {
    if (lvl != 0) {   // u = conv(nearest x2 (h)): data gradient at the fine resolution, then the 2x2 sum
      bf16_t* du = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(du);
      CTTA_TRY(vconv_dgrad(c, V->d_upsample[lvl], dh, H, W, du, false));
      H /= 2; W /= 2;
      bf16_t* dl = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(dl);
      RUN(c, ctta_pool2_sum(du, dl, B, H, W, ch, 0, stream));
      dh = dl;
    }
    for (int b = (int)V->up[lvl].size() - 1; b >= 0; --b) {
      CTTA_TRY(bwd_vae_res(c, V->up[lvl][b], dh, &dh));
      ch = V->up[lvl][b].cin;
    }
  }
  CTTA_TRY(bwd_vae_res(c, V->mid2, dh, &dh));
  CTTA_TRY(bwd_vae_attn(c, V->attn, dh, &dh));
  CTTA_TRY(bwd_vae_res(c, V->mid1, dh, &dh));
  const int zc = V->d_conv_in.cout;
  bf16_t* dzin = A.get<bf16_t>((size_t)B * H * W * zc); ALLOC_OR_FAIL(dzin);
  CTTA_TRY(vconv_dgrad(c, V->d_conv_in, dh, H, W, dzin, false));
  if (!dry) {
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(post_quant_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dzin, zc,
                       V->pq_w, 1.0f / cfg.scale_factor, B, cfg.embed_dim, cfg.z_channels, H * W, gz);
    CTTA_LAUNCH_CHECK();
  }
  if (gn_need && c.gn_need > *gn_need) *gn_need = c.gn_need;
  return CTTA_OK;
}

static ctta_status vae_build(ctta_vae* V) {
  const ctta_vae_config& cfg = V->cfg;
  WeightStore& ws = V->store;
  const int nres = cfg.n_levels;
  int block_in = cfg.ch * cfg.ch_mult[nres - 1];
  V->block_in = block_in;
  CTTA_TRY(ws.add_vector("post_quant_conv.weight", cfg.z_channels * cfg.embed_dim, &V->pq_w));
  CTTA_TRY(ws.add_vector("post_quant_conv.bias", cfg.z_channels, &V->pq_b));
  const std::string p = "decoder.";
  const bool grad = cfg.enable_grad != 0;
  CTTA_TRY(make_conv(ws, p + "conv_in.", block_in, cfg.z_channels, 32, 3, 3, 1, 1, &V->conv_in));
  if (grad) CTTA_TRY(make_conv_dgrad_from(ws, V->conv_in, block_in, cfg.z_channels, &V->d_conv_in));
  CTTA_TRY(make_vae_res(ws, p + "mid.block_1.", block_in, block_in, &V->mid1, grad));
  CTTA_TRY(make_vae_attn(ws, p + "mid.attn_1.", block_in, &V->attn, grad));
  CTTA_TRY(make_vae_res(ws, p + "mid.block_2.", block_in, block_in, &V->mid2, grad));
  V->up.resize(nres);
  V->upsample.resize(nres);
  V->d_upsample.resize(nres);
  for (int lvl = nres - 1; lvl >= 0; --lvl) {
    const int block_out = cfg.ch * cfg.ch_mult[lvl];
    V->up[lvl].resize(cfg.num_res_blocks + 1);
    for (int b = 0; b <= cfg.num_res_blocks; ++b) {
      CTTA_TRY(make_vae_res(ws, p + "up." + std::to_string(lvl) + ".block." + std::to_string(b) + ".", block_in,
                            block_out, &V->up[lvl][b], grad));
      block_in = block_out;
    }
    if (lvl != 0) {
      CTTA_TRY(make_conv(ws, p + "up." + std::to_string(lvl) + ".upsample.conv.", block_in, block_in, block_in, 3, 3, 1,
                         1, &V->upsample[lvl]));
      if (grad) CTTA_TRY(make_conv_dgrad_from(ws, V->upsample[lvl], block_in, block_in, &V->d_upsample[lvl]));
    }
  }
  V->c_last = block_in;
  CTTA_TRY(make_gn(ws, p + "norm_out.", block_in, &V->norm_out));
  CTTA_TRY(make_conv(ws, p + "conv_out.", cfg.out_ch, block_in, block_in, 3, 3, 1, 1, &V->conv_out));
  if (grad) CTTA_TRY(add_small_conv_w(ws, p + "conv_out.weight", 1, block_in, 9, &V->out_w));
  return CTTA_OK;
}

extern "C" ctta_status ctta_vae_create(const ctta_vae_config* cfg, const ctta_tensor* weights, int n_weights,
                                       void* stream, ctta_vae** out) {
  CTTA_REQUIRE(cfg && weights && out, "vae_create: null pointer");
  CTTA_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= CTTA_MAX_LEVELS, "vae_create: n_levels=%d", cfg->n_levels);
  CTTA_REQUIRE(cfg->embed_dim <= 16 && cfg->z_channels <= 32, "vae_create: z_channels/embed_dim too large");
  CTTA_REQUIRE(cfg->out_ch == 1, "vae_create: out_ch=%d (only the 1-channel mel decoder is built)", cfg->out_ch);
  CTTA_REQUIRE(cfg->ch % 32 == 0, "vae_create: ch=%d must be a multiple of 32 (GroupNorm(32))", cfg->ch);
  CTTA_REQUIRE(cfg->scale_factor != 0.f, "vae_create: scale_factor is zero");
  hipStream_t s = (hipStream_t)stream;
  ctta_vae* V = new ctta_vae();
  V->cfg = *cfg;
  CTTA_REQUIRE(!cfg->enable_grad || cfg->z_channels % 8 == 0, "vae_create: enable_grad needs z_channels %% 8 == 0");
  ctta_status st = V->store.init(cfg->enable_grad ? estimate_store_bytes_training(weights, n_weights)
                                                  : estimate_store_bytes(weights, n_weights));
  if (st != CTTA_OK) { delete V; return st; }
  WeightTable wt;
  wt.build(weights, n_weights);
  st = vae_build(V);
  if (st == CTTA_OK) st = V->store.run_all(wt, s);
  size_t gn_need = 0;
  if (st == CTTA_OK) {
    V->arena.dry = true;
    V->arena.no_release = cfg->debug_taps != 0;
    st = vae_forward_impl(V, true, nullptr, cfg->max_batch, nullptr, s, &gn_need);
    if (st == CTTA_OK && cfg->enable_grad) {   // the differentiable forward keeps more, and the backward runs on top
      st = vae_forward_impl(V, true, nullptr, cfg->max_batch, nullptr, s, &gn_need, true);
      if (st == CTTA_OK) st = vae_backward_impl(V, true, nullptr, cfg->max_batch, nullptr, s, &gn_need);
    }
  }
  if (st == CTTA_OK) {
    const size_t bytes = V->arena.peak + (1 << 20);
    V->arena.dry = false;
    V->arena.cap = bytes;
    if (hipMalloc((void**)&V->arena.base, bytes) != hipSuccess ||
        hipMalloc((void**)&V->gn_scratch, (gn_need + 64 + (V->gn_fpart_floats = (size_t)cfg->max_batch *
                                                            ((size_t)cfg->latent_h * cfg->latent_w * 16 / 16 + 1) * 32 * 2)) * sizeof(float)) != hipSuccess) {
      ctta_set_error("vae_create: hipMalloc of %zu-byte activation arena failed", bytes);
      st = CTTA_ERR_NOMEM;
    } else {
      V->gn_scratch_floats = gn_need + 64;
      st = V->splitws.init();
    }
  }
  if (st == CTTA_OK && hipStreamSynchronize(s) != hipSuccess) { ctta_set_error("vae_create: stream sync failed"); st = CTTA_ERR_HIP; }
  if (st != CTTA_OK) { ctta_vae_destroy(V); return st; }
  *out = V;
  return CTTA_OK;
}

extern "C" void ctta_vae_destroy(ctta_vae* V) {
  if (!V) return;
  V->store.destroy();
  if (V->arena.base) (void)hipFree(V->arena.base);
  if (V->gn_scratch) (void)hipFree(V->gn_scratch);
  V->splitws.destroy();
  delete V;
}

extern "C" ctta_status ctta_vae_decode(ctta_vae* V, const float* z, int batch, float* mel, void* stream) {
  CTTA_REQUIRE(V && z && mel, "vae_decode: null pointer");
  CTTA_REQUIRE(batch >= 1 && batch <= V->cfg.max_batch, "vae_decode: batch %d outside [1,%d]", batch, V->cfg.max_batch);
  WsBind bind(V->splitws);
  return vae_forward_impl(V, false, z, batch, mel, (hipStream_t)stream, nullptr);
}

extern "C" ctta_status ctta_vae_decode_with_grad(ctta_vae* V, const float* z, int batch, float* mel, void* stream) {
  CTTA_REQUIRE(V && z && mel, "vae_decode_with_grad: null pointer");
  CTTA_REQUIRE(V->cfg.enable_grad, "vae_decode_with_grad: the handle was created without enable_grad");
  CTTA_REQUIRE(batch >= 1 && batch <= V->cfg.max_batch, "vae_decode_with_grad: batch %d outside [1,%d]", batch,
               V->cfg.max_batch);
  WsBind bind(V->splitws);
  return vae_forward_impl(V, false, z, batch, mel, (hipStream_t)stream, nullptr, true);
}
extern "C" ctta_status ctta_vae_decode_backward(ctta_vae* V, const float* grad_mel, int batch, float* grad_z, void* stream) {
  CTTA_REQUIRE(V && grad_mel && grad_z, "vae_decode_backward: null pointer");
  CTTA_REQUIRE(V->cfg.enable_grad && V->sv.B == batch,
               "vae_decode_backward: no differentiable forward of batch %d is pending on this handle", batch);
  WsBind bind(V->splitws);
  const ctta_status st = vae_backward_impl(V, false, grad_mel, batch, grad_z, (hipStream_t)stream, nullptr);
  V->sv.B = 0;   // the saved tensors are consumed (the backward's temporaries overwrote the arena above them)
  return st;
}

extern "C" size_t ctta_vae_arena_bytes(const ctta_vae* V) { return V ? V->arena.cap + V->store.arena.cap : 0; }
extern "C" int ctta_vae_num_taps(const ctta_vae* V) { return V ? (int)V->taps.size() : 0; }
extern "C" ctta_status ctta_vae_tap_info(const ctta_vae* V, int i, const char** name, int dims[4]) {
  CTTA_REQUIRE(V && i >= 0 && i < (int)V->taps.size(), "tap index out of range");
  const Tap& t = V->taps[i];
  *name = t.name.c_str();
  dims[0] = t.b; dims[1] = t.c; dims[2] = t.h; dims[3] = t.w;
  return CTTA_OK;
}
extern "C" ctta_status ctta_vae_tap_read(ctta_vae* V, int i, float* dst, void* stream) {
  CTTA_REQUIRE(V && i >= 0 && i < (int)V->taps.size() && dst, "tap index out of range");
  const Tap& t = V->taps[i];
  CTTA_REQUIRE(t.ptr, "tap '%s' has not been produced yet", t.name.c_str());
  return ctta_nhwc_bf16_to_nchw_f32(t.ptr, dst, t.b, t.c, t.h, t.w, t.c_stride, stream);
}

// ====================================================================================== VAE encoder
// AutoencoderKL.encode (autoencoder.py:80-85): Encoder.forward (modules.py:519-543) + quant_conv, the training-side
// latent encoder (tools/train_utils.py:155-162 encodes every batch).  Same blocks as the decoder; Downsample
// (modules.py:87-92) = zero-pad right/bottom by one + 3x3 stride-2 conv, which is conv_gemm with pad 0 and the
// output extent H/2 x W/2 (the out-of-image tap reads zeros).
struct ctta_vae_encoder {
  ctta_vae_config cfg;
  WeightStore store;
  Arena arena;
  std::vector<Tap> taps;
  ConvLayer conv_in;
  std::vector<std::vector<VaeRes>> down;   // [level][block]
  std::vector<ConvLayer> downsample;       // [level] (last level unused)
  VaeRes mid1, mid2;
  VaeAttn attn;
  GNLayer norm_out;
  ConvLayer conv_out;
  float *q_w = nullptr, *q_b = nullptr;    // quant_conv (1x1) in fp32
  int block_in = 0, zc2 = 0;
  float* gn_scratch = nullptr;
  size_t gn_scratch_floats = 0;
  size_t gn_fpart_floats = 0;
  SplitWs splitws;
};

// moments[b][o][hw] = quant_conv(h)[o] per pixel, h = conv_out result [M][zc2] bf16 -> NCHW fp32
__global__ void quant_moments_kernel(const bf16_t* __restrict__ h, const float* __restrict__ w, const float* __restrict__ b,
                                     int B, int cin, int cout, int HW, float* __restrict__ out) {
  const long long pix = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (pix >= (long long)B * HW) return;
  const int bb = (int)(pix / HW), hw = (int)(pix - (long long)bb * HW);
  float in[32];
  for (int c = 0; c < cin; ++c) in[c] = bf2f(h[(size_t)pix * cin + c]);
  for (int o = 0; o < cout; ++o) {
    float acc = b[o];
    for (int c = 0; c < cin; ++c) acc += w[o * cin + c] * in[c];
    out[((size_t)bb * cout + o) * HW + hw] = acc;
  }
}

static ctta_status vae_encode_impl(ctta_vae_encoder* E, bool dry, const float* mel, int B, float* moments,
                                   hipStream_t stream, size_t* gn_need) {
  const ctta_vae_config& cfg = E->cfg;
  VCtx c;
  c.arena = &E->arena; c.stream = stream; c.dry = dry;
  c.taps = cfg.debug_taps ? &E->taps : nullptr;
  c.gn_scratch = E->gn_scratch; c.gn_scratch_floats = E->gn_scratch_floats;
  if (E->gn_scratch && gn_fuse_enabled()) { c.gn_fpart = E->gn_scratch + E->gn_scratch_floats; c.gn_fpart_floats = E->gn_fpart_floats; }
  c.gn_groups = 32;
  c.B = B;
  Arena& A = E->arena;
  A.reset();
  const int scale = 1 << (cfg.n_levels - 1);
  int H = cfg.latent_h * scale, W = cfg.latent_w * scale;
  const int cpad = 8;
  bf16_t* xin = A.get<bf16_t>((size_t)B * H * W * cpad); ALLOC_OR_FAIL(xin);
  RUN(c, ctta_nchw_f32_to_nhwc_bf16(mel, xin, B, 1, H, W, cpad, 1.0f, stream));
  int ch = cfg.ch;
  bf16_t* h = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(h);
  CTTA_TRY(run_conv2d(c, E->conv_in, xin, B, H, W, false, h, nullptr, 0, nullptr, 0));
  add_tap(c, "conv_in", h, B, ch, H, W, ch);
  for (int lvl = 0; lvl < cfg.n_levels; ++lvl) {
    for (size_t b = 0; b < E->down[lvl].size(); ++b) {
      CTTA_TRY(run_vae_res(c, E->down[lvl][b], h, H, W, &h));
      ch = E->down[lvl][b].cout;
      add_tap(c, "down." + std::to_string(lvl) + ".block." + std::to_string(b), h, B, ch, H, W, ch);
    }
    if (lvl != cfg.n_levels - 1) {
      const int ho = H / 2, wo = W / 2;
      bf16_t* d_ = A.get<bf16_t>((size_t)B * ho * wo * ch); ALLOC_OR_FAIL(d_);
      const ConvLayer& L = E->downsample[lvl];
      ctta_conv_desc d;
      desc_init(&d);
      d.x0 = h; d.c0 = L.cin_pad;
      d.batch = B; d.hi = H; d.wi = W; d.ho = ho; d.wo = wo;
      d.kh = 3; d.kw = 3; d.stride_h = d.stride_w = 2; d.pad_h = d.pad_w = 0;
      d.w = L.p.w; d.k_pad = L.p.k_pad; d.n = L.p.n; d.bias = L.p.bias;
      d.out = d_; d.ldc = L.p.n;
      RUN(c, ctta_conv_gemm(&d, c.stream));
      h = d_; H = ho; W = wo;
      add_tap(c, "down." + std::to_string(lvl) + ".downsample", h, B, ch, H, W, ch);
    }
  }
  CTTA_TRY(run_vae_res(c, E->mid1, h, H, W, &h));
  CTTA_TRY(run_vae_attn(c, &E->attn, h, H, W, &h));
  CTTA_TRY(run_vae_res(c, E->mid2, h, H, W, &h));
  add_tap(c, "mid.block_2", h, B, ch, H, W, ch);
  bf16_t* a = A.get<bf16_t>((size_t)B * H * W * ch); ALLOC_OR_FAIL(a);
  CTTA_TRY(vgn(c, E->norm_out, h, a, H * W, true));
  bf16_t* co = A.get<bf16_t>((size_t)B * H * W * E->zc2); ALLOC_OR_FAIL(co);
  CTTA_TRY(run_conv2d(c, E->conv_out, a, B, H, W, false, co, nullptr, 0, nullptr, 0));
  if (!dry) {
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(quant_moments_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, co, E->q_w,
                       E->q_b, B, E->zc2, 2 * cfg.embed_dim, H * W, moments);
    CTTA_LAUNCH_CHECK();
  }
  if (gn_need) *gn_need = c.gn_need;
  return CTTA_OK;
}

static ctta_status vae_encoder_build(ctta_vae_encoder* E) {
  const ctta_vae_config& cfg = E->cfg;
  WeightStore& ws = E->store;
  const int nres = cfg.n_levels;
  const std::string p = "encoder.";
  CTTA_TRY(make_conv(ws, p + "conv_in.", cfg.ch, 1, 8, 3, 3, 1, 1, &E->conv_in));
  int block_in = cfg.ch;
  E->down.resize(nres);
  E->downsample.resize(nres);
  for (int lvl = 0; lvl < nres; ++lvl) {
    const int block_out = cfg.ch * cfg.ch_mult[lvl];
    E->down[lvl].resize(cfg.num_res_blocks);
    for (int b = 0; b < cfg.num_res_blocks; ++b) {
      CTTA_TRY(make_vae_res(ws, p + "down." + std::to_string(lvl) + ".block." + std::to_string(b) + ".", block_in,
                            block_out, &E->down[lvl][b]));
      block_in = block_out;
    }
    if (lvl != nres - 1)
      CTTA_TRY(make_conv(ws, p + "down." + std::to_string(lvl) + ".downsample.conv.", block_in, block_in, block_in, 3, 3,
                         2, 0, &E->downsample[lvl]));
  }
  E->block_in = block_in;
  CTTA_TRY(make_vae_res(ws, p + "mid.block_1.", block_in, block_in, &E->mid1));
  CTTA_TRY(make_vae_attn(ws, p + "mid.attn_1.", block_in, &E->attn));
  CTTA_TRY(make_vae_res(ws, p + "mid.block_2.", block_in, block_in, &E->mid2));
  CTTA_TRY(make_gn(ws, p + "norm_out.", block_in, &E->norm_out));
  E->zc2 = 2 * cfg.z_channels;
  CTTA_TRY(make_conv(ws, p + "conv_out.", E->zc2, block_in, block_in, 3, 3, 1, 1, &E->conv_out));
  CTTA_TRY(ws.add_vector("quant_conv.weight", 2 * cfg.embed_dim * E->zc2, &E->q_w));
  CTTA_TRY(ws.add_vector("quant_conv.bias", 2 * cfg.embed_dim, &E->q_b));
  return CTTA_OK;
}

extern "C" void ctta_vae_encoder_destroy(ctta_vae_encoder* E) {
  if (!E) return;
  E->store.destroy();
  if (E->arena.base) (void)hipFree(E->arena.base);
  if (E->gn_scratch) (void)hipFree(E->gn_scratch);
  E->splitws.destroy();
  delete E;
}

extern "C" ctta_status ctta_vae_encoder_create(const ctta_vae_config* cfg, const ctta_tensor* weights, int n_weights,
                                               void* stream, ctta_vae_encoder** out) {
  CTTA_REQUIRE(cfg && weights && out, "vae_encoder_create: null pointer");
  CTTA_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= CTTA_MAX_LEVELS, "vae_encoder_create: n_levels=%d", cfg->n_levels);
  CTTA_REQUIRE(cfg->z_channels * 2 <= 32 && cfg->z_channels * 2 % 4 == 0 && cfg->embed_dim * 2 <= 32,
               "vae_encoder_create: z_channels/embed_dim unsupported");
  CTTA_REQUIRE(cfg->out_ch == 1, "vae_encoder_create: in_channels=%d (only the 1-channel mel encoder is built)", cfg->out_ch);
  CTTA_REQUIRE(cfg->ch % 32 == 0, "vae_encoder_create: ch=%d must be a multiple of 32 (GroupNorm(32))", cfg->ch);
  hipStream_t s = (hipStream_t)stream;
  ctta_vae_encoder* E = new ctta_vae_encoder();
  E->cfg = *cfg;
  ctta_status st = E->store.init(estimate_store_bytes(weights, n_weights));
  if (st != CTTA_OK) { delete E; return st; }
  WeightTable wt;
  wt.build(weights, n_weights);
  st = vae_encoder_build(E);
  if (st == CTTA_OK) st = E->store.run_all(wt, s);
  size_t gn_need = 0;
  if (st == CTTA_OK) {
    E->arena.dry = true;
    E->arena.no_release = cfg->debug_taps != 0;
    st = vae_encode_impl(E, true, nullptr, cfg->max_batch, nullptr, s, &gn_need);
  }
  if (st == CTTA_OK) {
    const size_t bytes = E->arena.peak + (1 << 20);
    E->arena.dry = false;
    E->arena.cap = bytes;
    if (hipMalloc((void**)&E->arena.base, bytes) != hipSuccess ||
        hipMalloc((void**)&E->gn_scratch, (gn_need + 64 + (E->gn_fpart_floats = (size_t)cfg->max_batch *
                                                            ((size_t)cfg->latent_h * cfg->latent_w * 16 / 16 + 1) * 32 * 2)) * sizeof(float)) != hipSuccess) {
      ctta_set_error("vae_encoder_create: hipMalloc of %zu-byte activation arena failed", bytes);
      st = CTTA_ERR_NOMEM;
    } else {
      E->gn_scratch_floats = gn_need + 64;
      st = E->splitws.init();
    }
  }
  if (st == CTTA_OK && hipStreamSynchronize(s) != hipSuccess) { ctta_set_error("vae_encoder_create: stream sync failed"); st = CTTA_ERR_HIP; }
  if (st != CTTA_OK) { ctta_vae_encoder_destroy(E); return st; }
  *out = E;
  return CTTA_OK;
}

extern "C" ctta_status ctta_vae_encode(ctta_vae_encoder* E, const float* mel, int batch, float* moments, void* stream) {
  CTTA_REQUIRE(E && mel && moments, "vae_encode: null pointer");
  CTTA_REQUIRE(batch >= 1 && batch <= E->cfg.max_batch, "vae_encode: batch %d outside [1,%d]", batch, E->cfg.max_batch);
  WsBind bind(E->splitws);
  return vae_encode_impl(E, false, mel, batch, moments, (hipStream_t)stream, nullptr);
}
extern "C" ctta_status ctta_vae_encoder_load_weights(ctta_vae_encoder* E, const ctta_tensor* weights, int n_weights,
                                                     void* stream) {
  CTTA_REQUIRE(E && weights, "vae_encoder_load_weights: null pointer");
  WeightTable wt;
  wt.build(weights, n_weights);
  return E->store.run_all(wt, (hipStream_t)stream);
}
extern "C" int ctta_vae_encoder_num_taps(const ctta_vae_encoder* E) { return E ? (int)E->taps.size() : 0; }
extern "C" ctta_status ctta_vae_encoder_tap_info(const ctta_vae_encoder* E, int i, const char** name, int dims[4]) {
  CTTA_REQUIRE(E && i >= 0 && i < (int)E->taps.size(), "tap index out of range");
  const Tap& t = E->taps[i];
  *name = t.name.c_str();
  dims[0] = t.b; dims[1] = t.c; dims[2] = t.h; dims[3] = t.w;
  return CTTA_OK;
}
extern "C" ctta_status ctta_vae_encoder_tap_read(ctta_vae_encoder* E, int i, float* dst, void* stream) {
  CTTA_REQUIRE(E && i >= 0 && i < (int)E->taps.size() && dst, "tap index out of range");
  const Tap& t = E->taps[i];
  CTTA_REQUIRE(t.ptr, "tap '%s' has not been produced yet", t.name.c_str());
  return ctta_nhwc_bf16_to_nchw_f32(t.ptr, dst, t.b, t.c, t.h, t.w, t.c_stride, stream);
}

// ====================================================================================== HiFi-GAN
struct Conv1d {
  PackedW p;
  int cin = 0, cout = 0, k = 1, pad = 0, dil = 1;
};
struct ConvT1d {
  PackedW p;
  int cin = 0, cout = 0, k = 0, u = 0, pad = 0, taps = 0;
};
struct HResBlock {
  Conv1d c1[3], c2[3];
  bf16_t* f1[3] = {nullptr, nullptr, nullptr};   // fragment-major copies of c1 / c2 for the fused unit kernel (resunit.hip)
  bf16_t* f2[3] = {nullptr, nullptr, nullptr};
  bool fused = false;
  bool chained = false;                      // all three units in one launch (ctta_reschain_conv1d)
  Conv1d d1[3], d2[3];                       // enable_grad: data-gradient operands
  const bf16_t* sv_xt[3] = {nullptr, nullptr, nullptr};    // leaky_relu(convs1[m](.)) of the differentiable forward
  const bf16_t* sv_ract[3] = {nullptr, nullptr, nullptr};  // leaky_relu of the residual stream entering unit m
};

struct ctta_hifigan {
  ctta_hifigan_config cfg;
  WeightStore store;
  Arena arena;
  SplitWs splitws;
  std::vector<Tap> taps;
  Conv1d conv_pre;
  std::vector<ConvT1d> ups;
  std::vector<HResBlock> res;   // [n_ups * n_kernels]
  float *post_w = nullptr, *post_b = nullptr;
  int c_last = 0;
  // enable_grad (decode_to_waveform(allow_grad=True), hifigan/utilities.py:79-81)
  Conv1d d_pre;
  std::vector<Conv1d> d_ups;                 // strided convolutions: the ConvTranspose1d data gradients
  struct Saved {
    const bf16_t* xa0 = nullptr;             // leaky_relu(conv_pre(mel))
    std::vector<const bf16_t*> xa_out;       // per stage: leaky_relu(sum_j resblock_j / n_kernels)
    std::vector<int> len;                    // per stage: output length
    int B = 0, frames = 0;
  } sv;
};

static ctta_status make_conv1d(WeightStore& ws, const std::string& p, int cout, int cin, int k, int dil, Conv1d* L) {
  const int K = k * cin, k_pad = round_up(K, 64), n_pad = round_up(cout, 4);
  std::vector<int32_t> ro(n_pad, -1), co(k_pad, -1);
  for (int r = 0; r < cout; ++r) ro[r] = r * cin * k;
  for (int x = 0; x < k; ++x)
    for (int c = 0; c < cin; ++c) co[x * cin + c] = c * k + x;
  CTTA_TRY(ws.add_matrix(p + "weight", {cout, cin, k}, ro, co, nullptr, nullptr, 0, &L->p.w));
  CTTA_TRY(ws.add_vector(p + "bias", cout, n_pad, {{0, 0, cout}}, &L->p.bias));
  L->p.n = n_pad; L->p.k_pad = k_pad;
  L->cin = cin; L->cout = cout; L->k = k; L->dil = dil;
  L->pad = (k * dil - dil) / 2;   // get_padding, hifigan/models.py:18-19
  return CTTA_OK;
}

// ConvTranspose1d(cin, cout, k, stride u, padding (k-u)//2) as `u` phase convolutions in one GEMM:
// with s = t + pad, r = s % u, q = s / u:  out[t] = sum_m W[:, :, r + m*u] x[q - m].
// Row n = r*cout + co of the packed matrix, column (tap kw, ci) with m = taps-1-kw.
static ctta_status make_convt1d(WeightStore& ws, const std::string& p, int cin, int cout, int k, int u, ConvT1d* L) {
  const int taps = (k + u - 1) / u;
  const int K = taps * cin, k_pad = round_up(K, 64), n = u * cout;
  std::vector<int32_t> ro(n), ra(n), co(k_pad, -1), ca(k_pad, 0);
  for (int r = 0; r < u; ++r)
    for (int o = 0; o < cout; ++o) { ro[r * cout + o] = o * k + r; ra[r * cout + o] = r; }
  for (int kw = 0; kw < taps; ++kw) {
    const int m = taps - 1 - kw;
    for (int c = 0; c < cin; ++c) { co[kw * cin + c] = c * cout * k + m * u; ca[kw * cin + c] = m * u; }
  }
  CTTA_TRY(ws.add_matrix(p + "weight", {cin, cout, k}, ro, co, &ra, &ca, k, &L->p.w));
  std::vector<WeightStore::Seg> segs;
  for (int r = 0; r < u; ++r) segs.push_back({0, r * cout, cout});
  CTTA_TRY(ws.add_vector(p + "bias", cout, n, segs, &L->p.bias));
  L->p.n = n; L->p.k_pad = k_pad;
  L->cin = cin; L->cout = cout; L->k = k; L->u = u; L->pad = (k - u) / 2; L->taps = taps;
  return CTTA_OK;
}

// Data-gradient operands.  Conv1d ("same" padding, dilation d): dX = conv(dY, taps reversed, channels swapped), same
// dilation, padding (k-1)d - pad = pad; built as bf16 transposes of the forward pack.  ConvTranspose1d(stride u):
// dX[q][ci] = sum_{k,co} dY[q*u - pad + k][co] W[ci][co][k], a stride-u convolution over dY.
static ctta_status make_conv1d_dgrad(WeightStore& ws, const Conv1d& F, Conv1d* D) {
  ConvLayer f, d;
  f.p = F.p; f.cin_pad = F.cin; f.cout = F.cout; f.kh = 1; f.kw = F.k; f.pad = 0;
  CTTA_TRY(make_conv_dgrad_from(ws, f, F.cout, F.cin, &d));
  D->p = d.p; D->cin = F.cout; D->cout = F.cin; D->k = F.k; D->dil = F.dil; D->pad = (F.k - 1) * F.dil - F.pad;
  return CTTA_OK;
}
static ctta_status make_convt1d_dgrad(WeightStore& ws, const std::string& p, const ConvT1d& F, Conv1d* D) {
  const int K = F.k * F.cout, k_pad = round_up(K, 64), n_pad = round_up(F.cin, 4);
  std::vector<int32_t> ro(n_pad, -1), co(k_pad, -1);
  for (int ci = 0; ci < F.cin; ++ci) ro[ci] = ci * F.cout * F.k;
  for (int kk = 0; kk < F.k; ++kk)
    for (int o = 0; o < F.cout; ++o) co[kk * F.cout + o] = o * F.k + kk;
  CTTA_TRY(ws.add_matrix(p + "weight", {F.cin, F.cout, F.k}, ro, co, nullptr, nullptr, 0, &D->p.w));
  D->p.bias = nullptr; D->p.n = n_pad; D->p.k_pad = k_pad;
  D->cin = F.cout; D->cout = F.cin; D->k = F.k; D->dil = F.u /* stride */; D->pad = F.pad;
  return CTTA_OK;
}

// Output-side activation plumbing of one vocoder conv: every LeakyReLU of Generator.forward /
// ResBlock.forward is applied ONCE, in the epilogue of the conv that produces its argument
// (out_slope: the only consumer wants lrelu(out); out2: both out and lrelu(out) are needed),
// instead of on every K-step re-read of the operand.
struct OutAct {
  float out_slope = 0.f;     // > 0: out = leaky_relu(v, out_slope)
  bf16_t* out2 = nullptr;    // != null: out2 = leaky_relu(out, out2_slope)
  float out2_slope = 0.f;
};

static ctta_status run_conv1d(RunCtx& c, const Conv1d& L, const bf16_t* x, int B, int len, bf16_t* out,
                              const bf16_t* res, bool accumulate, float alpha, const OutAct& oa) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = x; d.c0 = L.cin;
  d.batch = B; d.hi = 1; d.wi = len; d.ho = 1; d.wo = len;
  d.kh = 1; d.kw = L.k; d.pad_w = L.pad; d.dil_w = L.dil;
  d.w = L.p.w; d.k_pad = L.p.k_pad; d.n = L.p.n; d.bias = L.p.bias;
  d.res = res; d.res_ld = L.cout;
  d.accumulate = accumulate ? 1 : 0; d.alpha = alpha;
  if (oa.out_slope > 0.f) { d.out_act = 3; d.out_slope = oa.out_slope; }
  d.out2 = oa.out2; d.out2_slope = oa.out2_slope;
  d.out = out; d.ldc = L.cout;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

static inline int convt_out_len(const ConvT1d& L, int len) { return (len - 1) * L.u - 2 * L.pad + L.k; }

static ctta_status run_convt1d(RunCtx& c, const ConvT1d& L, const bf16_t* x, int B, int len, bf16_t* out,
                               const OutAct& oa) {
  const int lout = convt_out_len(L, len);
  const int Q = (lout - 1 + L.pad) / L.u + 1;
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = x; d.c0 = L.cin;
  d.batch = B; d.hi = 1; d.wi = len; d.ho = 1; d.wo = Q;
  d.kh = 1; d.kw = L.taps; d.pad_w = L.taps - 1;
  d.w = L.p.w; d.k_pad = L.p.k_pad; d.n = L.p.n; d.bias = L.p.bias;
  d.out2 = oa.out2; d.out2_slope = oa.out2_slope;
  d.out = out; d.ldc = L.u * L.cout;
  d.out_batch_stride = (int64_t)lout * L.cout;
  d.out_offset = -(int64_t)L.pad * L.cout;
  d.out_limit = (int64_t)lout * L.cout;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

extern "C" int64_t ctta_hifigan_out_len(const ctta_hifigan* G, int frames) {
  if (!G) return 0;
  int len = frames;
  for (const ConvT1d& u : G->ups) len = convt_out_len(u, len);
  return len;
}

static ctta_status hifigan_forward_impl(ctta_hifigan* G, bool dry, const float* mel, int B, int frames, float* wav,
                                        hipStream_t stream, bool grad = false) {
  const ctta_hifigan_config& cfg = G->cfg;
  RunCtx c;
  c.arena = &G->arena; c.stream = stream; c.dry = dry;
  c.taps = cfg.debug_taps ? &G->taps : nullptr;
  c.gn_scratch = nullptr; c.gn_scratch_floats = 0;
  Arena& A = G->arena;
  A.reset();
  const int nk = cfg.n_kernels;
  bf16_t* m = A.get<bf16_t>((size_t)B * frames * cfg.num_mels); ALLOC_OR_FAIL(m);
  RUN(c, ctta_rows_f32_to_bf16(mel, m, (int64_t)B * frames, cfg.num_mels, cfg.num_mels, stream));
  int len = frames, ch = cfg.upsample_initial_channel;
  // xa always holds leaky_relu(x): the raw conv_pre / stage outputs have no other consumer
  bf16_t* xa = A.get<bf16_t>((size_t)B * len * ch); ALLOC_OR_FAIL(xa);
  {
    OutAct oa; oa.out_slope = 0.1f;
    CTTA_TRY(run_conv1d(c, G->conv_pre, m, B, len, xa, nullptr, false, 1.0f, oa));
  }
  add_tap(c, "lrelu.conv_pre", xa, B, ch, 1, len, ch);
  G->sv.B = 0;
  if (grad) {
    G->sv.B = B; G->sv.frames = frames; G->sv.xa0 = xa;
    G->sv.xa_out.assign(cfg.n_ups, nullptr);
    G->sv.len.assign(cfg.n_ups, 0);
  }
  for (int i = 0; i < cfg.n_ups; ++i) {
    const ConvT1d& U = G->ups[i];
    const int lout = convt_out_len(U, len);
    const size_t elems = (size_t)B * lout * U.cout;
    bf16_t* y = A.get<bf16_t>(elems); ALLOC_OR_FAIL(y);        // x = ups[i](leaky_relu(x, 0.1))
    bf16_t* ya = A.get<bf16_t>(elems); ALLOC_OR_FAIL(ya);      // leaky_relu(x, 0.1) for the 3 resblocks
    {
      bool all_fused = !grad;
      for (int j = 0; j < nk; ++j) all_fused = all_fused && G->res[i * nk + j].fused;
      OutAct oa;
      if (!all_fused) { oa.out2 = ya; oa.out2_slope = 0.1f; }   // the fused units apply leaky_relu themselves
      CTTA_TRY(run_convt1d(c, U, xa, B, len, y, oa));
    }
    len = lout; ch = U.cout;
    add_tap(c, "ups." + std::to_string(i), y, B, ch, 1, len, ch);
    bf16_t* xs = A.get<bf16_t>(elems); ALLOC_OR_FAIL(xs);
    const size_t mk = A.mark();
    bf16_t* xt = A.get<bf16_t>(elems); ALLOC_OR_FAIL(xt);
    bf16_t* ra = A.get<bf16_t>(elems); ALLOC_OR_FAIL(ra);
    bf16_t* rb = A.get<bf16_t>(elems); ALLOC_OR_FAIL(rb);
    bf16_t* aa = A.get<bf16_t>(elems); ALLOC_OR_FAIL(aa);
    bf16_t* ab = A.get<bf16_t>(elems); ALLOC_OR_FAIL(ab);
    const bool last_stage = i == cfg.n_ups - 1;
    for (int j = 0; j < nk; ++j) {
      HResBlock& R = G->res[i * nk + j];
      const bf16_t* r = y;       // residual stream (raw)
      const bf16_t* ract = ya;   // leaky_relu(r, 0.1)
      if (R.chained && !grad) {  // one launch per ResBlock: residual stream and intermediates stay in LDS across the 3 units
        const bool fin = j == nk - 1;
        const int dils[3] = {R.c1[0].dil, R.c1[1].dil, R.c1[2].dil};
        const void* w1[3] = {R.f1[0], R.f1[1], R.f1[2]};
        const void* w2[3] = {R.f2[0], R.f2[1], R.f2[2]};
        const float* b1[3] = {R.c1[0].p.bias, R.c1[1].p.bias, R.c1[2].p.bias};
        const float* b2[3] = {R.c2[0].p.bias, R.c2[1].p.bias, R.c2[2].p.bias};
        RUN(c, ctta_reschain_conv1d(y, B, len, ch, R.c1[0].k, dils, w1, b1, w2, b2, 0.1f, xs, j > 0 ? 1 : 0,
                                    fin ? 1.0f / (float)nk : 1.0f, fin ? (last_stage ? 0.01f : 0.1f) : 0.f, c.stream));
        continue;
      }
      if (R.fused && !grad) {    // one launch per unit: x -> x + conv2(lrelu(conv1(lrelu(x)))), intermediate kept in LDS
        for (int mth = 0; mth < 3; ++mth) {
          const bool last = mth == 2, fin = j == nk - 1;
          bf16_t* dst = last ? xs : ((r == ra) ? rb : ra);
          const float out_slope = (last && fin) ? (last_stage ? 0.01f : 0.1f) : 0.f;
          RUN(c, ctta_resunit_conv1d(r, B, len, ch, R.c1[mth].k, R.c1[mth].dil, R.f1[mth], R.c1[mth].p.bias, R.f2[mth],
                                     R.c2[mth].p.bias, 0.1f, dst, (last && j > 0) ? 1 : 0,
                                     (last && fin) ? 1.0f / (float)nk : 1.0f, out_slope, c.stream));
          r = dst;
        }
        continue;
      }
      for (int mth = 0; mth < 3; ++mth) {
        if (grad) {   // every activated tensor is a LeakyReLU mask of the backward: no buffer reuse
          xt = A.get<bf16_t>(elems); ALLOC_OR_FAIL(xt);
          R.sv_xt[mth] = xt; R.sv_ract[mth] = ract;
        }
        OutAct o1; o1.out_slope = 0.1f;                 // xt is only ever consumed through leaky_relu
        CTTA_TRY(run_conv1d(c, R.c1[mth], ract, B, len, xt, nullptr, false, 1.0f, o1));
        if (grad) {   // the backward's LeakyReLU masks, readable by the parity tests
          const std::string tn = "mask.res." + std::to_string(i) + "." + std::to_string(j) + "." + std::to_string(mth);
          add_tap(c, tn + ".a", ract, B, ch, 1, len, ch);
          add_tap(c, tn + ".b", xt, B, ch, 1, len, ch);
        }
        const bool last = mth == 2;
        if (!last) {
          bf16_t* dst = (r == ra) ? rb : ra;
          bf16_t* dact = (ract == aa) ? ab : aa;
          if (grad) { dact = A.get<bf16_t>(elems); ALLOC_OR_FAIL(dact); }
          OutAct o2; o2.out2 = dact; o2.out2_slope = 0.1f;
          CTTA_TRY(run_conv1d(c, R.c2[mth], xt, B, len, dst, r, false, 1.0f, o2));
          r = dst; ract = dact;
        } else {
          // the block's last conv adds its residual AND folds into xs = (sum_j resblock_j(x)) / nk;
          // the final fold also applies the NEXT consumer's leaky_relu (slope 0.1, or 0.01 before conv_post)
          const bool fin = j == nk - 1;
          OutAct o2;
          if (fin) o2.out_slope = last_stage ? 0.01f : 0.1f;
          CTTA_TRY(run_conv1d(c, R.c2[mth], xt, B, len, xs, r, j > 0, fin ? 1.0f / (float)nk : 1.0f, o2));
        }
      }
    }
    if (!grad) A.release(mk);
    xa = xs;
    if (grad) { G->sv.xa_out[i] = xa; G->sv.len[i] = len; }
    add_tap(c, "lrelu.stage." + std::to_string(i), xa, B, ch, 1, len, ch);
  }
  // x = tanh(conv_post(leaky_relu(x)))  -- the leaky_relu (default slope 0.01, models.py:113) is already in xa
  RUN(c, ctta_conv_small_n(xa, ch, B, 1, len, 1, 7, 0, 3, G->post_w, G->post_b, 1, 0, 0.f, 2, wav, nullptr, stream));
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ vocoder input gradient
// d wav / d mel of the frozen Generator (Generator.forward hifigan/models.py:101-117 in reverse): tanh and conv_post
// in one small kernel, then per stage the three ResBlocks (two data-gradient convolutions + two LeakyReLU masks per
// unit, residual stream folded into the mask kernel), the ConvTranspose1d data gradient as a strided convolution,
// and conv_pre's data gradient written as the fp32 (B, frames, num_mels) mel gradient.
static ctta_status run_conv1d_dgrad(RunCtx& c, const Conv1d& D, const bf16_t* dy, int B, int len_in, int len_out,
                                    int stride, void* out, bool f32) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = dy; d.c0 = D.cin;
  d.batch = B; d.hi = 1; d.wi = len_in; d.ho = 1; d.wo = len_out;
  d.kh = 1; d.kw = D.k; d.pad_w = D.pad;
  if (stride > 0) d.stride_w = stride; else d.dil_w = D.dil;
  d.w = D.p.w; d.k_pad = D.p.k_pad; d.n = D.p.n;
  d.out = out; d.ldc = D.cout; d.out_f32 = f32 ? 1 : 0;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

static ctta_status hifigan_backward_impl(ctta_hifigan* G, bool dry, const float* gwav, const float* wav, int B, float* gmel,
                                         hipStream_t stream) {
  const ctta_hifigan_config& cfg = G->cfg;
  RunCtx c;
  c.arena = &G->arena; c.stream = stream; c.dry = dry;
  c.taps = cfg.debug_taps ? &G->taps : nullptr; c.gn_scratch = nullptr; c.gn_scratch_floats = 0;
  Arena& A = G->arena;   // continues above the differentiable forward's saved tensors
  const int nk = cfg.n_kernels, last = cfg.n_ups - 1;
  int len = G->sv.len[last], ch = G->c_last;
  bf16_t* g = A.get<bf16_t>((size_t)B * len * ch); ALLOC_OR_FAIL(g);    // gradient w.r.t. the stage's activated output
  RUN(c, ctta_conv_cout1_dgrad(gwav, wav, G->post_w, B, 1, len, 1, 7, 0, 3, ch, nullptr, 0.f, g, stream));
  for (int i = last; i >= 0; --i) {
    add_tap(c, "grad.lrelu.stage." + std::to_string(i), g, B, ch, 1, len, ch);
    const ConvT1d& U = G->ups[i];
    const int len_prev = i > 0 ? G->sv.len[i - 1] : G->sv.frames;
    const size_t elems = (size_t)B * len * ch;
    bf16_t* gs = A.get<bf16_t>(elems); ALLOC_OR_FAIL(gs);             // d / d(resblock output), 1/n_kernels folded in
    RUN(c, ctta_lrelu_bwd(g, G->sv.xa_out[i], i == last ? 0.01f : 0.1f, 1.0f / (float)nk, nullptr, gs, (int64_t)elems, 0,
                          stream));
    bf16_t* gy = A.get<bf16_t>(elems); ALLOC_OR_FAIL(gy);             // d / d(ups[i] output), summed over the blocks
    bf16_t* gprev = A.get<bf16_t>((size_t)B * len_prev * U.cin); ALLOC_OR_FAIL(gprev);
    const size_t mk = A.mark();
    bf16_t* gxt = A.get<bf16_t>(elems); ALLOC_OR_FAIL(gxt);
    bf16_t* gu = A.get<bf16_t>(elems); ALLOC_OR_FAIL(gu);
    bf16_t* gra = A.get<bf16_t>(elems); ALLOC_OR_FAIL(gra);
    bf16_t* ra = A.get<bf16_t>(elems); ALLOC_OR_FAIL(ra);
    bf16_t* rb = A.get<bf16_t>(elems); ALLOC_OR_FAIL(rb);
    for (int j = 0; j < nk; ++j) {
      const HResBlock& R = G->res[i * nk + j];
      const bf16_t* gr = gs;
      for (int m = 2; m >= 0; --m) {   // r' = convs2(leaky_relu(convs1(leaky_relu(r)))) + r
        CTTA_TRY(run_conv1d_dgrad(c, R.d2[m], gr, B, len, len, 0, gxt, false));
        RUN(c, ctta_lrelu_bwd(gxt, R.sv_xt[m], 0.1f, 1.0f, nullptr, gu, (int64_t)elems, 0, stream));
        CTTA_TRY(run_conv1d_dgrad(c, R.d1[m], gu, B, len, len, 0, gra, false));
        if (m > 0) {
          bf16_t* dst = (gr == ra) ? rb : ra;
          RUN(c, ctta_lrelu_bwd(gra, R.sv_ract[m], 0.1f, 1.0f, gr, dst, (int64_t)elems, 0, stream));
          gr = dst;
        } else {
          RUN(c, ctta_lrelu_bwd(gra, R.sv_ract[0], 0.1f, 1.0f, gr, gy, (int64_t)elems, j > 0 ? 1 : 0, stream));
        }
      }
    }
    add_tap(c, "grad.ups." + std::to_string(i), gy, B, ch, 1, len, ch);
    CTTA_TRY(run_conv1d_dgrad(c, G->d_ups[i], gy, B, len, len_prev, U.u, gprev, false));
    A.release(mk);
    g = gprev; len = len_prev; ch = U.cin;
  }
  const size_t elems = (size_t)B * len * ch;
  bf16_t* gpre = A.get<bf16_t>(elems); ALLOC_OR_FAIL(gpre);
  RUN(c, ctta_lrelu_bwd(g, G->sv.xa0, 0.1f, 1.0f, nullptr, gpre, (int64_t)elems, 0, stream));
  CTTA_TRY(run_conv1d_dgrad(c, G->d_pre, gpre, B, len, len, 0, gmel, true));
  return CTTA_OK;
}

static ctta_status hifigan_build(ctta_hifigan* G) {
  const ctta_hifigan_config& cfg = G->cfg;
  WeightStore& ws = G->store;
  const std::string P = "vocoder.";
  const int c0 = cfg.upsample_initial_channel;
  const bool grad = cfg.enable_grad != 0;
  CTTA_TRY(make_conv1d(ws, P + "conv_pre.", c0, cfg.num_mels, 7, 1, &G->conv_pre));
  if (grad) CTTA_TRY(make_conv1d_dgrad(ws, G->conv_pre, &G->d_pre));
  G->ups.resize(cfg.n_ups);
  G->d_ups.resize(cfg.n_ups);
  G->res.resize((size_t)cfg.n_ups * cfg.n_kernels);
  int ch = c0;
  for (int i = 0; i < cfg.n_ups; ++i) {
    const int cin = c0 >> i, cout = c0 >> (i + 1);
    CTTA_REQUIRE(cout % 8 == 0, "hifigan: channel count %d not a multiple of 8", cout);
    CTTA_TRY(make_convt1d(ws, P + "ups." + std::to_string(i) + ".", cin, cout, cfg.upsample_kernel_sizes[i],
                          cfg.upsample_rates[i], &G->ups[i]));
    if (grad) CTTA_TRY(make_convt1d_dgrad(ws, P + "ups." + std::to_string(i) + ".", G->ups[i], &G->d_ups[i]));
    ch = cout;
    for (int j = 0; j < cfg.n_kernels; ++j) {
      HResBlock& R = G->res[(size_t)i * cfg.n_kernels + j];
      const std::string rp = P + "resblocks." + std::to_string(i * cfg.n_kernels + j) + ".";
      const int k = cfg.resblock_kernel_sizes[j];
      for (int m = 0; m < 3; ++m) {
        CTTA_TRY(make_conv1d(ws, rp + "convs1." + std::to_string(m) + ".", ch, ch, k, cfg.resblock_dilations[j][m], &R.c1[m]));
        CTTA_TRY(make_conv1d(ws, rp + "convs2." + std::to_string(m) + ".", ch, ch, k, 1, &R.c2[m]));
        if (grad) {
          CTTA_TRY(make_conv1d_dgrad(ws, R.c1[m], &R.d1[m]));
          CTTA_TRY(make_conv1d_dgrad(ws, R.c2[m], &R.d2[m]));
        }
      }
      // narrow stages: the plain forward runs each unit (conv1 -> lrelu -> conv2 -> + x) as ONE fused launch that reads
      // its weights fragment-major; the copies are re-derived from the packed operands after every (re)load
      R.fused = !grad;
      for (int m = 0; m < 3; ++m) R.fused = R.fused && ctta_resunit_supported(ch, k, cfg.resblock_dilations[j][m]) != 0;
      // the chained form (one launch per ResBlock) is correct and tested, but measured SLOWER than three unit launches
      // once those lost their serialised staging loads (round 3, B = 32: C = 32 k = 3 0.76 vs 0.63 ms, k = 7 1.34 vs 0.81 ms:
      // the recomputed halo and two workgroups per CU cost more than the two saved HBM round trips): opt-in only
      R.chained = false;      // (ctta_reschain_conv1d stays an exported, tested operator; this engine does not take it)
      if (R.fused) {
        for (int m = 0; m < 3; ++m) {
          R.f1[m] = ws.arena.get<bf16_t>((size_t)ch * k * ch);
          R.f2[m] = ws.arena.get<bf16_t>((size_t)ch * k * ch);
          if (!R.f1[m] || !R.f2[m]) { ctta_set_error("weight store exhausted (fragment-major vocoder weights)"); return CTTA_ERR_NOMEM; }
        }
        HResBlock* Rp = &R;
        const int chc = ch;
        ws.jobs.push_back([Rp, chc, k](const WeightTable&, hipStream_t s) -> ctta_status {
          for (int m = 0; m < 3; ++m) {
            CTTA_TRY(ctta_frag_pack(Rp->c1[m].p.w, chc, Rp->c1[m].p.k_pad, k * chc, Rp->f1[m], s));
            CTTA_TRY(ctta_frag_pack(Rp->c2[m].p.w, chc, Rp->c2[m].p.k_pad, k * chc, Rp->f2[m], s));
          }
          return CTTA_OK;
        });
      }
    }
  }
  G->c_last = ch;
  CTTA_TRY(add_small_conv_w(ws, P + "conv_post.weight", 1, ch, 7, &G->post_w));
  CTTA_TRY(ws.add_vector(P + "conv_post.bias", 1, &G->post_b));
  return CTTA_OK;
}

extern "C" ctta_status ctta_hifigan_create(const ctta_hifigan_config* cfg, const ctta_tensor* weights, int n_weights,
                                           void* stream, ctta_hifigan** out) {
  CTTA_REQUIRE(cfg && weights && out, "hifigan_create: null pointer");
  CTTA_REQUIRE(cfg->n_ups >= 1 && cfg->n_ups <= CTTA_MAX_UPS && cfg->n_kernels >= 1 && cfg->n_kernels <= 4,
               "hifigan_create: bad n_ups/n_kernels");
  CTTA_REQUIRE(cfg->num_mels % 8 == 0, "hifigan_create: num_mels=%d must be a multiple of 8", cfg->num_mels);
  hipStream_t s = (hipStream_t)stream;
  ctta_hifigan* G = new ctta_hifigan();
  G->cfg = *cfg;
  ctta_status st = G->store.init(cfg->enable_grad ? estimate_store_bytes_training(weights, n_weights)
                                                  : estimate_store_bytes(weights, n_weights));
  if (st != CTTA_OK) { delete G; return st; }
  WeightTable wt;
  wt.build(weights, n_weights);
  st = hifigan_build(G);
  if (st == CTTA_OK) st = G->store.run_all(wt, s);
  if (st == CTTA_OK) {
    G->arena.dry = true;
    G->arena.no_release = cfg->debug_taps != 0;
    st = hifigan_forward_impl(G, true, nullptr, cfg->max_batch, cfg->max_frames, nullptr, s);
    if (st == CTTA_OK && cfg->enable_grad) {
      st = hifigan_forward_impl(G, true, nullptr, cfg->max_batch, cfg->max_frames, nullptr, s, true);
      if (st == CTTA_OK) st = hifigan_backward_impl(G, true, nullptr, nullptr, cfg->max_batch, nullptr, s);
      G->sv.B = 0;
    }
  }
  if (st == CTTA_OK) {
    const size_t bytes = G->arena.peak + (1 << 20);
    G->arena.dry = false;
    G->arena.cap = bytes;
    if (hipMalloc((void**)&G->arena.base, bytes) != hipSuccess) {
      ctta_set_error("hifigan_create: hipMalloc of %zu-byte activation arena failed", bytes);
      st = CTTA_ERR_NOMEM;
    } else {
      st = G->splitws.init();
    }
  }
  if (st == CTTA_OK && hipStreamSynchronize(s) != hipSuccess) { ctta_set_error("hifigan_create: stream sync failed"); st = CTTA_ERR_HIP; }
  if (st != CTTA_OK) { ctta_hifigan_destroy(G); return st; }
  *out = G;
  return CTTA_OK;
}

extern "C" void ctta_hifigan_destroy(ctta_hifigan* G) {
  if (!G) return;
  G->store.destroy();
  if (G->arena.base) (void)hipFree(G->arena.base);
  G->splitws.destroy();
  delete G;
}

extern "C" ctta_status ctta_hifigan_forward(ctta_hifigan* G, const float* mel, int batch, int frames, float* wav,
                                            void* stream) {
  CTTA_REQUIRE(G && mel && wav, "hifigan_forward: null pointer");
  CTTA_REQUIRE(batch >= 1 && batch <= G->cfg.max_batch && frames >= 1 && frames <= G->cfg.max_frames,
               "hifigan_forward: batch %d / frames %d outside the handle's limits (%d, %d)", batch, frames,
               G->cfg.max_batch, G->cfg.max_frames);
  WsBind bind(G->splitws);
  return hifigan_forward_impl(G, false, mel, batch, frames, wav, (hipStream_t)stream);
}

extern "C" ctta_status ctta_hifigan_forward_with_grad(ctta_hifigan* G, const float* mel, int batch, int frames, float* wav,
                                                      void* stream) {
  CTTA_REQUIRE(G && mel && wav, "hifigan_forward_with_grad: null pointer");
  CTTA_REQUIRE(G->cfg.enable_grad, "hifigan_forward_with_grad: the handle was created without enable_grad");
  CTTA_REQUIRE(batch >= 1 && batch <= G->cfg.max_batch && frames >= 1 && frames <= G->cfg.max_frames,
               "hifigan_forward_with_grad: batch %d / frames %d outside the handle's limits (%d, %d)", batch, frames,
               G->cfg.max_batch, G->cfg.max_frames);
  WsBind bind(G->splitws);
  return hifigan_forward_impl(G, false, mel, batch, frames, wav, (hipStream_t)stream, true);
}
extern "C" ctta_status ctta_hifigan_backward(ctta_hifigan* G, const float* grad_wav, const float* wav, int batch, int frames,
                                             float* grad_mel, void* stream) {
  CTTA_REQUIRE(G && grad_wav && wav && grad_mel, "hifigan_backward: null pointer");
  CTTA_REQUIRE(G->cfg.enable_grad && G->sv.B == batch && G->sv.frames == frames,
               "hifigan_backward: no differentiable forward of batch %d x %d frames is pending on this handle", batch, frames);
  WsBind bind(G->splitws);
  const ctta_status st = hifigan_backward_impl(G, false, grad_wav, wav, batch, grad_mel, (hipStream_t)stream);
  G->sv.B = 0;
  return st;
}

extern "C" size_t ctta_hifigan_arena_bytes(const ctta_hifigan* G) { return G ? G->arena.cap + G->store.arena.cap : 0; }
extern "C" int ctta_hifigan_num_taps(const ctta_hifigan* G) { return G ? (int)G->taps.size() : 0; }
extern "C" ctta_status ctta_hifigan_tap_info(const ctta_hifigan* G, int i, const char** name, int dims[4]) {
  CTTA_REQUIRE(G && i >= 0 && i < (int)G->taps.size(), "tap index out of range");
  const Tap& t = G->taps[i];
  *name = t.name.c_str();
  dims[0] = t.b; dims[1] = t.c; dims[2] = t.h; dims[3] = t.w;
  return CTTA_OK;
}
extern "C" ctta_status ctta_hifigan_tap_read(ctta_hifigan* G, int i, float* dst, void* stream) {
  CTTA_REQUIRE(G && i >= 0 && i < (int)G->taps.size() && dst, "tap index out of range");
  const Tap& t = G->taps[i];
  CTTA_REQUIRE(t.ptr, "tap '%s' has not been produced yet", t.name.c_str());
  return ctta_nhwc_bf16_to_nchw_f32(t.ptr, dst, t.b, t.c, t.h, t.w, t.c_stride, stream);
}
