// Backward pass of the student U-Net for the consistency-distillation step (included by
// engine_unet.hip; same translation unit).  The reference gets these gradients from torch autograd
// over UNet2DConditionGuidedModel.forward (train.py:332-346 -> accelerator.backward(loss)); here the
// tape the training forward recorded is replayed in reverse with hand-written operators:
//   * data gradients   : conv_gemm against the re-packed (rotated / transposed) weights,
//   * weight gradients : one split-M GEMM  dW[k][n] = sum_m Q[k][m] * dY^T[n][m]  per layer, where
//                        Q = im2col(X)^T plus an all-ones row (bias gradient) and, for the resnets'
//                        conv1, one indicator row per sample (time-embedding gradient),
//   * attention        : flash-style backward (ctta_attention_bwd): probabilities rebuilt from the forward's
//                        log-sum-exp, dV = P^T dO, dP = dO V^T, dS = P (dP - D) scale, dQ = dS K, dK = dS^T Q,
//   * norms / GEGLU / embedding MLP: the kernels in backward.hip.
// Gradients are ACCUMULATED into the caller's fp32 tensors (same names / shapes as the state dict),
// i.e. `.grad` semantics; the caller zeroes them.  Activation gradients travel in bf16.

struct GradTable {
  std::unordered_map<std::string, float*> map;
  void build(const ctta_tensor* g, int n) {
    map.clear();
    for (int i = 0; i < n; ++i) map[g[i].name] = const_cast<float*>(g[i].data);   // ctta_tensor is shared with the (read-only) weight tables
  }
};

struct BCtx : UCtx {
  const GradTable* grads = nullptr;
  float* dtemb_all = nullptr;   // [B][temb_total] fp32, filled from the conv1 indicator rows
};

// One weight-gradient job: `w` is the context its launches use (scratch slot as arena; side stream when enabled).
struct WgJob {
  BCtx w;
  int slot = 0;
  bool async = false;
};
static ctta_status wg_begin(BCtx& c, WgJob* j) {
  ctta_unet::WgradSide& W = c.U->wg;
  j->slot = W.next;
  W.next = (W.next + 1) % ctta_unet::WgradSide::NS;
  // ("wgrad_stream" is read per job: switched off between two backward passes -- bench.py's profiled step, whose per-launch
  // event brackets must not overlap -- the jobs run on the caller's stream in the same slots; the handle's arena policy,
  // W.enabled, was fixed when it was created)
  j->async = !c.dry && W.enabled && W.stream != nullptr && ctta_opt(CTTA_OPT_WGRAD_STREAM) != 0;
  j->w = c;
  j->w.arena = &W.slot[j->slot];
  W.slot[j->slot].reset();
  if (!j->async && !c.dry && W.in_use[j->slot]) {      // a side-stream job of an earlier pass may still own the slot
    CTTA_CHECK_HIP(hipStreamWaitEvent(c.stream, W.freed[j->slot], 0));
    W.in_use[j->slot] = false;
  }
  if (j->async) {
    j->w.stream = W.stream;
    // the slot's previous job must have drained before the main stream writes dY^T into it.  Jobs that read their operands
    // in place (ctta_wgrad_tn) never touch the slot from the main stream, but they keep this wait: it is the back-pressure
    // that lets the main stream run at most two jobs ahead of the side stream.  Without it the main stream races to the
    // block's wg_join and idles there while the side stream drains alone: 92.4 vs 83.9 ms per pipelined step (round 4).
    if (W.in_use[j->slot]) CTTA_CHECK_HIP(hipStreamWaitEvent(c.stream, W.freed[j->slot], 0));
  }
  return CTTA_OK;
}
// dY^T is in the slot (written on the main stream): hand over to the side stream
static ctta_status wg_handoff(BCtx& c, WgJob& j) {
  if (!j.async) return CTTA_OK;
  ctta_unet::WgradSide& W = c.U->wg;
  CTTA_CHECK_HIP(hipEventRecord(W.ready[j.slot], c.stream));
  CTTA_CHECK_HIP(hipStreamWaitEvent(W.stream, W.ready[j.slot], 0));
  return CTTA_OK;
}
static ctta_status wg_end(BCtx& c, WgJob& j) {
  if (!j.async) return CTTA_OK;
  ctta_unet::WgradSide& W = c.U->wg;
  CTTA_CHECK_HIP(hipEventRecord(W.freed[j.slot], W.stream));
  W.in_use[j.slot] = true;
  W.dirty = true;
  return CTTA_OK;
}
// the main stream waits for every weight-gradient job issued so far (block boundaries: the caller may start the
// all-reduce of the block's gradients; before the embedding MLPs, which read d temb)
static ctta_status wg_join(BCtx& c) {
  ctta_unet::WgradSide& W = c.U->wg;
  if (c.dry || !W.dirty) return CTTA_OK;
  CTTA_CHECK_HIP(hipEventRecord(W.joined, W.stream));
  CTTA_CHECK_HIP(hipStreamWaitEvent(c.stream, W.joined, 0));
  W.dirty = false;
  W.joined_on = c.stream;
  return CTTA_OK;
}

static ctta_status grad_ptr(BCtx& c, const std::string& key, float** out) {
  *out = nullptr;
  if (c.dry) return CTTA_OK;
  auto it = c.grads->map.find(key);
  if (it == c.grads->map.end() || !it->second) {
    ctta_set_error("unet_backward: no gradient tensor for '%s'", key.c_str());
    return CTTA_ERR_MISSING_KEY;
  }
  *out = it->second;
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ weight gradients
// slabs[s][n][r]: one row per OUTPUT channel n, r = (cin, kh, kw) index of the weight row (then the bias column,
// then per-sample columns) -- the state dict's own row layout, so the scatter is a coalesced streaming add.
struct Slabs { float* p = nullptr; int S = 1, R = 0, ld = 0, N = 0; };

static int pick_splits(int64_t M, int R, int N) {
  const int64_t tiles = (int64_t)((R + 127) / 128) * ((N + 127) / 128);
  int S = 1;
  while (tiles * S < 256 && S < 16 && M / (2 * S) >= 256) S *= 2;
  return S;
}

// slabs[s][n][r] = sum_{m in segment s} dY[m][n] * Q[r][m];  x is the layer input in NHWC (a linear is the 1x1 case
// with batch=1, hi=rows), dy [M][N] contiguous.  Rows of Q: (channel, tap) then the all-ones row, then `nb`
// per-sample indicator rows.  Both GEMM operands are made m-contiguous (dY^T, im2col^T).
static constexpr bool wgrad_inplace_on() { return true; }   // operands read where they lie (round 5); the copies route serves the geometries it does not
// Linear layers: both operands are read WHERE THEY LIE by ctta_wgrad_tn on the side stream (no dY^T, no X^T copy).  dY stays
// valid until the block's wg_join because the backward's arena releases nothing while the side stream is on
// (unet_backward_begin_impl), and no dY is rewritten in place: the transformer's running token-stream gradient moves to a
// new buffer at every LayerNorm backward (ctta_layernorm_bwd_add).
static bool wgrad_inplace_ok(int taps, bool ups, int stride, int pad, int nb, int C, int N) {
  return wgrad_inplace_on() && taps == 1 && !ups && stride == 1 && pad == 0 && nb == 0 && C % 8 == 0 && N % 8 == 0;
}
static constexpr bool wgrad_direct_on() { return true; }    // one-split layers add their tiles straight into the gradient tensors
// `dm` (may be NULL): the layer's pack map when the caller allows the DIRECT form -- one split: the kernel adds its tiles
// straight into the gradient tensors and `out->p` comes back NULL (nothing to scatter).
static ctta_status wgrad_slabs(BCtx& cm, WgJob& job, const bf16_t* x, int C, int B, int hi, int wi, bool ups, int ho, int wo,
                               int kh, int kw, int stride, int pad, const bf16_t* dy, int N, int nb, Slabs* out,
                               const PackMap* dm = nullptr) {
  BCtx& c = job.w;
  Arena& A = *c.arena;
  const int64_t M = (int64_t)B * ho * wo;
  const int K = kh * kw * C;
  const int R = K + 1 + nb;
  const int ld = round_up(R, 4);
  if (wgrad_inplace_ok(kh * kw, ups, stride, pad, nb, C, N) && M < (1LL << 31) - 4096) {
    const int64_t tiles = (int64_t)((N + 127) / 128) * ((C + 127) / 128);
    int S = 1;
    // 128 x 128 tiles at two workgroups per CU; the level-0 / level-1 linears have 4..16 tiles over 9 216..36 864 rows: up to 64
    // splits of >= 4 chunks (16 splits left a 256 x 256 layer with 64 workgroups walking 36 chunks each: 75 us per launch)
    constexpr int tn_cap = 32, tn_target = 256;    // splits aimed at 256 workgroups, at most 32 (round 5)
    // round 5: 256 workgroups / <= 32 splits instead of 512 / 64.  In a replayed step ONE kernel has the device to itself 79 % of
    // the time (profiles/gaps_distill_pipelined_r05.txt): a launch that fills every CU pushes the other streams' kernels behind
    // it, a narrower, longer one runs BESIDE them, and every halving of the splits halves the slab bytes.  Pipelined step
    // 77.5 -> 76.2 ms at (256, 16..32); too narrow (64 workgroups, 8 splits) loses 4 ms (profiles/ab_r05_wgrad_splits.txt).
    while (tiles * S < tn_target && S < tn_cap && M / (2 * S) >= 128) S *= 2;
    const int mp = (int)round_up64(M, 64 * S);
    if (S == 1 && dm && wgrad_direct_on() && dm->n <= N && (dm->bkey.empty() || dm->n_bias <= dm->n)) {
      float *gw, *gb = nullptr;
      CTTA_TRY(grad_ptr(c, dm->wkey, &gw));
      if (!dm->bkey.empty()) CTTA_TRY(grad_ptr(c, dm->bkey, &gb));
      if (!cm.dry) CTTA_TRY(wg_handoff(cm, job));
      RUN(c, ctta_wgrad_tn_direct(dy, N, N, x, C, C, (int)M, mp, dm->k_ident > 0 ? dm->k_ident : K, dm->n, dm->ro,
                                  dm->k_ident > 0 ? nullptr : dm->co, gw, dm->n_bias, dm->bidx, gb, c.stream));
      out->p = nullptr; out->S = 1; out->R = R; out->ld = ld; out->N = N;
      return CTTA_OK;
    }
    float* slabs = A.get<float>((size_t)S * N * ld); ALLOC_OR_FAIL(slabs);
    if (!cm.dry) CTTA_TRY(wg_handoff(cm, job));
    RUN(c, ctta_wgrad_tn(dy, N, N, x, C, C, (int)M, mp, S, K, slabs, (int64_t)N * ld, ld, c.stream));
    out->p = slabs; out->S = S; out->R = R; out->ld = ld; out->N = N;
    return CTTA_OK;
  }
  // the layer input read in place through LDS transpose reads (wgrad_gemm.hip) wherever the geometry allows: every linear
  // and every 3x3 stride-1 convolution of the U-Net except conv_in (8 channels); the stride-2 / upsampling samplers and
  // conv_in keep the im2col^T route below
  // Linears (taps == 1) stay on the transposed-operand route by default: there the old Q is ONE transposed copy of X, not
  // nine, and the implicit kernel's 1-tap tiling reads LDS once per MFMA (round-3 profile: 7.5 ms of side-stream kernel
  // time per step against 3.9 ms for ALL gradient GEMMs before); CTTA_WGRAD_IMPLICIT_LINEAR=1 routes them through it too.
  constexpr bool implicit_linear = false;
  const int taps = kh * kw;
  const bool implicit = !ups && stride == 1 && ho == hi && wo == wi &&
                        ((taps == 9 && kh == 3 && pad == 1) || (taps == 1 && pad == 0 && implicit_linear)) &&
                        M < (1LL << 31) - 4096 && ctta_wgrad_implicit_supported(taps, C, hi, wi, C, N) != 0;
  int S = pick_splits(M, R, N);
  if (implicit) {   // 64 x 64 (x 9 taps) / 64 x 256 tiles, two workgroups per CU
    const int64_t tiles = (int64_t)((N + 63) / 64) * ((C + (taps == 9 ? 63 : 255)) / (taps == 9 ? 64 : 256));
    S = 1;
    constexpr int im_cap = 16, im_target = 256;    // (round 5: 256 workgroups, see the linears above)
    while (tiles * S < im_target && S < im_cap && M / (2 * S) >= 256) S *= 2;
  }
  const int mp = (int)round_up64(M, 64 * S);
  const int seg = mp / S;
  bf16_t* q = nullptr;
  if (!implicit) { q = A.get<bf16_t>((size_t)R * mp); ALLOC_OR_FAIL(q); }
  // 3x3 convolutions: dY read in place too (no convolution's dY is rewritten by the main stream later) when the arena policy
  // keeps it alive for the side stream
  const bool conv_inplace = implicit && kh * kw == 9 && wgrad_inplace_on() && N % 8 == 0 && (nb == 0 || (ho * wo) % 64 == 0) &&
                            (!job.async || c.U->wg.enabled);
  if (conv_inplace && S == 1 && nb == 0 && dm && wgrad_direct_on() && dm->k_ident > 0 && dm->n <= N &&
      (dm->bkey.empty() || dm->n_bias <= dm->n)) {
    float *gw, *gb = nullptr;
    CTTA_TRY(grad_ptr(c, dm->wkey, &gw));
    if (!dm->bkey.empty()) CTTA_TRY(grad_ptr(c, dm->bkey, &gb));
    if (!cm.dry) CTTA_TRY(wg_handoff(cm, job));
    RUN(c, ctta_wgrad_implicit_direct(dy, N, N, mp, x, C, C, B, hi, wi, (int)M, dm->k_ident, dm->n, dm->ro, gw, dm->n_bias, dm->bidx, gb,
                                      c.stream));
    out->p = nullptr; out->S = 1; out->R = R; out->ld = ld; out->N = N;
    return CTTA_OK;
  }
  if (conv_inplace) {
    float* slabs = A.get<float>((size_t)S * N * ld); ALLOC_OR_FAIL(slabs);
    if (!cm.dry) CTTA_TRY(wg_handoff(cm, job));
    RUN(c, ctta_wgrad_implicit_inplace(dy, N, N, mp, x, C, C, B, hi, wi, 9, (int)M, S, K, nb, slabs, (int64_t)N * ld, ld, c.stream));
    out->p = slabs; out->S = S; out->R = R; out->ld = ld; out->N = N;
    return CTTA_OK;
  }
  bf16_t* pt = A.get<bf16_t>((size_t)N * mp); ALLOC_OR_FAIL(pt);
  float* slabs = A.get<float>((size_t)S * N * ld); ALLOC_OR_FAIL(slabs);
  RUN(cm, ctta_transpose_bf16(dy, 0, (int)M, N, N, 0, pt, 0, mp, 1, cm.stream));   // dY lives in the main stream's arena
  if (!cm.dry) CTTA_TRY(wg_handoff(cm, job));
  if (implicit) {
    RUN(c, ctta_wgrad_implicit(pt, N, mp, x, C, C, B, hi, wi, taps, (int)M, S, K, nb, slabs, (int64_t)N * ld, ld, c.stream));
    out->p = slabs; out->S = S; out->R = R; out->ld = ld; out->N = N;
    return CTTA_OK;
  }
  RUN(c, ctta_im2col_t(x, C, B, hi, wi, ups ? 1 : 0, ho, wo, kh, kw, stride, pad, pad, 1, q, mp, nb, c.stream));
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = pt; d.c0 = seg; d.x_stride = mp;
  d.batch = 1; d.hi = N; d.wi = 1; d.ho = N; d.wo = 1;
  // the GEMM produces the K weight columns only; the bias column K (and the nb per-sample columns behind it) are row sums
  // of dY^T (ctta_wgrad_rowsum) -- as GEMM rows (the all-ones / indicator rows im2col_t still writes) they cost a whole
  // extra tile column: N = 257 ran 3 column tiles of 128 for 2 tiles of work
  const bool sums_apart = (nb == 0 || (ho * wo) % 8 == 0) && mp % (8 * S) == 0;
  d.w = q; d.k_pad = mp; d.n = sums_apart ? K : R;
  d.out = slabs; d.ldc = ld; d.out_f32 = 1;
  d.groups = S; d.x_group_stride = seg; d.w_group_stride = seg; d.out_group_stride = (int64_t)N * ld;
  if (job.async) ctta_conv_suppress_splitk(1);   // the handle's split-K slabs belong to the main stream's launches
  const ctta_status gst = c.dry ? CTTA_OK : ctta_conv_gemm(&d, c.stream);
  if (job.async) ctta_conv_suppress_splitk(0);
  CTTA_TRY(gst);
  if (sums_apart) RUN(c, ctta_wgrad_rowsum(pt, N, mp, (int)M, S, ho * wo, nb, slabs, (int64_t)N * ld, ld, K, c.stream));
  out->p = slabs; out->S = S; out->R = R; out->ld = ld; out->N = N;
  return CTTA_OK;
}

// slab rows [row0, row0 + m.n) -> weight / bias gradients of the layer described by `m`
static ctta_status scatter_wgrad(BCtx& c, const Slabs& sl, int k_rows, const PackMap& m, int row0 = 0) {
  const int64_t stride = (int64_t)sl.N * sl.ld;
  const float* base = sl.p + (size_t)row0 * sl.ld;
  float *gw, *gb = nullptr;
  CTTA_TRY(grad_ptr(c, m.wkey, &gw));
  if (!m.bkey.empty()) CTTA_TRY(grad_ptr(c, m.bkey, &gb));
  // weight rows and (same launch) the bias column of the slabs
  const bool bias_here = !m.bkey.empty() && m.n_bias <= m.n;
  const int bcol = bias_here ? k_rows : -1;
  if (m.k_ident > 0)
    RUN(c, ctta_wgrad_scatter_rows_bias(base, sl.S, stride, sl.ld, m.k_ident, m.n, m.ro, nullptr, gw, bcol, m.n_bias, m.bidx, gb, 1,
                                        c.stream));
  else
    RUN(c, ctta_wgrad_scatter_rows_bias(base, sl.S, stride, sl.ld, k_rows, m.n, m.ro, m.co, gw, bcol, m.n_bias, m.bidx, gb, 1,
                                        c.stream));
  if (!m.bkey.empty() && !bias_here)
    RUN(c, ctta_col_scatter(base, sl.S, stride, sl.ld, k_rows, 1, m.n_bias, m.bidx, gb, 0, 1, c.stream));
  return CTTA_OK;
}

static ctta_status conv_wgrad(BCtx& c, const ConvLayer& L, const PackMap& m, const bf16_t* x, int H, int W, bool ups,
                              const bf16_t* dy, int temb_off = -1, const Resnet* temb_of = nullptr) {
  WgJob job;
  CTTA_TRY(wg_begin(c, &job));
  BCtx& w = job.w;
  const int hi = ups ? 2 * H : H, wi = ups ? 2 * W : W;
  const int ho = (hi + 2 * L.pad - L.kh) / L.stride + 1, wo = (wi + 2 * L.pad - L.kw) / L.stride + 1;
  const int nb = temb_off >= 0 ? c.B : 0;
  Slabs sl;
  CTTA_TRY(wgrad_slabs(c, job, x, L.cin_pad, c.B, hi, wi, ups, ho, wo, L.kh, L.kw, L.stride, L.pad, dy, L.p.n, nb, &sl,
                       nb == 0 ? &m : nullptr));
  const int K = L.kh * L.kw * L.cin_pad;
  if (sl.p) CTTA_TRY(scatter_wgrad(w, sl, K, m));      // sl.p == NULL: the kernel added into the gradients itself
  if (nb > 0)   // d temb[b][off + n] = sum over the sample's pixels of dY: columns K+1 .. K+B of the slab
    RUN(w, ctta_col_scatter(sl.p, sl.S, (int64_t)sl.N * sl.ld, sl.ld, K + 1, nb, L.cout, nullptr,
                            c.dtemb_all + temb_off, c.U->temb_total, 0, w.stream));
  if (temb_of) {  // temb = time_emb_proj(SiLU(emb)): its weight / bias gradients need only this resnet's rows of d temb,
                  // which the scatter above just produced (same stream, in order)
    float *gw, *gb;
    CTTA_TRY(grad_ptr(w, temb_of->key + "time_emb_proj.weight", &gw));
    CTTA_TRY(grad_ptr(w, temb_of->key + "time_emb_proj.bias", &gb));
    RUN(w, ctta_linear_f32_bwd(c.U->ts.emb_silu, c.U->temb_w, c.dtemb_all + temb_of->temb_off, c.U->temb_total, nullptr,
                               nullptr, gw, gb, c.B, temb_of->cout, c.U->temb_dim, 0, 1, w.stream));
  }
  if (!c.dry) CTTA_TRY(wg_end(c, job));
  return CTTA_OK;
}

// linear y = x W^T (+b): x [rows][x_ld] (the first k_rows columns are the GEMM K), dy [rows][N]
static ctta_status linear_wgrad(BCtx& c, const PackMap& m, const PackMap* m2, const bf16_t* x, int x_ld, int64_t rows,
                                const bf16_t* dy, int N) {
  WgJob job;
  CTTA_TRY(wg_begin(c, &job));
  Slabs sl;
  CTTA_TRY(wgrad_slabs(c, job, x, x_ld, 1, (int)rows, 1, false, (int)rows, 1, 1, 1, 1, 0, dy, N, 0, &sl, m2 ? nullptr : &m));
  if (sl.p) CTTA_TRY(scatter_wgrad(job.w, sl, x_ld, m, 0));
  if (m2) CTTA_TRY(scatter_wgrad(job.w, sl, x_ld, *m2, m.n));   // fused [q | k]
  if (!c.dry) CTTA_TRY(wg_end(c, job));
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ data gradients
static ctta_status conv_dgrad(BCtx& c, const ConvLayer& D, const bf16_t* dy, int Hdy, int Wdy, bf16_t* dx, int Hout,
                              int Wout, bool accumulate) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = dy; d.c0 = D.cin_pad;
  d.batch = c.B; d.hi = Hdy; d.wi = Wdy; d.ho = Hout; d.wo = Wout;
  d.kh = D.kh; d.kw = D.kw; d.pad_h = d.pad_w = D.pad;
  d.w = D.p.w; d.k_pad = D.p.k_pad; d.n = D.p.n;
  d.out = dx; d.ldc = D.cout; d.accumulate = accumulate ? 1 : 0;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}
// dx [rows][ldx] (first n_out columns written) (+)= dy[:, :k] * W   with dy row stride dy_ld
static ctta_status linear_dgrad(BCtx& c, const PackedW& D, const bf16_t* dy, int dy_k, int dy_ld, int64_t rows, bf16_t* dx,
                                int n_out, int ldx, bool accumulate) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = dy; d.c0 = dy_k; d.x_stride = dy_ld;
  d.batch = 1; d.hi = (int)rows; d.wi = 1; d.ho = (int)rows; d.wo = 1;
  d.w = D.w; d.k_pad = D.k_pad; d.n = n_out;
  d.out = dx; d.ldc = ldx; d.accumulate = accumulate ? 1 : 0;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

static ctta_status gn_backward(BCtx& c, const GNLayer& g, const bf16_t* x, const bf16_t* dy, bf16_t* dx, int hw,
                               const float* stats, bool silu, bool accumulate_dx) {
  float *dg, *db;
  CTTA_TRY(grad_ptr(c, g.key + "weight", &dg));
  CTTA_TRY(grad_ptr(c, g.key + "bias", &db));
  const int groups = c.U->cfg.norm_num_groups;
  if (!c.dry && ctta_groupnorm_bwd_scratch_floats(c.B, hw, g.c, groups) > c.gn_scratch_floats) {
    ctta_set_error("groupnorm backward scratch too small");
    return CTTA_ERR_INVALID;
  }
  RUN(c, ctta_groupnorm_bwd(x, dy, dx, c.B, hw, g.c, groups, stats, g.gamma, g.beta, silu ? 1 : 0, accumulate_dx ? 1 : 0,
                            dg, db, 1, c.gn_scratch, c.stream));
  return CTTA_OK;
}
// dx = dx_add + dL/dx (dx_add NULL: plain).  The transformer's running token-stream gradient moves to a NEW buffer at every
// LayerNorm: the previous one stays as it was for the weight-gradient jobs that read it in place on the side stream.
static ctta_status ln_backward(BCtx& c, const LNLayer& l, const bf16_t* x, const bf16_t* dy, const bf16_t* dx_add, bf16_t* dx,
                               int64_t rows, int d, int ld) {
  float *dg, *db;
  CTTA_TRY(grad_ptr(c, l.key + "weight", &dg));
  CTTA_TRY(grad_ptr(c, l.key + "bias", &db));
  // the partial table of d gamma / d beta comes from THIS handle's arena and goes back at once: everything that may later
  // be placed on top of it is written by kernels enqueued on c.stream behind the fold
  Arena& A = *c.arena;
  const size_t mk = A.mark();
  const size_t nf = ctta_layernorm_bwd_scratch_floats(rows, ld);
  float* part = nf ? A.get<float>(nf) : nullptr;
  if (nf && !part) { ctta_set_error("arena exhausted (layernorm backward partials)"); return CTTA_ERR_NOMEM; }
  RUN(c, ctta_layernorm_bwd_ws(x, dy, dx_add, dx, rows, d, ld, l.gamma, 1e-5f, dg, db, part, nf, c.stream));
  A.release(mk);
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ resnet
// out = conv2(silu(gn2(t1))) + shortcut(x),  t1 = conv1(silu(gn1(x))) + temb   (resnet.py:549-597)
static ctta_status bwd_resnet(BCtx& c, Resnet& R, const bf16_t* dout, bf16_t** dx_p) {
  Arena& A = *c.arena;
  const Resnet::Saved& S = R.sv;
  const int H = S.H, W = S.W;
  const size_t M = (size_t)c.B * H * W;
  bf16_t* dx = A.get<bf16_t>(M * R.cin); ALLOC_OR_FAIL(dx);
  const size_t mk = A.mark();
  bf16_t* da2 = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(da2);
  CTTA_TRY(conv_dgrad(c, R.t2.d, dout, H, W, da2, H, W, false));
  CTTA_TRY(conv_wgrad(c, R.c2, R.t2.m, S.a2, H, W, false, dout));
  bf16_t* dt1 = A.get<bf16_t>(M * R.cout); ALLOC_OR_FAIL(dt1);
  CTTA_TRY(gn_backward(c, R.n2, S.t1, da2, dt1, H * W, S.st2, true, false));
  CTTA_TRY(conv_wgrad(c, R.c1, R.t1.m, S.a, H, W, false, dt1, R.temb_off, &R));   // + time_emb_proj weight / bias gradients
  bf16_t* da = A.get<bf16_t>(M * R.cin); ALLOC_OR_FAIL(da);
  CTTA_TRY(conv_dgrad(c, R.t1.d, dt1, H, W, da, H, W, false));
  if (R.has_sc) {
    CTTA_TRY(conv_dgrad(c, R.tsc.d, dout, H, W, dx, H, W, false));
    CTTA_TRY(conv_wgrad(c, R.sc, R.tsc.m, S.x, H, W, false, dout));
  } else {
    RUN(c, ctta_add_slices(dout, R.cout, nullptr, 0, dx, R.cin, (int64_t)M, R.cin, c.stream));
  }
  CTTA_TRY(gn_backward(c, R.n1, S.x, da, dx, H * W, S.st1, true, true));
  A.release(mk);
  *dx_p = dx;
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ attention
// q [B][nq][ldq], k [B][krows][ldk] (head h at columns h*64), vt [B][hp][vt_ld]; out / dO [B*nq][hp].
// Writes dq [B*nq][lddq], dk [B*krows][lddk], dv [B*krows][hp] for keys < nk (head-padded lanes come
// out zero).  Flash-style (no score matrix).  Round 5: Q, K, dO are read where they lie and V as the forward keeps it
// (transposed) -- ctta_attention_bwd_inplace takes the operands it needs in the other orientation out of its own LDS tiles
// through transposing reads; the four whole-tensor transposes per call of rounds 1-4 (CTTA_ATTN_BWD_INPLACE=0) are gone.
static constexpr bool attn_bwd_inplace_on() { return true; }
static ctta_status bwd_attention(BCtx& c, int heads, int dh, const bf16_t* q, int ldq, const bf16_t* k, int ldk, int krows,
                                 const bf16_t* vt, int vt_ld, const float* bias, int nq, int nk, const bf16_t* out,
                                 const bf16_t* dO, int hp, const float* lse, bf16_t* dq, int lddq, bf16_t* dk, int lddk,
                                 bf16_t* dv) {
  Arena& A = *c.arena;
  const int B = c.B;
  const int nk64 = round_up(nk, 64), nq64 = round_up(nq, 64);
  const size_t mk = A.mark();
  const bool inplace = attn_bwd_inplace_on() && vt_ld % 8 == 0 && vt_ld >= nk;
  bf16_t *vn = nullptr, *kt = nullptr, *qt = nullptr, *dot = nullptr;
  if (!inplace) {
    vn = A.get<bf16_t>((size_t)B * vt_ld * hp); ALLOC_OR_FAIL(vn);      // V   [B][vt_ld][hp]
    kt = A.get<bf16_t>((size_t)B * hp * nk64); ALLOC_OR_FAIL(kt);       // K^T [B][hp][nk64]
    qt = A.get<bf16_t>((size_t)B * hp * nq64); ALLOC_OR_FAIL(qt);       // Q^T
    dot = A.get<bf16_t>((size_t)B * hp * nq64); ALLOC_OR_FAIL(dot);     // dO^T
  }
  float* dsum = A.get<float>((size_t)B * heads * nq); ALLOC_OR_FAIL(dsum);
  float* part = nullptr;
  int64_t part_floats = 0;
  if (nk <= 128) {   // cross-attention: room for 32 query splits of the dk/dv kernel
    part_floats = (int64_t)32 * 2 * B * krows * hp;
    part = A.get<float>((size_t)part_floats); ALLOC_OR_FAIL(part);
  }
  if (inplace) {
    RUN(c, ctta_attention_bwd_inplace(q, ldq, k, ldk, krows, vt, vt_ld, bias, out, hp, dO, hp, lse, dsum, dq, lddq, dk, lddk, dv, hp,
                                      B, heads, nq, nk, 1.0f / sqrtf((float)dh), part, part_floats, c.stream));
    A.release(mk);
    return CTTA_OK;
  }
  RUN(c, ctta_transpose_bf16(vt, (int64_t)hp * vt_ld, hp, vt_ld, vt_ld, 0, vn, (int64_t)vt_ld * hp, hp, B, c.stream));
  RUN(c, ctta_transpose_bf16(k, (int64_t)krows * ldk, nk, hp, ldk, 0, kt, (int64_t)hp * nk64, nk64, B, c.stream));
  RUN(c, ctta_transpose_bf16(q, (int64_t)nq * ldq, nq, hp, ldq, 0, qt, (int64_t)hp * nq64, nq64, B, c.stream));
  RUN(c, ctta_transpose_bf16(dO, (int64_t)nq * hp, nq, hp, hp, 0, dot, (int64_t)hp * nq64, nq64, B, c.stream));
  RUN(c, ctta_attention_bwd(q, ldq, k, ldk, krows, vn, hp, vt_ld, kt, nk64, qt, dot, nq64, bias, out, hp, dO, hp, lse, dsum,
                            dq, lddq, dk, lddk, dv, hp, B, heads, nq, nk, 1.0f / sqrtf((float)dh), part, part_floats, c.stream));
  A.release(mk);
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ transformer
// transformer_2d.py:218-332 + attention.py:276-334 in reverse (see run_transformer for the forward)
static ctta_status bwd_transformer(BCtx& c, Transformer& T, const bf16_t* dout, bf16_t** dx_p) {
  Arena& A = *c.arena;
  const Transformer::Saved& S = T.sv;
  const int N = S.H * S.W;
  const int64_t M = (int64_t)c.B * N;
  const int cp = T.cp, hp = T.hp, ffp = T.ffp;
  const int vt_ld = round_up(N, 8);
  bf16_t* dx = A.get<bf16_t>((size_t)M * T.c); ALLOC_OR_FAIL(dx);
  const size_t mk = A.mark();
  // out = proj_out(s3) + x
  bf16_t* ds = A.get<bf16_t>((size_t)M * cp); ALLOC_OR_FAIL(ds);   // running gradient of the token stream
  bf16_t* ds_next[3];                                              // ... after ln3, ln2, ln1 (never rewritten in place)
  for (int i = 0; i < 3; ++i) { ds_next[i] = A.get<bf16_t>((size_t)M * cp); ALLOC_OR_FAIL(ds_next[i]); }
  CTTA_TRY(linear_dgrad(c, T.t_proj_out.d, dout, T.c, T.c, M, ds, cp, cp, false));
  CTTA_TRY(linear_wgrad(c, T.t_proj_out.m, nullptr, S.s3, cp, M, dout, T.proj_out.n));
  {  // s3 = ff2(geglu(ff1(ln3(s2)))) + s2
    const size_t m2 = A.mark();
    bf16_t* dgg = A.get<bf16_t>((size_t)M * ffp); ALLOC_OR_FAIL(dgg);
    CTTA_TRY(linear_dgrad(c, T.t_ff2.d, ds, cp, cp, M, dgg, ffp, ffp, false));
    CTTA_TRY(linear_wgrad(c, T.t_ff2.m, nullptr, S.gg, ffp, M, ds, cp));
    bf16_t* df = A.get<bf16_t>((size_t)M * 2 * ffp); ALLOC_OR_FAIL(df);
    RUN(c, ctta_geglu_bwd(S.f, dgg, df, M, ffp, 1, c.stream));
    bf16_t* dn = A.get<bf16_t>((size_t)M * cp); ALLOC_OR_FAIL(dn);
    CTTA_TRY(linear_dgrad(c, T.t_ff1.d, df, 2 * ffp, 2 * ffp, M, dn, cp, cp, false));
    CTTA_TRY(linear_wgrad(c, T.t_ff1.m, nullptr, S.n3, cp, M, df, 2 * ffp));
    CTTA_TRY(ln_backward(c, T.ln3, S.s2, dn, ds, ds_next[0], M, T.inner, cp));
    ds = ds_next[0];
    A.release(m2);
  }
  {  // s2 = out2(attn(q2(ln2(s1)), k2(enc), v2(enc))) + s1
    const size_t m2 = A.mark();
    const int Lp = c.Lp;
    bf16_t* datt = A.get<bf16_t>((size_t)M * hp); ALLOC_OR_FAIL(datt);
    CTTA_TRY(linear_dgrad(c, T.t_out2.d, ds, cp, cp, M, datt, hp, hp, false));
    CTTA_TRY(linear_wgrad(c, T.t_out2.m, nullptr, S.att2, hp, M, ds, cp));
    bf16_t* dq = A.get<bf16_t>((size_t)M * hp); ALLOC_OR_FAIL(dq);
    const size_t kv = (size_t)c.B * Lp * hp;
    bf16_t* dk = A.get<bf16_t>(kv); ALLOC_OR_FAIL(dk);
    bf16_t* dv = A.get<bf16_t>(kv); ALLOC_OR_FAIL(dv);
    if (!c.dry) {   // rows of padded text positions are never written by the per-head GEMMs
      CTTA_CHECK_HIP(ctta_zero_async(dk, kv * sizeof(bf16_t), c.stream));
      CTTA_CHECK_HIP(ctta_zero_async(dv, kv * sizeof(bf16_t), c.stream));
    }
    CTTA_TRY(bwd_attention(c, T.heads, T.dh, S.q2, hp, S.k2, hp, Lp, S.vt2, Lp, c.mask_bias, N, c.L, S.att2, datt, hp, S.lse2,
                           dq, hp, dk, hp, dv));
    bf16_t* dn = A.get<bf16_t>((size_t)M * cp); ALLOC_OR_FAIL(dn);
    CTTA_TRY(linear_dgrad(c, T.t_q2.d, dq, hp, hp, M, dn, cp, cp, false));
    CTTA_TRY(linear_wgrad(c, T.t_q2.m, nullptr, S.n2, cp, M, dq, hp));
    CTTA_TRY(linear_wgrad(c, T.t_k2.m, nullptr, c.enc_bf, c.U->xp, (int64_t)c.B * Lp, dk, hp));
    CTTA_TRY(linear_wgrad(c, T.t_v2.m, nullptr, c.enc_bf, c.U->xp, (int64_t)c.B * Lp, dv, hp));
    CTTA_TRY(ln_backward(c, T.ln2, S.s1, dn, ds, ds_next[1], M, T.inner, cp));
    ds = ds_next[1];
    A.release(m2);
  }
  {  // s1 = out1(attn(q1(n1), k1(n1), v1(n1))) + s0,  n1 = ln1(s0)
    const size_t m2 = A.mark();
    bf16_t* datt = A.get<bf16_t>((size_t)M * hp); ALLOC_OR_FAIL(datt);
    CTTA_TRY(linear_dgrad(c, T.t_out1.d, ds, cp, cp, M, datt, hp, hp, false));
    CTTA_TRY(linear_wgrad(c, T.t_out1.m, nullptr, S.att1, hp, M, ds, cp));
    bf16_t* dqk = A.get<bf16_t>((size_t)M * 2 * hp); ALLOC_OR_FAIL(dqk);
    bf16_t* dv = A.get<bf16_t>((size_t)M * hp); ALLOC_OR_FAIL(dv);
    CTTA_TRY(bwd_attention(c, T.heads, T.dh, S.qk, 2 * hp, S.qk + hp, 2 * hp, N, S.vt, vt_ld, nullptr, N, N, S.att1, datt, hp,
                           S.lse1, dqk, 2 * hp, dqk + hp, 2 * hp, dv));
    bf16_t* dn = A.get<bf16_t>((size_t)M * cp); ALLOC_OR_FAIL(dn);
    CTTA_TRY(linear_dgrad(c, T.t_q1.d, dqk, hp, 2 * hp, M, dn, cp, cp, false));
    CTTA_TRY(linear_dgrad(c, T.t_k1.d, dqk + hp, hp, 2 * hp, M, dn, cp, cp, true));
    CTTA_TRY(linear_dgrad(c, T.t_v1.d, dv, hp, hp, M, dn, cp, cp, true));
    CTTA_TRY(linear_wgrad(c, T.t_q1.m, &T.t_k1.m, S.n1, cp, M, dqk, 2 * hp));
    CTTA_TRY(linear_wgrad(c, T.t_v1.m, nullptr, S.n1, cp, M, dv, hp));
    CTTA_TRY(ln_backward(c, T.ln1, S.s0, dn, ds, ds_next[2], M, T.inner, cp));
    ds = ds_next[2];
    A.release(m2);
  }
  // s0 = proj_in(gn(x))
  bf16_t* dg = A.get<bf16_t>((size_t)M * T.c); ALLOC_OR_FAIL(dg);
  CTTA_TRY(linear_dgrad(c, T.t_proj_in.d, ds, cp, cp, M, dg, T.c, T.c, false));
  CTTA_TRY(linear_wgrad(c, T.t_proj_in.m, nullptr, S.g, T.c, M, ds, cp));
  RUN(c, ctta_add_slices(dout, T.c, nullptr, 0, dx, T.c, M, T.c, c.stream));
  CTTA_TRY(gn_backward(c, T.norm, S.x, dg, dx, N, S.st, false, true));
  A.release(mk);
  *dx_p = dx;
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ embeddings
static ctta_status bwd_embeddings(BCtx& c) {
  ctta_unet* U = c.U;
  Arena& A = *c.arena;
  const ctta_unet::TrainSaved& S = U->ts;
  const int B = c.B, T = U->temb_dim, total = U->temb_total, c0 = U->cfg.block_out_channels[0];
  // temb_all = Linear(SiLU(emb)) over the concatenated time_emb_proj table (its weight gradients were taken per resnet)
  float* demb = A.get<float>((size_t)B * T); ALLOC_OR_FAIL(demb);
  RUN(c, ctta_linear_f32_bwd(S.emb_silu, U->temb_w, c.dtemb_all, total, S.emb, demb, nullptr, nullptr, B, total, T, 0, 0,
                             c.stream));
  // emb = linear_2(SiLU(linear_1(feat)))  for the time (and guidance) branch; d emb feeds both
  auto mlp = [&](const std::string& name, const float* feat, int k, const float* hpre, const float* hid, const float* w1,
                 const float* w2) -> ctta_status {
    float *gw1, *gb1, *gw2, *gb2;
    CTTA_TRY(grad_ptr(c, name + "linear_1.weight", &gw1));
    CTTA_TRY(grad_ptr(c, name + "linear_1.bias", &gb1));
    CTTA_TRY(grad_ptr(c, name + "linear_2.weight", &gw2));
    CTTA_TRY(grad_ptr(c, name + "linear_2.bias", &gb2));
    float* dh = A.get<float>((size_t)B * T); ALLOC_OR_FAIL(dh);
    RUN(c, ctta_linear_f32_bwd(hid, w2, demb, T, hpre, dh, gw2, gb2, B, T, T, 0, 1, c.stream));
    RUN(c, ctta_linear_f32_bwd(feat, w1, dh, T, nullptr, nullptr, gw1, gb1, B, T, k, 0, 1, c.stream));
    return CTTA_OK;
  };
  CTTA_TRY(mlp("time_embedding.", S.tfeat, c0, S.t_h1pre, S.t_h1, U->t_w1, U->t_w2));
  if (U->cfg.guided) CTTA_TRY(mlp("guidance_embedding.", S.gfeat, T, S.g_h1pre, S.g_h1, U->g_w1, U->g_w2));
  return CTTA_OK;
}

// ------------------------------------------------------------------------------ whole network
// The backward pass is resumable block by block (out head, up blocks, mid block, down blocks, then conv_in + the
// embedding MLPs): after each step all parameter gradients of that block are final, so the caller can start the
// data-parallel all-reduce of that block's slice of the flat gradient buffer while the next block computes.
static void bctx_init(BCtx& c, ctta_unet* U, bool dry, const GradTable* grads, hipStream_t stream) {
  const ctta_unet::TrainSaved& S = U->ts;
  c.arena = &U->arena; c.stream = stream; c.dry = dry; c.taps = nullptr;
  c.gn_scratch = U->gn_scratch; c.gn_scratch_floats = U->gn_scratch_floats;
  c.U = U; c.B = S.B; c.L = S.L; c.Lp = S.Lp; c.train = true;
  c.enc_bf = S.enc_bf; c.mask_bias = S.mask_bias; c.grads = grads;
  c.dtemb_all = U->bw.dtemb_all;
}

static ctta_status unet_backward_begin_impl(ctta_unet* U, bool dry, const bf16_t* dpred, const GradTable* grads,
                                            hipStream_t stream) {
  const ctta_unet_config& cfg = U->cfg;
  const ctta_unet::TrainSaved& S = U->ts;
  Arena& A = U->arena;
  A.off = S.arena_off;
  // With the weight-gradient side stream on, a linear layer's dY is read in place by the SIDE stream some time after the main
  // stream has moved on: nothing the backward allocates is released before the next training forward resets the arena
  // (the dry run sizes it under the same policy; a few GB more at batch 9 -- HBM is what this part has).
  A.no_release = U->wg.enabled && wgrad_inplace_on();
  const int B = S.B, H = cfg.height, W = cfg.width, c0 = cfg.block_out_channels[0];
  U->bw.dtemb_all = A.get<float>((size_t)B * U->temb_total); ALLOC_OR_FAIL(U->bw.dtemb_all);
  U->bw.dskip.assign((size_t)S.n_skips, nullptr);
  U->bw.pos = U->tape.size();
  BCtx c;
  bctx_init(c, U, dry, grads, stream);
  if (!dry) {
    // Every earlier backward ended with wg_join: when that join was on THIS stream, everything the side stream did for
    // it is ordered before what this call enqueues, so no scratch slot is "in use" any more.  Without this the first
    // jobs of a backward waited on `freed` events recorded by the previous one -- harmless eagerly, but inside a hipGraph
    // capture that is a dependency on uncaptured work (hipErrorStreamCaptureIsolation).
    // The same when this call is the first one inside / after a capture: events recorded on the other side of that
    // boundary must not be waited on (torch's graph context synchronises the device when it opens and closes).
    ctta_unet::WgradSide& Wg = U->wg;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cs);
    const bool capturing = cs == hipStreamCaptureStatusActive;
    if (!Wg.dirty && (Wg.joined_on == stream || capturing != Wg.was_capturing))
      for (int i = 0; i < ctta_unet::WgradSide::NS; ++i) Wg.in_use[i] = false;
    Wg.was_capturing = capturing;
  }
  // ---- conv_out, conv_norm_out
  const size_t M0 = (size_t)B * H * W;
  bf16_t* dh = A.get<bf16_t>(M0 * c0); ALLOC_OR_FAIL(dh);
  const size_t mk = A.mark();
  bf16_t* da = A.get<bf16_t>(M0 * c0); ALLOC_OR_FAIL(da);
  CTTA_TRY(conv_dgrad(c, U->t_conv_out.d, dpred, H, W, da, H, W, false));
  ConvLayer co;   // geometry of conv_out for the weight gradient (the forward runs it on the small-N kernel)
  co.cin_pad = c0; co.cout = cfg.out_channels; co.kh = co.kw = 3; co.stride = 1; co.pad = 1; co.p.n = 8;
  CTTA_TRY(conv_wgrad(c, co, U->t_conv_out.m, S.a_out, H, W, false, dpred));
  CTTA_TRY(gn_backward(c, U->norm_out, S.h_last, da, dh, H * W, S.st_out, true, false));
  A.release(mk);
  CTTA_TRY(wg_join(c));   // the out head's gradients are final when this call returns
  U->bw.dh = dh;
  U->bw.active = true;
  return CTTA_OK;
}

// replays the tape entries of ONE block (all entries at the current position that share its block id); the last
// step also runs the embedding MLPs.  *block_done = id of the finished block, *finished = 1 after the last step.
static ctta_status unet_backward_next_impl(ctta_unet* U, bool dry, const GradTable* grads, hipStream_t stream,
                                           int* block_done, int* finished) {
  const ctta_unet::TrainSaved& S = U->ts;
  Arena& A = U->arena;
  BCtx c;
  bctx_init(c, U, dry, grads, stream);
  if (!dry) {
    // begin / every earlier next ended with wg_join on this stream: no scratch slot is in use any more, and the `freed`
    // events of the previous call must not be waited on when this call is captured into its OWN hipGraph (segmented
    // capture of the data-parallel step: they were recorded in another capture).
    ctta_unet::WgradSide& Wg = U->wg;
    if (!Wg.dirty && Wg.joined_on == stream)
      for (int i = 0; i < ctta_unet::WgradSide::NS; ++i) Wg.in_use[i] = false;
  }
  const int B = S.B;
  bf16_t* dh = U->bw.dh;
  std::vector<bf16_t*>& dskip = U->bw.dskip;
  CTTA_REQUIRE(U->bw.pos > 0, "unet_backward_next: nothing left to replay");
  const int block = U->tape[U->bw.pos - 1].block;
  while (U->bw.pos > 0 && U->tape[U->bw.pos - 1].block == block) {
    TapeOp& op = U->tape[--U->bw.pos];
    const size_t M = (size_t)B * op.H * op.W;
    switch (op.kind) {
      case TapeOp::RESNET: CTTA_TRY(bwd_resnet(c, *op.R, dh, &dh)); break;
      case TapeOp::TRANSFORMER: CTTA_TRY(bwd_transformer(c, *op.T, dh, &dh)); break;
      case TapeOp::UPSAMPLE: {   // y = conv(nearest_x2(x)): dx = 2x2 sum-pool of the conv data gradient
        bf16_t* dx = A.get<bf16_t>(M * op.ch); ALLOC_OR_FAIL(dx);
        const size_t mk = A.mark();
        bf16_t* dup = A.get<bf16_t>(M * 4 * op.ch); ALLOC_OR_FAIL(dup);
        CTTA_TRY(conv_dgrad(c, op.Lv->tsampler.d, dh, 2 * op.H, 2 * op.W, dup, 2 * op.H, 2 * op.W, false));
        RUN(c, ctta_pool2_sum(dup, dx, B, op.H, op.W, op.ch, 0, stream));
        CTTA_TRY(conv_wgrad(c, op.Lv->sampler, op.Lv->tsampler.m, op.x, op.H, op.W, true, dh));
        A.release(mk);
        dh = dx;
        break;
      }
      case TapeOp::DOWNSAMPLE: {   // stride-2 conv: zero-insert dY, then a stride-1 data-gradient conv
        bf16_t* dx = A.get<bf16_t>(M * op.ch); ALLOC_OR_FAIL(dx);
        const size_t mk = A.mark();
        const int ho = (op.H + 2 - 3) / 2 + 1, wo = (op.W + 2 - 3) / 2 + 1;
        bf16_t* dz = A.get<bf16_t>(M * op.ch); ALLOC_OR_FAIL(dz);
        RUN(c, ctta_zero_insert2(dh, dz, B, ho, wo, op.H, op.W, op.ch, stream));
        CTTA_TRY(conv_dgrad(c, op.Lv->tsampler.d, dz, op.H, op.W, dx, op.H, op.W, false));
        CTTA_TRY(conv_wgrad(c, op.Lv->sampler, op.Lv->tsampler.m, op.x, op.H, op.W, false, dh));
        A.release(mk);
        dh = dx;
        break;
      }
      case TapeOp::CONCAT: {   // torch.cat([h, skip], 1): split the gradient
        bf16_t* d0 = A.get<bf16_t>(M * op.ch); ALLOC_OR_FAIL(d0);
        bf16_t* d1 = A.get<bf16_t>(M * op.skc); ALLOC_OR_FAIL(d1);
        const int ld = op.ch + op.skc;
        RUN(c, ctta_add_slices(dh, ld, nullptr, 0, d0, op.ch, (int64_t)M, op.ch, stream));
        RUN(c, ctta_add_slices(dh + op.ch, ld, nullptr, 0, d1, op.skc, (int64_t)M, op.skc, stream));
        dskip[op.skip_idx] = d1;
        dh = d0;
        break;
      }
      case TapeOp::SKIP_PUSH: {
        bf16_t* ds = dskip[op.skip_idx];
        CTTA_REQUIRE(ds, "internal: skip %d has no gradient", op.skip_idx);
        RUN(c, ctta_add_slices(dh, op.ch, ds, op.ch, dh, op.ch, (int64_t)M, op.ch, stream));
        break;
      }
      case TapeOp::CONV_IN:
        CTTA_TRY(conv_wgrad(c, U->conv_in, U->t_conv_in.m, op.x, op.H, op.W, false, dh));
        break;
    }
  }
  U->bw.dh = dh;
  *block_done = block;
  *finished = 0;
  CTTA_TRY(wg_join(c));     // this block's weight gradients (and d temb rows) are complete
  if (U->bw.pos == 0) {
    CTTA_TRY(bwd_embeddings(c));
    *finished = 1;
    U->bw.active = false;
  }
  return CTTA_OK;
}

static ctta_status unet_backward_impl(ctta_unet* U, bool dry, const bf16_t* dpred, const GradTable* grads,
                                      hipStream_t stream, size_t* gn_need) {
  CTTA_TRY(unet_backward_begin_impl(U, dry, dpred, grads, stream));
  int block = 0, fin = 0;
  while (!fin) CTTA_TRY(unet_backward_next_impl(U, dry, grads, stream, &block, &fin));
  if (gn_need) *gn_need = 0;
  return CTTA_OK;
}

extern "C" ctta_status ctta_unet_forward_train(ctta_unet* U, const float* sample, const float* timesteps,
                                               const double* guidance, const float* enc, const uint8_t* mask, int batch,
                                               int text_len, float* out, void* stream) {
  CTTA_REQUIRE(U && sample && timesteps && enc && out, "unet_forward_train: null pointer");
  CTTA_REQUIRE(U->cfg.enable_training, "unet_forward_train: handle was created without enable_training");
  CTTA_REQUIRE(!U->cfg.guided || guidance, "unet_forward_train: guidance is required by the guided U-Net");
  CTTA_REQUIRE(batch >= 1 && batch <= U->cfg.max_batch, "unet_forward_train: batch %d outside [1,%d]", batch, U->cfg.max_batch);
  CTTA_REQUIRE(text_len >= 1 && text_len <= U->cfg.max_text_len, "unet_forward_train: text_len %d outside [1,%d]", text_len,
               U->cfg.max_text_len);
  WsBind bind(U->splitws);
  return unet_forward_impl(U, false, sample, timesteps, guidance, enc, mask, batch, text_len, out, (hipStream_t)stream,
                           nullptr, true);
}

extern "C" ctta_status ctta_unet_backward(ctta_unet* U, const void* dout_nhwc, const ctta_tensor* grads, int n_grads,
                                          void* stream) {
  CTTA_REQUIRE(U && dout_nhwc && grads, "unet_backward: null pointer");
  CTTA_REQUIRE(U->cfg.enable_training, "unet_backward: handle was created without enable_training");
  CTTA_REQUIRE(U->ts.valid, "unet_backward: no training forward to differentiate (call ctta_unet_forward_train first)");
  GradTable gt;
  gt.build(grads, n_grads);
  WsBind bind(U->splitws);
  const ctta_status st = unet_backward_impl(U, false, (const bf16_t*)dout_nhwc, &gt, (hipStream_t)stream, nullptr);
  U->ts.valid = false;   // one backward per training forward
  return st;
}

// Block-wise variant for overlapping the data-parallel gradient all-reduce with the backward pass:
//   begin (out head) -> next ... next until *finished.  Block ids: 0 = conv_in + embedding MLPs, 1..n = down blocks,
//   n+1 = mid block, n+2..2n+1 = up blocks, 2n+2 = conv_norm_out + conv_out (final after `begin`).
extern "C" ctta_status ctta_unet_backward_begin(ctta_unet* U, const void* dout_nhwc, const ctta_tensor* grads, int n_grads,
                                                void* stream) {
  CTTA_REQUIRE(U && dout_nhwc && grads, "unet_backward_begin: null pointer");
  CTTA_REQUIRE(U->cfg.enable_training, "unet_backward_begin: handle was created without enable_training");
  CTTA_REQUIRE(U->ts.valid, "unet_backward_begin: no training forward to differentiate (call ctta_unet_forward_train first)");
  GradTable gt;
  gt.build(grads, n_grads);
  U->ts.valid = false;
  WsBind bind(U->splitws);
  return unet_backward_begin_impl(U, false, (const bf16_t*)dout_nhwc, &gt, (hipStream_t)stream);
}
extern "C" ctta_status ctta_unet_backward_next(ctta_unet* U, const ctta_tensor* grads, int n_grads, void* stream,
                                               int* block_done, int* finished) {
  CTTA_REQUIRE(U && grads && block_done && finished, "unet_backward_next: null pointer");
  CTTA_REQUIRE(U->bw.active, "unet_backward_next: call ctta_unet_backward_begin first");
  GradTable gt;
  gt.build(grads, n_grads);
  WsBind bind(U->splitws);
  return unet_backward_next_impl(U, false, &gt, (hipStream_t)stream, block_done, finished);
}
