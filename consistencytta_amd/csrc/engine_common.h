// Host-side runtime shared by the three engines: device arenas, the state-dict weight table,
// weight packing into the conv_gemm layouts, and thin layer launchers.
#pragma once
#include <stdlib.h>

#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

// ---------------------------------------------------------------------------------- arenas
// Stack (bump) allocator over one hipMalloc.  `dry` arenas only measure (sizing pass).
struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool dry = true;
  bool no_release = false;   // debug taps keep every intermediate alive

  void* alloc(size_t bytes) {
    const size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    if (off > peak) peak = off;
    if (dry) return reinterpret_cast<void*>((uintptr_t)0x1000 + a);   // never dereferenced
    if (off > cap) return nullptr;
    return base + a;
  }
  template <typename T>
  T* get(size_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
  size_t mark() const { return off; }
  void release(size_t m) { if (!no_release) off = m; }
  void reset() { off = 0; }
};

// Per-handle split-K workspace of conv_gemm: allocated with the handle, bound to the calling host thread while one of
// the handle's entry points enqueues work (WsBind), so two handles on two streams never share partial-sum slabs.
struct SplitWs {
  void* p = nullptr;
  size_t bytes = 0;
  ctta_status init() {
    bytes = ctta_conv_workspace_bytes();
    if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; ctta_set_error("hipMalloc of the split-K workspace failed"); return CTTA_ERR_NOMEM; }
    // the stream-K header (ticket / finished / epoch words and the flags) starts at zero; the launches keep it consistent
    if (hipMemset(p, 0, ctta_conv_workspace_header_bytes()) != hipSuccess) { destroy(); ctta_set_error("hipMemset of the stream-K header failed"); return CTTA_ERR_HIP; }
    return CTTA_OK;
  }
  void destroy() { if (p) (void)hipFree(p); p = nullptr; }
};
struct WsBind {   // re-entrant: the previous binding of the thread (a raw caller's, or an outer handle's) comes back
  void* prev = nullptr;
  size_t prev_bytes = 0;
  int prev_hdr = 0;
  explicit WsBind(const SplitWs& w) {
    ctta_conv_bound_workspace(&prev, &prev_bytes);
    prev_hdr = ctta_conv_bound_workspace_header();
    ctta_conv_bind_workspace_ex(w.p, w.bytes, 1);
  }
  ~WsBind() { ctta_conv_bind_workspace_ex(prev, prev_bytes, prev_hdr); }
  WsBind(const WsBind&) = delete;
  WsBind& operator=(const WsBind&) = delete;
};

struct Tap {
  std::string name;
  const void* ptr;
  int b, c, h, w, c_stride;
  bool f32_nchw;
};

// ---------------------------------------------------------------------------------- weights
struct WeightTable {
  std::unordered_map<std::string, const ctta_tensor*> map;
  void build(const ctta_tensor* w, int n) {
    map.clear();
    for (int i = 0; i < n; ++i) map[w[i].name] = &w[i];
  }
  const ctta_tensor* find(const std::string& k) const {
    auto it = map.find(k);
    return it == map.end() ? nullptr : it->second;
  }
};

static inline int64_t tensor_numel(const ctta_tensor* t) {
  int64_t n = 1;
  for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
  return n;
}

// A packed bf16 GEMM operand [n_rows][k_pad] plus its fp32 bias (owned copies).
struct PackedW {
  bf16_t* w = nullptr;
  float* bias = nullptr;
  int n = 0;       // valid rows handed to conv_gemm (multiple of 4)
  int k_pad = 0;
};

// Where a packed operand came from, kept for the backward pass: the inverse of the packing
// (ctta_wgrad_scatter) adds a [K][N] gradient slab back into the reference parameter layout.
struct PackMap {
  std::string wkey, bkey;
  int32_t *ro = nullptr, *co = nullptr;   // device copies of the pack maps (row_off[n], col_off[k])
  int32_t* bidx = nullptr;                // bias placement (NULL = identity over n_bias entries)
  int n = 0;                              // packed rows (GEMM N, padding included)
  int n_bias = 0;
  int k_ident = 0;                        // conv: weight rows are (cin, kh, kw)-contiguous = the row order of
                                          // ctta_im2col_t, so the gradient scatters with the identity column map
                                          // over the first k_ident slab rows
};

// Persistent device storage for packed weights / fp32 copies / index maps.
// (Re)loading a state dict = two table-driven launches: every bf16 pack job in one kernel, every fp32
// vector / table copy in another (ctta_pack_weight_multi / ctta_copy_segments_multi).  The job tables
// live on the device and are only re-uploaded when a source pointer changed -- in training the
// parameters sit in one flat buffer, so the per-step re-pack after AdamW / EMA costs two launches.
struct WeightStore {
  Arena arena;
  std::vector<std::function<ctta_status(const WeightTable&, hipStream_t)>> jobs;   // irregular one-offs

  struct PackSpec {
    std::string key;
    std::vector<int64_t> shape;
    const int32_t *ro, *co, *ra, *ca;
    int aux_limit, n_rows, k_pad;
    bf16_t* dst;
    int src_row_len;   // > 0: rows are contiguous source ranges (row-staged pack kernel), 0: generic gather
  };
  struct CopySpec {
    std::string key;
    int64_t numel;      // expected element count of the source tensor
    int src_start, count;
    float* dst;
  };
  std::vector<PackSpec> packs;
  std::vector<CopySpec> copies;
  std::vector<ctta_tpose_job> tposes;   // bf16 -> bf16 derivations (data-gradient operands); static pointers
  ctta_tpose_job* d_tpose = nullptr;
  int tpose_blocks = 0;
  bool tpose_uploaded = false;
  std::vector<ctta_pack_job> h_pack, h_pack_prev;
  // jobs of the row-staged kernel, one table per LDS class: [0] rows of <= 9216 floats (36 KB: four 256-thread
  // blocks per CU), [1] longer rows (one 1024-thread block per CU)
  static constexpr int kRowsLdsSmall = 9216, kRowsLdsLarge = 36 * 1024;
  struct RowClass {
    std::vector<ctta_pack_job> h, h_prev;
    ctta_pack_job* d = nullptr;
    size_t cap = 0;
    int blocks = 0, lds_floats = 0;
  } rowc[2];
  std::vector<ctta_copy_seg> h_copy, h_copy_prev;
  ctta_pack_job* d_pack = nullptr;
  ctta_copy_seg* d_copy = nullptr;
  size_t d_pack_cap = 0, d_copy_cap = 0;
  int pack_blocks = 0;

  ctta_status init(size_t bytes) {
    arena.dry = false;
    arena.cap = bytes;
    CTTA_CHECK_HIP(hipMalloc((void**)&arena.base, bytes));
    CTTA_CHECK_HIP(hipMemset(arena.base, 0, bytes));   // padding of the fp32 vectors stays zero for good
    return CTTA_OK;
  }
  void destroy() {
    if (arena.base) (void)hipFree(arena.base);
    if (d_pack) (void)hipFree(d_pack);
    for (RowClass& c : rowc) { if (c.d) (void)hipFree(c.d); c.d = nullptr; c.cap = 0; }
    if (d_copy) (void)hipFree(d_copy);
    if (d_tpose) (void)hipFree(d_tpose);
    arena.base = nullptr; d_pack = nullptr; d_copy = nullptr; d_tpose = nullptr;
  }

  static ctta_status find_checked(const WeightTable& wt, const std::string& key, const ctta_tensor** out) {
    const ctta_tensor* t = wt.find(key);
    if (!t) { ctta_set_error("missing state-dict key '%s'", key.c_str()); return CTTA_ERR_MISSING_KEY; }
    *out = t;
    return CTTA_OK;
  }

  ctta_status run_all(const WeightTable& wt, hipStream_t s) {
    // ---- bf16 packs
    h_pack.clear();
    for (RowClass& c : rowc) { c.h.clear(); c.blocks = 0; c.lds_floats = 0; }
    int blk = 0;
    for (const PackSpec& p : packs) {
      const ctta_tensor* t;
      CTTA_TRY(find_checked(wt, p.key, &t));
      if (t->ndim != (int)p.shape.size()) {
        ctta_set_error("key '%s': rank %d, expected %d", p.key.c_str(), t->ndim, (int)p.shape.size());
        return CTTA_ERR_INVALID;
      }
      for (int i = 0; i < t->ndim; ++i)
        if (t->shape[i] != p.shape[i]) {
          ctta_set_error("size mismatch for '%s' dim %d: %lld vs expected %lld", p.key.c_str(), i,
                         (long long)t->shape[i], (long long)p.shape[i]);
          return CTTA_ERR_INVALID;
        }
      ctta_pack_job j;
      memset(&j, 0, sizeof(j));   // tables are compared bytewise: no indeterminate padding
      j.src = t->data; j.row_off = p.ro; j.col_off = p.co; j.row_aux = p.ra; j.col_aux = p.ca;
      j.aux_limit = p.aux_limit; j.n_rows = p.n_rows; j.k_pad = p.k_pad; j.dst = p.dst;
      j.src_row_len = p.src_row_len;
      if (p.src_row_len > 0 && p.src_row_len <= kRowsLdsLarge) {   // source rows staged in LDS
        RowClass& c = rowc[p.src_row_len <= kRowsLdsSmall ? 0 : 1];
        const int cap = p.src_row_len <= kRowsLdsSmall ? kRowsLdsSmall : kRowsLdsLarge;
        int rpb = cap / p.src_row_len;
        if (rpb > 64) rpb = 64;
        if (rpb > p.n_rows) rpb = p.n_rows;
        j.rows_per_block = rpb;
        j.block0 = c.blocks;
        c.blocks += (p.n_rows + rpb - 1) / rpb;
        if (rpb * p.src_row_len > c.lds_floats) c.lds_floats = rpb * p.src_row_len;
        c.h.push_back(j);
        continue;
      }
      j.block0 = blk;
      blk += (int)(((int64_t)p.n_rows * p.k_pad + CTTA_PACK_ELEMS_PER_BLOCK - 1) / CTTA_PACK_ELEMS_PER_BLOCK);
      h_pack.push_back(j);
    }
    pack_blocks = blk;
    // ---- fp32 copies, chunked so that one block moves at most CTTA_COPY_ELEMS_PER_BLOCK elements
    h_copy.clear();
    for (const CopySpec& c : copies) {
      const ctta_tensor* t;
      CTTA_TRY(find_checked(wt, c.key, &t));
      if (tensor_numel(t) != c.numel) {
        ctta_set_error("size mismatch for '%s': %lld elements, expected %lld", c.key.c_str(),
                       (long long)tensor_numel(t), (long long)c.numel);
        return CTTA_ERR_INVALID;
      }
      for (int off = 0; off < c.count; off += CTTA_COPY_ELEMS_PER_BLOCK) {
        ctta_copy_seg g;
        memset(&g, 0, sizeof(g));
        g.src = t->data + c.src_start + off; g.dst = c.dst + off;
        g.count = c.count - off < CTTA_COPY_ELEMS_PER_BLOCK ? c.count - off : CTTA_COPY_ELEMS_PER_BLOCK;
        h_copy.push_back(g);
      }
    }
    CTTA_TRY(sync_table(h_pack, h_pack_prev, (void**)&d_pack, &d_pack_cap, sizeof(ctta_pack_job)));
    for (int ci = 0; ci < 2; ++ci) {
      RowClass& c = rowc[ci];
      CTTA_TRY(sync_table(c.h, c.h_prev, (void**)&c.d, &c.cap, sizeof(ctta_pack_job)));
      if (!c.h.empty())
        CTTA_TRY(ctta_pack_weight_rows_multi(c.d, (int)c.h.size(), c.blocks, c.lds_floats, ci == 0 ? 256 : 1024, s));
    }
    CTTA_TRY(sync_table(h_copy, h_copy_prev, (void**)&d_copy, &d_copy_cap, sizeof(ctta_copy_seg)));
    if (!h_pack.empty()) CTTA_TRY(ctta_pack_weight_multi(d_pack, (int)h_pack.size(), pack_blocks, s));
    if (!h_copy.empty()) CTTA_TRY(ctta_copy_segments_multi(d_copy, (int)h_copy.size(), s));
    if (!tposes.empty()) {   // after the forward packs: their transposes / tap rotations
      if (!tpose_uploaded) {
        CTTA_CHECK_HIP(hipMalloc((void**)&d_tpose, tposes.size() * sizeof(ctta_tpose_job)));
        CTTA_CHECK_HIP(hipMemcpy(d_tpose, tposes.data(), tposes.size() * sizeof(ctta_tpose_job), hipMemcpyHostToDevice));
        tpose_uploaded = true;
      }
      CTTA_TRY(ctta_transpose_multi(d_tpose, (int)tposes.size(), tpose_blocks, s));
    }
    for (auto& j : jobs) CTTA_TRY(j(wt, s));
    return CTTA_OK;
  }

  // dst[c][r] = src[r][c] (r < rows, c < cols), columns [rows, wcols) of each dst row zero; runs after the packs
  void add_transpose(const bf16_t* src, bf16_t* dst, int rows, int cols, int src_ld, int dst_ld, int wcols) {
    ctta_tpose_job j;
    memset(&j, 0, sizeof(j));
    j.src = src; j.dst = dst; j.rows = rows; j.cols = cols; j.src_ld = src_ld; j.dst_ld = dst_ld; j.wcols = wcols;
    j.block0 = tpose_blocks; j.tiles_r = (wcols + 63) / 64;
    tpose_blocks += j.tiles_r * ((cols + 63) / 64);
    tposes.push_back(j);
  }

  // uploads a job table when it differs from what the device already holds (blocking copy: tables are a
  // few hundred KB and change only when the caller hands over different tensors)
  template <typename T>
  static ctta_status sync_table(const std::vector<T>& cur, std::vector<T>& prev, void** dev, size_t* cap, size_t elem) {
    if (cur.empty()) return CTTA_OK;
    const size_t bytes = cur.size() * elem;
    if (*dev && prev.size() == cur.size() && memcmp(prev.data(), cur.data(), bytes) == 0) return CTTA_OK;
    if (bytes > *cap) {
      CTTA_CHECK_HIP(hipDeviceSynchronize());
      if (*dev) (void)hipFree(*dev);
      *dev = nullptr;
      CTTA_CHECK_HIP(hipMalloc(dev, bytes));
      *cap = bytes;
    } else {
      CTTA_CHECK_HIP(hipDeviceSynchronize());   // an earlier launch may still be reading the old table
    }
    CTTA_CHECK_HIP(hipMemcpy(*dev, cur.data(), bytes, hipMemcpyHostToDevice));
    prev = cur;
    return CTTA_OK;
  }

  // uploads a host int vector once (index maps are structural, not weight dependent)
  ctta_status upload(const std::vector<int32_t>& v, int32_t** out) {
    int32_t* d = arena.get<int32_t>(v.size());
    if (!d) { ctta_set_error("weight store exhausted"); return CTTA_ERR_NOMEM; }
    CTTA_CHECK_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    *out = d;
    return CTTA_OK;
  }

  // Generic packed matrix: dst[r][k] = src[row_off[r] + col_off[k]] (or 0).
  ctta_status add_matrix(const std::string& key, const std::vector<int64_t>& expect_shape,
                         const std::vector<int32_t>& row_off, const std::vector<int32_t>& col_off,
                         const std::vector<int32_t>* row_aux, const std::vector<int32_t>* col_aux,
                         int aux_limit, bf16_t** out, bf16_t* dst_preallocated = nullptr,
                         PackMap* pm = nullptr) {
    const int n_rows = (int)row_off.size(), k_pad = (int)col_off.size();
    bf16_t* dst = dst_preallocated ? dst_preallocated : arena.get<bf16_t>((size_t)n_rows * k_pad);
    if (!dst) { ctta_set_error("weight store exhausted at %s", key.c_str()); return CTTA_ERR_NOMEM; }
    int32_t *dro, *dco, *dra = nullptr, *dca = nullptr;
    CTTA_TRY(upload(row_off, &dro));
    CTTA_TRY(upload(col_off, &dco));
    if (aux_limit > 0) { CTTA_TRY(upload(*row_aux, &dra)); CTTA_TRY(upload(*col_aux, &dca)); }
    *out = dst;
    if (pm) { pm->wkey = key; pm->ro = dro; pm->co = dco; pm->n = n_rows; }
    // rows that are contiguous source ranges (every plain conv / linear weight: row r = src[r * rowlen ...]) take the
    // row-staged kernel; anything else (the ConvTranspose phase packs) the generic gather
    int src_row_len = 0;
    if (!expect_shape.empty() && expect_shape[0] > 0 && k_pad % 8 == 0) {
      int64_t numel = 1;
      for (int64_t d : expect_shape) numel *= d;
      const int64_t rowlen = numel / expect_shape[0];
      bool ok = rowlen > 0 && rowlen < (1 << 30);
      for (int32_t c : col_off) ok = ok && c < rowlen;
      for (int32_t r : row_off) ok = ok && (r < 0 || (r % rowlen == 0 && r + rowlen <= numel));
      if (ok) src_row_len = (int)rowlen;
    }
    packs.push_back({key, expect_shape, dro, dco, dra, dca, aux_limit, n_rows, k_pad, dst, src_row_len});
    return CTTA_OK;
  }

  // fp32 vector copy with optional placement: dst has n_pad entries, src[i] lands at dst_pos(i)
  // given as (count, dst_start) segments; the rest stays zero (the store is zero-filled at init).
  struct Seg { int src_start, dst_start, count; };
  ctta_status add_vector(const std::string& key, int src_len, int n_pad, const std::vector<Seg>& segs,
                         float** out) {
    float* dst = arena.get<float>((size_t)n_pad);
    if (!dst) { ctta_set_error("weight store exhausted at %s", key.c_str()); return CTTA_ERR_NOMEM; }
    *out = dst;
    for (const Seg& g : segs) copies.push_back({key, (int64_t)src_len, g.src_start, g.count, dst + g.dst_start});
    return CTTA_OK;
  }
  ctta_status add_vector(const std::string& key, int len, float** out) {
    return add_vector(key, len, len, {{0, 0, len}}, out);
  }
  // copies into a caller-chosen place of a bigger owned fp32 buffer (concatenated tables)
  ctta_status add_copy_into(const std::string& key, int64_t numel, float* dst) {
    copies.push_back({key, numel, 0, (int)numel, dst});
    return CTTA_OK;
  }
};

// ---------------------------------------------------------------------------------- conv layers
struct ConvLayer {
  PackedW p;
  int cin_pad = 0;   // channels the packed K layout assumes per tap (c0 + c1 at run time)
  int cout = 0;
  int kh = 1, kw = 1, stride = 1, pad = 0, dil_w = 1;
};

// conv2d weight (cout, cin, kh, kw) -> [rows][ (kh,kw,c) ] with cin padded to cin_pad.
// rows: n_rows_pad >= cout entries; row r maps to source row row_map[r] (or -1).
static inline ctta_status make_conv(WeightStore& ws, const std::string& prefix, int cout, int cin,
                                    int cin_pad, int kh, int kw, int stride, int pad, ConvLayer* L,
                                    bool with_bias = true, PackMap* pm = nullptr) {
  const int K = kh * kw * cin_pad;
  const int k_pad = round_up(K, 64);
  const int n_pad = round_up(cout, 4);
  std::vector<int32_t> ro(n_pad, -1), co(k_pad, -1);
  for (int r = 0; r < cout; ++r) ro[r] = r * cin * kh * kw;
  for (int y = 0; y < kh; ++y)
    for (int x = 0; x < kw; ++x)
      for (int c = 0; c < cin; ++c) co[(y * kw + x) * cin_pad + c] = c * kh * kw + y * kw + x;
  const std::vector<int64_t> shape = {cout, cin, kh, kw};
  CTTA_TRY(ws.add_matrix(prefix + "weight", shape, ro, co, nullptr, nullptr, 0, &L->p.w, nullptr, pm));
  if (with_bias) CTTA_TRY(ws.add_vector(prefix + "bias", cout, n_pad, {{0, 0, cout}}, &L->p.bias));
  if (pm && with_bias) { pm->bkey = prefix + "bias"; pm->n_bias = cout; }
  if (pm) pm->k_ident = cin * kh * kw;
  L->p.n = n_pad; L->p.k_pad = k_pad;
  L->cin_pad = cin_pad; L->cout = cout; L->kh = kh; L->kw = kw; L->stride = stride; L->pad = pad;
  return CTTA_OK;
}

// Linear weight (n_src, k_src) with row/column placement maps.
//   row_map[r] = source row or -1 ; col_map[k] = source col or -1.
static inline ctta_status make_linear(WeightStore& ws, const std::string& wkey, const std::string& bkey,
                                      int n_src, int k_src, const std::vector<int32_t>& row_map,
                                      const std::vector<int32_t>& col_map, PackedW* P,
                                      bf16_t* dst_preallocated = nullptr, PackMap* pm = nullptr) {
  const int n_pad = (int)row_map.size();
  const int k_pad = (int)col_map.size();
  std::vector<int32_t> ro(n_pad), co(k_pad);
  for (int r = 0; r < n_pad; ++r) ro[r] = row_map[r] < 0 ? -1 : row_map[r] * k_src;
  for (int k = 0; k < k_pad; ++k) co[k] = col_map[k];
  CTTA_TRY(ws.add_matrix(wkey, {n_src, k_src}, ro, co, nullptr, nullptr, 0, &P->w, dst_preallocated, pm));
  if (pm && !bkey.empty()) {
    pm->bkey = bkey; pm->n_bias = n_pad;
    CTTA_TRY(ws.upload(row_map, &pm->bidx));
  }
  if (!bkey.empty()) {
    std::vector<WeightStore::Seg> segs;
    for (int r = 0; r < n_pad; ++r)
      if (row_map[r] >= 0) {
        if (!segs.empty() && segs.back().src_start + segs.back().count == row_map[r] &&
            segs.back().dst_start + segs.back().count == r)
          segs.back().count++;
        else
          segs.push_back({row_map[r], r, 1});
      }
    CTTA_TRY(ws.add_vector(bkey, n_src, n_pad, segs, &P->bias));
  }
  P->n = n_pad; P->k_pad = k_pad;
  return CTTA_OK;
}

// Data-gradient operands.  conv: dX = conv(dY, W rotated 180 deg, in/out channels swapped), so the
// packed rows are the forward INPUT channels and K = (kh, kw, cout); linear: W^T.
// Two ways to fill them: from the fp32 state dict with a pack job (make_conv_dgrad, used where no bf16 forward
// operand exists), or -- cheaper, no fp32 re-read -- as bf16 transposes of the forward operand that the same
// load just packed (make_*_dgrad_from): tap t of the forward pack, a [cout][cin] matrix, becomes the
// [cin][cout] block of tap T-1-t.
static inline ctta_status make_conv_dgrad(WeightStore& ws, const std::string& prefix, int cout, int cin, int kh,
                                          int kw, int pad, ConvLayer* D, int cout_pad = 0) {
  if (cout_pad < cout) cout_pad = cout;   // channel stride of the dY operand
  const int K = kh * kw * cout_pad;
  const int k_pad = round_up(K, 64);
  const int n_pad = round_up(cin, 4);
  std::vector<int32_t> ro(n_pad, -1), co(k_pad, -1);
  for (int ci = 0; ci < cin; ++ci) ro[ci] = ci * kh * kw;
  for (int a = 0; a < kh; ++a)
    for (int b = 0; b < kw; ++b)
      for (int o = 0; o < cout; ++o) co[(a * kw + b) * cout_pad + o] = o * cin * kh * kw + (kh - 1 - a) * kw + (kw - 1 - b);
  CTTA_TRY(ws.add_matrix(prefix + "weight", {cout, cin, kh, kw}, ro, co, nullptr, nullptr, 0, &D->p.w));
  D->p.bias = nullptr; D->p.n = n_pad; D->p.k_pad = k_pad;
  D->cin_pad = cout_pad; D->cout = cin; D->kh = kh; D->kw = kw; D->stride = 1; D->pad = kh - 1 - pad;
  return CTTA_OK;
}
static inline ctta_status make_conv_dgrad_from(WeightStore& ws, const ConvLayer& F, int cout, int cin, ConvLayer* D) {
  CTTA_REQUIRE(cout % 8 == 0 && cin % 8 == 0, "data-gradient pack: channel counts must be multiples of 8");
  const int T = F.kh * F.kw;
  const int k_pad = round_up(T * cout, 64), n_pad = round_up(cin, 4);
  bf16_t* dst = ws.arena.get<bf16_t>((size_t)n_pad * k_pad);   // zero-initialised store: padding stays zero
  if (!dst) { ctta_set_error("weight store exhausted"); return CTTA_ERR_NOMEM; }
  for (int t = 0; t < T; ++t)
    ws.add_transpose(F.p.w + (size_t)t * F.cin_pad, dst + (size_t)(T - 1 - t) * cout, cout, cin, F.p.k_pad, k_pad, cout);
  D->p.w = dst; D->p.bias = nullptr; D->p.n = n_pad; D->p.k_pad = k_pad;
  D->cin_pad = cout; D->cout = cin; D->kh = F.kh; D->kw = F.kw; D->stride = 1; D->pad = F.kh - 1 - F.pad;
  return CTTA_OK;
}
// W^T of a packed linear operand F ([n_rows][k_pad], rows may be a slice of a fused operand)
static inline ctta_status make_linear_dgrad_from(WeightStore& ws, const bf16_t* fw, int n_rows, int k_pad, PackedW* D) {
  const int kd = round_up(n_rows, 64);
  bf16_t* dst = ws.arena.get<bf16_t>((size_t)k_pad * kd);
  if (!dst) { ctta_set_error("weight store exhausted"); return CTTA_ERR_NOMEM; }
  ws.add_transpose(fw, dst, n_rows, k_pad, k_pad, kd, kd);
  D->w = dst; D->bias = nullptr; D->n = k_pad; D->k_pad = kd;
  return CTTA_OK;
}

static inline std::vector<int32_t> identity_map(int n_valid, int n_pad) {
  std::vector<int32_t> m(n_pad, -1);
  for (int i = 0; i < n_valid; ++i) m[i] = i;
  return m;
}
// heads*dh features -> heads*64 padded features
static inline std::vector<int32_t> head_pad_map(int heads, int dh) {
  std::vector<int32_t> m((size_t)heads * 64, -1);
  for (int h = 0; h < heads; ++h)
    for (int d = 0; d < dh; ++d) m[h * 64 + d] = h * dh + d;
  return m;
}

// ---------------------------------------------------------------------------------- run context
struct RunCtx {
  Arena* arena;
  hipStream_t stream;
  bool dry;
  std::vector<Tap>* taps;   // non-null when debug taps are recorded
  float* gn_scratch;        // groupnorm scratch (sized for the largest call)
  size_t gn_scratch_floats;
  // GroupNorm statistics from the producing convolution's epilogue: convolutions / linears whose output may feed a
  // GroupNorm write per-tile partial sums to gn_fpart; the GroupNorm that consumes exactly that tensor next skips its own
  // statistics pass (ctta_groupnorm_from_partials).  gn_ready_* describe the tensor the partials in gn_fpart belong to.
  float* gn_fpart = nullptr;
  size_t gn_fpart_floats = 0;
  int gn_groups = 0;
  const void* gn_ready_x = nullptr;
  int gn_ready_chunks = 0, gn_ready_c = 0;
};

static inline void gn_emit_setup(RunCtx& c, ctta_conv_desc* d, int64_t rows_per_sample, int64_t samples) {
  c.gn_ready_x = nullptr;
  c.gn_ready_chunks = 0;
  if (c.dry || !c.gn_fpart || c.gn_groups <= 0 || d->n % c.gn_groups || rows_per_sample < 64) return;
  (void)samples;
  d->gn_part = c.gn_fpart; d->gn_groups = c.gn_groups; d->gn_hw = (int)rows_per_sample;
  d->gn_part_floats = (int64_t)c.gn_fpart_floats;
}
static inline void gn_emit_done(RunCtx& c, const ctta_conv_desc& d) {
  if (c.dry || !d.gn_part) return;
  const int chunks = ctta_conv_last_gn_chunks();
  if (chunks > 0) { c.gn_ready_x = d.out; c.gn_ready_chunks = chunks; c.gn_ready_c = d.n; }
}


// fused GroupNorm statistics (engine_common.h: gn_emit_setup): partial-sum buffer appended to the GroupNorm scratch
// On by default since round 3 (ctta_set_gn_fuse / CTTA_GN_FUSE=0 turn it off).  Round 2 measured it neutral (the
// statistics then rode in the ROLLED wide-store epilogue, 16.7 us per 256x256 tile); in the straight-line epilogue they
// cost two packed adds and two packed FMAs per 4 outputs (+1..7 us per launch, +4 % on the 128-channel VAE layers) and
// save GroupNorm's statistics pass: conv + GroupNorm 193 -> 177 us (U-Net level 0), 683 -> 632 us (VAE 256 ch), 1005 ->
// 948 us (VAE 128 ch), generation 364 -> 369 clips/s (tools/gn_fuse_ab.py, profiles/README.md).
// Trade-off, stated: the per-tile partial sums make a sample's (mean, rstd) depend -- in the last fp32 bits -- on the
// TILE SHAPE its convolution ran with, which is chosen from the batch size.  A sample's result is still bit-independent
// of its batch mates and of its position at a FIXED batch size (what data-parallel sharding needs); across batch sizes
// it now agrees to bf16 round-off instead of bit for bit (tests/test_engines_gpu.py, both properties; with the fusion
// off the old bit-for-bit property across sizes is asserted too).
static inline bool gn_fuse_enabled() { return ctta_gn_fuse_on(); }

#define RUN(ctx, expr)                 \
  do {                                 \
    if (!(ctx).dry) CTTA_TRY(expr);    \
  } while (0)

#define ALLOC_OR_FAIL(ptr)                                           \
  do {                                                               \
    if (!(ptr)) { ctta_set_error("activation arena exhausted"); return CTTA_ERR_NOMEM; } \
  } while (0)

static inline void desc_init(ctta_conv_desc* d) {
  memset(d, 0, sizeof(*d));
  d->kh = d->kw = 1;
  d->stride_h = d->stride_w = 1;
  d->dil_h = d->dil_w = 1;
  d->alpha = 1.0f;
  d->groups = 1;
}

// 2-D convolution over NHWC bf16 (optionally fused x2 nearest upsample on the input)
static inline ctta_status run_conv2d(RunCtx& c, const ConvLayer& L, const bf16_t* x, int B, int H, int W,
                                     bool upsample, bf16_t* out, const float* rowvec, int rowvec_ld,
                                     const bf16_t* res, int res_ld) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = x; d.c0 = L.cin_pad;
  d.batch = B;
  d.hi = upsample ? 2 * H : H; d.wi = upsample ? 2 * W : W; d.upsample = upsample ? 1 : 0;
  d.kh = L.kh; d.kw = L.kw; d.stride_h = d.stride_w = L.stride; d.pad_h = d.pad_w = L.pad;
  d.ho = (d.hi + 2 * L.pad - L.kh) / L.stride + 1;
  d.wo = (d.wi + 2 * L.pad - L.kw) / L.stride + 1;
  d.w = L.p.w; d.k_pad = L.p.k_pad; d.n = L.p.n; d.bias = L.p.bias;
  d.rowvec = rowvec; d.rowvec_ld = rowvec_ld; d.res = res; d.res_ld = res_ld;
  d.out = out; d.ldc = L.p.n;
  gn_emit_setup(c, &d, (int64_t)d.ho * d.wo, B);
  RUN(c, ctta_conv_gemm(&d, c.stream));
  gn_emit_done(c, d);
  return CTTA_OK;
}

// Linear over rows of a padded bf16 matrix [rows][k_pad] -> [rows][ldc]
static inline ctta_status run_linear(RunCtx& c, const PackedW& P, const bf16_t* x, int x_ld, int64_t rows,
                                     bf16_t* out, int ldc, const bf16_t* res, int res_ld) {
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = x; d.c0 = x_ld;
  d.batch = 1; d.hi = (int)rows; d.wi = 1; d.ho = (int)rows; d.wo = 1;
  d.w = P.w; d.k_pad = P.k_pad; d.n = P.n; d.bias = P.bias;
  d.res = res; d.res_ld = res_ld;
  d.out = out; d.ldc = ldc;
  c.gn_ready_x = nullptr;      // a linear overwrites nothing the partials describe, but ends the producer -> GroupNorm adjacency
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

// V^T = Wv * X^T per batch: Wv packed as the PIXEL operand [hp][k_pad], X [B][tokens][k_pad] as the
// "weight" operand; writes vt [B][hp][vt_ld] (keys contiguous), optional per-row bias.
static inline ctta_status run_vt(RunCtx& c, const PackedW& Wv, const bf16_t* x, int B, int tokens,
                                 int tokens_valid_n, bf16_t* vt, int vt_ld, float* out_f32 = nullptr) {
  (void)out_f32;
  ctta_conv_desc d;
  desc_init(&d);
  d.x0 = Wv.w; d.c0 = Wv.k_pad;
  d.batch = 1; d.hi = Wv.n; d.wi = 1; d.ho = Wv.n; d.wo = 1;
  d.w = x; d.k_pad = Wv.k_pad; d.n = tokens_valid_n;
  d.out = vt; d.ldc = vt_ld;
  d.groups = B; d.x_group_stride = 0; d.w_group_stride = (int64_t)tokens * Wv.k_pad;
  d.out_group_stride = (int64_t)Wv.n * vt_ld;
  RUN(c, ctta_conv_gemm(&d, c.stream));
  return CTTA_OK;
}

struct GNLayer {
  float* gamma = nullptr;
  float* beta = nullptr;
  int c = 0;
  std::string key;   // state-dict prefix ("...norm1.")
};
static inline ctta_status make_gn(WeightStore& ws, const std::string& prefix, int c, GNLayer* g) {
  g->c = c; g->key = prefix;
  CTTA_TRY(ws.add_vector(prefix + "weight", c, &g->gamma));
  CTTA_TRY(ws.add_vector(prefix + "bias", c, &g->beta));
  return CTTA_OK;
}
static inline ctta_status run_gn(RunCtx& c, const GNLayer& g, const bf16_t* x, bf16_t* y, int B, int hw,
                                 int groups, float eps, bool silu, float* stats = nullptr) {
  if (!c.dry && ctta_groupnorm_scratch_floats(B, hw, g.c, groups) > c.gn_scratch_floats) {
    ctta_set_error("groupnorm scratch too small");
    return CTTA_ERR_INVALID;
  }
  if (!c.dry && c.gn_ready_x == (const void*)x && c.gn_ready_chunks > 0 && c.gn_ready_c == g.c && groups == c.gn_groups) {
    // the convolution that produced x already summed it per (sample, row tile, group)
    const int chunks = c.gn_ready_chunks;
    c.gn_ready_x = nullptr;
    c.gn_ready_chunks = 0;
    CTTA_TRY(ctta_groupnorm_from_partials(x, y, B, hw, g.c, groups, g.gamma, g.beta, eps, silu ? 1 : 0, c.gn_fpart, chunks,
                                          c.gn_scratch, stats, c.stream));
    return CTTA_OK;
  }
  RUN(c, ctta_groupnorm_stats_out(x, y, B, hw, g.c, groups, g.gamma, g.beta, eps, silu ? 1 : 0, c.gn_scratch, stats,
                                  c.stream));
  return CTTA_OK;
}

static inline void add_tap(RunCtx& c, const std::string& name, const void* p, int b, int ch, int h, int w,
                           int c_stride, bool f32 = false) {
  if (c.taps && c.dry) c.taps->push_back({name, nullptr, b, ch, h, w, c_stride, f32});
  if (c.taps && !c.dry)
    for (auto& t : *c.taps)
      if (t.name == name) { t.ptr = p; t.b = b; }
}

static inline size_t estimate_store_bytes(const ctta_tensor* w, int n) {
  size_t total = 64 << 20;
  for (int i = 0; i < n; ++i) total += (size_t)tensor_numel(&w[i]) * 2 * 3 + 8192;   // bf16 x padding + maps
  return total;
}
static inline size_t estimate_store_bytes_training(const ctta_tensor* w, int n) {
  size_t total = 64 << 20;   // forward packs + data-gradient packs + fp32 tables
  for (int i = 0; i < n; ++i) total += (size_t)tensor_numel(&w[i]) * 2 * 5 + 16384;
  return total;
}
