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
#include "conv_gemm_kernel.h"

#include <stdlib.h>

CTTA_CONV_VARIANTS_ALL(CTTA_CONV_DECLARE)
CTTA_CONV_VARIANTS_KIND(CTTA_CONV_DECLARE_K)

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
    // the partial sums are requested four slabs at a time and added in slab order (the plain loop waits for every load
    // before it asks for the next: S - 1 serial round trips per lane)
    int s2 = 1;
    for (; s2 + 4 <= S; s2 += 4) {
      float4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const float4*>(src + (size_t)(s2 + u) * slab_stride);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += q[u].x; a.y += q[u].y; a.z += q[u].z; a.w += q[u].w; }
    }
    for (; s2 < S; ++s2) {
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
// The first SK_HDR_WORDS 32-bit words of a workspace are the stream-K header (ConvParams::sk_hdr): zero when the workspace is
// created, kept consistent by the stream-K launches themselves afterwards.  Partial slabs / slots start behind it.
static const size_t kSplitWsBytes = (size_t)192 << 20;
static thread_local float* t_ws = nullptr;
static thread_local size_t t_ws_bytes = 0;
static thread_local int t_ws_hdr = 0;     // the bound workspace's header was zeroed by its owner: stream-K launches may use it
extern "C" void ctta_conv_bind_workspace(void* ws, size_t bytes) { t_ws = (float*)ws; t_ws_bytes = ws ? bytes : 0; t_ws_hdr = 0; }
extern "C" void ctta_conv_bind_workspace_ex(void* ws, size_t bytes, int header_zeroed) {
  t_ws = (float*)ws; t_ws_bytes = ws ? bytes : 0; t_ws_hdr = (ws && header_zeroed) ? 1 : 0;
}
extern "C" void ctta_conv_bound_workspace(void** ws, size_t* bytes) { if (ws) *ws = t_ws; if (bytes) *bytes = t_ws_bytes; }
extern "C" int ctta_conv_bound_workspace_header(void) { return t_ws_hdr; }
extern "C" size_t ctta_conv_workspace_bytes(void) { return kSplitWsBytes; }
extern "C" size_t ctta_conv_workspace_header_bytes(void) { return (size_t)SK_HDR_WORDS * 4; }
#include <mutex>
static float* splitk_workspace(size_t* bytes, bool* hdr_ok = nullptr) {
  if (t_ws) { *bytes = t_ws_bytes; if (hdr_ok) *hdr_ok = t_ws_hdr != 0; return t_ws; }
  static std::mutex mu;
  static float* per_dev[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!per_dev[dev]) {
    if (hipMalloc((void**)&per_dev[dev], kSplitWsBytes) != hipSuccess) per_dev[dev] = nullptr;
    else if (hipMemset(per_dev[dev], 0, (size_t)SK_HDR_WORDS * 4) != hipSuccess) { (void)hipFree(per_dev[dev]); per_dev[dev] = nullptr; }
  }
  *bytes = kSplitWsBytes;
  if (hdr_ok) *hdr_ok = per_dev[dev] != nullptr;
  return per_dev[dev];
}
static int cu_count() { return ctta_cu_count(); }

// ------------------------------------------------------------------------------------------
struct Variant {
  const char* name;
  int bm, bn, bk;
  int wm, wn;
  int mode;
  int stages;
  int kind;     // 0: one tile per workgroup; 2: stream-K (one persistent launch)
  void (*launch)(const ConvParams&, dim3, hipStream_t);
  ctta_status (*prepare)();
};

#define VARIANT(BM, BN, BK, WM, WN, G, S) \
  {#BM "x" #BN "x" #BK "_w" #WM "x" #WN "_m" #G "_s" #S, BM, BN, BK, WM, WN, G, S, 0, \
   launch_variant<BM, BN, BK, WM, WN, G, S, 0>, prepare_variant<BM, BN, BK, WM, WN, G, S, 0>}
#define VARIANT_K(BM, BN, BK, WM, WN, G, S, KIND, TAG) \
  {#BM "x" #BN "x" #BK "_w" #WM "x" #WN "_m" #G "_s" #S TAG, BM, BN, BK, WM, WN, G, S, KIND, \
   launch_variant<BM, BN, BK, WM, WN, G, S, KIND>, prepare_variant<BM, BN, BK, WM, WN, G, S, KIND>}

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
    VARIANT(512, 128, 32, 4, 2, 2, 2),  // 35  N = 128 layers: 8 waves of 128x64 (the big tile's wave shape) over 512 rows
    VARIANT(512, 128, 64, 4, 2, 2, 2),  // 36  ... with BK = 64: the whole 160 KB of LDS
    VARIANT(64, 128, 64, 2, 2, 2, 4),   // 37  deeper rings for thin K-heavy launches (latency-bound: one K tile in flight per
    VARIANT(128, 128, 64, 2, 2, 2, 3),  // 38  workgroup is ~1.1 us per K step whatever the tile)
    VARIANT(128, 64, 64, 2, 2, 2, 3),   // 39
    VARIANT(128, 128, 64, 2, 2, 2, 4),  // 40
    VARIANT_K(256, 256, 64, 2, 4, 2, 2, 2, "_sk"),     // 41  stream-K (one persistent launch, in-launch fold): twins of 29 / 31 / 17
    VARIANT_K(256, 128, 64, 2, 2, 2, 2, 2, "_sk"),     // 42
    VARIANT_K(128, 128, 64, 2, 2, 2, 2, 2, "_sk"),     // 43
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

static bool xcd_default() { return ctta_opt(CTTA_OPT_XCD) != 0; }
// Settled constants (their A/B switches went with round 6; the sweeps and A/Bs that fixed them are in profiles/ and LABNOTES.md)
static constexpr int kSplitkMinNk = 32;       // K tiles from which a launch with few output tiles is split over K
static constexpr int kSplitkTiles = 192;      // ... "few": fewer output tiles than this
static constexpr int kSplitkTarget = 512;     // workgroups a split launch aims for
static constexpr int kSplitkMax = 8;          // most splits
static constexpr int kSplitkMinSteps = 8;     // K tiles every split keeps at least
static constexpr int kBigTileMinK = 512;      // the 256x256x64 tile from this K up (straight-line epilogue, profiles/sweep_r02*.json)
static thread_local unsigned long long* t_stamps = nullptr;
extern "C" void ctta_conv_debug_stamps(void* buf) { t_stamps = (unsigned long long*)buf; }
unsigned long long* ctta_debug_stamps_current() { return t_stamps; }
static thread_local int t_no_splitk = 0;
extern "C" void ctta_conv_suppress_splitk(int on) { t_no_splitk = on ? 1 : 0; }
static bool splitk_default() { return ctta_opt(CTTA_OPT_SPLITK) != 0 && !t_no_splitk; }

// Tile choice from the on-device sweep (tools/sweep_conv.py, profiles/sweep_r01*.json); ids index
// kVariants (1-based).  Returns a register-staged id (1..8); the caller adds +8 / +16 for the
// direct-to-LDS twins.  kBigTile (256x256x64, 8 waves of 128x64) is chosen separately: it halves the
// L1->LDS bytes per FLOP, which is what bounds the 128-wide tiles (64 B/clk/CU vs 512 MFMA-cycles).
static const int kBigTile = 29;
static const int kBigTileSk = 41;     // its stream-K twin
// 0: no; 1: the 256x256x64 tile, one tile per workgroup; 2: the deep case below (stream-K where the workspace allows, else 1)
static int want_big_tile(long long M, int N, long long K, int groups) {
  const long long t256 = ((M + 255) / 256) * ((N + 255) / 256) * groups;
  // one workgroup per CU: 288 tiles (the distillation teacher's batch 18 at level 0) are two rounds of the 256 CUs with the
  // second one 12 % full -- 659-741 TFLOP/s against 828-910 on the thin-grid tile (profiles/sweep_r03.txt, t18 rows)
  // (a ragged last row tile that alone opens a round is cut off into its own small launch: judge the rest)
  long long tq = t256;
  const long long t_cut = (M / 256) * ((N + 255) / 256) * groups;
  if (M % 256 != 0 && groups == 1 && t_cut >= 1 && (t_cut + 255) / 256 < (t256 + 255) / 256) tq = t_cut;
  const long long rounds = (tq + 255) / 256;
  const bool fills = tq >= 1024 || tq * 10 >= rounds * 256 * 7;
  if (N >= 256 && (N % 256 == 0 || N >= 1024) && K >= kBigTileMinK && t256 >= 192 && fills) return 1;
  // Round 6 (profiles/sweep_r06_streamk_v3_coop_fold.txt, weights cold): a DEEP launch (K >= 8192) with 64 .. 191 big tiles is
  // bound by the bytes its workgroups stage per CU-clock, and the small tiles that fill every CU stage the most per FLOP.  On
  // the big tile: as stream-K (one persistent launch, one workgroup per CU, partial tiles folded in the launch) 4608 x 1024 x
  // 9216 822 TFLOP/s, 4096 x 1024 x 9216 845, 4096 x 1024 x 18432 1063, 18432 x 512 x 9216 1014; with the two-pass split-K
  // (7-8 splits) 764 / 727 / 982 / 968; round 5's choices (64x128x64, 3-stage ring, + split-K) 628 / 737 / 757 / 686.
  // Not when the 128x128x64 tile fills its 512 slots evenly (8192 x 1024 x 9216: 1049 vs 1050), which pick_variant tests first;
  // not below 64 tiles (2304 x 1024 x 9216: 548 vs 676 -- the fold traffic does not shrink with M).
  const long long t128 = ((M + 127) / 128) * ((N + 127) / 128) * groups;
  const bool even128 = t128 >= 400 && t128 < 1024 && t128 * 100 >= ((t128 + 511) / 512) * 512 * 85;
  return (N >= 512 && N % 256 == 0 && K >= 8192 && groups == 1 && t256 >= 64 && t256 < 192 && !even128) ? 2 : 0;
}
static int pick_variant(long long M, int N, long long K, int groups) {
  if (N <= 32) {                                           // 256x32; few row tiles (the per-sample cross-attention
    const long long t256 = ((M + 255) / 256) * groups;     // K / V^T projections: 18..180 workgroups walking K = 1024
    return t256 >= 512 ? 4 : 5;                            // one latency-bound tile at a time): 64x64 quadruples them
  }
  if (N <= 64) return K >= 512 ? 8 : 5;                    // 128x64 / 64x64
  const long long t128 = ((M + 127) / 128) * ((N + 127) / 128) * groups;
  if (t128 < 200) return 5;                                // too few 128x128 tiles to fill 256 CUs
  // 128x128x64 when its tiles fill the CUs about evenly (two resident workgroups per CU: 512 slots).  The batch-32 level-2
  // linears (M = 8192, t128 = 512: 736-824 vs 588-682 TFLOP/s on the thin-grid tile) and the Heun teacher's batch 16 at
  // level 1 (t128 = 512: 1024 vs 740, 1109 vs 786, 928 vs 665) take it; batch 18 (t128 = 576 = 2.25 tiles per slot pair)
  // and batch 9 (288) do NOT: 723 vs 866, 665 vs 768 -- profiles/sweep_r03.txt, u32 / t16 / t18 rows.
  if (t128 >= 400 && t128 < 1024 && K >= 512 && t128 * 100 >= ((t128 + 511) / 512) * 512 * 85) return 1;
  if (t128 < 1024 && (K < 4096 || t128 < 400 || N <= 512)) return 6;                    // thin grids (distillation micro-batch): 64x128x64 doubles the workgroups
  // batch 18 at level 0 (M = 73728, N = 256: 1152 tiles = 2.25 rounds of 512 slots): the thin-grid tile beats 128x128x32
  // (828-910 vs 730, round 3) -- and from K = 2048 up 128x128x64 beats both (round 6, profiles/sweep_r06_streamk_v3_coop_fold.txt:
  // K = 4608 942 vs 798 on the 3-stage ring, K = 2304 866 vs 746-779)
  if (t128 < 1536 && K >= 1024 && N <= 512) return K >= 2048 ? 1 : 6;
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
  p.wide_store = (!d->out_f32 && !scalar_store && d->ldc % 4 == 0 && d->n % 4 == 0 &&
                  (p.plain_out || (!geglu && p.obs % 4 == 0 && d->out_offset % 4 == 0 && d->out_limit % 4 == 0)) &&
                  (!d->res || d->res_ld % 4 == 0) && (!d->rowvec || d->rowvec_ld % 4 == 0))
                     ? 1 : 0;
  {
    p.wide_f32 = (d->out_f32 && !scalar_store && d->ldc % 4 == 0 && d->n % 4 == 0 && p.plain_out && !d->res && !d->accumulate &&
                  !d->out2 && d->out_act == 0 && d->alpha == 1.0f && !d->rowvec && !d->bias_m && !geglu && !d->gn_part) ? 1 : 0;
  }
  p.epi_fast_geglu = (geglu && p.wide_store && p.plain_out &&
                      M * (long long)d->ldc * 2 < 0x7FFFFF00LL) ? 1 : 0;
  // (a per-sample strided, shifted, clipped destination -- the ConvTranspose upsamplers -- takes it too since round 6: one
  // descriptor per sample does the clipping; its rows cover the sample: (howo + 1) * ldc elements at most)
  const bool strided_ok = !p.plain_out && !d->res && !d->accumulate && !d->gn_part && !d->rowvec && d->out_limit > 0 &&
                          d->out_limit * 2 < 0x7FFFFF00LL && ((long long)p.howo + 1) * d->ldc * 2 < 0x7FFFFF00LL;
  p.epi_fast = (p.wide_store && (p.plain_out ? M * (long long)d->ldc * 2 < 0x7FFFFF00LL : strided_ok) &&
                (!d->res || M * (long long)d->res_ld * 2 < 0x7FFFFF00LL) && !geglu && !d->bias_m && !(d->gn_part && (d->accumulate || d->out2)) && !(d->accumulate && d->out2) &&
                (d->out_act == 0 || (d->out_act == 3 && d->out_slope >= 0.f && d->out_slope <= 1.f)) &&
                (!d->out2 || (d->out2_slope >= 0.f && d->out2_slope <= 1.f))) ? 1 : 0;
  p.epi_act = (d->alpha != 1.0f || d->out_act == 3) ? 1 : 0;
  if (p.epi_act && d->gn_part) p.epi_fast = 0;   // the statistics instantiations carry no scale / activation
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
  bool thin_ring = false;
  if (vid <= 0 || vid > kNumVariants) {
    const int big = (!d->in_act && !geglu && fast_ok(kVariants[kBigTile - 1].bk)) ? want_big_tile(M, d->n, K, groups) : 0;
    if (big) {
      vid = kBigTile;
      if (big == 2 && splitk_default() && ctta_opt(CTTA_OPT_STREAMK)) {      // stream-K needs a workspace with a live header
        size_t wsb = 0;
        bool hdr_ok = false;
        if (splitk_workspace(&wsb, &hdr_ok) && hdr_ok && !t_stamps) vid = kBigTileSk;
      }
      // (round 1 sent short-K launches with a residual / second output / accumulate to the 256x128x32 tile because the
      // big tile's rolled epilogue could not hide behind another workgroup; with the straight-line epilogue the big tile
      // wins there too: profiles/sweep_r02_epi.json, 670 vs 626 TFLOP/s at K = 768)
    } else {
      vid = pick_variant(M, d->n, K, groups);
      // fused GEGLU: 128x128x32 through the wide-store epilogue (its read-back loop is rolled, so the 16-fragment tile
      // keeps its accumulators in registers); the direct epilogue only exists in the <= 8-fragment tiles (64x128x64)
      // (round 3, after the GELU rewrite: 256x128x32 with 8 waves of 64x64 wins from K = 512 up and on the batch-9 / 16
      // shapes -- 796 vs 727, 919 vs 795, 625 vs 591 TFLOP/s, profiles/sweep_r03.txt; the K = 256 batch-32 launch stays)
      if (geglu) vid = !p.wide_store ? 6 : ((K >= 512 || M < 100000) && fast_ok(32)) ? 28 : 2;
      // deep and narrow (few 128x128 tiles, long K): the 128x128 tile with split-K beats small tiles that only
      // exist to create workgroups (measured: M=1152, N=1024, K=9216 at 176 TFLOP/s on 64x64 tiles)
      const long long t128 = ((M + 127) / 128) * ((d->n + 127) / 128);
      if (splitk_default() && groups == 1 && K >= 4096 && d->n >= 256 && t128 < 192 && !scalar_store && !geglu &&
          d->out_limit == 0 && d->out_offset == 0)
        // M = 1024 / 1152 (teacher batches, level 3): 64x128x64 + split-K 502-523 vs 414-450 TFLOP/s on 128x128x64;
        // M = 576 (batch 9, level 3): 128x64x64 346 vs 295
      {
        // round 5 (profiles/sweep_r05_thin.txt, weights cold): a K step of these launches takes ~1750 clocks whatever the
        // tile (one K tile in flight per workgroup, ~2 workgroups per CU: tools/thin_timeline.py), so the tile that does
        // the most work per step while split-K still fills one round of the CUs wins: M <= 640: 64x128x64 with the 3-stage
        // ring and 7 splits (504 workgroups on 512 slots) 373 vs 304 TFLOP/s on 128x64x64; M <= 1280: 128x128x64 with 7
        // splits 571 vs 420 on 64x128x64
        thin_ring = M <= 640;
        vid = M <= 640 ? 6 : 1;
      }
      if (!d->in_act && vid <= 8) vid += fast_ok(kVariants[vid - 1].bk) ? 16 : 8;
      // 64 < N <= 128 with enough rows: the 256x128x32 tile (8 waves of 64x64) stages 25 % fewer bytes per FLOP than
      // 128x128 / 64x128 and, with the wide-store epilogue, wins from K = 384 up (sweep: +12..22 %)
      const long long t28 = ((M + 255) / 256) * groups;
      if (!geglu && !d->in_act && d->n > 64 && d->n <= 128 && K >= 384 && t28 >= 512 && fast_ok(32) &&
          vid != 1 + 16) {
        vid = 28;
        // ... and from K = 1024 up with >= 2 rounds of 512-row tiles: 512x128x64 (8 waves of 128x64 = the big tile's wave
        // shape, all 160 KB of LDS): 978 vs 937-959 (K = 1152), 1074 vs 961 (K = 2304), 895 vs 806 (k = 11 conv1d) TFLOP/s
        if (K >= 1024 && (M + 511) / 512 * groups >= 512 && fast_ok(64)) vid = 36;
      }
    }
  }
  {   // K-heavy launches on the 64x128x64 tile whose workgroups fill whole rounds at TWO per CU: the 3-stage ring (72 KB of
      // LDS instead of 48: two K tiles in flight while one is consumed -- inside the pipeline the weights of these layers
      // arrive cold from HBM, 1.3 us per K step with one tile in flight).  Teacher loop at batch 16 (M = 4096 x N = 1024:
      // 512 workgroups; M = 16384 x N = 512: 1024): 67.7 -> 70.1 U-Net queries/s.  Batch 9 / 18 (576, 1152 workgroups: 1.1 and
      // 2.25 rounds of 512 slots where the 2-stage tile has 768) lose 2.3 ms of the distillation step with it, so the rule
      // looks at the round fill, like the tile rules above (A/B of round 3).
    // (the same move for the 128x128x64 tile -- 128x128x32 with a 3-stage ring, 48 KB -- measured slower at batch 32 and 16:
    // 25.4 vs 24.8 ms and 14.7 vs 14.5 ms per U-Net forward, tools/r3_probe35.sh)
    if (d->tile <= 0 && vid == 22 && K >= 4096 && !geglu) {
      long long wgs = ((M + 63) / 64) * ((d->n + 127) / 128) * groups;
      if (wgs < kSplitkTiles && groups == 1 && splitk_default()) {      // the split-K factor the launch below will choose
        const long long nk = (K + 63) / 64;
        long long sp = 512 / wgs;
        if (sp > 8) sp = 8;
        if (sp > nk / 8) sp = nk / 8;
        if (sp > 1) wgs *= sp;
      }
      const long long rounds = (wgs + 511) / 512;
      if ((wgs >= 512 && wgs * 100 >= rounds * 512 * 85) || (thin_ring && wgs > 384 && wgs <= 512)) vid = 27;
    }
  }
  CTTA_REQUIRE(!(kVariants[vid - 1].mode != 0 && d->in_act), "conv_gemm: in_act needs a register-staged variant (tile 1..8)");
  if (geglu) {   // direct epilogue: <= 8-fragment tiles (64x64, 64x128, 128x64, 256x32); wide-store: also the 128x128 tiles
    const Variant& gv = kVariants[vid - 1];
    const int frags = (gv.bm / gv.wm / 16) * (gv.bn / gv.wn / 16);     // accumulator fragments per wave
    CTTA_REQUIRE(frags <= 8 || (p.wide_store && frags <= 16),
                 "conv_gemm: the fused GEGLU epilogue needs a tile with <= 8 fragments per wave (or, wide-store, <= 16): got %s",
                 gv.name);
  }
  CTTA_REQUIRE(kVariants[vid - 1].mode != 2 || fast_ok(kVariants[vid - 1].bk),
               "conv_gemm: variant %s needs (c0+c1) %% BK == 0, one source and <= 32 taps", kVariants[vid - 1].name);
  const Variant& v = kVariants[vid - 1];
  if (!conv_wide_f32_ok(v.bm, v.bn, v.bk, v.wm, v.wn, v.mode, v.stages)) p.wide_f32 = 0;
  p.x_bytes = (unsigned)x_bytes; p.w_bytes = (unsigned)w_bytes;
  p.zero = zero_page();
  CTTA_REQUIRE(p.zero, "conv_gemm: could not allocate the zero page");
  p.nk = (int)((K + v.bk - 1) / v.bk);
  CTTA_REQUIRE((long long)p.nk * v.bk <= d->k_pad, "conv_gemm: k_pad too small for BK");
  CTTA_TRY(v.prepare());
  dim3 grid((unsigned)((M + v.bm - 1) / v.bm), (unsigned)((d->n + v.bn - 1) / v.bn), (unsigned)groups);
  p.ksplit = 1; p.nk_split = p.nk;
  if (v.kind >= 2) {
    // Stream-K: one persistent launch, at most one workgroup per CU slot; every workgroup walks an equal share of the (tile, K
    // step) items and the partial tiles are folded inside the launch in K order (ConvParams::sk_hdr, conv_gemm_sk_kernel)
    CTTA_REQUIRE(groups == 1 && !geglu, "conv_gemm: stream-K variant %s takes ungrouped launches without the fused GEGLU", v.name);
    size_t wsb = 0;
    bool hdr_ok = false;
    float* wsp = splitk_workspace(&wsb, &hdr_ok);
    CTTA_REQUIRE(wsp && hdr_ok, "conv_gemm: stream-K needs a workspace whose header was zeroed (ctta_conv_bind_workspace_ex)");
    const long long T = (long long)grid.x * grid.y;
    CTTA_REQUIRE(2 * T <= SK_MAX_GRID, "conv_gemm: stream-K takes at most %d output tiles (got %lld)", SK_MAX_GRID / 2, T);
    const long long items = T * p.nk;
    const int per_cu = (int)((160 * 1024) / ((size_t)v.stages * (v.bm + v.bn) * v.bk * 2));
    long long G = (long long)cu_count() * (per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu);
    if (ctta_opt(CTTA_OPT_STREAMK_GRID) > 0) G = ctta_opt(CTTA_OPT_STREAMK_GRID);
    if (G > items / 4) G = items / 4;          // >= 4 K steps per workgroup
    if (G > SK_MAX_GRID) G = SK_MAX_GRID;
    if (G < 1) G = 1;
    const size_t slot = (size_t)v.bm * v.bn * 4;
    if ((size_t)SK_HDR_WORDS * 4 + (size_t)2 * G * slot > wsb) G = (long long)((wsb - (size_t)SK_HDR_WORDS * 4) / (2 * slot));   // two slots per workgroup
    CTTA_REQUIRE(G >= 1, "conv_gemm: workspace too small for stream-K");
    if (G >= 8) G &= ~7LL;      // whole rounds of the 8 XCDs (the chunk arithmetic of the kernel needs it)
    // XCD chunks (conv_gemm_sk_kernel): the most chunks of whole tiles whose largest is within 6 % of the mean
    int nch = 1;
    for (int c = 8; c > 1; c >>= 1) {
      if (G % 8 != 0 || T < c) continue;
      const long long big = (T + c - 1) / c;
      if (big * c * 100 <= T * 106 && (T / c) * p.nk >= 4 * (G / c)) { nch = c; break; }
    }
    if (xcd_default() == false) nch = 1;
    p.sk_chunks = nch;
    p.m_tiles = (int)grid.x; p.n_tiles = (int)grid.y; p.sk_tiles = (int)T;
    p.sk_m_inner = w_bytes > x_bytes ? 1 : 0;       // weight-dominated: the row tiles of one weight slab run next to each other
    p.sk_hdr = reinterpret_cast<unsigned*>(wsp);
    p.sk_slots = wsp + SK_HDR_WORDS;
    const bool prof_sk = ctta_prof_active();
    if (prof_sk) ctta_prof_begin(0, vid + ((p.epi_fast || p.epi_fast_geglu) ? 0 : 100), M, d->n, K, groups, (hipStream_t)stream);
    // (no GroupNorm statistics from this launch: split tiles leave through the fold, not through the statistics epilogue;
    // ctta_conv_last_gn_chunks() stays 0 and the caller runs its statistics pass)
    v.launch(p, dim3((unsigned)G, 1, 1), (hipStream_t)stream);
    if (prof_sk) ctta_prof_end((hipStream_t)stream);
    CTTA_LAUNCH_CHECK();
    return CTTA_OK;
  }
  // split-K: deep, narrow problems (the 1024-channel levels at small batch: M <= 2304, K = 9216 / 18432) launch
  // too few workgroups to fill 256 CUs; split the K walk over blockIdx.z and reduce in a second pass
  const long long tiles = (long long)grid.x * grid.y;
  int splits = 1;
  float* ws = nullptr;
  size_t ws_bytes = 0;
  if (splitk_default() && groups == 1 && !scalar_store && !geglu && d->out_limit == 0 && d->out_offset == 0 &&
      tiles < kSplitkTiles && p.nk >= kSplitkMinNk && (ws = splitk_workspace(&ws_bytes)) != nullptr &&
      ws_bytes > (size_t)SK_HDR_WORDS * 4) {
    ws += SK_HDR_WORDS; ws_bytes -= (size_t)SK_HDR_WORDS * 4;       // the stream-K header stays untouched
    splits = (int)(kSplitkTarget / tiles);
    if (splits > kSplitkMax) splits = kSplitkMax;
    if (splits > p.nk / kSplitkMinSteps) splits = p.nk / kSplitkMinSteps;
    if (splits < 1) splits = 1;
    const int ld = (d->n + 3) / 4 * 4;
    if ((long long)splits * M * ld * 4 > (long long)ws_bytes) splits = 1;
  }
  const bool prof = ctta_prof_active();
  if (prof) ctta_prof_begin(0, vid + ((p.epi_fast || p.epi_fast_geglu) ? 0 : 100), M, d->n, K, groups, (hipStream_t)stream);   // +100: generic epilogue
  // Ragged last row tile of a 256-row-tile launch: when it alone opens another round of the CUs, it leaves this launch
  // (one row tile fewer) and runs as a second launch with 64x64 tiles behind it (the same rows, the same epilogue; rows
  // are independent, so this is exact).  M = 163872, N = 512: 1282 -> 1280 big tiles = 5 rounds instead of 5 + a round
  // of two half-idle workgroups, plus ~16 small workgroups.
  int tail_rows = 0;
  {
    const long long slots = 256LL * (vid == 28 ? 2 : 1);
    const long long t_all = (long long)grid.x * grid.y, t_cut = (long long)(grid.x - 1) * grid.y;
    if (d->tile <= 0 && v.bm == 256 && v.mode != 0 && splits == 1 && groups == 1 && M % 256 != 0 && grid.x > 1 &&
        !geglu && !d->gn_part && !t_stamps && (t_all + slots - 1) / slots > (t_cut + slots - 1) / slots) {
      tail_rows = (int)(M % 256);
      grid.x -= 1;
    }
  }
  // (launches whose weights outweigh their activations take the weight-slab mapping below instead, whatever their row tiles)
  const bool slab_pref = groups == 1 && grid.y >= 2 && w_bytes > x_bytes && tail_rows == 0;
  if (splits == 1 && groups == 1 && xcd_default() && grid.x >= 64 && !slab_pref) {
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
  // Weight-slab affinity (see ConvParams::slab_total): every split-K launch, and unsplit launches whose few row tiles the
  // M-range mapping above does not take (grid.x < 64) when the weights outweigh the activations and there are slabs enough
  // to give every XCD its own.
  bool slab = false;
  if (groups == 1 && xcd_default() && p.xcd_per == 0) {
    if (splits > 1) slab = true;
    else if (grid.y >= 2 && (long long)grid.x * grid.y >= 16 && w_bytes > x_bytes && tail_rows == 0) slab = true;
  }
  if (slab && splits == 1) {
    p.m_tiles = (int)grid.x; p.n_tiles = (int)grid.y;
    p.slab_total = p.m_tiles * p.n_tiles;
    p.slab_per = (p.slab_total + 7) / 8;
    grid = dim3((unsigned)(8 * p.slab_per), 1, 1);
  }
  if (splits > 1) {
    const int ld = (d->n + 3) / 4 * 4;
    ConvParams q = p;   // first pass: raw partial sums
    q.nk_split = (p.nk + splits - 1) / splits;
    splits = (p.nk + q.nk_split - 1) / q.nk_split;   // every split owns at least one K-tile
    q.ksplit = splits;
    if (slab) {
      q.m_tiles = (int)grid.x; q.n_tiles = (int)grid.y;
      q.slab_total = q.m_tiles * q.n_tiles * splits;
      q.slab_per = (q.slab_total + 7) / 8;
    }
    q.bias = nullptr; q.bias_m = nullptr; q.rowvec = nullptr; q.res = nullptr; q.out_act = 0; q.alpha = 1.0f;
    q.accumulate = 0; q.out2 = nullptr; q.out = ws; q.ldc = ld; q.out_f32 = 1; q.obs = (long long)p.howo * ld;
    q.wide_store = 0;
    q.wide_f32 = conv_wide_f32_ok(v.bm, v.bn, v.bk, v.wm, v.wn, v.mode, v.stages) ? 1 : 0;
    q.ogs = (long long)M * ld;
    grid.z = (unsigned)splits;
    if (slab) grid = dim3((unsigned)(8 * q.slab_per), 1, 1);
    v.launch(q, grid, (hipStream_t)stream);
    const long long total = M * (ld / 4);
    int fb = (int)((total + 255) / 256);
    if (fb > 4096) fb = 4096;
    splitk_finish_kernel<<<dim3(fb), dim3(256), 0, (hipStream_t)stream>>>(p, ws, splits, M * ld, ld);
  } else {
    v.launch(p, grid, (hipStream_t)stream);
    if (tail_rows > 0) {
      const int tid_ = 5 + ((v.mode == 2 && fast_ok(64)) ? 16 : 8);   // 64x64x64, descriptor staging where a K tile never straddles a tap
      const Variant& tv = kVariants[tid_ - 1];
      CTTA_TRY(tv.prepare());
      ConvParams t = p;
      t.m_off = (int)(M - tail_rows);
      t.xcd_per = 0; t.m_tiles = 0; t.n_tiles = 0; t.n_inner = 0; t.slab_total = 0; t.slab_per = 0;
      t.nk = (int)((K + tv.bk - 1) / tv.bk);
      t.nk_split = t.nk;
      tv.launch(t, dim3((unsigned)((tail_rows + tv.bm - 1) / tv.bm), (unsigned)((d->n + tv.bn - 1) / tv.bn), 1), (hipStream_t)stream);
    }
  }
  if (prof) ctta_prof_end((hipStream_t)stream);
  CTTA_LAUNCH_CHECK();
  return CTTA_OK;
}
