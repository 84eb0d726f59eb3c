// Shared device/host helpers for the gfx950 kernels of libctta_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ctta.h"

typedef uint16_t bf16_t;  // raw bfloat16 bit pattern

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction); the shift/add/compare/select sequence it replaces cost ~6 VALU ops per value in every epilogue.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ void unpack8(const uint4& v, float f[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack8(const float f[8]) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// sigmoid / SiLU with the hardware reciprocal (v_rcp_f32, 1 ulp): an IEEE fp32 division costs ten VALU ops
// (div_scale x2, rcp, 4 fma, div_fmas, div_fixup) per element, and the results are rounded to bf16 anyway.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

// Exact-GELU gate g * Phi(g), Phi(g) = 0.5 * (1 + erf(g / sqrt 2)) (F.gelu as attention.py:430-432 / nn.GELU call it) with
// Abramowitz-Stegun 7.1.26 for the error function (|error| <= 1.5e-7: four orders below the bf16 rounding of the result):
// h = 0.5 * erfc(|g| / sqrt 2) = poly(t) * exp(-g^2 / 2), t = 1 / (1 + p |g| / sqrt 2); Phi = h for g < 0 (no
// cancellation in the tail), 1 - h otherwise.  One v_rcp, one v_exp and a dozen FMA-class operations, branch-free.
// ocml's erff is a two-branch polynomial (~35 instructions under exec masks per element), which made the GEGLU epilogue
// of the K = 256 ff1 GEMM as long as its main loop.
__device__ __forceinline__ float gelu_erf_f(float g) {
  const float ax = fabsf(g);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
  float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  poly = fmaf(poly, t, 0.5f * 1.421413741f);
  poly = fmaf(poly, t, 0.5f * -0.284496736f);
  poly = fmaf(poly, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(g * g * (-0.5f * 1.4426950408889634f));
  const float h = poly * t * e;
  return g * (g > 0.f ? 1.0f - h : h);
}

// Phi(g) and phi(g) of the GELU derivative  d/dg [g Phi(g)] = Phi(g) + g phi(g)  from the same rational fit and the same
// exponential as gelu_erf_f (the backward kernels used ocml's erff plus a second exponential per element).
__device__ __forceinline__ void gelu_cdf_pdf(float g, float& cdf, float& pdf) {
  const float ax = fabsf(g);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
  float poly = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  poly = fmaf(poly, t, 0.5f * 1.421413741f);
  poly = fmaf(poly, t, 0.5f * -0.284496736f);
  poly = fmaf(poly, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(g * g * (-0.5f * 1.4426950408889634f));
  const float h = poly * t * e;
  cdf = g > 0.f ? 1.0f - h : h;
  pdf = 0.3989422804014327f * e;
}

// Row sums of the LayerNorm kernels: the total of GL = 32 / 64 consecutive lanes in every one of them, folded on the VALU (DPP
// row rotations inside 16 lanes, v_permlane16/32_swap across them); ONE summation order for every kernel that normalises rows.
__device__ __forceinline__ float dpp_row_sum(float v) {   // total of the 16 lanes of a DPP row, in every lane
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
template <int GL>
__device__ __forceinline__ float ln_group_sum(float v) {
  v = dpp_row_sum(v);
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if constexpr (GL == 64) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host side ---------------------------------------------------------------------------
// Library options (include/ctta.h: ctta_set_option / ctta_get_option).  The ONLY switches of the library: nothing in it reads
// the environment.  Values are plain ints read at the call that uses them.
enum CttaOption {
  CTTA_OPT_XCD = 0,            // XCD-aware tile order of the conv_gemm launches (default 1)
  CTTA_OPT_SPLITK,             // K decomposition of deep thin launches: two-pass split-K / stream-K (default 1; 0: one tile per workgroup, whole K)
  CTTA_OPT_STREAMK,            // stream-K (one persistent launch, in-launch fold) where the rules choose it (default 1); 0: two-pass split-K only
  CTTA_OPT_STREAMK_GRID,       // tuning: workgroups of a stream-K launch (0 = one per CU slot)
  CTTA_OPT_WGRAD_STREAM,       // weight-gradient launches on the handle's side stream (default 1); read per backward call
  CTTA_OPT_GN_FUSE,            // GroupNorm statistics from the producing convolution's epilogue (default 1)
  CTTA_OPT_FUSED_RES,          // HiFi-GAN ResBlock units as fused pair kernels (default 1)
  CTTA_OPT_FFN_FUSE,           // transformer feed-forward of the 256-wide level as one row-tile kernel (ffn_fused.hip) in the inference forward
  CTTA_OPT_COUNT
};
extern int g_ctta_opt[CTTA_OPT_COUNT];
int ctta_cu_count();   // api.hip
static inline int ctta_opt(CttaOption o) { return g_ctta_opt[o]; }
void ctta_set_error(const char* fmt, ...);
bool ctta_prof_active();
bool ctta_gn_fuse_on();
unsigned long long* ctta_debug_stamps_current();   // ctta_conv_debug_stamps' buffer of this host thread (or null)
void ctta_prof_begin(int kind, int variant, long long m, long long n, long long k, long long groups, hipStream_t s);
void ctta_prof_end(hipStream_t s);
hipError_t ctta_zero_async(void* ptr, size_t bytes, hipStream_t s);   // zero-fill kernel (no memset nodes in hipGraphs)

#define CTTA_CHECK_HIP(expr)                                                         \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ctta_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return CTTA_ERR_HIP;                                                           \
    }                                                                                \
  } while (0)

#define CTTA_REQUIRE(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      ctta_set_error(__VA_ARGS__);             \
      return CTTA_ERR_INVALID;                 \
    }                                          \
  } while (0)

#define CTTA_TRY(expr)                    \
  do {                                    \
    ctta_status _s = (expr);              \
    if (_s != CTTA_OK) return _s;         \
  } while (0)

#define CTTA_LAUNCH_CHECK()                                                          \
  do {                                                                               \
    hipError_t _e = hipGetLastError();                                               \
    if (_e != hipSuccess) {                                                          \
      ctta_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return CTTA_ERR_HIP;                                                           \
    }                                                                                \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
