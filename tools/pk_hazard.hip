// Synthetic victim for the packed-fp32 finding (LABNOTES.md 5): every lane iterates acc = fma(acc, a, b) on a pair of
// values, once with the packed instruction (v_pk_fma_f32) and once with two scalar v_fma_f32, from identical inputs.
// variant 1 / 2 take the multiplier through the op_sel_hi / op_sel source swizzles the SLP vectoriser emits.
// Without interference the packed and the scalar results agree bit for bit in every lane.
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void pk_victim(const float* __restrict__ in, float* __restrict__ out_pk, float* __restrict__ out_sc, int iters, int variant) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float a = in[0], b = in[1];
  f32x2 p = {in[2] + i * 1e-6f, in[3] - i * 1e-6f};
  float s0 = p.x, s1 = p.y;
  const f32x2 av = {a, a}, bv = {b, b};
  const f32x2 a_lo = {a, 999.0f}, a_hi = {999.0f, a};      // operand pairs whose OTHER half must never be read
  for (int k = 0; k < iters; ++k) {
    if (variant == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(av), "v"(bv));
    else if (variant == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(p) : "v"(p), "v"(a_lo), "v"(bv));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p) : "v"(p), "v"(a_hi), "v"(bv));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(s0), "v"(a), "v"(b));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(s1), "v"(a), "v"(b));
  }
  out_pk[2 * i] = p.x; out_pk[2 * i + 1] = p.y;
  out_sc[2 * i] = s0; out_sc[2 * i + 1] = s1;
}
extern "C" int pk_launch(const float* in, float* out_pk, float* out_sc, int blocks, int iters, int variant, void* stream) {
  hipLaunchKernelGGL(pk_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out_pk, out_sc, iters, variant);
  return (int)hipGetLastError();
}

// ---- synthetic co-runners: which resource of the neighbour matters?
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern "C" __global__ void co_mfma(float* out, int iters) {          // matrix pipe only: no memory traffic, no LDS
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * i); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int k = 0; k < iters; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
extern "C" __global__ void co_stream(const float4* __restrict__ in, float4* __restrict__ out, long long n) {   // HBM only
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = in[i];
}
extern "C" __global__ void co_valu(float* out, int iters) {          // plain fp32 VALU only
  float x = threadIdx.x * 1e-3f, y = 0.5f;
  for (int k = 0; k < iters; ++k) { x = __builtin_fmaf(x, 0.999f, y); y = __builtin_fmaf(y, 0.998f, x); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}
extern "C" int co_launch(int kind, void* a, void* b, long long n, int iters, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(co_mfma, dim3(2048), dim3(256), 0, s, (float*)a, iters);
  else if (kind == 1) hipLaunchKernelGGL(co_stream, dim3(4096), dim3(256), 0, s, (const float4*)a, (float4*)b, n);
  else hipLaunchKernelGGL(co_valu, dim3(2048), dim3(256), 0, s, (float*)a, iters);
  return (int)hipGetLastError();
}
