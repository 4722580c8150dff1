// ode_common.h -- device helpers shared by ode.hip and odernn.hip: the transposed 16x16 mat-vec batch on
// v_mfma_f32_16x16x4_f32 (see the header comment of ode.hip for the lane mapping) and the activations.
#pragma once
#include "common.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 matvec(const f32x4 w, const f32x4 x, f32x4 c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) c = MFMA16(w[r], x[r], c);
  return c;
}
// tanh(x) = 1 - 2/(exp(2x)+1) on the hardware exp/rcp units (v_exp_f32, v_rcp_f32): absolute error ~1e-7, saturates
// correctly at +-inf; libm's tanhf costs ~10x more instructions and dominated the solve (4 tanh per lane per RHS).
__device__ __forceinline__ float fast_tanh(float x) {
  const float t = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}
__device__ __forceinline__ f32x4 tanh4(const f32x4 u) {
  return f32x4{fast_tanh(u[0]), fast_tanh(u[1]), fast_tanh(u[2]), fast_tanh(u[3])};
}
__device__ __forceinline__ f32x4 lrelu4(const f32x4 u) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = u[r] > 0.f ? u[r] : 0.2f * u[r];
  return o;
}
__device__ __forceinline__ f32x4 lrelu_grad4(const f32x4 pre, const f32x4 gr) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = pre[r] > 0.f ? gr[r] : 0.2f * gr[r];
  return o;
}


#define LDT 20  // LDS row stride (floats) of a 16x16 transpose tile: conflict-free b128 writes / b32 reads

// sum over the 16 trajectories of a wave (lanes sharing g)
__device__ __forceinline__ f32x4 sum_over_samples(f32x4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] += __shfl_xor(v[r], 1);
    v[r] += __shfl_xor(v[r], 2);
    v[r] += __shfl_xor(v[r], 4);
    v[r] += __shfl_xor(v[r], 8);
  }
  return v;
}


__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ f32x4 sigmoid4(const f32x4 u) {
  return f32x4{fast_sigmoid(u[0]), fast_sigmoid(u[1]), fast_sigmoid(u[2]), fast_sigmoid(u[3])};
}
