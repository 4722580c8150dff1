// valu_common.h -- the VECTOR-ALU / DPP mapping of the 16-wide mat-vecs shared by ode_valu.hip and odernn_valu.hip:
// one lane = one (trajectory, feature) pair, the 16 features of a trajectory in one DPP row (16 lanes), 4 trajectories
// per wave.  See the header comment of ode_valu.hip.
#pragma once
#include "common.h"

template <int R> __device__ __forceinline__ float row_ror(float v) {
  if (R == 0) return v;
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (R & 15), 0xf, 0xf, false));
}

struct W16 { float w[16]; };

// out_i = bias_i + sum_r W.w[r] * y[src(i, r)].  hipcc does not fold a DPP move into v_fmac (it emits v_mov_b32_dpp +
// v_fmac, or SLP-packs the FMAs into v_pk_fma_f32, which has no DPP form): the 16 instructions are written out.  The
// leading s_nop 1 covers the 2 wait states a DPP read needs after a VALU write of the same VGPR (the rotated operand
// is produced by compiler-scheduled code just before the block); accumulators are plain operands.
__device__ __forceinline__ float mv16(const W16& W, float y, float bias) {
  float a0, a1, a2, a3;
  asm("s_nop 1\n\t"
      "v_fma_f32 %0, %5, %4, %21\n\t"
      "v_mul_f32_dpp %1, %4, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %2, %4, %7 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %3, %4, %8 row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %4, %9 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %4, %10 row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %4, %11 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %3, %4, %12 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %4, %13 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %4, %14 row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %4, %15 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %3, %4, %16 row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %4, %17 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %4, %18 row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %4, %19 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %3, %4, %20 row_ror:15 row_mask:0xf bank_mask:0xf"
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
      : "v"(y), "v"(W.w[0]), "v"(W.w[1]), "v"(W.w[2]), "v"(W.w[3]), "v"(W.w[4]), "v"(W.w[5]), "v"(W.w[6]), "v"(W.w[7]), "v"(W.w[8]), "v"(W.w[9]), "v"(W.w[10]), "v"(W.w[11]), "v"(W.w[12]), "v"(W.w[13]), "v"(W.w[14]), "v"(W.w[15]), "v"(bias));
  return (a0 + a1) + (a2 + a3);
}
// The same product as ONE dependent chain (16 instructions instead of 19, no adds at the end): for kernels that run two
// waves per SIMD and are bound by vector-ALU issue, not by the length of the dependent chain (odernn_valu.hip).
__device__ __forceinline__ float mv16c(const W16& W, float y, float bias) {
  float a0;
  asm("s_nop 1\n\t"
      "v_fma_f32 %0, %2, %1, %18\n\t"
      "v_fmac_f32_dpp %0, %1, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %4 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %5 row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %7 row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %8 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %9 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %11 row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %12 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %13 row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %14 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %15 row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %16 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %1, %17 row_ror:15 row_mask:0xf bank_mask:0xf"
      : "=&v"(a0)
      : "v"(y), "v"(W.w[0]), "v"(W.w[1]), "v"(W.w[2]), "v"(W.w[3]), "v"(W.w[4]), "v"(W.w[5]), "v"(W.w[6]), "v"(W.w[7]), "v"(W.w[8]), "v"(W.w[9]), "v"(W.w[10]), "v"(W.w[11]), "v"(W.w[12]), "v"(W.w[13]), "v"(W.w[14]), "v"(W.w[15]), "v"(bias));
  return a0;
}
// G.w[r] += p * q[src(i, r)]   (per-lane slice of the outer product p q^T)
__device__ __forceinline__ void outer16(W16& G, float p, float q) {
  asm("s_nop 1\n\t"
      "v_fmac_f32 %0, %17, %16\n\t"
      "v_fmac_f32_dpp %1, %17, %16 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %17, %16 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %3, %17, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %4, %17, %16 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %5, %17, %16 row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %6, %17, %16 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %7, %17, %16 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %8, %17, %16 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %9, %17, %16 row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %10, %17, %16 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %11, %17, %16 row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %12, %17, %16 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %13, %17, %16 row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %14, %17, %16 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %15, %17, %16 row_ror:15 row_mask:0xf bank_mask:0xf"
      : "+v"(G.w[0]), "+v"(G.w[1]), "+v"(G.w[2]), "+v"(G.w[3]), "+v"(G.w[4]), "+v"(G.w[5]), "+v"(G.w[6]), "+v"(G.w[7]), "+v"(G.w[8]), "+v"(G.w[9]), "+v"(G.w[10]), "+v"(G.w[11]), "+v"(G.w[12]), "+v"(G.w[13]), "+v"(G.w[14]), "+v"(G.w[15])
      : "v"(p), "v"(q));
}
// src[r] = the row-local lane whose value rotation r delivers to this lane
__device__ __forceinline__ void probe_sources(int i, int* src) {
  const float fi = (float)i;
  src[0] = i;
  src[1] = (int)row_ror<1>(fi);   src[2] = (int)row_ror<2>(fi);   src[3] = (int)row_ror<3>(fi);
  src[4] = (int)row_ror<4>(fi);   src[5] = (int)row_ror<5>(fi);   src[6] = (int)row_ror<6>(fi);
  src[7] = (int)row_ror<7>(fi);   src[8] = (int)row_ror<8>(fi);   src[9] = (int)row_ror<9>(fi);
  src[10] = (int)row_ror<10>(fi); src[11] = (int)row_ror<11>(fi); src[12] = (int)row_ror<12>(fi);
  src[13] = (int)row_ror<13>(fi); src[14] = (int)row_ror<14>(fi); src[15] = (int)row_ror<15>(fi);
}
// rows of a row-major [rows][ld] matrix: w[r] = M[(row0 + i) * ld + col0 + src[r]]
__device__ __forceinline__ void load_rows(W16& W, const float* M, int ld, int row0, int col0, int i, const int* src) {
#pragma unroll
  for (int r = 0; r < 16; ++r) W.w[r] = M[(row0 + i) * ld + col0 + src[r]];
}
// transposed: w[r] = M[(row0 + src[r]) * ld + col0 + i]  (so that mv16 computes M^T v)
__device__ __forceinline__ void load_cols(W16& W, const float* M, int ld, int row0, int col0, int i, const int* src) {
#pragma unroll
  for (int r = 0; r < 16; ++r) W.w[r] = M[(row0 + src[r]) * ld + col0 + i];
}
__device__ __forceinline__ float lrelu1(float u) { return u > 0.f ? u : 0.2f * u; }


// ---- reductions ---------------------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave, the same value in every lane, fixed order: four row rotations make every lane hold
// its row's sum, the four row sums are read out with v_readlane and added as (r0 + r1) + (r2 + r3).  ~10 VALU issue slots
// against six dependent ds_bpermute round trips (~100+ cycles each) for the __shfl_xor butterfly.
__device__ __forceinline__ float wave_sum(float v) {
  v += row_ror<8>(v);
  v += row_ror<4>(v);
  v += row_ror<2>(v);
  v += row_ror<1>(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
// sum over the 4 trajectories of a wave (lanes i, i+16, i+32, i+48), the result in every lane, fixed order
__device__ __forceinline__ float traj_sum(float v) {
  const int b = __builtin_bit_cast(int, v);
  const int i = threadIdx.x & 15;
  // four readlanes cannot take a per-lane index; the row partner values come through LDS-less cross-lane reads
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((i) << 2, b));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((i + 16) << 2, b));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((i + 32) << 2, b));
  const float e = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((i + 48) << 2, b));
  return (a + c) + (d + e);
}
