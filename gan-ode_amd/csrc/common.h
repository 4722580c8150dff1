// common.h -- launch helpers shared by the HIP translation units of libgode.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gode.h"

#define GODE_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gode_act(float v, int act) {
  if (act == GODE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == GODE_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  return v;
}
// derivative of act evaluated at pre-activation z
__device__ __forceinline__ float gode_act_grad(float z, int act) {
  if (act == GODE_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == GODE_ACT_LRELU) return z > 0.f ? 1.f : 0.2f;
  return 1.f;
}

// Raw-buffer view of an operand for the LDS-DMA loads (the resource type and its builtins exist in the device pass only; the
// host pass, which only needs the kernel's launch stub, sees empty shells).
struct BufRsrc {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t r;
#endif
};
__device__ __forceinline__ BufRsrc make_buf_rsrc(const float* p, uint32_t bytes) {
  BufRsrc b;
#if defined(__HIP_DEVICE_COMPILE__)
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);   // raw buffer, 32-bit words
#else
  (void)p; (void)bytes;
#endif
  return b;
}
// 16 bytes per lane, memory[voff + soff] -> LDS[lds + 16 * lane]: voff per lane (VGPR), soff wave-uniform (SGPR); a voff beyond
// the buffer's extent delivers zeros
__device__ __forceinline__ void buf_dma16(const BufRsrc& b, float* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#else
  (void)b; (void)lds; (void)voff; (void)soff;
#endif
}

static inline bool gode_strides_are_channels_last(const int64_t* s) { return s[0] == 0 && s[1] == 0 && s[2] == 0 && s[3] == 0 && s[4] == 0; }

// Division by a run-time invariant (Granlund-Montgomery, branch-free, exact for all 32-bit n): the host builds the
// magic once per launch, the kernels pay a v_mul_hi + 3 ALU ops instead of a ~30-instruction software divide.
struct FastDiv { uint32_t mul, sh1, sh2, d; };
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f; f.d = d;
  uint32_t l = 0; while ((1ull << l) < d) ++l;
  f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.sh1 = l < 1 ? l : 1; f.sh2 = l > 0 ? l - 1 : 0;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  const uint32_t t = __umulhi(f.mul, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
