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

static inline bool gode_strides_are_channels_last(const int64_t* s) { return s[0] == 0 && s[1] == 0 && s[2] == 0 && s[3] == 0 && s[4] == 0; }
