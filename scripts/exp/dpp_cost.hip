// Micro-benchmark: cycles per v_fmac_f32_dpp (row_ror) vs plain v_fmac_f32, one wave per SIMD and 4 waves per SIMD,
// 4 and 16 independent accumulators.   hipcc --offload-arch=gfx950 -O3 scripts/exp/dpp_cost.hip -o /tmp/dpp_cost && /tmp/dpp_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
template <int MODE>
__global__ void k(float* out, long long* cyc, float x) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = x + i + threadIdx.x;
  float y = x * threadIdx.x, w = 1.0001f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < REP; ++it) {
    if (MODE == 0) {        // 16 dpp fmac, 4 accumulators (mv16 pattern)
      asm volatile(
        "v_fmac_f32_dpp %0, %4, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %4, %5 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %5 row_ror:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %4, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %5 row_ror:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %4, %5 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %5 row_ror:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %5 row_ror:9 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %4, %5 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %5 row_ror:11 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %4, %5 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %5 row_ror:13 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %4, %5 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %4, %5 row_ror:15 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %4, %5 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(y), "v"(w));
    } else if (MODE == 1) { // 16 plain fmac, 4 accumulators
      asm volatile(
        "v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5\n\t"
        "v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5\n\t"
        "v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5\n\t"
        "v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(y), "v"(w));
    } else if (MODE == 2) { // 15 mov_dpp + 16 plain fmac (what hipcc emits)
      float t[15];
      asm volatile(
        "v_mov_b32_dpp %0, %15 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %15 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %15 row_ror:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %15 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %4, %15 row_ror:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %15 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %6, %15 row_ror:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %15 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %8, %15 row_ror:9 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %9, %15 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %10, %15 row_ror:11 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %15 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %12, %15 row_ror:13 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %13, %15 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %14, %15 row_ror:15 row_mask:0xf bank_mask:0xf"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]),
          "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]) : "v"(y));
      for (int j = 0; j < 15; ++j) a[j & 3] = fmaf(t[j], w, a[j & 3]);
      a[0] = fmaf(y, w, a[0]);
    } else if (MODE == 3) { // 16 dpp fmac, 16 accumulators (outer16 pattern)
      asm volatile(
        "v_fmac_f32_dpp %0, %16, %17 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %16, %17 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %16, %17 row_ror:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %16, %17 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %16, %17 row_ror:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %5, %16, %17 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %16, %17 row_ror:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %7, %16, %17 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %8, %16, %17 row_ror:9 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %9, %16, %17 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %10, %16, %17 row_ror:11 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %11, %16, %17 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %12, %16, %17 row_ror:13 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %13, %16, %17 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %14, %16, %17 row_ror:15 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %15, %16, %17 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]),
          "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(y), "v"(w));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  const char* names[4] = {"16 v_fmac_dpp / 4 acc", "16 v_fmac / 4 acc", "15 v_mov_dpp + 16 v_fma / 4 acc", "16 v_fmac_dpp / 16 acc"};
  for (int threads : {64, 256, 1024}) {
    for (int m = 0; m < 4; ++m) {
      long long h = 0;
      for (int rep = 0; rep < 3; ++rep) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.f);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      }
      printf("threads/block %4d  %-32s %8.1f cycles per group of 16 (s_memtime ticks)\n", threads, names[m], (double)h / REP);
    }
  }
  return 0;
}
