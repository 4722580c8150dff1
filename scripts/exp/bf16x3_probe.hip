// bf16x3_probe.hip -- feasibility probe for a 3-way split-bf16 GEMM on gfx950 ("what 9.5 ms would take").
//
// Question: if every fp32 operand is stored pre-split as three bf16 planes (a = a0 + a1 + a2) and the products
// a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 run on v_mfma_f32_32x32x16_bf16 (6 MFMAs of 32 cycles per K = 16 instead of
// 8 fp32 MFMAs of 64 cycles), how fast does a 128x128-tile LDS-DMA kernel of the igemm_fast shape actually go?  The
// arithmetic ceiling is 2.67x the fp32 MFMA's; the operand stream is 1.5x the bytes in 1/2.67 of the time.
//
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/probe scripts/exp/bf16x3_probe.hip && /tmp/probe
// Prints the launch time and the fp32-equivalent TFLOP/s (2*M*N*K / time) for K slabs of 32 and 16, plus the max error
// of one output tile against a double-precision product of the ORIGINAL fp32 operands (i.e. the split's accuracy).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// C[M][N] = A[M][K] * B[N][K]^T, operands as three bf16 planes each ([3][rows][K]).  128x128 tile, 4 waves (2x2), each
// wave 64x64 = 2x2 MFMA blocks.  LDS image per plane and operand: [128 rows][SK k] bf16, 16-byte chunks XOR-swizzled
// through the DMA's source address (the destination is lane-linear), one barrier per slab, two buffers.
template <int SK>
__global__ void __launch_bounds__(256) probe_kernel(const __bf16* A, const __bf16* B, float* C, int M, int N, int K) {
  constexpr int ROWB = SK * 2;                 // bytes per row and plane
  constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (4 at SK = 32, 2 at SK = 16)
  constexpr int PLANE = 128 * ROWB;            // bytes per plane image
  constexpr int BUF = 6 * PLANE;               // A0 A1 A2 B0 B1 B2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mblk = blockIdx.x, nblk = blockIdx.y;
  const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;

  // DMA: one wave instruction = 1 KiB = 64 / CPR rows of one plane.  Lane l: row l / CPR, LDS chunk l % CPR <- K chunk (l % CPR) ^ key(row)
  constexpr int RPI = 64 / CPR;                // rows per instruction
  constexpr int IPP = 128 / RPI;               // instructions per plane
  const int drow = lane / CPR, dch = lane % CPR;
  auto key = [](int row) { return CPR == 4 ? (row >> 2) & 3 : (row >> 3) & 1; };
  auto dma = [&](int slab, int buf) {
    // 6 planes x IPP instructions, dealt over the 4 waves
    for (int q = wave; q < 6 * IPP; q += 4) {
      const int plane = q / IPP, ins = q - plane * IPP;
      const int row = ins * RPI + drow;
      const bool isB = plane >= 3;
      const __bf16* src = isB ? B + (int64_t)(plane - 3) * planeB + (int64_t)(nblk * 128 + row) * K
                              : A + (int64_t)plane * planeA + (int64_t)(mblk * 128 + row) * K;
      src += slab * SK + ((dch ^ key(row)) * 8);
      char* dst = smem + buf * BUF + plane * PLANE + ins * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fr = lane & 31, fk = lane >> 5;    // fragment: row fr, k group fk (8 consecutive k)
  auto frag = [&](int buf, int plane, int row, int kstep) -> bf8 {
    const int chunk = kstep * 2 + fk;          // 16-byte chunk index inside the row (k = 8 * chunk)
    const char* p = smem + buf * BUF + plane * PLANE + row * ROWB + ((chunk ^ key(row)) * 16);
    return *reinterpret_cast<const bf8*>(p);
  };
  const int nslab = K / SK;
  dma(0, 0);
  for (int s = 0; s < nslab; ++s) {
    __syncthreads();
#ifndef PROBE_NO_DMA      // ablation: -DPROBE_NO_DMA = fragment reads + MFMA only (results wrong)
    if (s + 1 < nslab) dma(s + 1, (s + 1) & 1);
#endif
    const int buf = s & 1;
#pragma unroll
    for (int ks = 0; ks < SK / 16; ++ks) {
      bf8 a[2][3], b[2][3];
#ifdef PROBE_NO_READ     // ablation: -DPROBE_NO_READ = DMA + MFMA only, fragments read once (results wrong)
      if (s == 0)
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = frag(buf, p, wm * 64 + i * 32 + fr, ks);
          b[i][p] = frag(buf, 3 + p, wn * 64 + i * 32 + fr, ks);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);     // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) {
        const int row = mblk * 128 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = nblk * 128 + wn * 64 + j * 32 + (lane & 31);
        C[(int64_t)row * N + col] = acc[i][j][r];
      }
}

static uint16_t f2bf(float f) {                 // round to nearest even
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int SK>
static void run(const char* name, const __bf16* dA, const __bf16* dB, float* dC, int M, int N, int K) {
  const size_t lds = 2 * 6 * 128 * SK * 2;
  CHECK(hipFuncSetAttribute((const void*)probe_kernel<SK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(M / 128, N / 128);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe_kernel<SK>, grid, dim3(256), lds, 0, dA, dB, dC, M, N, K);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_kernel<SK>, grid, dim3(256), lds, 0, dA, dB, dC, M, N, K);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-28s M=%d N=%d K=%d  %8.1f us  %7.1f fp32-equivalent TFLOP/s  (LDS %zu KB per workgroup)\n", name, M, N, K, ms * 1e3,
         2.0 * M * N * K / ms / 1e9, lds / 1024);
}

int main() {
  const int M = 32768, N = 256, K = 2048;      // the ConvT 512 -> 256 layer's per-phase GEMM, 4 phases worth of rows
  std::vector<float> A((size_t)M * K), B((size_t)N * K);
  srand(1);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  std::vector<uint16_t> Ap(3 * A.size()), Bp(3 * B.size());
  auto split = [](const std::vector<float>& X, std::vector<uint16_t>& P) {
    const size_t n = X.size();
    for (size_t i = 0; i < n; ++i) {
      const uint16_t h0 = f2bf(X[i]); const float r1 = X[i] - bf2f(h0);
      const uint16_t h1 = f2bf(r1); const float r2 = r1 - bf2f(h1);
      P[i] = h0; P[n + i] = h1; P[2 * n + i] = f2bf(r2);
    }
  };
  split(A, Ap); split(B, Bp);
  __bf16 *dA, *dB; float* dC;
  CHECK(hipMalloc(&dA, Ap.size() * 2)); CHECK(hipMalloc(&dB, Bp.size() * 2)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMemcpy(dA, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
  run<32>("bf16x3, K slabs of 32", dA, dB, dC, M, N, K);
  run<16>("bf16x3, K slabs of 16", dA, dB, dC, M, N, K);
  // accuracy of the split against fp64 on the first 128 x 128 tile (and of plain fp32 accumulation for comparison)
  std::vector<float> Ct((size_t)128 * N);
  CHECK(hipMemcpy(Ct.data(), dC, Ct.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst32 = 0, scale = 0;
  for (int r = 0; r < 128; ++r)
    for (int c = 0; c < 128; ++c) {
      double ref = 0; float f32 = 0.f;
      for (int k = 0; k < K; ++k) { ref += (double)A[(size_t)r * K + k] * B[(size_t)c * K + k]; f32 += A[(size_t)r * K + k] * B[(size_t)c * K + k]; }
      worst = fmax(worst, fabs(Ct[(size_t)r * N + c] - ref)); worst32 = fmax(worst32, fabs((double)f32 - ref)); scale = fmax(scale, fabs(ref));
    }
  printf("max |bf16x3 - fp64| / max|ref| = %.3g   (sequential fp32 accumulation: %.3g)\n", worst / scale, worst32 / scale);
  return 0;
}
