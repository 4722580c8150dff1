// conv_patch.hip -- forward of the discriminators' FIRST layers (1-4 input channels, 8-64 taps: Conv3d(3, 64, 4) of the
// UCF video discriminator, Conv2d(3, 64, 4) of its image discriminator, models/mocogan.py:72,138) without an im2col.
//
// With 3 input channels the generic igemm kernel gathers its A operand one scalar at a time (K = 192 is 64 taps x 3
// channels, nothing is 16-byte contiguous): 35 TFLOP/s.  Here -- the forward twin of wgrad_patch_kernel -- a slab is
// (up to) 32 consecutive positions of ONE output row; the kd*kh input rows it touches are staged in LDS once
// ([row][w][ci], any global strides, zero fill), and the MFMA's A operand (rows = positions, k = (tap, channel)) is read
// straight out of that patch: position q, column j = (row r, kw, ci) sits at r*PL + (q*sw + kw)*Ci + ci, i.e. a per-lane
// base q*sw*Ci, a per-row scalar and an immediate.  The weights of a wave's 32 output channels live in registers for
// the whole launch (kd*kh*kw*Ci / 2 VGPRs: one value per MFMA k step).  A workgroup works on 4 / (Co / 32) slabs at a
// time, one 32x32 output block per wave; patches are double-buffered.  Output: raw [positions][Co], as gode_igemm.
#include <stdint.h>
#include "common.h"
#include "conv_geom.h"

#define CP_TAB 128          // slabs whose corners are decoded at a time (LDS table)
#define CP_MAXCOL 2         // patch columns staged per thread and row

struct CpArgs {
  gode_conv_geom g;
  const float* x; const float* w; float* out;
  int32_t xsN, xsD, xsH, xsW, xsC;
  int32_t Kp;                // row length of the packed FPROP panel Bp[co][Kp], k = tap * Ci + ci
  int32_t CB, SPI;           // 32-column blocks of Co; slabs per workgroup iteration (= 4 / CB)
  int32_t PL, LWC;           // padded patch row length (floats); staged columns per row ((31*sw + kw) * Ci)
  int32_t nseg, segs, iters_per_wg;
  FastDiv dseg, dHo, dDo, dCi;
};

template <int KD, int KH, int KC2>       // KC2 = kw * Ci / 2: MFMA k steps per patch row
__global__ void __launch_bounds__(256) conv_patch_fprop_kernel(const CpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float cp_smem[];
  constexpr int PR = KD * KH;
  __shared__ int tab[CP_TAB][6];         // per slab: x corner offset, bw, masks (depth | row << 8), valid positions, first out row (lo, hi)
  const gode_conv_geom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int Ci = g.Ci, Co = g.Co, sw = g.sw;
  const int patch = PR * a.PL;
  const int my_slab = wave / a.CB, cb = wave - my_slab * a.CB;     // this wave's slab of the iteration and column block

  // weights of this wave's 32 output channels: value k step (r, ks) = W[cb*32 + fr][(r * kw * Ci) + 2 ks + fh]
  float wreg[PR * KC2];
#pragma unroll
  for (int r = 0; r < PR; ++r)
#pragma unroll
    for (int ks = 0; ks < KC2; ++ks) wreg[r * KC2 + ks] = a.w[(int64_t)(cb * 32 + fr) * a.Kp + r * 2 * KC2 + 2 * ks + fh];

  // staging: the threads of group sg = tid / TG stage slab sg of the iteration; a thread owns <= CP_MAXCOL columns of every row
  const int TG = 256 / a.SPI, sg = tid / TG, tg = tid - sg * TG;
  int c_off[CP_MAXCOL], c_iwl[CP_MAXCOL], c_lds[CP_MAXCOL];
#pragma unroll
  for (int i = 0; i < CP_MAXCOL; ++i) {
    const int c = tg + TG * i;
    const bool ok = c < a.LWC;
    const uint32_t iwl = ok ? fdiv((uint32_t)c, a.dCi) : 0u;
    const int ci = ok ? c - (int)iwl * Ci : 0;
    c_off[i] = (int)iwl * a.xsW + ci * a.xsC;
    c_iwl[i] = ok ? (int)iwl : (1 << 28);            // (never in range)
    c_lds[i] = ok ? c : -1;
  }
  const int a_lane = fr * sw * Ci + fh;              // A-operand address: + r * PL + 2 ks

  const int it0 = blockIdx.x * a.iters_per_wg;
  const int nit_all = (a.segs + a.SPI - 1) / a.SPI;
  const int it1 = it0 + a.iters_per_wg < nit_all ? it0 + a.iters_per_wg : nit_all;
  float rx[CP_MAXCOL][PR];
  auto fetch = [&](int t) {                          // t: table slot of this group's slab
    const int base = __builtin_amdgcn_readfirstlane(tab[t][0]), bw = __builtin_amdgcn_readfirstlane(tab[t][1]);
    const unsigned masks = (unsigned)__builtin_amdgcn_readfirstlane(tab[t][2]);
#pragma unroll
    for (int i = 0; i < CP_MAXCOL; ++i) {
      const bool cok = (unsigned)(bw + c_iwl[i]) < (unsigned)g.Wi;
#pragma unroll
      for (int r = 0; r < PR; ++r) {
        const int kd = r / KH, kh = r - kd * KH;
        const bool ok = cok && ((masks >> kd) & (masks >> (8 + kh)) & 1u);
        const float v = a.x[ok ? base + kd * a.xsD + kh * a.xsH + c_off[i] : 0];      // unconditional load, clamped address
        rx[i][r] = ok ? v : 0.f;
      }
    }
  };
  auto stage = [&](int buf) {
    float* P = cp_smem + (buf * a.SPI + sg) * patch;
#pragma unroll
    for (int i = 0; i < CP_MAXCOL; ++i)
      if (c_lds[i] >= 0) {
#pragma unroll
        for (int r = 0; r < PR; ++r) P[r * a.PL + c_lds[i]] = rx[i][r];
      }
  };

  int parity = 0;
  for (int c0 = it0; c0 < it1; c0 += CP_TAB / 4) {   // chunks of iterations whose slabs fit the table (SPI <= 4)
    const int cn = it1 - c0 < CP_TAB / 4 ? it1 - c0 : CP_TAB / 4;
    __syncthreads();                                 // (the previous chunk's table and buffers are no longer read)
    if (tid < cn * a.SPI) {
      const int sidx = c0 * a.SPI + tid;
      if (sidx < a.segs) {
        const uint32_t orow = fdiv((uint32_t)sidx, a.dseg), seg = sidx - orow * a.nseg;
        const uint32_t t1 = fdiv(orow, a.dHo), qh = orow - t1 * g.Ho;
        const uint32_t img = fdiv(t1, a.dDo), qd = t1 - img * g.Do;
        const int qw0 = (int)seg * 32;
        const int bd = (int)qd * g.sd - g.pd, bh = (int)qh * g.sh - g.ph, bw = qw0 * sw - g.pw;
        unsigned masks = 0;
        for (int k = 0; k < KD; ++k) masks |= ((unsigned)(bd + k) < (unsigned)g.Di ? 1u : 0u) << k;
        for (int k = 0; k < KH; ++k) masks |= ((unsigned)(bh + k) < (unsigned)g.Hi ? 1u : 0u) << (8 + k);
        const int64_t m0 = (int64_t)orow * g.Wo + qw0;
        tab[tid][0] = (int)img * a.xsN + bd * a.xsD + bh * a.xsH + bw * a.xsW;
        tab[tid][1] = bw; tab[tid][2] = (int)masks;
        tab[tid][3] = g.Wo - qw0 < 32 ? g.Wo - qw0 : 32;
        tab[tid][4] = (int)(uint32_t)m0; tab[tid][5] = (int)(m0 >> 32);
      } else {                                       // past the last slab: nothing in range, nothing stored
        tab[tid][0] = 0; tab[tid][1] = 0; tab[tid][2] = 0; tab[tid][3] = 0; tab[tid][4] = 0; tab[tid][5] = 0;
      }
    }
    __syncthreads();
    fetch(sg); stage(parity);
    __syncthreads();
    for (int t = 0; t < cn; ++t) {
      const int buf = parity;
      if (t + 1 < cn) fetch((t + 1) * a.SPI + sg);
      const float* P = cp_smem + (buf * a.SPI + my_slab) * patch + a_lane;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int r = 0; r < PR; ++r) {
        const float* Pr = P + r * a.PL;
#pragma unroll
        for (int ks = 0; ks < KC2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Pr[2 * ks], wreg[r * KC2 + ks], acc, 0, 0, 0);
      }
      // rows = positions ((r & 3) + 8 (r >> 2) + 4 fh), columns = output channels cb*32 + fr: 128-byte runs per row
      const int slot = t * a.SPI + my_slab;
      const int npos = __builtin_amdgcn_readfirstlane(tab[slot][3]);
      const int64_t m0 = ((int64_t)__builtin_amdgcn_readfirstlane(tab[slot][5]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(tab[slot][4]);
      float* dst = a.out + m0 * Co + cb * 32 + fr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pos = (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (pos < npos) dst[(int64_t)pos * Co] = acc[r];
      }
      if (t + 1 < cn) stage(buf ^ 1);
      parity ^= 1;
      __syncthreads();
    }
  }
}

// Returns 1 if the op was launched here, 0 if it is not this kernel's shape (the caller goes on), < 0 on errors.
int gode_launch_conv_patch_fprop(const gode_igemm_op* op, const int64_t* gs, hipStream_t st) {
  const gode_conv_geom& g = op->g;
  if (op->dir != GODE_FPROP || op->groups == 2 || op->stats || op->scale || op->shift || op->act != GODE_ACT_NONE ||
      op->epilogue != GODE_EPI_RAW || op->tile != 0)
    return 0;
  const int KC = g.kw * g.Ci, Kt = g.kd * g.kh * KC;
  if (g.Ci > 4 || (g.Co != 32 && g.Co != 64 && g.Co != 128) || KC % 2 != 0 || Kt < 8 || g.Wo < 16) return 0;
  CpArgs A;
  A.g = g; A.x = op->src; A.w = op->wpack; A.out = op->out;
  A.xsN = (int)gs[0]; A.xsD = (int)gs[1]; A.xsH = (int)gs[2]; A.xsW = (int)gs[3]; A.xsC = (int)gs[4];
  A.Kp = (Kt + 3) & ~3;
  A.CB = g.Co / 32; A.SPI = 4 / A.CB;
  A.LWC = (31 * g.sw + g.kw) * g.Ci;
  if (A.LWC > CP_MAXCOL * (256 / A.SPI)) return 0;
  const int PL = A.LWC;        // (the row stride only separates different instructions' reads: no padding needed)
  A.PL = PL;
  const int PR = g.kd * g.kh;
  const size_t lds = (size_t)2 * A.SPI * PR * PL * sizeof(float);
  if (lds > 60 * 1024) return 0;
  A.nseg = (g.Wo + 31) / 32;
  const int64_t segs = (int64_t)g.N * g.Do * g.Ho * A.nseg;
  if (segs >= (1ll << 31) || (int64_t)g.N * g.Do * g.Ho * g.Wo * g.Co >= (1ll << 31)) return 0;
  A.segs = (int)segs;
  const int nit = (A.segs + A.SPI - 1) / A.SPI;
  int wgs = nit / 4 < 1 ? 1 : nit / 4;               // >= 4 iterations per workgroup, two workgroups per CU at most
  if (wgs > 512) wgs = 512;
  A.iters_per_wg = (nit + wgs - 1) / wgs;
  wgs = (nit + A.iters_per_wg - 1) / A.iters_per_wg;
  A.dseg = make_fastdiv((uint32_t)A.nseg); A.dHo = make_fastdiv((uint32_t)g.Ho); A.dDo = make_fastdiv((uint32_t)g.Do);
  A.dCi = make_fastdiv((uint32_t)g.Ci);
  void (*kern)(const CpArgs) = nullptr;
  const int KC2 = KC / 2;
#define CP_PICK(D, H, K2) if (g.kd == D && g.kh == H && KC2 == K2) kern = conv_patch_fprop_kernel<D, H, K2>;
  CP_PICK(4, 4, 6)      // Conv3d(3, ., 4): UCF video discriminator
  CP_PICK(1, 4, 6)      // Conv2d(3, ., 4): UCF image discriminator
  CP_PICK(2, 2, 1)      // Conv3d(1, ., 2): MNIST video discriminator
  CP_PICK(1, 4, 2)      // Conv2d(1, ., 4): MNIST image discriminator
#undef CP_PICK
  if (!kern) return 0;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, st, A);
  GODE_LAUNCH_CHECK();
  return 1;
}
