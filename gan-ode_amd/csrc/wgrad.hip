// wgrad.hip -- weight gradients of Conv2d/Conv3d/ConvTranspose2d as a split-K fp32-MFMA GEMM.
//
//   dW[co][tap][ci] = sum over y positions m of  Y[m][co] * X[pos(m,tap)][ci]
// (Y = y-side tensor, dense [M][Co]; X = x-side tensor gathered at the tap offset, zero outside).  The reduction
// index m is the MFMA k index, so both LDS tiles keep the memory order [position][channel] and are read along the
// channel axis (consecutive lanes -> consecutive channels, conflict free; the FAST kernel reads one ds_read_b64 per
// operand and k by giving MFMA row r of sub-tile i the channel 2r+i); no transpose is ever materialised.  Transform-free
// operands (the heavy layers) are staged global -> LDS by global_load_lds_dwordx4 into two buffers.
// The reduction is cut into `splits` slabs (grid.z), each writing its own [Co][taps*Ci] partial; a second kernel
// sums the partials in fixed order and scatters into PyTorch's canonical W[co][ci][taps] layout (deterministic,
// no float atomics).  The BN+activation of the producing layer is fused into whichever operand is an activation.
#include <stdlib.h>
#include "common.h"
#include "conv_geom.h"

struct WgradArgs {
  gode_conv_geom g;
  const float* x; const float* y; const float* scale; const float* shift; float* work;
  int32_t xsN, xsD, xsH, xsW, xsC;
  int32_t act, xform_on_y;
  int32_t M, chunk, Kt, taps;
  FastDiv dWo, dHo, dDo;
  uint32_t x_bytes;     // extent of x (LDS-DMA kernels address the operands as raw buffers: 32-bit byte offsets); 0: too large
};

template <int WM, int WN, int TM, int TN, bool VECX, bool VECY>
__global__ void __launch_bounds__(WM* WN * 64) wgrad_kernel(const WgradArgs a) {
  constexpr int NT = WM * WN * 64, BI = WM * TM * 32, BJ = WN * TN * 32;
  static_assert(NT == 256, "loader assumes 256 threads");
  constexpr int YC = BI / 4, XC = BJ / 4;           // float4 chunks per tile row
  constexpr int YR = NT / YC, XR = NT / XC;         // tile rows per loader pass
  constexpr int YP = 32 / YR, XP = 32 / XR;         // passes over the 32-position slab
  static_assert(YP >= 1 && XP >= 1, "tile too narrow for the loader");
  __shared__ __attribute__((aligned(16))) float smem[2 * 32 * (BI + BJ)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int jblk = blockIdx.x, iblk = blockIdx.y, z = blockIdx.z;
  const gode_conv_geom& g = a.g;
  const int m_begin = z * a.chunk;
  const int m_end = m_begin + a.chunk < a.M ? m_begin + a.chunk : a.M;
  const int nslab = m_end > m_begin ? (m_end - m_begin + 31) >> 5 : 0;

  // ---- per-thread fixed columns
  const int ych = tid % YC, yr0 = tid / YC;
  const int xch = tid % XC, xr0 = tid / XC;
  const int co0 = iblk * BI + ych * 4;
  const int j0 = jblk * BJ + xch * 4;
  int xkd[4], xkh[4], xkw[4], xci[4];
  bool xok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = j0 + e;
    xok[e] = j < a.Kt;
    const int tap = xok[e] ? j / g.Ci : 0;
    xci[e] = xok[e] ? j - tap * g.Ci : 0;
    xkw[e] = tap % g.kw; xkh[e] = (tap / g.kw) % g.kh; xkd[e] = tap / (g.kw * g.kh);
  }
  f32x4 xsc = {1, 1, 1, 1}, xsh = {0, 0, 0, 0}, ysc = {1, 1, 1, 1}, ysh = {0, 0, 0, 0};
  const bool xf = a.scale != nullptr;
  if (xf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!a.xform_on_y && xok[e]) { xsc[e] = a.scale[xci[e]]; xsh[e] = a.shift[xci[e]]; }
      if (a.xform_on_y && co0 + e < g.Co) { ysc[e] = a.scale[co0 + e]; ysh[e] = a.shift[co0 + e]; }
    }
  }
  const int xact = a.xform_on_y ? GODE_ACT_NONE : a.act, yact = a.xform_on_y ? a.act : GODE_ACT_NONE;

  f32x4 ry[YP], rx[XP];

  auto fetch = [&](int slab) {
    const int mb = m_begin + slab * 32;
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      const int m = mb + yr0 + p * YR;
      f32x4 v = {0, 0, 0, 0};
      if (m < m_end) {
        if (VECY) {
          if (co0 < g.Co) v = *reinterpret_cast<const f32x4*>(a.y + (int64_t)m * g.Co + co0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (co0 + e < g.Co) v[e] = a.y[(int64_t)m * g.Co + co0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (co0 + e < g.Co) ? gode_act(v[e] * ysc[e] + ysh[e], yact) : 0.f;
      }
      ry[p] = v;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int m = mb + xr0 + p * XR;
      f32x4 v = {0, 0, 0, 0};
      if (m < m_end) {
        const int qw = m % g.Wo; int t = m / g.Wo;
        const int qh = t % g.Ho; t /= g.Ho;
        const int qd = t % g.Do; const int img = t / g.Do;
        const int bd = qd * g.sd - g.pd, bh = qh * g.sh - g.ph, bw = qw * g.sw - g.pw;
        const int base = img * a.xsN;
        if (VECX) {
          const int id = bd + xkd[0], ih = bh + xkh[0], iw = bw + xkw[0];
          if (xok[0] && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi) {
            v = *reinterpret_cast<const f32x4*>(a.x + base + id * a.xsD + ih * a.xsH + iw * a.xsW + xci[0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gode_act(v[e] * xsc[e] + xsh[e], xact);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int id = bd + xkd[e], ih = bh + xkh[e], iw = bw + xkw[e];
            if (xok[e] && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi)
              v[e] = gode_act(a.x[base + id * a.xsD + ih * a.xsH + iw * a.xsW + xci[e] * a.xsC] * xsc[e] + xsh[e], xact);
          }
        }
      }
      rx[p] = v;
    }
  };
  auto stage = [&](int buf) {
    float* Ys = smem + buf * 32 * (BI + BJ);
    float* Xs = Ys + 32 * BI;
#pragma unroll
    for (int p = 0; p < YP; ++p) *reinterpret_cast<f32x4*>(Ys + (yr0 + p * YR) * BI + ych * 4) = ry[p];
#pragma unroll
    for (int p = 0; p < XP; ++p) *reinterpret_cast<f32x4*>(Xs + (xr0 + p * XR) * BJ + xch * 4) = rx[p];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nslab > 0) {
    fetch(0);
    stage(0);
  }
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) fetch(s + 1);
    const float* Ys = smem + buf * 32 * (BI + BJ) + fh * BI + wm * TM * 32 + fr;
    const float* Xs = smem + buf * 32 * (BI + BJ) + 32 * BI + fh * BJ + wn * TN * 32 + fr;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = Ys[ks * 2 * BI + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Xs[ks * 2 * BJ + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nslab) stage(buf ^ 1);
    __syncthreads();
  }

  float* dst = a.work + (int64_t)z * g.Co * a.Kt;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = jblk * BJ + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = iblk * BI + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.Co && col < a.Kt) dst[(int64_t)row * a.Kt + col] = acc[i][j][r];
      }
    }
}

// FAST path (vector loads on both operands): branch-free loop body (clamped loads + masks, magic-number position
// decode).  GL=false: register staging into ONE LDS buffer (32 KB at 128x128), BN/activation fused into the load.

// GL: transform-free operands go global -> LDS by `global_load_lds_dwordx4` (the [position][channel] image is already
// lane-linear: thread t of a loader pass owns 16-byte chunk t), two LDS buffers, one barrier per 32-position slab.
template <int WM, int WN, int TM, int TN, bool GL>
__global__ void __launch_bounds__(WM* WN * 64) wgrad_fast_kernel(const WgradArgs a) {
  constexpr int NT = WM * WN * 64, BI = WM * TM * 32, BJ = WN * TN * 32;
  constexpr int YC = BI / 4, XC = BJ / 4, YR = NT / YC, XR = NT / XC, YP = 32 / YR, XP = 32 / XR;
  static_assert(YP >= 1 && XP >= 1, "tile too narrow for the loader");
  static_assert(!GL || (YC % 64 == 0 || 64 % YC == 0), "loader pass must be lane-linear");
  constexpr int BUF = 32 * (BI + BJ);
  __shared__ __attribute__((aligned(16))) float smem[(GL ? 2 : 1) * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int jblk = blockIdx.x, iblk = blockIdx.y, z = blockIdx.z;
  const gode_conv_geom& g = a.g;
  const int m_begin = z * a.chunk;
  const int m_end = m_begin + a.chunk < a.M ? m_begin + a.chunk : a.M;
  const int nslab = m_end > m_begin ? (m_end - m_begin + 31) >> 5 : 0;

  const int ych = tid % YC, yr0 = tid / YC, xch = tid % XC, xr0 = tid / XC;
  const int co0 = iblk * BI + ych * 4, j0 = jblk * BJ + xch * 4;
  const bool yok = co0 < g.Co, xok = j0 < a.Kt;
  const int tap = xok ? j0 / g.Ci : 0;
  const int xci = xok ? j0 - tap * g.Ci : 0;
  const int xkw = tap % g.kw, xkh = (tap / g.kw) % g.kh, xkd = tap / (g.kw * g.kh);
  f32x4 xsc = {1, 1, 1, 1}, xsh = {0, 0, 0, 0}, ysc = {1, 1, 1, 1}, ysh = {0, 0, 0, 0};
  if (a.scale != nullptr) {
    if (!a.xform_on_y && xok) { xsc = *reinterpret_cast<const f32x4*>(a.scale + xci); xsh = *reinterpret_cast<const f32x4*>(a.shift + xci); }
    if (a.xform_on_y && yok) { ysc = *reinterpret_cast<const f32x4*>(a.scale + co0); ysh = *reinterpret_cast<const f32x4*>(a.shift + co0); }
  }
  const float nslope = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);
  const float xneg = a.xform_on_y ? 1.f : nslope, yneg = a.xform_on_y ? nslope : 1.f;
  const float* yptr = a.y + (yok ? co0 : 0);
  const float* xptr = a.x + xci;

  f32x4 ry[YP], rx[XP];
  unsigned ymask = 0, xmask = 0;
  auto fetch = [&](int slab) {
    const int mb = m_begin + slab * 32;
    ymask = 0; xmask = 0;
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      const int m = mb + yr0 + p * YR;
      const bool ok = m < m_end && yok;
      ry[p] = *reinterpret_cast<const f32x4*>(yptr + (int64_t)(ok ? m : m_begin) * g.Co);
      ymask |= (ok ? 1u : 0u) << p;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const uint32_t m = mb + xr0 + p * XR;
      const uint32_t t1 = fdiv(m, a.dWo), qw = m - t1 * g.Wo;
      const uint32_t t2 = fdiv(t1, a.dHo), qh = t1 - t2 * g.Ho;
      const uint32_t img = fdiv(t2, a.dDo), qd = t2 - img * g.Do;
      const int id = (int)qd * g.sd - g.pd + xkd, ih = (int)qh * g.sh - g.ph + xkh, iw = (int)qw * g.sw - g.pw + xkw;
      const bool ok = (int)m < m_end && xok && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi &&
                      (unsigned)iw < (unsigned)g.Wi;
      const int off = ok ? (int)img * a.xsN + id * a.xsD + ih * a.xsH + iw * a.xsW : 0;
      rx[p] = *reinterpret_cast<const f32x4*>(xptr + off);
      xmask |= (ok ? 1u : 0u) << p;
    }
  };
  // LDS-DMA issue of one slab (GL): the loader's (row, chunk) of a pass is tid-linear, so wave w writes 1 KiB at pass base +
  // 256*w floats.  Both operands are raw buffers.  y: the resource is re-based to the slab (three scalar operations), a lane's
  // offset inside the slab is a constant, and the rows past m_end fall outside the re-based extent (zeros from the range
  // check).  x: the position (image, depth, row, column) of a lane's rows is carried from slab to slab -- a slab advances
  // every row by 32 positions, i.e. by fixed digit increments with at most one carry per digit -- instead of being decoded
  // from the row number with three divisions per load; padding taps and tail rows get an offset beyond the buffer.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  BufRsrc rsrcX;
  int yvoff[YP];
  int xm[XP], xiw[XP], xih[XP], xid[XP], xoff[XP];
  // digit increments of a 32-position step and the offset corrections of the carries (wave-uniform)
  const int stepw = 32 % g.Wo, c1 = 32 / g.Wo, steph = c1 % g.Ho, c2 = c1 / g.Ho, stepd = c2 % g.Do, stepn = c2 / g.Do;
  const int spanw = g.Wo * g.sw, spanh = g.Ho * g.sh, spand = g.Do * g.sd;
  const int off_step = stepw * g.sw * a.xsW + steph * g.sh * a.xsH + stepd * g.sd * a.xsD + stepn * a.xsN;
  const int off_cw = g.sh * a.xsH - spanw * a.xsW, off_ch = g.sd * a.xsD - spanh * a.xsH, off_cd = a.xsN - spand * a.xsD;
  const int iw_lim = xkw - g.pw + spanw, ih_lim = xkh - g.ph + spanh, id_lim = xkd - g.pd + spand;
  if (GL) {
    rsrcX = make_buf_rsrc(a.x, a.x_bytes);
#pragma unroll
    for (int p = 0; p < YP; ++p) yvoff[p] = yok ? ((yr0 + p * YR) * g.Co + co0) * 4 : (int)0x80000000;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const uint32_t m = m_begin + xr0 + p * XR;
      const uint32_t t1 = fdiv(m, a.dWo), qw = m - t1 * g.Wo;
      const uint32_t t2 = fdiv(t1, a.dHo), qh = t1 - t2 * g.Ho;
      const uint32_t img = fdiv(t2, a.dDo), qd = t2 - img * g.Do;
      xm[p] = (int)m;
      xid[p] = (int)qd * g.sd - g.pd + xkd; xih[p] = (int)qh * g.sh - g.ph + xkh; xiw[p] = (int)qw * g.sw - g.pw + xkw;
      xoff[p] = (int)img * a.xsN + xid[p] * a.xsD + xih[p] * a.xsH + xiw[p] * a.xsW + xci;
    }
  }
  int dma_mb = m_begin;          // first position of the slab the next dma() call fetches (slabs are fetched in order)
  auto dma = [&](int buf) {
    float* Ys = smem + buf * BUF + wave_u * 256;
    float* Xs = smem + buf * BUF + 32 * BI + wave_u * 256;
    const int left = m_end - dma_mb;
    const BufRsrc rsrcY = make_buf_rsrc(a.y + (int64_t)dma_mb * g.Co, (uint32_t)((left > 32 ? 32 : left) * g.Co) * 4u);
#pragma unroll
    for (int p = 0; p < YP; ++p) buf_dma16(rsrcY, Ys + p * YR * BI, yvoff[p], 0);
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const bool ok = xm[p] < m_end && xok && (unsigned)xid[p] < (unsigned)g.Di && (unsigned)xih[p] < (unsigned)g.Hi &&
                      (unsigned)xiw[p] < (unsigned)g.Wi;
      buf_dma16(rsrcX, Xs + p * XR * BJ, ok ? xoff[p] * 4 : (int)0x80000000, 0);
      // advance this row by 32 positions
      xm[p] += 32;
      int o = xoff[p] + off_step;
      int iw = xiw[p] + stepw * g.sw;
      const bool cw = iw >= iw_lim;
      iw -= cw ? spanw : 0; o += cw ? off_cw : 0;
      int ih = xih[p] + steph * g.sh + (cw ? g.sh : 0);
      const bool ch = ih >= ih_lim;
      ih -= ch ? spanh : 0; o += ch ? off_ch : 0;
      int id = xid[p] + stepd * g.sd + (ch ? g.sd : 0);
      const bool cd = id >= id_lim;
      id -= cd ? spand : 0; o += cd ? off_cd : 0;
      xiw[p] = iw; xih[p] = ih; xid[p] = id; xoff[p] = o;
    }
    dma_mb += 32;
  };
  auto stage = [&]() {
    float* Ys = smem;
    float* Xs = smem + 32 * BI;
#pragma unroll
    for (int p = 0; p < YP; ++p) {
      f32x4 v;
      const bool ok = (ymask >> p) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) { float t = ry[p][e] * ysc[e] + ysh[e]; t = t > 0.f ? t : t * yneg; v[e] = ok ? t : 0.f; }
      *reinterpret_cast<f32x4*>(Ys + (yr0 + p * YR) * BI + ych * 4) = v;
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      f32x4 v;
      const bool ok = (xmask >> p) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) { float t = rx[p][e] * xsc[e] + xsh[e]; t = t > 0.f ? t : t * xneg; v[e] = ok ? t : 0.f; }
      *reinterpret_cast<f32x4*>(Xs + (xr0 + p * XR) * BJ + xch * 4) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // MFMA row r of sub-tile i is output channel TM*r + i (and column c of sub-tile j is weight column TN*c + j): a
  // lane then reads its TM (TN) operands of one k as ONE ds_read_b32/b64 instead of TM (TN) separate reads
  typedef float __attribute__((ext_vector_type(TM))) av_t;
  typedef float __attribute__((ext_vector_type(TN))) bv_t;
  const int fr = lane & 31, fh = lane >> 5;
  const float* Yr = smem + fh * BI + wm * TM * 32 + fr * TM;
  const float* Xr = smem + 32 * BI + fh * BJ + wn * TN * 32 + fr * TN;
  auto mma_slab = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const av_t af = *reinterpret_cast<const av_t*>(Yr + buf * BUF + ks * 2 * BI);
      const bv_t bf = *reinterpret_cast<const bv_t*>(Xr + buf * BUF + ks * 2 * BJ);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  if (GL) {
    // same schedule as igemm_fast_kernel's LDS-DMA loop: fragments one group of 16 MFMAs (4 k pairs) ahead of the matrix
    // pipe, the slab barrier before the last group
    av_t fa[2][4]; bv_t fb[2][4];
    auto frag = [&](int buf, int kg, int set) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fa[set][q] = *reinterpret_cast<const av_t*>(Yr + buf * BUF + (kg * 4 + q) * 2 * BI);
        fb[set][q] = *reinterpret_cast<const bv_t*>(Xr + buf * BUF + (kg * 4 + q) * 2 * BJ);
      }
    };
    auto mma = [&](int set) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][q][i], fb[set][q][j], acc[i][j], 0, 0, 0);
    };
    if (nslab > 0) {
      dma(0);
      __syncthreads();                   // (waits vmcnt(0)) slab 0 has landed
      frag(0, 0, 0);
      for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) dma(buf ^ 1);
        frag(buf, 1, 1); __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0);
        frag(buf, 2, 0); __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0);
        frag(buf, 3, 1); __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                 // my reads of slab s are done, slab s+1 has landed (vmcnt(0) + barrier)
        frag(buf ^ 1, 0, 0);             // (after the last slab: a stale buffer, never used)
        __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (nslab > 0) {
    fetch(0);
    for (int s = 0; s + 1 < nslab; ++s) {
      stage();
      __syncthreads();
      fetch(s + 1);
      mma_slab(0);
      __syncthreads();
    }
    stage();
    __syncthreads();
    mma_slab(0);
  }

  float* dst = a.work + (int64_t)z * g.Co * a.Kt;
  const int col0 = jblk * BJ + wn * TN * 32 + (lane & 31) * TN;      // Kt % 4 == 0 here, so col0 < Kt covers the pair
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = iblk * BI + wm * TM * 32 + TM * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) + i;
      if (row < g.Co && col0 < a.Kt) {
        bv_t v;
#pragma unroll
        for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
        *reinterpret_cast<bv_t*>(dst + (int64_t)row * a.Kt + col0) = v;
      }
    }
}

// THIN path: very few weight columns (Kt = taps*Ci <= 16: first discriminator layers with 1/3 input channels, the
// generator's 64->1 output layer).  An MFMA tile would be >90% padding; instead every thread keeps 4 output channels
// x Kt taps in registers, strides over positions, and the workgroup combines its position lanes through LDS.  Output:
// one [Co][Kt] slab per workgroup in the same layout the split-K reduce consumes.
#define THIN_MAXKT 16
__global__ void __launch_bounds__(256) wgrad_thin_kernel(const WgradArgs a, int chunk_thin) {
  const gode_conv_geom& g = a.g;
  const int C4 = g.Co >> 2;                 // float4 channel lanes (Co % 4 == 0, C4 <= 256)
  const int CL = C4 < 64 ? C4 : 64;
  const int ML = 256 / CL;                  // position lanes
  const int tid = threadIdx.x, cl = tid % CL, ml = tid / CL;
  const int Kt = a.Kt;
  __shared__ __attribute__((aligned(16))) float red[4][256][4];   // 16 KB: taps are combined four at a time
  const float nslope = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);
  const float xneg = a.xform_on_y ? 1.f : nslope, yneg = a.xform_on_y ? nslope : 1.f;
  const int m0 = blockIdx.x * chunk_thin;
  const int m1 = m0 + chunk_thin < a.M ? m0 + chunk_thin : a.M;
  for (int cg = 0; cg < C4; cg += CL) {     // channel groups (one pass unless Co > 256)
    const int c4 = cg + cl;
    const bool cok = ml < ML && c4 < C4;
    f32x4 acc[THIN_MAXKT];
#pragma unroll
    for (int j = 0; j < THIN_MAXKT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ysc = {1, 1, 1, 1}, ysh = {0, 0, 0, 0};
    if (cok && a.scale && a.xform_on_y) { ysc = *reinterpret_cast<const f32x4*>(a.scale + c4 * 4); ysh = *reinterpret_cast<const f32x4*>(a.shift + c4 * 4); }
    if ((CL & 15) == 0) {
      // Shared gathers (round 3): the 16 channel lanes of a DPP row work on the SAME position, and each used to gather all
      // Kt x values itself (Kt bounds-checked scalar loads of the same addresses in 16 lanes: the kernel was bound by those
      // ~20 instructions per tap, 39 us for the 27 MB of the MNIST video discriminator's first layer).  Now lane j of a row
      // gathers tap j only (its tap coordinates are fixed for the whole launch) and the row reads the Kt values from each
      // other (ds_bpermute): ~2.5x fewer instructions per position.
      const int lj = tid & 15;
      const bool tap_ok = lj < Kt;
      int tci = 0, tkw = 0, tkh = 0, tkd = 0;
      if (tap_ok) { int tap = lj / g.Ci; tci = lj - tap * g.Ci; tkw = tap % g.kw; tkh = (tap / g.kw) % g.kh; tkd = tap / (g.kw * g.kh); }
      float xsc = 1.f, xsh = 0.f;
      if (tap_ok && a.scale && !a.xform_on_y) { xsc = a.scale[tci]; xsh = a.shift[tci]; }
      const bool xf = !a.xform_on_y;
      for (int m = m0 + ml; m < m1; m += ML) {       // (ml < ML always here: CL * ML == 256)
        f32x4 yv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < C4) {
          yv = *reinterpret_cast<const f32x4*>(a.y + (int64_t)m * g.Co + c4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { float t = yv[e] * ysc[e] + ysh[e]; yv[e] = t > 0.f ? t : t * yneg; }
        }
        const uint32_t t1 = fdiv((uint32_t)m, a.dWo), qw = m - t1 * g.Wo;
        const uint32_t t2 = fdiv(t1, a.dHo), qh = t1 - t2 * g.Ho;
        const uint32_t img = fdiv(t2, a.dDo), qd = t2 - img * g.Do;
        const int id = (int)qd * g.sd - g.pd + tkd, ih = (int)qh * g.sh - g.ph + tkh, iw = (int)qw * g.sw - g.pw + tkw;
        float xmine = 0.f;
        if (tap_ok && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi) {
          xmine = a.x[(int)img * a.xsN + id * a.xsD + ih * a.xsH + iw * a.xsW + tci * a.xsC];
          if (xf) { xmine = xmine * xsc + xsh; xmine = xmine > 0.f ? xmine : xmine * xneg; }
        }
#pragma unroll
        for (int j = 0; j < THIN_MAXKT; ++j) {
          if (j < Kt) acc[j] += yv * __shfl(xmine, j, 16);      // (Kt is launch-uniform)
        }
      }
    } else if (cok) {
      for (int m = m0 + ml; m < m1; m += ML) {
        f32x4 yv = *reinterpret_cast<const f32x4*>(a.y + (int64_t)m * g.Co + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { float t = yv[e] * ysc[e] + ysh[e]; yv[e] = t > 0.f ? t : t * yneg; }
        const uint32_t t1 = fdiv((uint32_t)m, a.dWo), qw = m - t1 * g.Wo;
        const uint32_t t2 = fdiv(t1, a.dHo), qh = t1 - t2 * g.Ho;
        const uint32_t img = fdiv(t2, a.dDo), qd = t2 - img * g.Do;
        const int bd = (int)qd * g.sd - g.pd, bh = (int)qh * g.sh - g.ph, bw = (int)qw * g.sw - g.pw;
        int ci = 0, kw = 0, kh = 0, kd = 0;
#pragma unroll
        for (int j = 0; j < THIN_MAXKT; ++j) {
          if (j < Kt) {
            const int id = bd + kd, ih = bh + kh, iw = bw + kw;
            float xv = 0.f;
            if ((unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi) {
              xv = a.x[(int)img * a.xsN + id * a.xsD + ih * a.xsH + iw * a.xsW + ci * a.xsC];
              if (!a.xform_on_y) {
                const float sc = a.scale ? a.scale[ci] : 1.f, sh = a.scale ? a.shift[ci] : 0.f;
                xv = xv * sc + sh; xv = xv > 0.f ? xv : xv * xneg;
              }
            }
            acc[j] += yv * xv;
            if (++ci == g.Ci) { ci = 0; if (++kw == g.kw) { kw = 0; if (++kh == g.kh) { kh = 0; ++kd; } } }
          }
        }
      }
    }
    // combine the position lanes through LDS (fixed order q = 0..ML-1): thread (tap j, channel lane c) sums its column
    float* dst = a.work + (int64_t)blockIdx.x * g.Co * Kt;
#pragma unroll
    for (int j0 = 0; j0 < THIN_MAXKT; j0 += 4) {
      if (j0 < Kt) {           // wave-uniform
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (j0 + jj < Kt) *reinterpret_cast<f32x4*>(red[jj][tid]) = cok ? acc[j0 + jj] : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        const int nj = Kt - j0 < 4 ? Kt - j0 : 4;
        for (int i = tid; i < nj * CL; i += 256) {
          const int jj = i / CL, c = i - jj * CL;
          if (cg + c < C4) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < ML; ++q) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(red[jj][q * CL + c]);
              s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(int64_t)((cg + c) * 4 + e) * Kt + j0 + jj] = s[e];
          }
        }
      }
    }
  }
}


// PATCH path: first layers of the discriminators and the generator's RGB head at UCF size (Ci <= 4 but 16-64 taps, so
// Kt = 48 / 192: too wide for the THIN path, and an im2col of a 3-channel tensor is one scalar gather per element -- the
// generic kernel spends its time there: 17 TFLOP/s).  Here a slab is (up to) 32 consecutive positions of ONE output row;
// the kd*kh input rows they touch are staged in LDS once, [row][w][ci] with any global strides, and the MFMA's X
// operand is read straight out of that patch: column j = (row r, kw, ci) of position q sits at r*PL + (q*sw + kw)*Ci + ci,
// so the per-lane part of the address is fixed and the position part is an immediate.  No im2col tile exists anywhere.
// The padded row length PL = kw*Ci (mod 32) lets the 32 column lanes of a ds_read_b32 fall into distinct banks.
#define WP_MAXB 4          // 32x32 output blocks per wave
#define WP_MAXE 16         // patch elements staged per thread and slab
struct WpArgs {
  WgradArgs w;
  int32_t RB, CB;          // 32-row blocks of Co, 32-column blocks of Kt
  int32_t PR, PL, LW;      // patch rows (kd*kh), padded row length (floats), w positions per row ((32-1)*sw + kw)
  int32_t nseg;            // 32-position segments per output row
  FastDiv dseg;
  int32_t segs, segs_per_wg;
  int32_t yvec;            // y rows may be read as float4
};

#define WP_TAB 128         // slabs whose corners are decoded at a time (LDS table)
// NBW: 32x32 output blocks per wave.  Wave w owns row block w % RB and the column blocks w / RB + (4 / RB) * k, so one
// Y fragment serves all its MFMAs of a k step; column blocks past CB compute zeros and are not stored.  CO = Co.
template <int NBW, int CO>
__global__ void __launch_bounds__(256) wgrad_patch_kernel(const WpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float wp_smem[];
  constexpr int RB = CO / 32, CSTEP = 4 / RB, YC = CO / 4;
  __shared__ int tab[WP_TAB][6];                     // per slab: x corner offset, bw, masks, valid positions, first y row (lo, hi)
  const WgradArgs& a = p.w;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int Ci = a.g.Ci, Kt = a.Kt, kh = a.g.kh, kw = a.g.kw, sw = a.g.sw;
  const int patch = p.PR * p.PL;
  const int bufsz = patch + 32 * CO + 4;             // [patch | Y slab | dump slot of the threads without an element]
  const int nelem = p.PR * p.LW * Ci;
  const int ne = (nelem + 255) >> 8;                 // patch elements per thread (uniform)

  // per-thread patch elements, fixed over the slabs: global offset from the slab's corner, LDS slot, and (kd, kh, local w)
  // packed 5:5:22.  A thread past the last element gets kd = 31: never in range, its zero goes to the dump slot.
  int e_goff[WP_MAXE], e_lds[WP_MAXE], e_key[WP_MAXE];
#pragma unroll
  for (int i = 0; i < WP_MAXE; ++i) {
    const int e = tid + 256 * i;
    const bool ok = e < nelem;
    const int r = ok ? e / (p.LW * Ci) : 0, rem = ok ? e - r * (p.LW * Ci) : 0;
    const int iwl = rem / Ci, ci = rem - iwl * Ci;
    const int kdi = r / kh, khi = r - kdi * kh;
    e_goff[i] = kdi * a.xsD + khi * a.xsH + iwl * a.xsW + ci * a.xsC;
    e_lds[i] = ok ? r * p.PL + iwl * Ci + ci : patch + 32 * CO;
    e_key[i] = (ok ? kdi : 31) | (khi << 5) | (iwl << 10);
  }
  const int rb = wave % RB;
  int b_x[NBW], b_colj[NBW];
  bool b_col[NBW];
#pragma unroll
  for (int k = 0; k < NBW; ++k) {
    const int cb = wave / RB + CSTEP * k;
    const int j = cb * 32 + fr;
    b_col[k] = cb < p.CB && j < Kt;
    const int tap = b_col[k] ? j / Ci : 0, ci = b_col[k] ? j - tap * Ci : 0;
    const int kwi = tap % kw, r = tap / kw;
    b_x[k] = r * p.PL + kwi * Ci + ci + fh * sw * Ci;
    b_colj[k] = j;
  }
  const int y_off = patch + fh * CO + rb * 32 + fr;
  const float yneg = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);
  const bool yxf = a.scale != nullptr;               // (host: only with xform_on_y)
  constexpr int NYC = (32 * YC + 255) / 256;         // float4 chunks of the Y slab per thread (256 % YC == 0)
  const int yc4 = tid % YC, ypos0 = tid / YC;        // chunk i of this thread: position ypos0 + i * (256 / YC), same channels
  f32x4 ysc = {1, 1, 1, 1}, ysh = {0, 0, 0, 0};
  if (yxf) { ysc = *reinterpret_cast<const f32x4*>(a.scale + yc4 * 4); ysh = *reinterpret_cast<const f32x4*>(a.shift + yc4 * 4); }

  f32x16 acc[NBW];
#pragma unroll
  for (int k = 0; k < NBW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  const int s0 = blockIdx.x * p.segs_per_wg;
  const int s1 = s0 + p.segs_per_wg < p.segs ? s0 + p.segs_per_wg : p.segs;
  float rx[WP_MAXE];
  f32x4 ry[NYC];
  auto fetch = [&](int t) {                          // t: slot of the slab in the table
    const int base = __builtin_amdgcn_readfirstlane(tab[t][0]), bw = __builtin_amdgcn_readfirstlane(tab[t][1]);
    const unsigned masks = __builtin_amdgcn_readfirstlane(tab[t][2]);     // depth taps in range | row taps in range << 5
    const int npos = __builtin_amdgcn_readfirstlane(tab[t][3]);
    const int64_t m0 = ((int64_t)__builtin_amdgcn_readfirstlane(tab[t][5]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane(tab[t][4]);
#pragma unroll
    for (int i = 0; i < WP_MAXE; ++i) {
      if (i < ne) {                                  // uniform
        const unsigned key = (unsigned)e_key[i];
        const bool ok = ((masks >> (key & 31)) & (masks >> (5 + ((key >> 5) & 31))) & 1u) && (unsigned)(bw + (int)(key >> 10)) < (unsigned)a.g.Wi;
        const float v = a.x[ok ? base + e_goff[i] : 0];      // unconditional load, clamped address
        rx[i] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NYC; ++i) {
      const int pos = ypos0 + i * (256 / YC);
      const bool ok = pos < npos;
      const float* src = a.y + (m0 + (ok ? pos : 0)) * CO + yc4 * 4;
      f32x4 v;
      if (p.yvec) v = *reinterpret_cast<const f32x4*>(src);
      else { v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t2 = v[e];
        if (yxf) { t2 = t2 * ysc[e] + ysh[e]; t2 = fmaxf(t2, t2 * yneg); }
        v[e] = ok ? t2 : 0.f;
      }
      ry[i] = v;
    }
  };
  auto stage = [&](int buf) {
    float* P = wp_smem + buf * bufsz;
#pragma unroll
    for (int i = 0; i < WP_MAXE; ++i) if (i < ne) P[e_lds[i]] = rx[i];
#pragma unroll
    for (int i = 0; i < NYC; ++i)
      if (ypos0 + i * (256 / YC) < 32) *reinterpret_cast<f32x4*>(P + patch + (tid + 256 * i) * 4) = ry[i];
  };

  const int qstep = 2 * sw * Ci;
  int parity = 0;
  for (int c0 = s0; c0 < s1; c0 += WP_TAB) {         // chunks of WP_TAB slabs
    const int cn = s1 - c0 < WP_TAB ? s1 - c0 : WP_TAB;
    __syncthreads();                                 // (the previous chunk's table and LDS buffers are no longer read)
    if (tid < cn) {
      const uint32_t sidx = c0 + tid;
      const uint32_t orow = fdiv(sidx, p.dseg), seg = sidx - orow * p.nseg;
      const uint32_t t1 = fdiv(orow, a.dHo), qh = orow - t1 * a.g.Ho;
      const uint32_t img = fdiv(t1, a.dDo), qd = t1 - img * a.g.Do;
      const int qw0 = (int)seg * 32;
      const int bd = (int)qd * a.g.sd - a.g.pd, bh = (int)qh * a.g.sh - a.g.ph, bw = qw0 * sw - a.g.pw;
      unsigned masks = 0;
      for (int k = 0; k < a.g.kd; ++k) masks |= ((unsigned)(bd + k) < (unsigned)a.g.Di ? 1u : 0u) << k;
      for (int k = 0; k < kh; ++k) masks |= ((unsigned)(bh + k) < (unsigned)a.g.Hi ? 1u : 0u) << (5 + k);
      const int64_t m0 = (int64_t)orow * a.g.Wo + qw0;
      tab[tid][0] = (int)img * a.xsN + bd * a.xsD + bh * a.xsH + bw * a.xsW;
      tab[tid][1] = bw; tab[tid][2] = (int)masks;
      tab[tid][3] = a.g.Wo - qw0 < 32 ? a.g.Wo - qw0 : 32;
      tab[tid][4] = (int)(uint32_t)m0; tab[tid][5] = (int)(m0 >> 32);
    }
    __syncthreads();
    fetch(0); stage(parity);
    __syncthreads();
    for (int t = 0; t < cn; ++t) {
      const int buf = parity;
      if (t + 1 < cn) fetch(t + 1);
      const float* P = wp_smem + buf * bufsz;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float av = P[y_off + ks * 2 * CO];
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
          const float xv = P[b_x[k] + ks * qstep];
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_col[k] ? xv : 0.f, acc[k], 0, 0, 0);
        }
      }
      if (t + 1 < cn) stage(buf ^ 1);
      parity ^= 1;
      __syncthreads();
    }
  }

  float* dst = a.work + (int64_t)blockIdx.x * CO * Kt;
#pragma unroll
  for (int k = 0; k < NBW; ++k) {
    if (b_col[k]) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[(int64_t)(rb * 32 + 4 * fh + (r & 3) + 8 * (r >> 2)) * Kt + b_colj[k]] = acc[k][r];
    }
  }
}

// sums the split-K slabs in fixed order (fp64 running sum) and scatters into the canonical layout; VEC4: four
// consecutive (co, j) entries per thread with float4 slab reads (Kt % 4 == 0)
template <bool VEC4>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* work, float* dw, int Co, int Ci, int taps,
                                                           int splits, const int32_t* co_perm, int accumulate) {
  const int Kt = Ci * taps;
  const int64_t total = (int64_t)Co * Kt;
  constexpr int V = VEC4 ? 4 : 1;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; i < total; i += (int64_t)gridDim.x * 256 * V) {
    double sd[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sd[e] = 0.0;
#pragma unroll 8
    for (int zz = 0; zz < splits; ++zz) {
      if (VEC4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(work + (int64_t)zz * total + i);
#pragma unroll
        for (int e = 0; e < V; ++e) sd[e] += (double)v[e];
      } else {
        sd[0] += (double)work[(int64_t)zz * total + i];
      }
    }
    int co = (int)(i / Kt);
    const int j0 = (int)(i - (int64_t)co * Kt);
    if (co_perm) { co = co_perm[co]; if (co < 0) continue; }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int j = j0 + e, tap = j / Ci, ci = j - tap * Ci;
      float* dptr = dw + ((int64_t)co * Ci + ci) * taps + tap;
      const float sv = (float)sd[e];
      *dptr = accumulate ? *dptr + sv : sv;
    }
  }
}

// few outputs, many slabs (thin path, up to 1024 slabs): 16 outputs x 16 slab lanes per workgroup -- each lane's chain
// is splits/16 loads, 8 in flight (64 x 4 lanes meant 256 serial loads per thread: ~30 us of pure latency);
// fixed-order combine through LDS
#define RS_OUT 16
#define RS_LANES 16
__global__ void __launch_bounds__(256) wgrad_reduce_small_kernel(const float* work, float* dw, int Co, int Ci, int taps,
                                                                 int splits, const int32_t* co_perm, int accumulate) {
  __shared__ double red[RS_LANES][RS_OUT];
  const int Kt = Ci * taps, total = Co * Kt;
  const int o = threadIdx.x & (RS_OUT - 1), ln = threadIdx.x / RS_OUT;
  const int i = blockIdx.x * RS_OUT + o;
  double sd = 0.0;
  if (i < total) {
#pragma unroll 8
    for (int zz = ln; zz < splits; zz += RS_LANES) sd += (double)work[(int64_t)zz * total + i];
  }
  red[ln][o] = sd;
  __syncthreads();
  if (ln == 0 && i < total) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < RS_LANES; ++q) tot += red[q][o];
    const float sv = (float)tot;
    int co = i / Kt;
    const int j = i - co * Kt, tap = j / Ci, ci = j - tap * Ci;
    if (co_perm) { co = co_perm[co]; if (co < 0) return; }
    float* dptr = dw + ((int64_t)co * Ci + ci) * taps + tap;
    *dptr = accumulate ? *dptr + sv : sv;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 3: 256 output channels x 128 weight columns, 8 waves (FAST kernel only): 25 % fewer operand bytes per MFMA than 128 x 128.
// Measured: decoder layers 332 -> 310 and 313 -> 304 us, MNIST video-D layers 2 / 3 at N = 64 119 -> 112 and 144 -> 135 us;
// a 0.6-GFLOP problem (image-D layer 2) loses 15 %, hence the size floor.
static int wg_tile(const gode_conv_geom& g) {
  const int64_t macs = (int64_t)g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw;
  static const char* tenv = getenv("GODE_WGRAD_TILE");       // tuning knob (scripts/exp/sweep_wgrad.py): 2 = 128x128, 3 = 256x128, 4 = 128x256
  if (tenv) {
    const int t = atoi(tenv);
    if (t == 2 && g.Co > 64) return 2;
    if (t == 3 && g.Co % 256 == 0 && g.Ci % 4 == 0) return 3;
    if (t == 4 && g.Co % 128 == 0 && g.Ci % 4 == 0 && (g.kd * g.kh * g.kw * g.Ci) % 256 == 0) return 4;
  }
  if (g.Co % 256 == 0 && g.Ci % 4 == 0 && macs >= (1ll << 31)) return 3;
  // 128 x 256, 8 waves: the decoder's last stride-2 layer (326 -> 321 us); with only two column tiles (MNIST video-D
  // layer 1, Kt = 512) it needs 128 position splits and loses 12 %
  if (g.Co == 128 && g.Ci % 4 == 0 && (g.kd * g.kh * g.kw * g.Ci) % 256 == 0 && g.kd * g.kh * g.kw * g.Ci >= 1024 && macs >= (1ll << 31)) return 4;
  return g.Co <= 32 ? 0 : (g.Co <= 64 ? 1 : 2);
}
static int wg_bi(int t) { return t == 0 ? 32 : (t == 1 ? 64 : (t == 3 ? 256 : 128)); }
static int wg_bj(int t) { return t == 4 ? 256 : 128; }

// Position splits of the weight-gradient GEMM.  All workgroups carry equal work, so the launch time is the busiest
// CU's share: the grid (tiles x splits) should be a multiple of the 256 CUs and give each CU >= 2 resident workgroups
// (measured on the decoder layers: 512 workgroups 96-105 TFLOP/s, 640 workgroups 84-88, 256 workgroups 90-99).  More
// splits also cost one more [Co][Kt] slab to write and reduce, ~114 positions' worth of GEMM time each.
extern "C" int gode_wgrad_auto_splits(const gode_conv_geom* g) {
  const int taps = g->kd * g->kh * g->kw, Kt = taps * g->Ci;
  const int64_t M = (int64_t)g->N * g->Do * g->Ho * g->Wo;
  const int wt = wg_tile(*g);
  const bool w8 = wt >= 3;          // 8-wave tiles: one workgroup per CU (96 KB of LDS), its waves cover each other's barriers
  const int64_t tiles = (int64_t)gode_ceil_div(g->Co, wg_bi(wt)) * gode_ceil_div(Kt, wg_bj(wt));
  int64_t cap = (M + 63) / 64;
  if (cap > 256) cap = 256;
  if (cap < 1) cap = 1;
  int best = 1; double best_cost = 1e300;
  for (int64_t s = 1; s <= cap; ++s) {
    const double blocks = (double)(tiles * s);
    const double rounds = (double)((tiles * s + 255) / 256);
    const double balance = blocks / (rounds * 256.0);                 // busiest CU's share vs the mean
    const double overlap = (blocks >= 512.0 || w8) ? 1.0 : 0.92;      // a lone 4-wave workgroup per CU cannot hide its barriers
    const double cost = (1.0 + 114.0 * (double)s / (double)M) / (balance * overlap);
    if (cost < best_cost - 1e-12) { best_cost = cost; best = (int)s; }
  }
  return best;
}

static bool wg_thin(const gode_conv_geom& g) {
  const int Kt = g.kd * g.kh * g.kw * g.Ci;
  return Kt <= THIN_MAXKT && g.Co % 4 == 0 && (256 % ((g.Co / 4) < 64 ? (g.Co / 4) : 64)) == 0;
}
static int wg_thin_blocks(const gode_conv_geom& g) {
  const int64_t M = (int64_t)g.N * g.Do * g.Ho * g.Wo;
  int64_t b = (M + 63) / 64;       // ~4 positions per thread (the per-position gather chain is serial), up to 1024 workgroups
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
// PATCH path eligibility and launch shape (a function of the op's geometry and transform only, so that the work-size
// query and the launch agree)
static bool wg_patch(const gode_wgrad_op* op, WpArgs* P) {
  const gode_conv_geom& g = op->g;
  const int Kt = g.kd * g.kh * g.kw * g.Ci;
  // (Kt <= 16 stays on the THIN path: measured on the MNIST first layers, 33 / 36 us there against 44 / 84 us here)
  if (op->splits > 0 || g.Ci > 4 || Kt <= THIN_MAXKT || (g.Co != 32 && g.Co != 64 && g.Co != 128) || g.Wo < 16) return false;
  if (op->scale && !op->xform_on_y) return false;
  const int RB = g.Co / 32, CB = (Kt + 31) / 32;
  if ((CB + 4 / RB - 1) / (4 / RB) > WP_MAXB) return false;
  const int PR = g.kd * g.kh, LW = 31 * g.sw + g.kw;
  if (PR > 255 || LW > 32767 || PR * LW * g.Ci > 256 * WP_MAXE) return false;
  int PL = LW * g.Ci;
  while (PL % 32 != (g.kw * g.Ci) % 32) ++PL;
  if (2 * (PR * PL + 32 * g.Co + 4) * 4 > 64 * 1024 || 256 % (g.Co / 4) != 0 || PR > 30) return false;
  // the kernel packs tap-validity masks into one word: depth taps in bits 0..4, row taps from bit 5 (e_key's 5-bit fields)
  if (g.kd > 5 || g.kh > 27) return false;
  const int nseg = (g.Wo + 31) / 32;
  const int64_t segs = (int64_t)g.N * g.Do * g.Ho * nseg;
  if (segs >= (1ll << 31)) return false;
  if (P) {
    P->RB = RB; P->CB = CB; P->PR = PR; P->PL = PL; P->LW = LW; P->nseg = nseg; P->dseg = make_fastdiv((uint32_t)nseg); P->segs = (int)segs;
    int wgs = (int)(segs / 2 < 1 ? 1 : segs / 2);            // >= 2 segments per workgroup, two workgroups per CU at most
    if (wgs > 512) wgs = 512;
    P->segs_per_wg = (int)((segs + wgs - 1) / wgs);
  }
  return true;
}
static int wg_patch_blocks(const gode_wgrad_op* op) {
  WpArgs P;
  if (!wg_patch(op, &P)) return 0;
  return (P.segs + P.segs_per_wg - 1) / P.segs_per_wg;
}

static int wg_splits(const gode_wgrad_op* op) {
  if (const int pb = wg_patch_blocks(op)) return pb;
  if (op->splits <= 0 && wg_thin(op->g)) return wg_thin_blocks(op->g);
  return op->splits > 0 ? op->splits : gode_wgrad_auto_splits(&op->g);
}

extern "C" int64_t gode_wgrad_work_size(const gode_wgrad_op* op) {
  const int taps = op->g.kd * op->g.kh * op->g.kw;
  return (int64_t)wg_splits(op) * op->g.Co * taps * op->g.Ci;
}

static bool wg_generic_forced() {
  static const bool f = getenv("GODE_WGRAD_GENERIC") != nullptr;     // read once per process
  return f;
}
template <int WM, int WN, int TM, int TN>
static int wg_launch(const WgradArgs& A, bool vx, bool vy, int splits, hipStream_t st) {
  constexpr int BI = WM * TM * 32, BJ = WN * TN * 32;
  dim3 grid(gode_ceil_div(A.Kt, BJ), gode_ceil_div(A.g.Co, BI), splits), block(WM * WN * 64);
  static const char* genv = getenv("GODE_WGRAD_GLDS");
  const bool glds = (genv ? atoi(genv) != 0 : true) && A.scale == nullptr && A.act == GODE_ACT_NONE && A.x_bytes != 0;
  if (vx && vy && !wg_generic_forced()) {
    if (glds) hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, true>), grid, block, 0, st, A);
    else hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, false>), grid, block, 0, st, A);
  }
  else if (vx && vy) hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, true, true>), grid, block, 0, st, A);
  else if (vx) hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, true, false>), grid, block, 0, st, A);
  else if (vy) hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, false, true>), grid, block, 0, st, A);
  else hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, false, false>), grid, block, 0, st, A);
  GODE_LAUNCH_CHECK();
  return 0;
}

extern "C" int gode_wgrad(const gode_wgrad_op* op, void* stream) {
  if (!op || !op->x || !op->y || !op->work || !op->dw) return GODE_E_ARG;
  const gode_conv_geom& g = op->g;
  IgemmGeom chk;
  int rc = gode_build_igemm_geom(g, GODE_FPROP, &chk);  // validates the conv relation
  if (rc) return rc;
  WgradArgs A;
  A.g = g; A.x = op->x; A.y = op->y; A.scale = op->scale; A.shift = op->shift; A.work = op->work;
  int64_t xs[5];
  if (gode_strides_are_channels_last(op->xs)) {
    xs[4] = 1; xs[3] = g.Ci; xs[2] = (int64_t)g.Wi * xs[3]; xs[1] = (int64_t)g.Hi * xs[2]; xs[0] = (int64_t)g.Di * xs[1];
  } else {
    for (int i = 0; i < 5; ++i) xs[i] = op->xs[i];
  }
  int64_t span = 1 + (int64_t)(g.N - 1) * xs[0] + (int64_t)(g.Di - 1) * xs[1] + (int64_t)(g.Hi - 1) * xs[2] +
                 (int64_t)(g.Wi - 1) * xs[3] + (int64_t)(g.Ci - 1) * xs[4];
  const int64_t M = (int64_t)g.N * g.Do * g.Ho * g.Wo;
  if (span >= (1ll << 31) || M * g.Co >= (1ll << 31)) return GODE_E_SHAPE;
  A.xsN = (int)xs[0]; A.xsD = (int)xs[1]; A.xsH = (int)xs[2]; A.xsW = (int)xs[3]; A.xsC = (int)xs[4];
  A.act = op->act; A.xform_on_y = op->xform_on_y;
  A.x_bytes = (span * 4 < (1ll << 31) && M * g.Co * 4 < (1ll << 31)) ? (uint32_t)(span * 4) : 0u;
  A.M = (int)M; A.taps = g.kd * g.kh * g.kw; A.Kt = A.taps * g.Ci;
  A.dWo = make_fastdiv(g.Wo); A.dHo = make_fastdiv(g.Ho); A.dDo = make_fastdiv(g.Do);
  const int splits = wg_splits(op);
  A.chunk = (int)(((M + splits - 1) / splits + 31) / 32 * 32);
  const bool vx = xs[4] == 1 && g.Ci % 4 == 0 && xs[0] % 4 == 0 && xs[1] % 4 == 0 && xs[2] % 4 == 0 && xs[3] % 4 == 0 &&
                  (uintptr_t)op->x % 16 == 0;
  const bool vy = g.Co % 4 == 0 && (uintptr_t)op->y % 16 == 0;
  hipStream_t st = (hipStream_t)stream;
  const int t = wg_tile(g);
  WpArgs WP;
  const bool patch = wg_patch(op, &WP);
  if (patch) {
    WP.w = A;
    WP.yvec = (uintptr_t)op->y % 16 == 0;
    const size_t lds = (size_t)2 * (WP.PR * WP.PL + 32 * g.Co + 4) * sizeof(float);
    const int nbw = (WP.CB + 4 / WP.RB - 1) / (4 / WP.RB);      // column blocks per wave
    void (*kern)(const WpArgs) = nullptr;
#define WP_PICK(N, C) if (nbw == N && g.Co == C) kern = wgrad_patch_kernel<N, C>;
    WP_PICK(1, 32) WP_PICK(2, 32) WP_PICK(3, 32) WP_PICK(4, 32)
    WP_PICK(1, 64) WP_PICK(2, 64) WP_PICK(3, 64) WP_PICK(4, 64)
    WP_PICK(1, 128) WP_PICK(2, 128) WP_PICK(3, 128) WP_PICK(4, 128)
#undef WP_PICK
    if (!kern) return GODE_E_SHAPE;
    hipLaunchKernelGGL(kern, dim3(splits), dim3(256), lds, st, WP);
    GODE_LAUNCH_CHECK();
    rc = 0;
  } else if (op->splits <= 0 && wg_thin(g) && vy) {
    const int chunk_thin = (int)((M + splits - 1) / splits);
    hipLaunchKernelGGL(wgrad_thin_kernel, dim3(splits), dim3(256), 0, st, A, chunk_thin);
    GODE_LAUNCH_CHECK();
    rc = 0;
  } else if (t == 0) rc = wg_launch<1, 4, 1, 1>(A, vx, vy, splits, st);
  else if (t == 1) rc = wg_launch<2, 2, 1, 2>(A, vx, vy, splits, st);
  else if (t >= 3 && vx && vy && !wg_generic_forced()) {
    dim3 grid(gode_ceil_div(A.Kt, wg_bj(t)), gode_ceil_div(g.Co, wg_bi(t)), splits), block(512);
    const bool glds = A.scale == nullptr && A.act == GODE_ACT_NONE && A.x_bytes != 0;
    if (t == 3) {
      if (glds) hipLaunchKernelGGL((wgrad_fast_kernel<4, 2, 2, 2, true>), grid, block, 0, st, A);
      else hipLaunchKernelGGL((wgrad_fast_kernel<4, 2, 2, 2, false>), grid, block, 0, st, A);
    } else {
      if (glds) hipLaunchKernelGGL((wgrad_fast_kernel<2, 4, 2, 2, true>), grid, block, 0, st, A);
      else hipLaunchKernelGGL((wgrad_fast_kernel<2, 4, 2, 2, false>), grid, block, 0, st, A);
    }
    GODE_LAUNCH_CHECK();
    rc = 0;
  }
  else rc = wg_launch<2, 2, 2, 2>(A, vx, vy, splits, st);
  if (rc) return rc;
  const int64_t total = (int64_t)g.Co * A.Kt;
  if ((total <= 8192 || (patch && total <= 32768)) && splits >= 16) {   // (patch path: hundreds of slabs of a [Co][Kt] of a few thousand)
    hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((int)((total + RS_OUT - 1) / RS_OUT)), dim3(256), 0, st, op->work, op->dw, g.Co,
                       g.Ci, A.taps, splits, op->co_perm, op->accumulate);
    GODE_LAUNCH_CHECK();
    return 0;
  }
  const bool v4 = A.Kt % 4 == 0 && (uintptr_t)op->work % 16 == 0;
  int blocks = (int)((total / (v4 ? 4 : 1) + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  if (v4) hipLaunchKernelGGL((wgrad_reduce_kernel<true>), dim3(blocks), dim3(256), 0, st, op->work, op->dw, g.Co, g.Ci,
                             A.taps, splits, op->co_perm, op->accumulate);
  else hipLaunchKernelGGL((wgrad_reduce_kernel<false>), dim3(blocks), dim3(256), 0, st, op->work, op->dw, g.Co, g.Ci,
                          A.taps, splits, op->co_perm, op->accumulate);
  GODE_LAUNCH_CHECK();
  return 0;
}
