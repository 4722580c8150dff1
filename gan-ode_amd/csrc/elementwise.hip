// elementwise.hip -- the HBM-bound companions of the GEMM kernels: BatchNorm statistics finalisation and backward,
// BCE-with-logits, Adam(L2).  All reductions are fixed-order (partials -> one reducer), so results are
// run-to-run reproducible.  Reference semantics: nn.BatchNorm2d/3d train mode (models/mocogan.py:76-85,143-156,
// 202-212), nn.BCEWithLogitsLoss (mnist_moco_ode.py:89), torch.optim.Adam(weight_decay) (mnist_moco_ode.py:86-88).
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm finalize
__global__ void __launch_bounds__(256) bn_finalize_kernel(const gode_bn_finalize_op a) {
  const int c = blockIdx.x, tid = threadIdx.x;
  __shared__ double red[2][256];
  if (a.training) {
    // stats[which][column][row] (column = rep*C + c): consecutive threads read consecutive rows of one column.
    // groups == 2: two BatchNorm batches in one launch.  Their partial rows are described by up to 8 segments
    // (begin, split, end): rows [begin, split) of a segment belong to group 0, [split, end) to group 1 -- one segment with
    // begin = 0, split = rows0, end = rows for a grouped FPROP pass (gode_igemm groups == 2), one per stride phase for a
    // transposed-convolution stack whose rows are [first batch; second batch] (gode_igemm_stats_segments).  The groups are
    // finalised one after the other in `order` (0: group 0 first), so the running statistics see the two momentum updates
    // in the order of the reference's two forward calls.
    const int reps = a.ncols / a.C;
    const int ngroups = a.groups == 2 ? 2 : 1;
    for (int gi = 0; gi < ngroups; ++gi) {
      const int grp = ngroups == 2 ? (a.order ? 1 - gi : gi) : 0;
      double s1 = 0.0, s2 = 0.0;
      const int nseg = (ngroups == 2 && !a.stats1) ? (a.nseg > 0 ? a.nseg : 1) : 1;
      for (int sg = 0; sg < nseg; ++sg) {
        int r_lo, r_hi;
        if (ngroups == 1 || a.stats1) { r_lo = 0; r_hi = (grp == 1 && a.stats1) ? a.rows1 : a.rows; }
        else if (a.nseg > 0) { r_lo = grp == 0 ? a.seg[3 * sg] : a.seg[3 * sg + 1]; r_hi = grp == 0 ? a.seg[3 * sg + 1] : a.seg[3 * sg + 2]; }
        else { r_lo = grp == 0 ? 0 : a.rows0; r_hi = grp == 0 ? a.rows0 : a.rows; }
        // (stats1: the second group's partial sums are an array of their own, written by a launch of its own)
        const float* st = (grp == 1 && a.stats1) ? a.stats1 : a.stats;
        const int stride = (grp == 1 && a.stats1) ? a.rows1 : a.rows;
        const int nr = r_hi - r_lo;
        const int64_t items = (int64_t)nr * reps;
        for (int64_t i = tid; i < items; i += 256) {
          const int rep = (int)(i / nr); const int r = r_lo + (int)(i - (int64_t)rep * nr);
          const int64_t o = (int64_t)(rep * a.C + c) * stride + r;
          s1 += (double)st[o];
          s2 += (double)st[(int64_t)a.ncols * stride + o];
        }
      }
      __syncthreads();
      red[0][tid] = s1; red[1][tid] = s2;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
      }
      if (tid == 0) {
        const double n = (double)((grp == 1 && a.count1 > 0) ? a.count1 : a.count);
        const double mean = red[0][0] / n;
        double var = red[1][0] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
        const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
        const float sc = g * invstd;
        const int o = grp * a.C + c;
        a.mean[o] = (float)mean; a.invstd[o] = invstd;
        a.scale[o] = sc; a.shift[o] = b - (float)mean * sc;
        if (a.running_mean) {
          const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
          a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
          a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
        }
        if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
      }
    }
  } else if (tid == 0) {
    const float invstd = 1.f / sqrtf(a.running_var[c] + a.eps);
    const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
    const float sc = g * invstd;
    const int ngroups = a.groups == 2 ? 2 : 1;
    for (int grp = 0; grp < ngroups; ++grp) {      // eval mode: both groups normalise with the running statistics
      const int o = grp * a.C + c;
      if (a.mean) a.mean[o] = a.running_mean[c];
      if (a.invstd) a.invstd[o] = invstd;
      a.scale[o] = sc; a.shift[o] = b - a.running_mean[c] * sc;
    }
  }
}

extern "C" int gode_bn_finalize(const gode_bn_finalize_op* op, void* stream) {
  if (!op || op->C <= 0 || !op->scale || !op->shift) return GODE_E_ARG;
  if (op->training) {
    if (!op->stats || !op->mean || !op->invstd || op->ncols % op->C != 0 || op->count <= 0) return GODE_E_ARG;
    if (op->stats1 && (op->groups != 2 || op->rows1 <= 0 || op->rows <= 0)) return GODE_E_ARG;
    if (op->groups == 2 && !op->stats1 && op->nseg == 0 && (op->rows0 <= 0 || op->rows0 >= op->rows)) return GODE_E_ARG;
    if (op->groups == 2 && (op->nseg < 0 || op->nseg > 8)) return GODE_E_ARG;
    for (int i = 0; op->groups == 2 && i < op->nseg; ++i)
      if (!(0 <= op->seg[3 * i] && op->seg[3 * i] <= op->seg[3 * i + 1] && op->seg[3 * i + 1] <= op->seg[3 * i + 2] && op->seg[3 * i + 2] <= op->rows)) return GODE_E_ARG;
  } else if (!op->running_mean || !op->running_var) return GODE_E_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(op->C), dim3(256), 0, (hipStream_t)stream, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const gode_bn_apply_op a) {
  const int64_t n4 = a.M * a.C / 4;
  const float neg = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % a.C) + ((a.M0 > 0 && i * 4 >= a.M0 * a.C) ? a.C : 0);   // second image group: + C
    f32x4 v = *reinterpret_cast<const f32x4*>(a.y + i * 4);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.scale) { sc = *reinterpret_cast<const f32x4*>(a.scale + c); sh = *reinterpret_cast<const f32x4*>(a.shift + c); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float t = v[e] * sc[e] + sh[e]; v[e] = fmaxf(t, t * neg); }
    *reinterpret_cast<f32x4*>(a.out + i * 4) = v;
  }
}

extern "C" int gode_bn_apply(const gode_bn_apply_op* op, void* stream) {
  if (!op || !op->y || !op->out || op->M <= 0 || op->C <= 0 || op->C % 4 != 0) return GODE_E_ARG;
  if ((op->scale == nullptr) != (op->shift == nullptr)) return GODE_E_ARG;
  int64_t b = (op->M * op->C / 4 + 255) / 256;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((int)(b > 8192 ? 8192 : (b < 1 ? 1 : b))), dim3(256), 0, (hipStream_t)stream, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// col2im of a thin-output transposed convolution (see gode_col2im_op): one thread per output pixel, <= ceil(kh/sh) *
// ceil(kw/sw) taps of C floats each; every element of cols is read exactly once per launch.
__global__ void __launch_bounds__(256) col2im_kernel(const gode_col2im_op a) {
  const int64_t total = (int64_t)a.N * a.Ho * a.Wo;
  const int KC = a.kh * a.kw * a.C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ow = (int)(i % a.Wo); const int64_t t = i / a.Wo;
    const int oh = (int)(t % a.Ho), n = (int)(t / a.Ho);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kh = (oh + a.ph) % a.sh; kh < a.kh; kh += a.sh) {
      const int ih = (oh + a.ph - kh) / a.sh;
      if (oh + a.ph - kh < 0 || ih >= a.Hi) continue;
      for (int kw = (ow + a.pw) % a.sw; kw < a.kw; kw += a.sw) {
        const int iw = (ow + a.pw - kw) / a.sw;
        if (ow + a.pw - kw < 0 || iw >= a.Wi) continue;
        const float* src = a.cols + ((int64_t)(n * a.Hi + ih) * a.Wi + iw) * KC + (kh * a.kw + kw) * a.C;
        for (int c = 0; c < a.C; ++c) acc[c] += src[c];
      }
    }
    float* dst = a.out + i * a.C;
    for (int c = 0; c < a.C; ++c) dst[c] = a.epilogue == GODE_EPI_TANH ? tanhf(acc[c]) : acc[c];
  }
}

// Row-group form: a workgroup owns the sh output rows q*sh - ph ... q*sh - ph + sh - 1 of one image; they draw on the
// R = ceil(kh / sh) input rows q, q-1, ... only, which are contiguous in cols ([Wi][kh*kw*C] each) and go to LDS with
// coalesced float4 loads (the per-pixel form above reads 12-byte pieces 192 B apart: 43 us for the UCF head's 50 MB;
// this one ~20).  Same tap order per output element, so the same sums bit for bit.
__global__ void __launch_bounds__(256) col2im_rows_kernel(const gode_col2im_op a, int R, int nq) {
  extern __shared__ __attribute__((aligned(16))) float c2i_lds[];
  const int KC = a.kh * a.kw * a.C, rowlen = a.Wi * KC;
  const int q = blockIdx.x % nq, n = blockIdx.x / nq;
  for (int j = 0; j < R; ++j) {
    const int ih = q - j;
    if (ih < 0 || ih >= a.Hi) continue;                         // (never read below)
    const f32x4* src = reinterpret_cast<const f32x4*>(a.cols + ((int64_t)(n * a.Hi + ih) * a.Wi) * KC);
    f32x4* dst = reinterpret_cast<f32x4*>(c2i_lds + j * rowlen);
    for (int i = threadIdx.x; i < rowlen / 4; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < a.sh * a.Wo; o += 256) {
    const int r = o / a.Wo, ow = o - r * a.Wo;
    const int oh = q * a.sh - a.ph + r;
    if (oh < 0 || oh >= a.Ho) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kh = (oh + a.ph) % a.sh; kh < a.kh; kh += a.sh) {   // ascending kh = ascending j: the per-pixel form's order
      const int ih = (oh + a.ph - kh) / a.sh, j = q - ih;
      if (ih < 0 || ih >= a.Hi) continue;
      for (int kw = (ow + a.pw) % a.sw; kw < a.kw; kw += a.sw) {
        const int iw = (ow + a.pw - kw) / a.sw;
        if (ow + a.pw - kw < 0 || iw >= a.Wi) continue;
        const float* src = c2i_lds + j * rowlen + iw * KC + (kh * a.kw + kw) * a.C;
        for (int c = 0; c < a.C; ++c) acc[c] += src[c];
      }
    }
    float* dst = a.out + (((int64_t)n * a.Ho + oh) * a.Wo + ow) * a.C;
    for (int c = 0; c < a.C; ++c) dst[c] = a.epilogue == GODE_EPI_TANH ? tanhf(acc[c]) : acc[c];
  }
}

// 3-D form, taps and strides known at compile time (the shape that matters: 4x4x4 taps, strides 1x2x2 -- the UCF video
// discriminator's first layer): no divisions by run-time values, the <= 16 loads of a voxel are issued together.  Same tap
// order (kd, kh, kw ascending) as the generic kernel below: bit-identical sums.
template <int KD, int KH, int KW, int SD, int SH, int SW, int CC>
__global__ void __launch_bounds__(256) col2im3d_fixed_kernel(const gode_col2im_op a) {
  const uint32_t total = (uint32_t)a.N * a.Do * a.Ho * a.Wo;           // (< 2^31: checked by the launcher)
  constexpr int KC = KD * KH * KW * CC;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t ow = i % (uint32_t)a.Wo; uint32_t t = i / (uint32_t)a.Wo;
    const uint32_t oh = t % (uint32_t)a.Ho; t /= (uint32_t)a.Ho;
    const uint32_t od = t % (uint32_t)a.Do, n = t / (uint32_t)a.Do;
    float acc[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc[c] = 0.f;
    const int d0 = ((int)od + a.pd) % SD, h0 = ((int)oh + a.ph) % SH, w0 = ((int)ow + a.pw) % SW;
#pragma unroll
    for (int jd = 0; jd < (KD + SD - 1) / SD; ++jd) {
      const int kd = d0 + jd * SD, td = (int)od + a.pd - kd, id = td / SD;
      if (kd >= KD || td < 0 || id >= a.Di) continue;
#pragma unroll
      for (int jh = 0; jh < (KH + SH - 1) / SH; ++jh) {
        const int kh = h0 + jh * SH, th = (int)oh + a.ph - kh, ih = th / SH;
        if (kh >= KH || th < 0 || ih >= a.Hi) continue;
#pragma unroll
        for (int jw = 0; jw < (KW + SW - 1) / SW; ++jw) {
          const int kw = w0 + jw * SW, tw = (int)ow + a.pw - kw, iw = tw / SW;
          if (kw >= KW || tw < 0 || iw >= a.Wi) continue;
          const float* src = a.cols + ((((int64_t)n * a.Di + id) * a.Hi + ih) * a.Wi + iw) * KC + ((kd * KH + kh) * KW + kw) * CC;
#pragma unroll
          for (int c = 0; c < CC; ++c) acc[c] += src[c];
        }
      }
    }
    float* dst = a.out + (int64_t)i * CC;
#pragma unroll
    for (int c = 0; c < CC; ++c) dst[c] = a.epilogue == GODE_EPI_TANH ? tanhf(acc[c]) : acc[c];
  }
}

// 3-D form: one thread per output voxel, <= ceil(kd/sd) * ceil(kh/sh) * ceil(kw/sw) taps of C floats each, fixed tap order
__global__ void __launch_bounds__(256) col2im3d_kernel(const gode_col2im_op a) {
  const int64_t total = (int64_t)a.N * a.Do * a.Ho * a.Wo;
  const int KC = a.kd * a.kh * a.kw * a.C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ow = (int)(i % a.Wo); int64_t t = i / a.Wo;
    const int oh = (int)(t % a.Ho); t /= a.Ho;
    const int od = (int)(t % a.Do), n = (int)(t / a.Do);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kd = (od + a.pd) % a.sd; kd < a.kd; kd += a.sd) {
      const int id = (od + a.pd - kd) / a.sd;
      if (od + a.pd - kd < 0 || id >= a.Di) continue;
      for (int kh = (oh + a.ph) % a.sh; kh < a.kh; kh += a.sh) {
        const int ih = (oh + a.ph - kh) / a.sh;
        if (oh + a.ph - kh < 0 || ih >= a.Hi) continue;
        for (int kw = (ow + a.pw) % a.sw; kw < a.kw; kw += a.sw) {
          const int iw = (ow + a.pw - kw) / a.sw;
          if (ow + a.pw - kw < 0 || iw >= a.Wi) continue;
          const float* src = a.cols + ((((int64_t)n * a.Di + id) * a.Hi + ih) * a.Wi + iw) * KC + ((kd * a.kh + kh) * a.kw + kw) * a.C;
          for (int c = 0; c < a.C; ++c) acc[c] += src[c];
        }
      }
    }
    float* dst = a.out + i * a.C;
    for (int c = 0; c < a.C; ++c) dst[c] = a.epilogue == GODE_EPI_TANH ? tanhf(acc[c]) : acc[c];
  }
}

extern "C" int gode_col2im(const gode_col2im_op* op, void* stream) {
  if (!op || !op->cols || !op->out || op->N <= 0 || op->C <= 0 || op->C > 4 || op->kh <= 0 || op->kw <= 0 || op->sh <= 0 ||
      op->sw <= 0 || op->ph < 0 || op->pw < 0 || op->Hi <= 0 || op->Wi <= 0)
    return GODE_E_ARG;
  if (op->Ho != (op->Hi - 1) * op->sh - 2 * op->ph + op->kh || op->Wo != (op->Wi - 1) * op->sw - 2 * op->pw + op->kw) return GODE_E_SHAPE;
  if (op->epilogue != GODE_EPI_RAW && op->epilogue != GODE_EPI_TANH) return GODE_E_ARG;
  if (op->kd > 0) {
    if (op->sd <= 0 || op->pd < 0 || op->Di <= 0 || op->Do != (op->Di - 1) * op->sd - 2 * op->pd + op->kd) return GODE_E_SHAPE;
    const int64_t voxels = (int64_t)op->N * op->Do * op->Ho * op->Wo;
    int64_t nb = (voxels + 255) / 256; if (nb > 16384) nb = 16384;
    if (op->kd == 4 && op->kh == 4 && op->kw == 4 && op->sd == 1 && op->sh == 2 && op->sw == 2 && op->C == 3 && voxels < (1ll << 31)) {
      hipLaunchKernelGGL((col2im3d_fixed_kernel<4, 4, 4, 1, 2, 2, 3>), dim3((int)nb), dim3(256), 0, (hipStream_t)stream, *op);
      GODE_LAUNCH_CHECK();
      return 0;
    }
    hipLaunchKernelGGL(col2im3d_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, *op);
    GODE_LAUNCH_CHECK();
    return 0;
  }
  const int KC = op->kh * op->kw * op->C, R = (op->kh + op->sh - 1) / op->sh;
  const int nq = (op->Ho - 1 + op->ph) / op->sh + 1;             // row groups that contain an output row
  const int64_t lds = (int64_t)R * op->Wi * KC * sizeof(float);
  if ((op->Wi * KC) % 4 == 0 && ((uintptr_t)op->cols % 16) == 0 && lds <= 48 * 1024 && (int64_t)op->N * nq < (1ll << 31)) {
    hipLaunchKernelGGL(col2im_rows_kernel, dim3(op->N * nq), dim3(256), (size_t)lds, (hipStream_t)stream, *op, R, nq);
    GODE_LAUNCH_CHECK();
    return 0;
  }
  int64_t nb = ((int64_t)op->N * op->Ho * op->Wo + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(col2im_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm + activation backward
#define BNB_ROWS 256   // rows per reduction block
#define BNB_CL 16      // float4 channel lanes per block (64 channels)

__device__ __forceinline__ float gz_of(float g, float y, float sc, float sh, int act) {
  if (act == GODE_ACT_TANH_OUT) return g * (1.f - y * y);
  return g * gode_act_grad(y * sc + sh, act);
}

// rank-1 upstream gradient (gode_bn_bwd_op.r1_s): the scalar of row r = (image, h, w), zero outside the crop
__device__ __forceinline__ float r1_scalar(const gode_bn_bwd_op& a, uint32_t r) {       // (M < 2^31: 32-bit divisions)
  const uint32_t t = r / (uint32_t)a.r1_W, w = r - t * (uint32_t)a.r1_W;
  const uint32_t n = t / (uint32_t)a.r1_H, h = t - n * (uint32_t)a.r1_H;
  const int hh = (int)h - a.r1_off, ww = (int)w - a.r1_off;
  return ((unsigned)hh < (unsigned)a.r1_h && (unsigned)ww < (unsigned)a.r1_wd) ? a.r1_s[((int64_t)n * a.r1_h + hh) * a.r1_wd + ww] : 0.f;
}

// partial[blk][2][C] (DOUBLE): sum g_z, sum g_z * xhat   (C % 4 == 0, C/4 <= 256).
// Both sums cancel heavily (signed gradients), and their error is fed back into every element of g_y, so they are
// accumulated in fp64 end to end; the kernel stays HBM-bound (2 fp64 FMAs per 8 bytes read).
// groups == 2 (one discriminator pass over [real; fake], gode_bn_bwd_op.groups): rows [0, M/2) and [M/2, M) are two
// BatchNorm batches with their own statistics ([2][C] arrays, group-major) -- row chunks never straddle the halves
// (chunks [0, rows0) belong to group 0), the finalisation forms one coefficient set per group, and both groups' sums go
// into dgamma / dbeta in group order: the arithmetic of two separate backward passes in one launch triple.
struct BnbGroups { int groups, rows0, rows; int64_t M0; };
__device__ __forceinline__ BnbGroups bnb_groups(const gode_bn_bwd_op& a) {
  BnbGroups G;
  G.groups = a.groups == 2 ? 2 : 1;
  G.M0 = G.groups == 2 ? (a.M0 > 0 ? a.M0 : a.M / 2) : a.M;          // rows [0, M0): group 0, [M0, M): group 1
  G.rows0 = (int)((G.M0 + BNB_ROWS - 1) / BNB_ROWS);
  G.rows = G.groups == 2 ? G.rows0 + (int)((a.M - G.M0 + BNB_ROWS - 1) / BNB_ROWS) : G.rows0;
  return G;
}

__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const gode_bn_bwd_op a) {
  // block = (row chunk blockIdx.x) x (channel group blockIdx.y); threads = CL float4 channel lanes x RL row lanes
  const int C4 = a.C >> 2, tid = threadIdx.x;
  const int CL = C4 < BNB_CL ? C4 : BNB_CL;
  const int rl = 256 / CL;
  const int cl = tid % CL, rlane = tid / CL;
  const int c4 = blockIdx.y * CL + cl;
  const BnbGroups G = bnb_groups(a);
  const int grp = (int)blockIdx.x >= G.rows0 ? 1 : 0;
  __shared__ double red[2][256][4];
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  const bool active = rlane < rl && c4 < C4;
  if (active) {
    const int c = c4 * 4, cs = grp * a.C + c;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + cs), sh = *reinterpret_cast<const f32x4*>(a.shift + cs);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cs), is = *reinterpret_cast<const f32x4*>(a.invstd + cs);
    const int64_t r0 = grp * G.M0 + (int64_t)((int)blockIdx.x - grp * G.rows0) * BNB_ROWS;
    const int64_t rend = grp == 0 ? G.M0 : a.M;
    const int64_t r1 = r0 + BNB_ROWS < rend ? r0 + BNB_ROWS : rend;
    const float* gsrc = a.gin ? a.gin : a.g;
    f32x4 w1 = {0.f, 0.f, 0.f, 0.f};
    if (a.r1_s) w1 = *reinterpret_cast<const f32x4*>(a.r1_w + c);
    for (int64_t r = r0 + rlane; r < r1; r += rl) {
      f32x4 g;
      if (a.r1_s) { const float s = r1_scalar(a, (uint32_t)r); g = w1 * s; }
      else g = *reinterpret_cast<const f32x4*>(gsrc + r * a.C + c);
      const f32x4 y = *reinterpret_cast<const f32x4*>(a.y + r * a.C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gz = gz_of(g[e], y[e], sc[e], sh[e], a.act);
        s1[e] += (double)gz;
        s2[e] += (double)gz * (double)((y[e] - mu[e]) * is[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; }
  __syncthreads();
  if (tid < CL && blockIdx.y * CL + tid < C4) {
    double t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
    for (int j = 0; j < rl; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { t1[e] += red[0][j * CL + tid][e]; t2[e] += red[1][j * CL + tid][e]; }
    double* dst = reinterpret_cast<double*>(a.work) + (int64_t)blockIdx.x * 2 * a.C + (blockIdx.y * CL + tid) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[e] = t1[e]; dst[a.C + e] = t2[e]; }
  }
}

// one block per channel: reduce partials, write dgamma/dbeta and the three apply coefficients (per group)
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const gode_bn_bwd_op a) {
  const int c = blockIdx.x, tid = threadIdx.x;
  __shared__ double red[2][256];
  const double* part = reinterpret_cast<const double*>(a.work);
  const BnbGroups G = bnb_groups(a);
  float* coef = a.work + (int64_t)G.rows * 4 * a.C;
  float dbeta_acc = 0.f, dgamma_acc = 0.f;
  if (tid == 0 && a.accumulate) { dbeta_acc = a.dbeta ? a.dbeta[c] : 0.f; dgamma_acc = a.dgamma ? a.dgamma[c] : 0.f; }
  for (int grp = 0; grp < G.groups; ++grp) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = (grp == 0 ? 0 : G.rows0) + tid; r < (grp == 0 ? G.rows0 : G.rows); r += 256) {
      s1 += part[(int64_t)r * 2 * a.C + c];
      s2 += part[(int64_t)r * 2 * a.C + a.C + c];
    }
    __syncthreads();
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
      __syncthreads();
    }
    if (tid == 0) {
      dbeta_acc += (float)red[0][0]; dgamma_acc += (float)red[1][0];          // group order: what two passes would add
      const double Mg = (double)(grp == 0 ? G.M0 : a.M - (G.groups == 2 ? G.M0 : 0));
      const double invM = 1.0 / Mg;
      const int cs = grp * a.C + c;
      const double g = a.gamma ? (double)a.gamma[c] : 1.0, is = (double)a.invstd[cs];
      const double k = g * is;
      float* cf = coef + (int64_t)grp * 3 * a.C;
      // eval mode (running statistics are constants of the forward): g_y = gamma*invstd * g_z, no batch terms
      cf[c] = (float)k;                                                      // * g_z
      cf[a.C + c] = a.eval_mode ? 0.f : (float)(k * is * red[1][0] * invM);  // * (mean - y)
      cf[2 * a.C + c] = a.eval_mode ? 0.f : (float)(k * red[0][0] * invM);   // subtracted constant
    }
  }
  if (tid == 0) {
    if (a.dbeta) a.dbeta[c] = dbeta_acc;
    if (a.dgamma) a.dgamma[c] = dgamma_acc;
  }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const gode_bn_bwd_op a) {
  const int64_t n4 = a.M * a.C / 4;
  const BnbGroups G = bnb_groups(a);
  const float* coef = a.work + (int64_t)G.rows * 4 * a.C;
  const float* gsrc = a.gin ? a.gin : a.g;
  const int64_t n4_0 = G.M0 * a.C / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int grp = (G.groups == 2 && i >= n4_0) ? 1 : 0;
    const int c = (int)((i * 4) % a.C), cs = grp * a.C + c;
    const float* cf = coef + (int64_t)grp * 3 * a.C;
    f32x4 g;
    if (a.r1_s) g = *reinterpret_cast<const f32x4*>(a.r1_w + c) * r1_scalar(a, (uint32_t)((i * 4) / a.C));
    else g = *reinterpret_cast<const f32x4*>(gsrc + i * 4);
    const f32x4 y = *reinterpret_cast<const f32x4*>(a.y + i * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + cs), sh = *reinterpret_cast<const f32x4*>(a.shift + cs);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + cs);
    const f32x4 cA = *reinterpret_cast<const f32x4*>(cf + c), cB = *reinterpret_cast<const f32x4*>(cf + a.C + c),
                cC = *reinterpret_cast<const f32x4*>(cf + 2 * a.C + c);
    // g_y = k*g_z - k*dbeta/M - k*invstd*dgamma/M*(y - mean); (mean - y) is formed first so that a large |mean|
    // does not cancel against a separately rounded constant
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = cA[e] * gz_of(g[e], y[e], sc[e], sh[e], a.act) + cB[e] * (mu[e] - y[e]) - cC[e];
    *reinterpret_cast<f32x4*>(a.g + i * 4) = g;
  }
}

// no BatchNorm: g *= act'(y) (or tanh-from-output), any C
__global__ void __launch_bounds__(256) act_bwd_kernel(float* g, const float* gin, const float* y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    g[i] = gz_of(gin[i], y[i], 1.f, 0.f, act);
}

extern "C" int64_t gode_bn_bwd_work_size(int64_t M, int32_t C) {
  // (sized for the two-group form: one more chunk row, two coefficient sets)
  const int64_t rows = (M + BNB_ROWS - 1) / BNB_ROWS + 2;
  return rows * 4 * C + 6 * (int64_t)C;   // fp64 partials (2 floats each) + 3 coefficient vectors per group
}

static int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int gode_bn_bwd(const gode_bn_bwd_op* op, void* stream) {
  if (!op || !op->g || !op->y || op->M <= 0 || op->C <= 0) return GODE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!op->mean) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(op->M * op->C)), dim3(256), 0, st, op->g,
                       op->gin ? op->gin : op->g, op->y, op->M * op->C, op->act);
    GODE_LAUNCH_CHECK();
    return 0;
  }
  if (op->C % 4 != 0 || !op->invstd || !op->scale || !op->shift || !op->work) return GODE_E_ARG;
  if (op->r1_s && (!op->r1_w || op->r1_H <= 0 || op->r1_W <= 0 || op->r1_h <= 0 || op->r1_wd <= 0 || op->r1_off < 0 ||
                   op->M % ((int64_t)op->r1_H * op->r1_W) != 0 || op->M >= (1ll << 31) || ((uintptr_t)op->r1_w % 16) != 0)) return GODE_E_ARG;
  if (op->groups == 2 && op->M0 == 0 && (op->M % 2 != 0)) return GODE_E_ARG;
  if (op->groups == 2 && (op->M0 < 0 || op->M0 >= op->M)) return GODE_E_ARG;
  const int C4 = op->C / 4, CL = C4 < BNB_CL ? C4 : BNB_CL;
  if (256 % CL != 0) return GODE_E_SHAPE;
  const int groups = op->groups == 2 ? 2 : 1;
  const int64_t M0 = groups == 2 ? (op->M0 > 0 ? op->M0 : op->M / 2) : op->M;
  const int rows = (int)((M0 + BNB_ROWS - 1) / BNB_ROWS) + (groups == 2 ? (int)((op->M - M0 + BNB_ROWS - 1) / BNB_ROWS) : 0);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(rows, (C4 + CL - 1) / CL), dim3(256), 0, st, *op);
  GODE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(op->C), dim3(256), 0, st, *op);
  GODE_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(op->M * op->C / 4)), dim3(256), 0, st, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// BCE with logits against a constant target, mean reduction (single block: n is a few thousand at most)
__global__ void __launch_bounds__(256) bce_kernel(const gode_bce_op a) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double s = 0.0;
  const float inv = 1.f / (float)a.n;
  for (int64_t i = tid; i < a.n; i += 256) {
    const float x = a.logits[i];
    const float l = fmaxf(x, 0.f) - x * a.target + log1pf(expf(-fabsf(x)));
    s += (double)l;
    if (a.grad) {
      const float sig = 1.f / (1.f + expf(-x));
      a.grad[i] = a.gscale * (sig - a.target) * inv;
    }
  }
  red[tid] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (tid < k) red[tid] += red[tid + k];
    __syncthreads();
  }
  if (tid == 0 && a.loss) {
    const float v = (float)(red[0] / (double)a.n);
    *a.loss = a.accumulate ? *a.loss + v : v;
  }
}

extern "C" int gode_bce_logits(const gode_bce_op* op, void* stream) {
  if (!op || !op->logits || op->n <= 0) return GODE_E_ARG;
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Adam with L2-coupled weight decay (operation order of torch.optim.Adam's single-tensor path)
__global__ void __launch_bounds__(256) adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float b1,
                                                   float b2, float eps, float wd, float gscale, float step_size,
                                                   float bc2_sqrt) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pi = p[i];
    const float gi = g[i] * gscale + wd * pi;
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

extern "C" int gode_adam_l2(const gode_adam_op* op, void* stream) {
  if (!op || !op->p || !op->g || !op->m || !op->v || op->n <= 0 || op->step < 1) return GODE_E_ARG;
  const double bc1 = 1.0 - pow((double)op->beta1, (double)op->step);
  const double bc2 = 1.0 - pow((double)op->beta2, (double)op->step);
  const float step_size = (float)((double)op->lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(op->n)), dim3(256), 0, (hipStream_t)stream, op->p, op->g, op->m,
                     op->v, op->n, op->beta1, op->beta2, op->eps, op->weight_decay, op->gscale, step_size, bc2_sqrt);
  GODE_LAUNCH_CHECK();
  return 0;
}

// One update rule for both entry points; float4 lanes when the four pointers of a tensor are 16-byte aligned (torch
// allocations and arena slices of whole float4s are), scalar tail / fallback otherwise.  HBM-bound: 16 B read + 12 B
// written per parameter.
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float b1, float b2, float eps, float wd,
                                          float gscale, float step_size, float bc2_sqrt) {
  const float gi = g * gscale + wd * p;
  const float mi = m + (gi - m) * (1.f - b1);
  const float vi = v * b2 + (1.f - b2) * gi * gi;
  m = mi; v = vi;
  p = p - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}
__device__ __forceinline__ void adam_tensor(const gode_adam_tensor& t, float b1, float b2, float eps, float wd, float gscale,
                                            float step_size, float bc2_sqrt) {
  const bool vec = (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0;
  const int64_t n4 = vec ? t.n >> 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 p = reinterpret_cast<const f32x4*>(t.p)[i], m = reinterpret_cast<const f32x4*>(t.m)[i], v = reinterpret_cast<const f32x4*>(t.v)[i];
    const f32x4 g = reinterpret_cast<const f32x4*>(t.g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = p[e], me = m[e], ve = v[e];
      adam_elem(pe, g[e], me, ve, b1, b2, eps, wd, gscale, step_size, bc2_sqrt);
      p[e] = pe; m[e] = me; v[e] = ve;
    }
    reinterpret_cast<f32x4*>(t.m)[i] = m; reinterpret_cast<f32x4*>(t.v)[i] = v; reinterpret_cast<f32x4*>(t.p)[i] = p;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (int64_t)gridDim.x * 256) {
    float p = t.p[i], m = t.m[i], v = t.v[i];
    adam_elem(p, t.g[i], m, v, b1, b2, eps, wd, gscale, step_size, bc2_sqrt);
    t.m[i] = m; t.v[i] = v; t.p[i] = p;
  }
}
static int adam_blocks(int64_t max_n) {       // the largest tensor sets the grid: ~2 float4 per thread, others leave early
  int64_t b = (max_n + 2047) / 2048;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

__global__ void __launch_bounds__(256) adam_multi_kernel(const gode_adam_tensor* table, float b1, float b2, float eps,
                                                         float wd, float gscale, float step_size, float bc2_sqrt) {
  adam_tensor(table[blockIdx.y], b1, b2, eps, wd, gscale, step_size, bc2_sqrt);
}

extern "C" int gode_adam_multi(const gode_adam_tensor* table, int32_t count, int64_t max_n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float gscale, int32_t step, void* stream) {
  if (!table || count <= 0 || max_n <= 0 || step < 1) return GODE_E_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const int bx = adam_blocks(max_n);
  hipLaunchKernelGGL(adam_multi_kernel, dim3(bx, count), dim3(256), 0, (hipStream_t)stream, table, beta1, beta2, eps,
                     weight_decay, gscale, (float)((double)lr / bc1), (float)sqrt(bc2));
  GODE_LAUNCH_CHECK();
  return 0;
}

// The same update with the two step-dependent coefficients read from DEVICE memory (coef[0] = lr / (1 - beta1^step),
// coef[1] = sqrt(1 - beta2^step), written by the host in the arithmetic of gode_adam_multi): a launch recorded in a
// HIP graph stays valid while the step count advances -- the host refreshes coef before each replay.
__global__ void __launch_bounds__(256) adam_multi_dev_kernel(const gode_adam_tensor* table, float b1, float b2, float eps,
                                                             float wd, float gscale, const float* coef) {
  adam_tensor(table[blockIdx.y], b1, b2, eps, wd, gscale, coef[0], coef[1]);
}

extern "C" int gode_adam_multi_dev(const gode_adam_tensor* table, int32_t count, int64_t max_n, float beta1, float beta2,
                                   float eps, float weight_decay, float gscale, const float* coef, void* stream) {
  if (!table || count <= 0 || max_n <= 0 || !coef) return GODE_E_ARG;
  const int bx = adam_blocks(max_n);
  hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(bx, count), dim3(256), 0, (hipStream_t)stream, table, beta1, beta2, eps,
                     weight_decay, gscale, coef);
  GODE_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) scale_kernel(float* out, const float* a, int64_t n, float alpha, int acc) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = acc ? out[i] + a[i] * alpha : a[i] * alpha;
}

extern "C" int gode_scale(float* out, const float* a, int64_t n, float alpha, int accumulate, void* stream) {
  if (!out || !a || n <= 0) return GODE_E_ARG;
  hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, out, a, n, alpha, accumulate);
  GODE_LAUNCH_CHECK();
  return 0;
}
