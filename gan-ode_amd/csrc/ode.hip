// ode.hip -- the motion-latent Neural-ODE of MoCoGAN-ODE fused into one launch each way.
//
// Reference arithmetic: models/mocogan_ode.py:6-17 (ODEFunc: Linear(16,16)-Tanh-Linear(16,16)), :123-129 (pre-net
// Linear(16,64)-LReLU-Linear(64,16)-LReLU), :133-148 (odeint_adjoint(..., linspace(0,1,T), method='rk4')), and
// torchdiffeq's fixed-grid rk4 (Kutta 3/8 rule) + continuous-adjoint backward (restated in oracle/ode_ref.py).
// In PyTorch one solve is ~360 dependent micro-kernels; here it is one wave-resident loop.
//
// Mapping: one wave = 16 trajectories.  Every 16x16 mat-vec batch is exactly one v_mfma_f32_16x16x4_f32 tile
// computed TRANSPOSED (D = W * Y^T): lane (s = lane&15, g = lane>>4) ends up holding features 4g..4g+3 of
// trajectory s in its 4 accumulator registers.  Because the MFMA k order is free as long as A and B agree, step r
// of the next product takes k = 4g+r: the accumulator registers of one product are fed straight back as the B
// operand of the next and the weights are pre-loaded in that k order -- no cross-lane traffic in the whole solve.
// Only the parameter-gradient outer products of the adjoint pass (sum over trajectories) need a transpose; it goes
// through a 1.25 KB per-matrix LDS tile and is again an MFMA (k = trajectory).
#include <stdlib.h>
#include "common.h"

#include "ode_common.h"

// ---------------------------------------------------------------------------------------------------------------
// Launched with 256 threads when there is a content code to broadcast: wave 0 runs the (strictly serial) solve of the
// 16 trajectories, waves 1-3 write the content columns of the same latent rows at the same time on the other SIMDs
// (the broadcast used to run ahead of the solve on the one wave: 31 us per launch at N = 32, 21 us for the solve alone).
__global__ void __launch_bounds__(256) ode_fwd_kernel(const gode_ode_fwd_op a) {
  const int n0 = blockIdx.x * 16;
  const int T = a.T;
  if (threadIdx.x >= 64) {
    // content columns 16..65 (the same 50 values on all T rows of a trajectory) + zero pad, one float4 per thread and
    // turn; rows are 200 B in `content`, so the four values are gathered one by one (L1-resident: 3.2 KB per workgroup)
    const int rows_per = a.sel_t ? 1 : T;
    const int q4 = (a.zcols - 16) >> 2;      // float4 chunks of content + pad per latent row
    const int total4 = 16 * rows_per * q4;
    for (int i = threadIdx.x - 64; i < total4; i += 192) {
      const int rr = i / q4, q = i - rr * q4;
      const int ns = rr / rows_per, tt = rr - ns * rows_per;
      if (n0 + ns < a.N) {
        const float* c = a.content + (int64_t)(n0 + ns) * 50 + 4 * q;
        f32x4 v = zero4();
        if (q < 12) v = f32x4{c[0], c[1], c[2], c[3]};
        else if (q == 12) v = f32x4{c[0], c[1], 0.f, 0.f};
        *reinterpret_cast<f32x4*>(a.z + ((int64_t)(n0 + ns) * rows_per + tt) * a.zcols + 16 + 4 * q) = v;
      }
    }
    return;
  }
  const int l = threadIdx.x, s = l & 15, g = l >> 4;
  const int n = n0 + s;
  const bool valid = n < a.N;

  f32x4 y = valid ? ld4(a.x + n * 16 + 4 * g) : zero4();
  if (a.prenet) {
    f32x4 acc = ld4(a.p.bb + 4 * g);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x4 h = matvec(ld4(a.p.Wa + (16 * m + s) * 16 + 4 * g), y, ld4(a.p.ba + 16 * m + 4 * g));
      acc = matvec(ld4(a.p.Wb + s * 64 + 16 * m + 4 * g), lrelu4(h), acc);
    }
    y = lrelu4(acc);
  }
  const f32x4 w1 = ld4(a.p.W1 + s * 16 + 4 * g), w2 = ld4(a.p.W2 + s * 16 + 4 * g);
  const f32x4 b1 = ld4(a.p.b1 + 4 * g), b2 = ld4(a.p.b2 + 4 * g);
  auto f = [&](const f32x4 yy) { return matvec(w2, tanh4(matvec(w1, yy, b1)), b2); };

  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto emit = [&](int t) {
    if (!valid) return;
    if (a.traj) *reinterpret_cast<f32x4*>(a.traj + ((int64_t)n * T + t) * 16 + 4 * g) = y;
    if (a.sel_t == nullptr) *reinterpret_cast<f32x4*>(a.z + ((int64_t)n * T + t) * a.zcols + 4 * g) = y;
    else if (t == tsel) *reinterpret_cast<f32x4*>(a.z + (int64_t)n * a.zcols + 4 * g) = y;
  };
  emit(0);
  const float third = 1.0f / 3.0f;
  if (a.grid_dt != nullptr) {
    // solver grid != output times: integrate step by step, emit every output whose time falls in the step just taken
    int jo = 1;
    for (int i = 0; i < a.G; ++i) {
      const float dt = a.grid_dt[i];
      const f32x4 y0 = y;
      const f32x4 k1 = f(y);
      const f32x4 k2 = f(y + dt * k1 * third);
      const f32x4 k3 = f(y + dt * (k2 - k1 * third));
      const f32x4 k4 = f(y + dt * (k1 - k2 + k3));
      const f32x4 y1 = y0 + (k1 + 3.f * (k2 + k3) + k4) * dt * 0.125f;
      while (jo < T && a.emit_at[jo] == i) {
        const float w = a.emit_w[jo];
        y = w == 1.f ? y1 : (w == 0.f ? y0 : y0 + w * (y1 - y0));
        emit(jo);
        ++jo;
      }
      y = y1;
    }
    return;
  }
  for (int j = 0; j + 1 < T; ++j) {
    const float dt = a.dt[j] / (float)a.substeps;
    for (int ss = 0; ss < a.substeps; ++ss) {
      const f32x4 k1 = f(y);
      const f32x4 k2 = f(y + dt * k1 * third);
      const f32x4 k3 = f(y + dt * (k2 - k1 * third));
      const f32x4 k4 = f(y + dt * (k1 - k2 + k3));
      y = y + (k1 + 3.f * (k2 + k3) + k4) * dt * 0.125f;
    }
    emit(j + 1);
  }

}

int gode_launch_ode_dopri5(const gode_ode_fwd_op* op, hipStream_t st);   // odernn.hip
int gode_launch_ode_fwd_valu(const gode_ode_fwd_op* op, hipStream_t st);  // ode_valu.hip (default mapping)
int gode_launch_ode_bwd_valu(const gode_ode_bwd_op* op, hipStream_t st);
// Two mappings of the same arithmetic (different summation order): the VALU/DPP kernels of ode_valu.hip (4 trajectories
// per wave, short dependent chains: 14 / 33 us forward / adjoint at the config size N = 32 against 21 / 53 us) and the
// MFMA-chain kernels of this file (16 trajectories per wave: 67 vs 43 TFLOP/s forward at N = 2^20).  Default: VALU up
// to 8192 trajectories, MFMA above; GODE_ODE_MFMA=1 / =0 forces one (the A/B of profiles/r02_ode_kernels.txt).
static bool ode_use_mfma(int N) {
  static const char* e = getenv("GODE_ODE_MFMA");
  if (e) return atoi(e) != 0;
  return N > 8192;
}
extern "C" int gode_ode_fwd(const gode_ode_fwd_op* op, void* stream) {
  if (op && op->method == 1) {
    if (!op->x || !op->z || !op->tout || op->N <= 0 || op->T < 1 || !(op->rtol > 0.f) || !(op->atol >= 0.f)) return GODE_E_ARG;
    if (op->zcols < 68 || op->zcols % 4 != 0) return GODE_E_ARG;
    if (!op->p.W1 || !op->p.b1 || !op->p.W2 || !op->p.b2) return GODE_E_ARG;
    if (op->prenet && (!op->p.Wa || !op->p.ba || !op->p.Wb || !op->p.bb)) return GODE_E_ARG;
    return gode_ode_fwd_multi(op, 1, stream);       // odernn_valu.hip (VALU kernels; MFMA fallback inside)
  }
  if (op && op->method != 0) return GODE_E_ARG;
  if (!op || !op->x || !op->z || !op->dt || op->N <= 0 || op->T < 1 || op->substeps < 1) return GODE_E_ARG;
  if (op->grid_dt && (!op->emit_at || !op->emit_w || op->G < 1)) return GODE_E_ARG;
  if (op->zcols < 68 || op->zcols % 4 != 0) return GODE_E_ARG;
  if (!op->p.W1 || !op->p.b1 || !op->p.W2 || !op->p.b2) return GODE_E_ARG;
  if (op->prenet && (!op->p.Wa || !op->p.ba || !op->p.Wb || !op->p.bb)) return GODE_E_ARG;
  if (!ode_use_mfma(op->N)) return gode_launch_ode_fwd_valu(op, (hipStream_t)stream);
  hipLaunchKernelGGL(ode_fwd_kernel, dim3((op->N + 15) / 16), dim3(op->content ? 256 : 64), 0, (hipStream_t)stream, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// adjoint backward
#define OFF_WA 0
#define OFF_BA 1024
#define OFF_WB 1088
#define OFF_BB 2112
#define OFF_W1 2128
#define OFF_B1 2384
#define OFF_W2 2400
#define OFF_B2 2656
__global__ void __launch_bounds__(64) ode_bwd_kernel(const gode_ode_bwd_op a) {
  __shared__ __attribute__((aligned(16))) float tile[4][16 * LDT];
  const int l = threadIdx.x, s = l & 15, g = l >> 4;
  const int n0 = blockIdx.x * 16, n = n0 + s;
  const bool valid = n < a.N;
  const int T = a.T;
  float* part = a.work + (int64_t)blockIdx.x * GODE_ODE_NPARAM;

  const f32x4 w1 = ld4(a.p.W1 + s * 16 + 4 * g), w2 = ld4(a.p.W2 + s * 16 + 4 * g);
  const f32x4 b1 = ld4(a.p.b1 + 4 * g), b2 = ld4(a.p.b2 + 4 * g);
  f32x4 w1t, w2t;  // transposed operands: (W^T)[s][4g+r] = W[4g+r][s]
#pragma unroll
  for (int r = 0; r < 4; ++r) { w1t[r] = a.p.W1[(4 * g + r) * 16 + s]; w2t[r] = a.p.W2[(4 * g + r) * 16 + s]; }

  // dst += sum over trajectories of P[s][i] * Q[s][j]; P, Q given in D layout (lane (s,g) holds cols 4g..4g+3).
  // result in D layout: lane (j = lane&15, g) register r = dst[4g + r][j]
  auto outer = [&](f32x4 dst, const f32x4 P, const f32x4 Q) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(&tile[0][s * LDT + 4 * g]) = P;
    *reinterpret_cast<f32x4*>(&tile[1][s * LDT + 4 * g]) = Q;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) dst = MFMA16(tile[0][(4 * g + r) * LDT + s], tile[1][(4 * g + r) * LDT + s], dst);
    return dst;
  };

  f32x4 gW1 = zero4(), gW2 = zero4(), gb1 = zero4(), gb2 = zero4();
  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto upstream = [&](int t) {
    if (!valid) return zero4();
    if (a.sel_t == nullptr) return ld4(a.gz + ((int64_t)n * T + t) * a.zcols + 4 * g);
    return t == tsel ? ld4(a.gz + (int64_t)n * a.zcols + 4 * g) : zero4();
  };

  f32x4 adj = upstream(T - 1);
  f32x4 ky, ka;
  // one stage of the reversed augmented dynamics at (ys, as); c = RK weight * dt/8 for the parameter integrals
  auto stage = [&](const f32x4 ys, const f32x4 as, float c) {
    const f32x4 h = tanh4(matvec(w1, ys, b1));
    const f32x4 fv = matvec(w2, h, b2);
    const f32x4 v = matvec(w2t, as, zero4());
    const f32x4 du = v * (1.f - h * h);
    ka = matvec(w1t, du, zero4());
    ky = -fv;
    const f32x4 ca = c * as, cdu = c * du;
    gb2 += ca; gb1 += cdu;
    gW2 = outer(gW2, ca, h);
    gW1 = outer(gW1, cdu, ys);
  };

  const float third = 1.0f / 3.0f;
  for (int i = T - 1; i >= 1; --i) {
    f32x4 y = valid ? ld4(a.traj + ((int64_t)n * T + i) * 16 + 4 * g) : zero4();
    const int s0 = a.bstep_off ? a.bstep_off[i - 1] : 0;
    const int ns = a.bstep_off ? a.bstep_off[i] - s0 : a.substeps;
    const float dt_eq = a.dt[i - 1] / (float)a.substeps;
    for (int ss = 0; ss < ns; ++ss) {
      const float dt = a.bstep_off ? a.bstep_dt[s0 + ss] : dt_eq;
      stage(y, adj, dt * 0.125f);
      const f32x4 ky1 = ky, ka1 = ka;
      stage(y + dt * ky1 * third, adj + dt * ka1 * third, 3.f * dt * 0.125f);
      const f32x4 ky2 = ky, ka2 = ka;
      stage(y + dt * (ky2 - ky1 * third), adj + dt * (ka2 - ka1 * third), 3.f * dt * 0.125f);
      const f32x4 ky3 = ky, ka3 = ka;
      stage(y + dt * (ky1 - ky2 + ky3), adj + dt * (ka1 - ka2 + ka3), dt * 0.125f);
      y = y + (ky1 + 3.f * (ky2 + ky3) + ky) * dt * 0.125f;
      adj = adj + (ka1 + 3.f * (ka2 + ka3) + ka) * dt * 0.125f;
    }
    adj = adj + upstream(i - 1);
  }

  // ODEFunc parameter partials
  {
    const f32x4 sb1 = sum_over_samples(gb1), sb2 = sum_over_samples(gb2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      part[OFF_W1 + (4 * g + r) * 16 + s] = gW1[r];
      part[OFF_W2 + (4 * g + r) * 16 + s] = gW2[r];
    }
    if (s == 0) {
      *reinterpret_cast<f32x4*>(part + OFF_B1 + 4 * g) = sb1;
      *reinterpret_cast<f32x4*>(part + OFF_B2 + 4 * g) = sb2;
    }
  }

  // pre-net backward; adj = dL/d(pre-net output)
  if (a.prenet) {
    const f32x4 x = valid ? ld4(a.x + n * 16 + 4 * g) : zero4();
    f32x4 hpre[4];
    f32x4 acc = ld4(a.p.bb + 4 * g);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      hpre[m] = matvec(ld4(a.p.Wa + (16 * m + s) * 16 + 4 * g), x, ld4(a.p.ba + 16 * m + 4 * g));
      acc = matvec(ld4(a.p.Wb + s * 64 + 16 * m + 4 * g), lrelu4(hpre[m]), acc);
    }
    const f32x4 g0 = lrelu_grad4(acc, adj);
    const f32x4 sbb = sum_over_samples(g0);
    if (s == 0) *reinterpret_cast<f32x4*>(part + OFF_BB + 4 * g) = sbb;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 dWb = outer(zero4(), g0, lrelu4(hpre[m]));
      f32x4 wbt;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        part[OFF_WB + (4 * g + r) * 64 + 16 * m + s] = dWb[r];
        wbt[r] = a.p.Wb[(4 * g + r) * 64 + 16 * m + s];
      }
      const f32x4 gh = lrelu_grad4(hpre[m], matvec(wbt, g0, zero4()));
      const f32x4 sba = sum_over_samples(gh);
      if (s == 0) *reinterpret_cast<f32x4*>(part + OFF_BA + 16 * m + 4 * g) = sba;
      const f32x4 dWa = outer(zero4(), gh, x);
#pragma unroll
      for (int r = 0; r < 4; ++r) part[OFF_WA + (16 * m + 4 * g + r) * 16 + s] = dWa[r];
    }
  } else {
    for (int i = l; i < OFF_W1; i += 64) part[i] = 0.f;
  }
}

// `first`: entries below it are not touched (no pre-net: only the ODEFunc block [OFF_W1, NPARAM) exists, and `grads`
// may then point 2128 floats before a caller-owned region that holds just that block).  Rows b = 0, stride, 2*stride, ...
__global__ void __launch_bounds__(256) ode_bwd_reduce_kernel(const float* work, float* grads, int nblk, int stride, int accumulate,
                                                             int first) {
  const int i = first + blockIdx.x * 256 + threadIdx.x;
  if (i >= GODE_ODE_NPARAM) return;
  float s = 0.f;
  for (int b = 0; b < nblk; b += stride) s += work[(int64_t)b * GODE_ODE_NPARAM + i];
  grads[i] = accumulate ? grads[i] + s : s;
}
// Many workgroups (N >> 1024): a first level sums chunks of `chunk` consecutive partial rows in parallel, in place (the
// sum of rows [c*chunk, (c+1)*chunk) replaces row c*chunk; a thread only ever touches its own column of its own chunk),
// the kernel above then adds every chunk-th row.  Fixed order at both levels.  (A single level left 11 workgroups
// walking 65,536 rows each: 25 of the 28 ms of the N = 2^20 adjoint.)
__global__ void __launch_bounds__(256) ode_bwd_reduce_chunks_kernel(float* work, int nblk, int chunk, int first) {
  const int i = first + blockIdx.x * 256 + threadIdx.x;
  if (i >= GODE_ODE_NPARAM) return;
  const int b0 = blockIdx.y * chunk, b1 = b0 + chunk < nblk ? b0 + chunk : nblk;
  float s = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) s += work[(int64_t)b * GODE_ODE_NPARAM + i];
  work[(int64_t)b0 * GODE_ODE_NPARAM + i] = s;
}

extern "C" int64_t gode_ode_bwd_work_size(int32_t N) { return (int64_t)((N + 15) / 16) * GODE_ODE_NPARAM; }

int gode_launch_ode_dopri5_bwd(const gode_ode_bwd_op* op, hipStream_t st);   // adj_adaptive.hip

static int ode_bwd_reduce(const gode_ode_bwd_op* op, int nblk, hipStream_t st);

// the round-2 adaptive adjoint (adj_adaptive.hip, norm per 64-trajectory workgroup) with its reduction: the fallback of
// gode_ode_bwd_multi above the co-residency limit
int gode_launch_ode_dopri5_bwd_mfma(const gode_ode_bwd_op* op, hipStream_t st) {
  const int rc = gode_launch_ode_dopri5_bwd(op, st);
  if (rc) return rc;
  return ode_bwd_reduce(op, (op->N + 15) / 16, st);
}

extern "C" int gode_ode_bwd(const gode_ode_bwd_op* op, void* stream) {
  const bool adaptive = op && op->method == 1 && op->substeps == 0;
  if (!op || !op->traj || !op->gz || !op->work || !op->grads || op->N <= 0 || op->T < 1) return GODE_E_ARG;
  if (adaptive ? (!op->tout || !(op->rtol > 0.f) || !(op->atol >= 0.f)) : (!op->dt || op->substeps < 1)) return GODE_E_ARG;
  if (op->method != 0 && op->method != 1) return GODE_E_ARG;
  if (op->zcols < 16 || op->zcols % 4 != 0) return GODE_E_ARG;
  if (!op->p.W1 || !op->p.b1 || !op->p.W2 || !op->p.b2) return GODE_E_ARG;
  if (op->prenet && (!op->x || !op->p.Wa || !op->p.ba || !op->p.Wb || !op->p.bb)) return GODE_E_ARG;
  if ((op->bstep_off == nullptr) != (op->bstep_dt == nullptr)) return GODE_E_ARG;
  const int nblk = (op->N + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (adaptive) return gode_ode_bwd_multi(op, 1, stream);      // odernn_valu.hip (reduces into grads itself)
  if (!ode_use_mfma(op->N)) {
    const int rc = gode_launch_ode_bwd_valu(op, st);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(ode_bwd_kernel, dim3(nblk), dim3(64), 0, st, *op);
    GODE_LAUNCH_CHECK();
  }
  return ode_bwd_reduce(op, nblk, st);
}

static int ode_bwd_reduce(const gode_ode_bwd_op* op, int nblk, hipStream_t st) {
  const int first = op->prenet ? 0 : OFF_W1;
  const int gx = (GODE_ODE_NPARAM - first + 255) / 256;
  int stride = 1;
  if (nblk > 128) {
    stride = 64;
    hipLaunchKernelGGL(ode_bwd_reduce_chunks_kernel, dim3(gx, (nblk + stride - 1) / stride), dim3(256), 0, st, op->work, nblk, stride, first);
    GODE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(ode_bwd_reduce_kernel, dim3(gx), dim3(256), 0, st, op->work, op->grads, nblk, stride, op->accumulate, first);
  GODE_LAUNCH_CHECK();
  return 0;
}
