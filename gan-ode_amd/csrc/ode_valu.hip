// ode_valu.hip -- the fixed-grid RK4 solve and its adjoint on the VECTOR ALU with DPP row rotations (the default since
// round 2; the MFMA-chain kernels of ode.hip remain selectable with GODE_ODE_MFMA=1 for the A/B in profiles/).
//
// Reference arithmetic: as ode.hip (models/mocogan_ode.py:6-17,123-148; torchdiffeq fixed-grid rk4 = Kutta 3/8 and its
// continuous adjoint, restated in oracle/ode_ref.py).
//
// Mapping: one lane = one (trajectory, feature) pair; the 16 features of a trajectory sit in one DPP row (16 lanes), a
// wave holds 4 trajectories, a 256-thread workgroup 16.  A 16x16 mat-vec is 16 v_fmac_f32 whose second operand is the
// state rotated within the row (`row_ror:r`, r = 0..15) and whose first operand is the weight the lane needs for that
// rotation, W[i][src(i, r)], held in 16 VGPRs per matrix; four independent accumulators keep the dependent chain at 4.
// Against the MFMA mapping of ode.hip (16 trajectories per wave, four chained v_mfma_f32_16x16x4_f32 of 40 cycles
// dependent latency each per mat-vec, 4 tanh per lane):
//   * latency per right-hand side: 2 x (16 FMA issue slots) + ONE tanh per lane instead of 2 x 160 cycles of dependent
//     MFMA + four tanh -- the config-size launch (N = 32: eight waves) is bound by exactly that chain;
//   * throughput: the same 1,024 useful FLOPs per trajectory and right-hand side on the same vector ALUs (fp32 MFMA and
//     fp32 VALU share one peak), without the accumulator-to-operand moves;
//   * adjoint: the parameter-gradient outer products sum_traj P_i Q_j become 16 more FMAs per matrix into PER-LANE
//     accumulators (lane i, rotation r holds dW[i][src(i, r)]) that live in registers for the whole solve and are
//     reduced over trajectories ONCE at the end -- no LDS transpose, no barrier inside the time loop (ode.hip: two
//     barriers + an LDS round trip per outer product, eight per RK4 step).
// Which lane a rotation delivers is read off the hardware once per kernel (the lane index is rotated the same way), so
// the weight arrangement cannot disagree with the DPP semantics.
#include "common.h"
#include "ode_common.h"
#include "valu_common.h"

// All 2,672 parameters (10.7 KB) go to LDS with ONE round of coalesced loads before anything else: the per-lane weight
// arrangement is then gathered from LDS.  (Gathering it from global memory cost the config-size adjoint launch more
// than its whole time loop: twelve dependent batches of L2-missing loads in the pre-net backward, 60k of 126k cycles.)
#define VP_WA 0
#define VP_BA 1024
#define VP_WB 1088
#define VP_BB 2112
#define VP_W1 2128
#define VP_B1 2384
#define VP_W2 2400
#define VP_B2 2656
__device__ __forceinline__ void stage_params(const gode_ode_params& p, int prenet, float* lds) {
  const int t = threadIdx.x;
  if (prenet) {
    for (int k = t; k < 1024; k += 256) { lds[VP_WA + k] = p.Wa[k]; lds[VP_WB + k] = p.Wb[k]; }
    if (t < 64) lds[VP_BA + t] = p.ba[t];
    if (t < 16) lds[VP_BB + t] = p.bb[t];
  }
  lds[VP_W1 + t] = p.W1[t];
  lds[VP_W2 + t] = p.W2[t];
  if (t < 16) { lds[VP_B1 + t] = p.b1[t]; lds[VP_B2 + t] = p.b2[t]; }
  __syncthreads();
}

// content columns 16..65 (the same 50 values on all T rows of a trajectory) + zero pad: one float4 per thread and turn
__device__ __forceinline__ void broadcast_content(const gode_ode_fwd_op& a, int n0) {
  const int rows_per = a.sel_t ? 1 : a.T;
  const int q4 = (a.zcols - 16) >> 2;
  const int total4 = 16 * rows_per * q4;
  for (int k = threadIdx.x; k < total4; k += 256) {
    const int rr = k / q4, q = k - rr * q4;
    const int ns = rr / rows_per, tt = rr - ns * rows_per;
    if (n0 + ns < a.N) {
      const float* c = a.content + (int64_t)(n0 + ns) * 50 + 4 * q;
      f32x4 v = zero4();
      if (q < 12) v = f32x4{c[0], c[1], c[2], c[3]};
      else if (q == 12) v = f32x4{c[0], c[1], 0.f, 0.f};
      *reinterpret_cast<f32x4*>(a.z + ((int64_t)(n0 + ns) * rows_per + tt) * a.zcols + 16 + 4 * q) = v;
    }
  }
}

__global__ void __launch_bounds__(256) ode_fwd_valu_kernel(const gode_ode_fwd_op a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, tl = lane >> 4;
  const int n0 = blockIdx.x * 16, n = n0 + wave * 4 + tl;
  const bool valid = n < a.N;
  const int T = a.T;
  __shared__ float P[GODE_ODE_NPARAM];
  float y = valid ? a.x[n * 16 + i] : 0.f;
  stage_params(a.p, a.prenet, P);
  if (a.content) broadcast_content(a, n0);      // stores only; nothing below waits for them

  int src[16];
  probe_sources(i, src);
  if (a.prenet) {
    // Linear(16,64) -> LReLU -> Linear(64,16) -> LReLU: the 64 hidden units as four row-resident chunks
    float acc = P[VP_BB + i];
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
      W16 wa, wb;
      load_rows(wa, P + VP_WA, 16, 16 * m, 0, i, src);
      load_rows(wb, P + VP_WB, 64, 0, 16 * m, i, src);
      const float h = lrelu1(mv16(wa, y, P[VP_BA + 16 * m + i]));
      acc = mv16(wb, h, acc);
    }
    y = lrelu1(acc);
  }
  W16 w1, w2;
  load_rows(w1, P + VP_W1, 16, 0, 0, i, src);
  load_rows(w2, P + VP_W2, 16, 0, 0, i, src);
  const float b1 = P[VP_B1 + i], b2 = P[VP_B2 + i];
  auto f = [&](float yy) { return mv16(w2, fast_tanh(mv16(w1, yy, b1)), b2); };

  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto emit = [&](int t) {
    if (!valid) return;
    if (a.traj) a.traj[((int64_t)n * T + t) * 16 + i] = y;
    if (a.sel_t == nullptr) a.z[((int64_t)n * T + t) * a.zcols + i] = y;
    else if (t == tsel) a.z[(int64_t)n * a.zcols + i] = y;
  };
  emit(0);
  const float third = 1.0f / 3.0f;
  if (a.grid_dt != nullptr) {
    int jo = 1;
    for (int s = 0; s < a.G; ++s) {
      const float dt = a.grid_dt[s];
      const float y0 = y;
      const float k1 = f(y);
      const float k2 = f(y + dt * k1 * third);
      const float k3 = f(y + dt * (k2 - k1 * third));
      const float k4 = f(y + dt * (k1 - k2 + k3));
      const float y1 = y0 + (k1 + 3.f * (k2 + k3) + k4) * dt * 0.125f;
      while (jo < T && a.emit_at[jo] == s) {
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
      const float k1 = f(y);
      const float k2 = f(y + dt * k1 * third);
      const float k3 = f(y + dt * (k2 - k1 * third));
      const float k4 = f(y + dt * (k1 - k2 + k3));
      y = y + (k1 + 3.f * (k2 + k3) + k4) * dt * 0.125f;
    }
    emit(j + 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// adjoint backward (same parameter-block offsets as ode.hip / gode_ode_params order)
#define VOFF_WA 0
#define VOFF_BA 1024
#define VOFF_WB 1088
#define VOFF_BB 2112
#define VOFF_W1 2128
#define VOFF_B1 2384
#define VOFF_W2 2400
#define VOFF_B2 2656

// sum over the 4 trajectories of a wave (lanes i, i+16, i+32, i+48), result in every lane
__device__ __forceinline__ float sum_traj(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

__global__ void __launch_bounds__(256) ode_bwd_valu_kernel(const gode_ode_bwd_op a) {
  __shared__ float R[4][GODE_ODE_NPARAM];     // per-wave image of the gradient vector (combined once, at the end)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, tl = lane >> 4;
  const int n0 = blockIdx.x * 16, n = n0 + wave * 4 + tl;
  const bool valid = n < a.N;
  const int T = a.T;
  float* part = a.work + (int64_t)blockIdx.x * GODE_ODE_NPARAM;
  const float x = (valid && a.prenet) ? a.x[n * 16 + i] : 0.f;     // needed by the pre-net backward at the very end
#ifdef GODE_ODE_STAMPS   // diagnostic build only (scripts/exp/ode_stamps.sh): where the config-size launch spends its cycles
  const long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  long long st1 = 0, st2 = 0, st3 = 0;
#endif

  __shared__ float P[GODE_ODE_NPARAM];
  stage_params(a.p, a.prenet, P);
  int src[16];
  probe_sources(i, src);
  W16 w1, w2, w1t, w2t, gW1, gW2;
  load_rows(w1, P + VP_W1, 16, 0, 0, i, src);
  load_rows(w2, P + VP_W2, 16, 0, 0, i, src);
  load_cols(w1t, P + VP_W1, 16, 0, 0, i, src);
  load_cols(w2t, P + VP_W2, 16, 0, 0, i, src);
#pragma unroll
  for (int r = 0; r < 16; ++r) { gW1.w[r] = 0.f; gW2.w[r] = 0.f; }
  const float b1 = P[VP_B1 + i], b2 = P[VP_B2 + i];
  float gb1 = 0.f, gb2 = 0.f;

  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto upstream = [&](int t) -> float {
    if (!valid) return 0.f;
    if (a.sel_t == nullptr) return a.gz[((int64_t)n * T + t) * a.zcols + i];
    return t == tsel ? a.gz[(int64_t)n * a.zcols + i] : 0.f;
  };
  float adj = upstream(T - 1);
  float ky, ka;
  // one stage of the reversed augmented dynamics at (ys, as); c = RK weight * dt/8 for the parameter integrals
  auto stage = [&](float ys, float as, float c) {
    const float h = fast_tanh(mv16(w1, ys, b1));
    const float fv = mv16(w2, h, b2);
    const float v = mv16(w2t, as, 0.f);
    const float du = v * (1.f - h * h);
    ka = mv16(w1t, du, 0.f);
    ky = -fv;
    const float ca = c * as, cdu = c * du;
    gb2 += ca; gb1 += cdu;
    outer16(gW2, ca, h);
    outer16(gW1, cdu, ys);
  };

#ifdef GODE_ODE_STAMPS
  st1 = __builtin_amdgcn_s_memtime();
#endif
  const float third = 1.0f / 3.0f;
  // the stored state and the upstream gradient of the NEXT interval are requested before the four stages of the
  // current one, so the (dependent, ~1 us each at N = 32) loads land under ~600 cycles of arithmetic
  float y_next = (valid && T > 1) ? a.traj[((int64_t)n * T + (T - 1)) * 16 + i] : 0.f;
  for (int it = T - 1; it >= 1; --it) {
    float y = y_next;
    const float up = upstream(it - 1);
    if (it > 1) y_next = valid ? a.traj[((int64_t)n * T + (it - 1)) * 16 + i] : 0.f;
    const int s0 = a.bstep_off ? a.bstep_off[it - 1] : 0;
    const int ns = a.bstep_off ? a.bstep_off[it] - s0 : a.substeps;
    const float dt_eq = a.dt[it - 1] / (float)a.substeps;
    for (int ss = 0; ss < ns; ++ss) {
      const float dt = a.bstep_off ? a.bstep_dt[s0 + ss] : dt_eq;
      stage(y, adj, dt * 0.125f);
      const float ky1 = ky, ka1 = ka;
      stage(y + dt * ky1 * third, adj + dt * ka1 * third, 3.f * dt * 0.125f);
      const float ky2 = ky, ka2 = ka;
      stage(y + dt * (ky2 - ky1 * third), adj + dt * (ka2 - ka1 * third), 3.f * dt * 0.125f);
      const float ky3 = ky, ka3 = ka;
      stage(y + dt * (ky1 - ky2 + ky3), adj + dt * (ka1 - ka2 + ka3), dt * 0.125f);
      y = y + (ky1 + 3.f * (ky2 + ky3) + ky) * dt * 0.125f;
      adj = adj + (ka1 + 3.f * (ka2 + ka3) + ka) * dt * 0.125f;
    }
    adj = adj + up;
  }

#ifdef GODE_ODE_STAMPS
  st2 = __builtin_amdgcn_s_memtime();
#endif
  // ---- reduce over trajectories: 4 per wave by shuffles; every wave deposits its sums for ALL parameters in its own
  // LDS image of the gradient vector, ONE barrier at the very end, then the four images are added in fixed order and
  // written out coalesced.  (Writing each matrix to global memory as it was finished put a store drain -- the
  // s_waitcnt vmcnt(0) of the next barrier -- behind each of 17 pieces: 47k cycles.)
  auto put_matrix = [&](const W16& Gm, int off, int ld, int row0, int col0) {
    float t[16];      // the 16 shuffles of a level are independent: issue them all, then add (latency paid once per level)
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = __shfl_xor(Gm.w[r], 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] += Gm.w[r];
    float u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = __shfl_xor(t[r], 32);
    if (tl == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) R[wave][off + (row0 + i) * ld + col0 + src[r]] = t[r] + u[r];
    }
  };
  auto put_vector = [&](float v, int off) {
    const float s = sum_traj(v);
    if (tl == 0) R[wave][off + i] = s;
  };
  put_matrix(gW1, VOFF_W1, 16, 0, 0);
  put_matrix(gW2, VOFF_W2, 16, 0, 0);
  put_vector(gb1, VOFF_B1);
  put_vector(gb2, VOFF_B2);

  // ---- pre-net backward; adj = dL/d(pre-net output)
  if (a.prenet) {
    float hpre[4];
    float acc = P[VP_BB + i];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      W16 wa, wb;
      load_rows(wa, P + VP_WA, 16, 16 * m, 0, i, src);
      load_rows(wb, P + VP_WB, 64, 0, 16 * m, i, src);
      hpre[m] = mv16(wa, x, P[VP_BA + 16 * m + i]);
      acc = mv16(wb, lrelu1(hpre[m]), acc);
    }
    const float g0 = acc > 0.f ? adj : 0.2f * adj;
    put_vector(g0, VOFF_BB);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      W16 G, wbt;
#pragma unroll
      for (int r = 0; r < 16; ++r) G.w[r] = 0.f;
      outer16(G, g0, lrelu1(hpre[m]));                       // dWb[i][16m + j] = sum g0_i * lrelu(hpre)_j
      put_matrix(G, VOFF_WB, 64, 0, 16 * m);
      load_cols(wbt, P + VP_WB, 64, 0, 16 * m, i, src);      // (Wb chunk)^T
      const float t = mv16(wbt, g0, 0.f);
      const float gh = hpre[m] > 0.f ? t : 0.2f * t;
      put_vector(gh, VOFF_BA + 16 * m);
#pragma unroll
      for (int r = 0; r < 16; ++r) G.w[r] = 0.f;
      outer16(G, gh, x);                                     // dWa[16m + i][j] = sum gh_i * x_j
      put_matrix(G, VOFF_WA, 16, 16 * m, 0);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < GODE_ODE_NPARAM; k += 256)
    part[k] = (k < VOFF_W1 && !a.prenet) ? 0.f : (R[0][k] + R[1][k]) + (R[2][k] + R[3][k]);
#ifdef GODE_ODE_STAMPS
  st3 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long rt1 = __builtin_amdgcn_s_memrealtime();
    printf("ode_bwd_valu stamps (cycles): setup %lld  time-loop %lld  epilogue %lld  total %lld | %.2f us, clock %.0f MHz (adj %g)\n",
           st1 - st0, st2 - st1, st3 - st2, st3 - st0, (double)(rt1 - rt0) / 100.0, (double)(st3 - st0) / ((double)(rt1 - rt0) / 100.0), (double)adj);
  }
#endif
}

int gode_launch_ode_fwd_valu(const gode_ode_fwd_op* op, hipStream_t st) {
  hipLaunchKernelGGL(ode_fwd_valu_kernel, dim3((op->N + 15) / 16), dim3(256), 0, st, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}
int gode_launch_ode_bwd_valu(const gode_ode_bwd_op* op, hipStream_t st) {
  hipLaunchKernelGGL(ode_bwd_valu_kernel, dim3((op->N + 15) / 16), dim3(256), 0, st, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}
