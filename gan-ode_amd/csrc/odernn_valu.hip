// odernn_valu.hip -- the ODE-RNN motion latent (models/mocogan_ode_rnn.py:39-51; BASELINE configs[4]) on the VECTOR ALU
// with DPP row rotations: the default kernels since round 3 (odernn.hip / adj_adaptive.hip keep the MFMA-chain kernels
// as the fallback above GODE_ODERNN_SYNC_MAX_N trajectories per launch).
//
//   h_0 ~ N(0,1);  per frame:  h' = odeint_adjoint(ODEFunc, h, [0,1])[-1]  (torchdiffeq default: dopri5, rtol 1e-7,
//   atol 1e-9),  h = GRUCell(e_t, h'),  e_t ~ N(0,1);  latent row t = h_{t+1}.
//
// Why this mapping.  The solve is one long dependent chain: 16 frames x ~7 trial steps x 6 right-hand sides at batch 32,
// on ONE workgroup.  The MFMA mapping (16 trajectories per wave, 4 values per lane) spent ~6,000 cycles per trial step:
// ~600 instructions of a single wave per SIMD, each paying its full issue + dependency latency (48 dependent MFMAs with
// their hazard nops, 4 tanh per lane, six ds_bpermute round trips per norm).  Here one lane owns one (trajectory,
// feature) value -- 8 waves for 32 trajectories, two per SIMD covering each other's DPP / transcendental wait states --
// and the algebra is arranged so that ONE 16x16 mat-vec per stage sits on the dependent chain:
//
//   f(y) = W2 tanh(W1 y + b1) + b2.  With u = W1 y + b1 integrated instead of y, a stage input is
//       u_i = u_0 + dt * sum_j a_ij (W1 k_j) = u_0 + dt * sum_j a_ij q_j,    q_j = M h_j + c,  h_j = tanh(u_j),
//   M = W1 W2 and c = W1 b2 (formed once per launch, accumulated in fp64).  The 5th-order solution, the error estimate
//   and the dense-output value are linear in the h_j, so each costs one W2 mat-vec per trial step:
//       y_1 = y_0 + dt (W2 sum_j b_j h_j + b2),   err = dt W2 sum_j e_j h_j   (sum_j e_j = 0),
//       y(x) = y_0 + dt (W2 (wS Hb + wM Hm + w1 h_1 + w7 h_7) + b2 (wS + wM/2 + w1 + w7))   [4th-order interpolant].
//   Same tableau, controller, norm and dense output as torchdiffeq (oracle/ode_ref.py:dopri5_solve); only the order of
//   the floating-point operations inside a right-hand side differs (states agree with the oracle to ~1e-6).
//
// Error norm: the RMS over ALL trajectories of the call, as torchdiffeq takes it.  A workgroup holds 32 trajectories;
// the workgroups of a call exchange their partial sums through `sync` (store partial, release-increment a counter, spin
// until every workgroup of the call has arrived, add the partials in fixed order): every workgroup reaches bit-identical
// accept / reject decisions, and re-runs are bit-identical (no float atomics).  Every spin is bounded; a stalled call
// reports a negative step count instead of hanging.  Several independent solves share one launch (`_multi`).
#include "common.h"
#include "ode_common.h"
#include "valu_common.h"

#ifndef RV_WAVES
#define RV_WAVES 8                   // (scripts/exp/odernn_waves.sh builds 4 to measure one wave per SIMD)
#endif
#define RV_TRAJ (4 * RV_WAVES)
#define RV_THREADS (64 * RV_WAVES)
#define RV_MAX_JOBS 8
#define RV_MAX_WGS (GODE_ODERNN_SYNC_MAX_N / RV_TRAJ)
#define RV_MAX_TRIALS 20000          // per frame; torchdiffeq's max_num_steps is 2^31 - 1
#define XS_VALS 8
#define XS_HDR 16

#define RO_W1 0
#define RO_B1 256
#define RO_W2 272
#define RO_B2 528
#define RO_WIH 544
#define RO_WHH 1312
#define RO_BIH 2080
#define RO_BHH 2128
#define RO_N 2176
#define RP_M 2176
#define RP_C 2432
#define RP_N 2448

// ---- dopri5 tableau (torchdiffeq _DORMAND_PRINCE_SHAMPINE_TABLEAU) ---------------------------------------------------
#define A21 (1.f / 5.f)
#define A31 (3.f / 40.f)
#define A32 (9.f / 40.f)
#define A41 (44.f / 45.f)
#define A42 (-56.f / 15.f)
#define A43 (32.f / 9.f)
#define A51 (19372.f / 6561.f)
#define A52 (-25360.f / 2187.f)
#define A53 (64448.f / 6561.f)
#define A54 (-212.f / 729.f)
#define A61 (9017.f / 3168.f)
#define A62 (-355.f / 33.f)
#define A63 (46732.f / 5247.f)
#define A64 (49.f / 176.f)
#define A65 (-5103.f / 18656.f)
#define B1 (35.f / 384.f)
#define B3 (500.f / 1113.f)
#define B4 (125.f / 192.f)
#define B5 (-2187.f / 6784.f)
#define B6 (11.f / 84.f)
#define E1 (35.f / 384.f - 1951.f / 21600.f)
#define E3 (500.f / 1113.f - 22642.f / 50085.f)
#define E4 (125.f / 192.f - 451.f / 720.f)
#define E5 (-2187.f / 6784.f + 12231.f / 42400.f)
#define E6 (11.f / 84.f - 649.f / 6300.f)
#define E7 (-1.f / 60.f)
#define C1 (6025192743.f / 30085553152.f / 2.f)
#define C3 (51252292925.f / 65400821598.f / 2.f)
#define C4 (-2691868925.f / 45128329728.f / 2.f)
#define C5 (187940372067.f / 1594534317056.f / 2.f)
#define C6 (-1776094331.f / 19743644256.f / 2.f)
#define C7 (11237099.f / 235043384.f / 2.f)

// x^p through the hardware log2 / exp2 units (1 ulp each): the step-size controller needs 3-4 digits
__device__ __forceinline__ float fast_pow(float x, float p) { return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)); }

// torchdiffeq _optimal_step_size (order 5): safety 0.9, ifactor 10, dfactor 0.2; a step with ratio < 1 never shrinks
__device__ __forceinline__ float step_factor(float ratio) {
  if (ratio == 0.f) return 10.f;
  const float f = 0.9f * fast_pow(ratio, -0.2f);
  return fminf(10.f, fmaxf(f, ratio < 1.f ? 1.f : 0.2f));
}
// weights of the 4th-order interpolant at abscissa x in terms of (y1 - y0, ymid - y0, dt f0, dt f1)
__device__ __forceinline__ void interp_weights(float x, float& wS, float& wM, float& w1c, float& w7c) {
  const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
  wS = -8.f * x4 + 14.f * x3 - 5.f * x2;
  wM = 16.f * x4 - 32.f * x3 + 16.f * x2;
  w1c = -2.f * x4 + 5.f * x3 - 4.f * x2 + x;
  w7c = 2.f * x4 - 3.f * x3 + x2;
}

// ---- whole-batch sums ------------------------------------------------------------------------------------------------
struct XSync {
  int* counter;     // sync[0]: arrivals so far, over all exchanges of the launch
  float* slots;     // sync + XS_HDR: [2][nwg][XS_VALS] partial sums, double-buffered by exchange parity
  int nwg, wg, seq;
};
struct RedLds {
  float red[2][RV_WAVES][XS_VALS];   // per-wave partials, double-buffered: one barrier per sum
  float bc[2][XS_VALS];              // cross-workgroup totals broadcast to the workgroup
  int dead;                          // a spin timed out somewhere: every thread of the workgroup bails out
};

// v[0..K): per-thread values -> the sums over ALL threads of ALL workgroups of the call, in every thread, fixed order.
template <int K>
__device__ __forceinline__ void batch_sum(float* v, RedLds* R, int& par, XSync& X) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) R->red[par][wave][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < RV_WAVES; ++w) s += R->red[par][w][k];
    v[k] = s;
  }
  if (X.nwg > 1) {
    float* slot = X.slots + (size_t)(X.seq & 1) * X.nwg * XS_VALS;
    if (wave == 0) {
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) __hip_atomic_store(&slot[X.wg * XS_VALS + k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(X.counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (X.seq + 1) * X.nwg;
        int spins = 0;
        while (__hip_atomic_load(X.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 20)) { R->dead = 1; break; }     // ~seconds: a workgroup of the call never arrived
        }
      }
      float t[K];
#pragma unroll
      for (int k = 0; k < K; ++k) t[k] = 0.f;
      for (int w = lane; w < X.nwg; w += 64) {
#pragma unroll
        for (int k = 0; k < K; ++k) t[k] += __hip_atomic_load(&slot[w * XS_VALS + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) t[k] = wave_sum(t[k]);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) R->bc[par][k] = t[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = R->bc[par][k];
    X.seq += 1;
  }
  par ^= 1;
}

// Stage all 2,176 parameters in LDS with one round of coalesced loads, then M = W1 W2 and c = W1 b2 (fp64 accumulation).
__device__ __forceinline__ void stage_rnn_params(const gode_odernn_params& p, float* P, bool fused) {
  const int t = threadIdx.x;
  for (int k = t; k < 256; k += RV_THREADS) { P[RO_W1 + k] = p.W1[k]; P[RO_W2 + k] = p.W2[k]; }
  for (int k = t; k < 768; k += RV_THREADS) { P[RO_WIH + k] = p.Wih[k]; P[RO_WHH + k] = p.Whh[k]; }
  if (t < 16) { P[RO_B1 + t] = p.b1[t]; P[RO_B2 + t] = p.b2[t]; }
  if (t < 48) { P[RO_BIH + t] = p.bih[t]; P[RO_BHH + t] = p.bhh[t]; }
  __syncthreads();
  if (fused) {
    if (t < 256) {
      const int i = t >> 4, j = t & 15;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += (double)P[RO_W1 + i * 16 + k] * (double)P[RO_W2 + k * 16 + j];
      P[RP_M + t] = (float)s;
    } else if (t < 272) {
      const int i = t - 256;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += (double)P[RO_W1 + i * 16 + k] * (double)P[RO_B2 + k];
      P[RP_C + i] = (float)s;
    }
    __syncthreads();
  }
}

// content columns 16..65 (the same 50 values on all rows of a trajectory) + zero pad for this workgroup's trajectories
__device__ __forceinline__ void rnn_broadcast_content(const float* content, float* z, const int32_t* sel_t, int N, int T, int zcols, int n0) {
  const int rows_per = sel_t ? 1 : T;
  const int q4 = (zcols - 16) >> 2;
  const int total4 = RV_TRAJ * rows_per * q4;
  for (int k = threadIdx.x; k < total4; k += RV_THREADS) {
    const int rr = k / q4, q = k - rr * q4;
    const int ns = rr / rows_per, tt = rr - ns * rows_per;
    if (n0 + ns < N) {
      const float* c = content + (int64_t)(n0 + ns) * 50 + 4 * q;
      f32x4 v = zero4();
      if (q < 12) v = f32x4{c[0], c[1], c[2], c[3]};
      else if (q == 12) v = f32x4{c[0], c[1], 0.f, 0.f};
      *reinterpret_cast<f32x4*>(z + ((int64_t)(n0 + ns) * rows_per + tt) * zcols + 16 + 4 * q) = v;
    }
  }
}

struct RnnFwdJobs { gode_odernn_fwd_op op[RV_MAX_JOBS]; int32_t nblk[RV_MAX_JOBS]; int32_t count; };

__global__ void __launch_bounds__(RV_THREADS) odernn_fwd_valu_kernel(const RnnFwdJobs J) {
  __shared__ float P[RP_N];
  __shared__ RedLds R;
  int job = 0, wg = blockIdx.x;
  while (job + 1 < J.count && wg >= J.nblk[job]) { wg -= J.nblk[job]; ++job; }
  const gode_odernn_fwd_op a = J.op[job];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, tl = lane >> 4;
  const int n0 = wg * RV_TRAJ, n = n0 + wave * 4 + tl;
  const bool valid = n < a.N;
  const int T = a.T;
  const float inv_count = 1.f / ((float)a.N * 16.f);
  float rtol = a.rtol, atol = a.atol;
  XSync X;
  X.nwg = J.nblk[job]; X.wg = wg; X.seq = 0;
  X.counter = a.sync; X.slots = a.sync ? reinterpret_cast<float*>(a.sync) + XS_HDR : nullptr;
  int par = 0;
  if (threadIdx.x == 0) R.dead = 0;

#ifdef GODE_ODE_STAMPS
  const long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  long long acc_stage = 0, acc_mv = 0, acc_sum = 0, acc_ctl = 0, last_end = 0, acc_trials = 0, st_loop = 0;
#endif
  float h = valid ? a.noise[(int64_t)n * 16 + i] : 0.f;
  stage_rnn_params(a.p, P, true);
  if (a.content) rnn_broadcast_content(a.content, a.z, a.sel_t, a.N, T, a.zcols, n0);      // stores only

  int src[16];
  probe_sources(i, src);
  W16 Mw, w1, w2, wih[3], whh[3];
  load_rows(Mw, P + RP_M, 16, 0, 0, i, src);
  load_rows(w1, P + RO_W1, 16, 0, 0, i, src);
  load_rows(w2, P + RO_W2, 16, 0, 0, i, src);
  float bih[3], bhh[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    load_rows(wih[q], P + RO_WIH, 16, 16 * q, 0, i, src);
    load_rows(whh[q], P + RO_WHH, 16, 16 * q, 0, i, src);
    bih[q] = P[RO_BIH + 16 * q + i]; bhh[q] = P[RO_BHH + 16 * q + i];
  }
  const float b1 = P[RO_B1 + i], b2 = P[RO_B2 + i], cq = P[RP_C + i];
  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  if (valid && a.hs) a.hs[((int64_t)n * (T + 1)) * 16 + i] = h;

#ifdef GODE_ODE_STAMPS
  st_loop = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < T; ++t) {
#ifdef GODE_ODE_STAMPS
    last_end = 0;
#endif
    // (this frame's GRU input is requested now: a dependent ~2 us HBM round trip that would otherwise sit between the
    // solve and the GRU on every frame -- 25 % of the launch in the stamped build)
    const float e = valid ? a.noise[((int64_t)(t + 1) * a.N + n) * 16 + i] : 0.f;
    // ---- h' = y(1), y' = f(y), y(0) = h.  The clock and the step size are fp64, as torchdiffeq keeps them.
    float y0 = h;
    float u0 = mv16c(w1, y0, b1);
    float h1 = fast_tanh(u0);
    float q1 = mv16c(Mw, h1, cq);
    double dtd;
    {   // torchdiffeq _select_initial_step (order 4)
      const float f0 = mv16c(w2, h1, b2);
      const float sc = atol + fabsf(y0) * rtol;
      float v[2] = {valid ? (y0 / sc) * (y0 / sc) : 0.f, valid ? (f0 / sc) * (f0 / sc) : 0.f};
      batch_sum<2>(v, &R, par, X);
      const float d0 = sqrtf(v[0] * inv_count), d1 = sqrtf(v[1] * inv_count);
      const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
      const float df = mv16c(w2, fast_tanh(fmaf(h0, q1, u0)) - h1, 0.f);          // f(y0 + h0 f0) - f0
      float w[1] = {valid ? (df / sc) * (df / sc) : 0.f};
      batch_sum<1>(w, &R, par, X);
      const float d2 = sqrtf(w[0] * inv_count) / h0;
      const float hh = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : fast_pow(0.01f / fmaxf(d1, d2), 0.2f);
      dtd = (double)fminf(100.f * h0, hh);
    }
    double tcur = 0.0;
    float yend = y0;
    int steps = 0;
    bool stalled = false;
    for (;;) {
      if (R.dead || steps >= RV_MAX_TRIALS || !(tcur + dtd > tcur)) { stalled = true; break; }
#ifdef GODE_ODE_STAMPS
      const long long sA = __builtin_amdgcn_s_memtime();
#endif
      const float dt = (float)dtd;
      const float h2 = fast_tanh(fmaf(dt * A21, q1, u0));
      const float q2 = mv16c(Mw, h2, cq);
      const float h3 = fast_tanh(fmaf(dt, A31 * q1 + A32 * q2, u0));
      const float q3 = mv16c(Mw, h3, cq);
      const float h4 = fast_tanh(fmaf(dt, A41 * q1 + A42 * q2 + A43 * q3, u0));
      const float q4 = mv16c(Mw, h4, cq);
      const float h5 = fast_tanh(fmaf(dt, A51 * q1 + A52 * q2 + A53 * q3 + A54 * q4, u0));
      const float q5 = mv16c(Mw, h5, cq);
      const float h6 = fast_tanh(fmaf(dt, A61 * q1 + A62 * q2 + A63 * q3 + A64 * q4 + A65 * q5, u0));
      const float q6 = mv16c(Mw, h6, cq);
      const float u7 = fmaf(dt, B1 * q1 + B3 * q3 + B4 * q4 + B5 * q5 + B6 * q6, u0);
      const float h7 = fast_tanh(u7);
#ifdef GODE_ODE_STAMPS
      asm volatile("" ::"v"(h7));
      const long long sB = __builtin_amdgcn_s_memtime();
#endif
      const float Hb = B1 * h1 + B3 * h3 + B4 * h4 + B5 * h5 + B6 * h6;
      const float He = E1 * h1 + E3 * h3 + E4 * h4 + E5 * h5 + E6 * h6 + E7 * h7;
      const float y1 = fmaf(dt, mv16c(w2, Hb, b2), y0);
      const float err = dt * mv16c(w2, He, 0.f);
      const float tol = atol + rtol * fmaxf(fabsf(y0), fabsf(y1));
      const float rr = err / tol;
      float v[1] = {valid ? rr * rr : 0.f};
      const float q7 = mv16c(Mw, h7, cq);            // next step's q1 if this one is accepted: issued under the sum's latency
#ifdef GODE_ODE_STAMPS
      asm volatile("" ::"v"(q7), "v"(v[0]));
      const long long sC = __builtin_amdgcn_s_memtime();
#endif
      batch_sum<1>(v, &R, par, X);
      const float ratio = sqrtf(v[0] * inv_count);
      ++steps;
#ifdef GODE_ODE_STAMPS
      {
        const long long sD = __builtin_amdgcn_s_memtime();
        acc_stage += sB - sA; acc_mv += sC - sB; acc_sum += sD - sC; acc_trials += 1;
        if (last_end) acc_ctl += sA - last_end;
        last_end = sD;
      }
#endif
      if (ratio <= 1.f) {
        if (tcur + dtd >= 1.0) {   // dense output at t = 1 (4th-order interpolant through the mid-point)
          const float x = (float)((1.0 - tcur) / dtd);
          float wS, wM, w1c, w7c;
          interp_weights(x, wS, wM, w1c, w7c);
          const float Hm = C1 * h1 + C3 * h3 + C4 * h4 + C5 * h5 + C6 * h6 + C7 * h7;
          const float Hx = wS * Hb + wM * Hm + w1c * h1 + w7c * h7;
          yend = fmaf(dt, mv16c(w2, Hx, b2 * (wS + 0.5f * wM + w1c + w7c)), y0);
          break;
        }
        tcur += dtd; y0 = y1; u0 = u7; h1 = h7; q1 = q7;
      }
      dtd *= (double)step_factor(ratio);
    }
    if (a.nsteps && wg == 0 && threadIdx.x == 0) a.nsteps[t] = stalled ? -steps - 1 : steps;
    if (stalled) return;              // uniform over the workgroup (and, through the shared norm, over the call)
    if (valid && a.hp) a.hp[((int64_t)n * T + t) * 16 + i] = yend;
    // ---- GRUCell(e_t, h')
    const float r = fast_sigmoid(mv16c(wih[0], e, bih[0]) + mv16c(whh[0], yend, bhh[0]));
    const float zg = fast_sigmoid(mv16c(wih[1], e, bih[1]) + mv16c(whh[1], yend, bhh[1]));
    const float nn = fast_tanh(mv16c(wih[2], e, bih[2]) + r * mv16c(whh[2], yend, bhh[2]));
    h = (1.f - zg) * nn + zg * yend;
    if (valid) {
      if (a.hs) a.hs[((int64_t)n * (T + 1) + t + 1) * 16 + i] = h;
      if (a.sel_t == nullptr) a.z[((int64_t)n * T + t) * a.zcols + i] = h;
      else if (t == tsel) a.z[(int64_t)n * a.zcols + i] = h;
    }
  }
#ifdef GODE_ODE_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long st1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    printf("odernn_fwd_valu: %lld cycles in %.1f us -> %.0f MHz; set-up %lld; %lld trial steps: 6 stages %lld | y1, err, q7 mat-vecs %lld | "
           "batch sum %lld | controller %lld; rest of the frames (initial step, dense output, GRU, stores) %lld\n",
           st1 - st0, (double)(rt1 - rt0) / 100.0, (double)(st1 - st0) / ((double)(rt1 - rt0) / 100.0), st_loop - st0, acc_trials,
           acc_stage, acc_mv, acc_sum, acc_ctl, (st1 - st_loop) - acc_stage - acc_mv - acc_sum - acc_ctl);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
extern "C" int64_t gode_odernn_sync_size(int32_t N) {
  const int64_t nwg = (N + RV_TRAJ - 1) / RV_TRAJ;
  return nwg > 1 ? XS_HDR + 2 * nwg * (2 * 544 + 8) : 0;      // the adjoint's exchange (theta images + 2 sums) is the larger one
}

static bool rnn_fwd_args_ok(const gode_odernn_fwd_op* op) {
  if (!op || !op->noise || !op->z || op->N <= 0 || op->T < 1 || !(op->rtol > 0.f) || !(op->atol >= 0.f)) return false;
  if (op->zcols < 68 || op->zcols % 4 != 0) return false;
  if (!op->p.W1 || !op->p.b1 || !op->p.W2 || !op->p.b2 || !op->p.Wih || !op->p.Whh || !op->p.bih || !op->p.bhh) return false;
  if (op->N > RV_TRAJ && op->N <= GODE_ODERNN_SYNC_MAX_N && !op->sync) return false;
  return true;
}

int gode_launch_odernn_fwd_mfma(const gode_odernn_fwd_op* op, hipStream_t st);   // odernn.hip (fallback above the co-residency limit)

extern "C" int gode_odernn_fwd_multi(const gode_odernn_fwd_op* ops, int32_t count, void* stream) {
  if (!ops || count < 1 || count > RV_MAX_JOBS) return GODE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  RnnFwdJobs J;
  J.count = 0;
  int spinning = 0;      // workgroups that wait for each other (solves of more than one workgroup)
  for (int k = 0; k < count; ++k) {
    if (!rnn_fwd_args_ok(&ops[k])) return GODE_E_ARG;
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    if (nblk > 1) spinning += nblk;
  }
  for (int k = 0; k < count; ++k) {
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    if (spinning > RV_MAX_WGS && nblk > 1) {
      // too many workgroups to guarantee co-residency of a spinning launch: these solves take the MFMA kernel with one
      // norm per 64-trajectory workgroup (see the header); one-workgroup solves stay below
      const int rc = gode_launch_odernn_fwd_mfma(&ops[k], st);
      if (rc) return rc;
      continue;
    }
    if (nblk > 1) {
      const hipError_t e = hipMemsetAsync(ops[k].sync, 0, XS_HDR * sizeof(int32_t), st);
      if (e != hipSuccess) return (int)e;
    }
    J.op[J.count] = ops[k]; J.nblk[J.count] = nblk; J.count += 1;
  }
  if (J.count == 0) return 0;
  int grid = 0;
  for (int k = 0; k < J.count; ++k) grid += J.nblk[k];
  hipLaunchKernelGGL(odernn_fwd_valu_kernel, dim3(grid), dim3(RV_THREADS), 0, st, J);
  GODE_LAUNCH_CHECK();
  return 0;
}

extern "C" int gode_odernn_fwd(const gode_odernn_fwd_op* op, void* stream) { return gode_odernn_fwd_multi(op, 1, stream); }

// =================================================================================================================
// Backward: GRU backward + torchdiffeq's ADAPTIVE adjoint of every unit-interval solve (per frame one adjoint call:
// dopri5 on the augmented state (y, a, g_theta) from t = 1 back to t = 0 with the "mixed" norm -- the maximum over the
// RMS norms of y, a and each ODEFunc parameter tensor; oracle/ode_ref.py:_Adjoint.backward).
//
// Same lane mapping as the forward.  It makes the parameter components cheap: with one (trajectory, feature) value per
// lane, a lane's a_i and h_j ARE the A / B operands of v_mfma_f32_16x16x4_f32 for k = the wave's 4 trajectories, so the
// batch sum of outer products  sum_n a_n (x) h_n  is ONE MFMA per matrix and stage, accumulating in registers -- no LDS
// transposes (adj_adaptive.hip spent most of its ~13,000 cycles per trial step on them).  Per trial step a wave
// accumulates the three weighted stage sums the controller needs (5th-order increment S, error estimate E, mid-point M:
// the A operand is scaled by the tableau weight), the waves' images meet in LDS once for the norm, and each thread owns
// two elements of the workgroup total, so the accepted state's total stays in registers.
// =================================================================================================================
#define TH_W1 0
#define TH_B1 256
#define TH_W2 272
#define TH_B2 528
#define TH_N 544
#define GP_LD 17                     // padded row stride of the GRU matrices in LDS (conflict-free row / column gathers)
#define GP_WIH 0
#define GP_WHH (48 * GP_LD)
#define GP_N (96 * GP_LD)
#define XB_VALS (2 * TH_N + 8)       // per workgroup and exchange: theta images (V, S) + the y / a partial sums

struct Acc4 { f32x4 W1, W2; float b1, b2; };      // W*: MFMA D layout (lane (g, s) reg r = [4g + r][s]); b*: per lane (trajectory, feature)
__device__ __forceinline__ void acc_zero(Acc4& A) { A.W1 = zero4(); A.W2 = zero4(); A.b1 = 0.f; A.b2 = 0.f; }

struct AdjLdsV {
  float img[RV_WAVES][2][TH_N];      // per-wave images (slot 0: V / E, slot 1: S)
  float gru[3][TH_N];                // GRU gradient totals of the workgroup, one 544-block per gate (Wih_q, Whh_q, bih_q, bhh_q);
                                     // every element is owned by one thread (element e: thread e % 512)
};

struct AdjV {
  W16 w1, w2, w1t, w2t;
  float b1, b2, rtol, atol, inv_ya;
  int i, tl, wave, lane;
  bool valid;
  AdjLdsV* L;
  RedLds* R;
  XSync X;
  float* xslots;                     // sync + XS_HDR: [2][nwg][XB_VALS]
  int par;
  float gt[2];                       // this thread's two elements of the accepted theta state's BATCH total
  float st[2];                       // ... of the last trial's 5th-order increment (kept from the norm pass)

  // reversed-time augmented dynamics at (y, a): dy = -f(y), da = +a^T df/dy; h and du are the operands of the theta terms
  __device__ __forceinline__ void eval(float y, float a, float& ky, float& ka, float& h, float& du) {
    h = fast_tanh(mv16c(w1, y, b1));
    const float fv = mv16c(w2, h, b2);
    const float v = mv16c(w2t, a, 0.f);
    du = v * (1.f - h * h);
    ka = mv16c(w1t, du, 0.f);
    ky = -fv;
  }
  // A += c * (theta dynamics at a stage):  dW2 += c a (x) h,  dW1 += c du (x) y,  db2 += c a,  db1 += c du
  __device__ __forceinline__ void theta_acc(Acc4& A, float c, float a, float h, float du, float y) {
    const float ca = c * a, cdu = c * du;
    A.W2 = MFMA16(ca, h, A.W2);
    A.W1 = MFMA16(cdu, y, A.W1);
    A.b2 += ca; A.b1 += cdu;
  }
  __device__ __forceinline__ void put_image(int slot, const Acc4& V, float scale) {
    float* im = L->img[wave][slot];
    const float sb1 = traj_sum(V.b1) * scale, sb2 = traj_sum(V.b2) * scale;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      im[TH_W1 + (4 * tl + r) * 16 + i] = V.W1[r] * scale;
      im[TH_W2 + (4 * tl + r) * 16 + i] = V.W2[r] * scale;
    }
    if (tl == 0) { im[TH_B1 + i] = sb1; im[TH_B2 + i] = sb2; }
  }
  __device__ __forceinline__ float img_total(int slot, int e) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < RV_WAVES; ++w) t += L->img[w][slot][e];
    return t;
  }
  __device__ __forceinline__ void arrive_and_wait() {     // thread 0 only
    __hip_atomic_fetch_add(X.counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const int target = (X.seq + 1) * X.nwg;
    int spins = 0;
    while (__hip_atomic_load(X.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 20)) { R->dead = 1; break; }
    }
  }
  // torchdiffeq's mixed norm of an augmented vector: its y / a parts per lane (vy, va, already divided by their scale), its
  // theta part as the per-wave images in LDS -- mode 0: slot 0 = V, scale = atol + rtol |G|;  mode 1: slot 0 = E, slot 1 = S,
  // scale = atol + rtol max(|G|, |G + S|);  mode -1: no theta part (the state itself at the start of a call: G = 0);
  // mode 2: the accepted theta state G itself (a later output interval of the same adjoint call: G carried over), whose
  // batch total every thread already owns -- no images.  The batch totals of S stay in st[] for the accepted-state update.
  __device__ __forceinline__ float mixed(float vy, float va, int mode) {
    float v[6];
    v[0] = valid ? vy * vy : 0.f;
    v[1] = valid ? va * va : 0.f;
    v[2] = v[3] = v[4] = v[5] = 0.f;
    float V[2] = {0.f, 0.f}, S[2] = {0.f, 0.f};
    if (mode == 0 || mode == 1) {
      __syncthreads();                     // the images written by put_image are complete
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = (int)threadIdx.x + RV_THREADS * k;
        if (e < TH_N) { V[k] = img_total(0, e); if (mode == 1) S[k] = img_total(1, e); }
      }
    }
    // workgroup sums of the y / a parts (one barrier; it also orders the image reads above before the next put_image)
    {
      const float a0 = wave_sum(v[0]), a1 = wave_sum(v[1]);
      if (lane == 0) { R->red[par][wave][0] = a0; R->red[par][wave][1] = a1; }
      __syncthreads();
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int w = 0; w < RV_WAVES; ++w) { s0 += R->red[par][w][0]; s1 += R->red[par][w][1]; }
      v[0] = s0; v[1] = s1;
      par ^= 1;
    }
    if (X.nwg > 1) {
      // the theta components are BATCH sums, and so are the y / a sums of squares: every workgroup's totals cross through
      // global memory once per norm and are added in workgroup order, so all workgroups form bit-identical values
      float* slot = xslots + (size_t)(X.seq & 1) * X.nwg * XB_VALS;
      float* mine = slot + (size_t)X.wg * XB_VALS;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = (int)threadIdx.x + RV_THREADS * k;
        if (e < TH_N) { mine[e] = V[k]; mine[TH_N + e] = S[k]; }
      }
      if (threadIdx.x == 0) { mine[2 * TH_N] = v[0]; mine[2 * TH_N + 1] = v[1]; }
      __syncthreads();                     // (waits for every thread's stores; thread 0's release then publishes them)
      if (threadIdx.x == 0) arrive_and_wait();
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      float s0 = 0.f, s1 = 0.f;
      V[0] = V[1] = S[0] = S[1] = 0.f;
      for (int w = 0; w < X.nwg; ++w) {
        const float* o = slot + (size_t)w * XB_VALS;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int e = (int)threadIdx.x + RV_THREADS * k;
          if (e < TH_N) { V[k] += o[e]; S[k] += o[TH_N + e]; }
        }
        s0 += o[2 * TH_N]; s1 += o[2 * TH_N + 1];
      }
      v[0] = s0; v[1] = s1;
      X.seq += 1;
    }
    st[0] = S[0]; st[1] = S[1];
    if (mode >= 0) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = (int)threadIdx.x + RV_THREADS * k;
        if (e < TH_N) {
          const float G = gt[k];
          const float sc = mode == 1 ? atol + rtol * fmaxf(fabsf(G), fabsf(G + S[k])) : atol + rtol * fabsf(G);
          const float q = (mode == 2 ? G : V[k]) / sc;
          const int comp = e < TH_B1 ? 0 : (e < TH_W2 ? 1 : (e < TH_B2 ? 2 : 3));
          v[2 + comp] += q * q;
        }
      }
      // workgroup sums of the four theta components (every workgroup holds the same batch totals: no second exchange)
      float w4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w4[k] = wave_sum(v[2 + k]);
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) R->red[par][wave][k] = w4[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < RV_WAVES; ++w) s += R->red[par][w][k];
        v[2 + k] = s;
      }
      par ^= 1;
    }
    float m = fmaxf(v[0] * inv_ya, v[1] * inv_ya);
    m = fmaxf(m, fmaxf(v[2] * (1.f / 256.f), v[4] * (1.f / 256.f)));      // W1, W2: 256 elements each
    m = fmaxf(m, fmaxf(v[3] * (1.f / 16.f), v[5] * (1.f / 16.f)));        // b1, b2
    return sqrtf(m);
  }

  // batch total of the per-wave images in slot 0, added to this thread's elements of the accepted theta state
  __device__ __forceinline__ void commit_image_total() {
    __syncthreads();
    float V[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = (int)threadIdx.x + RV_THREADS * k;
      if (e < TH_N) V[k] = img_total(0, e);
    }
    if (X.nwg > 1) {
      float* slot = xslots + (size_t)(X.seq & 1) * X.nwg * XB_VALS;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = (int)threadIdx.x + RV_THREADS * k;
        if (e < TH_N) slot[(size_t)X.wg * XB_VALS + e] = V[k];
      }
      __syncthreads();
      if (threadIdx.x == 0) arrive_and_wait();
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      V[0] = V[1] = 0.f;
      for (int w = 0; w < X.nwg; ++w) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int e = (int)threadIdx.x + RV_THREADS * k;
          if (e < TH_N) V[k] += slot[(size_t)w * XB_VALS + e];
        }
      }
      X.seq += 1;
    }
    gt[0] += V[0]; gt[1] += V[1];
    __syncthreads();                       // the images may be rewritten
  }

  // One adjoint solve: integrate (y, a, G) from tau0 to tau1 (tau = -t increasing).  On return a0 holds a at tau1 (4th-order
  // interpolant of the last accepted step) and gt[] this thread's two elements of the BATCH total of G(tau1) (the
  // per-wave shares never leave the trial step that produced them).  steps: running trial count (returned negative when the
  // solve stalled).
  // tau0 -> tau1: the interval in reversed time; fresh: the theta state starts at zero (a new adjoint call), else it is carried
  // over from the previous output interval of the same call (gt[] holds its batch total) and the solver restarts.
  __device__ void solve(float y0, float& a0, double tau0, double tau1, int& steps, bool fresh) {
    if (fresh) gt[0] = gt[1] = 0.f;
    float k1y, k1a, h1, du1;
    eval(y0, a0, k1y, k1a, h1, du1);
    double dtd;
    {   // _select_initial_step (order 4) under the mixed norm
      const float scy = atol + fabsf(y0) * rtol, sca = atol + fabsf(a0) * rtol;
      const float d0 = mixed(y0 / scy, a0 / sca, fresh ? -1 : 2);
      Acc4 X1; acc_zero(X1);
      theta_acc(X1, 1.f, a0, h1, du1, y0);
      put_image(0, X1, 1.f);
      const float d1 = mixed(k1y / scy, k1a / sca, 0);
      const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
      float fy, fa, hb, dub;
      const float yb = fmaf(h0, k1y, y0), ab = fmaf(h0, k1a, a0);
      eval(yb, ab, fy, fa, hb, dub);
      Acc4 Xd; acc_zero(Xd);
      theta_acc(Xd, 1.f, ab, hb, dub, yb);
      theta_acc(Xd, -1.f, a0, h1, du1, y0);
      put_image(0, Xd, 1.f);
      const float d2 = mixed((fy - k1y) / scy, (fa - k1a) / sca, 0) / h0;
      const float hh = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : fast_pow(0.01f / fmaxf(d1, d2), 0.2f);
      dtd = (double)fminf(100.f * h0, hh);
    }
    double tcur = tau0;
    const int step_limit = steps + RV_MAX_TRIALS;
    for (;;) {
      if (R->dead || steps >= step_limit || !(tcur + dtd > tcur)) { steps = -steps - 1; return; }
      const float dt = (float)dtd;
      Acc4 S, E, M;
      acc_zero(S); acc_zero(E); acc_zero(M);
      theta_acc(S, B1, a0, h1, du1, y0); theta_acc(E, E1, a0, h1, du1, y0); theta_acc(M, C1, a0, h1, du1, y0);
      float h, du, ys, as;
      ys = fmaf(dt * A21, k1y, y0); as = fmaf(dt * A21, k1a, a0);
      float k2y, k2a; eval(ys, as, k2y, k2a, h, du);                        // stage 2 has zero weight in S, E, M
      ys = fmaf(dt, A31 * k1y + A32 * k2y, y0); as = fmaf(dt, A31 * k1a + A32 * k2a, a0);
      float k3y, k3a; eval(ys, as, k3y, k3a, h, du);
      theta_acc(S, B3, as, h, du, ys); theta_acc(E, E3, as, h, du, ys); theta_acc(M, C3, as, h, du, ys);
      ys = fmaf(dt, A41 * k1y + A42 * k2y + A43 * k3y, y0); as = fmaf(dt, A41 * k1a + A42 * k2a + A43 * k3a, a0);
      float k4y, k4a; eval(ys, as, k4y, k4a, h, du);
      theta_acc(S, B4, as, h, du, ys); theta_acc(E, E4, as, h, du, ys); theta_acc(M, C4, as, h, du, ys);
      ys = fmaf(dt, A51 * k1y + A52 * k2y + A53 * k3y + A54 * k4y, y0); as = fmaf(dt, A51 * k1a + A52 * k2a + A53 * k3a + A54 * k4a, a0);
      float k5y, k5a; eval(ys, as, k5y, k5a, h, du);
      theta_acc(S, B5, as, h, du, ys); theta_acc(E, E5, as, h, du, ys); theta_acc(M, C5, as, h, du, ys);
      ys = fmaf(dt, A61 * k1y + A62 * k2y + A63 * k3y + A64 * k4y + A65 * k5y, y0);
      as = fmaf(dt, A61 * k1a + A62 * k2a + A63 * k3a + A64 * k4a + A65 * k5a, a0);
      float k6y, k6a; eval(ys, as, k6y, k6a, h, du);
      theta_acc(S, B6, as, h, du, ys); theta_acc(E, E6, as, h, du, ys); theta_acc(M, C6, as, h, du, ys);
      const float y1 = fmaf(dt, B1 * k1y + B3 * k3y + B4 * k4y + B5 * k5y + B6 * k6y, y0);
      const float a1 = fmaf(dt, B1 * k1a + B3 * k3a + B4 * k4a + B5 * k5a + B6 * k6a, a0);
      float k7y, k7a, h7, du7; eval(y1, a1, k7y, k7a, h7, du7);
      theta_acc(E, E7, a1, h7, du7, y1); theta_acc(M, C7, a1, h7, du7, y1);
      const float erry = dt * (E1 * k1y + E3 * k3y + E4 * k4y + E5 * k5y + E6 * k6y + E7 * k7y);
      const float erra = dt * (E1 * k1a + E3 * k3a + E4 * k4a + E5 * k5a + E6 * k6a + E7 * k7a);
      put_image(0, E, dt);
      put_image(1, S, dt);
      const float toly = atol + rtol * fmaxf(fabsf(y0), fabsf(y1)), tola = atol + rtol * fmaxf(fabsf(a0), fabsf(a1));
      const float ratio = mixed(erry / toly, erra / tola, 1);
      ++steps;
      if (ratio <= 1.f) {
        if (tcur + dtd >= tau1) {
          // last step: the value at tau1 off the 4th-order interpolant through (z0, z_mid, z1, f0, f1); the abscissa is
          // formed from the fp32-rounded times, as torchdiffeq does.  z(x) = z0 + wS (z1 - z0) + wM (z_mid - z0)
          // + dt (w1 f0 + w7 f1); for the (linear) theta components the same weights apply to S, M, X1, X7.
          const float x = ((float)tau1 - (float)tcur) / ((float)(tcur + dtd) - (float)tcur);
          float wS, wM, w1c, w7c;
          interp_weights(x, wS, wM, w1c, w7c);
          const float dmid = C1 * k1a + C3 * k3a + C4 * k4a + C5 * k5a + C6 * k6a + C7 * k7a;
          const float aend = a0 + wS * (a1 - a0) + dt * (wM * dmid + w1c * k1a + w7c * k7a);
          Acc4 dG;
          dG.W1 = wS * S.W1 + wM * M.W1; dG.W2 = wS * S.W2 + wM * M.W2;
          dG.b1 = wS * S.b1 + wM * M.b1; dG.b2 = wS * S.b2 + wM * M.b2;
          theta_acc(dG, w1c, a0, h1, du1, y0);
          theta_acc(dG, w7c, a1, h7, du7, y1);
          put_image(0, dG, dt);
          commit_image_total();
          a0 = aend;
          return;
        }
        // accepted, not last: G += dt S -- its batch total is in st[] from the norm pass
        gt[0] += st[0]; gt[1] += st[1];
        tcur += dtd; y0 = y1; a0 = a1; k1y = k7y; k1a = k7a; h1 = h7; du1 = du7;
      }
      dtd *= (double)step_factor(ratio);
    }
  }
};

struct RnnBwdJobs { gode_odernn_bwd_op op[RV_MAX_JOBS]; int32_t nblk[RV_MAX_JOBS]; int32_t count; int32_t direct; };

__global__ void __launch_bounds__(RV_THREADS) odernn_bwd_valu_kernel(const RnnBwdJobs J) {
  __shared__ float P[TH_N + 96];       // W1, b1, W2, b2 (gradient-vector order), bih, bhh
  __shared__ float GP[GP_N];           // Wih, Whh with padded rows
  __shared__ AdjLdsV lds;
  __shared__ RedLds R;
  int job = 0, wg = blockIdx.x;
  while (job + 1 < J.count && wg >= J.nblk[job]) { wg -= J.nblk[job]; ++job; }
  const gode_odernn_bwd_op a = J.op[job];
  AdjV S;
  S.lane = threadIdx.x & 63; S.wave = threadIdx.x >> 6; S.i = S.lane & 15; S.tl = S.lane >> 4;
  S.L = &lds; S.R = &R; S.par = 0;
  const int i = S.i, tl = S.tl, wave = S.wave;
  const int n0 = wg * RV_TRAJ, n = n0 + wave * 4 + tl;
  const bool valid = n < a.N;
  S.valid = valid;
  const int T = a.T;
  S.inv_ya = 1.f / ((float)a.N * 16.f);
  S.rtol = a.rtol; S.atol = a.atol;
  S.X.nwg = J.nblk[job]; S.X.wg = wg; S.X.seq = 0; S.X.counter = a.sync;
  S.xslots = a.sync ? reinterpret_cast<float*>(a.sync) + XS_HDR : nullptr;
  S.st[0] = S.st[1] = 0.f;
  if (threadIdx.x == 0) R.dead = 0;
  {
    const int t = threadIdx.x;
    if (t < 256) { P[TH_W1 + t] = a.p.W1[t]; P[TH_W2 + t] = a.p.W2[t]; }
    if (t < 16) { P[TH_B1 + t] = a.p.b1[t]; P[TH_B2 + t] = a.p.b2[t]; }
    if (t < 48) { P[TH_N + t] = a.p.bih[t]; P[TH_N + 48 + t] = a.p.bhh[t]; }
    for (int k = t; k < 768; k += RV_THREADS) {
      const int row = k >> 4, col = k & 15;
      GP[GP_WIH + row * GP_LD + col] = a.p.Wih[k];
      GP[GP_WHH + row * GP_LD + col] = a.p.Whh[k];
    }
    for (int k = t; k < 3 * TH_N; k += RV_THREADS) (&lds.gru[0][0])[k] = 0.f;
  }
  __syncthreads();
  {
    int src[16];
    probe_sources(i, src);
    load_rows(S.w1, P + TH_W1, 16, 0, 0, i, src);
    load_rows(S.w2, P + TH_W2, 16, 0, 0, i, src);
    load_cols(S.w1t, P + TH_W1, 16, 0, 0, i, src);
    load_cols(S.w2t, P + TH_W2, 16, 0, 0, i, src);
  }
  S.b1 = P[TH_B1 + i]; S.b2 = P[TH_B2 + i];

  float gth[2] = {0.f, 0.f};           // this thread's two elements of the ODEFunc gradient (batch total over all frames)
  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  float carry = 0.f;
  // the three per-frame inputs (upstream gradient, GRU input, post-ODE state) are requested one frame ahead: dependent
  // HBM round trips otherwise, ~2 us per frame on the critical path
  auto load_up = [&](int t) -> float {
    if (!valid || t < 0) return 0.f;
    if (a.sel_t == nullptr) return a.gz[((int64_t)n * T + t) * a.zcols + i];
    return t == tsel ? a.gz[(int64_t)n * a.zcols + i] : 0.f;
  };
  float up_n = load_up(T - 1);
  float e_n = valid ? a.noise[((int64_t)T * a.N + n) * 16 + i] : 0.f;
  float hp_n = valid ? a.hp[((int64_t)n * T + (T - 1)) * 16 + i] : 0.f;
  for (int t = T - 1; t >= 0; --t) {
    const float up = up_n, e = e_n, hp = hp_n;
    if (t > 0) {
      up_n = load_up(t - 1);
      e_n = valid ? a.noise[((int64_t)t * a.N + n) * 16 + i] : 0.f;
      hp_n = valid ? a.hp[((int64_t)n * T + (t - 1)) * 16 + i] : 0.f;
    }
    const float gh = carry + up;                                                  // dL/dh_{t+1}
    float adj;
    {   // GRU recompute + backward; the weight arrangement is gathered from LDS per frame (it would not fit the register file
        // next to the solver's four matrices)
      int src[16];
      probe_sources(i, src);
      float gi[3], hh[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        W16 w;
        load_rows(w, GP + GP_WIH, GP_LD, 16 * q, 0, i, src);
        gi[q] = mv16c(w, e, P[TH_N + 16 * q + i]);
        load_rows(w, GP + GP_WHH, GP_LD, 16 * q, 0, i, src);
        hh[q] = mv16c(w, hp, P[TH_N + 48 + 16 * q + i]);
      }
      const float r = fast_sigmoid(gi[0] + hh[0]);
      const float zg = fast_sigmoid(gi[1] + hh[1]);
      const float hn = hh[2];
      const float nn = fast_tanh(gi[2] + r * hn);
      const float dn_pre = gh * (1.f - zg) * (1.f - nn * nn);
      const float dz_pre = gh * (hp - nn) * zg * (1.f - zg);
      const float dr_pre = dn_pre * hn * r * (1.f - r);
      const float dgi[3] = {dr_pre, dz_pre, dn_pre};
      const float dgh[3] = {dr_pre, dz_pre, dn_pre * r};
      adj = gh * zg;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        W16 wt;
        load_cols(wt, GP + GP_WHH, GP_LD, 16 * q, 0, i, src);      // (Whh_q)^T
        adj = mv16c(wt, dgh[q], adj);
        // this frame's parameter-gradient terms of gate q: the wave's outer-product sums (one MFMA each: the per-lane values
        // are the A / B operands for k = the wave's 4 trajectories) meet in LDS and are added to the thread-owned totals
        const f32x4 pWih = MFMA16(dgi[q], e, zero4()), pWhh = MFMA16(dgh[q], hp, zero4());
        float* im = &lds.img[wave][0][0];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          im[(4 * tl + rr) * 16 + i] = pWih[rr];
          im[256 + (4 * tl + rr) * 16 + i] = pWhh[rr];
        }
        const float sbi = traj_sum(dgi[q]), sbh = traj_sum(dgh[q]);
        if (tl == 0) { im[512 + i] = sbi; im[528 + i] = sbh; }
        __syncthreads();
        for (int k = threadIdx.x; k < TH_N; k += RV_THREADS) lds.gru[q][k] += S.img_total(0, k);
        __syncthreads();
      }
    }
    // adjoint call of this frame's solve: state (y = h', a = adj, g_theta = 0) at t = 1 back to t = 0
    int steps = 0;
    S.solve(hp, adj, -1.0, 0.0, steps, true);
    if (a.nsteps && wg == 0 && threadIdx.x == 0) a.nsteps[t] = steps;
    if (steps < 0) break;                // stalled: uniform over the call
    gth[0] += S.gt[0]; gth[1] += S.gt[1];
    carry = adj;                                                                 // dL/dh_t
  }
  // ---- output: a launch of ONE workgroup writes (or adds to) grads directly; otherwise every workgroup writes one row of
  // its op's `work` (the ODEFunc block holds the BATCH total in every workgroup of an op: row 0 carries it, the other rows
  // carry zeros) and odernn_bwd_rows_kernel adds the rows in fixed order
  float* out = J.direct ? a.grads : a.work + (int64_t)wg * RO_N;
  const bool direct_acc = J.direct && a.accumulate;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = (int)threadIdx.x + RV_THREADS * k;
    if (e < TH_N) {
      const float v = wg == 0 ? gth[k] : 0.f;
      out[e] = direct_acc ? out[e] + v : v;
    }
  }
  for (int q = 0; q < 3; ++q) {
    for (int k = threadIdx.x; k < TH_N; k += RV_THREADS) {
      const float v = lds.gru[q][k];
      const int dst = k < 256 ? RO_WIH + 256 * q + k : (k < 512 ? RO_WHH + 256 * q + (k - 256) : (k < 528 ? RO_BIH + 16 * q + (k - 512) : RO_BHH + 16 * q + (k - 528)));
      out[dst] = direct_acc ? out[dst] + v : v;
    }
  }
}

// rows of `work` (one per workgroup; the ops that share a grads pointer, in array order) -> grads, fixed order
struct RowsArgs { const float* work[RV_MAX_JOBS]; int32_t rows[RV_MAX_JOBS]; int32_t count, accumulate; float* grads; };
__global__ void __launch_bounds__(256) odernn_bwd_rows_kernel(const RowsArgs A) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= RO_N) return;
  float s = 0.f;
  for (int j = 0; j < A.count; ++j)
    for (int b = 0; b < A.rows[j]; ++b) s += A.work[j][(int64_t)b * RO_N + k];
  A.grads[k] = A.accumulate ? A.grads[k] + s : s;
}

int gode_launch_odernn_bwd_mfma(const gode_odernn_bwd_op* op, hipStream_t st);   // odernn.hip: fixed substeps / fallback

static bool rnn_bwd_args_ok(const gode_odernn_bwd_op* op) {
  if (!op || !op->noise || !op->hp || !op->gz || !op->work || !op->grads || op->N <= 0 || op->T < 1 || op->substeps < 0) return false;
  if (op->substeps == 0 && (!(op->rtol > 0.f) || !(op->atol >= 0.f))) return false;
  if (op->zcols < 16 || op->zcols % 4 != 0) return false;
  if (!op->p.W1 || !op->p.b1 || !op->p.W2 || !op->p.b2 || !op->p.Wih || !op->p.Whh || !op->p.bih || !op->p.bhh) return false;
  if (op->substeps == 0 && op->N > RV_TRAJ && op->N <= GODE_ODERNN_SYNC_MAX_N && !op->sync) return false;
  return true;
}

extern "C" int64_t gode_odernn_bwd_work_size(int32_t N) { return (int64_t)((N + 15) / 16) * RO_N; }

static int rnn_bwd_launch(const gode_odernn_bwd_op* ops, int count, hipStream_t st) {
  RnnBwdJobs J;
  J.count = count;
  int grid = 0;
  for (int k = 0; k < count; ++k) {
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    if (nblk > 1) {
      const hipError_t e = hipMemsetAsync(ops[k].sync, 0, XS_HDR * sizeof(int32_t), st);
      if (e != hipSuccess) return (int)e;
    }
    J.op[k] = ops[k]; J.nblk[k] = nblk;
    grid += nblk;
  }
  J.direct = (count == 1 && grid == 1) ? 1 : 0;
  hipLaunchKernelGGL(odernn_bwd_valu_kernel, dim3(grid), dim3(RV_THREADS), 0, st, J);
  GODE_LAUNCH_CHECK();
  if (J.direct) return 0;
  bool done[RV_MAX_JOBS] = {false};
  for (int k = 0; k < count; ++k) {
    if (done[k]) continue;
    RowsArgs A;
    A.count = 0; A.grads = ops[k].grads; A.accumulate = ops[k].accumulate;     // later ops of the group add by construction
    for (int j = k; j < count; ++j) {
      if (!done[j] && ops[j].grads == ops[k].grads) { A.work[A.count] = ops[j].work; A.rows[A.count] = J.nblk[j]; A.count += 1; done[j] = true; }
    }
    hipLaunchKernelGGL(odernn_bwd_rows_kernel, dim3((RO_N + 255) / 256), dim3(256), 0, st, A);
    GODE_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int gode_odernn_bwd_multi(const gode_odernn_bwd_op* ops, int32_t count, void* stream) {
  if (!ops || count < 1 || count > RV_MAX_JOBS) return GODE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  int spinning = 0;
  bool fallback = false;
  for (int k = 0; k < count; ++k) {
    if (!rnn_bwd_args_ok(&ops[k])) return GODE_E_ARG;
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    if (ops[k].substeps > 0) fallback = true;
    else if (nblk > 1) spinning += nblk;
  }
  if (spinning > RV_MAX_WGS) fallback = true;
  if (!fallback) return rnn_bwd_launch(ops, count, st);
  // fixed Kutta-3/8 substeps, or more spinning workgroups than are guaranteed co-resident: the ops run one after the other
  // in array order (each with its own accumulate flag), the affected ones on the MFMA kernels of odernn.hip
  for (int k = 0; k < count; ++k) {
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    const bool mfma = ops[k].substeps > 0 || (nblk > 1 && spinning > RV_MAX_WGS);
    const int rc = mfma ? gode_launch_odernn_bwd_mfma(&ops[k], st) : rnn_bwd_launch(&ops[k], 1, st);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int gode_odernn_bwd(const gode_odernn_bwd_op* op, void* stream) { return gode_odernn_bwd_multi(op, 1, stream); }

// =================================================================================================================
// ode_method = "dopri5" on the plain Neural-ODE generators (gode_ode_fwd_op / gode_ode_bwd_op with method == 1; what
// BASELINE configs[3] words for UCF -- the reference's own call is rk4): pre-net, then ONE adaptive solve over the T output
// times, every output read off the 4th-order interpolant of the accepted step that covers it (RKAdaptiveStepsizeODESolver:
// advance while target > t1, then evaluate); the adjoint is ONE call over the T - 1 intervals with the theta state
// carried and the solver restarted per interval (oracle/ode_ref.py:_Adjoint.backward).  Same mapping, fused algebra,
// whole-batch norm and multi-solve launches as the ODE-RNN kernels above; the round-2 MFMA kernels (odernn.hip,
// adj_adaptive.hip) remain above GODE_ODERNN_SYNC_MAX_N trajectories per launch.
// =================================================================================================================
#define DP_WA 0
#define DP_BA 1024
#define DP_WB 1088
#define DP_BB 2112
#define DP_W1 2128
#define DP_B1 2384
#define DP_W2 2400
#define DP_B2 2656
#define DP_M GODE_ODE_NPARAM
#define DP_C (GODE_ODE_NPARAM + 256)
#define DP_N (GODE_ODE_NPARAM + 272)

__device__ __forceinline__ void stage_ode_params(const gode_ode_params& p, int prenet, float* P) {
  const int t = threadIdx.x;
  if (prenet) {
    for (int k = t; k < 1024; k += RV_THREADS) { P[DP_WA + k] = p.Wa[k]; P[DP_WB + k] = p.Wb[k]; }
    if (t < 64) P[DP_BA + t] = p.ba[t];
    if (t < 16) P[DP_BB + t] = p.bb[t];
  }
  for (int k = t; k < 256; k += RV_THREADS) { P[DP_W1 + k] = p.W1[k]; P[DP_W2 + k] = p.W2[k]; }
  if (t < 16) { P[DP_B1 + t] = p.b1[t]; P[DP_B2 + t] = p.b2[t]; }
  __syncthreads();
}

struct OdeFwdJobs { gode_ode_fwd_op op[RV_MAX_JOBS]; int32_t nblk[RV_MAX_JOBS]; int32_t count; };

__global__ void __launch_bounds__(RV_THREADS) ode_dopri5_fwd_valu_kernel(const OdeFwdJobs J) {
  __shared__ float P[DP_N];
  __shared__ RedLds R;
  int job = 0, wg = blockIdx.x;
  while (job + 1 < J.count && wg >= J.nblk[job]) { wg -= J.nblk[job]; ++job; }
  const gode_ode_fwd_op a = J.op[job];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, tl = lane >> 4;
  const int n0 = wg * RV_TRAJ, n = n0 + wave * 4 + tl;
  const bool valid = n < a.N;
  const int T = a.T;
  const float inv_count = 1.f / ((float)a.N * 16.f);
  const float rtol = a.rtol, atol = a.atol;
  XSync X;
  X.nwg = J.nblk[job]; X.wg = wg; X.seq = 0;
  X.counter = a.sync; X.slots = a.sync ? reinterpret_cast<float*>(a.sync) + XS_HDR : nullptr;
  int par = 0;
  if (threadIdx.x == 0) R.dead = 0;

  float y0 = valid ? a.x[(int64_t)n * 16 + i] : 0.f;
  stage_ode_params(a.p, a.prenet, P);
  if (threadIdx.x < 256) {           // M = W1 W2, c = W1 b2 (fp64 accumulation)
    const int r = threadIdx.x >> 4, j = threadIdx.x & 15;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += (double)P[DP_W1 + r * 16 + k] * (double)P[DP_W2 + k * 16 + j];
    P[DP_M + threadIdx.x] = (float)s;
  } else if (threadIdx.x < 272) {
    const int r = threadIdx.x - 256;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += (double)P[DP_W1 + r * 16 + k] * (double)P[DP_B2 + k];
    P[DP_C + r] = (float)s;
  }
  __syncthreads();
  if (a.content) rnn_broadcast_content(a.content, a.z, a.sel_t, a.N, T, a.zcols, n0);      // stores only

  int src[16];
  probe_sources(i, src);
  if (a.prenet) {
    // Linear(16,64) -> LReLU -> Linear(64,16) -> LReLU: the 64 hidden units as four row-resident chunks
    float acc = P[DP_BB + i];
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
      W16 wa, wb;
      load_rows(wa, P + DP_WA, 16, 16 * m, 0, i, src);
      load_rows(wb, P + DP_WB, 64, 0, 16 * m, i, src);
      const float hh = lrelu1(mv16c(wa, y0, P[DP_BA + 16 * m + i]));
      acc = mv16c(wb, hh, acc);
    }
    y0 = lrelu1(acc);
  }
  W16 Mw, w1, w2;
  load_rows(Mw, P + DP_M, 16, 0, 0, i, src);
  load_rows(w1, P + DP_W1, 16, 0, 0, i, src);
  load_rows(w2, P + DP_W2, 16, 0, 0, i, src);
  const float b1 = P[DP_B1 + i], b2 = P[DP_B2 + i], cq = P[DP_C + i];
  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto emit = [&](int t, float v) {
    if (!valid) return;
    if (a.traj) a.traj[((int64_t)n * T + t) * 16 + i] = v;
    if (a.sel_t == nullptr) a.z[((int64_t)n * T + t) * a.zcols + i] = v;
    else if (t == tsel) a.z[(int64_t)n * a.zcols + i] = v;
  };
  emit(0, y0);

  float u0 = mv16c(w1, y0, b1);
  float h1 = fast_tanh(u0);
  float q1 = mv16c(Mw, h1, cq);
  double dtd;
  {   // torchdiffeq _select_initial_step (order 4)
    const float f0 = mv16c(w2, h1, b2);
    const float sc = atol + fabsf(y0) * rtol;
    float v[2] = {valid ? (y0 / sc) * (y0 / sc) : 0.f, valid ? (f0 / sc) * (f0 / sc) : 0.f};
    batch_sum<2>(v, &R, par, X);
    const float d0 = sqrtf(v[0] * inv_count), d1 = sqrtf(v[1] * inv_count);
    const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    const float df = mv16c(w2, fast_tanh(fmaf(h0, q1, u0)) - h1, 0.f);
    float w[1] = {valid ? (df / sc) * (df / sc) : 0.f};
    batch_sum<1>(w, &R, par, X);
    const float d2 = sqrtf(w[0] * inv_count) / h0;
    const float hh = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : fast_pow(0.01f / fmaxf(d1, d2), 0.2f);
    dtd = (double)fminf(100.f * h0, hh);
  }
  double tcur = (double)a.tout[0], seg0 = tcur, seg1 = tcur;
  // the interpolant of the last accepted step, in the linear-in-h form: y(x) = sy + sdt (W2 (wS sHb + wM sHm + w1 sh1 + w7 sh7) + b2 (...))
  float sy = y0, sdt = 0.f, sHb = 0.f, sHm = 0.f, sh1 = 0.f, sh7 = 0.f;
  int steps = 0;
  bool stalled = false;
  for (int j = 1; j < T && !stalled; ++j) {
    const double target = (double)a.tout[j];
    while (target > seg1) {
      if (R.dead || steps >= RV_MAX_TRIALS || !(tcur + dtd > tcur)) { stalled = true; break; }
      const float dt = (float)dtd;
      const float h2 = fast_tanh(fmaf(dt * A21, q1, u0));
      const float q2 = mv16c(Mw, h2, cq);
      const float h3 = fast_tanh(fmaf(dt, A31 * q1 + A32 * q2, u0));
      const float q3 = mv16c(Mw, h3, cq);
      const float h4 = fast_tanh(fmaf(dt, A41 * q1 + A42 * q2 + A43 * q3, u0));
      const float q4 = mv16c(Mw, h4, cq);
      const float h5 = fast_tanh(fmaf(dt, A51 * q1 + A52 * q2 + A53 * q3 + A54 * q4, u0));
      const float q5 = mv16c(Mw, h5, cq);
      const float h6 = fast_tanh(fmaf(dt, A61 * q1 + A62 * q2 + A63 * q3 + A64 * q4 + A65 * q5, u0));
      const float q6 = mv16c(Mw, h6, cq);
      const float u7 = fmaf(dt, B1 * q1 + B3 * q3 + B4 * q4 + B5 * q5 + B6 * q6, u0);
      const float h7 = fast_tanh(u7);
      const float Hb = B1 * h1 + B3 * h3 + B4 * h4 + B5 * h5 + B6 * h6;
      const float He = E1 * h1 + E3 * h3 + E4 * h4 + E5 * h5 + E6 * h6 + E7 * h7;
      const float y1 = fmaf(dt, mv16c(w2, Hb, b2), y0);
      const float err = dt * mv16c(w2, He, 0.f);
      const float tol = atol + rtol * fmaxf(fabsf(y0), fabsf(y1));
      const float rr = err / tol;
      float v[1] = {valid ? rr * rr : 0.f};
      const float q7 = mv16c(Mw, h7, cq);
      batch_sum<1>(v, &R, par, X);
      const float ratio = sqrtf(v[0] * inv_count);
      ++steps;
      if (ratio <= 1.f) {
        sy = y0; sdt = dt; sHb = Hb; sh1 = h1; sh7 = h7;
        sHm = C1 * h1 + C3 * h3 + C4 * h4 + C5 * h5 + C6 * h6 + C7 * h7;
        seg0 = tcur; seg1 = tcur + dtd;
        tcur = seg1; y0 = y1; u0 = u7; h1 = h7; q1 = q7;
      }
      dtd *= (double)step_factor(ratio);
    }
    if (stalled) break;
    // (the abscissa is formed from the fp32-rounded times, as torchdiffeq does)
    const float x = ((float)target - (float)seg0) / ((float)seg1 - (float)seg0);
    float wS, wM, w1c, w7c;
    interp_weights(x, wS, wM, w1c, w7c);
    const float Hx = wS * sHb + wM * sHm + w1c * sh1 + w7c * sh7;
    emit(j, fmaf(sdt, mv16c(w2, Hx, b2 * (wS + 0.5f * wM + w1c + w7c)), sy));
  }
  if (a.nsteps && wg == 0 && threadIdx.x == 0) a.nsteps[0] = stalled ? -steps - 1 : steps;
}

struct OdeBwdJobs { gode_ode_bwd_op op[RV_MAX_JOBS]; int32_t nblk[RV_MAX_JOBS]; int32_t count; int32_t direct; };

__global__ void __launch_bounds__(RV_THREADS) ode_dopri5_bwd_valu_kernel(const OdeBwdJobs J) {
  __shared__ float P[GODE_ODE_NPARAM];
  __shared__ AdjLdsV lds;
  __shared__ RedLds R;
  int job = 0, wg = blockIdx.x;
  while (job + 1 < J.count && wg >= J.nblk[job]) { wg -= J.nblk[job]; ++job; }
  const gode_ode_bwd_op a = J.op[job];
  AdjV S;
  S.lane = threadIdx.x & 63; S.wave = threadIdx.x >> 6; S.i = S.lane & 15; S.tl = S.lane >> 4;
  S.L = &lds; S.R = &R; S.par = 0;
  const int i = S.i, tl = S.tl, wave = S.wave;
  const int n0 = wg * RV_TRAJ, n = n0 + wave * 4 + tl;
  const bool valid = n < a.N;
  S.valid = valid;
  const int T = a.T;
  S.inv_ya = 1.f / ((float)a.N * 16.f);
  S.rtol = a.rtol; S.atol = a.atol;
  S.X.nwg = J.nblk[job]; S.X.wg = wg; S.X.seq = 0; S.X.counter = a.sync;
  S.xslots = a.sync ? reinterpret_cast<float*>(a.sync) + XS_HDR : nullptr;
  S.st[0] = S.st[1] = 0.f; S.gt[0] = S.gt[1] = 0.f;
  if (threadIdx.x == 0) R.dead = 0;
  const float x = (valid && a.prenet) ? a.x[(int64_t)n * 16 + i] : 0.f;      // needed by the pre-net backward at the very end
  stage_ode_params(a.p, a.prenet, P);
  int src[16];
  probe_sources(i, src);
  load_rows(S.w1, P + DP_W1, 16, 0, 0, i, src);
  load_rows(S.w2, P + DP_W2, 16, 0, 0, i, src);
  load_cols(S.w1t, P + DP_W1, 16, 0, 0, i, src);
  load_cols(S.w2t, P + DP_W2, 16, 0, 0, i, src);
  S.b1 = P[DP_B1 + i]; S.b2 = P[DP_B2 + i];

  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto upstream = [&](int t) -> float {
    if (!valid) return 0.f;
    if (a.sel_t == nullptr) return a.gz[((int64_t)n * T + t) * a.zcols + i];
    return t == tsel ? a.gz[(int64_t)n * a.zcols + i] : 0.f;
  };
  float adj = upstream(T - 1);
  int steps = 0;
  for (int it = T - 1; it >= 1; --it) {
    const float y = valid ? a.traj[((int64_t)n * T + it) * 16 + i] : 0.f;
    S.solve(y, adj, -(double)a.tout[it], -(double)a.tout[it - 1], steps, it == T - 1);
    if (steps < 0) break;
    adj = adj + upstream(it - 1);
  }
  if (a.nsteps && wg == 0 && threadIdx.x == 0) a.nsteps[0] = steps;
  // ---- output row: [Wa 1024 | ba 64 | Wb 1024 | bb 16 | W1 256 | b1 16 | W2 256 | b2 16].  One launch of ONE workgroup writes (or
  // adds to) grads directly; otherwise one row of `work` per workgroup (the ODEFunc block holds the BATCH total in every
  // workgroup: row 0 carries it, the others zeros), added by ode_rows_kernel
  float* out = J.direct ? a.grads : a.work + (int64_t)wg * GODE_ODE_NPARAM;
  const bool acc_out = J.direct && a.accumulate;
  auto put = [&](int dst, float v) { out[dst] = acc_out ? out[dst] + v : v; };
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = (int)threadIdx.x + RV_THREADS * k;
    if (e < TH_N) put(DP_W1 + e, (wg == 0 && steps >= 0) ? S.gt[k] : 0.f);     // TH_* order == W1, b1, W2, b2
  }
  if (!a.prenet) {
    for (int k = threadIdx.x; k < DP_W1; k += RV_THREADS) put(k, 0.f);
    return;
  }
  // ---- pre-net backward; adj = dL/d(pre-net output).  With one (trajectory, feature) value per lane the batch sums of
  // outer products are one MFMA per 16x16 chunk (k = the wave's 4 trajectories); the waves' images meet in LDS per chunk.
  float hpre[4];
  float acc = P[DP_BB + i];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    W16 wa, wb;
    load_rows(wa, P + DP_WA, 16, 16 * m, 0, i, src);
    load_rows(wb, P + DP_WB, 64, 0, 16 * m, i, src);
    hpre[m] = mv16c(wa, x, P[DP_BA + 16 * m + i]);
    acc = mv16c(wb, lrelu1(hpre[m]), acc);
  }
  const float g0 = valid ? (acc > 0.f ? adj : 0.2f * adj) : 0.f;
#pragma unroll 1
  for (int m = 0; m < 4; ++m) {
    W16 wbt;
    load_cols(wbt, P + DP_WB, 64, 0, 16 * m, i, src);      // (Wb chunk)^T
    const float tt = mv16c(wbt, g0, 0.f);
    const float gh = valid ? (hpre[m] > 0.f ? tt : 0.2f * tt) : 0.f;
    const f32x4 dWb = MFMA16(g0, lrelu1(hpre[m]), zero4());   // [i][j] = sum_n g0_i lrelu(hpre_m)_j -> Wb[i][16m + j]
    const f32x4 dWa = MFMA16(gh, x, zero4());                 // [i][j] = sum_n gh_i x_j              -> Wa[16m + i][j]
    float* im = &lds.img[wave][0][0];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      im[(4 * tl + r) * 16 + i] = dWb[r];
      im[256 + (4 * tl + r) * 16 + i] = dWa[r];
    }
    const float sba = traj_sum(gh), sbb = traj_sum(g0);
    if (tl == 0) { im[512 + i] = sba; im[528 + i] = sbb; }
    __syncthreads();
    for (int k = threadIdx.x; k < TH_N; k += RV_THREADS) {
      const float v = S.img_total(0, k);
      if (k < 256) put(DP_WB + (k >> 4) * 64 + 16 * m + (k & 15), v);
      else if (k < 512) put(DP_WA + (16 * m + ((k - 256) >> 4)) * 16 + (k & 15), v);
      else if (k < 528) put(DP_BA + 16 * m + (k - 512), v);
      else if (m == 0) put(DP_BB + (k - 528), v);
    }
  }
}

// rows of `work` (ops sharing a grads pointer, in array order) -> grads, fixed order; row length 2,672
struct OdeRowsArgs { const float* work[RV_MAX_JOBS]; int32_t rows[RV_MAX_JOBS]; int32_t count, accumulate; float* grads; };
__global__ void __launch_bounds__(256) ode_rows_kernel(const OdeRowsArgs A) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= GODE_ODE_NPARAM) return;
  float s = 0.f;
  for (int j = 0; j < A.count; ++j)
    for (int b = 0; b < A.rows[j]; ++b) s += A.work[j][(int64_t)b * GODE_ODE_NPARAM + k];
  A.grads[k] = A.accumulate ? A.grads[k] + s : s;
}

int gode_launch_ode_dopri5(const gode_ode_fwd_op* op, hipStream_t st);          // odernn.hip (MFMA fallback)
int gode_launch_ode_dopri5_bwd_mfma(const gode_ode_bwd_op* op, hipStream_t st); // ode.hip (MFMA fallback incl. its reduce)

static bool ode5_fwd_ok(const gode_ode_fwd_op* op) {
  if (!op || op->method != 1 || !op->x || !op->z || !op->tout || op->N <= 0 || op->T < 1 || !(op->rtol > 0.f) || !(op->atol >= 0.f)) return false;
  if (op->zcols < 68 || op->zcols % 4 != 0) return false;
  if (!op->p.W1 || !op->p.b1 || !op->p.W2 || !op->p.b2) return false;
  if (op->prenet && (!op->p.Wa || !op->p.ba || !op->p.Wb || !op->p.bb)) return false;
  return true;
}

// `count` (<= 8) dopri5 latent solves (method == 1 ops) in one launch (ops: HOST array)
extern "C" int gode_ode_fwd_multi(const gode_ode_fwd_op* ops, int32_t count, void* stream) {
  if (!ops || count < 1 || count > RV_MAX_JOBS) return GODE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  OdeFwdJobs J;
  J.count = 0;
  int spinning = 0;
  for (int k = 0; k < count; ++k) {
    if (!ode5_fwd_ok(&ops[k])) return GODE_E_ARG;
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    if (nblk > 1 && ops[k].sync) spinning += nblk;
  }
  for (int k = 0; k < count; ++k) {
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    if (nblk > 1 && (!ops[k].sync || spinning > RV_MAX_WGS)) {       // no exchange buffer, or too many workgroups to be co-resident
      const int rc = gode_launch_ode_dopri5(&ops[k], st);
      if (rc) return rc;
      continue;
    }
    if (nblk > 1) {
      const hipError_t e = hipMemsetAsync(ops[k].sync, 0, XS_HDR * sizeof(int32_t), st);
      if (e != hipSuccess) return (int)e;
    }
    J.op[J.count] = ops[k]; J.nblk[J.count] = nblk; J.count += 1;
  }
  if (J.count == 0) return 0;
  int grid = 0;
  for (int k = 0; k < J.count; ++k) grid += J.nblk[k];
  hipLaunchKernelGGL(ode_dopri5_fwd_valu_kernel, dim3(grid), dim3(RV_THREADS), 0, st, J);
  GODE_LAUNCH_CHECK();
  return 0;
}

static int ode5_bwd_launch(const gode_ode_bwd_op* ops, int count, hipStream_t st) {
  OdeBwdJobs J;
  J.count = count;
  int grid = 0;
  for (int k = 0; k < count; ++k) {
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    if (nblk > 1) {
      const hipError_t e = hipMemsetAsync(ops[k].sync, 0, XS_HDR * sizeof(int32_t), st);
      if (e != hipSuccess) return (int)e;
    }
    J.op[k] = ops[k]; J.nblk[k] = nblk;
    grid += nblk;
  }
  J.direct = (count == 1 && grid == 1) ? 1 : 0;
  hipLaunchKernelGGL(ode_dopri5_bwd_valu_kernel, dim3(grid), dim3(RV_THREADS), 0, st, J);
  GODE_LAUNCH_CHECK();
  if (J.direct) return 0;
  bool done[RV_MAX_JOBS] = {false};
  for (int k = 0; k < count; ++k) {
    if (done[k]) continue;
    OdeRowsArgs A;
    A.count = 0; A.grads = ops[k].grads; A.accumulate = ops[k].accumulate;
    for (int j = k; j < count; ++j) {
      if (!done[j] && ops[j].grads == ops[k].grads) { A.work[A.count] = ops[j].work; A.rows[A.count] = J.nblk[j]; A.count += 1; done[j] = true; }
    }
    hipLaunchKernelGGL(ode_rows_kernel, dim3((GODE_ODE_NPARAM + 255) / 256), dim3(256), 0, st, A);
    GODE_LAUNCH_CHECK();
  }
  return 0;
}

// the adaptive adjoints (method == 1, substeps == 0) of `count` (<= 8) dopri5 solves in one launch
extern "C" int gode_ode_bwd_multi(const gode_ode_bwd_op* ops, int32_t count, void* stream) {
  if (!ops || count < 1 || count > RV_MAX_JOBS) return GODE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  int spinning = 0;
  bool fallback = false;
  for (int k = 0; k < count; ++k) {
    const gode_ode_bwd_op* op = &ops[k];
    if (op->method != 1 || op->substeps != 0 || !op->traj || !op->gz || !op->work || !op->grads || !op->tout || op->N <= 0 || op->T < 1 ||
        !(op->rtol > 0.f) || !(op->atol >= 0.f) || op->zcols < 16 || op->zcols % 4 != 0)
      return GODE_E_ARG;
    if (!op->p.W1 || !op->p.b1 || !op->p.W2 || !op->p.b2) return GODE_E_ARG;
    if (op->prenet && (!op->x || !op->p.Wa || !op->p.ba || !op->p.Wb || !op->p.bb)) return GODE_E_ARG;
    const int nblk = (op->N + RV_TRAJ - 1) / RV_TRAJ;
    if (nblk > 1) { if (!op->sync) fallback = true; else spinning += nblk; }
  }
  if (spinning > RV_MAX_WGS) fallback = true;
  if (!fallback) return ode5_bwd_launch(ops, count, st);
  for (int k = 0; k < count; ++k) {       // in array order, each with its own accumulate flag
    const int nblk = (ops[k].N + RV_TRAJ - 1) / RV_TRAJ;
    const bool mfma = nblk > 1 && (!ops[k].sync || spinning > RV_MAX_WGS);
    const int rc = mfma ? gode_launch_ode_dopri5_bwd_mfma(&ops[k], st) : ode5_bwd_launch(&ops[k], 1, st);
    if (rc) return rc;
  }
  return 0;
}
