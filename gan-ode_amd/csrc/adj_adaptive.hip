// adj_adaptive.hip -- torchdiffeq's ADAPTIVE adjoint on the device: the augmented state (y, a, g_theta) of every
// adjoint call is integrated backwards with dopri5 under the same controller as the forward solve and torchdiffeq's
// "mixed" norm (the maximum over the components' RMS norms; components = y, a and each ODEFunc parameter tensor).
//
// Reference: models/mocogan_ode_rnn.py:47-48 calls odeint_adjoint(ode_fn, h, [0, 1]) with defaults (dopri5, rtol 1e-7,
// atol 1e-9), so its backward integrates the adjoint adaptively too (SURVEY 3.3 tail); the restated algorithm is
// oracle/ode_ref.py:_Adjoint.backward + dopri5_solve (parity unpinned: torchdiffeq absent, no fixture).  Used by
// gode_odernn_bwd (substeps == 0) and gode_ode_bwd (method == 1, substeps == 0); the fixed-substep Kutta-3/8
// discretisation of round 1 stays available (substeps > 0).
//
// One workgroup = up to 64 trajectories (4 waves x 16, MFMA mapping of ode.hip); the error norm is taken jointly over
// the workgroup's trajectories (= the whole batch at the reference sizes; torchdiffeq: over the whole batch) -- the
// deviation already recorded for the forward solve.  The parameter components are batch sums: g_theta' = sum over
// trajectories of a (x) df/dtheta.  They are linear in the stage derivatives, so per trial step a wave accumulates three
// weighted sums of its stage outer products (5th-order solution S, error estimate E, mid-point M), the four waves'
// images are combined through LDS for the norm, and only accepted steps are added to the state.
#include <stdlib.h>
#include "ode_common.h"

#define ADJ_BLOCK 64
#define TH_W1 0
#define TH_B1 256
#define TH_W2 272
#define TH_B2 528
#define TH_N 544

struct Th { f32x4 W1, W2, b1, b2; };   // one wave's share: W* in MFMA D layout (lane (j, g) reg r = [4g + r][j]), b* per lane
__device__ __forceinline__ Th th_zero() { return Th{zero4(), zero4(), zero4(), zero4()}; }
__device__ __forceinline__ void th_axpy(Th& d, float c, const Th& x) { d.W1 += c * x.W1; d.W2 += c * x.W2; d.b1 += c * x.b1; d.b2 += c * x.b2; }
__device__ __forceinline__ Th th_sub(const Th& a, const Th& b) { return Th{a.W1 - b.W1, a.W2 - b.W2, a.b1 - b.b1, a.b2 - b.b2}; }
__device__ __forceinline__ Th th_scale(float c, const Th& a) { return Th{c * a.W1, c * a.W2, c * a.b1, c * a.b2}; }

__device__ __forceinline__ f32x4 abs4f(const f32x4 v) { return f32x4{fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3])}; }
__device__ __forceinline__ f32x4 max4f(const f32x4 a, const f32x4 b) { return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])}; }
__device__ __forceinline__ float sq4f(const f32x4 v) { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }

struct __attribute__((aligned(16))) AdjLds {
  float tile[4][2][16 * LDT];   // per-wave transpose tiles of the outer products
  float img[4][2][TH_N];        // per-wave images of theta-shaped quantities (slot 0: V / E, slot 1: S)
  float gtot[TH_N];             // workgroup total of the accepted theta state of the current adjoint call
  float red[2][4][8];
};

struct AdjSolver {
  f32x4 w1, w2, w1t, w2t, b1, b2;
  float rtol, atol, inv_ya;
  int s, g, wv, nw;
  int red_par = 0;
  bool valid;
  AdjLds* L;

  // sum of 6 values over the workgroup, every thread gets the totals (fixed order: deterministic, workgroup-uniform)
  __device__ __forceinline__ void block_sum6(float* v) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) L->red[red_par][wv][k] = v[k];
    }
    __syncthreads();       // one barrier per call: the slot sets alternate (see block_sum in odernn.hip)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float t = 0.f;
      for (int w = 0; w < nw; ++w) t += L->red[red_par][w][k];
      v[k] = t;
    }
    red_par ^= 1;
  }
  // The transpose tiles are per wave, and a wave's LDS operations execute in issue order: the only thing to guarantee is
  // that the compiler keeps the stores before the loads (and the previous call's loads before these stores) -- no
  // workgroup barrier (there were 24 of them per trial step).
  __device__ __forceinline__ f32x4 outer(const f32x4 P, const f32x4 Q) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *reinterpret_cast<f32x4*>(&L->tile[wv][0][s * LDT + 4 * g]) = P;
    *reinterpret_cast<f32x4*>(&L->tile[wv][1][s * LDT + 4 * g]) = Q;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 d = zero4();
#pragma unroll
    for (int r = 0; r < 4; ++r) d = MFMA16(L->tile[wv][0][(4 * g + r) * LDT + s], L->tile[wv][1][(4 * g + r) * LDT + s], d);
    return d;
  }
  // reversed-time augmented dynamics at (y, a): dy = -f(y), da = +a^T df/dy, dtheta = +a^T df/dtheta (this wave's share)
  __device__ __forceinline__ void eval(const f32x4 y, const f32x4 a, f32x4& ky, f32x4& ka, Th& X, bool want_theta) {
    const f32x4 h = tanh4(matvec(w1, y, b1));
    const f32x4 fv = matvec(w2, h, b2);
    const f32x4 du = matvec(w2t, a, zero4()) * (1.f - h * h);
    ka = matvec(w1t, du, zero4());
    ky = -fv;
    if (want_theta) { X.W2 = outer(a, h); X.W1 = outer(du, y); X.b2 = a; X.b1 = du; }
  }
  __device__ __forceinline__ void put_image(int slot, const Th& V) {
    float* im = L->img[wv][slot];
    const f32x4 sb1 = sum_over_samples(V.b1), sb2 = sum_over_samples(V.b2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      im[TH_W1 + (4 * g + r) * 16 + s] = V.W1[r];
      im[TH_W2 + (4 * g + r) * 16 + s] = V.W2[r];
    }
    if (s == 0) { *reinterpret_cast<f32x4*>(im + TH_B1 + 4 * g) = sb1; *reinterpret_cast<f32x4*>(im + TH_B2 + 4 * g) = sb2; }
  }
  __device__ __forceinline__ float img_total(int slot, int e) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += L->img[w][slot][e];
    return t;
  }
  // per-thread partial sums of (V/scale)^2 for the four parameter components (images must be in LDS, barrier done).
  // mode 0: V = image 0, scale = atol + rtol |G|        (initial step: f0, f1 - f0)
  // mode 1: V = image 0 (E), scale = atol + rtol max(|G|, |G + S|), S = image 1   (trial step)
  // mode 2: V = G itself, scale = atol + rtol |G|       (initial step: d0)
  __device__ __forceinline__ void theta_sq(int mode, float* out4) {
    out4[0] = out4[1] = out4[2] = out4[3] = 0.f;
    for (int e = threadIdx.x; e < TH_N; e += blockDim.x) {
      const float G = L->gtot[e];
      float V, sc;
      if (mode == 2) { V = G; sc = atol + rtol * fabsf(G); }
      else {
        V = img_total(0, e);
        sc = mode == 0 ? atol + rtol * fabsf(G) : atol + rtol * fmaxf(fabsf(G), fabsf(G + img_total(1, e)));
      }
      const float q = V / sc;
      const int comp = e < TH_B1 ? 0 : (e < TH_W2 ? 1 : (e < TH_B2 ? 2 : 3));
      out4[comp] += q * q;
    }
  }
  // torchdiffeq's mixed norm of an augmented vector given its y / a parts per lane and its theta part through theta_sq
  __device__ __forceinline__ float mixed(const f32x4 vy, const f32x4 va, int theta_mode) {
    float v[6];
    v[0] = valid ? sq4f(vy) : 0.f;
    v[1] = valid ? sq4f(va) : 0.f;
    __syncthreads();                      // the images written by put_image are complete
    theta_sq(theta_mode, v + 2);
    block_sum6(v);
    float m = fmaxf(v[0] * inv_ya, v[1] * inv_ya);
    m = fmaxf(m, fmaxf(v[2] * (1.f / 256.f), v[4] * (1.f / 256.f)));
    m = fmaxf(m, fmaxf(v[3] * (1.f / 16.f), v[5] * (1.f / 16.f)));
    return sqrtf(m);
  }

  // One adjoint call segment: integrate (y, a, G) from tau0 to tau1 (tau = -t increasing), return a and this wave's G
  // at tau1 read off the 4th-order interpolant of the last accepted step (RKAdaptiveStepsizeODESolver flow).
  // y is not returned: the caller resets it to the stored forward state, as odeint_adjoint does.  steps: trial count.
  // Returns false when the call stalled (trial-step limit, or a step size that no longer advances the fp64 clock:
  // torchdiffeq asserts there); the state is then only partially integrated and the caller reports a negative count.
  __device__ bool solve(f32x4 y0, f32x4& a0, Th& G, double tau0, double tau1, int& steps) {
    f32x4 k1y, k1a;
    Th X1;
    eval(y0, a0, k1y, k1a, X1, true);
    double dtd;
    {   // _initial_step(order 4) with the mixed norm
      const f32x4 scy = atol + abs4f(y0) * rtol, sca = atol + abs4f(a0) * rtol;
      const float d0 = mixed(y0 / scy, a0 / sca, 2);
      put_image(0, X1);
      const float d1 = mixed(k1y / scy, k1a / sca, 0);
      const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
      f32x4 fy, fa; Th Xb;
      eval(y0 + h0 * k1y, a0 + h0 * k1a, fy, fa, Xb, true);
      put_image(0, th_sub(Xb, X1));
      const float d2 = mixed((fy - k1y) / scy, (fa - k1a) / sca, 0) / h0;
      const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : powf(0.01f / fmaxf(d1, d2), 0.2f);
      dtd = (double)fminf(100.f * h0, h1);
    }
    double tcur = tau0;
    const int step_limit = steps + 100000;
    for (;;) {
      if (steps >= step_limit || !(tcur + dtd > tcur)) return false;
      const float dt = (float)dtd;
      f32x4 k2y, k2a, k3y, k3a, k4y, k4a, k5y, k5a, k6y, k6a, k7y, k7a;
      Th X, S = th_zero(), E = th_zero(), M = th_zero(), X7;
      th_axpy(S, 35.f / 384.f, X1); th_axpy(E, 35.f / 384.f - 1951.f / 21600.f, X1); th_axpy(M, 6025192743.f / 30085553152.f / 2.f, X1);
      eval(y0 + dt * (0.2f * k1y), a0 + dt * (0.2f * k1a), k2y, k2a, X, false);      // stage 2 has zero weight in S, E, M
      eval(y0 + dt * ((3.f / 40.f) * k1y + (9.f / 40.f) * k2y), a0 + dt * ((3.f / 40.f) * k1a + (9.f / 40.f) * k2a), k3y, k3a, X, true);
      th_axpy(S, 500.f / 1113.f, X); th_axpy(E, 500.f / 1113.f - 22642.f / 50085.f, X); th_axpy(M, 51252292925.f / 65400821598.f / 2.f, X);
      eval(y0 + dt * ((44.f / 45.f) * k1y + (-56.f / 15.f) * k2y + (32.f / 9.f) * k3y),
           a0 + dt * ((44.f / 45.f) * k1a + (-56.f / 15.f) * k2a + (32.f / 9.f) * k3a), k4y, k4a, X, true);
      th_axpy(S, 125.f / 192.f, X); th_axpy(E, 125.f / 192.f - 451.f / 720.f, X); th_axpy(M, -2691868925.f / 45128329728.f / 2.f, X);
      eval(y0 + dt * ((19372.f / 6561.f) * k1y + (-25360.f / 2187.f) * k2y + (64448.f / 6561.f) * k3y + (-212.f / 729.f) * k4y),
           a0 + dt * ((19372.f / 6561.f) * k1a + (-25360.f / 2187.f) * k2a + (64448.f / 6561.f) * k3a + (-212.f / 729.f) * k4a),
           k5y, k5a, X, true);
      th_axpy(S, -2187.f / 6784.f, X); th_axpy(E, -2187.f / 6784.f + 12231.f / 42400.f, X); th_axpy(M, 187940372067.f / 1594534317056.f / 2.f, X);
      eval(y0 + dt * ((9017.f / 3168.f) * k1y + (-355.f / 33.f) * k2y + (46732.f / 5247.f) * k3y + (49.f / 176.f) * k4y + (-5103.f / 18656.f) * k5y),
           a0 + dt * ((9017.f / 3168.f) * k1a + (-355.f / 33.f) * k2a + (46732.f / 5247.f) * k3a + (49.f / 176.f) * k4a + (-5103.f / 18656.f) * k5a),
           k6y, k6a, X, true);
      th_axpy(S, 11.f / 84.f, X); th_axpy(E, 11.f / 84.f - 649.f / 6300.f, X); th_axpy(M, -1776094331.f / 19743644256.f / 2.f, X);
      const f32x4 y1 = y0 + dt * ((35.f / 384.f) * k1y + (500.f / 1113.f) * k3y + (125.f / 192.f) * k4y + (-2187.f / 6784.f) * k5y + (11.f / 84.f) * k6y);
      const f32x4 a1 = a0 + dt * ((35.f / 384.f) * k1a + (500.f / 1113.f) * k3a + (125.f / 192.f) * k4a + (-2187.f / 6784.f) * k5a + (11.f / 84.f) * k6a);
      eval(y1, a1, k7y, k7a, X7, true);
      th_axpy(E, -1.f / 60.f, X7); th_axpy(M, 11237099.f / 235043384.f / 2.f, X7);
      const f32x4 erry = dt * ((35.f / 384.f - 1951.f / 21600.f) * k1y + (500.f / 1113.f - 22642.f / 50085.f) * k3y + (125.f / 192.f - 451.f / 720.f) * k4y +
                               (-2187.f / 6784.f + 12231.f / 42400.f) * k5y + (11.f / 84.f - 649.f / 6300.f) * k6y + (-1.f / 60.f) * k7y);
      const f32x4 erra = dt * ((35.f / 384.f - 1951.f / 21600.f) * k1a + (500.f / 1113.f - 22642.f / 50085.f) * k3a + (125.f / 192.f - 451.f / 720.f) * k4a +
                               (-2187.f / 6784.f + 12231.f / 42400.f) * k5a + (11.f / 84.f - 649.f / 6300.f) * k6a + (-1.f / 60.f) * k7a);
      const Th Sd = th_scale(dt, S), Ed = th_scale(dt, E);
      put_image(0, Ed);
      put_image(1, Sd);
      const f32x4 toly = atol + rtol * max4f(abs4f(y0), abs4f(y1)), tola = atol + rtol * max4f(abs4f(a0), abs4f(a1));
      const float ratio = mixed(erry / toly, erra / tola, 1);
      ++steps;
#ifdef GODE_ADJ_DEBUG
      if (threadIdx.x == 0 && blockIdx.x == 0) printf("  trial %d tcur %.6f dt %.6f ratio %.4g\n", steps, tcur, dtd, (double)ratio);
#endif
      if (ratio <= 1.f) {
        const bool last = tcur + dtd >= tau1;
        float wS = 1.f, wM = 0.f, w1c = 0.f, w7c = 0.f;       // G(tau) - G0 = wS*S + wM*Mid + dt*(w1c*X1 + w7c*X7)
        if (last) {
          // 4th-order interpolant through (z0, z_mid, z1, f0, f1); abscissa from the fp32-rounded times as torchdiffeq
          const float x = ((float)tau1 - (float)tcur) / ((float)(tcur + dtd) - (float)tcur);
          const f32x4 amid = a0 + dt * ((6025192743.f / 30085553152.f / 2.f) * k1a + (51252292925.f / 65400821598.f / 2.f) * k3a +
                                        (-2691868925.f / 45128329728.f / 2.f) * k4a + (187940372067.f / 1594534317056.f / 2.f) * k5a +
                                        (-1776094331.f / 19743644256.f / 2.f) * k6a + (11237099.f / 235043384.f / 2.f) * k7a);
          const f32x4 ca = 2.f * dt * (k7a - k1a) - 8.f * (a1 + a0) + 16.f * amid;
          const f32x4 cb = dt * (5.f * k1a - 3.f * k7a) + 18.f * a0 + 14.f * a1 - 32.f * amid;
          const f32x4 cc = dt * (k7a - 4.f * k1a) - 11.f * a0 - 5.f * a1 + 16.f * amid;
          const f32x4 cd = dt * k1a;
          a0 = a0 + x * (cd + x * (cc + x * (cb + x * ca)));
          // the same polynomial for the (linear) theta components: coefficients of S, Mid, X1, X7
          const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
          wS = -8.f * x4 + 14.f * x3 - 5.f * x2;
          wM = 16.f * x4 - 32.f * x3 + 16.f * x2;
          w1c = -2.f * x4 + 5.f * x3 - 4.f * x2 + x;
          w7c = 2.f * x4 - 3.f * x3 + x2;
        } else {
          a0 = a1;
        }
        Th dG = th_scale(wS, Sd);
        th_axpy(dG, wM * dt, M); th_axpy(dG, w1c * dt, X1); th_axpy(dG, w7c * dt, X7);
        th_axpy(G, 1.f, dG);
        // workgroup total of the state (the tolerance of the next step / next segment is relative to it)
        __syncthreads();
        put_image(0, dG);
        __syncthreads();
        for (int e = threadIdx.x; e < TH_N; e += blockDim.x) L->gtot[e] += img_total(0, e);
        __syncthreads();
        if (last) return true;
        tcur += dtd; y0 = y1; k1y = k7y; k1a = k7a; X1 = X7;
      }
      float fac;
      if (ratio == 0.f) fac = 10.f;
      else { fac = 0.9f * powf(ratio, -0.2f); fac = fminf(10.f, fmaxf(fac, ratio < 1.f ? 1.f : 0.2f)); }
      dtd *= (double)fac;
    }
  }
};

#define RNN_NPARAM 2176
#define RO_W1 0
#define RO_B1 256
#define RO_W2 272
#define RO_B2 528
#define RO_WIH 544
#define RO_WHH 1312
#define RO_BIH 2080
#define RO_BHH 2128

// ODE-RNN backward with the adaptive adjoint: per frame GRU backward, then ONE adjoint call over [1, 0] (its theta state
// starts at zero, as each odeint_adjoint call's does).  One partial row of `work` per wave (16 trajectories).
__global__ void __launch_bounds__(ADJ_BLOCK * 4) odernn_bwd_adaptive_kernel(const gode_odernn_bwd_op a) {
  __shared__ AdjLds lds;
  AdjSolver S;
  const int l = threadIdx.x & 63;
  S.s = l & 15; S.g = l >> 4; S.wv = threadIdx.x >> 6; S.nw = blockDim.x >> 6; S.L = &lds;
  const int s = S.s, g = S.g;
  const int n = blockIdx.x * ADJ_BLOCK + S.wv * 16 + s;
  S.valid = n < a.N;
  const bool valid = S.valid;
  const int T = a.T;
  const int nvalid = (a.N - blockIdx.x * ADJ_BLOCK) < ADJ_BLOCK ? (a.N - blockIdx.x * ADJ_BLOCK) : ADJ_BLOCK;
  S.inv_ya = 1.f / (float)(nvalid * 16);
  float rtol = a.rtol, atol = a.atol;
  asm volatile("" : "+v"(rtol), "+v"(atol));     // (see ode_dopri5_fwd_kernel: keeps hipcc off a two-SGPR packed op)
  S.rtol = rtol; S.atol = atol;
  float* part = a.work + (int64_t)(blockIdx.x * 4 + S.wv) * RNN_NPARAM;

  S.w1 = ld4(a.p.W1 + s * 16 + 4 * g); S.w2 = ld4(a.p.W2 + s * 16 + 4 * g);
  S.b1 = ld4(a.p.b1 + 4 * g); S.b2 = ld4(a.p.b2 + 4 * g);
#pragma unroll
  for (int r = 0; r < 4; ++r) { S.w1t[r] = a.p.W1[(4 * g + r) * 16 + s]; S.w2t[r] = a.p.W2[(4 * g + r) * 16 + s]; }
  // The GRU weights are re-read per frame (L1/L2-resident, 15 float4 per lane) and the GRU parameter gradients live in
  // LDS (one float4 slot per lane and tensor): together with the ~380 registers of the dopri5 stages they would not
  // fit the 512-register file -- a build that kept them in registers spilled 41 VGPRs to scratch (and, on the box,
  // returned gradients off by percents; this build uses no scratch: .private_segment_fixed_size 0).
  __shared__ __attribute__((aligned(16))) float ggru[12][ADJ_BLOCK * 4][4];     // gWih[3], gWhh[3], gbih[3], gbhh[3]
#pragma unroll
  for (int q = 0; q < 12; ++q) *reinterpret_cast<f32x4*>(ggru[q][threadIdx.x]) = zero4();
  Th Gacc = th_zero();
  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto upstream = [&](int t) {
    if (!valid) return zero4();
    if (a.sel_t == nullptr) return ld4(a.gz + ((int64_t)n * T + t) * a.zcols + 4 * g);
    return t == tsel ? ld4(a.gz + (int64_t)n * a.zcols + 4 * g) : zero4();
  };
  auto gacc = [&](int q, const f32x4 v) {
    f32x4* p = reinterpret_cast<f32x4*>(ggru[q][threadIdx.x]);
    *p = *p + v;
  };
  f32x4 carry = zero4();
  int steps = 0;
  for (int t = T - 1; t >= 0; --t) {
    const f32x4 gh = carry + upstream(t);                                        // dL/dh_{t+1}
    const f32x4 e = valid ? ld4(a.noise + ((int64_t)(t + 1) * a.N + n) * 16 + 4 * g) : zero4();
    const f32x4 hp = valid ? ld4(a.hp + ((int64_t)n * T + t) * 16 + 4 * g) : zero4();
    f32x4 adj;
    {
      f32x4 gi[3], hh[3];     // gate pre-activations: W_i* e + b_i*, W_h* h' + b_h*
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        gi[q] = matvec(ld4(a.p.Wih + (16 * q + s) * 16 + 4 * g), e, ld4(a.p.bih + 16 * q + 4 * g));
        hh[q] = matvec(ld4(a.p.Whh + (16 * q + s) * 16 + 4 * g), hp, ld4(a.p.bhh + 16 * q + 4 * g));
      }
      const f32x4 r = sigmoid4(gi[0] + hh[0]);
      const f32x4 zg = sigmoid4(gi[1] + hh[1]);
      const f32x4 hn = hh[2];
      const f32x4 nn = tanh4(gi[2] + r * hn);
      const f32x4 dn_pre = gh * (1.f - zg) * (1.f - nn * nn);
      const f32x4 dz_pre = gh * (hp - nn) * zg * (1.f - zg);
      const f32x4 dr_pre = dn_pre * hn * r * (1.f - r);
      const f32x4 dgi[3] = {dr_pre, dz_pre, dn_pre};
      const f32x4 dgh[3] = {dr_pre, dz_pre, dn_pre * r};
      adj = gh * zg;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        gacc(6 + q, dgi[q]); gacc(9 + q, dgh[q]);
        gacc(q, S.outer(dgi[q], e));
        gacc(3 + q, S.outer(dgh[q], hp));
        f32x4 whht;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) whht[rr] = a.p.Whh[(16 * q + 4 * g + rr) * 16 + s];   // (Whh_q)^T operand
        adj = matvec(whht, dgh[q], adj);
      }
    }
    // adjoint call of this frame's solve: state (y = h', a = adj, g_theta = 0) at t = 1 back to t = 0
    __syncthreads();
    for (int k = threadIdx.x; k < TH_N; k += blockDim.x) lds.gtot[k] = 0.f;
    __syncthreads();
    Th G = th_zero();
    const int before = steps;
    const bool ok = S.solve(hp, adj, G, -1.0, 0.0, steps);
    if (a.nsteps && blockIdx.x == 0 && threadIdx.x == 0) a.nsteps[t] = ok ? steps - before : -(steps - before) - 1;
    if (!ok) break;                  // (uniform over the workgroup: the decision comes from the joint norm)
    th_axpy(Gacc, 1.f, G);
    carry = adj;                                                                 // dL/dh_t
  }
  if (blockIdx.x * ADJ_BLOCK + S.wv * 16 >= a.N) return;      // a wave without trajectories owns no partial row
  const f32x4 sb1 = sum_over_samples(Gacc.b1), sb2 = sum_over_samples(Gacc.b2);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    part[RO_W1 + (4 * g + r) * 16 + s] = Gacc.W1[r];
    part[RO_W2 + (4 * g + r) * 16 + s] = Gacc.W2[r];
  }
  if (s == 0) { *reinterpret_cast<f32x4*>(part + RO_B1 + 4 * g) = sb1; *reinterpret_cast<f32x4*>(part + RO_B2 + 4 * g) = sb2; }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const f32x4 gWih = *reinterpret_cast<f32x4*>(ggru[q][threadIdx.x]), gWhh = *reinterpret_cast<f32x4*>(ggru[3 + q][threadIdx.x]);
    const f32x4 si = sum_over_samples(*reinterpret_cast<f32x4*>(ggru[6 + q][threadIdx.x]));
    const f32x4 sh = sum_over_samples(*reinterpret_cast<f32x4*>(ggru[9 + q][threadIdx.x]));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      part[RO_WIH + (16 * q + 4 * g + r) * 16 + s] = gWih[r];
      part[RO_WHH + (16 * q + 4 * g + r) * 16 + s] = gWhh[r];
    }
    if (s == 0) { *reinterpret_cast<f32x4*>(part + RO_BIH + 16 * q + 4 * g) = si; *reinterpret_cast<f32x4*>(part + RO_BHH + 16 * q + 4 * g) = sh; }
  }
}

int gode_launch_odernn_bwd_adaptive(const gode_odernn_bwd_op* op, hipStream_t st) {
  const int nblocks = (op->N + ADJ_BLOCK - 1) / ADJ_BLOCK;
  // every wave writes one partial row; rows of waves beyond N hold zeros (their lanes are masked)
  const int per = op->N < ADJ_BLOCK ? op->N : ADJ_BLOCK;
  hipLaunchKernelGGL(odernn_bwd_adaptive_kernel, dim3(nblocks), dim3(((per + 15) / 16) * 64), 0, st, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Plain Neural-ODE generator solved with dopri5 (gode_ode_bwd_op.method == 1): ONE adjoint call over the T output
// times -- the theta state is carried across the output intervals, the solver is restarted on every interval
// (oracle/ode_ref.py:_Adjoint.backward) -- followed by the pre-net backward of ode_bwd_kernel.
#define OFF_WA 0
#define OFF_BA 1024
#define OFF_WB 1088
#define OFF_BB 2112
#define OFF_W1 2128
#define OFF_B1 2384
#define OFF_W2 2400
#define OFF_B2 2656
__global__ void __launch_bounds__(ADJ_BLOCK * 4) ode_dopri5_bwd_kernel(const gode_ode_bwd_op a) {
  __shared__ AdjLds lds;
  AdjSolver S;
  const int l = threadIdx.x & 63;
  S.s = l & 15; S.g = l >> 4; S.wv = threadIdx.x >> 6; S.nw = blockDim.x >> 6; S.L = &lds;
  const int s = S.s, g = S.g;
  const int n = blockIdx.x * ADJ_BLOCK + S.wv * 16 + s;
  S.valid = n < a.N;
  const bool valid = S.valid;
  const int T = a.T;
  const int nvalid = (a.N - blockIdx.x * ADJ_BLOCK) < ADJ_BLOCK ? (a.N - blockIdx.x * ADJ_BLOCK) : ADJ_BLOCK;
  S.inv_ya = 1.f / (float)(nvalid * 16);
  float rtol = a.rtol, atol = a.atol;
  asm volatile("" : "+v"(rtol), "+v"(atol));
  S.rtol = rtol; S.atol = atol;
  float* part = a.work + (int64_t)(blockIdx.x * 4 + S.wv) * GODE_ODE_NPARAM;

  S.w1 = ld4(a.p.W1 + s * 16 + 4 * g); S.w2 = ld4(a.p.W2 + s * 16 + 4 * g);
  S.b1 = ld4(a.p.b1 + 4 * g); S.b2 = ld4(a.p.b2 + 4 * g);
#pragma unroll
  for (int r = 0; r < 4; ++r) { S.w1t[r] = a.p.W1[(4 * g + r) * 16 + s]; S.w2t[r] = a.p.W2[(4 * g + r) * 16 + s]; }
  const int tsel = (a.sel_t && valid) ? a.sel_t[n] : -1;
  auto upstream = [&](int t) {
    if (!valid) return zero4();
    if (a.sel_t == nullptr) return ld4(a.gz + ((int64_t)n * T + t) * a.zcols + 4 * g);
    return t == tsel ? ld4(a.gz + (int64_t)n * a.zcols + 4 * g) : zero4();
  };
  for (int k = threadIdx.x; k < TH_N; k += blockDim.x) lds.gtot[k] = 0.f;
  __syncthreads();
  Th G = th_zero();
  f32x4 adj = upstream(T - 1);
  int steps = 0;
  bool ok = true;
  for (int i = T - 1; i >= 1 && ok; --i) {
    const f32x4 y = valid ? ld4(a.traj + ((int64_t)n * T + i) * 16 + 4 * g) : zero4();
    ok = S.solve(y, adj, G, -(double)a.tout[i], -(double)a.tout[i - 1], steps);
    adj = adj + upstream(i - 1);
  }
  if (a.nsteps && threadIdx.x == 0) a.nsteps[blockIdx.x] = ok ? steps : -steps - 1;
  const bool wave_live = blockIdx.x * ADJ_BLOCK + S.wv * 16 < a.N;      // a wave without trajectories owns no partial row
  if (wave_live) {
    const f32x4 sb1 = sum_over_samples(G.b1), sb2 = sum_over_samples(G.b2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      part[OFF_W1 + (4 * g + r) * 16 + s] = G.W1[r];
      part[OFF_W2 + (4 * g + r) * 16 + s] = G.W2[r];
    }
    if (s == 0) { *reinterpret_cast<f32x4*>(part + OFF_B1 + 4 * g) = sb1; *reinterpret_cast<f32x4*>(part + OFF_B2 + 4 * g) = sb2; }
  }
  if (a.prenet) {     // pre-net backward; adj = dL/d(pre-net output)   (as ode_bwd_kernel)
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
    if (s == 0 && wave_live) *reinterpret_cast<f32x4*>(part + OFF_BB + 4 * g) = sbb;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 dWb = S.outer(g0, lrelu4(hpre[m]));
      f32x4 wbt;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (wave_live) part[OFF_WB + (4 * g + r) * 64 + 16 * m + s] = dWb[r];
        wbt[r] = a.p.Wb[(4 * g + r) * 64 + 16 * m + s];
      }
      const f32x4 gh = lrelu_grad4(hpre[m], matvec(wbt, g0, zero4()));
      const f32x4 sba = sum_over_samples(gh);
      if (s == 0 && wave_live) *reinterpret_cast<f32x4*>(part + OFF_BA + 16 * m + 4 * g) = sba;
      const f32x4 dWa = S.outer(gh, x);
#pragma unroll
      for (int r = 0; r < 4; ++r) if (wave_live) part[OFF_WA + (16 * m + 4 * g + r) * 16 + s] = dWa[r];
    }
  } else if (wave_live) {
    for (int k = l; k < OFF_W1; k += 64) part[k] = 0.f;
  }
}

int gode_launch_ode_dopri5_bwd(const gode_ode_bwd_op* op, hipStream_t st) {
  const int nblocks = (op->N + ADJ_BLOCK - 1) / ADJ_BLOCK;
  const int per = op->N < ADJ_BLOCK ? op->N : ADJ_BLOCK;
  hipLaunchKernelGGL(ode_dopri5_bwd_kernel, dim3(nblocks), dim3(((per + 15) / 16) * 64), 0, st, *op);
  GODE_LAUNCH_CHECK();
  return 0;
}
