// Host-side (g++) evaluation of the implicit-GEMM index algebra in gan-ode_amd/csrc/conv_geom.h: test-only code
// that lets the CPU suite verify the phase tables, the packed-weight map and the gather/scatter formulas against
// torch's own convolutions without a GPU.  It mirrors the address arithmetic of igemm.hip / wgrad.hip line by line
// (same fields, same formulas) but has no tiling and no MFMA.
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../gan-ode_amd/csrc/conv_geom.h"

extern "C" int hc_igemm(const gode_conv_geom* g, int dir, const float* src, const int64_t* gs_in, const float* w,
                        const int32_t* co_perm, float* out) {
  IgemmGeom G;
  int rc = gode_build_igemm_geom(*g, dir, &G);
  if (rc) return rc;
  int64_t gs[5];
  bool cl = true;
  for (int i = 0; i < 5; ++i) cl = cl && gs_in[i] == 0;
  if (cl) { gs[4] = 1; gs[3] = G.Cg; gs[2] = (int64_t)G.Gw * gs[3]; gs[1] = (int64_t)G.Gh * gs[2]; gs[0] = (int64_t)G.Gd * gs[1]; }
  else for (int i = 0; i < 5; ++i) gs[i] = gs_in[i];
  std::vector<float> wp(gode_pack_floats(G));
  for (int p = 0; p < G.nphase; ++p) {
    const PhaseGeom& P = G.ph[p];
    for (int n = 0; n < G.Ncols; ++n)
      for (int k = 0; k < P.Kp; ++k) {
        int64_t s = gode_pack_source(*g, dir, G, P, n, k, co_perm);
        wp[P.w_off + (int64_t)n * P.Kp + k] = s >= 0 ? w[s] : 0.f;
      }
  }
  for (int p = 0; p < G.nphase; ++p) {
    const PhaseGeom& P = G.ph[p];
    for (int m = 0; m < P.M; ++m) {
      const int qw = m % P.Mw; int t = m / P.Mw;
      const int qh = t % P.Mh; t /= P.Mh;
      const int qd = t % P.Md; const int img = t / P.Md;
      const int bd = qd * G.Sd + P.Od, bh = qh * G.Sh + P.Oh, bw = qw * G.Sw + P.Ow;
      const int64_t oo = ((((int64_t)img * G.Xd + qd * G.OSd + P.Pd) * G.Xh + qh * G.OSh + P.Ph) * G.Xw + qw * G.OSw + P.Pw) * G.Ncols;
      for (int n = 0; n < G.Ncols; ++n) {
        double acc = 0;
        for (int k = 0; k < P.K; ++k) {
          const int tap = k / G.Cg, c = k - tap * G.Cg;
          const int jw = tap % P.Tw, t2 = tap / P.Tw, jh = t2 % P.Th, jd = t2 / P.Th;
          const int id = bd + G.J * jd, ih = bh + G.J * jh, iw = bw + G.J * jw;
          if (id < 0 || id >= G.Gd || ih < 0 || ih >= G.Gh || iw < 0 || iw >= G.Gw) continue;
          acc += (double)src[img * gs[0] + id * gs[1] + ih * gs[2] + iw * gs[3] + c * gs[4]] * wp[P.w_off + (int64_t)n * P.Kp + k];
        }
        out[oo + n] = (float)acc;
      }
    }
  }
  return 0;
}

// dW[co][ci][taps] = sum_m y[m][co] * x[pos(m,tap)][ci]   (wgrad.hip's map; x, y channels-last)
extern "C" int hc_wgrad(const gode_conv_geom* gp, const float* x, const float* y, const int32_t* co_perm, float* dw) {
  const gode_conv_geom& g = *gp;
  const int taps = g.kd * g.kh * g.kw;
  const int64_t M = (int64_t)g.N * g.Do * g.Ho * g.Wo;
  for (int co = 0; co < g.Co; ++co)
    for (int j = 0; j < taps * g.Ci; ++j) {
      const int tap = j / g.Ci, ci = j - tap * g.Ci;
      const int kw = tap % g.kw, kh = (tap / g.kw) % g.kh, kd = tap / (g.kw * g.kh);
      double acc = 0;
      for (int64_t m = 0; m < M; ++m) {
        const int qw = m % g.Wo; int64_t t = m / g.Wo;
        const int qh = t % g.Ho; t /= g.Ho;
        const int qd = t % g.Do; const int img = (int)(t / g.Do);
        const int id = qd * g.sd - g.pd + kd, ih = qh * g.sh - g.ph + kh, iw = qw * g.sw - g.pw + kw;
        if (id < 0 || id >= g.Di || ih < 0 || ih >= g.Hi || iw < 0 || iw >= g.Wi) continue;
        acc += (double)y[m * g.Co + co] * x[((((int64_t)img * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.Ci + ci];
      }
      int c2 = co;
      if (co_perm) { c2 = co_perm[co]; if (c2 < 0) continue; }
      dw[((int64_t)c2 * g.Ci + ci) * taps + tap] = (float)acc;
    }
  return 0;
}
