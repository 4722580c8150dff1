// conv_geom.h -- index algebra of the implicit-GEMM convolutions, shared by the HIP kernels and by the host-side
// checker (tests/hostcheck) so that the gather/scatter maps can be verified on the CPU build box without a GPU.
//
// One sub-GEMM ("phase") computes  C[m][n] = sum_k A[m][k] * B[n][k]  with
//   m  <-> (img, qd, qh, qw)   position in the phase's M-space
//   k  <-> (tap=(jd,jh,jw), c) tap-major, gathered channel minor
//   A[m][k] = src[img][qd*S+O + J*jd][..][..][c]  (zero outside the tensor), transformed on load
//   out position of m = (qd*OS + P, ...)
// FPROP (y = conv(x)):  S=stride, O=-pad, J=+1, OS=1, P=0, one phase.
// DGRAD (x = convT(y)): one phase per residue phi of the x position modulo the stride:
//   taps kk = kk0 + j*s with kk0=(phi+pad) mod s;  iy = q + (phi+pad-kk0)/s - j  ->  S=1, O=(phi+pad-kk0)/s, J=-1,
//   OS=s, P=phi.
#pragma once
#include <stdint.h>
#include "../../include/gode.h"

#if defined(__HIPCC__)
#define GODE_HD __host__ __device__ inline
#else
#define GODE_HD inline
#endif

#define GODE_MAX_PHASES 8

struct PhaseGeom {
  int32_t Md, Mh, Mw;   // M-space extent per image
  int32_t M;            // N*Md*Mh*Mw
  int32_t Td, Th, Tw;   // taps per dim in this phase
  int32_t K;            // Td*Th*Tw*Cg
  int32_t Kp;           // K rounded up to a multiple of 4 (row length of the packed panel)
  int32_t Od, Oh, Ow;   // gather base offsets
  int32_t Pd, Ph, Pw;   // output position offsets
  int32_t kd0, kh0, kw0;// first canonical tap per dim (for packing / wgrad)
  int32_t row0;         // first partial-stats row (m-block index) of this phase
  int32_t img0;         // first image of this phase (0 except for the image groups of a grouped FPROP, see gode_igemm_op.groups)
  int64_t w_off;        // float offset of this phase's panel inside wpack
};

struct IgemmGeom {
  int32_t nphase;
  int32_t Cg;           // gathered channels (k minor)
  int32_t Ncols;        // output channels (n)
  int32_t Sd, Sh, Sw;   // gather stride
  int32_t J;            // +1 / -1 tap direction
  int32_t Gd, Gh, Gw;   // gathered tensor extent
  int32_t OSd, OSh, OSw;// output position stride
  int32_t Xd, Xh, Xw;   // output tensor extent
  int32_t kstep_d, kstep_h, kstep_w; // canonical tap step per dim (1 for FPROP, stride for DGRAD)
  int32_t fullk;        // DGRAD of a conv whose kernel covers its whole input (y is 1x1x1): a plain GEMM with
                        // columns n=(tap,ci) -- the generator's first ConvTranspose2d (models/mocogan.py:201)
  PhaseGeom ph[GODE_MAX_PHASES];
};

GODE_HD int gode_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Builds the phase table.  Returns 0 on success.
inline int gode_build_igemm_geom(const gode_conv_geom& g, int dir, IgemmGeom* out) {
  IgemmGeom& G = *out;
  G.fullk = 0;
  if (g.N <= 0 || g.Ci <= 0 || g.Co <= 0) return GODE_E_SHAPE;
  // the conv relation must hold: Do = (Di + 2p - k)/s + 1
  if ((g.Di + 2 * g.pd - g.kd) / g.sd + 1 != g.Do || (g.Hi + 2 * g.ph - g.kh) / g.sh + 1 != g.Ho ||
      (g.Wi + 2 * g.pw - g.kw) / g.sw + 1 != g.Wo)
    return GODE_E_SHAPE;
  if (dir == GODE_FPROP) {
    G.nphase = 1; G.Cg = g.Ci; G.Ncols = g.Co;
    G.Sd = g.sd; G.Sh = g.sh; G.Sw = g.sw; G.J = 1;
    G.Gd = g.Di; G.Gh = g.Hi; G.Gw = g.Wi;
    G.OSd = G.OSh = G.OSw = 1;
    G.Xd = g.Do; G.Xh = g.Ho; G.Xw = g.Wo;
    G.kstep_d = G.kstep_h = G.kstep_w = 1;
    PhaseGeom& p = G.ph[0];
    p.Md = g.Do; p.Mh = g.Ho; p.Mw = g.Wo; p.M = g.N * p.Md * p.Mh * p.Mw;
    p.Td = g.kd; p.Th = g.kh; p.Tw = g.kw; p.K = p.Td * p.Th * p.Tw * G.Cg; p.Kp = (p.K + 3) & ~3;
    p.Od = -g.pd; p.Oh = -g.ph; p.Ow = -g.pw; p.Pd = p.Ph = p.Pw = 0;
    p.kd0 = p.kh0 = p.kw0 = 0; p.row0 = 0; p.img0 = 0; p.w_off = 0;
    return 0;
  }
  if (dir != GODE_DGRAD) return GODE_E_ARG;
  if (g.Do == 1 && g.Ho == 1 && g.Wo == 1 && g.pd == 0 && g.ph == 0 && g.pw == 0 && g.Di == g.kd &&
      g.Hi == g.kh && g.Wi == g.kw && g.kd * g.kh * g.kw > 1) {
    G.fullk = 1; G.nphase = 1; G.Cg = g.Co; G.Ncols = g.kd * g.kh * g.kw * g.Ci;
    G.Sd = G.Sh = G.Sw = 1; G.J = 1; G.Gd = G.Gh = G.Gw = 1;
    G.OSd = G.OSh = G.OSw = 1; G.Xd = G.Xh = G.Xw = 1;
    G.kstep_d = G.kstep_h = G.kstep_w = 1;
    PhaseGeom& p = G.ph[0];
    p.Md = p.Mh = p.Mw = 1; p.M = g.N; p.Td = p.Th = p.Tw = 1; p.K = G.Cg; p.Kp = (p.K + 3) & ~3;
    p.Od = p.Oh = p.Ow = 0; p.Pd = p.Ph = p.Pw = 0; p.kd0 = p.kh0 = p.kw0 = 0; p.row0 = 0; p.img0 = 0; p.w_off = 0;
    return 0;
  }
  if (g.sd * g.sh * g.sw > GODE_MAX_PHASES) return GODE_E_SHAPE;
  G.Cg = g.Co; G.Ncols = g.Ci;
  G.Sd = G.Sh = G.Sw = 1; G.J = -1;
  G.Gd = g.Do; G.Gh = g.Ho; G.Gw = g.Wo;
  G.OSd = g.sd; G.OSh = g.sh; G.OSw = g.sw;
  G.Xd = g.Di; G.Xh = g.Hi; G.Xw = g.Wi;
  G.kstep_d = g.sd; G.kstep_h = g.sh; G.kstep_w = g.sw;
  int np = 0; int64_t woff = 0;
  for (int fd = 0; fd < g.sd; ++fd)
    for (int fh = 0; fh < g.sh; ++fh)
      for (int fw = 0; fw < g.sw; ++fw) {
        PhaseGeom& p = G.ph[np++];
        const int phi[3] = {fd, fh, fw};
        const int X[3] = {g.Di, g.Hi, g.Wi}, s[3] = {g.sd, g.sh, g.sw}, k[3] = {g.kd, g.kh, g.kw},
                  pad[3] = {g.pd, g.ph, g.pw};
        int Mx[3], T[3], O[3], k0[3];
        for (int a = 0; a < 3; ++a) {
          Mx[a] = X[a] > phi[a] ? (X[a] - phi[a] + s[a] - 1) / s[a] : 0;
          int r = (phi[a] + pad[a]) % s[a]; if (r < 0) r += s[a];
          k0[a] = r;
          T[a] = k[a] > r ? (k[a] - r + s[a] - 1) / s[a] : 0;
          O[a] = (phi[a] + pad[a] - r) / s[a];   // exact division
        }
        p.Md = Mx[0]; p.Mh = Mx[1]; p.Mw = Mx[2]; p.M = g.N * Mx[0] * Mx[1] * Mx[2];
        p.Td = T[0]; p.Th = T[1]; p.Tw = T[2]; p.K = T[0] * T[1] * T[2] * G.Cg; p.Kp = (p.K + 3) & ~3;
        p.Od = O[0]; p.Oh = O[1]; p.Ow = O[2]; p.Pd = fd; p.Ph = fh; p.Pw = fw;
        p.kd0 = k0[0]; p.kh0 = k0[1]; p.kw0 = k0[2];
        p.row0 = 0; p.img0 = 0; p.w_off = woff;
        woff += (int64_t)G.Ncols * p.Kp;
      }
  G.nphase = np;
  return 0;
}

inline int64_t gode_pack_floats(const IgemmGeom& G) {
  int64_t t = 0;
  for (int i = 0; i < G.nphase; ++i) t += (int64_t)G.Ncols * G.ph[i].Kp;
  return t;
}

// canonical weight offset W[co][ci][kd][kh][kw]
GODE_HD int64_t gode_w_index(const gode_conv_geom& g, int co, int ci, int kd, int kh, int kw) {
  return ((((int64_t)co * g.Ci + ci) * g.kd + kd) * g.kh + kh) * g.kw + kw;
}

// element (n, k) of phase p's packed panel -> canonical weight offset, or -1 for padding.  co_perm (nullable) maps
// an internal y-side channel to its canonical channel (<0: structural zero).
GODE_HD int64_t gode_pack_source(const gode_conv_geom& g, int dir, const IgemmGeom& G, const PhaseGeom& p, int n,
                                 int k, const int32_t* co_perm) {
  if (k >= p.K) return -1;
  int co, ci, kd, kh, kw;
  if (G.fullk) {  // n = (tap, ci), k = co
    const int tapn = n / g.Ci;
    ci = n - tapn * g.Ci; co = k;
    kd = tapn / (g.kh * g.kw); kh = (tapn / g.kw) % g.kh; kw = tapn % g.kw;
  } else {
    const int tap = k / G.Cg, c = k - tap * G.Cg;
    const int jw = tap % p.Tw, jh = (tap / p.Tw) % p.Th, jd = tap / (p.Tw * p.Th);
    kd = p.kd0 + jd * G.kstep_d; kh = p.kh0 + jh * G.kstep_h; kw = p.kw0 + jw * G.kstep_w;
    if (dir == GODE_FPROP) { co = n; ci = c; } else { co = c; ci = n; }
  }
  if (co_perm) { co = co_perm[co]; if (co < 0) return -1; }
  return gode_w_index(g, co, ci, kd, kh, kw);
}

// gathered coordinate of (m-space coordinate q, tap j) along one dim; caller range-checks against G extent
GODE_HD int gode_gather_coord(int q, int S, int O, int J, int j) { return q * S + O + J * j; }
