// igemm.hip -- implicit-GEMM convolution for gfx950 on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One kernel serves Conv2d/Conv3d forward (FPROP), their input gradients and the ConvTranspose2d forward
// (DGRAD, one sub-GEMM per stride phase so that no MFMA is spent on structurally-zero taps), and the plain GEMM of
// the generator's first layer (FULLK).  Reference layers: models/mocogan.py:72-89,138-159,200-215,
// models/mocogan_ode.py:66-84.
//
// Data flow per workgroup (256 threads = 4 waves, BMxBN output tile, K slabs of 32), three staging paths:
//   LDS-DMA (igemm_fast_kernel MODE 2; every heavy, transform-free GEMM): global --global_load_lds_dwordx4--> LDS
//     [row][32], 16-byte chunks XOR-swizzled through the source address, two buffers, one barrier per slab;
//   registers (MODE 0/1 and the generic igemm_kernel; BN/activation of the previous layer fused into the load, odd
//     channel counts, strided sources): float4 (or scalar) gathers -> registers -> ds_write_b128 -> LDS [row][36];
//   then ds_read_b128 fragments -> 32x32x2 MFMA, 2x2 (2x1, 1x1) tiles per wave.
// MFMA k-index trick: lane half h supplies k = 4h+j in step j, so one ds_read_b128 feeds four MFMAs.
// Epilogue: raw (or tanh) store, 128-B coalesced along channels, plus deterministic per-column partial sums for
// train-mode BatchNorm (no atomics).
#include <stdlib.h>
#include "common.h"
#include "conv_geom.h"

int gode_launch_conv_patch_fprop(const gode_igemm_op* op, const int64_t* gs, hipStream_t st);   // conv_patch.hip

struct IgemmArgs {
  IgemmGeom G;
  const float* src;
  const float* w;
  float* out;
  const float* scale;
  const float* shift;
  float* stats;
  int32_t gsN, gsD, gsH, gsW, gsC;
  int32_t act, epilogue;
  int32_t MB, NB, xcd_mode;   // FAST kernel: 1-D grid of MB*NB*nphase workgroups, decoded XCD-aware (see igemm_block_id)
  int32_t tapskip, dmajor;           // FAST kernel MODE 3; dmajor > 0: rows are (depth, image, h, w)-ordered with this many images
  int32_t stagger;                   // tuning knob: de-phase co-resident workgroups (units of s_sleep 64)
  int32_t ksplit, slabs_per_split;   // split-K: grid is ksplit copies of the above; partial tiles go to work
  float* work;
  int32_t out_numel;
  int32_t stats_rows;   // number of partial-statistics rows of the launch (the stats buffer is [2][ncols][stats_rows])
  uint32_t src_bytes, w_bytes;   // LDS-DMA kernels address both operands as raw buffers (32-bit byte offsets): their extents
};

// Workgroups are dealt round-robin to the 8 XCDs in linear id order (id and id+8 share an XCD and its 4 MiB L2);
// this is used for locality only.  xcd_mode 1 keeps all m-blocks of one (n-block, phase) "combo" on one XCD (its weight
// panel stays L2-resident), xcd_mode 2 keeps all combos of one m-block on one XCD back to back (the gathered
// activation rows are fetched once), xcd_mode 3 additionally gives each XCD a contiguous run of m-blocks (halo rows
// shared by neighbouring output rows hit the same L2; the default), 0 is the plain order.  Every mode is a bijection
// of [0, MB*NB*nphase).
__device__ __forceinline__ void igemm_block_id(const IgemmArgs& a, int& mblk, int& nblk, int& phase, int& split) {
  const int MB = a.MB, NC = a.NB * a.G.nphase;
  split = blockIdx.x / (MB * NC);
  const int id = blockIdx.x - split * (MB * NC);
  int combo;
  if (a.xcd_mode == 1) {
    const int x = id & 7, j = id >> 3;
    if ((NC & 7) == 0) { combo = x + 8 * (j / MB); mblk = j % MB; }
    else { const int parts = 8 / NC; combo = x % NC; mblk = j * parts + x / NC; }
  } else if (a.xcd_mode == 2) {
    const int x = id & 7, j = id >> 3;
    combo = j % NC; mblk = (j / NC) * 8 + x;
  } else if (a.xcd_mode == 3) {          // like 2, but every XCD owns a CONTIGUOUS run of m-blocks: neighbouring output
    const int x = id & 7, j = id >> 3;   // rows (which share input halo rows) hit the same L2
    combo = j % NC; mblk = x * (MB >> 3) + j / NC;
  } else {
    mblk = id % MB; combo = id / MB;
  }
  nblk = combo % a.NB; phase = combo / a.NB;
}

template <int WM, int WN, int TM, int TN, bool VEC>
__global__ void __launch_bounds__(WM* WN * 64) igemm_kernel(const IgemmArgs a) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, LDK = 36;
  constexpr int RPP = NT / 8;  // tile rows covered by one loader pass (8 threads x float4 per 32-wide row)
  constexpr int AP = BM / RPP, BP = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/loader mismatch");
  constexpr int BUF = (BM + BN) * LDK;
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF + BM * 5];
  int* rowinfo = reinterpret_cast<int*>(smem + 2 * BUF);  // [BM][4]: base, bd, bh, bw
  int* outoff = rowinfo + BM * 4;                          // [BM]

  const PhaseGeom& P = a.G.ph[blockIdx.z];
  const int mblk = blockIdx.x, nblk = blockIdx.y;
  if (mblk * BM >= P.M) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int Cg = a.G.Cg, Ncols = a.G.Ncols;

  for (int r = tid; r < BM; r += NT) {
    const int m = mblk * BM + r;
    int base = -1, bd = 0, bh = 0, bw = 0, oo = -1;
    if (m < P.M) {
      const int qw = m % P.Mw; int t = m / P.Mw;
      const int qh = t % P.Mh; t /= P.Mh;
      const int qd = t % P.Md; const int img = t / P.Md + P.img0;
      base = img * a.gsN;
      bd = qd * a.G.Sd + P.Od; bh = qh * a.G.Sh + P.Oh; bw = qw * a.G.Sw + P.Ow;
      oo = (((img * a.G.Xd + qd * a.G.OSd + P.Pd) * a.G.Xh + qh * a.G.OSh + P.Ph) * a.G.Xw + qw * a.G.OSw + P.Pw) *
           Ncols;
    }
    rowinfo[r * 4 + 0] = base; rowinfo[r * 4 + 1] = bd; rowinfo[r * 4 + 2] = bh; rowinfo[r * 4 + 3] = bw;
    outoff[r] = oo;
  }
  __syncthreads();

  int rbase[AP], rbd[AP], rbh[AP], rbw[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int r = (tid >> 3) + RPP * i;
    rbase[i] = rowinfo[r * 4 + 0]; rbd[i] = rowinfo[r * 4 + 1]; rbh[i] = rowinfo[r * 4 + 2]; rbw[i] = rowinfo[r * 4 + 3];
  }
  const int kchunk = (tid & 7) * 4;
  const float* wp = a.w + P.w_off;
  const int J = a.G.J, Gd = a.G.Gd, Gh = a.G.Gh, Gw = a.G.Gw;
  const int nslab = (P.Kp + 31) >> 5;

  f32x4 ra[AP], rb[BP], sc4, sh4;
  unsigned amask = 0;
  const bool xf = a.scale != nullptr;

  auto fetch = [&](int slab) {
    const int kk = slab * 32 + kchunk;
    amask = 0;
    sc4 = f32x4{1.f, 1.f, 1.f, 1.f}; sh4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (VEC) {
      if (kk < P.K) {
        const int tap = kk / Cg, c = kk - tap * Cg;
        const int jw = tap % P.Tw, t2 = tap / P.Tw, jh = t2 % P.Th, jd = t2 / P.Th;
        if (xf) { sc4 = *reinterpret_cast<const f32x4*>(a.scale + c); sh4 = *reinterpret_cast<const f32x4*>(a.shift + c); }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const int id = rbd[i] + J * jd, ih = rbh[i] + J * jh, iw = rbw[i] + J * jw;
          if (rbase[i] >= 0 && (unsigned)id < (unsigned)Gd && (unsigned)ih < (unsigned)Gh && (unsigned)iw < (unsigned)Gw) {
            ra[i] = *reinterpret_cast<const f32x4*>(a.src + rbase[i] + id * a.gsD + ih * a.gsH + iw * a.gsW + c);
            amask |= 1u << i;
          }
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kk + e;
        if (k < P.K) {
          const int tap = k / Cg, c = k - tap * Cg;
          const int jw = tap % P.Tw, t2 = tap / P.Tw, jh = t2 % P.Th, jd = t2 / P.Th;
          if (xf) { sc4[e] = a.scale[c]; sh4[e] = a.shift[c]; }
#pragma unroll
          for (int i = 0; i < AP; ++i) {
            const int id = rbd[i] + J * jd, ih = rbh[i] + J * jh, iw = rbw[i] + J * jw;
            if (rbase[i] >= 0 && (unsigned)id < (unsigned)Gd && (unsigned)ih < (unsigned)Gh && (unsigned)iw < (unsigned)Gw) {
              ra[i][e] = a.src[rbase[i] + id * a.gsD + ih * a.gsH + iw * a.gsW + c * a.gsC];
              amask |= 1u << (i * 4 + e);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int n = nblk * BN + (tid >> 3) + RPP * i;
      if (n < Ncols && kk < P.Kp) rb[i] = *reinterpret_cast<const f32x4*>(wp + (int64_t)n * P.Kp + kk);
      else rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  auto stage = [&](int buf) {
    float* As = smem + buf * BUF;
    float* Bs = As + BM * LDK;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = VEC ? ((amask >> i) & 1u) : ((amask >> (i * 4 + e)) & 1u);
        float t = ra[i][e];
        if (xf) t = gode_act(t * sc4[e] + sh4[e], a.act);
        else t = gode_act(t, a.act);
        v[e] = ok ? t : 0.f;
      }
      *reinterpret_cast<f32x4*>(As + ((tid >> 3) + RPP * i) * LDK + kchunk) = v;
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<f32x4*>(Bs + ((tid >> 3) + RPP * i) * LDK + kchunk) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(0);
  stage(0);
  __syncthreads();

  const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) fetch(s + 1);
    const float* As = smem + buf * BUF + (wm * TM * 32 + frag_row) * LDK + frag_k;
    const float* Bs = smem + buf * BUF + BM * LDK + (wn * TN * 32 + frag_row) * LDK + frag_k;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDK + kg * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDK + kg * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nslab) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int ccol = lane & 31, crow = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = nblk * BN + (wn * TN + j) * 32 + ccol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + crow;
        const int oo = outoff[row];
        if (oo >= 0 && col < Ncols) {
          float v = acc[i][j][r];
          if (a.epilogue == GODE_EPI_TANH) v = tanhf(v);
          a.out[oo + col] = v;
        }
      }
    }

  if (a.stats != nullptr) {
    float* sred = smem;  // [WM][BN][2]; all waves are past the last barrier of the main loop
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lane < 32) {
        const int c = (wn * TN + j) * 32 + lane;
        sred[(wm * BN + c) * 2 + 0] = s1;
        sred[(wm * BN + c) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      const int col = nblk * BN + c;
      if (col < Ncols) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += sred[(w * BN + c) * 2 + 0]; s2 += sred[(w * BN + c) * 2 + 1]; }
        // [which][column][row]: the finalize kernel then walks contiguous rows per channel
        a.stats[(int64_t)col * a.stats_rows + P.row0 + mblk] = s1;
        a.stats[((int64_t)Ncols + col) * a.stats_rows + P.row0 + mblk] = s2;
      }
    }
  }
}

#define IGEMM_MAX_LIVE_TAPS 64   // MODE 3 tap table (+1 slot read past the end)

// ---------------------------------------------------------------------------------------------------------------
// FAST variant for the layers that carry the FLOPs: vector gather, Cg % 32 == 0 (so a 32-wide K slab never straddles
// a tap and the tap decode is wave-uniform scalar work), K % 32 == 0.  The loop body is one basic block: loads are
// unconditional (clamped address + mask), the BN/activation transform is branch-free, and the staging work of slab
// s+1 is placed between the MFMA groups of slab s so that it issues in the shadow of the matrix pipe.
// MODE 1 : registers, two LDS buffers, one barrier per slab, staging interleaved with this wave's own MFMAs.
// MODE 0 : registers, one LDS buffer (half the LDS => twice the resident workgroups), two barriers per slab; the
//           staging of one workgroup is covered by the MFMAs of the others on the SIMD.  Default when XF.
// MODE 2 (LDS-DMA): the slab is written to LDS by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass);
//           two unpadded LDS buffers whose 16-byte chunks are XOR-swizzled through the SOURCE address (the DMA
//           destination is lane-linear), one barrier per slab.  Transform-free instantiation only.
// MODE 3 : MODE 2 over the LIVE taps of the tile only: taps that no row of the tile can use -- the depth planes beyond the
//           tensor's ends in the input gradient of a temporal k=4 convolution, 23-43 % of the K loop of the UCF video
//           discriminator's layers -- are left out of a per-tile tap table (workgroup-uniform).  Its own instantiation:
//           the bookkeeping costs the plain MODE 2 loop 5-12 % when compiled in.
template <int WM, int WN, int TM, int TN, int MODE, bool XF>
__global__ void __launch_bounds__(WM* WN * 64) igemm_fast_kernel(const IgemmArgs a) {
  constexpr bool DB = MODE == 1, GL = MODE >= 2, SK = MODE == 3;
  static_assert(!(GL && XF), "the LDS-DMA path cannot transform on load");
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, LDK = GL ? 32 : 36;
  constexpr int RPP = NT / 8;
  constexpr int AP = BM / RPP, BP = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/loader mismatch");
  static_assert(!GL || RPP % 16 == 0, "swizzle key must not depend on the loader pass");
  constexpr int BUF = (BM + BN) * LDK;
  constexpr int NBUF = (DB || GL) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float smem[NBUF * BUF + BM * 5 + (MODE == 3 ? 4 + IGEMM_MAX_LIVE_TAPS : 0)];
  int* rowinfo = reinterpret_cast<int*>(smem + NBUF * BUF);
  int* outoff = rowinfo + BM * 4;
  int* tapmask = outoff + BM;      // MODE 3 [3]: per dim, bit j set <=> some row of the tile gathers inside the tensor at tap j;
  int* taptab = tapmask + 4;       //        [3] = number of live taps, taptab[] = their indices in K order

  int mblk, nblk, phase, split;
  igemm_block_id(a, mblk, nblk, phase, split);
  const PhaseGeom& P = a.G.ph[phase];
  if (mblk * BM >= P.M) return;
  if (a.stagger > 0) {   // identical workgroups started together run in lockstep; offset them by quarters of a slab period
    const int q = (blockIdx.x >> 8) & 3;
    for (int i = 0; i < q * a.stagger; ++i) __builtin_amdgcn_s_sleep(64);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int Cg = a.G.Cg, Ncols = a.G.Ncols;

  if (SK && tid < 3) tapmask[tid] = 0;
  if (SK) __syncthreads();
  for (int r = tid; r < BM; r += NT) {
    const int m = mblk * BM + r;
    int base = -1, bd = 0, bh = 0, bw = 0, oo = -1;
    if (m < P.M) {
      const int qw = m % P.Mw; int t = m / P.Mw;
      const int qh = t % P.Mh; t /= P.Mh;
      int qd, img;
      if (SK && a.dmajor > 0) { img = t % a.dmajor; qd = t / a.dmajor; }    // depth-major rows: a tile sees few depth planes
      else { qd = t % P.Md; img = t / P.Md + P.img0; }
      base = img * a.gsN;
      bd = qd * a.G.Sd + P.Od; bh = qh * a.G.Sh + P.Oh; bw = qw * a.G.Sw + P.Ow;
      oo = (((img * a.G.Xd + qd * a.G.OSd + P.Pd) * a.G.Xh + qh * a.G.OSh + P.Ph) * a.G.Xw + qw * a.G.OSw + P.Pw) *
           Ncols;
      if (SK) {
        int md = 0, mh = 0, mw = 0;
        for (int j = 0; j < P.Td; ++j) md |= ((unsigned)(bd + a.G.J * j) < (unsigned)a.G.Gd ? 1 : 0) << j;
        for (int j = 0; j < P.Th; ++j) mh |= ((unsigned)(bh + a.G.J * j) < (unsigned)a.G.Gh ? 1 : 0) << j;
        for (int j = 0; j < P.Tw; ++j) mw |= ((unsigned)(bw + a.G.J * j) < (unsigned)a.G.Gw ? 1 : 0) << j;
        atomicOr(&tapmask[0], md); atomicOr(&tapmask[1], mh); atomicOr(&tapmask[2], mw);
      }
    }
    rowinfo[r * 4 + 0] = base; rowinfo[r * 4 + 1] = bd; rowinfo[r * 4 + 2] = bh; rowinfo[r * 4 + 3] = bw;
    outoff[r] = oo;
  }
  __syncthreads();
  if (SK) {
    // the taps some row of this tile can use, in K order (a tap is live if each of its three coordinates is in range for
    // some row: a superset of the exact set, exact for the temporal ends this mode exists for)
    if (tid == 0) {
      const int tmd = tapmask[0], tmh = tapmask[1], tmw = tapmask[2];
      int n = 0;
      for (int t = 0; t < P.Td * P.Th * P.Tw; ++t) {
        const int w_ = t % P.Tw, t2 = t / P.Tw, h_ = t2 % P.Th, d_ = t2 / P.Th;
        if (((tmd >> d_) & 1) & ((tmh >> h_) & 1) & ((tmw >> w_) & 1)) taptab[n++] = t;
      }
      tapmask[3] = n;
    }
    __syncthreads();
  }

  int rbase[AP], rbd[AP], rbh[AP], rbw[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int r = (tid >> 3) + RPP * i;
    rbase[i] = rowinfo[r * 4 + 0]; rbd[i] = rowinfo[r * 4 + 1]; rbh[i] = rowinfo[r * 4 + 2]; rbw[i] = rowinfo[r * 4 + 3];
  }
  // GL: LDS position (tid&7) of row r holds K chunk (tid&7) ^ key(r), key(r) = (r>>1)&7 -- two 128-byte rows share
  // one 256-byte bank row, so a 16-lane ds_read_b128 group (rows r..r+15, same chunk) touches 16 distinct slots
  const int kchunk = GL ? (((tid & 7) ^ ((tid >> 4) & 7)) * 4) : (tid & 7) * 4;
  const int J = a.G.J, Gd = a.G.Gd, Gh = a.G.Gh, Gw = a.G.Gw;
  // MODE 3 walks the LIVE slabs only (and a K split shares those out evenly)
  const int spt = Cg >> 5;         // slabs per tap
  const int nslab_all = SK ? __builtin_amdgcn_readfirstlane(tapmask[3]) * spt : P.K >> 5;
  const int per_split = SK ? (nslab_all + a.ksplit - 1) / a.ksplit : a.slabs_per_split;
  const int s_begin = split * per_split;
  const int s_last = a.ksplit > 1 ? (s_begin + per_split < nslab_all ? s_begin + per_split : nslab_all) : nslab_all;
  const int nslab = s_last > s_begin ? s_last - s_begin : 0;   // slabs of THIS workgroup
  // weight rows of this thread (clamped; rows >= Ncols are masked to zero)
  const float* wrow[BP];
  bool wok[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int n = nblk * BN + (tid >> 3) + RPP * i;
    wok[i] = n < Ncols;
    wrow[i] = a.w + P.w_off + (int64_t)(wok[i] ? n : Ncols - 1) * P.Kp + kchunk + (SK ? 0 : s_begin * 32);
  }
  const bool xf = a.scale != nullptr;
  const float neg = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);

  // wave-uniform tap odometer for the slab being fetched (starts at this workgroup's first slab)
  int c0, jd, jh, jw;
  int li = 0, wk = 0;      // MODE 3: index into the live-tap table; K offset of the odometer inside the packed panel
  auto live_tap = [&]() {
    const int t = __builtin_amdgcn_readfirstlane(taptab[li]);
    jw = t % P.Tw; const int t2 = t / P.Tw; jh = t2 % P.Th; jd = t2 / P.Th;
    wk = t * Cg;
  };
  if (SK) {
    li = s_begin / spt;
    c0 = (s_begin - li * spt) * 32;
    if (nslab > 0) live_tap(); else { jd = jh = jw = 0; }
  } else {
    const int k0 = s_begin * 32, tap0 = k0 / Cg;
    c0 = k0 - tap0 * Cg;
    jw = tap0 % P.Tw; const int t2 = tap0 / P.Tw; jh = t2 % P.Th; jd = t2 / P.Th;
  }
  f32x4 ra[AP], rb[BP], sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  unsigned amask = 0;
  int roff[AP];          // element offset of (row, current tap, channel 0); refreshed only when the tap changes
  bool tap_dirty = true;
  // GL: both operands are raw buffers.  A lane's byte offset (row, tap, K chunk) sits in a VGPR and changes once per tap; the
  // per-slab advance is the instruction's SCALAR offset, so issuing a slab costs one s_mov m0 + one buffer_load ... lds per
  // 1 KiB piece and no vector arithmetic.  Padding taps and tail rows carry an offset beyond the buffer: the hardware's
  // range check returns zeros for them (no zero page, no select).
  BufRsrc rsrcA, rsrcB;
  unsigned voffB[BP];
  if (GL) {
    rsrcA = make_buf_rsrc(a.src, a.src_bytes);
    rsrcB = make_buf_rsrc(a.w, a.w_bytes);
#pragma unroll
    for (int i = 0; i < BP; ++i) voffB[i] = (unsigned)((wrow[i] - a.w) * 4);
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  auto tap_refresh = [&]() {
    if (tap_dirty) {     // wave-uniform: once per tap (every Cg/32 slabs), not per slab
      amask = 0;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int id = rbd[i] + J * jd, ih = rbh[i] + J * jh, iw = rbw[i] + J * jw;
        const bool ok = rbase[i] >= 0 && (unsigned)id < (unsigned)Gd && (unsigned)ih < (unsigned)Gh && (unsigned)iw < (unsigned)Gw;
        roff[i] = ok ? rbase[i] + id * a.gsD + ih * a.gsH + iw * a.gsW + kchunk : 0;
        amask |= (ok ? 1u : 0u) << i;
      }
      tap_dirty = false;
    }
  };
  auto advance = [&]() {   // the tap odometer (uniform)
    c0 += 32;
    if (c0 >= Cg) {
      c0 = 0;
      tap_dirty = true;
      if (SK) { ++li; live_tap(); }      // (one entry past the table's end is read after the last slab and never used)
      else if (++jw == P.Tw) { jw = 0; if (++jh == P.Th) { jh = 0; ++jd; } }
    }
  };
  auto fetch = [&](int slab) {
    const int c = c0 + kchunk;
    if (XF && xf) { sc4 = *reinterpret_cast<const f32x4*>(a.scale + c); sh4 = *reinterpret_cast<const f32x4*>(a.shift + c); }
    tap_refresh();
#pragma unroll
    for (int i = 0; i < AP; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a.src + roff[i] + (((amask >> i) & 1u) ? c0 : 0));
#pragma unroll
    for (int i = 0; i < BP; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wrow[i] + slab * 32);
    advance();
  };
  // LDS-DMA issue of one slab into buffer `buf`: wave w fills rows 8w..8w+7 (+RPP per pass), 1 KiB per instruction
  auto dma = [&](int slab, int buf) {
    if (tap_dirty) {
      tap_refresh();
#pragma unroll
      for (int i = 0; i < AP; ++i) roff[i] = ((amask >> i) & 1u) ? roff[i] * 4 : (int)0x80000000;     // (GL: bytes from here on)
    }
    float* dstA = smem + buf * BUF + wave_u * 8 * LDK;
    float* dstB = dstA + BM * LDK;
#pragma unroll
    for (int i = 0; i < AP; ++i)
      buf_dma16(rsrcA, dstA + RPP * i * LDK, roff[i], c0 * 4);
    const int kb = (SK ? wk + c0 : slab * 32) * 4;
#pragma unroll
    for (int i = 0; i < BP; ++i)
      buf_dma16(rsrcB, dstB + RPP * i * LDK, (int)voffB[i], kb);
    advance();
  };
  auto stageA = [&](int buf) {
    float* As = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      f32x4 v;
      const bool ok = (amask >> i) & 1u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = ra[i][e];
        if (XF) { t = t * sc4[e] + sh4[e]; t = fmaxf(t, t * neg); }   // neg in [0,1]: max(t, neg*t) == act(t)
        v[e] = ok ? t : 0.f;
      }
      *reinterpret_cast<f32x4*>(As + ((tid >> 3) + RPP * i) * LDK + kchunk) = v;
    }
  };
  auto stageB = [&](int buf) {
    float* Bs = smem + buf * BUF + BM * LDK;
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      f32x4 v = rb[i];
      if (!wok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(Bs + ((tid >> 3) + RPP * i) * LDK + kchunk) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31, frag_k = GL ? 0 : (lane >> 5) * 4;
  const int fkey = (frag_row >> 1) & 7, fhalf = lane >> 5;
  const float* Abase = smem + (wm * TM * 32 + frag_row) * LDK + frag_k;
  const float* Bbase = smem + BM * LDK + (wn * TN * 32 + frag_row) * LDK + frag_k;
  auto mma_group = [&](int buf, int kg) {
    f32x4 af[TM], bf[TN];
    const int koff = GL ? (((2 * kg + fhalf) ^ fkey) * 4) : kg * 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Abase + buf * BUF + i * 32 * LDK + koff);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bbase + buf * BUF + j * 32 * LDK + koff);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
  };

  if (GL) {
    // Fragments run one MFMA group ahead of the matrix pipe, and the slab barrier sits BEFORE the last group of a slab:
    // by then this wave has read all its fragments of the slab (so the buffer may be refilled once everyone is past the
    // barrier) and its share of the next slab has landed; the last group's 16 MFMAs then cover the barrier's skew and the
    // first fragment reads of the next slab -- the matrix pipe never waits for LDS or for the DMA issue.
    f32x4 fa[2][TM], fb[2][TN];
    auto frag = [&](int buf, int kg, int set) {
      const int koff = ((2 * kg + fhalf) ^ fkey) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(Abase + buf * BUF + i * 32 * LDK + koff);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(Bbase + buf * BUF + j * 32 * LDK + koff);
    };
    auto mma = [&](int set) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][e], fb[set][j][e], acc[i][j], 0, 0, 0);
    };
    if (nslab > 0) {
      dma(0, 0);
      __syncthreads();                        // (waits vmcnt(0)) slab 0 has landed
      frag(0, 0, 0);
#ifdef GODE_ABLATION_BUILD   // timing-only bodies (WRONG results); never defined for libgode.so (scripts/exp/build_ablation.sh)
      if (a.stagger < 0) {
        const bool rd = a.stagger >= -2 || a.stagger == -4;      // -3: MFMAs only; -1: + fragment reads; -2: + barrier; -4: all but the barrier
        for (int s = 0; s < nslab; ++s) {
          const int buf = s & 1;
          if (a.stagger == -4 && s + 1 < nslab) dma(s + 1, buf ^ 1);
          if (rd) frag(buf, 1, 1);
          __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0);
          if (rd) frag(buf, 2, 0);
          __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0);
          if (rd) frag(buf, 3, 1);
          __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0);
          if (a.stagger == -2) __syncthreads();
          if (rd) frag(buf ^ 1, 0, 0);
          __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0);
        }
      } else
#endif
      for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) dma(s + 1, buf ^ 1);   // every wave is past the barrier below: nobody reads that buffer any more
        // (sched_barrier: the compiler otherwise sinks each fragment read to just before its first use, i.e. behind the
        // MFMA group it is meant to run under)
        frag(buf, 1, 1); __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0);
        frag(buf, 2, 0); __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0);
        frag(buf, 3, 1); __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                      // lgkmcnt(0) + vmcnt(0) + barrier: my reads of slab s are done, slab s+1 has landed
        frag(buf ^ 1, 0, 0);                  // (after the last slab: a stale buffer, never used)
        __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                          // the epilogue reuses the buffers
  } else if (nslab > 0) {
  fetch(0);
  if (DB) {
    stageA(0);
    stageB(0);
    __syncthreads();
    for (int s = 0; s + 1 < nslab; ++s) {
      const int buf = s & 1;
      fetch(s + 1);
      mma_group(buf, 0);
      mma_group(buf, 1);
      stageA(buf ^ 1);
      mma_group(buf, 2);
      stageB(buf ^ 1);
      mma_group(buf, 3);
      __syncthreads();
    }
    const int buf = (nslab - 1) & 1;
    mma_group(buf, 0); mma_group(buf, 1); mma_group(buf, 2); mma_group(buf, 3);
    __syncthreads();
#ifdef GODE_ABLATION_BUILD   // timing-only bodies (WRONG results); never defined for libgode.so (gan-ode_amd/build.py)
  } else if (a.stagger == -1) {        // ABLATION (timing only, wrong results): MFMA + LDS reads, no staging, no barriers
    stageA(0); stageB(0);
    __syncthreads();
    for (int s = 0; s + 1 < nslab; ++s) { mma_group(0, 0); mma_group(0, 1); mma_group(0, 2); mma_group(0, 3); }
  } else if (a.stagger == -2) {        // ABLATION: + barriers, still no staging
    stageA(0); stageB(0);
    __syncthreads();
    for (int s = 0; s + 1 < nslab; ++s) { __syncthreads(); mma_group(0, 0); mma_group(0, 1); mma_group(0, 2); mma_group(0, 3); __syncthreads(); }
  } else if (a.stagger == -3) {        // ABLATION: + global fetch, no LDS staging
    stageA(0); stageB(0);
    __syncthreads();
    for (int s = 0; s + 1 < nslab; ++s) { __syncthreads(); fetch(s + 1); mma_group(0, 0); mma_group(0, 1); mma_group(0, 2); mma_group(0, 3); __syncthreads(); }
    asm volatile("" :: "v"(ra[0][0]), "v"(rb[0][0]));
#endif
  } else {
    for (int s = 0; s + 1 < nslab; ++s) {
      stageA(0);
      stageB(0);
      __syncthreads();
      fetch(s + 1);
      mma_group(0, 0); mma_group(0, 1); mma_group(0, 2); mma_group(0, 3);
      __syncthreads();
    }
    stageA(0);
    stageB(0);
    __syncthreads();
    mma_group(0, 0); mma_group(0, 1); mma_group(0, 2); mma_group(0, 3);
    __syncthreads();
  }
  }

  const int ccol = lane & 31, crow = 4 * (lane >> 5);
  const bool partial = a.ksplit > 1;
  float* const dst = partial ? a.work + (int64_t)split * a.out_numel : a.out;
  const bool tanh_out = !partial && a.epilogue == GODE_EPI_TANH;
  if (mblk * BM + BM <= P.M && nblk * BN + BN <= Ncols && !tanh_out) {
    // interior tile (workgroup-uniform; all but the edge tiles): no per-element predicates, and the four row offsets
    // of an accumulator quad come from one ds_read_b128.  The predicated form below costs ~600 instructions per
    // wave -- 9 % of a 16-slab tile.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row0 = (wm * TM + i) * 32 + 8 * q + crow;
        const int o0 = outoff[row0], o1 = outoff[row0 + 1], o2 = outoff[row0 + 2], o3 = outoff[row0 + 3];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = nblk * BN + (wn * TN + j) * 32 + ccol;
          dst[o0 + col] = acc[i][j][4 * q + 0];
          dst[o1 + col] = acc[i][j][4 * q + 1];
          dst[o2 + col] = acc[i][j][4 * q + 2];
          dst[o3 + col] = acc[i][j][4 * q + 3];
        }
      }
  } else {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = nblk * BN + (wn * TN + j) * 32 + ccol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + crow;
        const int oo = outoff[row];
        if (oo >= 0 && col < Ncols) {
          float v = acc[i][j][r];
          if (tanh_out) v = tanhf(v);
          dst[oo + col] = v;
        }
      }
    }
  }

  if (a.stats != nullptr && !partial) {
    float* sred = smem;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (lane < 32) {
        const int c = (wn * TN + j) * 32 + lane;
        sred[(wm * BN + c) * 2 + 0] = s1;
        sred[(wm * BN + c) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += NT) {
      const int col = nblk * BN + c;
      if (col < Ncols) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += sred[(w * BN + c) * 2 + 0]; s2 += sred[(w * BN + c) * 2 + 1]; }
        // [which][column][row]: the finalize kernel then walks contiguous rows per channel
        a.stats[(int64_t)col * a.stats_rows + P.row0 + mblk] = s1;
        a.stats[((int64_t)Ncols + col) * a.stats_rows + P.row0 + mblk] = s2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
#define SPLITK_ROWS 16
// split-K finish: sums the partial tiles in fixed order, applies the epilogue and produces the BatchNorm partial sums
// (one stats row per SPLITK_ROWS output positions, so that even a 512-position output spreads over 32 workgroups).  Threads own columns, so every sum has a fixed order.
__global__ void __launch_bounds__(256) igemm_splitk_reduce_kernel(const float* work, float* out, float* stats, int positions,
                                                                  int Ncols, int ksplit, int epilogue) {
  const int row0 = blockIdx.x * SPLITK_ROWS;
  const int nrows = positions - row0 < SPLITK_ROWS ? positions - row0 : SPLITK_ROWS;
  const int64_t numel = (int64_t)positions * Ncols;
  for (int col = threadIdx.x; col < Ncols; col += 256) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int r = 0; r < nrows; ++r) {
      const int64_t i = (int64_t)(row0 + r) * Ncols + col;
      float v = 0.f;
#pragma unroll 4
      for (int sp = 0; sp < ksplit; ++sp) v += work[sp * numel + i];
      s1 += v; s2 += v * v;
      out[i] = epilogue == GODE_EPI_TANH ? tanhf(v) : v;
    }
    if (stats) {
      stats[(int64_t)col * gridDim.x + blockIdx.x] = s1;
      stats[((int64_t)Ncols + col) * gridDim.x + blockIdx.x] = s2;
    }
  }
}

// Vector form (Ncols % 4 == 0): a workgroup owns 16 positions x 64 columns, one float4 per thread, so even a
// 512-position x 512-column output spreads over 256 workgroups with ksplit independent 16-byte loads per thread (the
// column-loop form above ran 32 workgroups and took 84 us for 34 MB of partials).  Same summation orders as above.
__global__ void __launch_bounds__(256) igemm_splitk_reduce_vec_kernel(const float* work, float* out, float* stats, int positions,
                                                                      int Ncols, int ksplit, int epilogue) {
  __shared__ float red[2][SPLITK_ROWS][64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int row = blockIdx.x * SPLITK_ROWS + ty, col = blockIdx.y * 64 + tx * 4;
  const int64_t numel = (int64_t)positions * Ncols;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row < positions && col < Ncols) {
    const float* p = work + (int64_t)row * Ncols + col;
#pragma unroll 8
    for (int sp = 0; sp < ksplit; ++sp) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + sp * numel);
      v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    f32x4 o = v;
    if (epilogue == GODE_EPI_TANH) { o[0] = tanhf(v[0]); o[1] = tanhf(v[1]); o[2] = tanhf(v[2]); o[3] = tanhf(v[3]); }
    *reinterpret_cast<f32x4*>(out + (int64_t)row * Ncols + col) = o;
  }
  if (stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][ty][tx * 4 + e] = v[e]; red[1][ty][tx * 4 + e] = v[e] * v[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < SPLITK_ROWS; ++r) acc += red[which][r][c];
      if (blockIdx.y * 64 + c < Ncols) stats[((int64_t)which * Ncols + blockIdx.y * 64 + c) * gridDim.x + blockIdx.x] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Thin outputs (<= 4 columns) with a long K and few rows: the discriminators' last layers (Conv3d 512->1 k2,
// Conv2d 256->1 k4; models/mocogan.py:88,158).  A 128x32 MFMA tile would run ~128 serial K slabs in a handful of
// workgroups; here one wave owns one output position, its 64 lanes stride over K with float4 gathers and the
// partial dot products are combined with wave shuffles.
struct DotArgs { IgemmArgs a; FastDiv dCg; int32_t per_block; };   // per_block: 4 = one position per wave; 1 = the four
                                                                  // waves of a workgroup split the K of ONE position
__global__ void __launch_bounds__(256) conv_dot_kernel(const DotArgs d) {
  const IgemmArgs& a = d.a;
  const PhaseGeom& P = a.G.ph[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool share = d.per_block == 1;      // few positions with a very long K (UCF video-D last layer: 16 x 32768)
  const int m = share ? blockIdx.x : blockIdx.x * 4 + wave;
  __shared__ float part[4][4];
  if (m >= P.M) return;
  const int Cg = a.G.Cg, Ncols = a.G.Ncols;
  const int qw = m % P.Mw; int t = m / P.Mw;
  const int qh = t % P.Mh; t /= P.Mh;
  const int qd = t % P.Md; const int img = t / P.Md;
  const int base = img * a.gsN;
  const int bd = qd * a.G.Sd + P.Od, bh = qh * a.G.Sh + P.Oh, bw = qw * a.G.Sw + P.Ow;
  const float neg = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);
  const bool xf = a.scale != nullptr;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* wp = a.w + P.w_off;
  for (int k = lane * 4 + (share ? wave * 256 : 0); k < P.K; k += (share ? 1024 : 256)) {
    const int tap = (int)fdiv((uint32_t)k, d.dCg), c = k - tap * Cg;
    const int jw = tap % P.Tw, t2 = tap / P.Tw, jh = t2 % P.Th, jd = t2 / P.Th;
    const int id = bd + a.G.J * jd, ih = bh + a.G.J * jh, iw = bw + a.G.J * jw;
    const bool ok = (unsigned)id < (unsigned)a.G.Gd && (unsigned)ih < (unsigned)a.G.Gh && (unsigned)iw < (unsigned)a.G.Gw;
    if (ok) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.src + base + id * a.gsD + ih * a.gsH + iw * a.gsW + c);
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (xf) { sc = *reinterpret_cast<const f32x4*>(a.scale + c); sh = *reinterpret_cast<const f32x4*>(a.shift + c); }
#pragma unroll
      for (int e = 0; e < 4; ++e) { float u = v[e] * sc[e] + sh[e]; v[e] = u > 0.f ? u : u * neg; }
      for (int n = 0; n < Ncols; ++n) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wp + (int64_t)n * P.Kp + k);
        acc[n] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
      }
    }
  }
  const int oo = (((img * a.G.Xd + qd * a.G.OSd + P.Pd) * a.G.Xh + qh * a.G.OSh + P.Ph) * a.G.Xw + qw * a.G.OSw + P.Pw) * Ncols;
  for (int n = 0; n < Ncols; ++n) {
    float s = acc[n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (!share) {
      if (lane == 0) a.out[oo + n] = a.epilogue == GODE_EPI_TANH ? tanhf(s) : s;
    } else if (lane == 0) part[wave][n] = s;
  }
  if (share) {               // (all four waves of the workgroup work on the same m, so none has returned early)
    __syncthreads();
    if (threadIdx.x < Ncols) {
      const float s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
      a.out[oo + threadIdx.x] = a.epilogue == GODE_EPI_TANH ? tanhf(s) : s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Layers with <= 4 output columns and many positions: the generators' last layers (MNIST: ConvTranspose2d 64 -> 1,
// k1, a crop, models/mocogan_ode.py:82; UCF: ConvTranspose2d 64 -> 3, k4 s2, models/mocogan.py:213) with the previous
// BatchNorm+ReLU fused on load and the tanh on store, and the input gradient of the first discriminator layer.  A
// 32-column MFMA tile would be >= 87 % padding; these are streaming passes: Cg/4 lanes share one output position
// (float4 of channels each, all taps in a loop, shuffle reduction at the end), 64/(Cg/4) positions per
// wave-iteration, grid-stride, one grid row per stride phase, the phase's weight panel (<= 16 KB) in LDS.
// MNIST head 44 -> 28 us; UCF head 224 -> ~65 us.
#define PW_MAXW 4096
struct PwArgs { IgemmArgs a; FastDiv dMw[GODE_MAX_PHASES], dMh[GODE_MAX_PHASES], dMd[GODE_MAX_PHASES]; int32_t lpp; };

// ONE_TAP (1x1x1 layers, e.g. the MNIST head): the lane's weights live in registers, no LDS, no tap loop (28 us vs 34 us
// through the general form on the 134 MB MNIST head input).
template <bool ONE_TAP>
__global__ void __launch_bounds__(256) conv_pointwise_kernel(const PwArgs d) {
  const IgemmArgs& a = d.a;
  const int ph = blockIdx.y;
  const PhaseGeom& P = a.G.ph[ph];
  const int Ncols = a.G.Ncols, Cg = a.G.Cg;
  __shared__ __attribute__((aligned(16))) float wl[ONE_TAP ? 4 : PW_MAXW];      // [column][Kp]
  if (!ONE_TAP) {
    for (int i = threadIdx.x; i < Ncols * P.Kp; i += 256) wl[i] = a.w[P.w_off + i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, lpp = d.lpp, ppw = 64 / lpp;
  const int sub = lane / lpp, ch = (lane - sub * lpp) * 4;
  const float neg = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (a.scale) { sc = *reinterpret_cast<const f32x4*>(a.scale + ch); sh = *reinterpret_cast<const f32x4*>(a.shift + ch); }
  const int ntap = ONE_TAP ? 1 : P.Td * P.Th * P.Tw;
  f32x4 w1[4];
  if (ONE_TAP) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
      w1[n] = n < Ncols ? *reinterpret_cast<const f32x4*>(a.w + P.w_off + (int64_t)n * P.Kp + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t m0 = wave_id * ppw; m0 < P.M; m0 += nwaves * ppw) {
    const int m = (int)m0 + sub;
    const bool live = m < P.M;
    const uint32_t mm = live ? (uint32_t)m : 0u;
    const uint32_t t1 = fdiv(mm, d.dMw[ph]), qw = mm - t1 * P.Mw;
    const uint32_t t2 = fdiv(t1, d.dMh[ph]), qh = t1 - t2 * P.Mh;
    const uint32_t img = fdiv(t2, d.dMd[ph]), qd = t2 - img * P.Md;
    const int bd = (int)qd * a.G.Sd + P.Od, bh = (int)qh * a.G.Sh + P.Oh, bw = (int)qw * a.G.Sw + P.Ow;
    const int base = (int)img * a.gsN;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int jd = 0, jh = 0, jw = 0;
    for (int t = 0; t < ntap; ++t) {
      const int id = bd + a.G.J * jd, ih = bh + a.G.J * jh, iw = bw + a.G.J * jw;
      const bool ok = live && (unsigned)id < (unsigned)a.G.Gd && (unsigned)ih < (unsigned)a.G.Gh && (unsigned)iw < (unsigned)a.G.Gw;
      if (ok) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a.src + base + id * a.gsD + ih * a.gsH + iw * a.gsW + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float u = v[e] * sc[e] + sh[e]; v[e] = fmaxf(u, u * neg); }
        const float* wk = wl + t * Cg + ch;
#pragma unroll
        for (int n = 0; n < 4; ++n)
          if (ONE_TAP || n < Ncols) {
            const f32x4 w = ONE_TAP ? w1[n] : *reinterpret_cast<const f32x4*>(wk + n * P.Kp);
            acc[n] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
          }
      }
      if (++jw == P.Tw) { jw = 0; if (++jh == P.Th) { jh = 0; ++jd; } }
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) {
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] += __shfl_xor(acc[n], o);
    }
    if (live && ch == 0) {
      const int oo = ((((int)img * a.G.Xd + (int)qd * a.G.OSd + P.Pd) * a.G.Xh + (int)qh * a.G.OSh + P.Ph) * a.G.Xw +
                      (int)qw * a.G.OSw + P.Pw) * Ncols;
      for (int n = 0; n < Ncols; ++n) a.out[oo + n] = a.epilogue == GODE_EPI_TANH ? tanhf(acc[n]) : acc[n];
    }
  }
}

// Tiny-K layers (K = taps x channels, one phase; dispatched for K <= 4): the backward-data of the generator head (Conv
// 1 -> 64, k1: K = 1) writes 64x more than it reads, so it is an output-bound HBM pass: the weight matrix sits in LDS as
// [k][column], Ncols/4 lanes share a position (each gathers the K inputs itself -- same addresses, one broadcast -- and
// owns 4 columns), grid-stride over positions.  Any source strides.  63 us on the padded-K MFMA path, 27 us here.
__global__ void __launch_bounds__(256) conv_smallk_kernel(const PwArgs d) {
  const IgemmArgs& a = d.a;
  const PhaseGeom& P = a.G.ph[0];
  const int Ncols = a.G.Ncols, Cg = a.G.Cg, K = P.K;
  __shared__ __attribute__((aligned(16))) float wl[32 * 128];
  for (int i = threadIdx.x; i < K * Ncols; i += 256) {
    const int k = i / Ncols, n = i - k * Ncols;
    wl[i] = a.w[P.w_off + (int64_t)n * P.Kp + k];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, lpp = d.lpp, ppw = 64 / lpp;
  const int sub = lane / lpp, c4 = (lane - sub * lpp) * 4;
  const float neg = a.act == GODE_ACT_RELU ? 0.f : (a.act == GODE_ACT_LRELU ? 0.2f : 1.f);
  const bool xf = a.scale != nullptr;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t m0 = wave_id * ppw; m0 < P.M; m0 += nwaves * ppw) {
    const int m = (int)m0 + sub;
    if (m >= P.M) continue;
    const uint32_t t1 = fdiv((uint32_t)m, d.dMw[0]), qw = m - t1 * P.Mw;
    const uint32_t t2 = fdiv(t1, d.dMh[0]), qh = t1 - t2 * P.Mh;
    const uint32_t img = fdiv(t2, d.dMd[0]), qd = t2 - img * P.Md;
    const int bd = (int)qd * a.G.Sd + P.Od, bh = (int)qh * a.G.Sh + P.Oh, bw = (int)qw * a.G.Sw + P.Ow;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (int jd = 0; jd < P.Td; ++jd)
      for (int jh = 0; jh < P.Th; ++jh)
        for (int jw = 0; jw < P.Tw; ++jw) {
          const int id = bd + a.G.J * jd, ih = bh + a.G.J * jh, iw = bw + a.G.J * jw;
          const bool ok = (unsigned)id < (unsigned)a.G.Gd && (unsigned)ih < (unsigned)a.G.Gh && (unsigned)iw < (unsigned)a.G.Gw;
          const int off = (int)img * a.gsN + id * a.gsD + ih * a.gsH + iw * a.gsW;
          for (int c = 0; c < Cg; ++c, ++k) {
            float v = 0.f;
            if (ok) {
              v = a.src[off + c * a.gsC];
              if (xf) v = v * a.scale[c] + a.shift[c];
              v = fmaxf(v, v * neg);
            }
            const f32x4 w = *reinterpret_cast<const f32x4*>(wl + k * Ncols + c4);
            acc[0] += v * w[0]; acc[1] += v * w[1]; acc[2] += v * w[2]; acc[3] += v * w[3];
          }
        }
    const int oo = ((((int)img * a.G.Xd + (int)qd * a.G.OSd + P.Pd) * a.G.Xh + (int)qh * a.G.OSh + P.Ph) * a.G.Xw +
                    (int)qw * a.G.OSw + P.Pw) * Ncols;
    if (a.epilogue == GODE_EPI_TANH) { acc[0] = tanhf(acc[0]); acc[1] = tanhf(acc[1]); acc[2] = tanhf(acc[2]); acc[3] = tanhf(acc[3]); }
    *reinterpret_cast<f32x4*>(a.out + oo + c4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
enum { TILE_128x128 = 1, TILE_128x64 = 2, TILE_128x32 = 3, TILE_64x64 = 4, TILE_128x128_W8 = 5, TILE_256x128 = 6 };

static int tile_bm(int tile) { return tile == TILE_64x64 ? 64 : (tile == TILE_256x128 ? 256 : 128); }
static int tile_bn(int tile) { return (tile == TILE_128x128 || tile == TILE_128x128_W8 || tile == TILE_256x128) ? 128 : tile == TILE_128x32 ? 32 : 64; }

static int pick_tile(const IgemmGeom& G, int requested) {
  requested %= 10;   // +10: double-LDS-buffer variant of the same tile (tuning knob; default is single-buffer)
  if (requested >= TILE_128x128 && requested <= TILE_256x128) return requested;
  if (G.Ncols <= 32) return TILE_128x32;
  if (G.Ncols <= 64) return TILE_128x64;
  int64_t blocks = 0;
  for (int i = 0; i < G.nphase; ++i) blocks += (int64_t)gode_ceil_div(G.ph[i].M, 128) * gode_ceil_div(G.Ncols, 128);
  return blocks >= 192 ? TILE_128x128 : TILE_64x64;
}

// split-K plan: purely a function of the geometry (so gode_igemm_stats_rows / gode_igemm_work_size agree with the
// launch).  Used when the FAST kernel applies and the natural grid would leave most of the 256 CUs idle.
struct SplitPlan { int ksplit, slabs_per_split, positions; };

static bool fast_geometry(const IgemmGeom& G) {
  bool ok = G.Cg % 32 == 0;
  for (int i = 0; i < G.nphase; ++i) ok = ok && G.ph[i].K >= 32 && G.ph[i].K % 32 == 0 && G.ph[i].Kp == G.ph[i].K;
  return ok;
}

static bool strides_allow_vec(const gode_igemm_op* op, const IgemmGeom& G) {
  if (gode_strides_are_channels_last(op->gs)) return G.Cg % 4 == 0;
  return op->gs[4] == 1 && G.Cg % 4 == 0 && op->gs[0] % 4 == 0 && op->gs[1] % 4 == 0 && op->gs[2] % 4 == 0 &&
         op->gs[3] % 4 == 0;
}

static SplitPlan plan_split(const gode_igemm_op* op, const IgemmGeom& G, int tile) {
  const gode_conv_geom& g = op->g;
  SplitPlan sp; sp.ksplit = 1; sp.slabs_per_split = 0; sp.positions = g.N * G.Xd * G.Xh * G.Xw;
  if (!fast_geometry(G) || G.Ncols <= 4 || !strides_allow_vec(op, G) || op->tile >= 10) return sp;
  const int bm = tile == 4 ? 64 : 128, bn = (tile == 1 || tile == 5) ? 128 : (tile == 3 ? 32 : 64);
  int64_t blocks = 0; int max_slabs = 0, min_slabs = 1 << 30;
  for (int i = 0; i < G.nphase; ++i) {
    blocks += (int64_t)gode_ceil_div(G.ph[i].M, bm) * gode_ceil_div(G.Ncols, bn);
    const int ns = G.ph[i].K >> 5;
    if (ns > max_slabs) max_slabs = ns;
    if (ns < min_slabs) min_slabs = ns;
  }
  if (blocks >= 64 || min_slabs < 8) return sp;   // only grids that would leave >3/4 of the CUs idle
  int k = (int)((512 + blocks - 1) / blocks);
  if (k > min_slabs / 4) k = min_slabs / 4;
  if (k > 32) k = 32;
  if (k < 2) return sp;
  sp.slabs_per_split = gode_ceil_div(max_slabs, k);
  sp.ksplit = gode_ceil_div(max_slabs, sp.slabs_per_split);
  return sp;
}

static int pick_tile_split_aware(const gode_conv_geom& g, const IgemmGeom& G, int requested) {
  const int t = pick_tile(G, requested);
  if (requested % 10 != 0 || !fast_geometry(G) || G.Ncols <= 4) return t;
  // with split-K available a starved grid is better served by the big tile (unless there are very few rows)
  int maxM = 0;
  for (int i = 0; i < G.nphase; ++i) if (G.ph[i].M > maxM) maxM = G.ph[i].M;
  int64_t blocks = 0;
  for (int i = 0; i < G.nphase; ++i) blocks += (int64_t)gode_ceil_div(G.ph[i].M, 128) * gode_ceil_div(G.Ncols, 128);
  if (t == TILE_64x64 && maxM > 64 && blocks < 64) return G.Ncols <= 64 ? TILE_128x64 : TILE_128x128;
  (void)g;
  return t;
}

// Tile and K-split of a FAST-path GEMM from a small cost model instead of thresholds.  All workgroups of a launch
// carry equal work and the dispatcher deals them evenly, so the launch lasts as long as the busiest CU:
//   time = ceil(workgroups / 256) * (fixed + slabs * cycles_per_slab(tile) / efficiency) [+ split-K reduce]
// with the MFMA cycles of one 32-deep slab (64 per 32x32x2 instruction), the measured pipe efficiencies of the tiles
// (two or more resident workgroups per CU overlap each other's barriers; a lone one cannot) and ~4k cycles of
// prologue + epilogue.  Checked against measurements: ConvT 128->64 (4096 x 128x64) 314 us model / 297 us measured;
// UCF video-D layer-2 dgrad 128x128 578 / 514 us, for which the model prefers 128x64 (436 us).
// MODE 3 of the FAST kernel (live taps only) pays for its slower loop when the temporal taps that fall off the tensor's
// ends are > 15 % of the K loop (temporal stride 1: Do of the Di planes' taps are live)
static bool igemm_wants_tapskip(const gode_igemm_op* op, const IgemmGeom& G) {
  if (!(op->dir == GODE_DGRAD && op->groups != 2 && !G.fullk && op->g.sd == 1 && G.ph[0].Td > 1 && 100 * op->g.Do < 85 * op->g.Di))
    return false;
  for (int i = 0; i < G.nphase; ++i) if (G.ph[i].Td * G.ph[i].Th * G.ph[i].Tw >= IGEMM_MAX_LIVE_TAPS) return false;
  return true;
}

static void choose_fast(const gode_igemm_op* op, const IgemmGeom& G, int* tile_out, SplitPlan* sp_out, double* cycles_out = nullptr) {
  const gode_conv_geom& g = op->g;
  const int positions = g.N * G.Xd * G.Xh * G.Xw;
  const double out_bytes = 4.0 * positions * G.Ncols;
  int max_slabs = 0, min_slabs = 1 << 30;
  bool even_phases = true;        // phases of unequal size leave early-exit workgroups in the grid: measured, a K split
  for (int i = 0; i < G.nphase; ++i) {   // then gains nothing (video-D layer 3 dgrad: 101 us split 3 ways vs 64 us unsplit)
    const int ns = G.ph[i].K >> 5;
    if (ns > max_slabs) max_slabs = ns;
    if (ns < min_slabs) min_slabs = ns;
    even_phases = even_phases && G.ph[i].M == G.ph[0].M;
  }
  // 256x128 (8 waves, one workgroup per CU, 25 % less operand traffic per MFMA) last: 2-3 % ahead of 128x128 on the full
  // decoder grids and 5 % on the MNIST video-D layer-2 input gradient (scripts/sweep_tiles.py), behind it on grids that
  // do not fill whole rounds of 256 such tiles
  static const int tiles[5] = {TILE_128x128, TILE_128x64, TILE_64x64, TILE_128x32, TILE_256x128};
  static const int ksplits[10] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  const bool live_taps = igemm_wants_tapskip(op, G);
  double best = 1e300;
  int best_tile = G.Ncols <= 32 ? TILE_128x32 : TILE_128x64, best_k = 1;
  for (int ti = 0; ti < 5; ++ti) {
    const int t = tiles[ti], bm = tile_bm(t), bn = tile_bn(t);
    if (bn >= 2 * G.Ncols && bn > 32) continue;              // more than half of every tile would be padding
    if (t == TILE_128x32 && G.Ncols > 32) continue;
    if (t == TILE_256x128 && (live_taps || op->groups == 2)) continue;     // (measured on plain launches only)
    const double cps = t == TILE_256x128 ? 8192.0 : (t == TILE_128x128 ? 4096.0 : (t == TILE_128x64 ? 2048.0 : 1024.0));
    const int resident = t == TILE_256x128 ? 1 : (t == TILE_128x128 ? 2 : (t == TILE_128x64 ? 3 : 4));
    const double eff2 = t == TILE_64x64 ? 0.70 : (t == TILE_128x32 ? 0.62 : (t == TILE_256x128 ? 0.785 : 0.76));   // (64x64 re-measured with the round-3 loop: 0.62 before)
    int64_t tiles_n = 0;
    for (int i = 0; i < G.nphase; ++i) tiles_n += (int64_t)gode_ceil_div(G.ph[i].M, bm) * gode_ceil_div(G.Ncols, bn);
    if (t == TILE_256x128 && tiles_n < 192) continue;        // one workgroup per CU: needs a grid that covers the chip
    for (int ki = 0; ki < 10; ++ki) {
      const int k = ksplits[ki];
      // (uneven phases: no split once the unsplit grid already covers half the chip -- see above; tiny grids, e.g. the
      // image discriminator's 7x7 <- 3x3 input gradient with 25 tiles, still gain from one: 29.5 -> ~20 us)
      if (k > 1 && (min_slabs / k < 4 || out_bytes * k > 192e6 || min_slabs != max_slabs || (!even_phases && tiles_n >= 128))) break;
      // the 256x128 tile only unsplit (measured with K splits: never ahead); live-tap launches split only grids that do
      // not fill the chip (UCF video-D layer 2 input gradient at N = 16: 269 us unsplit, 318 us split 2 ways)
      if (k > 1 && (t == TILE_256x128 || (live_taps && tiles_n >= 256))) break;
      const int64_t blocks = tiles_n * k;
      // live-tap launches: tiles of the end planes are short, so the hardware's in-order dispatch evens the CUs out and
      // the whole-rounds quantisation does not apply to the big tile (measured, UCF video-D layer 2 input gradient at
      // N=32, 640 tiles of 128x128 vs 1280 of 128x64: 10 % apart although 640 is 2.5 rounds; at N=16 the smaller tile
      // with a K split stays ahead by 5 %)
      const double per_cu = (live_taps && t == TILE_128x128 && k == 1 && blocks >= 512) ? blocks / 256.0 : (double)((blocks + 255) / 256);
      const int slabs = gode_ceil_div(max_slabs, k);
      const double eff = ((per_cu >= 2 && resident >= 2) || t == TILE_256x128) ? eff2 : 0.8 * eff2;    // (8 waves cover their own barriers)
      double cyc = per_cu * (4000.0 + slabs * cps / eff);
      if (k > 1) cyc += 2.4e9 * (2.5e-6 + (k + 1) * out_bytes / 3e12);
      if (cyc < best * 0.97) { best = cyc; best_tile = t; best_k = k; }   // 3 % hysteresis toward the earlier (larger) choice
    }
  }
  if (cycles_out) *cycles_out = best;
  if (live_taps) {
    // Live-tap launches whose 128x128 grid does not cover the chip (UCF video-D layer 3 input gradient: 224 / 112 tiles): the
    // tiles of the end planes are a quarter as long as those of the middle planes and all of them are resident at once, so the
    // launch lasts as long as the CU that drew the long ones -- the estimate above (equal tiles) is 20-30 % off there.  Small
    // tiles even that out: 4-5 workgroups of 64x64 per CU (scripts/sweep_tiles.py, N = 32 / 16: 371 -> 288 us unsplit, 209 ->
    // 179 us split 3 ways; every other live-tap shape of the configs keeps the model's choice within 2 %).
    int64_t t128 = 0, t64 = 0;
    for (int i = 0; i < G.nphase; ++i) {
      t128 += (int64_t)gode_ceil_div(G.ph[i].M, 128) * gode_ceil_div(G.Ncols, 128);
      t64 += (int64_t)gode_ceil_div(G.ph[i].M, 64) * gode_ceil_div(G.Ncols, 64);
    }
    if (t128 < 256 && G.Ncols >= 64 && min_slabs == max_slabs) {
      int k = (int)((1152 + t64 / 2) / t64);
      if (k < 1) k = 1;
      while (k > 1 && (min_slabs / k < 8 || out_bytes * k > 192e6)) --k;
      best_tile = TILE_64x64; best_k = k;
    }
  }
  *tile_out = best_tile;
  sp_out->positions = positions;
  sp_out->ksplit = 1; sp_out->slabs_per_split = 0;
  if (best_k > 1) {
    sp_out->slabs_per_split = gode_ceil_div(max_slabs, best_k);
    sp_out->ksplit = gode_ceil_div(max_slabs, sp_out->slabs_per_split);
  }
}

static int prepare(const gode_igemm_op* op, IgemmArgs* A, int* tile, int* max_mblk, int* rows, SplitPlan* sp, int* rows0 = nullptr) {
  int rc = gode_build_igemm_geom(op->g, op->dir, &A->G);
  if (rc) return rc;
  const bool grouped = op->groups == 2;
  if (grouped) {
    // two image groups with SEPARATE BatchNorm partial statistics (one discriminator pass over [real; fake]): the
    // single FPROP phase becomes two phases over the two halves of the batch, so no tile straddles the groups and
    // each group's partial-statistics rows are contiguous
    if (op->dir != GODE_FPROP || op->g.N % 2 != 0 || A->G.nphase != 1) return GODE_E_ARG;
    A->G.ph[1] = A->G.ph[0];
    A->G.ph[0].M /= 2; A->G.ph[1].M = A->G.ph[0].M; A->G.ph[1].img0 = op->g.N / 2;
    A->G.nphase = 2;
  } else if (op->groups != 0 && op->groups != 1) return GODE_E_ARG;
  static const char* menv = getenv("GODE_IGEMM_MODEL");
  static const bool sweeping = getenv("GODE_IGEMM_SWEEP") != nullptr;     // calibration runs: GODE_IGEMM_FORCE="tile,k" is re-read per call
  const char* fenv = sweeping ? getenv("GODE_IGEMM_FORCE") : nullptr;
  int ftile = 0, fk = 0;
  if (fenv && sscanf(fenv, "%d,%d", &ftile, &fk) == 2 && op->tile == 0 && fast_geometry(A->G) && A->G.Ncols > 4 &&
      strides_allow_vec(op, A->G)) {
    int max_slabs = 0;
    for (int i = 0; i < A->G.nphase; ++i) if ((A->G.ph[i].K >> 5) > max_slabs) max_slabs = A->G.ph[i].K >> 5;
    *tile = ftile;
    sp->positions = op->g.N * A->G.Xd * A->G.Xh * A->G.Xw; sp->ksplit = 1; sp->slabs_per_split = 0;
    if (fk > 1) { sp->slabs_per_split = gode_ceil_div(max_slabs, fk); sp->ksplit = gode_ceil_div(max_slabs, sp->slabs_per_split); }
  } else if (op->tile == 0 && fast_geometry(A->G) && A->G.Ncols > 4 && strides_allow_vec(op, A->G) && (!menv || atoi(menv) != 0)) {
    choose_fast(op, A->G, tile, sp);
  } else {
  *tile = pick_tile_split_aware(op->g, A->G, op->tile);
  *sp = plan_split(op, A->G, *tile);
  }
  if (grouped && sp->ksplit > 1 && (sp->positions / 2) % SPLITK_ROWS != 0) {
    // the split-K finish emits one statistics row per SPLITK_ROWS output positions: the groups must not share a row
    sp->ksplit = 1; sp->slabs_per_split = 0;
  }
  const int bm = tile_bm(*tile);
  int r0 = 0, mx = 0;
  for (int i = 0; i < A->G.nphase; ++i) {
    A->G.ph[i].row0 = r0;
    const int mb = gode_ceil_div(A->G.ph[i].M, bm);
    r0 += mb;
    if (mb > mx) mx = mb;
  }
  *max_mblk = mx;
  *rows = sp->ksplit > 1 ? gode_ceil_div(sp->positions, SPLITK_ROWS) : r0;
  if (rows0) *rows0 = !grouped ? *rows : (sp->ksplit > 1 ? *rows / 2 : A->G.ph[1].row0);
  return 0;
}

// rows of the partial statistics that belong to image group 0 (grouped FPROP; == gode_igemm_stats_rows otherwise)
extern "C" int gode_igemm_stats_rows0(const gode_igemm_op* op) {
  IgemmArgs A; int tile, mx, rows, rows0; SplitPlan sp;
  int rc = prepare(op, &A, &tile, &mx, &rows, &sp, &rows0);
  return rc ? rc : rows0;
}

// The cost model's estimate (GPU cycles) for this op's launch, or -1 when the op does not take the modelled FAST path.
// A caller that can run a batch as one launch or as two launches over its parts (ConvStack with split_images) compares.
extern "C" double gode_igemm_model_cycles(const gode_igemm_op* op) {
  if (!op || op->groups == 2 || op->tile != 0) return -1.0;
  IgemmArgs A;
  if (gode_build_igemm_geom(op->g, op->dir, &A.G)) return -1.0;
  static const char* menv = getenv("GODE_IGEMM_MODEL");
  if (!fast_geometry(A.G) || A.G.Ncols <= 4 || !strides_allow_vec(op, A.G) || (menv && atoi(menv) == 0)) return -1.0;
  int tile; SplitPlan sp; double cyc = -1.0;
  choose_fast(op, A.G, &tile, &sp, &cyc);
  return cyc;
}

extern "C" int gode_igemm_stats_segments(const gode_igemm_op* op, int32_t split_images, int32_t* seg) {
  if (!op || !seg || op->groups == 2 || split_images <= 0 || split_images >= op->g.N) return GODE_E_ARG;
  IgemmArgs A; int tile, mx, rows; SplitPlan sp;
  int rc = prepare(op, &A, &tile, &mx, &rows, &sp);
  if (rc) return rc;
  if (sp.ksplit > 1) {
    // split-K finish: one statistics row per SPLITK_ROWS output positions, positions in image-major order
    const int64_t per_img = (int64_t)A.G.Xd * A.G.Xh * A.G.Xw;
    const int64_t b = (int64_t)split_images * per_img;
    if (b % SPLITK_ROWS != 0) return GODE_E_SHAPE;
    seg[0] = 0; seg[1] = (int32_t)(b / SPLITK_ROWS); seg[2] = rows;
    return 1;
  }
  const int bm = tile_bm(tile);
  for (int i = 0; i < A.G.nphase; ++i) {
    const PhaseGeom& P = A.G.ph[i];
    const int64_t per_img = (int64_t)P.Md * P.Mh * P.Mw;       // rows of a phase are image-major
    const int64_t b = (int64_t)split_images * per_img;
    if (P.M != (int64_t)op->g.N * per_img) return GODE_E_SHAPE;
    if (b % bm != 0) return GODE_E_SHAPE;                        // a tile would hold rows of both parts
    seg[3 * i] = P.row0; seg[3 * i + 1] = P.row0 + (int32_t)(b / bm); seg[3 * i + 2] = P.row0 + gode_ceil_div(P.M, bm);
  }
  return A.G.nphase;
}

extern "C" int gode_igemm_stats_rows(const gode_igemm_op* op) {
  IgemmArgs A; int tile, mx, rows; SplitPlan sp;
  int rc = prepare(op, &A, &tile, &mx, &rows, &sp);
  return rc ? rc : rows;
}

extern "C" int64_t gode_igemm_work_size(const gode_igemm_op* op) {
  IgemmArgs A; int tile, mx, rows; SplitPlan sp;
  int rc = prepare(op, &A, &tile, &mx, &rows, &sp);
  if (rc) return rc;
  return sp.ksplit > 1 ? (int64_t)sp.ksplit * sp.positions * A.G.Ncols : 0;
}

template <int WM, int WN, int TM, int TN>
static int launch(IgemmArgs& A, bool vec, int max_mblk, bool double_buf, const SplitPlan& sp, hipStream_t st) {
  constexpr int BN = WN * TN * 32;
  dim3 grid(max_mblk, gode_ceil_div(A.G.Ncols, BN), A.G.nphase), block(WM * WN * 64);
  static const bool generic_forced = getenv("GODE_IGEMM_GENERIC") != nullptr;     // read once per process
  bool fast = vec && fast_geometry(A.G) && (sp.ksplit > 1 || !generic_forced);
  if (sp.ksplit > 1 && (!fast || A.work == nullptr)) return GODE_E_ARG;   // the plan needs the FAST path + workspace
  A.ksplit = sp.ksplit; A.slabs_per_split = sp.slabs_per_split;
  if (fast) {
    const int MB = max_mblk, NB = (int)grid.y, NC = NB * A.G.nphase;
    A.MB = MB; A.NB = NB; A.xcd_mode = 0;
    static const char* senv = getenv("GODE_IGEMM_STAGGER");
    A.stagger = senv ? atoi(senv) : 0;
#ifndef GODE_ABLATION_BUILD
    if (A.stagger < 0) A.stagger = 0;   // the negative (ablation) selectors only exist in -DGODE_ABLATION_BUILD builds
#endif
    static const char* xenv = getenv("GODE_IGEMM_XCD");
    const int force = xenv ? atoi(xenv) : -1;
    const int64_t wbytes = gode_pack_floats(A.G) * 4;
    const bool ok1 = (NC % 8 == 0) || (NC < 8 && 8 % NC == 0 && MB % (8 / NC) == 0);
    const bool ok2 = MB % 8 == 0;
    // measured L2-miss (fabric) fetch per launch on the three decoder layers, forward / backward-data, raw FETCH_SIZE MB:
    // mode 0: 45/95/411, 76/139/161;  mode 1: 203/230/338, 197/222/161;  mode 2: 68/73/46, 76/93/161;
    // mode 3: 68/33/36, 60/89/154 -- at equal launch times (+-1.5 %), so the choice is made on traffic alone
    // Mode 3 makes every XCD read the whole weight pack once; when 8 copies of it outweigh the gathered tensor 3:1
    // (ConvT 512->256 forward: 8 x 8.4 MB vs 16.8 MB) the plain order, which spreads the panels, fetches less.
    const int64_t images = (int64_t)A.out_numel / ((int64_t)A.G.Xd * A.G.Xh * A.G.Xw * A.G.Ncols);
    const int64_t in_bytes = images * A.G.Gd * A.G.Gh * A.G.Gw * A.G.Cg * 4;
    int mode = ok2 ? (wbytes * 8 < 3 * in_bytes ? 3 : 0) : 0;
    // depth-major rows with dead-tap skipping: the tiles of the end planes are light, so a contiguous run per XCD would
    // leave the XCDs that own them idle -- interleave the m-blocks over the XCDs instead
    // -- and NOT the plain order either: it hands an XCD m-blocks 8 apart, i.e. (at 16 tiles per depth plane) only
    // planes of one weight: UCF video-D layer 2 at N = 32 runs 600 us in the plain order, 845 us with contiguous runs,
    // 470 us interleaved (scripts/exp/xcd_sweep.py; no other shape of either config moves by more than 2 %)
    if (A.dmajor > 0) mode = ok2 ? 2 : (ok1 ? 1 : 0);
    if (force == 0) mode = 0;
    if (force == 1 && ok1) mode = 1;
    if (force == 2 && ok2) mode = 2;
    if (force == 3 && ok2) mode = 3;
    A.xcd_mode = mode;
    dim3 g1(MB * NC * sp.ksplit);
    const bool has_xf = A.scale != nullptr || A.act != GODE_ACT_NONE;
    static const char* genv = getenv("GODE_IGEMM_GLDS");
    // default; GODE_IGEMM_GLDS=0 selects register staging (as do operands beyond the 2 GiB a raw buffer's offsets span)
    const bool glds = (genv ? atoi(genv) != 0 : true) && A.src_bytes != 0 && A.w_bytes != 0;
    if (!has_xf && glds && !double_buf) {
      if (A.tapskip) hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, 3, false>), g1, block, 0, st, A);
      else hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, 2, false>), g1, block, 0, st, A);
    } else if (double_buf) {
      if (has_xf) hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, 1, true>), g1, block, 0, st, A);
      else hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, 1, false>), g1, block, 0, st, A);
    } else {
      if (has_xf) hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, 0, true>), g1, block, 0, st, A);
      else hipLaunchKernelGGL((igemm_fast_kernel<WM, WN, TM, TN, 0, false>), g1, block, 0, st, A);
    }
    if (sp.ksplit > 1) {
      GODE_LAUNCH_CHECK();
      if (A.G.Ncols % 4 == 0 && ((uintptr_t)A.work % 16) == 0 && ((uintptr_t)A.out % 16) == 0)
        hipLaunchKernelGGL(igemm_splitk_reduce_vec_kernel, dim3(gode_ceil_div(sp.positions, SPLITK_ROWS), gode_ceil_div(A.G.Ncols, 64)),
                           dim3(256), 0, st, A.work, A.out, A.stats, sp.positions, A.G.Ncols, sp.ksplit, A.epilogue);
      else
        hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3(gode_ceil_div(sp.positions, SPLITK_ROWS)), dim3(256), 0, st, A.work, A.out,
                           A.stats, sp.positions, A.G.Ncols, sp.ksplit, A.epilogue);
    }
  }
  else if (vec) hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, true>), grid, block, 0, st, A);
  else hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, false>), grid, block, 0, st, A);
  GODE_LAUNCH_CHECK();
  return 0;
}

extern "C" int gode_igemm(const gode_igemm_op* op, void* stream) {
  if (!op || !op->src || !op->wpack || !op->out) return GODE_E_ARG;
  IgemmArgs A;
  int tile, max_mblk, rows;
  SplitPlan sp;
  int rc = prepare(op, &A, &tile, &max_mblk, &rows, &sp);
  if (rc) return rc;
  if (max_mblk == 0) return 0;
  const IgemmGeom& G = A.G;
  int64_t gs[5];
  if (gode_strides_are_channels_last(op->gs)) {
    gs[4] = 1; gs[3] = G.Cg; gs[2] = (int64_t)G.Gw * gs[3]; gs[1] = (int64_t)G.Gh * gs[2]; gs[0] = (int64_t)G.Gd * gs[1];
  } else {
    for (int i = 0; i < 5; ++i) gs[i] = op->gs[i];
  }
  // 32-bit offset arithmetic inside the kernel: the gathered span and the output must stay below 2^31 elements
  int64_t span = 1 + (int64_t)(op->g.N - 1) * gs[0] + (int64_t)(G.Gd - 1) * gs[1] + (int64_t)(G.Gh - 1) * gs[2] +
                 (int64_t)(G.Gw - 1) * gs[3] + (int64_t)(G.Cg - 1) * gs[4];
  int64_t outn = (int64_t)op->g.N * G.Xd * G.Xh * G.Xw * G.Ncols;
  for (int i = 0; i < 5; ++i) if (gs[i] < 0) return GODE_E_ARG;
  if (span >= (1ll << 31) || outn >= (1ll << 31)) return GODE_E_SHAPE;
  if (op->src && op->wpack && op->out && ((uintptr_t)op->wpack % 16) == 0) {      // first layers with 1-4 input channels: conv_patch.hip
    const int pr = gode_launch_conv_patch_fprop(op, gs, (hipStream_t)stream);
    if (pr != 0) return pr < 0 ? pr : 0;
  }
  A.src = op->src; A.w = op->wpack; A.out = op->out; A.scale = op->scale; A.shift = op->shift; A.stats = op->stats;
  A.gsN = (int)gs[0]; A.gsD = (int)gs[1]; A.gsH = (int)gs[2]; A.gsW = (int)gs[3]; A.gsC = (int)gs[4];
  A.act = op->act; A.epilogue = op->epilogue;
  A.stagger = 0;
  // input gradient of a convolution with temporal taps: depth-major tiles, so that the taps falling off the tensor's ends
  // are dead for whole tiles (skipped in the FAST kernel's K loop)
  A.dmajor = 0;
  A.tapskip = igemm_wants_tapskip(op, G);
  // Depth-major rows: every tile sees one or two planes (a small plane) or a slice of one (a plane of >= 128 rows, which
  // would fill tiles by itself in image-major order too -- but depth-major with the interleaved XCD order below is 4-6 %
  // faster there as well: UCF video-D layer 1 input gradient 833 -> 800 us at N = 32, 405 -> 381 at N = 16).
  if (A.tapskip) A.dmajor = op->g.N;
  A.stats_rows = rows;
  A.src_bytes = span * 4 < (1ll << 31) ? (uint32_t)(span * 4) : 0u;
  A.w_bytes = gode_pack_floats(G) * 4 < (1ll << 31) ? (uint32_t)(gode_pack_floats(G) * 4) : 0u;
  A.work = op->work; A.out_numel = (int32_t)outn; A.ksplit = 1; A.slabs_per_split = 0; A.MB = 0; A.NB = 0; A.xcd_mode = 0;
  if ((op->scale == nullptr) != (op->shift == nullptr)) return GODE_E_ARG;
  const bool vec = gs[4] == 1 && (G.Cg % 4) == 0 && (gs[0] % 4) == 0 && (gs[1] % 4) == 0 && (gs[2] % 4) == 0 &&
                   (gs[3] % 4) == 0 && ((uintptr_t)op->src % 16) == 0 &&
                   (op->scale == nullptr || ((uintptr_t)op->scale % 16 == 0 && (uintptr_t)op->shift % 16 == 0));
  if (((uintptr_t)op->wpack % 16) != 0) return GODE_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (op->groups != 2) {      // (the thin-output streaming kernels below know nothing of image groups; grouped ops carry BN stats)
    int maxM = 0, minK = 1 << 30; bool kp_ok = true;
    for (int i = 0; i < G.nphase; ++i) {
      if (G.ph[i].M > maxM) maxM = G.ph[i].M;
      if (G.ph[i].K < minK) minK = G.ph[i].K;
      kp_ok = kp_ok && G.ph[i].Kp == G.ph[i].K;
    }
    const int lpp = G.Cg / 4;
    int maxK = 0;
    for (int i = 0; i < G.nphase; ++i) if (G.ph[i].Kp > maxK) maxK = G.ph[i].Kp;
    const bool one_tap = G.nphase == 1 && G.ph[0].K == G.Cg;
    // the multi-tap form needs many positions to beat the padded MFMA path (measured: M = 25 k slower, M >= 65 k faster)
    if (vec && kp_ok && G.Ncols <= 4 && G.Ncols * maxK <= PW_MAXW && op->stats == nullptr && op->tile == 0 &&
        maxM >= (one_tap ? 4096 : 65536) && (lpp == 4 || lpp == 8 || lpp == 16 || lpp == 32 || lpp == 64)) {
      PwArgs D; D.a = A; D.lpp = lpp;
      for (int i = 0; i < G.nphase; ++i) {
        D.dMw[i] = make_fastdiv((uint32_t)G.ph[i].Mw); D.dMh[i] = make_fastdiv((uint32_t)G.ph[i].Mh);
        D.dMd[i] = make_fastdiv((uint32_t)G.ph[i].Md);
      }
      int64_t blocks = ((int64_t)maxM + (64 / lpp) * 4 * 8 - 1) / ((64 / lpp) * 4 * 8);     // ~8 position groups per wave
      if (blocks > (one_tap ? 4096 : 2048)) blocks = one_tap ? 4096 : 2048;
      if (one_tap) hipLaunchKernelGGL(conv_pointwise_kernel<true>, dim3((int)blocks, 1), dim3(256), 0, st, D);
      else hipLaunchKernelGGL(conv_pointwise_kernel<false>, dim3((int)blocks, G.nphase), dim3(256), 0, st, D);
      GODE_LAUNCH_CHECK();
      return 0;
    }
    {
      const int nl = G.Ncols / 4;
      // K <= 4 only: at K = 8 / 16 (first discriminator layers) the per-lane gather chain made it no faster than the
      // padded MFMA path (24.9 vs 22 us, 15.7 vs 10 us); at K = 1 it is 27 us against 63 us
      if (G.nphase == 1 && G.ph[0].K <= 4 && G.Ncols % 4 == 0 && G.Ncols <= 128 && op->stats == nullptr && op->tile == 0 &&
          maxM >= 2048 && (nl == 4 || nl == 8 || nl == 16 || nl == 32) && ((uintptr_t)op->out % 16) == 0) {
        PwArgs D; D.a = A; D.lpp = nl;
        D.dMw[0] = make_fastdiv((uint32_t)G.ph[0].Mw); D.dMh[0] = make_fastdiv((uint32_t)G.ph[0].Mh); D.dMd[0] = make_fastdiv((uint32_t)G.ph[0].Md);
        int64_t blocks = ((int64_t)maxM + (64 / nl) * 4 * 4 - 1) / ((64 / nl) * 4 * 4);     // ~4 position groups per wave
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv_smallk_kernel, dim3((int)blocks), dim3(256), 0, st, D);
        GODE_LAUNCH_CHECK();
        return 0;
      }
    }
    if (vec && kp_ok && G.Ncols <= 4 && op->stats == nullptr && maxM <= 16384 && minK >= 512 && op->tile == 0) {
      DotArgs D; D.a = A; D.dCg = make_fastdiv((uint32_t)G.Cg);
      D.per_block = (maxM <= 256 && minK >= 4096) ? 1 : 4;
      hipLaunchKernelGGL(conv_dot_kernel, dim3(D.per_block == 1 ? maxM : gode_ceil_div(maxM, 4), G.nphase), dim3(256), 0, st, D);
      GODE_LAUNCH_CHECK();
      return 0;
    }
  }
  const bool sb = op->tile >= 10;   // request the double-buffered FAST variant
  switch (tile) {
    case TILE_128x128: return launch<2, 2, 2, 2>(A, vec, max_mblk, sb, sp, st);
    case TILE_128x64: return launch<2, 2, 2, 1>(A, vec, max_mblk, sb, sp, st);
    case TILE_128x32: return launch<4, 1, 1, 1>(A, vec, max_mblk, sb, sp, st);
    case TILE_64x64: return launch<2, 2, 1, 1>(A, vec, max_mblk, sb, sp, st);
    case TILE_128x128_W8: return launch<4, 2, 1, 2>(A, vec, max_mblk, sb, sp, st);
    case TILE_256x128: return launch<4, 2, 2, 2>(A, vec, max_mblk, sb, sp, st);
  }
  return GODE_E_ARG;
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing: canonical W[co][ci][taps] -> per-phase panels Bp[n][Kp] (k contiguous, zero padded)
struct PackArgs {
  IgemmGeom G;
  gode_conv_geom g;
  int dir;
  const float* w;
  float* wp;
  const int32_t* co_perm;
};

__global__ void pack_kernel(const PackArgs a) {
  const PhaseGeom& P = a.G.ph[blockIdx.z];
  const int64_t total = (int64_t)a.G.Ncols * P.Kp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / P.Kp), k = (int)(i - (int64_t)n * P.Kp);
    const int64_t src = gode_pack_source(a.g, a.dir, a.G, P, n, k, a.co_perm);
    a.wp[P.w_off + i] = src >= 0 ? a.w[src] : 0.f;
  }
}

extern "C" int64_t gode_pack_size(const gode_conv_geom* g, int dir) {
  IgemmGeom G;
  int rc = gode_build_igemm_geom(*g, dir, &G);
  return rc ? rc : gode_pack_floats(G);
}

// Several panels in one launch (gode_run hands over every run of consecutive pack ops: after an optimiser step all
// panels of a network are stale at once, and ten 6-us launches cost more than the 20 MB they move).  The entries
// travel by value in the kernel arguments, so only what the index map needs is kept per phase.
#define PACK_BATCH 8
struct PackPhase { int32_t Th, Tw, K, Kp, kd0, kh0, kw0, pad_; int64_t w_off, total; };
struct PackEntry {
  const float* w; float* wp; const int32_t* co_perm;
  int32_t nphase, Cg, fullk, dir, ks_d, ks_h, ks_w, Ci, kd, kh, kw, Co;
  PackPhase ph[GODE_MAX_PHASES];
};
struct PackBatch { PackEntry e[PACK_BATCH]; };
static_assert(sizeof(PackBatch) <= 4096, "kernel-argument budget");

// same map as gode_pack_source (conv_geom.h), on the trimmed entry
__global__ void __launch_bounds__(256) pack_batch_kernel(const PackBatch b) {
  const PackEntry& E = b.e[blockIdx.z];
  if ((int)blockIdx.y >= E.nphase) return;
  const PackPhase& P = E.ph[blockIdx.y];
  const int taps = E.kd * E.kh * E.kw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / P.Kp), k = (int)(i - (int64_t)n * P.Kp);
    float v = 0.f;
    if (k < P.K) {
      int co, ci, kd, kh, kw;
      if (E.fullk) {
        const int tapn = n / E.Ci;
        ci = n - tapn * E.Ci; co = k;
        kd = tapn / (E.kh * E.kw); kh = (tapn / E.kw) % E.kh; kw = tapn % E.kw;
      } else {
        const int tap = k / E.Cg, c = k - tap * E.Cg;
        const int jw = tap % P.Tw, jh = (tap / P.Tw) % P.Th, jd = tap / (P.Tw * P.Th);
        kd = P.kd0 + jd * E.ks_d; kh = P.kh0 + jh * E.ks_h; kw = P.kw0 + jw * E.ks_w;
        if (E.dir == GODE_FPROP) { co = n; ci = c; } else { co = c; ci = n; }
      }
      if (E.co_perm) co = E.co_perm[co];
      if (co >= 0) v = E.w[((int64_t)co * E.Ci + ci) * taps + (kd * E.kh + kh) * E.kw + kw];
    }
    E.wp[P.w_off + i] = v;
  }
}


// Tile form of the same map for the panels that matter (both channel counts multiples of 16, a power-of-two number of taps,
// no padding columns, no channel permutation): the element-wise kernel above reads the canonical tensor with a stride of
// `taps` (FPROP) or `Ci*taps` (DGRAD) floats between neighbouring lanes -- 4 useful bytes per 64-byte fetch; UCF's 4x4x4
// video-discriminator weights took 48 us per launch of 8 panels, five times their HBM time.  Here a workgroup moves one
// [16 co][16 ci][TT taps] brick through LDS: canonical reads in runs of TT (<= 32) floats, panel writes in runs of 16.
#define PACK_T 16
__global__ void __launch_bounds__(256) pack_tile_kernel(const PackBatch b) {
  const PackEntry& E = b.e[blockIdx.z];
  const int taps = E.kd * E.kh * E.kw;
  const int TT = taps < 32 ? taps : 32, ntb = taps / TT;
  const int nci = E.Ci / PACK_T, nco = E.Co / PACK_T;
  if ((int)blockIdx.x >= nci * nco * ntb) return;
  __shared__ float brick[PACK_T * (PACK_T * 33 + 1)];
  constexpr int CS = PACK_T * 33 + 1;          // co stride (odd: lanes that differ in co hit different banks)
  const int tb = blockIdx.x % ntb, ib = (blockIdx.x / ntb) % nci, cb = blockIdx.x / (ntb * nci);
  const int co0 = cb * PACK_T, ci0 = ib * PACK_T, t0 = tb * TT;
  const int tid = threadIdx.x;
  const int lt = __builtin_ctz(TT);
  for (int r = 0; r < TT; ++r) {
    const int idx = tid + 256 * r;
    const int tl = idx & (TT - 1), ci_l = (idx >> lt) & (PACK_T - 1), co_l = idx >> (lt + 4);
    brick[co_l * CS + ci_l * 33 + tl] = E.w[((int64_t)(co0 + co_l) * E.Ci + ci0 + ci_l) * taps + t0 + tl];
  }
  __syncthreads();
  if (E.dir == GODE_FPROP) {                   // n = co, k = tap * Ci + ci (one phase, taps in canonical order)
    const PackPhase& P = E.ph[0];
    const int ci_l = tid & (PACK_T - 1), co_l = tid >> 4;
    float* dst = E.wp + P.w_off + (int64_t)(co0 + co_l) * P.Kp + ci0 + ci_l;
    for (int tl = 0; tl < TT; ++tl) dst[(int64_t)(t0 + tl) * E.Ci] = brick[co_l * CS + ci_l * 33 + tl];
  } else {                                     // n = ci, k = (tap of the stride phase) * Co + co
    const int co_l = tid & (PACK_T - 1), ci_l = tid >> 4;
    for (int p = 0; p < E.nphase; ++p) {
      const PackPhase& P = E.ph[p];
      float* dst = E.wp + P.w_off + (int64_t)(ci0 + ci_l) * P.Kp + co0 + co_l;
      int j = 0;                                // (nested counters: no divisions in the tap walk)
      for (int kd = P.kd0; kd < E.kd; kd += E.ks_d)
        for (int kh = P.kh0; kh < E.kh; kh += E.ks_h)
          for (int kw = P.kw0; kw < E.kw; kw += E.ks_w, ++j) {
            const int t = (kd * E.kh + kh) * E.kw + kw;
            if (t >= t0 && t < t0 + TT) dst[(int64_t)j * E.Cg] = brick[co_l * CS + ci_l * 33 + (t - t0)];
          }
    }
  }
}

static bool pack_tileable(const PackEntry& E) {
  const int taps = E.kd * E.kh * E.kw;
  if (E.co_perm || E.fullk || taps < 4 || taps > 64 || (taps & (taps - 1)) != 0 || E.Ci % PACK_T != 0 || E.Co % PACK_T != 0) return false;
  for (int i = 0; i < E.nphase; ++i) if (E.ph[i].K != E.ph[i].Kp) return false;
  return true;
}

extern "C" int gode_pack_batch_(const gode_pack_op* const* ops, int n, void* stream) {
  for (int at = 0; at < n; at += PACK_BATCH) {
    PackBatch B;
    const int cnt = n - at < PACK_BATCH ? n - at : PACK_BATCH;
    int64_t mx = 0;
    for (int e = 0; e < cnt; ++e) {
      const gode_pack_op* op = ops[at + e];
      if (!op || !op->w || !op->wpack) return GODE_E_ARG;
      IgemmGeom G;
      const int rc = gode_build_igemm_geom(op->g, op->dir, &G);
      if (rc) return rc;
      PackEntry& E = B.e[e];
      E.w = op->w; E.wp = op->wpack; E.co_perm = op->co_perm;
      E.nphase = G.nphase; E.Cg = G.Cg; E.fullk = G.fullk; E.dir = op->dir;
      E.ks_d = G.kstep_d; E.ks_h = G.kstep_h; E.ks_w = G.kstep_w;
      E.Ci = op->g.Ci; E.kd = op->g.kd; E.kh = op->g.kh; E.kw = op->g.kw; E.Co = op->g.Co;
      for (int i = 0; i < G.nphase; ++i) {
        const PhaseGeom& p = G.ph[i];
        PackPhase& q = E.ph[i];
        q.Th = p.Th; q.Tw = p.Tw; q.K = p.K; q.Kp = p.Kp; q.kd0 = p.kd0; q.kh0 = p.kh0; q.kw0 = p.kw0; q.pad_ = 0;
        q.w_off = p.w_off; q.total = (int64_t)G.Ncols * p.Kp;
        if (q.total > mx) mx = q.total;
      }
    }
    if (mx == 0) continue;
    // the tileable entries of the batch go through the brick kernel, the rest through the element-wise one
    PackBatch T, R;
    int nt = 0, nr = 0, tiles = 0; int64_t mr = 0;
    for (int e = 0; e < cnt; ++e) {
      const PackEntry& E = B.e[e];
      if (pack_tileable(E)) {
        const int taps = E.kd * E.kh * E.kw, TT = taps < 32 ? taps : 32;
        const int t = (E.Ci / PACK_T) * (E.Co / PACK_T) * (taps / TT);
        if (t > tiles) tiles = t;
        T.e[nt++] = E;
      } else {
        for (int i = 0; i < E.nphase; ++i) if (E.ph[i].total > mr) mr = E.ph[i].total;
        R.e[nr++] = E;
      }
    }
    if (nt > 0) {
      hipLaunchKernelGGL(pack_tile_kernel, dim3(tiles, 1, nt), dim3(256), 0, (hipStream_t)stream, T);
      GODE_LAUNCH_CHECK();
    }
    if (nr > 0 && mr > 0) {
      int blocks = (int)((mr + 1023) / 1024); if (blocks > 512) blocks = 512; if (blocks < 1) blocks = 1;
      hipLaunchKernelGGL(pack_batch_kernel, dim3(blocks, GODE_MAX_PHASES, nr), dim3(256), 0, (hipStream_t)stream, R);
      GODE_LAUNCH_CHECK();
    }
  }
  return 0;
}

extern "C" int gode_pack_weights(const gode_conv_geom* g, int dir, const float* w, float* wpack, const int32_t* co_perm,
                                 int32_t co_canon, void* stream) {
  (void)co_canon;
  if (!g || !w || !wpack) return GODE_E_ARG;
  PackArgs A;
  int rc = gode_build_igemm_geom(*g, dir, &A.G);
  if (rc) return rc;
  A.g = *g; A.dir = dir; A.w = w; A.wp = wpack; A.co_perm = co_perm;
  int64_t mx = 0;
  for (int i = 0; i < A.G.nphase; ++i) { int64_t t = (int64_t)A.G.Ncols * A.G.ph[i].Kp; if (t > mx) mx = t; }
  if (mx == 0) return 0;
  int blocks = (int)((mx + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks, 1, A.G.nphase), dim3(256), 0, (hipStream_t)stream, A);
  GODE_LAUNCH_CHECK();
  return 0;
}
