/* gode.h -- C ABI of libgode.so, the MI355X (gfx950) native library behind the MoCoGAN + Neural-ODE hot path.
 *
 * The reference (chechaohp/gan-ode) is pure Python on PyTorch and has no FFI of its own; its seam for this path is
 * the class surface imported by mnist_moco_ode.py:5-6 / ucf_moco_ode.py:5-6.  The Python host code in
 * gan-ode_amd/ mirrors that surface and lowers every call to the entry points below through ctypes.  Each entry
 * point cites the reference arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (PyTorch's allocator);
 *    the library never allocates, frees or keeps memory and has no mutable global state;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and never synchronises;
 *  - return value: 0 ok, <0 argument error (GODE_E_*), >0 a hipError_t from a launch;
 *  - activations are channels-last fp32: [N][D][H][W][C]; weights stay in PyTorch's canonical layout
 *    W[co][ci][kd][kh][kw] (Conv) == W[in][out][kh][kw] (ConvTranspose, seen as the conv whose dgrad it is) and
 *    are re-packed on the device by gode_pack_* into the K-contiguous panels the GEMM kernels read.
 */
#ifndef GODE_H
#define GODE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GODE_VERSION 105

enum { GODE_OK = 0, GODE_E_ARG = -1, GODE_E_SHAPE = -2, GODE_E_KIND = -3 };
enum { GODE_ACT_NONE = 0, GODE_ACT_RELU = 1, GODE_ACT_LRELU = 2,              /* LeakyReLU slope is 0.2 */
       GODE_ACT_TANH_OUT = 3 /* bn_bwd only: y holds tanh OUTPUT, g *= 1-y^2 */ };
enum { GODE_EPI_RAW = 0, GODE_EPI_TANH = 1 };
enum { GODE_FPROP = 0, GODE_DGRAD = 1 };

/* A regular convolution x[N,Di,Hi,Wi,Ci] -> y[N,Do,Ho,Wo,Co].  2-D layers use D=1,kd=1,sd=1,pd=0.  A
 * ConvTranspose layer is described by the convolution whose data-gradient it is (x = its output). */
typedef struct gode_conv_geom {
  int32_t N, Ci, Co;
  int32_t Di, Hi, Wi;
  int32_t Do, Ho, Wo;
  int32_t kd, kh, kw;
  int32_t sd, sh, sw;
  int32_t pd, ph, pw; /* may be negative (ConvTranspose k=1,p=2 of mocogan_ode.py:82 is a crop) */
} gode_conv_geom;

/* ---- implicit-GEMM convolution (fp32 MFMA 32x32x2) ---------------------------------------------------------
 * dir=GODE_FPROP: out=y, gathers x.  dir=GODE_DGRAD: out=x, gathers y (one sub-GEMM per stride phase; this is
 * the ConvTranspose2d forward of models/mocogan.py:200-215 and the Conv3d/Conv2d input-gradient).
 * The gathered operand is read through element strides gs[5] = {N,D,H,W,C} (lets the first discriminator layer
 * read the caller's [B,C,T,H,W] / transposed views in place, mnist_moco_ode.py:136-139) and transformed on load:
 *   a = act(v*scale[c] + shift[c])   (train-mode BatchNorm + ReLU/LeakyReLU of the PREVIOUS layer; nullable).
 * Zero padding is applied after the transform.  `wpack` comes from gode_pack_weights for the same geom/dir.
 * If stats != NULL, per-column partial sums (sum, sum of squares) of the raw outputs are written to
 * stats[2][ncols][rows] (row = m-block, rows = gode_igemm_stats_rows(op); deterministic), feeding
 * gode_bn_finalize, which reads each channel's rows contiguously. */
typedef struct gode_igemm_op {
  gode_conv_geom g;
  int32_t dir, act, epilogue, tile; /* tile: 0 auto */
  const float* src;
  const float* wpack;
  float* out;
  const float* scale;
  const float* shift;
  float* stats;
  int64_t gs[5];
  float* work; /* nullable; >= gode_igemm_work_size floats enables split-K for launches that would not fill the GPU */
  /* groups == 2 (FPROP with stats only, N even): the batch is two image groups [0, N/2) and [N/2, N) whose BatchNorm
   * partial statistics must stay separate -- one discriminator pass over [real; fake] keeps the per-pass batch
   * statistics of the reference's two calls (mnist_moco_ode.py:119-124,137-143).  The partial-statistics rows of group 0
   * are the first gode_igemm_stats_rows0(op) rows of every column, group 1 the rest.  0 / 1: one group. */
  int32_t groups, pad3_;
} gode_igemm_op;
int gode_igemm(const gode_igemm_op* op, void* stream);
/* floats of workspace the op can use for split-K (0: it never splits).  Deterministic from the op's geometry. */
int64_t gode_igemm_work_size(const gode_igemm_op* op);
/* number of partial-stats rows gode_igemm writes for this op (host-side, no GPU work) */
int gode_igemm_stats_rows(const gode_igemm_op* op);
/* rows of those that belong to image group 0 (== gode_igemm_stats_rows unless op->groups == 2) */
int gode_igemm_stats_rows0(const gode_igemm_op* op);
/* A stack whose batch is [first `split_images` images; the rest] with SEPARATE BatchNorm statistics per part (the generator's
 * video and image paths decoded in one pass): which partial-statistics rows belong to which part.  Writes up to 8 segments
 * {begin, split, end} (gode_bn_finalize_op.seg) and returns their number; GODE_E_SHAPE when a tile (or split-K statistics
 * chunk) of this op would straddle the boundary -- the caller then runs the two parts separately. */
int gode_igemm_stats_segments(const gode_igemm_op* op, int32_t split_images, int32_t* seg);
/* The launch-time estimate (GPU cycles) of the tile / split-K cost model for this op, or -1 when the op does not take the
 * modelled path.  Host-side, no GPU work: such a caller compares one launch over the whole batch with two over its parts. */
double gode_igemm_model_cycles(const gode_igemm_op* op);
/* floats needed for the packed weights of (geom, dir) */
int64_t gode_pack_size(const gode_conv_geom* g, int dir);
/* canonical W[co][ci][taps] -> packed panels.  co_perm (nullable, length g->Co): internal y-side channel c is
 * canonical output channel co_perm[c] (zero weights when co_perm[c] < 0); g->Co counts INTERNAL channels.  Used for
 * the generator's first layer, whose latent buffer is stored [motion 16|content 50|pad 6] while the reference
 * concatenates content first (models/mocogan.py:267). */
int gode_pack_weights(const gode_conv_geom* g, int dir, const float* w, float* wpack, const int32_t* co_perm,
                      int32_t co_canon, void* stream);

/* ---- weight gradient: dW[co][ci][taps] = sum_m y_grad[m][co] * act(x*scale+shift)[pos(m,tap)][ci] ----------
 * split over the reduction (output positions) into `splits` slabs of a workspace, then reduced deterministically
 * and scattered into the canonical layout (accumulate!=0 adds to dW).  xform_on_y!=0 applies the transform to
 * the y-side operand instead (ConvTranspose layers: their *input* is the y side). */
typedef struct gode_wgrad_op {
  gode_conv_geom g;
  int32_t act, xform_on_y, splits, accumulate;
  const float* x;  /* x side tensor (channels-last unless xs given) */
  const float* y;  /* y side tensor, channels-last */
  const float* scale;
  const float* shift;
  float* work;     /* >= gode_wgrad_work_size floats */
  float* dw;       /* canonical layout */
  const int32_t* co_perm; int64_t co_canon; /* as in gode_pack_weights, nullable / g.Co */
  int64_t xs[5];   /* element strides of x {N,D,H,W,C}; all zero = channels-last contiguous */
} gode_wgrad_op;
int gode_wgrad(const gode_wgrad_op* op, void* stream);
int64_t gode_wgrad_work_size(const gode_wgrad_op* op);
int gode_wgrad_auto_splits(const gode_conv_geom* g);

/* ---- BatchNorm (train mode: batch statistics, eps, momentum as nn.BatchNorm2d/3d defaults) ------------------
 * finalize: partial sums -> mean, invstd, and the fused affine scale=gamma*invstd, shift=beta-mean*scale that the
 * NEXT layer's operand load applies; updates running_mean/var (unbiased var) and num_batches_tracked.
 * training==0: scale/shift from the running statistics (genSamples' eval mode, mnist_moco_ode.py:31-35). */
typedef struct gode_bn_finalize_op {
  const float* stats; int32_t rows, ncols, C; int64_t count;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* mean; float* invstd; float* scale; float* shift;
  float momentum, eps; int32_t training, pad_;
  /* groups == 2: the first rows0 partial rows belong to image group 0, the rest to group 1 (count = elements per channel
   * and GROUP); mean / invstd / scale / shift are [2][C] (group-major); the running statistics receive the two
   * momentum updates in group order (what two successive forward calls do), num_batches_tracked += 2. */
  int32_t groups, rows0;
  /* groups == 2, general form (nseg > 0 overrides rows0): up to 8 row segments {begin, split, end} -- rows [begin, split) are
   * group 0, [split, end) group 1 (gode_igemm_stats_segments: one per stride phase of a transposed-convolution stack whose
   * rows are [first batch; second batch]); count1 (> 0): elements per channel of group 1 when the groups differ in size
   * (count: group 0); order != 0: group 1's momentum update is applied first (the reference called that batch first). */
  int32_t nseg, order; int64_t count1; int32_t seg[24];   /* 8 x {begin, split, end} */
  /* groups == 2, two-launch form (overrides nseg / rows0): group 1's partial sums are an array of their own,
   * stats1[2][ncols][rows1], written by a second gode_igemm launch over that batch alone; `stats` / `rows` are group 0's. */
  const float* stats1; int32_t rows1, pad3_;
} gode_bn_finalize_op;
int gode_bn_finalize(const gode_bn_finalize_op* op, void* stream);

/* out = act(y*scale[c] + shift[c]) over [M][C] (C % 4 == 0): materialises an activated tensor for consumers that are
 * MFMA-bound and re-read every element several times (the heavy GEMMs run ~10-20 % faster without the transform
 * in their operand path; light layers keep it fused).  scale==NULL: plain activation. */
typedef struct gode_bn_apply_op {
  const float* y; float* out; const float* scale; const float* shift; int64_t M; int32_t C, act;
  int64_t M0; /* > 0: rows >= M0 use scale + C / shift + C (second image group of a grouped pass); 0: one group */
} gode_bn_apply_op;
int gode_bn_apply(const gode_bn_apply_op* op, void* stream);

/* Overlap-add of per-pixel tap contributions: the second half of a thin-output ConvTranspose2d (the UCF generator's
 * RGB head, models/mocogan.py:213) computed as ONE plain GEMM over the input pixels -- gode_igemm, DGRAD of the geometry
 * {N = pixels, Di x Hi x Wi = 1 x kh x kw, Do = Ho = Wo = 1}: cols[pixel][(kh, kw, c)] -- followed by
 *   out[n][oh][ow][c] = epilogue( sum over taps with oh = ih*sh - ph + kh, ow = iw*sw - pw + kw of cols[(n, ih, iw)][(kh, kw, c)] )
 * in fixed tap order.  Every input activation is then read once instead of once per tap and phase. */
typedef struct gode_col2im_op {
  const float* cols; float* out;
  int32_t N, Hi, Wi, Ho, Wo, C, kh, kw, sh, sw, ph, pw;   /* Hi x Wi: the pixels of cols; Ho x Wo: out */
  int32_t epilogue, pad_;                                  /* GODE_EPI_RAW / GODE_EPI_TANH */
  /* 3-D form (kd > 0; all zero: the 2-D op above): cols pixels are (n, id, ih, iw) over Di x Hi x Wi with kd*kh*kw*C values
   * each (tap order kd, kh, kw), out is Do x Ho x Wo, od = id*sd - pd + kd.  The input gradient of a thin-INPUT Conv3d (the UCF
   * video discriminator's first layer, models/mocogan.py:100: 3 channels, 4x4x4 taps) is one GEMM over its output positions
   * + this overlap-add, instead of a gather of 1,024 values per input voxel. */
  int32_t Di, Do, kd, sd, pd, pad2_;
} gode_col2im_op;
int gode_col2im(const gode_col2im_op* op, void* stream);

/* backward of y -> BN -> act given g_a = dL/d(act output), all [M][C] channels-last:
 *   g_z = g_a * act'(scale*y+shift);  dbeta = sum g_z;  dgamma = sum g_z*xhat;
 *   g_y = gamma*invstd*(g_z - dbeta/M - xhat*dgamma/M)     written in place over g_a.
 * With mean==NULL the layer has no BN: g_y = g_a*act'(y).  work: >= gode_bn_bwd_work_size floats, 8-byte aligned
 * (the two sums are accumulated in fp64).
 * accumulate!=0 adds into dgamma/dbeta.
 * eval_mode!=0: the forward normalised with the RUNNING statistics (mean/invstd hold them), which are constants:
 *   g_y = gamma*invstd*g_z  (dgamma/dbeta as above, with xhat from the running statistics). */
typedef struct gode_bn_bwd_op {
  float* g; const float* y; int64_t M; int32_t C, act;
  const float* gamma; const float* mean; const float* invstd; const float* scale; const float* shift;
  float* dgamma; float* dbeta; float* work; int32_t accumulate, eval_mode;
  const float* gin; /* nullable: read g_a from here instead of g (g is then write-only); lets the caller's upstream
                       gradient tensor be consumed in place without being modified */
  /* groups == 2 (M even): rows [0, M/2) and [M/2, M) are two BatchNorm batches of one grouped pass -- mean / invstd /
   * scale / shift are [2][C] (group-major, as gode_bn_finalize writes them), the batch terms use each group's own sums,
   * dgamma / dbeta receive both groups' sums in group order (what the reference's two backward passes add up to). */
  int32_t groups, pad2_;
  int64_t M0;   /* groups == 2: rows of group 0 (0: M / 2) */
  /* Rank-1 upstream gradient (r1_s != NULL; g is then write-only and gin ignored): the layer feeds a 1x1 convolution to ONE
   * channel whose output is a centre crop (the MNIST generator's head, models/mocogan.py:151), so g_a[row][c] =
   * r1_w[c] * r1_s[crop(row)] and the [M][C] gradient tensor need not be written and read back before this op.  Rows are
   * (image, h, w) over r1_H x r1_W; r1_s is (image, h - r1_off, w - r1_off) over r1_h x r1_w, zero outside. */
  const float* r1_s; const float* r1_w; int32_t r1_H, r1_W, r1_h, r1_wd, r1_off, pad3_;
} gode_bn_bwd_op;
int gode_bn_bwd(const gode_bn_bwd_op* op, void* stream);
int64_t gode_bn_bwd_work_size(int64_t M, int32_t C);

/* ---- motion-latent ODE (models/mocogan_ode.py:6-17,123-148; torchdiffeq fixed-grid rk4 = Kutta 3/8) --------
 * forward: x[N][16] host-drawn noise -> pre-net Linear(16,64)/LReLU/Linear(64,16)/LReLU -> T-1 RK4(3/8) steps
 * of f(y)=W2 tanh(W1 y+b1)+b2 with step sizes dt[T-1] -> latent rows.  z is the generator's latent buffer
 * [rows][zcols] = [motion 16 | content 50 | zero pad] (zcols % 4 == 0, >= 68; the Python side uses 96 so that the
 * first decoder GEMM has K % 32 == 0); row n*T+t (sel_t==NULL) or row n holding time sel_t[n]
 * (sample_images' row selection, models/mocogan.py:287-295).  content[N][50] is broadcast over the T rows.
 * traj[N][T][16] (nullable) keeps the whole trajectory for the adjoint pass. */
typedef struct gode_ode_params {
  const float* Wa; const float* ba; const float* Wb; const float* bb; /* pre-net: [64,16],[64],[16,64],[16] */
  const float* W1; const float* b1; const float* W2; const float* b2; /* ODEFunc: [16,16],[16],[16,16],[16] */
} gode_ode_params;
typedef struct gode_ode_fwd_op {
  gode_ode_params p;
  const float* x; const float* content; const float* dt; const int32_t* sel_t;
  float* z; float* traj; int32_t N, T, substeps, prenet; /* prenet==0: linear=False (nn.Identity) */
  int32_t zcols, G;
  /* optional solver grid (torchdiffeq options={'step_size': h}: FixedGridODESolver integrates on its own grid and
   * interpolates the outputs linearly).  grid_dt[G] = step sizes of the grid; output j >= 1 is produced after grid
   * step emit_at[j] as y0 + emit_w[j]*(y1 - y0) (emit_w == 1: y1 itself).  grid_dt == NULL: the grid is the output
   * times (dt[T-1], `substeps` equal sub-steps each) -- the reference's own call. */
  const float* grid_dt; const int32_t* emit_at; const float* emit_w;
  /* method 1: torchdiffeq's dopri5 (adaptive Dormand-Prince 5(4), controller and 4th-order dense output as in
   * gode_odernn_fwd) over the T output times tout[T] (increasing, tout[0] = start); the error norm is taken jointly
   * over the trajectories of a workgroup (<= 64; torchdiffeq: over the whole batch).  nsteps[ceil(N/64)] (nullable)
   * receives the number of trial steps.  method 0: fixed-grid rk4 (above). */
  int32_t method, pad2_; float rtol, atol; const float* tout; int32_t* nsteps;
  /* method 1 with N > 32: >= gode_odernn_sync_size(N) words through which the workgroups exchange the whole-batch error norm
   * (see the ODE-RNN ops below; NULL or more than GODE_ODERNN_SYNC_MAX_N trajectories per launch: one norm per
   * 64-trajectory workgroup, nsteps per such workgroup).  With it nsteps[0] = trial steps of the call (negative: stalled). */
  int32_t* sync;
} gode_ode_fwd_op;
int gode_ode_fwd(const gode_ode_fwd_op* op, void* stream);
/* `count` (<= 8) independent method-1 solves in ONE launch (ops: HOST array): the dopri5 counterpart of
 * gode_odernn_fwd_multi below */
int gode_ode_fwd_multi(const gode_ode_fwd_op* ops, int32_t count, void* stream);
/* adjoint backward (torchdiffeq odeint_adjoint semantics: per output interval ONE reverse-time RK4(3/8) step of
 * (y, a, g_theta), y reset to the stored trajectory, a += upstream grad) followed by the pre-net backward.
 * gz[rows][zcols]: gradient wrt the latent rows (only the 16 motion columns are read).  Parameter gradients are
 * reduced deterministically through work (>= gode_ode_bwd_work_size floats) and written (accumulate!=0: added) to
 * grads, laid out as the 8 tensors of gode_ode_params in order (2672 floats). */
typedef struct gode_ode_bwd_op {
  gode_ode_params p;
  const float* x; const float* traj; const float* dt; const int32_t* sel_t; const float* gz;
  float* work; float* grads; int32_t N, T, substeps, prenet, accumulate, zcols;
  /* optional per-interval reverse step lists (same option: the adjoint solve of output interval j -> j-1 runs on the
   * grid torchdiffeq builds for the reversed span): steps bstep_dt[bstep_off[j-1] .. bstep_off[j]) for j = 1..T-1.
   * NULL: `substeps` equal steps of dt[j-1]/substeps. */
  const int32_t* bstep_off; const float* bstep_dt;
  /* method 1 with substeps == 0: torchdiffeq's ADAPTIVE adjoint of a dopri5 solve -- the augmented state (y, a, g_theta)
   * of the one adjoint call is integrated backwards over the output times tout[T] with dopri5 (rtol, atol), mixed norm
   * (max over the RMS norms of y, a and each ODEFunc parameter tensor, joint over a workgroup's <= 64 trajectories);
   * the theta state is carried across the output intervals, the solver restarts on each.  method 1 with substeps > 0:
   * the same adjoint discretised with `substeps` fixed Kutta-3/8 steps per interval (round-1 behaviour; dt required). */
  int32_t method, pad_; float rtol, atol; const float* tout;
  int32_t* nsteps;   /* nullable, adaptive adjoint only: trial steps of the adjoint call (nsteps[0]; without `sync` and N > 32:
                        [ceil(N/64)], one per workgroup); a negative entry reports a call that stalled (trial-step limit /
                        step-size underflow: torchdiffeq asserts there) */
  int32_t* sync;     /* adaptive adjoint with N > 32: as in gode_ode_fwd_op */
} gode_ode_bwd_op;
int gode_ode_bwd(const gode_ode_bwd_op* op, void* stream);
/* the adaptive adjoints (method 1, substeps 0) of `count` (<= 8) solves in one launch; ops sharing a grads pointer are
 * reduced in array order */
int gode_ode_bwd_multi(const gode_ode_bwd_op* ops, int32_t count, void* stream);
int64_t gode_ode_bwd_work_size(int32_t N);
#define GODE_ODE_NPARAM 2672

/* ---- ODE-RNN motion latent (models/mocogan_ode_rnn.py:41-53) -------------------------------
 * per frame: h' = odeint_adjoint(ODEFunc, h, [0,1])[-1] with torchdiffeq's default dopri5 (rtol, atol), then
 * h = GRUCell(e_t, h').  noise[T+1][N][16] = h_0, e_0..e_{T-1} drawn on the host.  z / content / sel_t as in
 * gode_ode_fwd (row t = h_{t+1}).  hs[N][T+1][16] (nullable), hp[N][T][16] (post-ODE states, kept for the backward).
 * Error norm: the RMS over ALL N trajectories of the call, as torchdiffeq takes it (models/mocogan_ode_rnn.py:47-48 hands
 * the whole batch to one odeint call): a workgroup integrates 32 trajectories; for N > 32 the workgroups of a call
 * exchange their partial sums of squares through `sync` (gode_odernn_sync_size(N) 4-byte words, caller-owned, zeroed by
 * the launch) and add them in fixed order, so every workgroup takes bit-identical accept / reject decisions and re-runs
 * are bit-identical.  All workgroups of a launch must be co-resident: N <= GODE_ODERNN_SYNC_MAX_N per launch (summed
 * over the ops of a _multi launch); above that the call falls back to one norm per 64-trajectory workgroup (a recorded
 * deviation, unused by any configuration).  nsteps[T] (nullable): dopri5 trial steps per frame (of workgroup 0 in the
 * fallback); a negative entry reports a solve that hit the trial-step limit or whose step size underflowed. */
typedef struct gode_odernn_params {
  const float* W1; const float* b1; const float* W2; const float* b2;         /* ODEFunc */
  const float* Wih; const float* Whh; const float* bih; const float* bhh;     /* GRUCell(16,16): [48,16]x2, [48]x2 */
} gode_odernn_params;
typedef struct gode_odernn_fwd_op {
  gode_odernn_params p;
  const float* noise; const float* content; const int32_t* sel_t;
  float* z; float* hs; float* hp; int32_t* nsteps;
  int32_t N, T; float rtol, atol; int32_t zcols, pad_;
  int32_t* sync;    /* N > 32: >= gode_odernn_sync_size(N) words; ignored (may be NULL) for N <= 32 */
} gode_odernn_fwd_op;
int gode_odernn_fwd(const gode_odernn_fwd_op* op, void* stream);
#define GODE_ODERNN_SYNC_MAX_N 4096
int64_t gode_odernn_sync_size(int32_t N);
/* `count` (<= 8) independent solves in ONE launch, each with its own whole-batch norm: one training iteration's
 * sample_videos / sample_images latents depend only on the generator's weights at the start of the iteration, so their
 * one-workgroup solves run side by side on different CUs instead of back to back (ops: HOST array). */
int gode_odernn_fwd_multi(const gode_odernn_fwd_op* ops, int32_t count, void* stream);
/* backward: GRU backward + continuous adjoint of every unit-interval solve.  substeps == 0 (what the Python side
 * uses): integrated ADAPTIVELY as torchdiffeq does -- per frame one adjoint call, dopri5 (rtol, atol) on the augmented
 * state (y, a, g_theta), mixed norm (max over the RMS norms of y, a and each ODEFunc parameter tensor) over the whole
 * batch of the call (sync as above).  substeps > 0: `substeps` fixed reverse-time Kutta-3/8 steps per frame (32 agree
 * with the adaptive result to ~1e-6).  grads: 2176 floats = W1,b1,W2,b2,Wih,Whh,bih,bhh.
 * work >= gode_odernn_bwd_work_size floats.  nsteps[T] (nullable, adaptive only): trial steps per frame's adjoint call,
 * negative = the solve stalled (trial-step limit / step-size underflow; torchdiffeq asserts there). */
typedef struct gode_odernn_bwd_op {
  gode_odernn_params p;
  const float* noise; const float* hp; const int32_t* sel_t; const float* gz;
  float* work; float* grads; int32_t N, T, substeps, accumulate, zcols, pad_;
  float rtol, atol;   /* substeps == 0: adaptive adjoint (dopri5 on (y, a, g_theta) per frame, mixed norm), as torchdiffeq */
  int32_t* sync;      /* adaptive adjoint with N > 32: as in the forward op */
  int32_t* nsteps;
} gode_odernn_bwd_op;
int gode_odernn_bwd(const gode_odernn_bwd_op* op, void* stream);
int64_t gode_odernn_bwd_work_size(int32_t N);
/* the adjoints of `count` (<= 8) independent solves (the video and the image path of one generator step) in one launch;
 * every op reduces into its own grads (accumulate as given, ops are reduced in array order) */
int gode_odernn_bwd_multi(const gode_odernn_bwd_op* ops, int32_t count, void* stream);
#define GODE_ODERNN_NPARAM 2176

/* ---- loss and optimiser (mnist_moco_ode.py:86-89) ------------------------------------------------------------
 * BCE-with-logits, mean reduced, constant target: loss (+)= mean(max(x,0) - x*t + log1p(exp(-|x|))) and
 * g[i] = gscale*(sigmoid(x[i]) - t)/n.  loss is one device float; accumulate!=0 adds to it. */
typedef struct gode_bce_op {
  const float* logits; float* grad; float* loss; int64_t n; float target, gscale; int32_t accumulate, pad_;
} gode_bce_op;
int gode_bce_logits(const gode_bce_op* op, void* stream);
/* torch.optim.Adam with L2-coupled weight_decay on one tensor: g+=wd*p; m,v update; bias-corrected step. */
typedef struct gode_adam_op {
  float* p; const float* g; float* m; float* v; int64_t n;
  float lr, beta1, beta2, eps, weight_decay, gscale; int32_t step, pad_; /* step >= 1; gscale: 1/world for DP */
} gode_adam_op;
int gode_adam_l2(const gode_adam_op* op, void* stream);
/* the same update on `count` tensors in ONE launch; `table` is a DEVICE array of {p, g, m, v, n} records */
typedef struct gode_adam_tensor { float* p; const float* g; float* m; float* v; int64_t n; } gode_adam_tensor;
int gode_adam_multi(const gode_adam_tensor* table, int32_t count, int64_t max_n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float gscale, int32_t step, void* stream);

/* gode_adam_multi with the step-dependent coefficients read from device memory: coef[0] = lr / (1 - beta1^step),
 * coef[1] = sqrt(1 - beta2^step), both evaluated in double and rounded to float exactly as gode_adam_multi does.  A
 * launch captured in a HIP graph then stays valid across steps (the host rewrites coef before every replay). */
int gode_adam_multi_dev(const gode_adam_tensor* table, int32_t count, int64_t max_n, float beta1, float beta2, float eps,
                        float weight_decay, float gscale, const float* coef, void* stream);

/* out[i] = a[i]*alpha (+ out[i] if accumulate); small utility for gradient bucket handling */
int gode_scale(float* out, const float* a, int64_t n, float alpha, int accumulate, void* stream);

/* ---- program runner: executes n ops back to back on the stream (one host call per network pass) -------------*/
enum { GODE_OP_IGEMM = 1, GODE_OP_WGRAD = 2, GODE_OP_BN_FINALIZE = 3, GODE_OP_BN_BWD = 4, GODE_OP_ODE_FWD = 5,
       GODE_OP_ODE_BWD = 6, GODE_OP_BCE = 7, GODE_OP_ADAM = 8, GODE_OP_PACK = 9, GODE_OP_ODERNN_FWD = 10,
       GODE_OP_ODERNN_BWD = 11, GODE_OP_BN_APPLY = 12, GODE_OP_COL2IM = 13 };
typedef struct gode_pack_op {
  gode_conv_geom g; int32_t dir, co_canon; const float* w; float* wpack; const int32_t* co_perm;
} gode_pack_op;
int gode_run(const int32_t* kinds, const void* const* ops, int32_t n, void* stream);

int gode_version(void);
/* size of every op struct as compiled, so the ctypes mirror can assert it matches (index = GODE_OP_*) */
int gode_sizeof(int kind);

#ifdef __cplusplus
}
#endif
#endif
