"""Host-side plans: turn one network pass (generator decoder, video/image discriminator) into a fixed list of
libgode ops over pre-allocated device buffers, executed by a single gode_run call.

A ConvStack is a chain of  conv -> [train-mode BatchNorm] -> activation  layers.  Every layer stores only its RAW
(pre-BatchNorm) output, channels-last; BatchNorm + activation are applied by the consumer kernel while it loads its
operand, so each activation tensor crosses HBM once per use.  Layers are described in conv orientation
(gode_conv_geom); a ConvTranspose2d layer is the DGRAD of its geometry, so the generator decoder
(models/mocogan.py:200-215, models/mocogan_ode.py:66-84) and the discriminators (models/mocogan.py:72-89,138-159)
share all kernels and this class.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L


def make_geom(N, Ci, Co, xi, yo, k, s, p) -> L.ConvGeom:
    """xi / yo / k / s / p are (d, h, w) triples in conv orientation."""
    return L.ConvGeom(N, Ci, Co, xi[0], xi[1], xi[2], yo[0], yo[1], yo[2], k[0], k[1], k[2], s[0], s[1], s[2],
                      p[0], p[1], p[2])


def conv_out(i, k, s, p):
    return (i + 2 * p - k) // s + 1


def dptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


@dataclass
class LayerSpec:
    geom: L.ConvGeom
    fwd_dir: int                 # L.FPROP (Conv) or L.DGRAD (ConvTranspose)
    act: int                     # activation applied to this layer's output (by the consumer)
    has_bn: bool
    epilogue: int = L.EPI_RAW    # only the last layer: tanh
    co_perm: Optional[torch.Tensor] = None  # int32 device tensor (generator layer 0)

    def out_dims(self):
        g = self.geom
        if self.fwd_dir == L.FPROP:
            return (g.N, g.Do, g.Ho, g.Wo, g.Co)
        return (g.N, g.Di, g.Hi, g.Wi, g.Ci)

    def in_dims(self):
        g = self.geom
        if self.fwd_dir == L.FPROP:
            return (g.N, g.Di, g.Hi, g.Wi, g.Ci)
        return (g.N, g.Do, g.Ho, g.Wo, g.Co)


@dataclass
class LayerParams:
    weight: torch.Tensor
    gamma: Optional[torch.Tensor] = None
    beta: Optional[torch.Tensor] = None
    running_mean: Optional[torch.Tensor] = None
    running_var: Optional[torch.Tensor] = None
    num_batches_tracked: Optional[torch.Tensor] = None


def _contig_strides(dims):
    """channels-last element strides {N,D,H,W,C} of a contiguous [N,D,H,W,C] buffer."""
    n, d, h, w, c = dims
    return (d * h * w * c, h * w * c, w * c, c, 1)


class ConvStack:
    """Forward/backward programs of one conv stack at a fixed batch size.

    input: either a plan-owned channels-last buffer (`self.x_in`, generator latent rows) or a caller tensor given
    per call with element strides (discriminators read the caller's video / image in place)."""

    def __init__(self, specs: Sequence[LayerSpec], params: Sequence[LayerParams], device, owns_input: bool,
                 momentum=0.1, eps=1e-5, pack_cache: Optional[dict] = None, groups: int = 1, split_images: int = 0,
                 split_order: int = 0, two_lane_backward: bool = False):
        """groups == 2: ONE pass over two image groups [first source; second source] (a discriminator applied to real
        and fake in the same launches, specs built for the joint batch): every BatchNorm keeps per-group batch
        statistics -- exactly what the reference's two forward calls compute -- while the GEMMs, weight gradients and
        elementwise passes run once over twice the rows.  Conv (FPROP) stacks with caller-owned inputs only."""
        self.specs, self.params, self.device = list(specs), list(params), device
        self.groups = groups
        # two_lane_backward: the weight gradients of a backward pass run on a second stream next to the input-gradient /
        # BatchNorm-backward chain: wgrad(l) only reads what the chain has finished (g[l]) and nothing reads what it writes,
        # so each run of weight-gradient ops waits for ONE event on the chain's stream and the streams join at the end.
        # Same kernels on the same inputs: bit-identical.  For stacks whose GEMM grids leave workgroup slots idle (the
        # discriminators at the configs' batch sizes).
        self.two_lane = bool(two_lane_backward)
        self._lane_stream = None
        # split_images > 0 (plan-owned input only): the batch is [first split_images images; the rest] -- two BatchNorm
        # batches of DIFFERENT size decoded in one pass (the generator's video and image paths: the reference calls `main`
        # on each, models/mocogan.py:276,293): per-part batch statistics, every GEMM / elementwise launch once over all
        # rows.  split_order: which part the reference called first (its momentum update of the running statistics comes
        # first).  The statistics rows of every layer must split at a tile boundary (gode_igemm_stats_segments); the
        # constructor raises ValueError otherwise and the caller runs the two parts separately.
        self.split, self.split_order = int(split_images), int(split_order)
        if self.split and (groups != 1 or not owns_input or not (0 < self.split < specs[0].geom.N)):
            raise ValueError("split ConvStack: plan-owned input, one launch group, 0 < split_images < N")
        self._G2 = groups == 2 or self.split > 0          # two BatchNorm groups: [2][C] statistics arrays
        self._last_parts = False
        if groups not in (1, 2) or (groups == 2 and (owns_input or specs[0].fwd_dir != L.FPROP or specs[0].has_bn
                                                     or specs[0].geom.N % 2)):
            raise ValueError("grouped ConvStack: two groups, Conv stack, caller-owned input, no BatchNorm on layer 0")
        self.nl = len(specs)
        self.momentum, self.eps = momentum, eps
        self.busy = False
        self.generation = 0      # bumped by every forward; an autograd node checks it before using the plan's buffers
        self._fwd_training = True
        # packed weight panels do not depend on the batch size: all plans of one module share them (and the record
        # of which weight version each panel was packed from)
        self.pack_cache = pack_cache if pack_cache is not None else {}
        lib = L.lib()
        f32 = dict(dtype=torch.float32, device=device)
        self.x_in = torch.zeros(specs[0].in_dims(), **f32) if owns_input else None
        # raw outputs of all but the last layer are plan-owned; the last is allocated per call
        self.y = [torch.empty(s.out_dims(), **f32) for s in specs[:-1]]
        # activated copies a[l] = act(BN(y[l])) for layers whose consumer is a heavy GEMM (see gode_bn_apply)
        self.a = [torch.empty(s.out_dims(), **f32) if self._materialize(l) else None for l, s in enumerate(specs[:-1])]
        self.out_dims = specs[-1].out_dims()
        self.wpack_f, self.wpack_b = [], [None] * self.nl
        self.stats, self.stat_rows = [], []
        self.mean, self.invstd, self.scale, self.shift = [], [], [], []
        # split stacks: layers whose GEMM runs as TWO launches (one per part) because the cost model prices the joint grid
        # above the two parts' grids -- a batch of 512 + 32 images whose 512 fill whole rounds of 256 workgroups exactly
        # spills 6 % of a round into a round of its own.  Everything else of the layer (statistics finalize, BatchNorm
        # apply / backward, weight gradient) still runs once over all rows.
        self._two_f, self._two_b, self._rows_ab = {}, {}, {}
        for l, s in enumerate(specs):
            n = lib.gode_pack_size(C.byref(s.geom), s.fwd_dir)
            if n <= 0:
                raise RuntimeError(f"bad geometry for pack ({n})")
            self.wpack_f.append(self._shared_pack(l, s.fwd_dir, n))
            C_out = s.out_dims()[4]
            two = self._two_f[l] = bool(self.split) and 0 < l < self.nl - 1 and self._two_launches(s.geom, s.fwd_dir)
            if s.has_bn:
                probe = L.IgemmOp(g=s.geom, dir=s.fwd_dir, tile=0, groups=groups)
                if two:     # the parts' partial sums: two arrays [2][ncols][rows of the part], one after the other
                    ra, rb = (lib.gode_igemm_stats_rows(C.byref(L.IgemmOp(g=self._part_geom(s.geom, n), dir=s.fwd_dir, tile=0)))
                              for n in (self.split, s.geom.N - self.split))
                    rows = ra + rb if ra > 0 and rb > 0 else -1
                    self._rows_ab[l] = (ra, rb)
                else:
                    rows = lib.gode_igemm_stats_rows(C.byref(probe))
                if rows <= 0:
                    raise RuntimeError(f"layer {l}: no partial-statistics layout for this geometry ({rows})")
                ncols = self._ncols(s)
                self.stat_rows.append(rows)
                self.stats.append(torch.empty(rows * 2 * ncols, **f32))
                # per-group [groups][C] (group-major)
                ng = 2 if self._G2 else 1
                self.mean.append(torch.empty(ng * C_out, **f32)); self.invstd.append(torch.empty(ng * C_out, **f32))
                self.scale.append(torch.empty(ng * C_out, **f32)); self.shift.append(torch.empty(ng * C_out, **f32))
                if self.split and not two:
                    seg = (C.c_int32 * 24)()
                    nseg = lib.gode_igemm_stats_segments(C.byref(probe), self.split, seg)
                    if nseg <= 0:
                        raise ValueError(f"layer {l}: the statistics rows do not split at image {self.split} ({nseg})")
                    self.__dict__.setdefault("_segs", {})[l] = (nseg, list(seg))
            else:
                self.stat_rows.append(0); self.stats.append(None)
                self.mean.append(None); self.invstd.append(None); self.scale.append(None); self.shift.append(None)
        self._ig_work = None    # split-K workspace shared by the plan's igemm ops (they run back to back)
        self._fwd = {}          # training flag -> Program
        self._bwd = {}          # (need_input_grad, need_param_grad, forward was train-mode) -> Program
        self.g = None           # gradient buffers (lazy)
        self._param_ptrs = None

    # -- helpers -------------------------------------------------------------------------------------------
    def _shared_pack(self, l, direction, n):
        key = ("buf", l, direction)
        buf = self.pack_cache.get(key)
        if buf is None or buf.numel() != n or buf.device != torch.device(self.device):
            buf = self.pack_cache[key] = torch.empty(n, dtype=torch.float32, device=self.device)
            self.pack_cache.pop(("ver", l, direction), None)
        return buf

    def _attach_work(self, igemm_ops):
        """Give every igemm op of a program the shared split-K workspace (sized for the largest request)."""
        lib = L.lib()
        need = max([lib.gode_igemm_work_size(C.byref(op)) for op in igemm_ops] + [0])
        if need > 0:
            if self._ig_work is None or self._ig_work.numel() < need:
                self._ig_work = torch.empty(need, dtype=torch.float32, device=self.device)
                for prog_ops in getattr(self, "_all_igemm", []):
                    for op in prog_ops:
                        op.work = self._ig_work.data_ptr()
            for op in igemm_ops:
                op.work = self._ig_work.data_ptr()
        self.__dict__.setdefault("_all_igemm", []).append(list(igemm_ops))

    def _run_stale_packs(self, packs, st):
        """packs: [(layer, dir, PackOp)].  Re-packs only panels whose weight changed since they were last packed."""
        stale = []
        for l, d, op in packs:
            w = self.params[l].weight
            ver = (w._version, getattr(w, "_gode_ver", 0), w.data_ptr())
            if self.pack_cache.get(("ver", l, d)) != ver:
                self.pack_cache[("ver", l, d)] = ver
                stale.append(op)
        if stale:
            L.Program(stale).run(st)

    @staticmethod
    def _ncols(s: LayerSpec):
        g = s.geom
        if s.fwd_dir == L.FPROP:
            return g.Co
        fullk = (g.Do == 1 and g.Ho == 1 and g.Wo == 1 and g.pd == 0 and g.ph == 0 and g.pw == 0 and
                 g.Di == g.kd and g.Hi == g.kh and g.Wi == g.kw and g.kd * g.kh * g.kw > 1)
        return g.kd * g.kh * g.kw * g.Ci if fullk else g.Ci

    def _materialize(self, l):
        """Materialise act(BN(y[l])) when the consuming layer l+1 is a heavy vector-path GEMM."""
        if l + 1 >= len(self.specs):
            return False
        prod, cons = self.specs[l], self.specs[l + 1]
        C_out = prod.out_dims()[4]
        g = cons.geom
        taps = g.kd * g.kh * g.kw
        macs = g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * taps
        heavy = C_out % 32 == 0 and macs >= (1 << 30) and (prod.has_bn or prod.act != L.ACT_NONE)
        if self._G2 and prod.has_bn:
            # per-group scale/shift: consumers read the activated copy, nothing downstream knows of groups -- except the light
            # LAST layer of a split stack (the generator's output convolution: its input is the largest activation of the
            # pass), which runs as one launch per part with that part's scale / shift instead (self._last_parts)
            if self.split and not heavy and l + 1 == self.nl - 1 and cons.epilogue == L.EPI_TANH and not cons.has_bn:
                self._last_parts = True
                return False
            return True
        return heavy

    def _head_as_gemm(self, l):
        """A thin-output ConvTranspose2d at the end of the stack (<= 4 output channels, > 1 tap, many pixels: the UCF
        generator's RGB head) runs as one plain GEMM over the input pixels + an overlap-add pass (gode_col2im):
        every input activation is read once instead of once per tap and phase.  Returns the GEMM's geometry or None."""
        s = self.specs[l]
        g = s.geom
        taps = g.kh * g.kw
        pixels = g.N * g.Ho * g.Wo
        if (l != self.nl - 1 or self.groups != 1 or s.fwd_dir != L.DGRAD or s.has_bn or g.kd != 1 or g.Di != 1 or g.Do != 1
                or g.Ci > 4 or taps == 1 or taps * g.Ci <= 4 or g.Co % 32 != 0 or pixels < 16384
                or (g.Ho == 1 and g.Wo == 1) or s.epilogue not in (L.EPI_RAW, L.EPI_TANH)):
            return None
        return make_geom(pixels, g.Ci, g.Co, (1, g.kh, g.kw), (1, 1, 1), (1, g.kh, g.kw), (1, 1, 1), (0, 0, 0))

    def _tail_as_gemm(self):
        """The input gradient of a thin-INPUT first layer with many taps (<= 4 input channels, >= 64 weight columns per output
        channel, many positions: the UCF video discriminator's Conv3d(3, 64, 4, ...)) as ONE plain GEMM over the layer's
        output positions + an overlap-add (gode_col2im, 3-D form): every gradient value is read once instead of once per
        tap of every input voxel it touches.  Returns the GEMM's geometry or None."""
        s = self.specs[0]
        g = s.geom
        taps = g.kd * g.kh * g.kw
        pos = g.N * g.Do * g.Ho * g.Wo
        if (self.groups != 1 or s.fwd_dir != L.FPROP or g.Ci > 4 or taps * g.Ci < 64 or g.Co % 32 != 0 or pos < 16384 or
                g.Di != (g.Do - 1) * g.sd - 2 * g.pd + g.kd or g.Hi != (g.Ho - 1) * g.sh - 2 * g.ph + g.kh or
                g.Wi != (g.Wo - 1) * g.sw - 2 * g.pw + g.kw):
            return None
        return make_geom(pos, g.Ci, g.Co, (g.kd, g.kh, g.kw), (1, 1, 1), (g.kd, g.kh, g.kw), (1, 1, 1), (0, 0, 0))

    def _in_src(self, l):
        """Tensor the consumer layer l reads (activated copy if materialised, else the raw producer output)."""
        if l == 0:
            return self.x_in
        return self.a[l - 1] if self.a[l - 1] is not None else self.y[l - 1]

    def _in_xform(self, l):
        """(scale, shift, act) the consumer of layer l-1's output applies on load."""
        if l == 0 or self.a[l - 1] is not None:
            return None, None, L.ACT_NONE
        s = self.specs[l - 1]
        return (self.scale[l - 1], self.shift[l - 1], s.act) if s.has_bn else (None, None, s.act)

    def _count(self, l):
        d = self.specs[l].out_dims()
        return d[0] * d[1] * d[2] * d[3]

    def param_ptrs(self):
        out = []
        for p in self.params:
            for t in (p.weight, p.gamma, p.beta, p.running_mean, p.running_var, p.num_batches_tracked):
                out.append(None if t is None else t.data_ptr())
        return tuple(out)

    def _refresh(self):
        """Programs hold raw parameter pointers; rebuild them if a module was moved/re-allocated."""
        ptrs = self.param_ptrs()
        if ptrs != self._param_ptrs:
            self._fwd, self._bwd, self._param_ptrs = {}, {}, ptrs
            self._all_igemm = []

    def invalidate_packs(self):
        """Forget which weight versions the packed panels were built from (next pass re-packs all of them)."""
        for k in [k for k in self.pack_cache if k[0] == "ver"]:
            del self.pack_cache[k]

    def weights_key(self):
        """Changes whenever any conv weight may have changed: torch's in-place version counter (torch optimisers,
        load_state_dict) plus the counter FusedAdam bumps (it updates through raw pointers)."""
        return tuple((p.weight._version, getattr(p.weight, "_gode_ver", 0)) for p in self.params)

    # -- forward -------------------------------------------------------------------------------------------
    @staticmethod
    def _part_geom(g, n):
        h = L.ConvGeom(*g.key())
        h.N = n
        return h

    def _two_launches(self, g, direction):
        """Split stack: is this layer's GEMM cheaper as one launch per part (cost model of the library)?"""
        lib = L.lib()
        cost = [lib.gode_igemm_model_cycles(C.byref(L.IgemmOp(g=self._part_geom(g, n), dir=direction, tile=0)))
                for n in (g.N, self.split, g.N - self.split)]
        return min(cost) > 0 and cost[1] + cost[2] + 12000.0 < cost[0]        # (+ ~5 us for the second launch)

    def _rank1_head(self, l, has_tanh):
        """Is layer l the stack's head AND a 1x1, stride-1 transposed convolution to ONE channel on a centre crop behind a
        BatchNorm layer (the MNIST generator's output layer)?  Its input gradient is then rank 1 (see _build_bwd)."""
        if l != self.nl - 1 or l == 0 or not has_tanh:
            return False
        s, g = self.specs[l], self.specs[l].geom
        return (s.fwd_dir == L.DGRAD and not s.has_bn and s.co_perm is None and self.specs[l - 1].has_bn and g.Ci == 1 and
                g.kd == g.kh == g.kw == 1 and g.sd == g.sh == g.sw == 1 and g.Di == g.Do == 1 and g.pd == 0 and
                g.ph == g.pw and g.Ho == g.Hi + 2 * g.ph and g.Wo == g.Wi + 2 * g.pw and self.params[l].weight.is_contiguous()
                and self.params[l].weight.data_ptr() % 16 == 0)

    def _half_geom(self, g):
        h = L.ConvGeom(*g.key())
        h.N = g.N // 2
        return h

    def _build_fwd(self, training: bool):
        ops = []
        packs = []
        patch = {}
        lib = L.lib()
        G2 = self.groups == 2
        for l, (s, p) in enumerate(zip(self.specs, self.params)):
            packs.append((l, s.fwd_dir, L.PackOp(g=s.geom, dir=s.fwd_dir, co_canon=0, w=dptr(p.weight),
                                                 wpack=dptr(self.wpack_f[l]), co_perm=dptr(s.co_perm))))
            sc, sh, act = self._in_xform(l)
            src = self._in_src(l)
            if G2 and l == 0:
                # the two groups come from two caller tensors: one launch each into the halves of y[0]
                hg = self._half_geom(s.geom)
                half = self.y[0].numel() // 2
                pair = []
                for k in range(2):
                    o = L.IgemmOp(g=hg, dir=s.fwd_dir, act=act, epilogue=s.epilogue, tile=0, src=None,
                                  wpack=dptr(self.wpack_f[l]), out=self.y[0].data_ptr() + 4 * half * k)
                    pair.append(o)
                    ops.append(o)
                patch["first"], patch["first2"] = pair
                if self.a[0] is not None:          # activated copy for a heavy consumer (no BatchNorm on layer 0)
                    d = s.out_dims()
                    ops.append(L.BnApplyOp(y=dptr(self.y[0]), out=dptr(self.a[0]), scale=None, shift=None,
                                           M=d[0] * d[1] * d[2] * d[3], C=d[4], act=s.act))
                continue
            hg = self._head_as_gemm(l)
            if hg is not None:
                n = lib.gode_pack_size(C.byref(hg), L.DGRAD)
                wp = self._shared_pack(l, "head_gemm", n)
                packs.append((l, "head_gemm", L.PackOp(g=hg, dir=L.DGRAD, co_canon=0, w=dptr(p.weight), wpack=dptr(wp),
                                                       co_perm=dptr(s.co_perm))))
                g = s.geom
                if getattr(self, "_head_cols", None) is None:
                    self._head_cols = torch.empty(hg.N * g.kh * g.kw * g.Ci, dtype=torch.float32, device=self.device)
                if self._last_parts:      # one pixel GEMM per part (that part's BatchNorm scale / shift), one overlap-add
                    pix, Cs = g.Ho * g.Wo, sc.numel() // 2
                    for n, img0, k in ((self.split, 0, 0), (g.N - self.split, self.split, 1)):
                        pg = self._part_geom(hg, n * pix)
                        ops.append(L.IgemmOp(g=pg, dir=L.DGRAD, act=act, epilogue=L.EPI_RAW, tile=0,
                                             src=src.data_ptr() + 4 * img0 * pix * g.Co, wpack=dptr(wp),
                                             out=self._head_cols.data_ptr() + 4 * img0 * pix * g.kh * g.kw * g.Ci,
                                             scale=sc.data_ptr() + 4 * Cs * k, shift=sh.data_ptr() + 4 * Cs * k))
                else:
                    ops.append(L.IgemmOp(g=hg, dir=L.DGRAD, act=act, epilogue=L.EPI_RAW, tile=0, src=dptr(src), wpack=dptr(wp),
                                         out=dptr(self._head_cols), scale=dptr(sc), shift=dptr(sh)))
                c2i = L.Col2imOp(cols=dptr(self._head_cols), out=None, N=g.N, Hi=g.Ho, Wi=g.Wo, Ho=g.Hi, Wo=g.Wi, C=g.Ci,
                                 kh=g.kh, kw=g.kw, sh=g.sh, sw=g.sw, ph=g.ph, pw=g.pw, epilogue=s.epilogue)
                ops.append(c2i)
                patch["last"] = c2i
                if l == 0:
                    raise RuntimeError("a one-layer stack cannot use the GEMM + col2im head")
                continue
            want_stats = s.has_bn and training
            if self._last_parts and l == self.nl - 1:
                # (split stack, light output layer: one launch per part with that part's BatchNorm scale / shift)
                per_in, per_out, Cs = int(np.prod(s.in_dims()[1:])), int(np.prod(s.out_dims()[1:])), sc.numel() // 2
                pair = []
                for n, img0, k in ((self.split, 0, 0), (s.geom.N - self.split, self.split, 1)):
                    pair.append(L.IgemmOp(g=self._part_geom(s.geom, n), dir=s.fwd_dir, act=act, epilogue=s.epilogue, tile=0,
                                          src=src.data_ptr() + 4 * img0 * per_in, wpack=dptr(self.wpack_f[l]), out=None,
                                          scale=sc.data_ptr() + 4 * Cs * k, shift=sh.data_ptr() + 4 * Cs * k))
                ops += pair
                patch["last"], patch["last2"] = pair[0], (pair[1], 4 * self.split * per_out)
                continue
            if self._two_f.get(l):
                ra = self._rows_ab[l][0] if s.has_bn else 0
                per_in, per_out = int(np.prod(s.in_dims()[1:])), int(np.prod(s.out_dims()[1:]))
                for n, img0, st_off in ((self.split, 0, 0), (s.geom.N - self.split, self.split, ra * 2 * self._ncols(s))):
                    ops.append(L.IgemmOp(g=self._part_geom(s.geom, n), dir=s.fwd_dir, act=act, epilogue=s.epilogue, tile=0,
                                         src=src.data_ptr() + 4 * img0 * per_in, wpack=dptr(self.wpack_f[l]),
                                         out=self.y[l].data_ptr() + 4 * img0 * per_out, scale=dptr(sc), shift=dptr(sh),
                                         stats=self.stats[l].data_ptr() + 4 * st_off if want_stats else None, groups=0))
                op = None
            else:
                op = L.IgemmOp(g=s.geom, dir=s.fwd_dir, act=act, epilogue=s.epilogue, tile=0, src=dptr(src),
                               wpack=dptr(self.wpack_f[l]), out=dptr(self.y[l]) if l < self.nl - 1 else None,
                               scale=dptr(sc), shift=dptr(sh),
                               stats=dptr(self.stats[l]) if want_stats else None,
                               groups=2 if (G2 and s.has_bn) else 0)
                ops.append(op)
            if l == 0:
                patch["first"] = op
            if l == self.nl - 1:
                patch["last"] = op
            if s.has_bn:
                rows0 = lib.gode_igemm_stats_rows0(C.byref(op)) if G2 else 0
                fin = L.BnFinalizeOp(stats=dptr(self.stats[l]), rows=self.stat_rows[l], ncols=self._ncols(s),
                                     C=s.out_dims()[4], count=self._count(l) // self.groups, gamma=dptr(p.gamma),
                                     beta=dptr(p.beta), running_mean=dptr(p.running_mean),
                                     running_var=dptr(p.running_var),
                                     num_batches_tracked=dptr(p.num_batches_tracked), mean=dptr(self.mean[l]),
                                     invstd=dptr(self.invstd[l]), scale=dptr(self.scale[l]),
                                     shift=dptr(self.shift[l]), momentum=self.momentum, eps=self.eps,
                                     training=1 if training else 0, groups=self.groups if G2 else 0, rows0=rows0)
                if self.split:
                    per_img = self._count(l) // s.geom.N
                    fin.groups, fin.order = 2, self.split_order
                    fin.count, fin.count1 = per_img * self.split, per_img * (s.geom.N - self.split)
                    if self._two_f.get(l):
                        ra, rb = self._rows_ab[l]
                        fin.rows, fin.rows1 = ra, rb
                        fin.stats1 = self.stats[l].data_ptr() + 4 * ra * 2 * self._ncols(s)
                    else:
                        nseg, seg = self._segs[l]
                        fin.nseg = nseg
                        for k in range(3 * nseg):
                            fin.seg[k] = seg[k]
                ops.append(fin)
            if l < self.nl - 1 and self.a[l] is not None:
                d = s.out_dims()
                M = d[0] * d[1] * d[2] * d[3]
                M0 = M // 2 if (G2 and s.has_bn) else ((M // d[0]) * self.split if (self.split and s.has_bn) else 0)
                ops.append(L.BnApplyOp(y=dptr(self.y[l]), out=dptr(self.a[l]), scale=dptr(self.scale[l]),
                                       shift=dptr(self.shift[l]), M=M, C=d[4], act=s.act, M0=M0))
        patch["packs"] = packs
        self._attach_work([op for op in ops if isinstance(op, L.IgemmOp)])
        return L.Program(ops), patch

    def forward(self, training: bool, x: Optional[torch.Tensor] = None, x_strides=None, pre_ops_program=None,
                x2: Optional[torch.Tensor] = None, x2_strides=None):
        """Runs the stack.  x (+ element strides {N,D,H,W,C}) is required when the plan does not own its input; a
        grouped stack takes the second group's tensor as x2.  Returns the freshly allocated last-layer output
        [N,D,H,W,C] (grouped: N = both groups, first group first)."""
        self._refresh()
        if training not in self._fwd:
            self._fwd[training] = self._build_fwd(training)
        prog, patch = self._fwd[training]
        out = torch.empty(self.out_dims, dtype=torch.float32, device=self.device)
        patch["last"].out = out.data_ptr()
        if "last2" in patch:
            patch["last2"][0].out = out.data_ptr() + patch["last2"][1]
        if self.x_in is None:
            first = patch["first"]
            first.src = x.data_ptr()
            for i in range(5):
                first.gs[i] = int(x_strides[i])
            # aliases WITHOUT autograd history: the plan must not keep the caller's graph (or, below, its own autograd
            # node, which holds the plan's lease) alive
            self._x_user, self._x_strides = x.detach(), tuple(int(v) for v in x_strides)
            if self.groups == 2:
                second = patch["first2"]
                second.src = x2.data_ptr()
                for i in range(5):
                    second.gs[i] = int(x2_strides[i])
                self._x2_user, self._x2_strides = x2.detach(), tuple(int(v) for v in x2_strides)
        st = stream_ptr()
        self._run_stale_packs(patch["packs"], st)
        if pre_ops_program is not None:
            pre_ops_program.run(st)
        prog.run(st)
        self.out = out.detach()      # `out` itself becomes the autograd node's output: holding it would be a cycle
        self.generation += 1
        self._fwd_training = training
        return out

    # -- backward ------------------------------------------------------------------------------------------
    def n_grad_floats(self):
        n = 0
        for p in self.params:
            n += p.weight.numel()
            if p.gamma is not None:
                n += p.gamma.numel() + p.beta.numel()
        return n

    def _build_bwd(self, need_input_grad: bool, need_param_grad: bool = True, training: bool = True):
        lib = L.lib()
        f32 = dict(dtype=torch.float32, device=self.device)
        if self.g is None:
            self.g = [torch.empty(s.out_dims(), **f32) for s in self.specs]  # grad wrt raw/activated outputs
            self.g_in = None
        ops, patch = [], {"dw": [], "dgamma": [], "dbeta": []}
        bpacks = []
        wg_work = 0
        bn_work = 0
        last = self.specs[-1]
        # ops that read the upstream gradient: patched per call to the caller's tensor when it is contiguous (no
        # staging copy); with a tanh epilogue only its backward reads it (and writes g[-1] for the others)
        readers = patch["gout_readers"] = []
        if last.epilogue == L.EPI_TANH:
            M = self._count(self.nl - 1)
            op = L.BnBwdOp(g=dptr(self.g[-1]), y=None, M=M, C=last.out_dims()[4], act=L.ACT_TANH_OUT)
            patch["tanh"] = op
            readers.append((op, "gin"))
            ops.append(op)
        for l in range(self.nl - 1, -1, -1):
            s, p = self.specs[l], self.params[l]
            rev = L.DGRAD if s.fwd_dir == L.FPROP else L.FPROP
            sc, sh, act = self._in_xform(l)
            src = self._in_src(l)
            # weight gradient
            if self.groups == 2 and l == 0:
                # two caller tensors: one launch per group over its half of g[0], the second adds to the first
                hg = self._half_geom(s.geom)
                half = self.g[0].numel() // 2
                pair = []
                for k in range(2):
                    pair.append(L.WgradOp(g=hg, act=act, xform_on_y=0, splits=0, accumulate=0, x=None,
                                          y=self.g[0].data_ptr() + 4 * half * k, scale=dptr(sc), shift=dptr(sh)))
                if need_param_grad:
                    for k, wk in enumerate(pair):
                        wg_work = max(wg_work, lib.gode_wgrad_work_size(C.byref(wk)))
                        patch["dw"].append((l, wk, k == 1))
                        ops.append(wk)
                    patch["wgrad0"], patch["wgrad0b"] = pair
                if need_input_grad:
                    raise NotImplementedError("a grouped stack does not return input gradients")
                continue
            if self._last_parts and l == self.nl - 1:
                # (split stack, light output layer: its input is read raw with the PART's BatchNorm scale / shift -- one
                # weight-gradient launch per part, the second adds to the first)
                if s.fwd_dir != L.DGRAD or "tanh" not in patch:
                    raise NotImplementedError("split stack: per-part output layer is a transposed convolution with tanh")
                per_in, per_out, Cs = int(np.prod(s.in_dims()[1:])), int(np.prod(s.out_dims()[1:])), sc.numel() // 2
                if need_param_grad:
                    for n, img0, k in ((self.split, 0, 0), (s.geom.N - self.split, self.split, 1)):
                        wk = L.WgradOp(g=self._part_geom(s.geom, n), act=act, xform_on_y=1, splits=0, accumulate=0,
                                       x=self.g[l].data_ptr() + 4 * img0 * per_out, y=src.data_ptr() + 4 * img0 * per_in,
                                       scale=sc.data_ptr() + 4 * Cs * k, shift=sh.data_ptr() + 4 * Cs * k, co_perm=dptr(s.co_perm))
                        wg_work = max(wg_work, lib.gode_wgrad_work_size(C.byref(wk)))
                        patch["dw"].append((l, wk, k == 1))
                        ops.append(wk)
                w = None
            elif s.fwd_dir == L.FPROP:
                w = L.WgradOp(g=s.geom, act=act, xform_on_y=0, splits=0, accumulate=0, x=dptr(src), y=dptr(self.g[l]),
                              scale=dptr(sc), shift=dptr(sh), co_perm=dptr(s.co_perm))
            else:
                w = L.WgradOp(g=s.geom, act=act, xform_on_y=1, splits=0, accumulate=0, x=dptr(self.g[l]), y=dptr(src),
                              scale=dptr(sc), shift=dptr(sh), co_perm=dptr(s.co_perm))
            if need_param_grad and w is not None:
                wg_work = max(wg_work, lib.gode_wgrad_work_size(C.byref(w)))
                patch["dw"].append((l, w, False))
                if l == 0:
                    patch["wgrad0"] = w
                if l == self.nl - 1 and "tanh" not in patch:
                    readers.append((w, "y" if s.fwd_dir == L.FPROP else "x"))
                ops.append(w)
            # input gradient
            if l > 0 or need_input_grad:
                rank1 = self._rank1_head(l, "tanh" in patch)
                tg = self._tail_as_gemm() if l == 0 else None
                dst = self.g_in if l == 0 else self.g[l - 1]
                if tg is not None:
                    # thin-input first layer: one GEMM over its output positions + a 3-D overlap-add (see _tail_as_gemm)
                    g = s.geom
                    wp = self._shared_pack(0, "tail_gemm", lib.gode_pack_size(C.byref(tg), L.DGRAD))
                    bpacks.append((0, "tail_gemm", L.PackOp(g=tg, dir=L.DGRAD, co_canon=0, w=dptr(p.weight), wpack=dptr(wp),
                                                            co_perm=dptr(s.co_perm))))
                    if getattr(self, "_tail_cols", None) is None:
                        self._tail_cols = torch.empty(tg.N * g.kd * g.kh * g.kw * g.Ci, **f32)
                    ops.append(L.IgemmOp(g=tg, dir=L.DGRAD, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=0, src=dptr(self.g[0]),
                                         wpack=dptr(wp), out=dptr(self._tail_cols)))
                    c2i = L.Col2imOp(cols=dptr(self._tail_cols), out=None, N=g.N, Hi=g.Ho, Wi=g.Wo, Ho=g.Hi, Wo=g.Wi, C=g.Ci,
                                     kh=g.kh, kw=g.kw, sh=g.sh, sw=g.sw, ph=g.ph, pw=g.pw, epilogue=L.EPI_RAW,
                                     Di=g.Do, Do=g.Di, kd=g.kd, sd=g.sd, pd=g.pd)
                    ops.append(c2i)
                    patch["dgrad0"] = c2i               # (its output, the input gradient, is allocated per call)
                elif rank1:
                    # the head is a 1x1 convolution to one channel: its input gradient is w (x) g -- rank 1 -- and the
                    # BatchNorm backward below forms it on the fly (gode_bn_bwd_op.r1_s) instead of reading it back: one
                    # 142-MB tensor less written and two less read per generator backward at configs[1]
                    pass
                else:
                    if self.wpack_b[l] is None:
                        self.wpack_b[l] = self._shared_pack(l, rev, lib.gode_pack_size(C.byref(s.geom), rev))
                    bpacks.append((l, rev, L.PackOp(g=s.geom, dir=rev, co_canon=0, w=dptr(p.weight),
                                                    wpack=dptr(self.wpack_b[l]), co_perm=dptr(s.co_perm))))
                    if l not in self._two_b:
                        self._two_b[l] = bool(self.split) and 0 < l < self.nl - 1 and self._two_launches(s.geom, rev)
                    if self._two_b[l]:        # (split stack: one launch per part, see _two_f)
                        per_in, per_out = int(np.prod(s.in_dims()[1:])), int(np.prod(s.out_dims()[1:]))
                        for n, img0 in ((self.split, 0), (s.geom.N - self.split, self.split)):
                            ops.append(L.IgemmOp(g=self._part_geom(s.geom, n), dir=rev, act=L.ACT_NONE, epilogue=L.EPI_RAW,
                                                 tile=0, src=self.g[l].data_ptr() + 4 * img0 * per_out,
                                                 wpack=dptr(self.wpack_b[l]), out=dst.data_ptr() + 4 * img0 * per_in))
                    else:
                        ig = L.IgemmOp(g=s.geom, dir=rev, act=L.ACT_NONE, epilogue=L.EPI_RAW, tile=0, src=dptr(self.g[l]),
                                       wpack=dptr(self.wpack_b[l]), out=dptr(dst))
                        if l == self.nl - 1 and "tanh" not in patch:
                            readers.append((ig, "src"))
                        if l == 0:
                            patch["dgrad0"] = ig        # its output (the input gradient) is allocated per call
                        ops.append(ig)
            if l > 0:
                sp, pp = self.specs[l - 1], self.params[l - 1]
                M, Cc = self._count(l - 1), sp.out_dims()[4]
                if sp.has_bn:
                    b = L.BnBwdOp(g=dptr(self.g[l - 1]), y=dptr(self.y[l - 1]), M=M, C=Cc, act=sp.act,
                                  gamma=dptr(pp.gamma), mean=dptr(self.mean[l - 1]), invstd=dptr(self.invstd[l - 1]),
                                  scale=dptr(self.scale[l - 1]), shift=dptr(self.shift[l - 1]), accumulate=0,
                                  eval_mode=0 if training else 1, groups=2 if self._G2 else 0,
                                  M0=(M // sp.out_dims()[0]) * self.split if self.split else 0)
                    if self._rank1_head(l, "tanh" in patch):
                        g = s.geom           # (FPROP view of the head: Ci = 1 on Hi x Wi  ->  Co channels on Ho x Wo, pad = crop offset)
                        b.r1_s, b.r1_w = dptr(self.g[l]), dptr(p.weight)
                        b.r1_H, b.r1_W, b.r1_h, b.r1_wd, b.r1_off = g.Ho, g.Wo, g.Hi, g.Wi, g.ph
                    # (groups == 2: per-group batch statistics ([2][C] arrays) in ONE reduce / finalize / apply triple;
                    # dgamma / dbeta receive both groups' sums in group order)
                    bn_work = max(bn_work, lib.gode_bn_bwd_work_size(M, Cc))
                    if need_param_grad:
                        patch["dgamma"].append((l - 1, b, False))
                else:
                    b = L.BnBwdOp(g=dptr(self.g[l - 1]), y=dptr(self.y[l - 1]), M=M, C=Cc, act=sp.act)
                patch.setdefault("bnb", []).append(b)
                ops.append(b)
        wg_buf = torch.empty(max(wg_work, 1), **f32)
        bn_buf = torch.empty(max(bn_work, 1), **f32)
        patch["keep"] = (wg_buf, bn_buf)      # owned by this program (several backward programs coexist)
        for _, w, _f in patch["dw"]:
            w.work = wg_buf.data_ptr()
        for b in patch.get("bnb", []):
            if b.mean:
                b.work = bn_buf.data_ptr()
        patch["packs"] = bpacks
        self._attach_work([op for op in ops if isinstance(op, L.IgemmOp)])
        if self.two_lane and need_param_grad:
            # maximal runs of weight-gradient ops ("W") and of everything else ("C", the chain): a W run depends on the C ops
            # before it, no C op depends on a W op
            segs, run, kind = [], [], None
            for op in ops:
                k = "W" if isinstance(op, L.WgradOp) else "C"
                if k != kind and run:
                    segs.append((kind, L.Program(run)))
                    run = []
                kind = k
                run.append(op)
            if run:
                segs.append((kind, L.Program(run)))
            if sum(1 for k, _ in segs if k == "W") > 0 and sum(1 for k, _ in segs if k == "C") > 0:
                patch["segments"] = segs
        return L.Program(ops), patch

    def _run_backward(self, prog, patch):
        """One backward program on the current stream -- or (two_lane_backward) its runs of weight-gradient ops on the lane
        stream, each behind an event that says the chain has produced what they read, joined before returning.  Not inside a
        stream capture: hipStreamEndCapture of ROCm 7.2 crashes on this fork pattern (several event edges into one forked
        stream from a stream that is itself a fork); a captured graph keeps the single-lane order."""
        segs = patch.get("segments")
        if segs is None or L.TRACE is not None or torch.cuda.is_current_stream_capturing():
            prog.run(stream_ptr())
            return
        main = torch.cuda.current_stream()
        if self._lane_stream is None or self._lane_stream.device != main.device:
            self._lane_stream = torch.cuda.Stream(device=main.device)
        side = self._lane_stream
        for kind, seg in segs:
            if kind == "C":
                seg.run(main.cuda_stream)
                continue
            ev = torch.cuda.Event()          # (a fresh event per dependency)
            ev.record(main)
            side.wait_event(ev)
            seg.run(side.cuda_stream)
        main.wait_stream(side)

    def _patch_user_input(self, patch):
        if self.x_in is None and "wgrad0" in patch:
            if self.specs[0].fwd_dir != L.FPROP:
                raise RuntimeError("caller-owned input is only supported for Conv (FPROP) stacks")
            srcs = [(patch["wgrad0"], self._x_user, self._x_strides)]
            if self.groups == 2:
                srcs.append((patch["wgrad0b"], self._x2_user, self._x2_strides))
            for w0, xu, xs in srcs:
                w0.x = xu.data_ptr()
                for i in range(5):
                    w0.xs[i] = xs[i]

    def backward(self, gout: torch.Tensor, need_input_grad: bool, need_param_grad: bool = True, into=None):
        """gout: gradient wrt the (activated) output, any strides, logical dims = self.out_dims.
        Returns (flat parameter gradients or None, per-layer views list, input gradient buffer or None).
        into: optional per-layer [(weight_grad, gamma_grad, beta_grad, accumulate)] -- the kernels then write (or
        add to) those caller-owned tensors directly and (None, None, g_in) is returned."""
        self._refresh()
        # an eval-mode forward normalised with the running statistics: its backward has no batch-statistics terms
        key = (need_input_grad, need_param_grad, self._fwd_training)
        if key not in self._bwd:
            self._bwd[key] = self._build_bwd(need_input_grad, need_param_grad, self._fwd_training)
        prog, patch = self._bwd[key]
        self._run_stale_packs(patch["packs"], stream_ptr())
        if gout.is_contiguous() and gout.numel() == self.g[-1].numel():
            gp = gout.data_ptr()                 # read in place (the kernels never write through these pointers)
        else:
            self.g[-1].copy_(gout)               # a strided upstream gradient (not produced by the training loop)
            gp = self.g[-1].data_ptr()
        for op, field in patch["gout_readers"]:
            setattr(op, field, gp)
        if "tanh" in patch:
            patch["tanh"].y = self.out.data_ptr()
        g_in = None
        if need_input_grad:
            # the input gradient leaves the plan (autograd hands it on), so it gets its own tensor per call
            g_in = torch.empty(self.specs[0].in_dims(), dtype=torch.float32, device=self.device)
            patch["dgrad0"].out = g_in.data_ptr()
        self.g_in = g_in
        if not need_param_grad:
            prog.run(stream_ptr())
            self.busy = False
            return None, None, g_in
        if into is not None:
            per = {l: t for l, t in enumerate(into)}
            for l, w, force in patch["dw"]:
                w.dw = per[l][0].data_ptr()
                w.accumulate = 1 if (per[l][3] or force) else 0
            for l, b, force in patch["dgamma"]:
                b.dgamma = per[l][1].data_ptr()
                b.dbeta = per[l][2].data_ptr()
                b.accumulate = 1 if (per[l][3] or force) else 0
            self._patch_user_input(patch)
            self._run_backward(prog, patch)
            self.busy = False
            return None, None, (self.g_in if need_input_grad else None)
        flat = torch.empty(self.n_grad_floats(), dtype=torch.float32, device=self.device)
        views, off = [], 0
        per_layer = {}
        for l, p in enumerate(self.params):
            n = p.weight.numel()
            wv = flat[off:off + n].view_as(p.weight); off += n
            gv = bv = None
            if p.gamma is not None:
                c = p.gamma.numel()
                gv = flat[off:off + c]; off += c
                bv = flat[off:off + c]; off += c
            per_layer[l] = (wv, gv, bv)
            views.append((wv, gv, bv))
        for l, w, force in patch["dw"]:
            w.dw = per_layer[l][0].data_ptr()
            w.accumulate = 1 if force else 0
        for l, b, force in patch["dgamma"]:
            b.dgamma = per_layer[l][1].data_ptr()
            b.dbeta = per_layer[l][2].data_ptr()
            b.accumulate = 1 if force else 0
        self._patch_user_input(patch)
        prog.run(stream_ptr())
        self.busy = False
        return flat, views, (self.g_in if need_input_grad else None)
