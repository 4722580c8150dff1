"""nn.Module surface of the hot path, mirroring the reference's classes (same names, constructor signatures,
state_dict keys, RNG consumption and return conventions) with all device arithmetic lowered to libgode.so.

Reference: models/mocogan.py (Noise :20-29, PatchImageDiscriminator :66-93, VideoDiscriminator :129-164,
VideoGenerator :185-301) and models/mocogan_ode.py (ODEFunc :6-17, VideoGenerator :20-54, VideoGeneratorMNIST
:57-111, VideoGeneratorMNISTODE :114-148).

The child modules under ``main`` / ``ode_fn`` / ``linear`` / ``recurrent`` are stock torch.nn layers used ONLY as
parameter containers (default init, state_dict keys, .cuda()/.to()); they are never called.  Every forward goes
through the HIP kernels and raises if the tensors are not on the GPU or libgode.so is missing.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .engine import ConvStack, LayerParams, LayerSpec, conv_out, dptr, make_geom, stream_ptr

Z_COLS = 96  # latent row: [motion 16 | content 50 | zero pad 30]: K % 32 == 0 puts the first decoder GEMM on the FAST path


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: this implementation runs only on an MI355X through libgode.so; move the module "
                           "to the GPU with .cuda() (there is no CPU or PyTorch fallback)")


class _Pool:
    """Plans own their activation buffers; a plan stays checked out between a forward that saved state for autograd
    and its backward, so two overlapping passes of the same shape (D(real), D(fake)) get distinct plans.  A forward
    whose backward never comes (loss dropped) gives its plan back when its autograd node is freed (_Lease).  Every
    forward bumps the plan's generation and its autograd node remembers it: a backward that arrives after the plan's
    buffers were reused (retain_graph second pass, or more than MAX_PLANS forwards of one shape kept alive) raises
    instead of computing on overwritten activations."""

    MAX_PLANS = 6

    def __init__(self):
        self.plans = {}
        self.pack_cache = {}     # packed weight panels shared by every plan of the module

    def get(self, key, factory):
        lst = self.plans.setdefault(key, [])
        for p in lst:
            if not p.busy:
                return p
        if len(lst) >= self.MAX_PLANS:
            # all checked out: reuse the oldest; its pending backward (if it ever comes) fails the generation check
            p = lst.pop(0)
            p.busy = False
            lst.append(p)
            return p
        p = factory()
        lst.append(p)
        return p

    def clear(self):
        self.plans.clear()
        self.pack_cache.clear()


class _Lease:
    """Held by an autograd node: identifies the forward (plan generation) whose buffers the node will need, and gives
    the plan back if the node dies without having run its backward."""

    __slots__ = ("plan", "generation")

    def __init__(self, plan):
        self.plan, self.generation = plan, _plan_generation(plan)

    def check(self, what):
        if _plan_generation(self.plan) != self.generation:
            raise RuntimeError(
                f"{what}: the activation buffers of this forward pass were reused by a later forward of the same shape "
                "before backward() ran (a second backward through a retained graph, or more than "
                f"{_Pool.MAX_PLANS} grad-enabled forwards of one shape kept alive at once); run the forward again")

    def __del__(self):
        try:
            if self.plan.busy and _plan_generation(self.plan) == self.generation:
                self.plan.busy = False
        except Exception:
            pass


def _plan_generation(plan):
    return plan.stack.generation if hasattr(plan, "stack") else plan.generation


# ==================================================================================================================
# generator
# ==================================================================================================================
class ODEFunc(nn.Module):
    """Right-hand side f(t, x) = W2 tanh(W1 x + b1) + b2 (models/mocogan_ode.py:6-17).  Evaluated inside the fused
    ODE kernels; calling it directly is not part of the hot path."""

    def __init__(self, dim, dim_hidden):
        super().__init__()
        self.fn = nn.Sequential(nn.Linear(dim, dim_hidden), nn.Tanh(), nn.Linear(dim_hidden, dim))

    def forward(self, t, x):
        raise RuntimeError("ODEFunc is evaluated inside libgode's fused RK4 kernels (gode_ode_fwd/bwd)")


def solver_grid_arrays(T, step_size):
    """torchdiffeq options={'step_size': h} (FixedGridODESolver) for t = linspace(0, 1, T): the solver walks its own
    grid arange(niters)*h + t0 (last point clamped to t_end) and interpolates the requested outputs linearly; the
    adjoint pass solves every output interval on the grid built the same way for the reversed span.  All arithmetic
    is fp32 torch on the host, operation for operation what torchdiffeq executes, so the device gets bit-identical
    step sizes and interpolation weights.  Returns CPU tensors: grid_dt[G], emit_at[T] (grid step after which output
    j is produced), emit_w[T] (interpolation weight; 0 / 1 = an end point itself), bstep_off[T], bstep_dt[...]."""
    t = torch.linspace(0, 1, T).float()

    def grid_of(tt):
        start, end = tt[0], tt[-1]
        niters = int(torch.ceil((end - start) / step_size + 1).item())
        gr = torch.arange(0, niters, dtype=tt.dtype) * step_size + start
        gr[-1] = end
        return gr

    gr = grid_of(t)
    gdt = gr[1:] - gr[:-1]
    emit_at = torch.full((T,), -1, dtype=torch.int32)
    emit_w = torch.zeros(T, dtype=torch.float32)
    j = 1
    for i in range(len(gr) - 1):
        a, b = gr[i], gr[i + 1]
        while j < T and bool(b >= t[j]):
            emit_at[j] = i
            emit_w[j] = 0.0 if bool(t[j] == a) else (1.0 if bool(t[j] == b) else (t[j] - a) / (b - a))
            j += 1
    assert j == T
    off, steps = [0], []
    for i in range(1, T):                      # adjoint of output interval i -> i-1: reversed span, t -> -t
        rg = grid_of(-t[i - 1:i + 1].flip(0))
        d = rg[1:] - rg[:-1]
        steps.append(d)
        off.append(off[-1] + len(d))
    return dict(G=len(gdt), grid_dt=gdt, emit_at=emit_at, emit_w=emit_w,
                bstep_off=torch.tensor(off, dtype=torch.int32), bstep_dt=torch.cat(steps))


class _GenPlan:
    """ODE solve + decoder for `n_traj` trajectories.  full: rows = n_traj*T; select: rows = n_traj (one chosen
    time per trajectory, sample_images)."""

    def __init__(self, gen: "VideoGenerator", n_traj: int, T: int, select: bool, zbuf=None):
        """zbuf: latent rows of a joint pass (_GenJointPlan) -- the plan then is the latent half only (no decoder stack)."""
        dev = gen.main[0].weight.device
        self.gen, self.n, self.T, self.select, self.device = gen, n_traj, T, select, dev
        self.rows = n_traj if select else n_traj * T
        f32 = dict(dtype=torch.float32, device=dev)
        # host-drawn inputs live in ONE device buffer [x 16 | content 50 | sel (int32 bits) 1] per trajectory block, so a
        # call stages them with a single H2D copy
        xs = self._x_shape(n_traj, T)
        nx = int(np.prod(xs))
        self._in = torch.zeros(nx + n_traj * 51, **f32)
        self.x = self._in[:nx].view(xs)
        self.content = self._in[nx:nx + n_traj * 50].view(n_traj, 50)
        self.sel = self._in[nx + n_traj * 50:].view(torch.int32) if select else None
        self.traj = torch.empty(n_traj, T, 16, **f32)
        tt = torch.linspace(0, 1, T).float()            # models/mocogan_ode.py:143 -- fp32 grid built on the host
        self.dt = (tt[1:] - tt[:-1]).to(dev) if T > 1 else torch.zeros(1, **f32)
        if zbuf is None:
            self.stack = ConvStack(gen._decoder_specs(self.rows), gen._decoder_params(), dev, owns_input=True,
                                   pack_cache=gen._pool.pack_cache, two_lane_backward=True)
        else:
            self.stack, self._zbuf = None, zbuf
        self.ode_work = torch.empty(L.lib().gode_ode_bwd_work_size(n_traj), **f32)
        self._ode_ptrs = None
        self.busy = False
        # pinned staging ring for the host-drawn noise: a pageable H2D copy would block the host until the stream
        # drains (no host/GPU overlap between consecutive calls); slots are recycled behind an event
        self._ring = []
        for _ in range(4):
            buf = torch.zeros(nx + n_traj * 51).pin_memory()
            self._ring.append(dict(buf=buf, x=buf[:nx].view(xs), c=buf[nx:nx + n_traj * 50].view(n_traj, 50),
                                   s=buf[nx + n_traj * 50:].view(torch.int32), ev=None))
        self._ring_i = 0

    @staticmethod
    def _x_shape(n_traj, T):
        """Shape of the host-drawn noise block: the ODE's initial states."""
        return (n_traj, 16)


    def _ode_params(self):
        g = self.gen
        lin = g.linear
        if isinstance(lin, nn.Identity):
            pre = (None, None, None, None)
        else:
            pre = (lin[0].weight, lin[0].bias, lin[2].weight, lin[2].bias)
        f = g.ode_fn.fn
        return pre + (f[0].weight, f[0].bias, f[2].weight, f[2].bias)

    def _solver_grid(self, step_size):
        gd = solver_grid_arrays(self.T, step_size)
        return {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in gd.items()}

    def _programs(self):
        ps = self._ode_params()
        ptrs = tuple(dptr(p) for p in ps) + (self.gen.ode_step_size, self.gen.ode_method, self.gen.ode_rtol,
                                             self.gen.ode_atol)
        if ptrs != self._ode_ptrs:
            self._ode_ptrs = ptrs
            op = L.OdeParams(*ptrs[:8])
            prenet = 0 if ps[0] is None else 1
            zbuf = getattr(self, "_zbuf", None)
            zbuf = self.stack.x_in if zbuf is None else zbuf
            zcols = getattr(self, "_zcols", Z_COLS)
            self.fwd_op = L.OdeFwdOp(p=op, x=dptr(self.x), content=dptr(self.content), dt=dptr(self.dt),
                                     sel_t=dptr(self.sel), z=dptr(zbuf), traj=dptr(self.traj), N=self.n,
                                     T=self.T, substeps=self.gen.ode_substeps, prenet=prenet, zcols=zcols)
            self.bwd_op = L.OdeBwdOp(p=op, x=dptr(self.x), traj=dptr(self.traj), dt=dptr(self.dt),
                                     sel_t=dptr(self.sel), gz=None, work=dptr(self.ode_work), grads=None, N=self.n,
                                     T=self.T, substeps=self.gen.ode_substeps, prenet=prenet, accumulate=0, zcols=zcols)
            self._grid = None
            if self.gen.ode_method == "dopri5":
                # torchdiffeq's adaptive solver over the output times; the adjoint call is integrated adaptively too
                # (adjoint_substeps == 0, gode_ode_bwd method 1) or with `adjoint_substeps` fixed Kutta-3/8 steps
                self._tout = torch.linspace(0, 1, self.T).float().to(self.device)
                self._nsteps = torch.zeros((self.n + 63) // 64, dtype=torch.int32, device=self.device)
                # more than one workgroup (32 trajectories each): whole-batch error norm through these words
                ns = L.lib().gode_odernn_sync_size(self.n)
                self._sync_f = torch.zeros(ns, dtype=torch.int32, device=self.device) if ns else None
                self._sync_b = torch.zeros(ns, dtype=torch.int32, device=self.device) if ns else None
                self.fwd_op.sync, self.bwd_op.sync = dptr(self._sync_f), dptr(self._sync_b)
                self.fwd_op.method, self.fwd_op.rtol, self.fwd_op.atol = 1, float(self.gen.ode_rtol), float(self.gen.ode_atol)
                self.fwd_op.tout, self.fwd_op.nsteps = dptr(self._tout), dptr(self._nsteps)
                self.bwd_op.method, self.bwd_op.rtol, self.bwd_op.atol = 1, float(self.gen.ode_rtol), float(self.gen.ode_atol)
                self.bwd_op.tout = dptr(self._tout)
                self._nsteps_bwd = torch.zeros((self.n + 63) // 64, dtype=torch.int32, device=self.device)
                self.bwd_op.nsteps = dptr(self._nsteps_bwd)
            elif self.gen.ode_method != "rk4":
                raise NotImplementedError(f"ode_method {self.gen.ode_method!r}: libgode implements 'rk4' (the reference's "
                                          "call) and 'dopri5'")
            elif self.gen.ode_step_size is not None and self.T > 1:
                gd = self._grid = self._solver_grid(float(self.gen.ode_step_size))
                self.fwd_op.G = gd["G"]
                self.fwd_op.grid_dt, self.fwd_op.emit_at, self.fwd_op.emit_w = dptr(gd["grid_dt"]), dptr(gd["emit_at"]), dptr(gd["emit_w"])
                self.bwd_op.bstep_off, self.bwd_op.bstep_dt = dptr(gd["bstep_off"]), dptr(gd["bstep_dt"])
            self.fwd_prog = L.Program([self.fwd_op])
        self.fwd_op.substeps = self.bwd_op.substeps = self.gen.ode_substeps
        if self.gen.ode_method == "dopri5":
            self.bwd_op.substeps = self.gen.adjoint_substeps

    def stage_inputs(self, x_host, content_host, sel_host):
        """Host-drawn inputs -> the plan's device block, on the current stream (one H2D copy)."""
        self._programs()
        feed = getattr(self.gen, "_feed", None)
        if feed is not None and feed.recording:
            # graph capture (GanTrainer(graph=True)): the inputs travel in the trainer's HostFeed block; the recorded
            # copy is device -> device from the block's static region for this call site
            nfl = self._in.numel()
            pin, dev = feed.noise_region(self.gen, self.select, self.n, self.T, nfl)
            self._pack_inputs(pin, x_host, content_host, sel_host)
            self._in.copy_(dev)
        else:
            slot = self._ring[self._ring_i]
            self._ring_i = (self._ring_i + 1) % len(self._ring)
            if slot["ev"] is not None:
                slot["ev"].synchronize()
            self._pack_inputs(slot["buf"], x_host, content_host, sel_host)
            self._in.copy_(slot["buf"], non_blocking=True)
            if slot["ev"] is None:
                slot["ev"] = torch.cuda.Event()
            slot["ev"].record()

    def forward(self, x_host, content_host, sel_host, training, keep):
        """x_host None: the latent rows of this call were produced ahead of time by VideoGenerator.prefetch_latents (on
        its own stream); only the decoder runs here, after the latent's event."""
        if x_host is None:
            torch.cuda.current_stream().wait_event(self._latent_ev)
            out = self.stack.forward(training)
        else:
            self.stage_inputs(x_host, content_host, sel_host)
            out = self.stack.forward(training, pre_ops_program=self.fwd_prog)
        self.busy = keep
        return out

    def _pack_inputs(self, buf, x_host, content_host, sel_host):
        """[x | content 50 per trajectory | sel (int32 bits) 1 per trajectory] -- the layout of self._in."""
        nx, n = x_host.numel(), self.n
        buf[:nx].copy_(x_host.reshape(-1))
        buf[nx:nx + n * 50].copy_(content_host.reshape(-1))
        if self.select:
            buf[nx + n * 50:nx + n * 51].view(torch.int32).copy_(sel_host)

    def _decoder_into(self, arena):
        """Per decoder layer (weight, gamma, beta gradient targets, accumulate?) inside the trainer's arena: the first
        writer of an optimiser step stores, later ones add."""
        into = []
        for lp in self.gen._decoder_params():
            wt, acc = arena.target(lp.weight)
            gt = bt = None
            if lp.gamma is not None:
                gt, _ = arena.target(lp.gamma)
                bt, _ = arena.target(lp.beta)
            into.append((wt, gt, bt, acc))
        return into

    def _decoder_pass_done(self):
        """The trainer may ask to be told when the LAST decoder backward of a generator step has been queued (the decoder
        block of the gradient arena is then complete and its all-reduce can start under the latent adjoint)."""
        cb = getattr(self.gen, "_on_decoder_grads", None)
        if cb is not None:
            self.gen._decoder_passes_left -= 1
            if self.gen._decoder_passes_left == 0:
                cb()

    def backward(self, gout, arena=None):
        if arena is not None:
            _, _, gz = self.stack.backward(gout, need_input_grad=True, into=self._decoder_into(arena))
            self._decoder_pass_done()
            self.latent_backward(gz.data_ptr(), arena, gz)
            self.busy = False
            return None, None
        flat, views, gz = self.stack.backward(gout, need_input_grad=True)
        motion = self.latent_backward(gz.data_ptr(), None, gz)
        self.busy = False
        return views, motion

    def latent_backward(self, gz_ptr, arena, keep):
        """Adjoint of the latent half given the gradient wrt its latent rows (gz_ptr: first row, Z_COLS floats per row).
        arena: gradients go straight into the trainer's arena (first writer of a step stores, later ones add) and the
        launch may be deferred into the trainer's batch; else -> list of gradient tensors (pre-net + ODEFunc order)."""
        self.bwd_op.gz = gz_ptr
        if arena is not None:
            ode = [q for q in self._ode_params() if q is not None]
            tgt = [arena.target(q) for q in ode]
            base, acc = tgt[0]
            off = base.data_ptr() - (0 if self._ode_ptrs[0] is not None else 2128 * 4)   # kernel offsets start at Wa
            self.bwd_op.grads = off
            self.bwd_op.accumulate = 1 if acc else 0
            batch = getattr(self.gen, "_adjoint_batch", None)
            if batch is not None and self.gen.ode_method == "dopri5" and self.gen.adjoint_substeps == 0:
                # adaptive adjoints of the video and the image path: one launch (see _RnnGenPlan.latent_backward)
                batch["ops"].append(self.bwd_op)
                batch["keep"].append(keep)
                if len(batch["ops"]) >= batch["expect"]:
                    self.gen.flush_adjoints()
            else:
                L.run_one(self.bwd_op, stream_ptr())
            return None
        grads = torch.empty(L.ODE_NPARAM, dtype=torch.float32, device=self.device)
        self.bwd_op.grads = grads.data_ptr()
        self.bwd_op.accumulate = 0
        L.run_one(self.bwd_op, stream_ptr())
        offs = [(0, 1024, (64, 16)), (1024, 64, (64,)), (1088, 1024, (16, 64)), (2112, 16, (16,)),
                (2128, 256, (16, 16)), (2384, 16, (16,)), (2400, 256, (16, 16)), (2656, 16, (16,))]
        if self._ode_ptrs[0] is None:
            offs = offs[4:]
        return [grads[o:o + n].view(shp) for o, n, shp in offs]


class _GenJointPlan:
    """sample_videos(nv) and sample_images(ni) of one generator step decoded in ONE pass: latent rows [nv*T video rows | ni
    image rows], two BatchNorm batches (ConvStack(split_images=nv*T)): the reference calls `main` on the 512-row and on
    the 32-row batch separately (models/mocogan.py:276,293) -- same arithmetic per element, but the image path's GEMMs
    (M = 32 rows: 45-65 TFLOP/s on their own) ride along in the video path's launches and every elementwise / reduction
    launch is issued once.  `vid` / `img`: the latent halves (plans without a decoder) writing into the joint buffer.
    images_first: the reference called sample_images first (the discriminator steps), so its BatchNorm momentum update
    comes first."""

    def __init__(self, gen, nv, ni, T, images_first):
        dev = gen.main[0].weight.device
        self.gen, self.nv, self.ni, self.T, self.device = gen, nv, ni, T, dev
        self.rows_v, self.rows_i = nv * T, ni
        self.stack = ConvStack(gen._decoder_specs(self.rows_v + self.rows_i), gen._decoder_params(), dev, owns_input=True,
                               pack_cache=gen._pool.pack_cache, split_images=self.rows_v, split_order=1 if images_first else 0,
                               two_lane_backward=True)       # (UCF G step 4.43 -> 4.34 ms, MNIST neutral)
        self.vid = gen._plan_cls(gen, nv, T, False, zbuf=self.stack.x_in[:self.rows_v])
        self.img = gen._plan_cls(gen, ni, T, True, zbuf=self.stack.x_in[self.rows_v:])
        self.images_first = images_first
        self.busy = False
        self._latent_ev = None

    def subplans(self):
        """in the order of the reference's calls (= the order of the host draws)"""
        return (self.img, self.vid) if self.images_first else (self.vid, self.img)

    def forward(self, host, training, keep):
        """host: [(x, content, sel) per sub-plan in call order], or None when the latents were prefetched."""
        if host is None:
            torch.cuda.current_stream().wait_event(self._latent_ev)
        else:
            subs = self.subplans()
            for sub, h in zip(subs, host):
                sub.stage_inputs(*h)
            self.gen._launch_latents(list(subs))
        out = self.stack.forward(training)
        self.busy = keep
        return out

    def backward(self, g_joint, arena):
        subs = (self.vid, self.img)
        offs = (0, 4 * self.rows_v * Z_COLS)
        if arena is not None:
            _, _, gz = self.stack.backward(g_joint, need_input_grad=True, into=self.vid._decoder_into(arena))
            self.vid._decoder_pass_done()
            for sub, off in zip(subs, offs):
                sub.latent_backward(gz.data_ptr() + off, arena, gz)
            self.busy = False
            return None, None
        flat, views, gz = self.stack.backward(g_joint, need_input_grad=True)
        motion = None
        for sub, off in zip(subs, offs):
            m = sub.latent_backward(gz.data_ptr() + off, None, gz)
            motion = m if motion is None else [a + b for a, b in zip(motion, m)]
        self.busy = False
        return views, motion


class _GenJointFn(torch.autograd.Function):
    """The joint generator pass as one autograd node with two outputs (video frames, image frames)."""

    @staticmethod
    def forward(ctx, plan, host, training, keep, n_dec, *params):
        out = plan.forward(host, training, keep)
        ctx.plan, ctx.n_dec, ctx.n_params = plan, n_dec, len(params)
        ctx.lease = _Lease(plan)
        ctx.out_shape = tuple(out.shape)
        return out.narrow(0, 0, plan.rows_v), out.narrow(0, plan.rows_v, plan.rows_i)

    @staticmethod
    def backward(ctx, gv, gi):
        plan = ctx.plan
        ctx.lease.check(type(plan.gen).__name__ + " (joint pass)")
        g = torch.empty(ctx.out_shape, dtype=torch.float32, device=plan.device)
        for part, gg in ((g.narrow(0, 0, plan.rows_v), gv), (g.narrow(0, plan.rows_v, plan.rows_i), gi)):
            if gg is None:
                part.zero_()
            else:
                part.copy_(gg)
        arena = getattr(plan.gen, "_gode_arena", None)
        if arena is not None and arena.active:
            plan.backward(g, arena)
            return (None,) * (5 + ctx.n_params)
        views, motion = plan.backward(g, None)
        grads = []
        for wv, gvw, bv in views:
            grads.append(wv)
            if gvw is not None:
                grads += [gvw, bv]
        assert len(grads) == ctx.n_dec and len(motion) == ctx.n_params - ctx.n_dec
        return (None, None, None, None, None, *grads, *motion)


class _GenFn(torch.autograd.Function):
    """Whole generator pass (pre-net + RK4 + decoder) as one autograd node; backward = decoder backward + adjoint."""

    @staticmethod
    def forward(ctx, plan, x_host, content_host, sel_host, training, keep, n_dec, *params):
        out = plan.forward(x_host, content_host, sel_host, training, keep)
        ctx.plan, ctx.n_dec, ctx.n_params = plan, n_dec, len(params)
        ctx.lease = _Lease(plan)
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        ctx.lease.check(type(plan.gen).__name__)
        arena = getattr(plan.gen, "_gode_arena", None)
        if arena is not None and arena.active:
            plan.backward(gout, arena)
            return (None,) * (7 + ctx.n_params)
        views, motion = plan.backward(gout)
        grads = []
        for wv, gv, bv in views:
            grads.append(wv)
            if gv is not None:
                grads += [gv, bv]
        assert len(grads) == ctx.n_dec and len(motion) == ctx.n_params - ctx.n_dec
        return (None, None, None, None, None, None, None, *grads, *motion)


class _LatentFn(torch.autograd.Function):
    """sample_z_m as its own autograd node: the fused ODE kernel alone, trajectory written as [N][T][16] = the
    reference's transpose(0,1).reshape(-1,16) row order."""

    @staticmethod
    def forward(ctx, gen, x_host, n, T, *params):
        dev = gen.main[0].weight.device
        lp = _GenPlan.__new__(_GenPlan)           # only the ODE half of a plan: no decoder buffers
        lp.gen, lp.n, lp.T, lp.select, lp.device = gen, n, T, False, dev
        f32 = dict(dtype=torch.float32, device=dev)
        lp.x = x_host.to(dev)
        lp.content, lp.sel = None, None
        lp.traj = torch.empty(n, T, 16, **f32)
        tt = torch.linspace(0, 1, T).float()
        lp.dt = (tt[1:] - tt[:-1]).to(dev) if T > 1 else torch.zeros(1, **f32)
        lp._z = torch.empty(n * T, 68, **f32)     # the kernel's latent-row output (unused columns stay unwritten)
        lp.ode_work = torch.empty(L.lib().gode_ode_bwd_work_size(n), **f32)
        lp._ode_ptrs = None
        lp._zbuf, lp._zcols = lp._z, 68
        lp._programs()
        lp.fwd_prog.run(stream_ptr())
        ctx.lp = lp
        return lp.traj.view(n * T, 16)

    @staticmethod
    def backward(ctx, gout):
        lp = ctx.lp
        g = gout.contiguous()
        grads = torch.empty(L.ODE_NPARAM, dtype=torch.float32, device=lp.device)
        op = lp.bwd_op
        op.gz, op.zcols, op.grads, op.accumulate = g.data_ptr(), 16, grads.data_ptr(), 0
        L.run_one(op, stream_ptr())
        offs = [(0, 1024, (64, 16)), (1024, 64, (64,)), (1088, 1024, (16, 64)), (2112, 16, (16,)),
                (2128, 256, (16, 16)), (2384, 16, (16,)), (2400, 256, (16, 16)), (2656, 16, (16,))]
        if lp._ode_ptrs[0] is None:
            offs = offs[4:]
        return (None, None, None, None, *[grads[o:o + k].view(shp) for o, k, shp in offs])


class VideoGenerator(nn.Module):
    """MoCoGAN generator whose motion latent is a Neural ODE (models/mocogan_ode.py:20-54), 64x64 decoder
    (models/mocogan.py:200-215).  `ode_substeps` (build extension, default 1 = the reference's one RK4 step per
    output interval) sub-divides each interval."""

    mnist = False
    _plan_cls = None   # set below (_GenPlan)
    # the reference passes method='rk4' (models/mocogan_ode.py:50,144); 'dopri5' is what BASELINE configs[3] words
    ode_method = "rk4"
    # adjoint_substeps (dopri5 only): 0 = the adjoint is integrated adaptively with the same controller and
    # torchdiffeq's mixed norm, as odeint_adjoint does; k > 0 = k fixed Kutta-3/8 steps per output interval
    ode_rtol, ode_atol, adjoint_substeps = 1e-7, 1e-9, 0

    def __init__(self, n_channels, dim_z_content, dim_z_category, dim_z_motion, video_length, ode_fn=ODEFunc,
                 dim_hidden=None, linear=True, ngf=64):
        super().__init__()
        if dim_z_category != 0:
            raise NotImplementedError("categorical latents are never used by the stage-3 scripts (dim_z_category=0)")
        if dim_z_motion != 16 or dim_z_content != 50:
            raise NotImplementedError("libgode's fused ODE kernels are specialised for dim_z_motion=16, "
                                      "dim_z_content=50 (mnist_moco_ode.py:78, ucf_moco_ode.py:80)")
        self.n_channels, self.dim_z_content, self.dim_z_category = n_channels, dim_z_content, dim_z_category
        self.dim_z_motion, self.video_length, self.ngf = dim_z_motion, video_length, ngf
        self.ode_substeps = 1
        self.ode_step_size = None    # = torchdiffeq options={'step_size': h}; None: the reference's call (grid = outputs)
        dim_z = dim_z_motion + dim_z_category + dim_z_content
        # construction order == the reference's, so that a given torch seed yields the same initial weights
        self.recurrent = nn.GRUCell(dim_z_motion, dim_z_motion)       # models/mocogan.py:198 (unused by ODE path)
        self.main = self._make_main(dim_z, ngf, n_channels, mnist=False)  # base class always builds the 64x64 stack
        self._init_ode_parts(ode_fn, dim_hidden, linear, dim_z, ngf)
        self._pool = _Pool()

    def _init_ode_parts(self, ode_fn, dim_hidden, linear, dim_z, ngf):
        if dim_hidden is None:
            # the reference passes no dim_hidden here and its ODEFunc then raises TypeError;
            # keep the failure mode but say why.
            raise TypeError("ODEFunc.__init__() missing 1 required positional argument: 'dim_hidden' "
                            "(pass dim_hidden=16; ucf_moco_ode.py:80 omits it and fails the same way)")
        self.ode_fn = self._make_ode_fn(ode_fn, dim_hidden)
        self.linear = self._make_prenet(linear)

    def _make_ode_fn(self, ode_fn, dim_hidden):
        """The fused RK4 / adjoint kernels hard-code ODEFunc's 16 -> 16 -> 16 tanh MLP (one MFMA tile per mat-vec):
        anything else would be integrated with the wrong right-hand side, so it is refused here."""
        hid = dim_hidden if dim_hidden else self.dim_z_motion
        if hid != 16:
            raise NotImplementedError(f"dim_hidden={hid}: libgode's fused ODE kernels are specialised for the "
                                      "16-16-16 ODEFunc of mnist_moco_ode.py:78 (dim_hidden=None -> dim_z_motion=16)")
        if not (isinstance(ode_fn, type) and issubclass(ode_fn, ODEFunc)):
            raise NotImplementedError("ode_fn must be gan_ode_amd's ODEFunc (W2 tanh(W1 x + b1) + b2, "
                                      "models/mocogan_ode.py:6-17): the right-hand side is evaluated inside the "
                                      "fused HIP kernels, an arbitrary Python callable cannot be")
        fn = ode_fn(dim=self.dim_z_motion, dim_hidden=hid)
        lin = [m for m in fn.fn] if hasattr(fn, "fn") else []
        if not (len(lin) == 3 and isinstance(lin[0], nn.Linear) and isinstance(lin[1], nn.Tanh)
                and isinstance(lin[2], nn.Linear) and tuple(lin[0].weight.shape) == (16, 16)
                and tuple(lin[2].weight.shape) == (16, 16)):
            raise NotImplementedError("ode_fn.fn must be Sequential(Linear(16,16), Tanh(), Linear(16,16))")
        return fn

    def _make_prenet(self, linear):
        if not linear:
            return nn.Identity()
        d = self.dim_z_motion
        return nn.Sequential(nn.Linear(d, 64), nn.LeakyReLU(0.2), nn.Linear(64, d), nn.LeakyReLU(0.2))

    @staticmethod
    def _make_main(dim_z, ngf, n_channels, mnist):
        w = [dim_z, ngf * 8, ngf * 4, ngf * 2, ngf]
        layers = []
        for i in range(4):
            st, pd = (1, 0) if i == 0 else (2, 1)
            layers += [nn.ConvTranspose2d(w[i], w[i + 1], 4, st, pd, bias=False), nn.BatchNorm2d(w[i + 1]), nn.ReLU(True)]
        if mnist:
            layers.append(nn.ConvTranspose2d(ngf, n_channels, kernel_size=1, stride=1, padding=2, bias=False))
        else:
            layers.append(nn.ConvTranspose2d(ngf, n_channels, 4, 2, 1, bias=False))
        layers.append(nn.Tanh())
        return nn.Sequential(*layers)

    # -- plan description ----------------------------------------------------------------------------------------
    def _decoder_specs(self, rows):
        ngf, nc = self.ngf, self.n_channels
        dev = self.main[0].weight.device
        perm = torch.tensor([50 + i for i in range(16)] + list(range(50)) + [-1] * (Z_COLS - 66), dtype=torch.int32,
                            device=dev)
        one = (1, 1, 1)
        specs = [LayerSpec(make_geom(rows, ngf * 8, Z_COLS, (1, 4, 4), one, (1, 4, 4), one, (0, 0, 0)), L.DGRAD,
                           L.ACT_RELU, True, co_perm=perm)]
        hw = 4
        for ci, co in ((ngf * 4, ngf * 8), (ngf * 2, ngf * 4), (ngf, ngf * 2)):
            specs.append(LayerSpec(make_geom(rows, ci, co, (1, hw * 2, hw * 2), (1, hw, hw), (1, 4, 4), (1, 2, 2),
                                             (0, 1, 1)), L.DGRAD, L.ACT_RELU, True))
            hw *= 2
        if self.mnist:   # ConvTranspose2d(ngf, C, 1, 1, padding=2): a 1x1 conv on the centre 28x28 crop
            specs.append(LayerSpec(make_geom(rows, nc, ngf, (1, 28, 28), (1, 32, 32), one, one, (0, 2, 2)), L.DGRAD,
                                   L.ACT_NONE, False, epilogue=L.EPI_TANH))
        else:
            specs.append(LayerSpec(make_geom(rows, nc, ngf, (1, 64, 64), (1, 32, 32), (1, 4, 4), (1, 2, 2), (0, 1, 1)),
                                   L.DGRAD, L.ACT_NONE, False, epilogue=L.EPI_TANH))
        return specs

    def _decoder_params(self):
        m = self.main
        out = []
        for i in range(4):
            conv, bn = m[3 * i], m[3 * i + 1]
            out.append(LayerParams(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked))
        out.append(LayerParams(m[12].weight))
        return out

    def _param_list(self):
        dec = []
        for p in self._decoder_params():
            dec.append(p.weight)
            if p.gamma is not None:
                dec += [p.gamma, p.beta]
        ode = []
        if not isinstance(self.linear, nn.Identity):
            ode += [self.linear[0].weight, self.linear[0].bias, self.linear[2].weight, self.linear[2].bias]
        f = self.ode_fn.fn
        ode += [f[0].weight, f[0].bias, f[2].weight, f[2].bias]
        return dec, ode

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if hasattr(self, "_pool"):
            self._pool.clear()   # parameter storage may have moved
            self.__dict__.pop("_labels", None)
            self.__dict__.pop("_latent_plans", None)
            self.__dict__.pop("_prefetched", None)
        return r

    def invalidate_packs(self):
        """Call after writing conv weights behind autograd's back (`p.data.copy_()`, `p.data.clamp_()`,
        `dist.broadcast(p.data)`): such writes bump neither torch's version counter nor FusedAdam's, so the packed
        weight panels would stay stale.  In-place ops on the parameter itself, optimiser steps, load_state_dict and
        .to()/.cuda() are tracked automatically."""
        for k in [k for k in self._pool.pack_cache if k[0] == "ver"]:
            del self._pool.pack_cache[k]

    def _zero_labels(self, n, device):
        """np.zeros(B) -> torch.from_numpy -> .cuda() of the reference (models/mocogan.py:235,279): float64 zeros,
        never written by anyone, so one cached tensor per batch size serves every call (no fill kernel per call)."""
        key = (n, device)
        t = self.__dict__.setdefault("_labels", {}).get(key)
        if t is None:
            t = self._labels[key] = torch.zeros(n, dtype=torch.float64, device=device)
        return t

    _gode_direct_grads = True      # the backward kernels can write into a trainer-owned GradArena

    def _arena_tail(self):
        """Pre-net + ODEFunc tensors in the adjoint kernel's output order (they must be contiguous in the arena)."""
        return self._param_list()[1]

    # -- host-side latent draws: RNG call order is part of the contract ---------------------------------------
    def _draw(self, num_samples, video_len):
        """NumPy normal for the content code first (models/mocogan.py:252), then torch.randn on the CPU generator
        for the ODE's initial noise (models/mocogan_ode.py:136)."""
        content = np.random.normal(0, 1, (num_samples, self.dim_z_content)).astype(np.float32)
        x = torch.randn(num_samples, self.dim_z_motion)
        return torch.from_numpy(content), x

    def _run(self, n_traj, T, select, x, content, sel, plan=None):
        _require_gpu(self.main[0].weight, type(self).__name__)
        if plan is None:
            plan = self._pool.get((n_traj, T, select), lambda: self._plan_cls(self, n_traj, T, select))
        dec, ode = self._param_list()
        # (grad mode is off inside Function.forward, so decide here whether the plan must be kept for a backward)
        keep = torch.is_grad_enabled() and any(p.requires_grad for p in dec + ode)
        return _GenFn.apply(plan, x, content, sel, self.training, keep, len(dec), *dec, *ode)

    # -- latents ahead of time ----------------------------------------------------------------------------------------
    def prefetch_latents(self, calls):
        """calls: [("videos" | "images", num_samples), ...] -- the sample_videos / sample_images calls that will follow,
        in order, with no optimiser step on THIS network in between (one training iteration: the generator's weights
        change only at its end).  All host draws are made now, in call order (the same NumPy / torch CPU generator
        consumption as the calls themselves would make), and the latent solves of all calls are issued at once on a side
        stream; the calls then only wait for the latents and run the decoder.  For the ODE-RNN generator the solves --
        one workgroup of a 256-CU GPU each, ~100 us of dependent arithmetic -- become ONE launch whose workgroups run side
        by side (gode_odernn_fwd_multi) instead of six launches back to back.  Results are bit-identical to the calls
        without prefetch."""
        _require_gpu(self.main[0].weight, type(self).__name__)
        if self.__dict__.get("_prefetched"):
            raise RuntimeError("prefetch_latents: the previous prefetch has not been consumed "
                               f"({len(self._prefetched)} calls left)")
        for kind, n in calls:
            if kind not in ("videos", "images", "pair_vi", "pair_iv"):
                raise ValueError(f"prefetch_latents: unknown call kind {kind!r}")
        main = torch.cuda.current_stream()
        lat = self.__dict__.get("_latent_stream")
        if lat is None or lat.device != main.device:
            lat = self._latent_stream = torch.cuda.Stream(device=main.device)
        plans, queue = [], []
        T = self.video_length
        per_shape = {}
        for kind, n in calls:
            per_shape[(kind, n)] = per_shape.get((kind, n), 0) + 1
        if max(per_shape.values(), default=0) > _Pool.MAX_PLANS:
            raise RuntimeError(f"prefetch_latents: more than {_Pool.MAX_PLANS} calls of one shape ahead (each holds its own "
                               "plan and buffers); announce fewer calls at a time")
        for kind, n in calls:                  # (plans are created on the caller's stream, like every other plan)
            if kind.startswith("pair"):        # n = (n_videos, n_images): sample_pair -- one joint plan, two latent halves
                plan = self._joint_plan(n[0], n[1], T, kind == "pair_iv")
                if plan is None:
                    raise RuntimeError(f"prefetch_latents: {kind}{n} cannot be decoded jointly; announce the two calls")
                plan.busy = True
                plans.append(plan)
                queue.append((kind, n, T, plan))
                continue
            select = kind == "images"
            plan = self._pool.get((n, T, select), lambda: self._plan_cls(self, n, T, select))
            plan.busy = True                   # reserved: a second call of the same shape gets its own plan and buffers
            plans.append(plan)
            queue.append((select, n, T, plan))
        lat.wait_stream(main)                  # weights written on the caller's stream (Adam) are complete
        with torch.cuda.stream(lat):
            latent_plans = []
            for plan, (select, n, _, _) in zip(plans, queue):
                # draw and stage call by call: a generator may hand out a reused host buffer (the ODE-RNN noise stack)
                for sub in (plan.subplans() if isinstance(plan, _GenJointPlan) else (plan,)):
                    sub.stage_inputs(*self._host_inputs(sub.select, sub.n, T))
                    latent_plans.append(sub)
            self._launch_latents(latent_plans)
            ev = torch.cuda.Event()
            ev.record(lat)
        for p in plans:
            p._latent_ev = ev
        self._prefetched = queue

    def _launch_latents(self, plans):
        st = stream_ptr()
        if self.ode_method == "dopri5":     # adaptive solves, one workgroup each at the config sizes: one launch per 8
            for k in range(0, len(plans), 8):
                chunk = plans[k:k + 8]
                arr = (L.OdeFwdOp * len(chunk))(*[p.fwd_op for p in chunk])
                L.call("ode_fwd_multi", lambda: L.check(L.lib().gode_ode_fwd_multi(arr, len(chunk), st), "gode_ode_fwd_multi"))
            return
        for p in plans:
            p.fwd_prog.run(st)

    def flush_adjoints(self):
        """Launch the adaptive adjoints collected during a backward pass (GanTrainer.g_step: video + image path)."""
        batch = getattr(self, "_adjoint_batch", None)
        if not batch or not batch["ops"]:
            return
        ops = batch["ops"]
        st = stream_ptr()
        for k in range(0, len(ops), 8):
            chunk = ops[k:k + 8]
            arr = (L.OdeBwdOp * len(chunk))(*chunk)
            L.call("ode_bwd_multi", lambda: L.check(L.lib().gode_ode_bwd_multi(arr, len(chunk), st), "gode_ode_bwd_multi"))
        batch["ops"], batch["keep"] = [], []

    def discard_prefetched(self):
        """Drop latents that were prefetched but will not be consumed (an exception in the middle of an iteration)."""
        for _, _, _, plan in self.__dict__.get("_prefetched") or []:
            plan.busy = False
        self._prefetched = []

    def _take_prefetched(self, select, n, T):
        q = self.__dict__.get("_prefetched")
        if not q:
            return None
        s0, n0, T0, plan = q[0]
        if (s0, n0, T0) != (select, n, T):
            name = lambda k: k if isinstance(k, str) else ("sample_images" if k else "sample_videos")      # noqa: E731
            raise RuntimeError(f"prefetch_latents promised {name(s0)}({n0}) next, got {name(select)}({n}, video_len={T})")
        q.pop(0)
        return plan

    # -- reference API ---------------------------------------------------------------------------------------------
    def _host_inputs(self, select, num_samples, T):
        """The host-drawn inputs of one sample_videos (select False) / sample_images (select True) call, consuming the
        NumPy and torch CPU generators exactly as the reference does: -> (x [n_traj, 16], content [n_traj, 50],
        sel int32 [n_traj] or None).  sample_images: the reference integrates B*T*2 trajectories and decodes B randomly
        chosen rows of the B*T*2*T latent rows (models/mocogan.py:287-295); all draws are made, only the chosen
        trajectories are kept."""
        if not select:
            content, x = self._draw(num_samples, T)
            return x, content, None
        n_all = num_samples * T * 2
        content, x = self._draw(n_all, T)
        j = np.sort(np.random.choice(n_all * T, num_samples, replace=False)).astype(np.int64)
        traj = torch.from_numpy(j // T)
        sel = torch.from_numpy((j % T).astype(np.int32))
        return self._take_trajectories(x, traj), content[traj].contiguous(), sel

    @staticmethod
    def _take_trajectories(x, traj):
        return x[traj].contiguous()

    def _fill_host_inputs(self, buf, select, num_samples, T):
        """Graph replay (HostFeed): the same draws, written straight into the feed block's region for this call site."""
        x, content, sel = self._host_inputs(select, num_samples, T)
        nx, n = x.numel(), content.shape[0]
        buf[:nx].copy_(x.reshape(-1))
        buf[nx:nx + n * 50].copy_(content.reshape(-1))
        if select:
            buf[nx + n * 50:nx + n * 51].view(torch.int32).copy_(sel)

    def sample_videos(self, num_samples, video_len=None):
        """-> (videos [B, C, T, H, W] fp32, float64 zero labels [B]); models/mocogan.py:271-285."""
        T = video_len if video_len is not None else self.video_length
        plan = self._take_prefetched(False, num_samples, T)
        if plan is not None:
            h = self._run(num_samples, T, False, None, None, None, plan=plan)
        else:
            x, content, _ = self._host_inputs(False, num_samples, T)
            h = self._run(num_samples, T, False, x, content, None)        # [B*T, 1, H, W, C]
        H, W = h.size(2), h.size(3)
        h = h.view(num_samples, T, H, W, self.n_channels).permute(0, 4, 1, 2, 3)
        return h, self._zero_labels(num_samples, h.device)

    def sample_images(self, num_samples):
        """-> (images [B, C, H, W], None); models/mocogan.py:287-295.  The reference integrates B*T*2 trajectories
        and decodes B randomly chosen rows of the B*T*2*T latent rows; the draws are reproduced exactly on the host
        and only the chosen trajectories are integrated.  For the fixed-grid rk4 call of the reference this is exact
        (trajectories are independent and unselected rows receive no gradient).  With ode_method="dopri5" it is a
        documented deviation: torchdiffeq accepts/rejects steps on an error norm over ALL B*T*2 trajectories, here
        the norm is taken over the selected ones, so the accepted step sequence differs (both solutions are within
        rtol 1e-7 of the exact flow; tests/test_gpu_modules.py::test_dopri5_method_against_oracle pins the size)."""
        T = self.video_length
        plan = self._take_prefetched(True, num_samples, T)
        if plan is not None:
            h = self._run(num_samples, T, True, None, None, None, plan=plan)
        else:
            x, content, sel = self._host_inputs(True, num_samples, T)
            h = self._run(num_samples, T, True, x, content, sel)
        return h.view(num_samples, h.size(2), h.size(3), self.n_channels).permute(0, 3, 1, 2), None

    def sample_pair(self, n_videos, n_images, images_first=False):
        """(sample_videos(n_videos), sample_images(n_images)) -- or, images_first, the same two calls in the other order --
        with the host draws of those two calls in that order, decoded in ONE pass over [video rows | image rows] with
        per-call BatchNorm batch statistics (_GenJointPlan): -> ((videos, labels), (images, None)).  Falls back to the two
        calls when the layers' statistics rows do not split at the batch boundary (unusual batch sizes)."""
        T = self.video_length
        _require_gpu(self.main[0].weight, type(self).__name__)
        plan = self._take_prefetched("pair_iv" if images_first else "pair_vi", (n_videos, n_images), T)
        host = None
        if plan is None:
            plan = self._joint_plan(n_videos, n_images, T, images_first)
            if plan is None:
                if images_first:
                    im = self.sample_images(n_images)
                    return self.sample_videos(n_videos), im
                vd = self.sample_videos(n_videos)
                return vd, self.sample_images(n_images)
            host = [self._host_inputs(sub.select, sub.n, T) for sub in plan.subplans()]
        dec, ode = self._param_list()
        keep = torch.is_grad_enabled() and any(p.requires_grad for p in dec + ode)
        hv, hi = _GenJointFn.apply(plan, host, self.training, keep, len(dec), *dec, *ode)
        H, W = hv.size(2), hv.size(3)
        vid = hv.view(n_videos, T, H, W, self.n_channels).permute(0, 4, 1, 2, 3)
        img = hi.view(n_images, H, W, self.n_channels).permute(0, 3, 1, 2)
        return (vid, self._zero_labels(n_videos, vid.device)), (img, None)

    def can_pair(self, n_videos, n_images, images_first=False):
        """Can sample_pair decode these two batches in one pass?  (Builds the plan on first use.)"""
        key = ("pair", n_videos, n_images, self.video_length, images_first)
        if key in self.__dict__.get("_no_joint", ()):
            return False
        return bool(self._pool.plans.get(key)) or self._joint_plan(n_videos, n_images, self.video_length, images_first) is not None

    def _joint_plan(self, nv, ni, T, images_first):
        """A free joint plan for this shape, or None when the shape cannot be decoded jointly."""
        key = ("pair", nv, ni, T, images_first)
        bad = self.__dict__.setdefault("_no_joint", set())
        if key in bad:
            return None
        try:
            return self._pool.get(key, lambda: _GenJointPlan(self, nv, ni, T, images_first))
        except ValueError:
            bad.add(key)
            return None

    def sample_z_content(self, num_samples, video_len=None):
        """models/mocogan.py:249-257 -> [N*T, 50], the same code on all T rows of a video."""
        T = video_len if video_len is not None else self.video_length
        c = np.repeat(np.random.normal(0, 1, (num_samples, self.dim_z_content)).astype(np.float32), T, axis=0)
        return torch.from_numpy(c).to(self.main[0].weight.device)

    def sample_z_categ(self, num_samples, video_len=None):
        """models/mocogan.py:231-247 with dim_z_category == 0 (the only case stage 3 uses): (None, np.zeros(N))."""
        return None, np.zeros(num_samples)

    def sample_z_m(self, num_samples, video_len=None):
        """models/mocogan_ode.py:133-148 (UCF class :39-54): torch.randn(N, 16) on the CPU generator -> pre-net ->
        odeint(rk4 on linspace(0, 1, T)) -> transpose(0, 1).reshape(-1, 16): [N*T, 16], row n*T + t.  One launch of the
        fused kernel (gode_ode_fwd); differentiable (gode_ode_bwd).  sample_videos()/sample_images() do NOT go
        through here -- they fuse the solve with the content broadcast and the decoder -- this method exists for
        the reference's public surface."""
        T = video_len if video_len is not None else self.video_length
        _require_gpu(self.main[0].weight, type(self).__name__)
        x = self._draw_motion(num_samples, T)
        _, ode = self._param_list()
        return _LatentFn.apply(self, x, num_samples, T, *ode)

    def _draw_motion(self, num_samples, T):
        return torch.randn(num_samples, self.dim_z_motion)

    def sample_z_video(self, num_samples, video_len=None):
        """models/mocogan.py:259-269: (cat([content, motion], dim=1) [N*T, 66], np.zeros(N)); RNG order: NumPy normal
        for the content first, then torch.randn for the motion noise."""
        z_content = self.sample_z_content(num_samples, video_len)
        z_category, z_category_labels = self.sample_z_categ(num_samples, video_len)
        z_motion = self.sample_z_m(num_samples, video_len)
        return torch.cat([z_content, z_motion], dim=1), z_category_labels

    def forward(self, *a, **k):
        raise RuntimeError("use sample_videos()/sample_images() (the reference never calls forward())")


VideoGenerator._plan_cls = _GenPlan


class VideoGeneratorMNIST(VideoGenerator):
    """28x28 decoder variant (models/mocogan_ode.py:57-111)."""

    mnist = True

    def _init_ode_parts(self, ode_fn, dim_hidden, linear, dim_z, ngf):
        self.ode_fn = self._make_ode_fn(ode_fn, dim_hidden)
        self.main = self._make_main(dim_z, ngf, self.n_channels, mnist=True)
        self.linear = self._make_prenet(linear)


class VideoGeneratorMNISTODE(VideoGeneratorMNIST):
    """The class mnist_moco_ode.py:6 imports (models/mocogan_ode.py:114-148)."""

    def _init_ode_parts(self, ode_fn, dim_hidden, linear, dim_z, ngf):
        super()._init_ode_parts(ode_fn, dim_hidden, linear, dim_z, ngf)
        self.ode_fn = self._make_ode_fn(ode_fn, dim_hidden)   # re-created, as the reference does
        self.linear = self._make_prenet(linear)


class _RnnGenPlan(_GenPlan):
    """ODE-RNN latent (gode_odernn_fwd/bwd) + decoder.  x_host carries the noise stack [T+1, n, 16]."""

    def __init__(self, gen, n_traj, T, select, zbuf=None):
        super().__init__(gen, n_traj, T, select, zbuf=zbuf)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.noise = self.x        # [T+1, n, 16]: h_0 and the per-frame GRU inputs (staged by the base class)
        self.hp = torch.empty(n_traj, T, 16, **f32)
        self.nsteps = torch.zeros(T, dtype=torch.int32, device=self.device)          # dopri5 trial steps per frame
        self.nsteps_bwd = torch.zeros(T, dtype=torch.int32, device=self.device)      # ... of each frame's adjoint call
        self.rnn_work = torch.empty(L.lib().gode_odernn_bwd_work_size(n_traj), **f32)
        # more than one workgroup (32 trajectories each): the whole-batch error norm is exchanged through these words
        ns = L.lib().gode_odernn_sync_size(n_traj)
        self.sync_f = torch.zeros(ns, dtype=torch.int32, device=self.device) if ns else None
        self.sync_b = torch.zeros(ns, dtype=torch.int32, device=self.device) if ns else None

    @staticmethod
    def _x_shape(n_traj, T):
        return (T + 1, n_traj, 16)

    def _rnn_params(self):
        g = self.gen
        f, r = g.ode_fn.fn, g.recurrent
        return (f[0].weight, f[0].bias, f[2].weight, f[2].bias, r.weight_ih, r.weight_hh, r.bias_ih, r.bias_hh)

    def _programs(self):
        ptrs = tuple(dptr(p) for p in self._rnn_params())
        if ptrs != self._ode_ptrs:
            self._ode_ptrs = ptrs
            op = L.OdeRnnParams(*ptrs)
            self.fwd_op = L.OdeRnnFwdOp(p=op, noise=dptr(self.noise), content=dptr(self.content), sel_t=dptr(self.sel),
                                        z=dptr(self.stack.x_in if self.stack is not None else self._zbuf), hs=None,
                                        hp=dptr(self.hp), nsteps=dptr(self.nsteps),
                                        N=self.n, T=self.T, rtol=self.gen.ode_rtol, atol=self.gen.ode_atol, zcols=Z_COLS,
                                        sync=dptr(self.sync_f))
            self.bwd_op = L.OdeRnnBwdOp(p=op, noise=dptr(self.noise), hp=dptr(self.hp), sel_t=dptr(self.sel), gz=None,
                                        work=dptr(self.rnn_work), grads=None, N=self.n, T=self.T,
                                        substeps=self.gen.adjoint_substeps, accumulate=0, zcols=Z_COLS,
                                        rtol=self.gen.ode_rtol, atol=self.gen.ode_atol, sync=dptr(self.sync_b),
                                        nsteps=dptr(self.nsteps_bwd))
            self.fwd_prog = L.Program([self.fwd_op])
        self.fwd_op.rtol, self.fwd_op.atol = self.gen.ode_rtol, self.gen.ode_atol
        self.bwd_op.rtol, self.bwd_op.atol = self.gen.ode_rtol, self.gen.ode_atol
        self.bwd_op.substeps = self.gen.adjoint_substeps

    def latent_backward(self, gz_ptr, arena, keep):
        self.bwd_op.gz = gz_ptr
        if arena is not None:
            # the 2176 ODEFunc + GRU gradients go straight into the trainer's arena: the eight tensors are its tail,
            # contiguous in the kernel's output order (VideoGeneratorMNISTODERNN._arena_tail)
            tgt = [arena.target(q) for q in self._rnn_params()]
            base, acc = tgt[0]
            self.bwd_op.grads = base.data_ptr()
            self.bwd_op.accumulate = 1 if acc else 0
            batch = getattr(self.gen, "_adjoint_batch", None)
            if batch is not None:
                # the trainer announced how many generator passes this backward holds (video + image path of the G step):
                # their adjoints -- one workgroup each -- go out as ONE launch once the last decoder backward is through
                batch["ops"].append(self.bwd_op)
                batch["keep"].append(keep)
                if len(batch["ops"]) >= batch["expect"]:
                    self.gen.flush_adjoints()
            else:
                L.run_one(self.bwd_op, stream_ptr())
            return None
        grads = torch.empty(L.ODERNN_NPARAM, dtype=torch.float32, device=self.device)
        self.bwd_op.grads = grads.data_ptr()
        self.bwd_op.accumulate = 0
        L.run_one(self.bwd_op, stream_ptr())
        return [grads[o:o + n].view(shp) for o, n, shp in _RNN_GRAD_OFFS]


_RNN_GRAD_OFFS = [(0, 256, (16, 16)), (256, 16, (16,)), (272, 256, (16, 16)), (528, 16, (16,)),
                  (544, 768, (48, 16)), (1312, 768, (48, 16)), (2080, 48, (48,)), (2128, 48, (48,))]


class _RnnLatentFn(torch.autograd.Function):
    """VideoGeneratorMNISTODERNN.sample_z_m as its own autograd node: gode_odernn_fwd alone (latent rows [N*T, 16],
    row n*T + t = h_{t+1}), backward = gode_odernn_bwd (GRU backward + adaptive adjoint per frame)."""

    @staticmethod
    def forward(ctx, gen, noise_host, n, T, *params):
        dev = gen.main[0].weight.device
        f32 = dict(dtype=torch.float32, device=dev)
        noise = noise_host.to(dev)                                    # [T+1, n, 16]: h_0, e_1 .. e_T
        hp = torch.empty(n, T, 16, **f32)
        z = torch.empty(n * T, 68, **f32)                             # the kernel's latent-row layout; columns 0..15 used
        nsteps = torch.zeros(T, dtype=torch.int32, device=dev)
        ns = L.lib().gode_odernn_sync_size(n)
        sync = torch.zeros(ns, dtype=torch.int32, device=dev) if ns else None
        prm = L.OdeRnnParams(*[dptr(p) for p in params])
        fop = L.OdeRnnFwdOp(p=prm, noise=dptr(noise), content=None, sel_t=None, z=dptr(z), hs=None, hp=dptr(hp),
                            nsteps=dptr(nsteps), N=n, T=T, rtol=gen.ode_rtol, atol=gen.ode_atol, zcols=68, sync=dptr(sync))
        L.run_one(fop, stream_ptr())
        ctx.keep = (gen, noise, hp, n, T, [p.detach() for p in params], sync)
        ctx.nsteps = nsteps
        return z[:, :16].contiguous()

    @staticmethod
    def backward(ctx, gout):
        gen, noise, hp, n, T, params, sync = ctx.keep
        dev = noise.device
        g = gout.contiguous()
        grads = torch.empty(L.ODERNN_NPARAM, dtype=torch.float32, device=dev)
        work = torch.empty(L.lib().gode_odernn_bwd_work_size(n), dtype=torch.float32, device=dev)
        prm = L.OdeRnnParams(*[dptr(p) for p in params])
        bop = L.OdeRnnBwdOp(p=prm, noise=dptr(noise), hp=dptr(hp), sel_t=None, gz=dptr(g), work=dptr(work),
                            grads=dptr(grads), N=n, T=T, substeps=gen.adjoint_substeps, accumulate=0, zcols=16,
                            rtol=gen.ode_rtol, atol=gen.ode_atol, sync=dptr(sync))
        L.run_one(bop, stream_ptr())
        return (None, None, None, None, *[grads[o:o + k].view(shp) for o, k, shp in _RNN_GRAD_OFFS])


class VideoGeneratorMNISTODERNN(VideoGeneratorMNIST):
    """ODE-RNN generator (models/mocogan_ode_rnn.py:21-53): per frame the hidden state is evolved by the Neural ODE
    over a unit interval (torchdiffeq default solver: dopri5, rtol 1e-7, atol 1e-9) and then updated by the GRU cell
    with fresh noise.  The pre-net `linear` is constructed (state_dict parity) but unused, as in the reference.
    Noise: the reference calls T.FloatTensor(n, d).normal_() (models/mocogan.py:297-301), i.e. the DEVICE generator on
    a GPU; here it is always drawn from the global torch CPU generator in the same order and copied to the device, so
    that runs are comparable with the CPU oracle at identical seeds ("ROCm .cuda() semantics")."""

    _plan_cls = _RnnGenPlan
    ode_rtol, ode_atol = 1e-7, 1e-9
    adjoint_substeps = 0      # 0: adaptive adjoint (torchdiffeq's behaviour); 32 was round 1's fixed discretisation

    def _init_ode_parts(self, ode_fn, dim_hidden, linear, dim_z, ngf):
        super()._init_ode_parts(ode_fn, dim_hidden, linear, dim_z, ngf)
        self.ode_fn = self._make_ode_fn(ode_fn, dim_hidden)
        self.linear = self._make_prenet(linear)

    def _param_list(self):
        dec, _ = super()._param_list()
        f, r = self.ode_fn.fn, self.recurrent
        return dec, [f[0].weight, f[0].bias, f[2].weight, f[2].bias, r.weight_ih, r.weight_hh, r.bias_ih, r.bias_hh]

    def _draw(self, num_samples, video_len):
        """content (NumPy) first, then h_0 and one e_t per frame from FloatTensor(...).normal_()
        (models/mocogan.py:252,297-301; mocogan_ode_rnn.py:44-47)."""
        content = np.random.normal(0, 1, (num_samples, self.dim_z_content)).astype(np.float32)
        # filled in place into a reused host buffer: the draws consume the generator exactly like the reference's
        # fresh FloatTensor(n, d).normal_() calls, without a ~1 MB malloc/free per call (large transient host
        # allocations next to a live HIP context showed up as 75-90 ms stalls on the GPU box)
        key = (num_samples, video_len)
        buf = self.__dict__.setdefault("_noise_bufs", {}).get(key)
        if buf is None:
            buf = self._noise_bufs[key] = torch.empty(video_len + 1, num_samples, self.dim_z_motion)
        for i in range(video_len + 1):
            buf[i].normal_()
        return torch.from_numpy(content), buf

    @staticmethod
    def _take_trajectories(noise, traj):
        return noise[:, traj].contiguous()        # noise stack [T+1, n, 16]: trajectories are the middle axis

    def _launch_latents(self, plans):
        """All prefetched ODE-RNN solves in one launch (<= 8 per launch), each with its own whole-batch error norm."""
        st = stream_ptr()
        for k in range(0, len(plans), 8):
            chunk = plans[k:k + 8]
            arr = (L.OdeRnnFwdOp * len(chunk))(*[p.fwd_op for p in chunk])
            L.call("odernn_fwd_multi", lambda: L.check(L.lib().gode_odernn_fwd_multi(arr, len(chunk), st), "gode_odernn_fwd_multi"))

    def flush_adjoints(self):
        """Launch the adjoints collected during a backward pass (see _RnnGenPlan.backward)."""
        batch = getattr(self, "_adjoint_batch", None)
        if not batch or not batch["ops"]:
            return
        ops = batch["ops"]
        st = stream_ptr()
        for k in range(0, len(ops), 8):
            chunk = ops[k:k + 8]
            arr = (L.OdeRnnBwdOp * len(chunk))(*chunk)
            L.call("odernn_bwd_multi", lambda: L.check(L.lib().gode_odernn_bwd_multi(arr, len(chunk), st), "gode_odernn_bwd_multi"))
        batch["ops"], batch["keep"] = [], []

    def _draw_motion(self, num_samples, T):
        """h_0 and one e_t per frame, each a fresh FloatTensor(n, d).normal_() on the global torch CPU generator
        (models/mocogan.py:297-301; call order of models/mocogan_ode_rnn.py:42-46) -> [T+1, n, 16]."""
        buf = torch.empty(T + 1, num_samples, self.dim_z_motion)
        for i in range(T + 1):
            buf[i].normal_()
        return buf

    def sample_z_m(self, num_samples, video_len=None):
        """models/mocogan_ode_rnn.py:39-51 -> [N*T, 16], row n*T + t = h_{t+1}: one gode_odernn_fwd launch, differentiable
        (gode_odernn_bwd).  sample_videos()/sample_images() fuse the same launch with the content broadcast and the
        decoder; this method is the reference's public surface (and what the inherited sample_z_video calls)."""
        T = video_len if video_len is not None else self.video_length
        _require_gpu(self.main[0].weight, type(self).__name__)
        noise = self._draw_motion(num_samples, T)
        _, rnn = self._param_list()
        return _RnnLatentFn.apply(self, noise, num_samples, T, *rnn)


# ==================================================================================================================
# discriminators
# ==================================================================================================================
class Noise(nn.Module):
    """Identity when use_noise is False, which is always the case in stage 3 (models/mocogan.py:20-29)."""

    def __init__(self, use_noise, sigma=0.2):
        super().__init__()
        if use_noise:
            raise NotImplementedError("use_noise=True is never used by the stage-3 scripts")
        self.use_noise, self.sigma = use_noise, sigma

    def forward(self, x):
        return x


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, strides, training, keep, arena, x, *params):
        out = plan.forward(training, x=x, x_strides=strides)
        plan.busy = keep
        ctx.plan = plan
        ctx.lease = _Lease(plan)
        ctx.x_shape = x.shape
        ctx.arena = arena
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        ctx.lease.check("discriminator")
        need_x = ctx.needs_input_grad[5]
        need_p = any(ctx.needs_input_grad[6:])
        arena = ctx.arena
        into = None
        if need_p and arena is not None and arena.active:
            into = []
            for lp in plan.params:
                wt, acc = arena.target(lp.weight)
                gt = bt = None
                if lp.gamma is not None:
                    gt, _ = arena.target(lp.gamma)
                    bt, _ = arena.target(lp.beta)
                into.append((wt, gt, bt, acc))
        flat, views, g_in = plan.backward(gout, need_input_grad=need_x, need_param_grad=need_p, into=into)
        grads = []
        if into is not None:
            grads = [None] * (len(ctx.needs_input_grad) - 6)
        elif need_p:
            for wv, gv, bv in views:
                grads.append(wv)
                if gv is not None:
                    grads += [gv, bv]
        else:
            grads = [None] * (len(ctx.needs_input_grad) - 6)
        gx = None
        if need_x:
            g = g_in                                       # [N, D, H, W, C] channels-last, allocated for this call
            gx = g.permute(0, 4, 1, 2, 3) if len(ctx.x_shape) == 5 else g.squeeze(1).permute(0, 3, 1, 2)
        return (None, None, None, None, None, gx, *grads)


class _DiscPairFn(torch.autograd.Function):
    """One pass of a discriminator over [first; second] (real and fake batches of equal shape) with per-group BatchNorm
    statistics: returns the joint raw output [2B, Do, Ho, Wo, 1]."""

    @staticmethod
    def forward(ctx, plan, strides1, strides2, training, keep, arena, x1, x2, *params):
        out = plan.forward(training, x=x1, x_strides=strides1, x2=x2, x2_strides=strides2)
        plan.busy = keep
        ctx.plan = plan
        ctx.lease = _Lease(plan)
        ctx.arena = arena
        return out

    @staticmethod
    def backward(ctx, gout):
        plan = ctx.plan
        ctx.lease.check("discriminator (paired pass)")
        if ctx.needs_input_grad[6] or ctx.needs_input_grad[7]:
            raise NotImplementedError("the paired discriminator pass is for discriminator updates (detached inputs)")
        need_p = any(ctx.needs_input_grad[8:])
        arena = ctx.arena
        nparams = len(ctx.needs_input_grad) - 8
        into = None
        if need_p and arena is not None and arena.active:
            into = []
            for lp in plan.params:
                wt, acc = arena.target(lp.weight)
                gt = bt = None
                if lp.gamma is not None:
                    gt, _ = arena.target(lp.gamma)
                    bt, _ = arena.target(lp.beta)
                into.append((wt, gt, bt, acc))
        flat, views, _ = plan.backward(gout, need_input_grad=False, need_param_grad=need_p, into=into)
        grads = [None] * nparams
        if into is None and need_p:
            grads = []
            for wv, gv, bv in views:
                grads.append(wv)
                if gv is not None:
                    grads += [gv, bv]
        return (None, None, None, None, None, None, None, None, *grads)


class _DiscBase(nn.Module):
    # ConvStack(two_lane_backward=...): the weight gradients of a backward pass on a second stream next to the input-gradient
    # chain.  UCF iteration 16.83 -> 16.53 ms, MNIST / ODE-RNN -0.3 / -0.5 % (same box, same run).
    two_lane_backward = True

    _gode_direct_grads = True

    def _specs(self, x_shape):
        raise NotImplementedError

    def _layer_params(self):
        out = []
        mods = list(self.main)
        for i, m in enumerate(mods):
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, (nn.BatchNorm2d, nn.BatchNorm3d)):
                    out.append(LayerParams(m.weight, nxt.weight, nxt.bias, nxt.running_mean, nxt.running_var,
                                           nxt.num_batches_tracked))
                else:
                    out.append(LayerParams(m.weight))
        return out

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if hasattr(self, "_pool"):
            self._pool.clear()
        return r

    def invalidate_packs(self):
        """See VideoGenerator.invalidate_packs."""
        for k in [k for k in self._pool.pack_cache if k[0] == "ver"]:
            del self._pool.pack_cache[k]

    def forward(self, input):
        _require_gpu(self.main[1].weight, type(self).__name__)
        _require_gpu(input, type(self).__name__ + " input")
        if input.dtype != torch.float32:
            raise RuntimeError("fp32 input expected")
        x = input
        if x.dim() == 5:
            strides = (x.stride(0), x.stride(2), x.stride(3), x.stride(4), x.stride(1))
        else:
            strides = (x.stride(0), 0, x.stride(2), x.stride(3), x.stride(1))
        key = tuple(x.shape)
        plan = self._pool.get(key, lambda: ConvStack(self._specs(x.shape), self._layer_params(),
                                                     self.main[1].weight.device, owns_input=False,
                                                     pack_cache=self._pool.pack_cache, two_lane_backward=self.two_lane_backward))
        params = []
        for p in self._layer_params():
            params.append(p.weight)
            if p.gamma is not None:
                params += [p.gamma, p.beta]
        keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        out = _DiscFn.apply(plan, strides, self.training, keep, getattr(self, "_gode_arena", None), x, *params)      # [B, Do, Ho, Wo, 1]
        h = out.permute(0, 4, 1, 2, 3)
        if input.dim() == 4:
            h = h.squeeze(2)       # (a squeeze, not h[:, :, 0]: select's backward allocates a zero tensor and copies)
        return h.squeeze(), None


    @staticmethod
    def _strides5(x):
        if x.dim() == 5:
            return (x.stride(0), x.stride(2), x.stride(3), x.stride(4), x.stride(1))
        return (x.stride(0), 0, x.stride(2), x.stride(3), x.stride(1))

    def forward_pair_joint(self, first, second):
        """D(first) and D(second) in ONE pass -- the same launches over [first; second] with PER-GROUP BatchNorm batch
        statistics, i.e. the arithmetic of the reference's two calls `dis(real)`, `dis(fake)` (mnist_moco_ode.py:119-124,
        137-143): running statistics receive first's update, then second's.  Returns the joint raw logits
        [2B, Do, Ho, Wo, 1] (first group first); GanTrainer feeds them to bce_with_logits_halves.  Twice the rows per
        launch fill the GPU better (video-D GEMMs +14 %) and every elementwise / reduction launch is issued once."""
        _require_gpu(self.main[1].weight, type(self).__name__)
        for t in (first, second):
            _require_gpu(t, type(self).__name__ + " input")
            if t.dtype != torch.float32:
                raise RuntimeError("fp32 input expected")
        if tuple(first.shape) != tuple(second.shape):
            raise RuntimeError("paired pass: both inputs must have the same shape")
        if first.requires_grad or second.requires_grad:
            raise RuntimeError("paired pass: inputs must be detached (it is the discriminator-update pass)")
        joint_shape = (2 * first.shape[0],) + tuple(first.shape[1:])
        key = ("pair",) + tuple(first.shape)
        plan = self._pool.get(key, lambda: ConvStack(self._specs(joint_shape), self._layer_params(),
                                                     self.main[1].weight.device, owns_input=False,
                                                     pack_cache=self._pool.pack_cache, groups=2,
                                                     two_lane_backward=self.two_lane_backward))
        params = []
        for p in self._layer_params():
            params.append(p.weight)
            if p.gamma is not None:
                params += [p.gamma, p.beta]
        keep = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _DiscPairFn.apply(plan, self._strides5(first), self._strides5(second), self.training, keep,
                                 getattr(self, "_gode_arena", None), first, second, *params)

    def forward_pair(self, first, second):
        """-> ((logits_first, None), (logits_second, None)) with forward()'s shapes."""
        out = self.forward_pair_joint(first, second)
        B = first.shape[0]
        res = []
        for h in (out[:B], out[B:]):
            h = h.permute(0, 4, 1, 2, 3)
            if first.dim() == 4:
                h = h.squeeze(2)
            res.append((h.squeeze(), None))
        return tuple(res)


class PatchImageDiscriminator(_DiscBase):
    """models/mocogan.py:66-93: Conv2d k4 s2 p1 x4 (C->ndf->2ndf->4ndf->1), BN on layers 2-3, LeakyReLU(0.2)."""

    def __init__(self, n_channels, ndf=64, use_noise=False, noise_sigma=None):
        super().__init__()
        self.use_noise, self.n_channels, self.ndf = use_noise, n_channels, ndf
        self.main = nn.Sequential(
            Noise(use_noise, sigma=noise_sigma), nn.Conv2d(n_channels, ndf, 4, 2, 1, bias=False),
            nn.LeakyReLU(0.2, inplace=True),
            Noise(use_noise, sigma=noise_sigma), nn.Conv2d(ndf, ndf * 2, 4, 2, 1, bias=False), nn.BatchNorm2d(ndf * 2),
            nn.LeakyReLU(0.2, inplace=True),
            Noise(use_noise, sigma=noise_sigma), nn.Conv2d(ndf * 2, ndf * 4, 4, 2, 1, bias=False),
            nn.BatchNorm2d(ndf * 4), nn.LeakyReLU(0.2, inplace=True),
            Noise(use_noise, sigma=noise_sigma), nn.Conv2d(ndf * 4, 1, 4, 2, 1, bias=False))
        self._pool = _Pool()

    def _specs(self, x_shape):
        B, Cc, H, W = x_shape
        ndf = self.ndf
        chans = [Cc, ndf, ndf * 2, ndf * 4, 1]
        specs = []
        for i in range(4):
            Ho, Wo = conv_out(H, 4, 2, 1), conv_out(W, 4, 2, 1)
            specs.append(LayerSpec(make_geom(B, chans[i], chans[i + 1], (1, H, W), (1, Ho, Wo), (1, 4, 4), (1, 2, 2),
                                             (0, 1, 1)), L.FPROP, L.ACT_LRELU if i < 3 else L.ACT_NONE, i in (1, 2)))
            H, W = Ho, Wo
        return specs


class VideoDiscriminator(_DiscBase):
    """models/mocogan.py:129-164: Conv3d ksize^3, stride (1,2,2), pad (0,1,1) x4 + a final ksize^3 s1 p0 conv."""

    def __init__(self, n_channels, n_output_neurons=1, bn_use_gamma=True, use_noise=False, noise_sigma=None, ndf=64,
                 ksize=4):
        super().__init__()
        if n_output_neurons != 1:
            raise NotImplementedError("n_output_neurons != 1 is only used by the categorical discriminator")
        self.n_channels, self.n_output_neurons, self.use_noise = n_channels, n_output_neurons, use_noise
        self.bn_use_gamma, self.ndf, self.ksize = bn_use_gamma, ndf, ksize
        st, pd = (1, 2, 2), (0, 1, 1)
        self.main = nn.Sequential(
            Noise(use_noise, sigma=noise_sigma), nn.Conv3d(n_channels, ndf, ksize, stride=st, padding=pd, bias=False),
            nn.LeakyReLU(0.2, inplace=True),
            Noise(use_noise, sigma=noise_sigma), nn.Conv3d(ndf, ndf * 2, ksize, stride=st, padding=pd, bias=False),
            nn.BatchNorm3d(ndf * 2), nn.LeakyReLU(0.2, inplace=True),
            Noise(use_noise, sigma=noise_sigma), nn.Conv3d(ndf * 2, ndf * 4, ksize, stride=st, padding=pd, bias=False),
            nn.BatchNorm3d(ndf * 4), nn.LeakyReLU(0.2, inplace=True),
            Noise(use_noise, sigma=noise_sigma), nn.Conv3d(ndf * 4, ndf * 8, ksize, stride=st, padding=pd, bias=False),
            nn.BatchNorm3d(ndf * 8), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv3d(ndf * 8, n_output_neurons, ksize, 1, 0, bias=False))
        self._pool = _Pool()

    def _specs(self, x_shape):
        B, Cc, D, H, W = x_shape
        ndf, k = self.ndf, self.ksize
        chans = [Cc, ndf, ndf * 2, ndf * 4, ndf * 8]
        specs = []
        for i in range(4):
            Do, Ho, Wo = conv_out(D, k, 1, 0), conv_out(H, k, 2, 1), conv_out(W, k, 2, 1)
            specs.append(LayerSpec(make_geom(B, chans[i], chans[i + 1], (D, H, W), (Do, Ho, Wo), (k, k, k), (1, 2, 2),
                                             (0, 1, 1)), L.FPROP, L.ACT_LRELU, i > 0))
            D, H, W = Do, Ho, Wo
        Do, Ho, Wo = conv_out(D, k, 1, 0), conv_out(H, k, 1, 0), conv_out(W, k, 1, 0)
        specs.append(LayerSpec(make_geom(B, chans[4], 1, (D, H, W), (Do, Ho, Wo), (k, k, k), (1, 1, 1), (0, 0, 0)),
                               L.FPROP, L.ACT_NONE, False))
        return specs
