"""The training iteration of mnist_moco_ode.py:113-163 (== ucf_moco_ode.py:115-165) restated as `train_step`, with
the loss and optimiser lowered to libgode kernels, plus single-node data parallelism: one process per GPU, one RCCL
all-reduce of a flat fp32 gradient bucket per optimiser step (torch.distributed backend "nccl" is RCCL on ROCm).

The reference has no `train_step` function (the body is inline in train()); the call sequence, the RNG draws and
the d_iters=2 structure below follow that body line by line.
"""
from __future__ import annotations

import contextlib
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib as L
from .engine import stream_ptr
from .modules import PatchImageDiscriminator, VideoDiscriminator, VideoGenerator, VideoGeneratorMNISTODE


def host_cpu_quota():
    """CPUs this process may actually use: the cgroup CPU quota if there is one, else the affinity mask size."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def limit_host_threads(n=None):
    """torch's intra-op pool defaults to one thread per visible core (128 on the MI355X host) while the container's
    cgroup quota is 16 CPUs: the idle pool spin-waits after every parallel region, the quota is exhausted and the
    whole process is throttled for the rest of the 100 ms period (seen as random 75-90 ms stalls in the host-side
    noise draws).  Call this once at start-up (bench.py does) to size the pool to the quota."""
    if n is None:
        import os
        # one process per GPU: the quota is shared by the ranks of this node
        local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
        n = max(1, min(16, host_cpu_quota() // max(1, local)))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return n


def freeze_host_gc():
    """One full-heap (generation-2) pass of Python's cyclic garbage collector over the heap that `import torch` and
    the plans leave behind takes 75-100 ms, and the first one of a run lands after ~60 training iterations: one
    iteration of 106 ms among 12.6 ms ones (scripts/diag_iter_values.py; the GPU drains and idles meanwhile).  Moving
    the long-lived objects to the permanent generation after warm-up makes later passes walk only young objects.
    GanTrainer calls this once after its second iteration; bench.py before its timed loops."""
    import gc
    gc.collect()
    gc.freeze()


# ------------------------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------------------------
_UNIT = {}


def unit_grad(device):
    """A cached device scalar 1.0: `loss.backward(gradient=unit_grad(dev))` spares autograd's ones_like fill, and the
    loss nodes below recognise it (by address) and hand their stored gradients on without multiplying."""
    t = _UNIT.get(device)
    if t is None:
        t = _UNIT[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


def _scaled(grad, gout):
    if gout.data_ptr() == unit_grad(gout.device).data_ptr():
        return grad
    return grad * gout      # a caller-supplied upstream gradient (not the training loop's path)


class _BceConstFn(torch.autograd.Function):
    """sum_k BCEWithLogits(logits_k, target_k) (mean-reduced each) for 1 or 2 logit tensors in one autograd node: the
    second term accumulates into the same device scalar (gode_bce_logits accumulate=1), so a discriminator loss
    `bce(real, 1) + bce(fake, 0)` (mnist_moco_ode.py:126-128) needs no elementwise add."""

    @staticmethod
    def forward(ctx, *args):
        pairs = [(args[i], args[i + 1]) for i in range(0, len(args), 2)]
        loss = torch.empty((), dtype=torch.float32, device=pairs[0][0].device)
        grads = []
        st = stream_ptr()
        for k, (logits, target) in enumerate(pairs):
            x = logits.contiguous()
            grad = torch.empty_like(x)
            op = L.BceOp(logits=x.data_ptr(), grad=grad.data_ptr(), loss=loss.data_ptr(), n=x.numel(),
                         target=float(target), gscale=1.0, accumulate=1 if k else 0)
            L.run_one(op, st)
            grads.append(grad)
        ctx.save_for_backward(*grads)
        return loss

    @staticmethod
    def backward(ctx, gout):
        out = []
        for g in ctx.saved_tensors:
            out += [_scaled(g, gout), None]
        return tuple(out)


def bce_with_logits_const(logits: torch.Tensor, target: float) -> torch.Tensor:
    """nn.BCEWithLogitsLoss()(logits, full_like(logits, target)) -- mean reduction (mnist_moco_ode.py:89,126-128)."""
    if not logits.is_cuda:
        raise RuntimeError("bce_with_logits_const runs only on the GPU through libgode.so")
    return _BceConstFn.apply(logits, target)


class _BceHalvesFn(torch.autograd.Function):
    """bce(x[:B], t_first) + bce(x[B:], t_second), each mean-reduced over its half, on ONE joint logits tensor (the
    paired discriminator pass): one autograd node, the gradient is one tensor in the joint layout."""

    @staticmethod
    def forward(ctx, joint, t_first, t_second):
        x = joint.contiguous()
        n = x.numel() // 2
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        st = stream_ptr()
        for k, tgt in enumerate((t_first, t_second)):
            L.run_one(L.BceOp(logits=x.data_ptr() + 4 * n * k, grad=grad.data_ptr() + 4 * n * k, loss=loss.data_ptr(), n=n,
                              target=float(tgt), gscale=1.0, accumulate=k), st)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return _scaled(grad, gout), None, None


def bce_with_logits_halves(joint: torch.Tensor, target_first: float, target_second: float) -> torch.Tensor:
    """nn.BCEWithLogitsLoss()(x[:B], t1) + nn.BCEWithLogitsLoss()(x[B:], t2) for the joint output of
    forward_pair_joint (mnist_moco_ode.py:126-128,145-147)."""
    if not joint.is_cuda or joint.shape[0] % 2:
        raise RuntimeError("bce_with_logits_halves: a CUDA tensor with an even leading dimension is expected")
    return _BceHalvesFn.apply(joint, target_first, target_second)


def bce_with_logits_pair(logits_a, target_a: float, logits_b, target_b: float) -> torch.Tensor:
    """bce_with_logits_const(a, ta) + bce_with_logits_const(b, tb) as one node (same arithmetic: each term is
    mean-reduced on its own, the two means are added in fp32)."""
    if not (logits_a.is_cuda and logits_b.is_cuda):
        raise RuntimeError("bce_with_logits_pair runs only on the GPU through libgode.so")
    return _BceConstFn.apply(logits_a, target_a, logits_b, target_b)


# ------------------------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------------------------
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps, weight_decay) semantics (L2-coupled decay, parameters whose .grad is None
    are skipped -- the generator's dead GRU cell, models/mocogan.py:198) on libgode's Adam kernel.  state_dict()
    uses torch.optim.Adam's keys (step / exp_avg / exp_avg_sq) so checkpoints interchange."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, weight_decay=1e-5):
        # the extra keys are torch.optim.Adam's remaining hyper-parameters at their defaults, carried so that a
        # state_dict saved here loads into torch.optim.Adam and vice versa (mnist_moco_ode.py:92-103,175-190)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False,
                                      fused=None, decoupled_weight_decay=False))

    @staticmethod
    def step_coefficients(lr, beta1, beta2, step):
        """(lr / (1 - beta1^step), sqrt(1 - beta2^step)) in gode_adam_multi's arithmetic: the hyper-parameters pass
        through C floats, the bias corrections are evaluated in double and rounded to float."""
        import numpy as np
        lr, b1, b2 = float(np.float32(lr)), float(np.float32(beta1)), float(np.float32(beta2))
        return (float(np.float32(lr / (1.0 - b1 ** float(step)))), float(np.float32((1.0 - b2 ** float(step)) ** 0.5)))

    @torch.no_grad()
    def step(self, closure=None, grads: Optional[dict] = None, gscale: float = 1.0, feed=None):
        """grads: optional {param: tensor} overriding .grad (views into an all-reduced bucket); gscale multiplies
        every gradient (1/world_size for data parallelism).  All tensors of a param group are updated by ONE launch
        (gode_adam_multi) driven by a small device table of pointers, uploaded only when a pointer changed.
        feed: a HostFeed (graph capture): the step-dependent coefficients are then read from the feed's device block
        (gode_adam_multi_dev), so the recorded launch stays valid while the step count advances."""
        assert closure is None
        import numpy as np
        lib = L.lib()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            rows, keep, step, max_n, dev = [], [], None, 0, None
            for p in group["params"]:
                g = grads.get(p) if grads is not None else p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam runs only on the GPU through libgode.so")
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                s_now = int(st["step"].item())
                if step is None:
                    step = s_now
                if s_now != step:       # parameters on different step counts: fall back to one launch each
                    if feed is not None:
                        raise RuntimeError("graph capture needs all parameters of a group on the same Adam step")
                    L.run_one(L.AdamOp(p=p.data_ptr(), g=g.data_ptr(), m=st["exp_avg"].data_ptr(),
                                       v=st["exp_avg_sq"].data_ptr(), n=p.numel(), lr=group["lr"], beta1=b1, beta2=b2,
                                       eps=group["eps"], weight_decay=group["weight_decay"], gscale=gscale,
                                       step=s_now), stream_ptr())
                else:
                    rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                 p.numel()))
                    max_n = max(max_n, p.numel())
                keep.append(g)
                p._gode_ver = getattr(p, "_gode_ver", 0) + 1     # updated through a raw pointer: bump our own version
                dev = p.device
            if rows:
                tbl = getattr(self, "_tables", None)
                if tbl is None:
                    tbl = self._tables = {}
                n = len(rows)
                ent = tbl.get(id(group))
                key = tuple(rows)
                if ent is None or ent.get("key") != key:
                    # (the pointers of parameters, arena gradients and moments do not change from step to step: the
                    # table is uploaded once, not per step)
                    if feed is not None:
                        raise RuntimeError("graph capture: the Adam pointer table must exist already (run eager "
                                           "iterations first)")
                    if ent is None or ent["host"].shape[0] != n:
                        ent = tbl[id(group)] = dict(host=torch.empty((n, 5), dtype=torch.int64).pin_memory(),
                                                    dev=torch.empty((n, 5), dtype=torch.int64, device=dev), ev=None)
                    if ent["ev"] is not None:
                        ent["ev"].synchronize()          # the previous upload from this pinned buffer has completed
                    ent["key"] = key
                    ent["host"].copy_(torch.from_numpy(np.asarray(rows, dtype=np.int64)))
                    ent["dev"].copy_(ent["host"], non_blocking=True)
                    if ent["ev"] is None:
                        ent["ev"] = torch.cuda.Event()
                    ent["ev"].record()
                if feed is not None:
                    coef = feed.adam_slot(self, gi, group["lr"], b1, b2, step)
                    L.call("adam_multi", lambda: L.check(
                        lib.gode_adam_multi_dev(ent["dev"].data_ptr(), n, max_n, b1, b2, group["eps"],
                                                group["weight_decay"], gscale, coef, stream_ptr()), "gode_adam_multi_dev"))
                else:
                    L.call("adam_multi", lambda: L.check(
                        lib.gode_adam_multi(ent["dev"].data_ptr(), n, max_n, group["lr"], b1, b2, group["eps"],
                                            group["weight_decay"], gscale, step, stream_ptr()), "gode_adam_multi"))
                self._keep = keep
        return None

    def advance_steps(self, group_index=0):
        """Host-side bookkeeping of one optimiser step that a replayed graph performed on the device: the `step`
        entries of the state (checkpoints, the next coefficient upload) move on by one."""
        for p in self.param_groups[group_index]["params"]:
            st = self.state.get(p)
            if st:
                st["step"] += 1

    def current_step(self, group_index=0):
        for p in self.param_groups[group_index]["params"]:
            st = self.state.get(p)
            if st:
                return int(st["step"].item())
        return 0


class HostFeed:
    """Everything the host contributes to one training iteration besides the launch itself -- the latent noise drawn
    from the NumPy / torch CPU generators (the reference's RNG contract) and Adam's step-dependent coefficients -- laid
    out in ONE pinned block that is uploaded with ONE copy into a static device block read by the captured graph.
    During capture the block's layout is recorded as a script (in program order); before every replay the script is
    walked again: same draws in the same order, coefficients for the new step counts."""

    FLOATS = 1 << 18

    def __init__(self, device):
        self.device = device
        self.dev = torch.zeros(self.FLOATS, dtype=torch.float32, device=device)
        self.pinned = [torch.zeros(self.FLOATS, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.events = [None, None]
        self.parity = 0
        self.script, self.size = [], 0
        self.recording = False

    # -- capture ---------------------------------------------------------------------------------------------------
    def _reserve(self, n):
        off = self.size
        self.size = off + ((n + 3) // 4) * 4                 # 16-byte aligned regions
        if self.size > self.FLOATS:
            raise RuntimeError("HostFeed block too small")
        return off

    def noise_region(self, gen, select, num_samples, T, nfloats):
        """-> (pinned view to fill now, device view the captured copy reads)"""
        assert self.recording
        off = self._reserve(nfloats)
        self.script.append(("noise", gen, select, num_samples, T, off, nfloats))
        return self.pinned[self.parity][off:off + nfloats], self.dev[off:off + nfloats]

    def adam_slot(self, opt, group_index, lr, b1, b2, step):
        assert self.recording
        off = self._reserve(2)
        self.script.append(("adam", opt, group_index, lr, b1, b2, off))
        c = FusedAdam.step_coefficients(lr, b1, b2, step)
        self.pinned[self.parity][off], self.pinned[self.parity][off + 1] = c[0], c[1]
        return self.dev[off:off + 2].data_ptr()

    # -- replay ----------------------------------------------------------------------------------------------------
    def begin(self):
        """Next pinned buffer (waits until the upload that last used it has been consumed)."""
        self.parity ^= 1
        ev = self.events[self.parity]
        if ev is not None:
            ev.synchronize()
        return self.pinned[self.parity]

    def fill_from_script(self):
        buf = self.begin()
        for ent in self.script:
            if ent[0] == "noise":
                _, gen, select, num_samples, T, off, nfloats = ent
                gen._fill_host_inputs(buf[off:off + nfloats], select, num_samples, T)
            else:
                _, opt, gi, lr, b1, b2, off = ent
                opt.advance_steps(gi)
                c = FusedAdam.step_coefficients(lr, b1, b2, opt.current_step(gi))
                buf[off], buf[off + 1] = c[0], c[1]

    def upload(self):
        """One H2D copy of the used part of the block on the current stream (ordered before the replay that reads it,
        after the replay that read the previous contents)."""
        n = self.size
        self.dev[:n].copy_(self.pinned[self.parity][:n], non_blocking=True)
        if self.events[self.parity] is None:
            self.events[self.parity] = torch.cuda.Event()
        self.events[self.parity].record()


# ------------------------------------------------------------------------------------------------------------------
# data parallel gradient bucket
# ------------------------------------------------------------------------------------------------------------------
class GradBucket:
    """Flat fp32 bucket over the parameters of ONE network that received a gradient.  gather() packs the grads,
    all_reduce() sums them over the process group (one collective per optimiser step; RCCL over xGMI on the GPU box,
    gloo in the CPU tests), views() hands the reduced slices to the optimiser without copying back."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = [p for p in params]
        self.flat = None
        self.live = []

    def gather(self):
        self.live = [p for p in self.params if p.grad is not None]
        if not self.live:
            return None
        n = sum(p.numel() for p in self.live)
        if self.flat is None or self.flat.numel() != n or self.flat.device != self.live[0].device:
            self.flat = torch.empty(n, dtype=torch.float32, device=self.live[0].device)
        off = 0
        for p in self.live:
            k = p.numel()
            self.flat[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        return self.flat

    def all_reduce(self, group=None):
        if self.flat is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def views(self):
        out, off = {}, 0
        for p in self.live:
            k = p.numel()
            out[p] = self.flat[off:off + k].view_as(p)
            off += k
        return out


class GradArena:
    """One flat fp32 gradient buffer per network, written by the backward kernels themselves: the first weight-gradient
    / BatchNorm-gradient / adjoint kernel that produces a parameter's gradient in a step stores it (and `p.grad`
    becomes a view of the arena), later ones (the second discriminator pass, the image path of the generator) add in
    place.  Replaces autograd's AccumulateGrad adds (64 small launches per iteration) and, under data parallelism,
    the pack-into-bucket copies: the arena IS the all-reduce bucket.  `order`: parameters in arena order (the
    generator puts its pre-net + ODEFunc tensors last, in the adjoint kernel's output order)."""

    def __init__(self, order: Sequence[torch.nn.Parameter]):
        self.params = list(order)
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, off = {}, 0
        for p in self.params:
            self.views[p] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.written = set()
        self.active = False

    def begin(self):
        """Start of an optimiser step (replaces zero_grad): every .grad is dropped, nothing is written yet."""
        for p in self.params:
            p.grad = None
        self.written.clear()
        self.active = True

    def end(self):
        self.active = False

    def target(self, p):
        """-> (tensor the kernel writes to, accumulate?) and binds p.grad to it on first use."""
        v = self.views[p]
        acc = p in self.written
        if not acc:
            self.written.add(p)
            p.grad = v
        return v, acc


# ------------------------------------------------------------------------------------------------------------------
# the iteration
# ------------------------------------------------------------------------------------------------------------------
def build_mnist(ngf=64, ndf=64):
    """The three networks of mnist_moco_ode.py:75-78."""
    return VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=ngf), VideoDiscriminator(1, ksize=2, ndf=ndf), \
        PatchImageDiscriminator(1, ndf=ndf)


def build_ucf(ngf=64, ndf=64):
    """ucf_moco_ode.py:77-80 with dim_hidden=16 (the shipped call omits it and raises)."""
    return VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=ngf), VideoDiscriminator(3, ndf=ndf), \
        PatchImageDiscriminator(3, ndf=ndf)


def _shards(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


class GanTrainer:
    """Owns the three networks and their optimisers; step() is one outer iteration of the reference loop.

    Data parallelism: with an initialised process group every rank runs the whole iteration on its own
    shard with per-replica BatchNorm statistics and ONE all-reduce (sum) of each network's flat gradient arena per
    optimiser step; 1/world is folded into the Adam kernel.  Construction broadcasts rank 0's parameters and buffers so
    that replicas start (and, with identical updates, stay) equal.  The same semantics are available on ONE GPU as
    *virtual replicas*: pass a list of shard tensors instead of a tensor and every optimiser step accumulates the
    shards' gradients in the arena (each shard its own forward/backward, own BatchNorm statistics) and applies
    1/(world * shards) -- which is how BASELINE configs[2] (batch 256 = 8 x 32) is parity-tested on one device."""

    def __init__(self, gen, dis_vid, dis_img, lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5, d_iters=2,
                 process_group=None, freeze_d_in_g_step=True, freeze_gc=False, direct_grads=True, sync_replicas=True,
                 overlap_image_d=True, graph=False, pair_d_passes=True, prefetch_latents=True, overlap_allreduce=True,
                 joint_generator_passes=True):
        self.gen, self.dis_vid, self.dis_img = gen, dis_vid, dis_img
        mk = lambda m: FusedAdam(m.parameters(), lr=lr, betas=betas, weight_decay=weight_decay)  # noqa: E731
        self.gen_opt, self.vid_opt, self.img_opt = mk(gen), mk(dis_vid), mk(dis_img)
        self.d_iters, self.group = d_iters, process_group
        self.freeze_d = freeze_d_in_g_step
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if self.world > 1 and sync_replicas:
            self.broadcast_state()
        self.buckets = {id(m): GradBucket(list(m.parameters())) for m in (gen, dis_vid, dis_img)}
        # overlap_image_d: inside step() the image discriminator's forward/backward/Adam (some 50 latency-bound
        # launches of 5-30 us each at batch 32) run on a side stream next to the video-discriminator step's big GEMMs.
        # Everything that touches the generator stays on the caller's stream in program order (its BatchNorm running
        # statistics are updated by every sample_* call), each stream executes the same kernels in the same order as
        # the serial schedule, so results are bit-identical to overlap_image_d=False.
        # pair_d_passes: in the discriminator steps D(real) and D(fake) run as ONE pass over [real; fake] with per-group
        # BatchNorm statistics (forward_pair_joint): same arithmetic per element, half the launches, fuller GEMM grids.
        self.pair_d = bool(pair_d_passes)
        # prefetch_latents: step() announces the iteration's sample_images / sample_videos calls to the generator up front
        # (VideoGenerator.prefetch_latents): the host draws are made in the same order, all latent solves go out at once on
        # a side stream.  Bit-identical results; what it buys is the ODE-RNN generator's six one-workgroup solves per
        # iteration running side by side in one launch, and the two adjoints of the G step likewise (flush_adjoints).
        self.prefetch = bool(prefetch_latents)
        # joint_generator_passes: the two generator calls between two generator updates (sample_images + sample_videos of
        # an inner discriminator pass; sample_videos + sample_images of the G step) are decoded in one pass over both
        # batches with per-call BatchNorm statistics (VideoGenerator.sample_pair)
        self.joint_g = bool(joint_generator_passes)
        self._paired_fake_vid = None
        # overlap_allreduce (world > 1): every all-reduce is issued asynchronously (the process group's own stream) and the
        # optimiser step that consumes it is DEFERRED to the point where that network is next needed -- a discriminator's
        # all-reduce then runs under the next no-grad generator forward, and the generator's arena goes out in two parts:
        # the decoder's 13 MB as soon as the last decoder backward kernel is queued (under the latent adjoint), the
        # ODEFunc / GRU tail after the adjoint.  Same kernels in the same per-stream order: bit-identical to the serial form.
        self.overlap_ar = bool(overlap_allreduce)
        self._pending = {}            # id(model) -> (opt, works, gscale, feed)
        self._early = {}              # id(model) -> (work, floats): the part of an arena whose all-reduce is already out
        self.skip_allreduce = False   # measurement only (bench.py): run the schedule without collectives
        self._side = None
        if overlap_image_d and next(dis_img.parameters()).is_cuda:
            self._side = torch.cuda.Stream(device=next(dis_img.parameters()).device)
        self._side_pending = False
        # graph: after two eager iterations (every plan, program and pointer table exists, the pack pattern is the
        # steady-state one) step() captures ONE whole iteration -- both streams, autograd's backward passes, the five
        # Adam launches -- in a HIP graph and from then on replays it: per iteration the host draws the latent noise
        # (same generators, same order: HostFeed), uploads it with one copy and launches the graph.  Measured at
        # BASELINE configs[1] the eager iteration is within ~25 % of being host-bound (scripts/host_floor.py: 7.5-8.2 ms
        # of Python + launch calls per iteration against ~11 ms of kernels); the replay needs ~2 ms of host time.
        # Single-process only (a captured RCCL all-reduce is not attempted); arena gradients required.
        self._graph_mode = bool(graph)
        self._graph = None
        self._feed = None
        if self._graph_mode and self.world > 1:
            raise NotImplementedError("GanTrainer(graph=True) is single-process; data-parallel runs use the eager schedule")
        # freeze_gc: call freeze_host_gc() after the second iteration (process-global, hence opt-in; bench.py and
        # long training loops want it, see its docstring)
        self._iters, self._freeze_gc = 0, freeze_gc
        # direct_grads: backward kernels write into a per-network GradArena (see its docstring); CPU tensors (the gloo
        # tests) and direct_grads=False keep the stock autograd accumulation + GradBucket
        self.arenas = {}
        if direct_grads:
            for m in (gen, dis_vid, dis_img):
                ps = list(m.parameters())
                if ps and ps[0].is_cuda and getattr(m, "_gode_direct_grads", False):
                    tail = m._arena_tail() if hasattr(m, "_arena_tail") else []
                    ids = {id(p) for p in tail}
                    arena = GradArena([p for p in ps if id(p) not in ids] + tail)
                    self.arenas[id(m)] = arena
                    m._gode_arena = arena

    def broadcast_state(self, src=0):
        """Every replica takes rank `src`'s parameters and buffers (BatchNorm running statistics included)."""
        for m in (self.gen, self.dis_vid, self.dis_img):
            with torch.no_grad():
                for t in list(m.parameters()) + list(m.buffers()):
                    dist.broadcast(t.detach(), src=src, group=self.group)
            if hasattr(m, "invalidate_packs"):
                m.invalidate_packs()

    # -- collectives ---------------------------------------------------------------------------------------------------
    def _all_reduce(self, flat, async_op):
        """One sum over the replicas of a flat fp32 gradient buffer (RCCL over xGMI; gloo in the rehearsals)."""
        if self.skip_allreduce:
            return None
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _flush(self, model):
        """The deferred optimiser step of `model`, if one is pending: wait for its all-reduce(s), then Adam."""
        ent = self._pending.pop(id(model), None)
        if ent is None:
            return
        opt, works, gscale, feed = ent
        for w in works:
            if w is not None:
                w.wait()              # (the current stream waits; the host does not block with RCCL)
        opt.step(gscale=gscale, feed=feed)

    def _early_decoder_allreduce(self):
        a = self.arenas[id(self.gen)]
        tail = sum(p.numel() for p in (self.gen._arena_tail() if hasattr(self.gen, "_arena_tail") else []))
        head = a.flat.numel() - tail
        self._early[id(self.gen)] = (self._all_reduce(a.flat[:head], True), head)

    def flush_all(self):
        for m in (self.dis_img, self.dis_vid, self.gen):
            self._flush(m)

    def time_collectives(self, reps=10):
        """Milliseconds of each of the iteration's all-reduces on its own (same buffers, nothing else running): what the
        fabric charges for them; compare with the iteration time of a run with skip_allreduce to see what is exposed."""
        out = {}
        if self.world <= 1:
            return out
        for name, m in (("dis_img", self.dis_img), ("dis_vid", self.dis_vid), ("gen", self.gen)):
            a = self.arenas.get(id(m))
            if a is None:
                continue
            buf = torch.zeros_like(a.flat)
            for _ in range(2):
                dist.all_reduce(buf, group=self.group)
            torch.cuda.synchronize()
            dist.barrier(group=self.group)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                dist.all_reduce(buf, group=self.group)
            e1.record()
            torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) / reps
        return out

    def _begin(self, model, opt):
        self._flush(model)            # the previous step's deferred update reads this arena
        a = self.arenas.get(id(model))
        if a is not None:
            # (re)bind: the autograd nodes look the arena up on the module, and a second trainer built on the same
            # networks (bench.py's graph-mode twin) would otherwise leave this trainer reducing an arena nobody writes
            model._gode_arena = a
            a.begin()
        else:
            opt.zero_grad()

    def _opt_step(self, model, opt, nshards=1):
        a = self.arenas.get(id(model))
        gscale = 1.0 / (self.world * nshards)
        if a is not None:
            if getattr(model, "_gode_arena", None) is not a:
                raise RuntimeError("another GanTrainer re-bound this network's gradient arena in the middle of an optimiser "
                                   "step: the backward kernels wrote into its arena, not this one")
            a.end()
            feed = getattr(opt, "_feed", None)
            feed = feed if (feed is not None and feed.recording) else None
            if self.world > 1 and self.overlap_ar and feed is None:
                # the arena is the bucket: no packing copies.  Parts that went out early (the generator's decoder block,
                # see g_step) are skipped here; the optimiser step waits in _flush, where the network is next needed
                early = self._early.pop(id(model), None)
                if early is None:
                    works = [self._all_reduce(a.flat, True)]
                else:
                    works = [early[0], self._all_reduce(a.flat[early[1]:], True)]
                self._pending[id(model)] = (opt, works, gscale, None)
                return
            if self.world > 1:       # serial form: one collective, then the update
                self._all_reduce(a.flat, False)
            opt.step(gscale=gscale, feed=feed)
            return
        if self.world > 1:
            b = self.buckets[id(model)]
            b.gather()
            b.all_reduce(self.group)
            opt.step(grads=b.views(), gscale=gscale)
        else:
            opt.step(gscale=gscale)

    @staticmethod
    def _mean(losses):
        if len(losses) == 1:
            return losses[0]
        return torch.stack(losses).mean()

    def d_image_step(self, real_img, _join=True, _defer=False, _pair_videos=0):
        """mnist_moco_ode.py:115-131.  real_img: [B,C,H,W], or a list of such shards (virtual replicas).
        _pair_videos (step() only): the batch size of the d_video_step that follows -- its fake videos are generated here,
        in the same decoder pass as this step's fake images (VideoGenerator.sample_pair), and kept for it."""
        shards = _shards(real_img)
        side = self._side
        main = torch.cuda.current_stream() if side is not None else None
        if side is not None and not self._side_pending:
            side.wait_stream(main)          # inputs prepared on the caller's stream; later passes only wait for `fake`
            self._side_pending = True
        on_side = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        with on_side():
            self._begin(self.dis_img, self.img_opt)
        losses = []
        for x in shards:
            B = x.shape[0]
            if not self.pair_d:
                with on_side():
                    pr, _ = self.dis_img(x)
            with torch.no_grad():                        # generator work stays on the caller's stream
                if _pair_videos:
                    self._paired_fake_vid, (fake, _) = self.gen.sample_pair(_pair_videos, B, images_first=True)
                else:
                    fake, _ = self.gen.sample_images(B)
            if side is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                fake.record_stream(side)
            with on_side():
                if self.pair_d:
                    loss = bce_with_logits_halves(self.dis_img.forward_pair_joint(x, fake), 1.0, 0.0)
                else:
                    pf, _ = self.dis_img(fake)
                    loss = bce_with_logits_pair(pr, 1.0, pf, 0.0)
                loss.backward(gradient=unit_grad(loss.device))
            losses.append(loss.detach())
        with on_side():
            self._opt_step(self.dis_img, self.img_opt, len(shards))
            if not _defer:
                self._flush(self.dis_img)      # called on its own: the update has landed when the call returns
            out = self._mean(losses)
        if _join:
            self._join_side()
        return out

    def _join_side(self):
        """The caller's stream waits for the image-discriminator work queued on the side stream."""
        if self._side is not None and self._side_pending:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_pending = False

    def d_video_step(self, real_vid, _defer=False):
        """mnist_moco_ode.py:133-150.  real_vid: [B,T,C,H,W], or a list of such shards."""
        shards = _shards(real_vid)
        losses = []
        begun = False
        for x in shards:
            B = x.shape[0]
            real = x.transpose(1, 2)                            # [B,T,C,H,W] -> [B,C,T,H,W] view, read in place
            if not self.pair_d and not begun:
                self._begin(self.dis_vid, self.vid_opt)
                begun = True
            if not self.pair_d:
                pr, _ = self.dis_vid(real)
            with torch.no_grad():
                fake, self._paired_fake_vid = self._paired_fake_vid, None
                if fake is not None:                     # generated by the d_image_step before (see _pair_videos)
                    fake = fake[0]
                    assert fake.shape[0] == B
                else:
                    fake, _ = self.gen.sample_videos(B)
            if not begun:
                # (paired pass) the generator forward above does not read this discriminator: its previous step's
                # all-reduce -- deferred, see overlap_allreduce -- has been running under it; now wait, update, begin
                self._begin(self.dis_vid, self.vid_opt)
                begun = True
            if self.pair_d:
                loss = bce_with_logits_halves(self.dis_vid.forward_pair_joint(real, fake), 1.0, 0.0)
            else:
                pf, _ = self.dis_vid(fake)
                loss = bce_with_logits_pair(pr, 1.0, pf, 0.0)
            loss.backward(gradient=unit_grad(loss.device))
            losses.append(loss.detach())
        self._opt_step(self.dis_vid, self.vid_opt, len(shards))
        if not _defer:
            self._flush(self.dis_vid)
        return self._mean(losses)

    def _can_pair(self, n_videos, n_images, images_first):
        return self.joint_g and hasattr(self.gen, "can_pair") and next(self.gen.parameters()).is_cuda and \
            self.gen.can_pair(n_videos, n_images, images_first)

    def g_step(self, B, shards=1):
        """mnist_moco_ode.py:153-163; `shards` virtual replicas of batch B each."""
        self._begin(self.gen, self.gen_opt)
        frozen = []
        if self.freeze_d:
            # the reference lets this backward also fill the discriminators' .grad, which the next zero_grad()
            # discards unused (mnist_moco_ode.py:116,134,162); skipping those weight-gradient GEMMs changes nothing
            for m in (self.dis_vid, self.dis_img):
                for p in m.parameters():
                    if p.requires_grad:
                        p.requires_grad_(False)
                        frozen.append(p)
        losses = []
        batch_adj = self.prefetch and id(self.gen) in self.arenas and hasattr(self.gen, "flush_adjoints")
        early_ar = self.world > 1 and self.overlap_ar and id(self.gen) in self.arenas
        try:
            joint = shards == 1 and self._can_pair(B, B, False)
            for _ in range(shards):
                if joint:          # both calls' rows in one decoder pass (two BatchNorm batches)
                    (fake_vid, _), (fake_img, _) = self.gen.sample_pair(B, B)
                else:
                    fake_vid, _ = self.gen.sample_videos(B)
                    fake_img, _ = self.gen.sample_images(B)
                side = self._side if shards == 1 else None
                if side is not None:
                    # D_img(fake images) -- ~40 launches on grids of a few workgroups -- runs on the side stream, under the
                    # video discriminator's GEMMs; autograd replays each node's backward on the stream of its forward, so
                    # the input-gradient half overlaps the same way.  Same kernels on the same inputs: bit-identical.
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    fake_img.record_stream(side)
                    with torch.cuda.stream(side):
                        pi, _ = self.dis_img(fake_img)
                    pv, _ = self.dis_vid(fake_vid)
                    main.wait_stream(side)
                    pi.record_stream(main)
                else:
                    pv, _ = self.dis_vid(fake_vid)
                    pi, _ = self.dis_img(fake_img)
                loss = bce_with_logits_pair(pv, 1.0, pi, 1.0)
                if batch_adj:      # the video and the image path: two latent adjoints, one launch
                    self.gen._adjoint_batch = dict(expect=2, ops=[], keep=[])
                if early_ar and shards == 1:
                    # both generator passes' decoder backward queued -> the decoder block of the arena is complete: its
                    # all-reduce goes out now, under the latent adjoint(s)
                    self.gen._decoder_passes_left = 1 if joint else 2
                    self.gen._on_decoder_grads = self._early_decoder_allreduce
                loss.backward(gradient=unit_grad(loss.device))
                self.gen._on_decoder_grads = None
                if batch_adj:
                    self.gen.flush_adjoints()
                    self.gen._adjoint_batch = None
                losses.append(loss.detach())
        finally:
            if batch_adj:
                self.gen._adjoint_batch = None
            for p in frozen:
                p.requires_grad_(True)
        self._opt_step(self.gen, self.gen_opt, shards)
        self._flush(self.gen)              # weights are current when the step returns (checkpoints, the next prefetch)
        return self._mean(losses)

    def step(self, real_imgs: Sequence, real_vids: Sequence):
        """real_imgs[i]: [B,C,H,W], real_vids[i]: [B,T,C,H,W] for i < d_iters (each may instead be a LIST of shard
        tensors: virtual replicas, see the class docstring).  Returns the three losses of the last inner pass as
        device scalars (the reference prints them every 100 iterations); with shards, their mean over the shards.
        In graph mode the returned scalars are static tensors that the next step() overwrites."""
        if self._graph_mode and self._iters >= 2 and not isinstance(real_imgs[0], (list, tuple)):
            return self._graph_step(real_imgs, real_vids)
        return self._eager_step(real_imgs, real_vids)

    def _eager_step(self, real_imgs, real_vids):
        first = _shards(real_imgs[0])
        B, nsh = first[0].shape[0], len(first)
        li = lv = None
        # (virtual replicas -- lists of shards, a parity-testing device -- repeat every call per shard: call by call there)
        # joint generator passes: the fake images of d_image_step i and the fake videos of d_video_step i come out of one
        # decoder pass (no generator update lies between the two calls), like the G step's two batches
        pair = [0] * self.d_iters
        if nsh == 1:
            for i in range(self.d_iters):
                nv, ni = _shards(real_vids[i])[0].shape[0], _shards(real_imgs[i])[0].shape[0]
                if len(_shards(real_vids[i])) == 1 and self._can_pair(nv, ni, True):
                    pair[i] = nv
        if self.prefetch and nsh == 1 and hasattr(self.gen, "prefetch_latents") and next(self.gen.parameters()).is_cuda:
            calls = []
            for i in range(self.d_iters):
                if pair[i]:
                    calls.append(("pair_iv", (pair[i], _shards(real_imgs[i])[0].shape[0])))
                    continue
                calls += [("images", x.shape[0]) for x in _shards(real_imgs[i])]
                calls += [("videos", x.shape[0]) for x in _shards(real_vids[i])]
            calls += [("pair_vi", (B, B))] if self._can_pair(B, B, False) else [("videos", B), ("images", B)]
            self.gen.prefetch_latents(calls)
        try:
            for i in range(self.d_iters):
                # (inside step() the discriminators' optimiser steps are deferred: see overlap_allreduce)
                li = self.d_image_step(real_imgs[i], _join=False, _defer=True, _pair_videos=pair[i])    # side stream
                lv = self.d_video_step(real_vids[i], _defer=True)
            if self._pending:
                # the G step reads both discriminators: their deferred updates land now (the image discriminator's on its
                # side stream, where its step ran)
                if self._side is not None:
                    with torch.cuda.stream(self._side):
                        self._flush(self.dis_img)
                else:
                    self._flush(self.dis_img)
                self._flush(self.dis_vid)
            self._join_side()                                         # the G step reads the updated image discriminator
            lg = self.g_step(B, nsh)
        except BaseException:
            self._paired_fake_vid = None
            if hasattr(self.gen, "discard_prefetched"):
                self.gen.discard_prefetched()
            raise
        self._iters += 1
        if self._freeze_gc and self._iters == 2:     # every plan and program exists now
            freeze_host_gc()
        return li, lv, lg

    # -- HIP-graph replay of the iteration ---------------------------------------------------------------------------
    def _graph_step(self, real_imgs, real_vids):
        key = (tuple(real_imgs[0].shape), tuple(real_vids[0].shape))
        if self._graph is not None and self._graph["key"] != key:
            raise RuntimeError("GanTrainer(graph=True): batch shapes changed after capture "
                               f"({self._graph['key']} -> {key}); build a new trainer for the new shapes")
        if self._graph is None:
            self._capture(real_imgs, real_vids, key)
        else:
            g = self._graph
            for dst, src in zip(g["imgs"] + g["vids"], list(real_imgs) + list(real_vids)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
            self._feed.fill_from_script()        # host draws + Adam coefficients for this iteration, in program order
            self._feed.upload()
            g["graph"].replay()
        for m in (self.gen, self.dis_vid, self.dis_img):
            m.invalidate_packs()                 # eager passes after a replay must re-pack (versions did not move)
        self._iters += 1
        return self._graph["losses"]

    def _capture(self, real_imgs, real_vids, key):
        dev = real_imgs[0].device
        for m in (self.gen, self.dis_vid, self.dis_img):
            if id(m) not in self.arenas:
                raise NotImplementedError("GanTrainer(graph=True) needs gradient arenas on all three networks "
                                          f"({type(m).__name__} has none)")
        imgs = [t.detach().clone() for t in real_imgs]      # static inputs: later iterations are copied into them
        vids = [t.detach().clone() for t in real_vids]
        # the graph must be self-contained with respect to the weights: every first-use pack of the iteration has to be
        # recorded.  Which panels are "stale" is host-side state -- an eager sample_videos() between the warm-up
        # iterations and this capture (or a load_state_dict / broadcast_state) would leave panels marked fresh, their
        # pack launches would be missing from the graph, and every replay would run that network on panels that are
        # never re-packed while Adam keeps moving the real weights.
        for m in (self.gen, self.dis_vid, self.dis_img):
            m.invalidate_packs()
        feed = self._feed = HostFeed(dev)
        self.gen._feed = feed
        for o in (self.gen_opt, self.vid_opt, self.img_opt):
            o._feed = feed
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        feed.begin()
        feed.recording = True
        try:
            with torch.cuda.graph(graph):
                losses = self._eager_step(imgs, vids)
        finally:
            feed.recording = False
        self._iters -= 1                        # _eager_step counted the capture pass; _graph_step counts it
        self._graph = dict(key=key, graph=graph, imgs=imgs, vids=vids, losses=losses)
        # nothing has executed yet: capture only recorded the launches.  The host inputs of THIS iteration were drawn
        # (and the Adam step counts advanced) while recording; upload them and run the iteration.
        feed.upload()
        graph.replay()


def train_step(trainer: GanTrainer, real_imgs, real_vids):
    """Functional spelling of GanTrainer.step for parity tests that read like oracle.mocogan_ref.train_step."""
    return trainer.step(real_imgs, real_vids)
