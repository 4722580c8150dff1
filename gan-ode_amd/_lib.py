"""ctypes binding of libgode.so (include/gode.h).  There is NO fallback: if the library is missing or a call
fails, a RuntimeError is raised -- the product path never runs on anything but the HIP kernels."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgode.so")

i32, i64, f32, ptr = C.c_int32, C.c_int64, C.c_float, C.c_void_p

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH_OUT = 0, 1, 2, 3
EPI_RAW, EPI_TANH = 0, 1
FPROP, DGRAD = 0, 1
OP_IGEMM, OP_WGRAD, OP_BN_FINALIZE, OP_BN_BWD, OP_ODE_FWD, OP_ODE_BWD, OP_BCE, OP_ADAM, OP_PACK = range(1, 10)
OP_ODERNN_FWD, OP_ODERNN_BWD, OP_BN_APPLY, OP_COL2IM = 10, 11, 12, 13
ODE_NPARAM = 2672
ODERNN_NPARAM = 2176


class ConvGeom(C.Structure):
    _fields_ = [(n, i32) for n in "N Ci Co Di Hi Wi Do Ho Wo kd kh kw sd sh sw pd ph pw".split()]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class IgemmOp(C.Structure):
    _fields_ = [("g", ConvGeom), ("dir", i32), ("act", i32), ("epilogue", i32), ("tile", i32), ("src", ptr),
                ("wpack", ptr), ("out", ptr), ("scale", ptr), ("shift", ptr), ("stats", ptr), ("gs", i64 * 5),
                ("work", ptr), ("groups", i32), ("pad3_", i32)]
    KIND = OP_IGEMM


class WgradOp(C.Structure):
    _fields_ = [("g", ConvGeom), ("act", i32), ("xform_on_y", i32), ("splits", i32), ("accumulate", i32),
                ("x", ptr), ("y", ptr), ("scale", ptr), ("shift", ptr), ("work", ptr), ("dw", ptr),
                ("co_perm", ptr), ("co_canon", i64), ("xs", i64 * 5)]
    KIND = OP_WGRAD


class BnFinalizeOp(C.Structure):
    _fields_ = [("stats", ptr), ("rows", i32), ("ncols", i32), ("C", i32), ("count", i64), ("gamma", ptr),
                ("beta", ptr), ("running_mean", ptr), ("running_var", ptr), ("num_batches_tracked", ptr),
                ("mean", ptr), ("invstd", ptr), ("scale", ptr), ("shift", ptr), ("momentum", f32), ("eps", f32),
                ("training", i32), ("pad_", i32), ("groups", i32), ("rows0", i32), ("nseg", i32), ("order", i32),
                ("count1", i64), ("seg", i32 * 24), ("stats1", ptr), ("rows1", i32), ("pad3_", i32)]
    KIND = OP_BN_FINALIZE


class BnBwdOp(C.Structure):
    _fields_ = [("g", ptr), ("y", ptr), ("M", i64), ("C", i32), ("act", i32), ("gamma", ptr), ("mean", ptr),
                ("invstd", ptr), ("scale", ptr), ("shift", ptr), ("dgamma", ptr), ("dbeta", ptr), ("work", ptr),
                ("accumulate", i32), ("eval_mode", i32), ("gin", ptr), ("groups", i32), ("pad2_", i32), ("M0", i64),
                ("r1_s", ptr), ("r1_w", ptr), ("r1_H", i32), ("r1_W", i32), ("r1_h", i32), ("r1_wd", i32), ("r1_off", i32),
                ("pad3_", i32)]
    KIND = OP_BN_BWD


class OdeParams(C.Structure):
    _fields_ = [(n, ptr) for n in "Wa ba Wb bb W1 b1 W2 b2".split()]


class OdeFwdOp(C.Structure):
    _fields_ = [("p", OdeParams), ("x", ptr), ("content", ptr), ("dt", ptr), ("sel_t", ptr), ("z", ptr),
                ("traj", ptr), ("N", i32), ("T", i32), ("substeps", i32), ("prenet", i32), ("zcols", i32), ("G", i32),
                ("grid_dt", ptr), ("emit_at", ptr), ("emit_w", ptr), ("method", i32), ("pad2_", i32), ("rtol", f32),
                ("atol", f32), ("tout", ptr), ("nsteps", ptr), ("sync", ptr)]
    KIND = OP_ODE_FWD


class OdeBwdOp(C.Structure):
    _fields_ = [("p", OdeParams), ("x", ptr), ("traj", ptr), ("dt", ptr), ("sel_t", ptr), ("gz", ptr),
                ("work", ptr), ("grads", ptr), ("N", i32), ("T", i32), ("substeps", i32), ("prenet", i32),
                ("accumulate", i32), ("zcols", i32), ("bstep_off", ptr), ("bstep_dt", ptr), ("method", i32),
                ("pad_", i32), ("rtol", f32), ("atol", f32), ("tout", ptr), ("nsteps", ptr), ("sync", ptr)]
    KIND = OP_ODE_BWD


class OdeRnnParams(C.Structure):
    _fields_ = [(n, ptr) for n in "W1 b1 W2 b2 Wih Whh bih bhh".split()]


class OdeRnnFwdOp(C.Structure):
    _fields_ = [("p", OdeRnnParams), ("noise", ptr), ("content", ptr), ("sel_t", ptr), ("z", ptr), ("hs", ptr),
                ("hp", ptr), ("nsteps", ptr), ("N", i32), ("T", i32), ("rtol", f32), ("atol", f32), ("zcols", i32),
                ("pad_", i32), ("sync", ptr)]
    KIND = OP_ODERNN_FWD


class OdeRnnBwdOp(C.Structure):
    _fields_ = [("p", OdeRnnParams), ("noise", ptr), ("hp", ptr), ("sel_t", ptr), ("gz", ptr), ("work", ptr),
                ("grads", ptr), ("N", i32), ("T", i32), ("substeps", i32), ("accumulate", i32), ("zcols", i32),
                ("pad_", i32), ("rtol", f32), ("atol", f32), ("sync", ptr), ("nsteps", ptr)]
    KIND = OP_ODERNN_BWD


class BnApplyOp(C.Structure):
    _fields_ = [("y", ptr), ("out", ptr), ("scale", ptr), ("shift", ptr), ("M", i64), ("C", i32), ("act", i32),
                ("M0", i64)]
    KIND = OP_BN_APPLY


class Col2imOp(C.Structure):
    _fields_ = [("cols", ptr), ("out", ptr)] + [(n, i32) for n in "N Hi Wi Ho Wo C kh kw sh sw ph pw epilogue pad_ Di Do kd sd pd pad2_".split()]
    KIND = OP_COL2IM


class BceOp(C.Structure):
    _fields_ = [("logits", ptr), ("grad", ptr), ("loss", ptr), ("n", i64), ("target", f32), ("gscale", f32),
                ("accumulate", i32), ("pad_", i32)]
    KIND = OP_BCE


class AdamOp(C.Structure):
    _fields_ = [("p", ptr), ("g", ptr), ("m", ptr), ("v", ptr), ("n", i64), ("lr", f32), ("beta1", f32),
                ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("gscale", f32), ("step", i32), ("pad_", i32)]
    KIND = OP_ADAM


class PackOp(C.Structure):
    _fields_ = [("g", ConvGeom), ("dir", i32), ("co_canon", i32), ("w", ptr), ("wpack", ptr), ("co_perm", ptr)]
    KIND = OP_PACK


_STRUCTS = {0: ConvGeom, OP_IGEMM: IgemmOp, OP_WGRAD: WgradOp, OP_BN_FINALIZE: BnFinalizeOp, OP_BN_BWD: BnBwdOp,
            OP_ODE_FWD: OdeFwdOp, OP_ODE_BWD: OdeBwdOp, OP_BCE: BceOp, OP_ADAM: AdamOp, OP_PACK: PackOp,
            OP_ODERNN_FWD: OdeRnnFwdOp, OP_ODERNN_BWD: OdeRnnBwdOp, OP_BN_APPLY: BnApplyOp, OP_COL2IM: Col2imOp}

EXPORTS = ["gode_igemm", "gode_igemm_stats_rows", "gode_igemm_stats_rows0", "gode_igemm_stats_segments", "gode_igemm_model_cycles", "gode_igemm_work_size", "gode_pack_size", "gode_pack_weights", "gode_wgrad",
           "gode_wgrad_work_size", "gode_wgrad_auto_splits", "gode_bn_finalize", "gode_bn_bwd",
           "gode_bn_bwd_work_size", "gode_bn_apply", "gode_col2im", "gode_ode_fwd", "gode_ode_bwd", "gode_ode_fwd_multi", "gode_ode_bwd_multi", "gode_ode_bwd_work_size", "gode_odernn_fwd",
           "gode_odernn_bwd", "gode_odernn_bwd_work_size", "gode_odernn_sync_size", "gode_odernn_fwd_multi",
           "gode_odernn_bwd_multi", "gode_bce_logits",
           "gode_adam_l2", "gode_adam_multi", "gode_adam_multi_dev", "gode_scale", "gode_run", "gode_version", "gode_sizeof"]

_lib = None


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is missing or inconsistent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                           "(hipcc --offload-arch=gfx950); there is no CPU or PyTorch fallback")
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"libgode.so does not export {name}")
    L.gode_sizeof.argtypes = [C.c_int]
    for kind, st in _STRUCTS.items():
        if L.gode_sizeof(kind) != C.sizeof(st):
            raise RuntimeError(f"ABI mismatch for op kind {kind}: C {L.gode_sizeof(kind)} != ctypes {C.sizeof(st)}")
    for name in ("gode_igemm", "gode_wgrad", "gode_bn_finalize", "gode_bn_bwd", "gode_ode_fwd", "gode_ode_bwd",
                 "gode_bce_logits", "gode_adam_l2", "gode_odernn_fwd", "gode_odernn_bwd", "gode_bn_apply", "gode_col2im"):
        getattr(L, name).argtypes = [ptr, ptr]
        getattr(L, name).restype = C.c_int
    L.gode_igemm_stats_rows.argtypes = [ptr]
    L.gode_igemm_stats_rows0.argtypes = [ptr]
    L.gode_igemm_stats_segments.argtypes = [ptr, i32, ptr]
    L.gode_igemm_model_cycles.argtypes = [ptr]
    L.gode_igemm_model_cycles.restype = C.c_double
    L.gode_igemm_work_size.argtypes = [ptr]
    L.gode_igemm_work_size.restype = i64
    L.gode_pack_size.argtypes = [ptr, C.c_int]
    L.gode_pack_size.restype = i64
    L.gode_pack_weights.argtypes = [ptr, C.c_int, ptr, ptr, ptr, i32, ptr]
    L.gode_wgrad_work_size.argtypes = [ptr]
    L.gode_wgrad_work_size.restype = i64
    L.gode_wgrad_auto_splits.argtypes = [ptr]
    L.gode_bn_bwd_work_size.argtypes = [i64, i32]
    L.gode_bn_bwd_work_size.restype = i64
    L.gode_ode_bwd_work_size.argtypes = [i32]
    L.gode_ode_bwd_work_size.restype = i64
    L.gode_odernn_bwd_work_size.argtypes = [i32]
    L.gode_odernn_bwd_work_size.restype = i64
    L.gode_odernn_sync_size.argtypes = [i32]
    L.gode_odernn_sync_size.restype = i64
    L.gode_ode_fwd_multi.argtypes = [ptr, i32, ptr]
    L.gode_ode_bwd_multi.argtypes = [ptr, i32, ptr]
    L.gode_odernn_fwd_multi.argtypes = [ptr, i32, ptr]
    L.gode_odernn_bwd_multi.argtypes = [ptr, i32, ptr]
    L.gode_scale.argtypes = [ptr, ptr, i64, f32, C.c_int, ptr]
    L.gode_adam_multi.argtypes = [ptr, i32, i64, f32, f32, f32, f32, f32, f32, i32, ptr]
    L.gode_adam_multi_dev.argtypes = [ptr, i32, i64, f32, f32, f32, f32, f32, ptr, ptr]
    L.gode_run.argtypes = [ptr, ptr, i32, ptr]
    _lib = L
    return L


def check(rc, what="gode call"):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (hipError)" if rc > 0 else " (argument/shape error)"))


# Measurement hook (bench.py's iteration roofline / per-class time split): when TRACE is a list, every op is launched
# on its own between two stream events and (op, start_event, end_event) is appended.  None (default): no overhead
# beyond one comparison per program.
TRACE = None


def _traced(op, stream, launch):
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    TRACE.append((op, e0, e1))


def call(tag, launch):
    """Entry points that take no op struct (gode_adam_multi, gode_scale): `tag` stands in for the op in TRACE."""
    if TRACE is not None:
        return _traced(tag, None, launch)
    launch()


class Program:
    """A fixed list of op structs executed by one gode_run call.  The structs stay alive (and patchable) here."""

    def __init__(self, ops):
        self.ops = list(ops)
        n = len(self.ops)
        self.kinds = (i32 * n)(*[op.KIND for op in self.ops])
        self.ptrs = (ptr * n)(*[C.addressof(op) for op in self.ops])
        self.n = n

    def run(self, stream):
        if TRACE is not None:
            for op in self.ops:
                run_one(op, stream)
            return
        check(lib().gode_run(self.kinds, self.ptrs, self.n, stream), "gode_run")


def run_one(op, stream):
    if op.KIND == OP_PACK:
        launch = lambda: check(lib().gode_pack_weights(C.byref(op.g), op.dir, op.w, op.wpack, op.co_perm, op.co_canon,  # noqa: E731
                                                       stream), "gode_pack_weights")
        return _traced(op, stream, launch) if TRACE is not None else launch()
    if TRACE is not None:
        return _traced(op, stream, lambda: _run_one(op, stream))
    _run_one(op, stream)


def _run_one(op, stream):
    fn = {OP_IGEMM: "gode_igemm", OP_WGRAD: "gode_wgrad", OP_BN_FINALIZE: "gode_bn_finalize", OP_BN_BWD: "gode_bn_bwd",
          OP_ODE_FWD: "gode_ode_fwd", OP_ODE_BWD: "gode_ode_bwd", OP_BCE: "gode_bce_logits",
          OP_ADAM: "gode_adam_l2", OP_ODERNN_FWD: "gode_odernn_fwd", OP_ODERNN_BWD: "gode_odernn_bwd",
          OP_BN_APPLY: "gode_bn_apply", OP_COL2IM: "gode_col2im"}[op.KIND]
    check(getattr(lib(), fn)(C.byref(op), stream), fn)
