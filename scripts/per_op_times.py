"""GPU box: runs two MNIST training iterations (so every plan exists), then re-launches every GEMM-shaped op of every
plan on its own (HIP events, 20 reps) and prints time, algorithmic FLOPs and TFLOP/s per op -- the table used to
find layers whose tile / split choice leaves the CUs unbalanced.
    python scripts/per_op_times.py [mnist|ucf]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
import gan_ode_amd._lib as L
from gan_ode_amd.engine import stream_ptr

cfg = sys.argv[1] if len(sys.argv) > 1 else "mnist"
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = (G.build_mnist() if cfg == "mnist" else G.build_ucf())
gen.cuda(); dv.cuda(); di.cuda()
B, T, Cc, HW = (32, 16, 1, 28) if cfg == "mnist" else (16, 16, 3, 64)
tr = G.GanTrainer(gen, dv, di)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(B, Cc, HW, HW, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(B, T, Cc, HW, HW, generator=g).cuda() for _ in range(2)]
for _ in range(2):
    tr.step(imgs, vids)
torch.cuda.synchronize()


def timeit(op, reps=20):
    st = stream_ptr()
    for _ in range(2):
        L.run_one(op, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.run_one(op, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def flops(op):
    g = op.g
    return 2.0 * g.N * g.Do * g.Ho * g.Wo * g.Co * g.Ci * g.kd * g.kh * g.kw


rows = []
seen = set()
for name, net in (("gen", gen), ("vidD", dv), ("imgD", di)):
    for key, plans in net._pool.plans.items():
        for plan in plans[:1]:
            stack = getattr(plan, "stack", plan)
            progs = [("fwd%d" % int(t), p[0]) for t, p in stack._fwd.items()]
            for bk, bp in stack._bwd.items():
                progs.append(("bwd", bp[0]))
            for pname, prog in progs:
                for op in prog.ops:
                    if not isinstance(op, (L.IgemmOp, L.WgradOp)):
                        continue
                    kind = "wgrad" if isinstance(op, L.WgradOp) else ("fprop" if op.dir == L.FPROP else "dgrad")
                    sig = (kind, op.g.key(), bool(op.scale))
                    if sig in seen:
                        continue
                    seen.add(sig)
                    ms = timeit(op)
                    gg = op.g
                    rows.append((ms, f"{name:5s} {str(key):22s} {pname:5s} {kind:6s} Ci={gg.Ci:4d} Co={gg.Co:4d} "
                                     f"in={gg.Di}x{gg.Hi}x{gg.Wi} out={gg.Do}x{gg.Ho}x{gg.Wo} k={gg.kd}{gg.kh}{gg.kw} "
                                     f"N={gg.N:4d} xf={int(bool(op.scale))}", flops(op)))
rows.sort(key=lambda r: -r[0])
for ms, desc, fl in rows:
    print(f"{ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF  {fl/1e9:7.2f} GF  {desc}")
