"""Per-op table of one training iteration (L.TRACE: every libgode op between two stream events): kind, geometry, us,
TFLOP/s.  python scripts/iter_ops.py [mnist|ucf|odernn] > gpurun_out/iter_ops.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
import gan_ode_amd._lib as L
from bench import _op_class_and_flop

cfg = sys.argv[1] if len(sys.argv) > 1 else "mnist"
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
if cfg == "ucf":
    gen, dv, di = G.build_ucf(); B, C_, HW = 16, 3, 64
elif cfg == "odernn":
    _, dv, di = G.build_mnist(); gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16); B, C_, HW = 32, 1, 28
else:
    gen, dv, di = G.build_mnist(); B, C_, HW = 32, 1, 28
gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(B, C_, HW, HW, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(B, 16, C_, HW, HW, generator=g).cuda() for _ in range(2)]
for _ in range(5):
    tr.step(imgs, vids)
torch.cuda.synchronize()
acc = {}
reps = 5
for rep in range(reps):
    L.TRACE = []
    tr.step(imgs, vids)
    torch.cuda.synchronize()
    trace, L.TRACE = L.TRACE, None
    for i, (op, e0, e1) in enumerate(trace):
        acc.setdefault(i, [op, 0.0])[1] += e0.elapsed_time(e1) * 1e3 / reps
rows = []
names = {1: "igemm", 2: "wgrad", 3: "bn_fin", 4: "bn_bwd", 5: "ode_fwd", 6: "ode_bwd", 7: "bce", 8: "adam", 9: "pack", 10: "odernn_fwd", 11: "odernn_bwd", 12: "bn_apply"}
tot = 0.0
for i in sorted(acc):
    op, us = acc[i]
    cls, fl = _op_class_and_flop(op)
    if isinstance(op, str):
        desc = op
    else:
        desc = names.get(op.KIND, str(op.KIND))
        if hasattr(op, "g") and hasattr(op.g, "N"):
            g_ = op.g
            desc += f" dir={getattr(op, 'dir', '-')} N={g_.N} Ci={g_.Ci} Co={g_.Co} in={g_.Di}x{g_.Hi}x{g_.Wi} out={g_.Do}x{g_.Ho}x{g_.Wo} k={g_.kd}{g_.kh}{g_.kw} s={g_.sd}{g_.sh}{g_.sw}"
            if op.KIND == 1:
                desc += f" xf={'y' if op.scale else 'n'} act={op.act} stats={'y' if op.stats else 'n'}"
        elif op.KIND in (3, 4, 12):
            desc += f" C={op.C} M={getattr(op, 'M', getattr(op, 'count', 0))}"
        elif op.KIND in (5, 6, 10, 11):
            desc += f" N={op.N} T={op.T}"
    tot += us
    rows.append((i, us, fl, desc))
print(f"# {cfg}: {len(rows)} ops, traced total {tot/1e3:.3f} ms")
for i, us, fl, desc in rows:
    print(f"{i:4d} {us:8.1f} us {fl/1e9:8.2f} GF {fl/us/1e6 if fl else 0:7.1f} TF  {desc}")
print("\n# by time")
for i, us, fl, desc in sorted(rows, key=lambda r: -r[1])[:60]:
    print(f"{i:4d} {us:8.1f} us {fl/1e9:8.2f} GF {fl/us/1e6 if fl else 0:7.1f} TF  {desc}")
