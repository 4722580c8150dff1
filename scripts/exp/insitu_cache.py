"""Why do the decoder GEMMs run 8-10 % slower inside an iteration than back to back?  Times decoder layer 3 (ConvT 128->64)
(a) repeated as is, (b) after its input was re-written by bn_apply (as in the real sequence), (c) after a 1 GiB fill
(caches cold), (d) after a small unrelated kernel chain.   python scripts/exp/insitu_cache.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gan_ode_amd as G
import gan_ode_amd._lib as L
from gan_ode_amd.engine import stream_ptr
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda()
with torch.no_grad():
    gen.sample_videos(32)
plan = gen._pool.plans[(32, 16, False)][0]
prog, _ = plan.stack._fwd[True]
ops = prog.ops
igemms = [i for i, op in enumerate(ops) if isinstance(op, L.IgemmOp)]
big = torch.empty(1 << 28, device="cuda")       # 1 GiB
def time_after(pre, op, reps=20):
    st = stream_ptr(); ts = []
    for _ in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); L.run_one(op, st); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for li in (1, 2, 3):
    op = ops[igemms[li]]
    # the bn_apply that produces this layer's input is the op just before it
    prev_apply = ops[igemms[li] - 1]
    assert isinstance(prev_apply, L.BnApplyOp), type(prev_apply)
    st = stream_ptr()
    a = time_after(lambda: None, op)
    b = time_after(lambda: L.run_one(prev_apply, st), op)
    c = time_after(lambda: big.fill_(1.0), op)
    d = time_after(lambda: [L.run_one(ops[igemms[0] + 1], st) for _ in range(3)], op)
    print(f"decoder layer {li}: repeated {a:.1f} us | after bn_apply of its input {b:.1f} | after 1 GiB fill {c:.1f} | after 3 small kernels {d:.1f}")
