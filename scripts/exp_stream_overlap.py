"""Do two HIP streams overlap on this box?  A: chain of large fp32 GEMM-like libgode launches (decoder layer), B: chain of
tiny kernels.  Prints serial sum vs concurrent wall."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
import gan_ode_amd._lib as L
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
with torch.no_grad():
    gen.sample_videos(32)
plan = gen._pool.plans[(32, 16, False)][0]
prog, _ = plan.stack._fwd[True]
big = [op for op in prog.ops if isinstance(op, L.IgemmOp)][1:4]
x = torch.rand(32, 1, 28, 28).cuda()
with torch.no_grad():
    di(x)
dplan = di._pool.plans[tuple(x.shape)][0]
small = dplan._fwd[True][0]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run_big(n):
    with torch.cuda.stream(sA):
        for _ in range(n):
            for op in big: L.run_one(op, sA.cuda_stream)
def run_small(n):
    with torch.cuda.stream(sB):
        for _ in range(n): small.run(sB.cuda_stream)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(2):
    run_big(3); run_small(10)
torch.cuda.synchronize()
NB, NS = 5, 40
tb = timed(lambda: run_big(NB)); ts = timed(lambda: run_small(NS))
def both():
    # interleave enqueues so that both queues are fed
    for i in range(NB):
        run_big(1); run_small(NS // NB)
tc = timed(both)
print(f"big alone {tb:.3f} ms ({NB * 3} launches), small alone {ts:.3f} ms ({NS} x {len(small.ops)} ops), concurrent {tc:.3f} ms, sum {tb + ts:.3f}")
