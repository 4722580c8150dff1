"""Host-side cost of one training iteration: wall time of issuing 20 iterations without draining vs with a final
sync, and a cProfile of the issue path (GPU box)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
B = 32
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(B, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(B, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
for _ in range(3): tr.step(imgs, vids)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): tr.step(imgs, vids)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"issue {1e3*(t1-t0)/20:.2f} ms/iter, with drain {1e3*(t2-t0)/20:.2f} ms/iter")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): tr.step(imgs, vids)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
