"""Host-side cost of gen.sample_videos(32): cProfile over 300 calls (GPU box)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
torch.manual_seed(0); np.random.seed(0)
gen, _, _ = G.build_mnist(); gen.cuda()
def run(n):
    with torch.no_grad():
        for _ in range(n):
            gen.sample_videos(32)
run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); run(300); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time/call {(t1 - t0) / 300 * 1e3:.3f} ms, incl. drain {(t2 - t0) / 300 * 1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable(); run(300); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
