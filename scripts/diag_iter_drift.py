"""Per-iteration wall time of 120 consecutive training iterations (sync after each), to expose drift / periodic stalls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
B = 32
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(B, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(B, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
tr = G.GanTrainer(gen, dv, di)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(imgs, vids)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    if i % 10 == 9:
        print(i + 1, " ".join(f"{t:5.1f}" for t in ts[-10:]), f" alloc={torch.cuda.memory_allocated()/1e9:.2f}GB reserved={torch.cuda.memory_reserved()/1e9:.2f}GB", flush=True)
