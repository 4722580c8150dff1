"""Soak: N training iterations (MNIST config, batch 32) -- losses stay finite, device memory stays flat, no plan leak."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gan_ode_amd as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist()
gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(32, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(32, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
mem0 = None
t0 = time.time()
for i in range(n):
    losses = [float(v) for v in tr.step(imgs, vids)]
    assert all(np.isfinite(losses)), (i, losses)
    if i == 20:
        torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated()
    if i % 50 == 0:
        print(i, [round(v, 4) for v in losses], torch.cuda.memory_allocated() >> 20, "MiB", flush=True)
torch.cuda.synchronize()
mem1 = torch.cuda.memory_allocated()
print("done", n, "iterations in", round(time.time() - t0, 1), "s; memory", mem0 >> 20, "->", mem1 >> 20, "MiB")
assert mem1 <= mem0 + (8 << 20), (mem0, mem1)
with torch.no_grad():
    v, _ = gen.sample_videos(4)
assert torch.isfinite(v).all()
