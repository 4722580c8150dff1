"""Soak (GPU box): N training iterations of a config on synthetic data; checks the losses stay finite, the allocator's
reserved memory stops growing after the first iterations, and reports iterations/s.   python scripts/soak.py [mnist|ucf|odernn] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G

cfg = sys.argv[1] if len(sys.argv) > 1 else "mnist"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
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
pool_i = [torch.rand(B, C_, HW, HW, generator=g).cuda() for _ in range(8)]
pool_v = [torch.rand(B, 16, C_, HW, HW, generator=g).cuda() for _ in range(8)]
mem = []
t0 = time.time()
for it in range(iters):
    imgs = [pool_i[(2 * it + k) % 8] for k in range(2)]
    vids = [pool_v[(2 * it + k) % 8] for k in range(2)]
    li, lv, lg = tr.step(imgs, vids)
    if it % 100 == 99 or it == iters - 1:
        torch.cuda.synchronize()
        vals = [float(li), float(lv), float(lg)]
        mem.append(torch.cuda.memory_reserved())
        print(f"it {it + 1}: losses {vals[0]:.4f} {vals[1]:.4f} {vals[2]:.4f}  reserved {mem[-1] / 2**20:.0f} MiB  {(it + 1) / (time.time() - t0):.1f} it/s", flush=True)
        assert all(np.isfinite(vals)), vals
assert len(mem) < 3 or mem[-1] <= mem[1] * 1.02, ("reserved memory keeps growing", mem)
print("soak ok")
