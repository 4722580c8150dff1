"""Host-side enqueue time of a training iteration vs its GPU time: python scripts/host_time.py [overlap 0|1]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
ov = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di, overlap_image_d=ov)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(32, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(32, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
for _ in range(5): tr.step(imgs, vids)
G.freeze_host_gc()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): tr.step(imgs, vids)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"overlap={ov}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/iter, total {1e3 * (t2 - t0) / K:.3f} ms/iter")
# host-only cost: the same loop while the GPU is idle-ish is not separable; report per-phase host time instead
for name, fn in (("d_img", lambda: tr.d_image_step(imgs[0])), ("d_vid", lambda: tr.d_video_step(vids[0])), ("g", lambda: tr.g_step(32))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  {name}: host {1e3 * (t1 - t0) / K:.3f} ms, total {1e3 * (t2 - t0) / K:.3f} ms")
