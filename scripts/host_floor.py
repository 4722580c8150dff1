"""Host-side floor of one training iteration: the same iteration at ngf=ndf=8 (GPU kernels are tiny, so wall time ~
host time), batch 32 so that the host RNG draws have BASELINE's sizes.  python scripts/host_floor.py"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(ngf=8, ndf=8); gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di, overlap_image_d=(os.environ.get("OVERLAP", "1") == "1"))
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(32, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(32, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
for _ in range(5): tr.step(imgs, vids)
G.freeze_host_gc()
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K): tr.step(imgs, vids)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"tiny nets: host issue {1e3 * (t1 - t0) / K:.3f} ms/iter, total {1e3 * (t2 - t0) / K:.3f} ms/iter")
for name, fn in (("d_img", lambda: tr.d_image_step(imgs[0])), ("d_vid", lambda: tr.d_video_step(vids[0])), ("g", lambda: tr.g_step(32)),
                 ("sample_videos", lambda: gen.sample_videos(32)), ("sample_images", lambda: gen.sample_images(32)),
                 ("draw_images", lambda: gen._draw(1024, 16)), ("draw_videos", lambda: gen._draw(32, 16))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad() if name.startswith("sample") else torch.enable_grad():
        for _ in range(K): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  {name}: host {1e3 * (t1 - t0) / K:.3f} ms, total {1e3 * (t2 - t0) / K:.3f} ms")
