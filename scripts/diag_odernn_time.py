import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
_, dv, di = G.build_mnist()
gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16)
gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
B = 32
g = torch.Generator().manual_seed(1)
img = torch.rand(B, 1, 28, 28, generator=g).cuda(); vid = torch.rand(B, 16, 1, 28, 28, generator=g).cuda()
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("sample_videos", t(lambda: gen.sample_videos(B)), "sample_images", t(lambda: gen.sample_images(B)))
for k, plans in gen._pool.plans.items():
    print(k, "nsteps", plans[0].nsteps.cpu().tolist())
print("d_img", t(lambda: tr.d_image_step(img)), "d_vid", t(lambda: tr.d_video_step(vid)), "g", t(lambda: tr.g_step(B)))
for k, plans in gen._pool.plans.items():
    print(k, "nsteps", plans[0].nsteps.cpu().tolist())
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tr.g_step(B); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.g_step(B); torch.cuda.synchronize()
    print("g_step", i, (time.perf_counter() - t0) * 1e3)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.d_image_step(img); tr.d_video_step(vid); torch.cuda.synchronize()
    print("d_step", i, (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(8):
    tr.g_step(B); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
