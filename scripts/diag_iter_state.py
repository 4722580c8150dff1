"""Why does bench.py's iteration time differ from a bare loop of tr.step?  Replays bench's phases and times the
iteration after each (GPU box).  Finding: the time per iteration depends on WHICH training iterations are timed --
iterations ~45-85 of a run on synthetic data take 15.3-18.4 ms, the ones before and after 12.3-12.5 ms, with the same
kernels and launch counts (data-dependent clocks / operand values, not host state: it does not follow queue depth,
plan counts, allocator state or a sleep)."""
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


def timed(fn, k, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / k, 1e3 * (t2 - t0) / k


order = sys.argv[1] if len(sys.argv) > 1 else "sample,iter,d,iter,g,iter"
K = int(os.environ.get("K", "20"))
tr = None
for ph in order.split(","):
    if ph == "sample":
        def f():
            with torch.no_grad(): gen.sample_videos(B)
        print("sample   issue/drain ms", timed(f, 100), flush=True)
    else:
        if tr is None:
            tr = G.GanTrainer(gen, dv, di)
        if ph == "iter": print("iter     issue/drain ms", timed(lambda: tr.step(imgs, vids), K), flush=True)
        if ph == "sleep":
            time.sleep(3.0); print("slept 3 s", flush=True)
        if ph == "prof":
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
            t0 = time.perf_counter()
            for _ in range(K): tr.step(imgs, vids)
            torch.cuda.synchronize(); pr.disable()
            print("prof     ms/iter", 1e3 * (time.perf_counter() - t0) / K)
            pstats.Stats(pr).sort_stats("tottime").print_stats(10)
        if ph == "d": print("d        issue/drain ms", timed(lambda: (tr.d_image_step(imgs[0]), tr.d_video_step(vids[0])), K), flush=True)
        if ph == "g": print("g        issue/drain ms", timed(lambda: tr.g_step(B), K), flush=True)
    for name, net in (("gen", gen), ("vidD", dv), ("imgD", di)):
        print("   ", name, {k: len(v) for k, v in net._pool.plans.items()}, flush=True)
