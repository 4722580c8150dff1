"""Host enqueue time of one gen.sample_videos(B) call vs its GPU time (GPU box):  python scripts/exp/host_headline.py [mnist|ucf|odernn]"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gan_ode_amd as G
cfg = sys.argv[1] if len(sys.argv) > 1 else "mnist"
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
if cfg == "ucf":
    gen, _, _ = G.build_ucf(); B = 16
elif cfg == "odernn":
    gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16); B = 32
else:
    gen, _, _ = G.build_mnist(); B = 32
gen.cuda()
with torch.no_grad():
    for _ in range(30):
        gen.sample_videos(B)
    G.freeze_host_gc()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        gen.sample_videos(B)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{cfg}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/call, total {1e3 * (t2 - t0) / K:.3f} ms/call")
    # host alone: the same calls with the GPU drained after each (no back-pressure)
    th = 0.0
    for _ in range(50):
        torch.cuda.synchronize()
        a = time.perf_counter()
        gen.sample_videos(B)
        th += time.perf_counter() - a
    print(f"   host time of one call on an idle GPU: {1e3 * th / 50:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        gen.sample_videos(B)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
