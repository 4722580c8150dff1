"""Runs a few sample_videos(32) calls (and optionally train iterations) for rocprofv3 --kernel-trace; the companion
summarise_trace() prints the per-kernel timeline of the LAST call with the idle gaps between kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()

mode = sys.argv[1] if len(sys.argv) > 1 else "sample"
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
B = 32
if mode == "sample":
    for _ in range(12):
        with torch.no_grad():
            gen.sample_videos(B)
    torch.cuda.synchronize()
else:
    tr = G.GanTrainer(gen, dv, di)
    g = torch.Generator().manual_seed(1)
    imgs = [torch.rand(B, 1, 28, 28, generator=g).cuda() for _ in range(2)]
    vids = [torch.rand(B, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
    for _ in range(4):
        tr.step(imgs, vids)
    torch.cuda.synchronize()
