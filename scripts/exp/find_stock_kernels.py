"""Which torch (non-libgode) GPU kernels does one training iteration launch?  torch profiler, aten ops with device time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gan_ode_amd as G
from torch.profiler import profile, ProfilerActivity
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(32, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(32, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
for _ in range(3): tr.step(imgs, vids)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(imgs, vids)
    torch.cuda.synchronize()
for e in prof.key_averages(group_by_stack_n=6):
    if e.key.startswith("aten::") and ("fill" in e.key or "zero" in e.key or "ones" in e.key or "add" in e.key or "mul" in e.key or "copy" in e.key or "stack" in e.key or "mean" in e.key):
        print(e.key, e.count, "dev_us", getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)))
        for fr in e.stack[:6]: print("     ", fr)
