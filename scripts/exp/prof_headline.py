"""The headline loop on its own (GPU box, under rocprofv3 --kernel-trace --stats): 60 gen.sample_videos(32) steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gan_ode_amd as G
G.limit_host_threads()
torch.manual_seed(0); np.random.seed(0)
gen, _, _ = G.build_mnist()
gen.cuda()
with torch.no_grad():
    for _ in range(60):
        gen.sample_videos(32)
torch.cuda.synchronize()
