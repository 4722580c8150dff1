import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gan_ode_amd as G
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(32, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(32, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
for _ in range(3): tr.step(imgs, vids)
torch.cuda.synchronize()
from torch.utils._python_dispatch import TorchDispatchMode
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("zeros", "zero_", "fill_", "ones")) and isinstance(out, torch.Tensor) and out.is_cuda:
            print("GPU", name, tuple(out.shape)); traceback.print_stack(limit=8)
        return out
with Spy():
    tr.step(imgs, vids)
torch.cuda.synchronize()
