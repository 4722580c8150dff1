import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gan_ode_amd as G
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_ucf(); gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
B = 16
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(B, 3, 64, 64, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(B, 16, 3, 64, 64, generator=g).cuda() for _ in range(2)]
for _ in range(4):
    tr.step(imgs, vids)
torch.cuda.synchronize()
