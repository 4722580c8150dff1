import sys, os
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
import gan_ode_amd._lib as L
def st(): return torch.cuda.current_stream().cuda_stream
for (M, Cc, act, actn) in [(131072, 128, L.ACT_RELU, "relu"), (8192, 128, L.ACT_RELU, "relu"), (524288, 64, L.ACT_RELU, "relu"), (32768, 256, L.ACT_RELU, "relu"), (8192,512,L.ACT_RELU,"relu")]:
    g = torch.Generator().manual_seed(1)
    y = (torch.randn(M, Cc, generator=g) * 2 + 1).double()
    ga = torch.randn(M, Cc, generator=g).double()
    bn = torch.nn.BatchNorm1d(Cc).double()
    yr = y.clone().requires_grad_(True)
    out = F.relu(bn(yr)); out.backward(ga)
    mean = y.mean(0); var = y.var(0, unbiased=False); invstd = 1 / torch.sqrt(var + 1e-5)
    d = dict(mean=mean.float().cuda(), invstd=invstd.float().cuda(), scale=invstd.float().cuda(), shift=(-mean * invstd).float().cuda())
    gam = torch.ones(Cc).cuda()
    yd, gd = y.float().cuda(), ga.float().cuda()
    dg, db = torch.empty(Cc).cuda(), torch.empty(Cc).cuda()
    work = torch.empty(L.lib().gode_bn_bwd_work_size(M, Cc), device="cuda")
    op = L.BnBwdOp(g=gd.data_ptr(), y=yd.data_ptr(), M=M, C=Cc, act=act, gamma=gam.data_ptr(), mean=d["mean"].data_ptr(), invstd=d["invstd"].data_ptr(), scale=d["scale"].data_ptr(), shift=d["shift"].data_ptr(), dgamma=dg.data_ptr(), dbeta=db.data_ptr(), work=work.data_ptr())
    L.run_one(op, st()); torch.cuda.synchronize()
    rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(M, Cc, "dgamma", rl2(dg.cpu(), bn.weight.grad), "dbeta", rl2(db.cpu(), bn.bias.grad), "g_y", rl2(gd.cpu(), yr.grad))
