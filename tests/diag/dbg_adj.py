import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
import gan_ode_amd._lib as L
from oracle import mocogan_ref as M
from oracle import ode_ref
from conftest import rel_err
def stream(): return torch.cuda.current_stream().cuda_stream
N, T = 20, 5
torch.manual_seed(N)
f = M.OdeRhs(16, 16); gru = torch.nn.GRUCell(16, 16)
noise = torch.randn(T + 1, N, 16)
h = [noise[0]]; hps = []
t01 = torch.tensor([0.0, 1.0])
for t in range(T):
    hp = ode_ref.odeint_adjoint(f, h[-1], t01)[-1]; hps.append(hp); h.append(gru(noise[t + 1], hp))
zref = torch.stack(h[1:], dim=1)
gup = torch.randn(N, T, 16, generator=torch.Generator().manual_seed(7))
(zref * gup).sum().backward()
ref_grads = [p.grad for p in list(f.parameters()) + [gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh]]
P = [p.detach().cuda() for p in list(f.parameters()) + [gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh]]
op = L.OdeRnnParams(*[p.data_ptr() for p in P])
nz, content = noise.cuda(), torch.randn(N, 50).cuda()
z = torch.full((N * T, 72), float("nan"), device="cuda"); hp_d = torch.empty(N, T, 16, device="cuda")
fop = L.OdeRnnFwdOp(p=op, noise=nz.data_ptr(), content=content.data_ptr(), sel_t=None, z=z.data_ptr(), hs=None, hp=hp_d.data_ptr(), nsteps=None, N=N, T=T, rtol=1e-7, atol=1e-9, zcols=72)
L.run_one(fop, stream())
gz = torch.zeros(N * T, 72, device="cuda"); gz.view(N, T, 72)[:, :, :16] = gup.cuda()
work = torch.empty(L.lib().gode_odernn_bwd_work_size(N), device="cuda")
res = {}
for substeps in (0, 32, 256):
    grads = torch.full((L.ODERNN_NPARAM,), float("nan"), device="cuda")
    bop = L.OdeRnnBwdOp(p=op, noise=nz.data_ptr(), hp=hp_d.data_ptr(), sel_t=None, gz=gz.data_ptr(), work=work.data_ptr(), grads=grads.data_ptr(), N=N, T=T, substeps=substeps, accumulate=0, zcols=72, rtol=1e-7, atol=1e-9)
    L.run_one(bop, stream()); torch.cuda.synchronize()
    res[substeps] = grads.cpu()
off = 0
for name, r in zip(("W1", "b1", "W2", "b2", "Wih", "Whh", "bih", "bhh"), ref_grads):
    n = r.numel()
    print(name, "adaptive vs oracle %.2e" % rel_err(res[0][off:off+n].view_as(r), r), " fixed32 vs oracle %.2e" % rel_err(res[32][off:off+n].view_as(r), r),
          " adaptive vs fixed256 %.2e" % rel_err(res[0][off:off+n], res[256][off:off+n]))
    off += n
