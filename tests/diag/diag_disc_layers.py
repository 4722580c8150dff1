"""Diagnostic (GPU box): layer-by-layer forward raw outputs and backward raw-output gradients of the video
discriminator (full width, batch 32) for the HIP path and the fp32 CPU oracle, both against the fp64 oracle."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gan_ode_amd as G
from oracle import mocogan_ref as M


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).abs().max() / b.abs().max())


torch.manual_seed(3)
dv = G.VideoDiscriminator(1, ksize=2)
o32 = M.VideoDisc(1, ksize=2); o32.load_state_dict(dv.state_dict())
o64 = copy.deepcopy(o32).double()
dv.cuda()
x = torch.rand(32, 1, 16, 28, 28) * 2 - 1


def run_oracle(o, xin):
    for m in o.main:
        if isinstance(m, torch.nn.LeakyReLU):
            m.inplace = False
    convs = [m for m in o.main if isinstance(m, torch.nn.Conv3d)]
    fw, bw = {}, {}
    hs = []
    for i, c in enumerate(convs):
        hs.append(c.register_forward_hook(lambda m, a, out, i=i: fw.__setitem__(i, out.detach().clone())))
        hs.append(c.register_full_backward_hook(lambda m, gi, go, i=i: bw.__setitem__(i, go[0].detach().clone())))
    out, _ = o(xin)
    loss = torch.nn.BCEWithLogitsLoss()(out, torch.ones_like(out))
    loss.backward()
    for h in hs:
        h.remove()
    return fw, bw


f32, b32 = run_oracle(o32, x.clone())
f64, b64 = run_oracle(o64, x.double())
xd = x.cuda().requires_grad_(True)
out, _ = dv(xd)
loss = G.bce_with_logits_const(out, 1.0)
plan = dv._pool.plans[tuple(xd.shape)][0]
loss.backward()
torch.cuda.synchronize()
cl = lambda t: t.permute(0, 2, 3, 4, 1)
for l in range(5):
    mine_y = (plan.y[l] if l < 4 else plan.out).cpu()
    mine_g = plan.g[l].cpu()
    print(f"L{l} fwd raw: hip-vs-f64 {rel(mine_y, cl(f64[l])):.2e} cpu32-vs-f64 {rel(f32[l], f64[l]):.2e} | "
          f"bwd g_raw: hip-vs-f64 {rel(mine_g, cl(b64[l])):.2e} cpu32-vs-f64 {rel(b32[l], b64[l]):.2e}")
    if l in (1, 2, 3):
        # sign agreement of the BatchNorm output z (the LeakyReLU kink) with the fp64 oracle
        z_mine = plan.y[l].cpu().double() * plan.scale[l].cpu().double() + plan.shift[l].cpu().double()
        bn = [m for m in o64.main if isinstance(m, torch.nn.BatchNorm3d)][l - 1]
        y64 = f64[l]
        mean = y64.mean(dim=(0, 2, 3, 4), keepdim=True); var = y64.var(dim=(0, 2, 3, 4), unbiased=False, keepdim=True)
        z64 = cl((y64 - mean) / torch.sqrt(var + 1e-5) * bn.weight.view(1, -1, 1, 1, 1) + bn.bias.view(1, -1, 1, 1, 1))
        y32 = f32[l].double()
        m32 = f32[l].mean(dim=(0, 2, 3, 4), keepdim=True); v32 = f32[l].var(dim=(0, 2, 3, 4), unbiased=False, keepdim=True)
        z32 = cl((f32[l] - m32) / torch.sqrt(v32 + 1e-5))
        flips_hip = int(((z_mine > 0) != (z64 > 0)).sum()); flips_cpu = int(((z32 > 0) != (z64 > 0)).sum())
        print(f"     BN-out sign flips vs f64: hip {flips_hip} cpu32 {flips_cpu} of {z64.numel()};"
              f" max|z err| hip {float((z_mine - z64).abs().max()):.2e} cpu32 {float((z32.double() - z64).abs().max()):.2e}"
              f"  max|mean/std| {float((mean.abs() / var.sqrt()).max()):.1f}")
print("input grad: hip-vs-cpu32", rel(xd.grad.cpu(), torch.zeros(1)) if False else "")
