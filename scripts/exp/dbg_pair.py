import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import copy, torch, numpy as np
import gan_ode_amd as G
from conftest import rel_err
torch.manual_seed(0)
for name, mk, shape in (("video", lambda: G.VideoDiscriminator(1, ksize=2, ndf=64), (32, 1, 16, 28, 28)), ("image", lambda: G.PatchImageDiscriminator(1, ndf=64), (32, 1, 28, 28))):
    dis = mk(); dis2 = copy.deepcopy(dis); dis.cuda(); dis2.cuda()
    real, fake = torch.rand(shape).cuda(), (torch.rand(shape) * 2 - 1).cuda()
    with torch.no_grad():
        joint = dis.forward_pair_joint(real, fake)
        (pr, _), (pf, _) = dis.forward_pair(real, fake)
        qr, _ = dis2(real); qf, _ = dis2(fake)
    print(name, "joint", tuple(joint.shape), "abs max", float(joint.abs().max()), "pr err", rel_err(pr.cpu(), qr.cpu()), "pf err", rel_err(pf.cpu(), qf.cpu()))
    plan = dis._pool.plans[("pair",) + shape][0]
    plan2 = [p for k, v in dis2._pool.plans.items() for p in v][0]
    for l in range(plan.nl - 1):
        y = plan.y[l]; B = y.shape[0] // 2
        print("  layer", l, "y pair vs single: first half err", rel_err(y[:B].cpu(), plan2.y[l].cpu()) if True else None, " (plan2 holds the LAST pass = fake) second half err", rel_err(y[B:].cpu(), plan2.y[l].cpu()),
              "| a exists", plan.a[l] is not None, "absmax a", float(plan.a[l].abs().max()) if plan.a[l] is not None else None)
