import sys; sys.path.insert(0, "/root/repo")
import torch, gan_ode_amd as G
for name, (gen, dv, di), B in (("mnist", G.build_mnist(), 32), ("ucf", G.build_ucf(), 16)):
    gen.cuda()
    p = gen._joint_plan(B, B, 16, False)
    with torch.no_grad():
        gen.sample_pair(B, B)
    print(name, "two fwd", p.stack._two_f)
    (v, _), (i, _) = gen.sample_pair(B, B)
    (v.sum() + i.sum()).backward()
    print(name, "two bwd", p.stack._two_b)
