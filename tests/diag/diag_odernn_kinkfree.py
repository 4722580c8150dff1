"""Where do the ODE-RNN generator's kink-free gradients differ from the oracle?  fp32 device vs fp32 oracle vs fp64 oracle,
videos only / images only."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import rel_err, seed_all
import gan_ode_amd as G
from oracle import mocogan_ref as M
G.limit_host_threads()
seed_all(123)
gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16)
ogen = M.GeneratorOdeRnn(1, 50, 0, 16, 16, mnist=True)
ogen.load_state_dict(gen.state_dict())
with torch.no_grad():
    for m in (gen, ogen):
        for mod in m.main:
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.bias.fill_(6.0)
o64 = copy.deepcopy(ogen).double()
gen.cuda()
B = 32
for which in ("videos", "images"):
    outs = []
    for m in (gen, ogen, o64):
        m.zero_grad()
        seed_all(124)
        x, _ = (m.sample_videos(B) if which == "videos" else m.sample_images(B))
        w = torch.randn(x.shape, generator=torch.Generator().manual_seed(3))
        (x * w.to(x.device).to(x.dtype)).sum().backward()
        outs.append(x.detach().cpu().double())
    print(which, "frames hip-vs-o32", rel_err(outs[0], outs[1]), "hip-vs-f64", rel_err(outs[0], outs[2]), "o32-vs-f64", rel_err(outs[1], outs[2]))
    r32, r64 = dict(ogen.named_parameters()), dict(o64.named_parameters())
    for k, p in gen.named_parameters():
        if r64[k].grad is None or k.startswith("main."):
            continue
        print(f"  {k:28s} hip-o32 {rel_err(p.grad.cpu(), r32[k].grad):.2e}  hip-f64 {rel_err(p.grad.cpu(), r64[k].grad):.2e}  o32-f64 {rel_err(r32[k].grad, r64[k].grad):.2e}")

print("---- combined loss, networks built as tests/test_gpu_configs.py::_odernn_pair(123) does")
seed_all(123)
_, dv, di = G.build_mnist()
gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16)
ogen = M.GeneratorOdeRnn(1, 50, 0, 16, 16, mnist=True)
ogen.load_state_dict(gen.state_dict())
with torch.no_grad():
    for m in (gen, ogen):
        for mod in m.main:
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.bias.fill_(6.0)
        m.main[12].weight.mul_(0.05)      # keeps the head's 6 * sum(w) offset inside tanh's linear range
o64 = copy.deepcopy(ogen).double()
gen.cuda()
for mode in ("combined", "videos", "images"):
    outs = []
    for m in (gen, ogen, o64):
        m.zero_grad()
        seed_all(124)
        vid, _ = m.sample_videos(B)
        img, _ = m.sample_images(B)
        wv = torch.randn(vid.shape, generator=torch.Generator().manual_seed(3)).to(vid.device).to(vid.dtype)
        wi = torch.randn(img.shape, generator=torch.Generator().manual_seed(4)).to(vid.device).to(vid.dtype)
        loss = (vid * wv).sum() * (mode != "images") + (img * wi).sum() * (mode != "videos")
        loss.backward()
        outs.append((vid.detach().cpu().double(), img.detach().cpu().double()))
    print(mode, "frames hip-o32", rel_err(outs[0][0], outs[1][0]), rel_err(outs[0][1], outs[1][1]))
    r32, r64 = dict(ogen.named_parameters()), dict(o64.named_parameters())
    for k, p in gen.named_parameters():
        if r64[k].grad is None:
            continue
        print(f"  {k:28s} hip-o32 {rel_err(p.grad.cpu(), r32[k].grad):.2e}  hip-f64 {rel_err(p.grad.cpu(), r64[k].grad):.2e}  o32-f64 {rel_err(r32[k].grad, r64[k].grad):.2e}")
