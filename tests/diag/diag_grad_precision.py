"""Diagnostic (GPU box): per-parameter gradient error of the HIP path and of the fp32 CPU oracle, both measured
against the oracle run in float64 on the same fp32 noise draws (full-width MNIST config, batch 32).
    python scripts/diag_grad_precision.py"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gan_ode_amd as G
from oracle import mocogan_ref as M


def seed_all(s):
    torch.manual_seed(s); np.random.seed(s)


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).abs().max() / b.abs().max())


def rl2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm())


seed_all(7)
gen, dv, di = G.build_mnist()
o32 = M.build_mnist()
for m, o in zip((gen, dv, di), o32):
    o.load_state_dict(m.state_dict())
o64 = [copy.deepcopy(o).double() for o in o32]
gen.cuda(); dv.cuda(); di.cuda()
B = 32


def run_oracle(models):
    g, v, i = models
    seed_all(8)
    vid, _ = g.sample_videos(B); img, _ = g.sample_images(B)
    pv, _ = v(vid); pi, _ = i(img)
    bce = torch.nn.BCEWithLogitsLoss()
    loss = bce(pv, torch.ones_like(pv)) + bce(pi, torch.ones_like(pi))
    loss.backward()
    return vid.detach(), float(loss.detach())


seed_all(8)
vid, _ = gen.sample_videos(B); img, _ = gen.sample_images(B)
pv, _ = dv(vid); pi, _ = di(img)
loss = G.bce_with_logits_const(pv, 1.0) + G.bce_with_logits_const(pi, 1.0)
loss.backward()
v32, l32 = run_oracle(o32)
v64, l64 = run_oracle(o64)
print(f"frames: hip-vs-f64 {rel(vid.detach().cpu(), v64):.2e}  cpu32-vs-f64 {rel(v32, v64):.2e}  hip-vs-cpu32 {rel(vid.detach().cpu(), v32):.2e}")
print(f"loss:   hip {float(loss.detach()):.8f} cpu32 {l32:.8f} f64 {l64:.8f}")
def chan_report(name, p, q, r):
    """per-output-channel relative error: shows whether an L2 outlier is ONE channel (a ReLU kink flip upstream)."""
    pg, qg, rg = p.grad.cpu().double(), q.grad.double(), r.grad.double()
    pg, qg, rg = pg.reshape(pg.shape[0], -1), qg.reshape(qg.shape[0], -1), rg.reshape(rg.shape[0], -1)
    if name.endswith("main.6.weight") or name.endswith("main.3.weight"):   # ConvTranspose: out channels on dim 1
        shp = p.grad.shape
        pg = p.grad.cpu().double().permute(1, 0, 2, 3).reshape(shp[1], -1)
        qg = q.grad.double().permute(1, 0, 2, 3).reshape(shp[1], -1); rg = r.grad.double().permute(1, 0, 2, 3).reshape(shp[1], -1)
    eh = ((pg - rg).norm(dim=1) / rg.norm()).tolist(); ec = ((qg - rg).norm(dim=1) / rg.norm()).tolist()
    top = sorted(range(len(eh)), key=lambda i: -eh[i])[:4]
    med = sorted(eh)[len(eh) // 2]
    print(f"   {name}: hip top channels {[(i, f'{eh[i]:.1e}') for i in top]} median {med:.1e}; cpu32 max {max(ec):.1e}")


for k in ("main.7.bias", "main.6.weight", "main.4.bias"):
    P, Q, R = dict(gen.named_parameters()), dict(o32[0].named_parameters()), dict(o64[0].named_parameters())
    chan_report("gen." + k, P[k], Q[k], R[k])

for tag, m, a, b in (("gen", gen, o32[0], o64[0]), ("vid", dv, o32[1], o64[1]), ("img", di, o32[2], o64[2])):
    for (k, p), (_, q), (_, r) in zip(m.named_parameters(), a.named_parameters(), b.named_parameters()):
        if r.grad is None:
            continue
        print(f"{tag}.{k:24s} max-norm: hip-vs-f64 {rel(p.grad.cpu(), r.grad):.2e} cpu32-vs-f64 {rel(q.grad, r.grad):.2e}"
              f" | L2: hip-vs-f64 {rl2(p.grad.cpu(), r.grad):.2e} cpu32-vs-f64 {rl2(q.grad, r.grad):.2e}")
