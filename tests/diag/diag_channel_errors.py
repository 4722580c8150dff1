"""Per-tensor, per-output-channel gradient error of the HIP path and of the stock fp32 CPU kernels against the float64
oracle (same fp32 draws): G-step gradients at BASELINE sizes.  python scripts/diag_channel_errors.py [mnist|ucf]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import gan_ode_amd as G
from oracle import mocogan_ref as M
from test_gpu_configs import channel_errors

cfg = sys.argv[1] if len(sys.argv) > 1 else "mnist"
G.limit_host_threads()
torch.manual_seed(81); np.random.seed(81)
nets = G.build_ucf() if cfg == "ucf" else G.build_mnist()
o32 = M.build_ucf() if cfg == "ucf" else M.build_mnist()
for m, o in zip(nets, o32):
    o.load_state_dict(m.state_dict())
o64 = [copy.deepcopy(o).double() for o in o32]
for m in nets:
    m.cuda()
gen, dv, di = nets
B = 16 if cfg == "ucf" else 32
torch.manual_seed(82); np.random.seed(82)
vid, _ = gen.sample_videos(B); img, _ = gen.sample_images(B)
pv, _ = dv(vid); pi, _ = di(img)
G.bce_with_logits_pair(pv, 1.0, pi, 1.0).backward()
bce = torch.nn.BCEWithLogitsLoss()
for og, ov, oi in (o32, o64):
    torch.manual_seed(82); np.random.seed(82)
    rvid, _ = og.sample_videos(B); rimg, _ = og.sample_images(B)
    rpv, _ = ov(rvid); rpi, _ = oi(rimg)
    (bce(rpv, torch.ones_like(rpv)) + bce(rpi, torch.ones_like(rpi))).backward()
print(f"{'tensor':34s} {'hip<1e-3':>8s} {'cpu<1e-3':>8s} {'hip med':>9s} {'cpu med':>9s} {'hip max':>9s} {'cpu max':>9s} {'hip-cpu med':>11s}")
for m, a, b in zip(nets, o32, o64):
    for (k, p), (_, q), (_, r) in zip(m.named_parameters(), a.named_parameters(), b.named_parameters()):
        if r.grad is None:
            continue
        eh, ec = channel_errors(p.grad.cpu(), r.grad), channel_errors(q.grad, r.grad)
        ehc = channel_errors(p.grad.cpu(), q.grad.double())
        print(f"{type(m).__name__[:14] + '.' + k:34s} {float((eh < 1e-3).double().mean()):8.3f} {float((ec < 1e-3).double().mean()):8.3f} "
              f"{float(eh.median()):9.2e} {float(ec.median()):9.2e} {float(eh.max()):9.2e} {float(ec.max()):9.2e} {float(ehc.median()):11.2e}")
