import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gan_ode_amd as G
from oracle import mocogan_ref as M
torch.manual_seed(0); np.random.seed(0)
gen, dv, di = G.build_mnist(); ogen, _, _ = M.build_mnist()
ogen.load_state_dict(gen.state_dict()); gen.cuda()
for B in (64, 256):
    torch.manual_seed(1); np.random.seed(1)
    with torch.no_grad():
        v, _ = gen.sample_videos(B)
    torch.cuda.synchronize()
    print(B, tuple(v.shape), bool(torch.isfinite(v).all()), float(v.abs().max()))
    if B == 64:
        torch.manual_seed(1); np.random.seed(1)
        with torch.no_grad():
            ov, _ = ogen.sample_videos(B)
        err = float((v.cpu() - ov).abs().max() / ov.abs().max())
        print("rel err vs oracle", err)
        assert err < 1e-4
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(10): gen.sample_videos(256)
torch.cuda.synchronize()
print("B=256: %.2f ms/step, %.0f videos/s" % ((time.perf_counter() - t0) * 100, 2560 / (time.perf_counter() - t0)))
dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di)
g = torch.Generator().manual_seed(1)
imgs = [torch.rand(128, 1, 28, 28, generator=g).cuda() for _ in range(2)]
vids = [torch.rand(128, 16, 1, 28, 28, generator=g).cuda() for _ in range(2)]
for _ in range(2): l = tr.step(imgs, vids)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = tr.step(imgs, vids)
torch.cuda.synchronize()
print("B=128 iteration %.1f ms" % ((time.perf_counter() - t0) * 200), [float(x) for x in l])
