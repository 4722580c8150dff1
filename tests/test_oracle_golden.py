"""CPU: the oracle restatement (oracle/mocogan_ref.py + oracle/ode_ref.py) against fixtures produced by the
reference's own classes (oracle/make_goldens.py).  Tolerance: the two run the same stock torch CPU kernels in the
same order, so agreement is expected to ~1e-6 relative; 1e-5 is asserted (north_star's bar is 1e-4)."""
import numpy as np
import pytest
import torch

from conftest import golden, load_sd, rel_err, seed_all
from oracle import mocogan_ref as M
from oracle import ode_ref

TOL = 1e-5


def test_ode_fixture_forward_and_adjoint():
    g = golden("ode_rk4.npz")
    f = load_sd(M.OdeRhs(16, 16), g, "w")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    sol = ode_ref.odeint_adjoint(f, x, torch.from_numpy(g["t"]), method="rk4")
    assert rel_err(sol.detach(), g["sol"]) < TOL
    assert torch.equal(sol[0].detach(), x.detach())  # first output time is y0 itself
    sol.backward(torch.from_numpy(g["grad_sol"]))
    assert rel_err(x.grad, g["grad_x"]) < TOL
    for k, p in f.named_parameters():
        assert rel_err(p.grad, g[f"g/{k}"]) < TOL, k


@pytest.mark.parametrize("tag,mnist,n_vid,n_img", [("mnist_tiny", True, 4, 4), ("ucf_tiny", False, 1, 3)])
def test_generator_fixture(tag, mnist, n_vid, n_img):
    g = golden(f"gen_{tag}.npz")
    s = int(g["seed"])
    gen = M.Generator(1 if mnist else 3, 50, 0, 16, 16, dim_hidden=16, ngf=8, mnist=mnist)
    load_sd(gen, g, "w")
    seed_all(s + 1)
    vid, labels = gen.sample_videos(n_vid)
    seed_all(s + 2)
    img, _ = gen.sample_images(n_img)
    assert vid.shape == g["videos"].shape and img.shape == g["images"].shape
    assert labels.dtype == torch.float64 and np.array_equal(labels.numpy(), g["labels"])
    assert rel_err(vid.detach(), g["videos"]) < TOL
    assert rel_err(img.detach(), g["images"]) < TOL
    wv, wi = torch.from_numpy(g["wv"].astype(np.float32)), torch.from_numpy(g["wi"].astype(np.float32))
    ((vid * wv).sum() + (img * wi).sum()).backward()
    for k, p in gen.named_parameters():
        ref = g[f"g/{k}"]
        if ref.size == 0:
            assert p.grad is None, k  # the dead GRU cell never receives a gradient
        else:
            assert rel_err(p.grad, ref) < 5e-5, k
    for k, v in gen.state_dict().items():
        if "running_" in k or "num_batches" in k:
            assert rel_err(v, g[f"w_after/{k}"]) < TOL, k
    gen.eval()
    seed_all(s + 4)
    with torch.no_grad():
        ev, _ = gen.sample_videos(n_vid)
    assert rel_err(ev, g["videos_eval"]) < TOL


@pytest.mark.parametrize("tag,ctor", [
    ("vid_mnist_tiny", lambda: M.VideoDisc(1, ksize=2, ndf=8)),
    ("vid_ucf_tiny", lambda: M.VideoDisc(3, ndf=8)),
    ("img_mnist_tiny", lambda: M.PatchImageDisc(1, ndf=8)),
    ("img_ucf_tiny", lambda: M.PatchImageDisc(3, ndf=8)),
])
def test_discriminator_fixture(tag, ctor):
    g = golden(f"disc_{tag}.npz")
    dis = load_sd(ctor(), g, "w")
    x = torch.from_numpy(g["x"].astype(np.float32)).requires_grad_(True)
    logits, _ = dis(x)
    assert logits.shape == g["logits"].shape
    assert rel_err(logits.detach(), g["logits"]) < TOL
    loss = torch.nn.BCEWithLogitsLoss()(logits, torch.ones_like(logits))
    assert rel_err(loss.detach(), g["loss"]) < TOL
    loss.backward()
    assert rel_err(x.grad, g["grad_x"]) < 5e-5
    for k, p in dis.named_parameters():
        assert rel_err(p.grad, g[f"g/{k}"]) < 5e-5, k


@pytest.mark.parametrize("tag,build,iters", [("mnist_tiny", M.build_mnist, 2), ("ucf_tiny", M.build_ucf, 1)])
def test_train_step_fixture(tag, build, iters):
    g = golden(f"train_{tag}.npz")
    s = int(g["seed"])
    gen, dv, di = build(ngf=8, ndf=8)
    for m, p in ((gen, "gen"), (dv, "vid"), (di, "img")):
        load_sd(m, g, f"w0/{p}")
    opts = M.make_optimizers(gen, dv, di)
    for it in range(iters):
        imgs = [torch.from_numpy(g[f"real_img/{it}/{i}"].astype(np.float32)) for i in range(2)]
        vids = [torch.from_numpy(g[f"real_vid/{it}/{i}"].astype(np.float32)) for i in range(2)]
        seed_all(s + 1 + it)
        losses = M.train_step(gen, dv, di, opts, imgs, vids)
        assert np.allclose([float(v) for v in losses], g["losses"][it], rtol=2e-5, atol=0)
    for m, p in ((gen, "gen"), (dv, "vid"), (di, "img")):
        for k, v in m.state_dict().items():
            ref = g[f"w1/{p}/{k}"]
            if v.dtype == torch.int64:
                assert int(v) == int(ref), k
            else:
                # Adam's first steps move every weight by ~lr regardless of gradient scale: compare absolutely
                assert float((v - torch.from_numpy(ref)).abs().max()) < 2e-5, (p, k)
    gen.eval()
    seed_all(s + 50)
    with torch.no_grad():
        ev, _ = gen.sample_videos(g["videos_eval"].shape[0])
    assert rel_err(ev, g["videos_eval"]) < 1e-3
