"""Generates tests/golden/*.npz by running the REFERENCE's own model classes on CPU in the build container.

Run from the repo root:   python oracle/make_goldens.py
Needs /root/reference (read-only); never runs on the GPU box.  The reference's classes are imported, not copied:
``models.mocogan`` imports as is; ``models.mocogan_ode`` needs the third-party package ``torchdiffeq`` which is
absent from the image, so oracle/ode_ref.py (the restated integrator, "parity unpinned", see its header) is
registered under that name before the import.  Every layer shape, default init, RNG call order and reshape in the
fixtures therefore comes from the reference itself; only the integrator arithmetic is the restatement's.

The reference has no callable train step (the loop body is inline in train(), mnist_moco_ode.py:113-163), so
``_ref_train_iteration`` below drives the reference's classes through the same sequence of calls.

Fixtures are data only: seeds, initial weights (tiny widths, ngf=ndf=8, so the files stay small), inputs and the
reference's outputs / gradients / post-step weights.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def _import_reference():
    sys.path.insert(0, REF)          # so that `models` is the reference's package, not the drop-in one
    sys.path.insert(1, REPO)
    from oracle import ode_ref
    shim = types.ModuleType("torchdiffeq")
    shim.odeint = ode_ref.odeint
    shim.odeint_adjoint = ode_ref.odeint_adjoint
    sys.modules["torchdiffeq"] = shim
    import models.mocogan as base
    import models.mocogan_ode as ode
    assert base.__file__.startswith(REF) and ode.__file__.startswith(REF)
    return base, ode, ode_ref


def seed(s):
    torch.manual_seed(s)
    np.random.seed(s)


def sd_np(module, prefix):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def bn_buffers_np(module, prefix):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()
            if "running_" in k or "num_batches" in k}


def h16(x):
    """Round to fp16-representable values so the array can be stored exactly as float16 (halves the file)."""
    return x.half().float()


def grads_np(module, prefix):
    return {f"{prefix}/{k}": (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(0, np.float32))
            for k, p in module.named_parameters()}


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def golden_ode(ode, ode_ref):
    seed(11)
    f = ode.ODEFunc(16, 16)
    x = torch.randn(8, 16, requires_grad=True)
    t = torch.linspace(0, 1, 16).float()
    sol = ode_ref.odeint_adjoint(f, x, t, method="rk4")
    g = torch.randn_like(sol)
    sol.backward(g)
    arrays = dict(x=x.detach().numpy(), t=t.numpy(), sol=sol.detach().numpy(), grad_sol=g.numpy(),
                  grad_x=x.grad.numpy())
    arrays.update(sd_np(f, "w"))
    arrays.update(grads_np(f, "g"))
    save("ode_rk4.npz", **arrays)


def golden_generator(ode, tag, ctor, n_vid, n_img, s):
    seed(s)
    gen = ctor()
    arrays = sd_np(gen, "w")
    seed(s + 1)
    vid, labels = gen.sample_videos(n_vid)
    seed(s + 2)
    img, _ = gen.sample_images(n_img)
    seed(s + 3)
    wv, wi = h16(torch.randn_like(vid)), h16(torch.randn_like(img))
    ((vid * wv).sum() + (img * wi).sum()).backward()
    arrays.update(grads_np(gen, "g"))
    arrays.update(bn_buffers_np(gen, "w_after"))  # BN running stats after the two train-mode calls
    arrays.update(videos=vid.detach().numpy(), labels=labels.numpy(), images=img.detach().numpy(),
                  wv=wv.numpy().astype(np.float16), wi=wi.numpy().astype(np.float16), seed=np.int64(s))
    gen.eval()
    seed(s + 4)
    with torch.no_grad():
        ev, _ = gen.sample_videos(n_vid)
    arrays["videos_eval"] = ev.numpy()
    save(f"gen_{tag}.npz", **arrays)


def golden_disc(base, tag, ctor, x, s):
    seed(s)
    dis = ctor()
    arrays = sd_np(dis, "w")
    x = h16(x).requires_grad_(True)
    logits, _ = dis(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, torch.ones_like(logits))
    loss.backward()
    arrays.update(grads_np(dis, "g"))
    arrays.update(bn_buffers_np(dis, "w_after"))
    arrays.update(x=x.detach().numpy().astype(np.float16), logits=logits.detach().numpy(), loss=loss.detach().numpy(),
                  grad_x=x.grad.numpy())
    save(f"disc_{tag}.npz", **arrays)


def _ref_train_iteration(gen, dis_vid, dis_img, opts, real_imgs, real_vids, batch, d_iters=2):
    """Drives the reference classes through the call sequence of mnist_moco_ode.py:113-163."""
    gen_opt, vid_opt, img_opt = opts
    bce = torch.nn.BCEWithLogitsLoss()
    for i in range(d_iters):
        img_opt.zero_grad()
        pr, _ = dis_img(real_imgs[i])
        with torch.no_grad():
            fake, _ = gen.sample_images(batch)
        pf, _ = dis_img(fake)
        li = bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf))
        li.backward()
        img_opt.step()
        vid_opt.zero_grad()
        pr, _ = dis_vid(real_vids[i].transpose(1, 2))
        with torch.no_grad():
            fake, _ = gen.sample_videos(batch)
        pf, _ = dis_vid(fake)
        lv = bce(pr, torch.ones_like(pr)) + bce(pf, torch.zeros_like(pf))
        lv.backward()
        vid_opt.step()
    gen_opt.zero_grad()
    fv, _ = gen.sample_videos(batch)
    fi, _ = gen.sample_images(batch)
    pv, _ = dis_vid(fv)
    pi, _ = dis_img(fi)
    lg = bce(pv, torch.ones_like(pv)) + bce(pi, torch.ones_like(pi))
    lg.backward()
    gen_opt.step()
    return li.item(), lv.item(), lg.item()


def golden_train_step(base, ode, tag, mk, C, HW, B, s, iters=2, n_eval=4):
    seed(s)
    gen, dis_vid, dis_img = mk()
    arrays = {}
    for m, p in ((gen, "gen"), (dis_vid, "vid"), (dis_img, "img")):
        arrays.update(sd_np(m, f"w0/{p}"))
    adam = lambda m: torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)  # noqa: E731
    opts = (adam(gen), adam(dis_vid), adam(dis_img))
    g = torch.Generator().manual_seed(s + 100)
    losses = []
    for it in range(iters):
        real_imgs = [h16(torch.rand(B, C, HW, HW, generator=g)) for _ in range(2)]
        real_vids = [h16(torch.rand(B, 16, C, HW, HW, generator=g)) for _ in range(2)]
        for i in range(2):
            arrays[f"real_img/{it}/{i}"] = real_imgs[i].numpy().astype(np.float16)
            arrays[f"real_vid/{it}/{i}"] = real_vids[i].numpy().astype(np.float16)
        seed(s + 1 + it)
        losses.append(_ref_train_iteration(gen, dis_vid, dis_img, opts, real_imgs, real_vids, B))
    arrays["losses"] = np.asarray(losses, np.float64)
    for m, p in ((gen, "gen"), (dis_vid, "vid"), (dis_img, "img")):
        arrays.update(sd_np(m, f"w1/{p}"))
    gen.eval()
    seed(s + 50)
    with torch.no_grad():
        ev, _ = gen.sample_videos(n_eval)
    arrays["videos_eval"] = ev.numpy()
    arrays["seed"] = np.int64(s)
    save(f"train_{tag}.npz", **arrays)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    base, ode, ode_ref = _import_reference()
    golden_ode(ode, ode_ref)

    golden_generator(ode, "mnist_tiny", lambda: ode.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8), 4, 4, 21)
    golden_generator(ode, "ucf_tiny", lambda: ode.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8), 1, 3, 31)

    g = torch.Generator().manual_seed(5)
    golden_disc(base, "vid_mnist_tiny", lambda: base.VideoDiscriminator(1, ksize=2, ndf=8),
                torch.rand(3, 1, 16, 28, 28, generator=g), 41)
    golden_disc(base, "vid_ucf_tiny", lambda: base.VideoDiscriminator(3, ndf=8),
                torch.rand(1, 3, 16, 64, 64, generator=g) * 2 - 1, 42)
    golden_disc(base, "img_mnist_tiny", lambda: base.PatchImageDiscriminator(1, ndf=8),
                torch.rand(5, 1, 28, 28, generator=g), 43)
    golden_disc(base, "img_ucf_tiny", lambda: base.PatchImageDiscriminator(3, ndf=8),
                torch.rand(3, 3, 64, 64, generator=g) * 2 - 1, 44)

    golden_train_step(base, ode, "mnist_tiny",
                      lambda: (ode.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8),
                               base.VideoDiscriminator(1, ksize=2, ndf=8), base.PatchImageDiscriminator(1, ndf=8)),
                      C=1, HW=28, B=8, s=61)
    golden_train_step(base, ode, "ucf_tiny",
                      lambda: (ode.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8),
                               base.VideoDiscriminator(3, ndf=8), base.PatchImageDiscriminator(3, ndf=8)),
                      C=3, HW=64, B=2, s=71, iters=1, n_eval=1)


if __name__ == "__main__":
    main()
