"""GPU: the drop-in module surface (sample_videos / sample_images / discriminators / train step) on libgode.so
against (a) the golden fixtures produced by the reference's own classes and (b) the CPU oracle on identical seeds,
at tiny widths and at BASELINE.json's full configuration (batch 32, 16x1x28x28, ngf=ndf=64).
Tolerance: 1e-4 relative fp32 on frames and losses (north_star); parameter gradients 5e-4 (long fp32 reductions in
a different order); post-Adam weights compared absolutely (Adam normalises the step to ~lr)."""
import numpy as np
import pytest
import torch

from conftest import golden, load_sd, rel_err, seed_all

import gan_ode_amd as G
from oracle import mocogan_ref as M

pytestmark = pytest.mark.gpu
TOL = 1e-4
GTOL = 5e-4


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64).flatten()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def robust_rel(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64).flatten()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64).flatten()
    return float((a - b).abs().median() / (b.norm() / b.numel() ** 0.5).clamp_min(1e-300))


def _f32(a):
    return torch.from_numpy(np.asarray(a).astype(np.float32))


@pytest.mark.parametrize("tag,mnist,n_vid,n_img", [("mnist_tiny", True, 4, 4), ("ucf_tiny", False, 1, 3)])
def test_generator_against_reference_fixture(tag, mnist, n_vid, n_img):
    g = golden(f"gen_{tag}.npz")
    s = int(g["seed"])
    gen = (G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8) if mnist
           else G.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8))
    load_sd(gen, g, "w")
    gen.cuda()
    seed_all(s + 1)
    vid, labels = gen.sample_videos(n_vid)
    seed_all(s + 2)
    img, _ = gen.sample_images(n_img)
    assert tuple(vid.shape) == g["videos"].shape and tuple(img.shape) == g["images"].shape
    assert labels.dtype == torch.float64 and labels.is_cuda and float(labels.abs().sum()) == 0.0
    assert rel_err(vid.detach().cpu(), g["videos"]) < TOL
    assert rel_err(img.detach().cpu(), g["images"]) < TOL
    ((vid * _f32(g["wv"]).cuda()).sum() + (img * _f32(g["wi"]).cuda()).sum()).backward()
    for k, p in gen.named_parameters():
        ref = g[f"g/{k}"]
        if ref.size == 0:
            assert p.grad is None, k
        else:
            # median-based error: robust against a single (Leaky)ReLU/BatchNorm kink flipping between two fp32
            # evaluations (see test_full_width_mnist_batch32_against_oracle); max-norm only as a sanity bound
            # In these tiny nets a BatchNorm channel averages only 256..4096 elements, so ONE flipped kink is a
            # 1/256 = 4e-3 relative change of that channel's gradient and shifts everything upstream by about as
            # much; 5e-3 bounds one flip, a wrong formula is O(0.1..1).
            assert robust_rel(p.grad.cpu(), ref) < 5e-3, (k, robust_rel(p.grad.cpu(), ref), rel_err(p.grad.cpu(), ref))
            assert rel_err(p.grad.cpu(), ref) < 5e-2, (k, rel_err(p.grad.cpu(), ref))
    for k, v in gen.state_dict().items():
        if "running_" in k:
            assert rel_err(v.cpu(), g[f"w_after/{k}"]) < TOL, k
        if "num_batches" in k:
            assert int(v) == int(g[f"w_after/{k}"]), k
    gen.eval()
    seed_all(s + 4)
    with torch.no_grad():
        ev, _ = gen.sample_videos(n_vid)
    assert rel_err(ev.cpu(), g["videos_eval"]) < TOL


@pytest.mark.parametrize("tag,mnist,n_vid,n_img", [("mnist_tiny", True, 4, 4), ("ucf_tiny", False, 1, 3)])
def test_fixture_generators_kink_free_gradients_at_1e_4(tag, mnist, n_vid, n_img):
    """The fixture test above bounds the generator gradients at a median 5e-3 / max 5e-2 because ONE flipped
    BatchNorm/ReLU kink moves a channel of these tiny nets by 4e-3.  The same nets and call sizes with every kink out of
    reach (BatchNorm beta = +6: all pre-activations positive; head weights x 0.05 so that the offset stays inside tanh's
    linear range -- see tests/test_gpu_configs.py::test_motion_latent_gradients_kink_free_full_width) against the oracle,
    which the fixtures pin: every parameter gradient at 1e-4 max-norm."""
    g = golden(f"gen_{tag}.npz")
    s = int(g["seed"])
    gen = (G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8) if mnist
           else G.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8))
    ogen = M.Generator(1 if mnist else 3, 50, 0, 16, 16, dim_hidden=None if mnist else 16, ngf=8, mnist=mnist)
    load_sd(gen, g, "w")
    with torch.no_grad():
        for mod in gen.main:
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.bias.fill_(6.0)
        gen.main[12].weight.mul_(0.05)
    ogen.load_state_dict({k: v.detach().clone() for k, v in gen.state_dict().items()})
    gen.cuda()
    seed_all(s + 1)
    vid, _ = gen.sample_videos(n_vid)
    seed_all(s + 2)
    img, _ = gen.sample_images(n_img)
    seed_all(s + 1)
    ovid, _ = ogen.sample_videos(n_vid)
    seed_all(s + 2)
    oimg, _ = ogen.sample_images(n_img)
    assert rel_err(vid.detach().cpu(), ovid.detach()) < TOL and rel_err(img.detach().cpu(), oimg.detach()) < TOL
    wv, wi = _f32(g["wv"]), _f32(g["wi"])
    ((vid * wv.cuda()).sum() + (img * wi.cuda()).sum()).backward()
    ((ovid * wv).sum() + (oimg * wi).sum()).backward()
    ref = dict(ogen.named_parameters())
    errs = {}
    for k, p in gen.named_parameters():
        if ref[k].grad is None:
            assert p.grad is None, k
        else:
            errs[k] = rel_err(p.grad.cpu(), ref[k].grad)
    bad = {k: e for k, e in errs.items() if e >= 1e-4}
    assert not bad, (bad, errs)


@pytest.mark.parametrize("tag,ctor", [
    ("vid_mnist_tiny", lambda: G.VideoDiscriminator(1, ksize=2, ndf=8)),
    ("vid_ucf_tiny", lambda: G.VideoDiscriminator(3, ndf=8)),
    ("img_mnist_tiny", lambda: G.PatchImageDiscriminator(1, ndf=8)),
    ("img_ucf_tiny", lambda: G.PatchImageDiscriminator(3, ndf=8)),
])
def test_discriminator_against_reference_fixture(tag, ctor):
    g = golden(f"disc_{tag}.npz")
    dis = load_sd(ctor(), g, "w").cuda()
    x = _f32(g["x"]).cuda().requires_grad_(True)
    logits, none = dis(x)
    assert none is None and tuple(logits.shape) == g["logits"].shape
    assert rel_err(logits.detach().cpu(), g["logits"]) < TOL
    loss = G.bce_with_logits_const(logits, 1.0)
    assert rel_err(loss.detach().cpu(), g["loss"]) < TOL
    loss.backward()
    assert rel_err(x.grad.cpu(), g["grad_x"]) < GTOL
    for k, p in dis.named_parameters():
        assert rel_err(p.grad.cpu(), g[f"g/{k}"]) < GTOL, k
    for k, v in dis.state_dict().items():
        if "running_" in k:
            assert rel_err(v.cpu(), g[f"w_after/{k}"]) < TOL, k


def test_discriminator_reads_transposed_view_in_place():
    """mnist_moco_ode.py:136-139 feeds real.transpose(1, 2); UCF-shaped data makes that view non-contiguous."""
    gen = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    dis, ref = G.VideoDiscriminator(3, ndf=8), M.VideoDisc(3, ndf=8)
    ref.load_state_dict(dis.state_dict())
    dis.cuda()
    real = torch.rand(2, 16, 3, 64, 64, generator=gen)
    out, _ = dis(real.cuda().transpose(1, 2))
    want, _ = ref(real.transpose(1, 2))
    assert rel_err(out.detach().cpu(), want.detach()) < TOL


@pytest.mark.parametrize("tag,build,obuild,iters", [("mnist_tiny", G.build_mnist, M.build_mnist, 2),
                                                    ("ucf_tiny", G.build_ucf, M.build_ucf, 1)])
def test_train_step_against_reference_fixture(tag, build, obuild, iters):
    g = golden(f"train_{tag}.npz")
    s = int(g["seed"])
    gen, dv, di = build(ngf=8, ndf=8)
    for m, p in ((gen, "gen"), (dv, "vid"), (di, "img")):
        load_sd(m, g, f"w0/{p}")
        m.cuda()
    tr = G.GanTrainer(gen, dv, di)
    for it in range(iters):
        imgs = [_f32(g[f"real_img/{it}/{i}"]).cuda() for i in range(2)]
        vids = [_f32(g[f"real_vid/{it}/{i}"]).cuda() for i in range(2)]
        seed_all(s + 1 + it)
        losses = G.train_step(tr, imgs, vids)
        got = [float(v) for v in losses]
        assert np.allclose(got, g["losses"][it], rtol=2e-4, atol=0), (got, g["losses"][it])
    for m, p in ((gen, "gen"), (dv, "vid"), (di, "img")):
        for k, v in m.state_dict().items():
            ref = g[f"w1/{p}/{k}"]
            if v.dtype == torch.int64:
                assert int(v) == int(ref), k
            elif "running_" in k:
                assert rel_err(v.cpu(), ref) < 2e-4, (p, k)
            else:
                # After 1-2 Adam steps every weight has moved by ~lr=2e-4 times sign-like m/sqrt(v): a wrong gradient
                # shows as ~50% of the entries off by 2*lr.  Entries whose gradient is within fp32 rounding noise of
                # zero may legitimately flip sign between two correct fp32 evaluations, so a tiny fraction is allowed.
                d = (v.cpu() - torch.from_numpy(ref)).abs()
                assert float((d > 6e-5).float().mean()) < 2e-3, (p, k, float(d.max()))
                assert float(d.median()) < 2e-6, (p, k)
    gen.eval()
    seed_all(s + 50)
    with torch.no_grad():
        ev, _ = gen.sample_videos(g["videos_eval"].shape[0])
    assert rel_err(ev.cpu(), g["videos_eval"]) < 2e-3


def test_full_width_mnist_batch32_against_oracle():
    """BASELINE.json configs[1]: batch 32, 16x1x28x28, ngf=ndf=64 -- one generator pass, both discriminators and all
    gradients against the CPU oracle on identical weights and seeds.  Frames, logits and loss: 1e-4 relative against
    the fp32 oracle.  Gradients run through 9 train-mode BatchNorms and (Leaky)ReLU kinks: one pre-activation within
    fp32 rounding of zero flips its derivative (1 vs 0.2) and moves single gradient entries by percents in BOTH fp32
    evaluations (measured: tests/diag/diag_disc_layers.py, tests/diag/diag_grad_precision.py), so gradients are judged in
    ROBUST relative error (median |a-b| over the tensor / rms of the reference) against the oracle run in float64 on
    the same fp32 draws: see the comment at the assertion for the bound.  A single flipped kink upstream shifts ONE channel of a BatchNorm bias gradient and of the adjacent
    weight gradient by ~0.5% while every other channel agrees to ~1e-5 (tests/diag/diag_grad_precision.py prints the
    per-channel picture), so the plain L2 error only gets a sanity bound."""
    import copy
    seed_all(7)
    gen, dv, di = G.build_mnist()
    o32 = M.build_mnist()
    for m, o in zip((gen, dv, di), o32):
        o.load_state_dict(m.state_dict())
    o64 = [copy.deepcopy(o).double() for o in o32]
    gen.cuda(); dv.cuda(); di.cuda()
    B = 32
    seed_all(8)
    vid, _ = gen.sample_videos(B)
    img, _ = gen.sample_images(B)
    pv, _ = dv(vid)
    pi, _ = di(img)
    loss = G.bce_with_logits_const(pv, 1.0) + G.bce_with_logits_const(pi, 1.0)
    loss.backward()

    def oracle(models):
        og, ov, oi = models
        seed_all(8)
        rvid, _ = og.sample_videos(B)
        rimg, _ = og.sample_images(B)
        rpv, _ = ov(rvid)
        rpi, _ = oi(rimg)
        bce = torch.nn.BCEWithLogitsLoss()
        rl = bce(rpv, torch.ones_like(rpv)) + bce(rpi, torch.ones_like(rpi))
        rl.backward()
        return rvid.detach(), rimg.detach(), rpv.detach(), rpi.detach(), rl.detach()

    rvid, rimg, rpv, rpi, rloss = oracle(o32)
    dvid, _, _, _, dloss = oracle(o64)
    assert vid.shape == (B, 1, 16, 28, 28) and pv.shape == (B, 11, 2, 2) and pi.shape == (B,)
    assert rel_err(vid.detach().cpu(), rvid) < TOL and rel_err(vid.detach().cpu(), dvid) < TOL
    assert rel_err(img.detach().cpu(), rimg) < TOL
    assert rel_err(pv.detach().cpu(), rpv) < TOL and rel_err(pi.detach().cpu(), rpi) < TOL
    assert abs(float(loss.detach()) - float(rloss)) / abs(float(rloss)) < TOL
    assert abs(float(loss.detach()) - float(dloss)) / abs(float(dloss)) < TOL
    worst = []
    for m, a, b in zip((gen, dv, di), o32, o64):
        for (k, p), (_, q), (_, r) in zip(m.named_parameters(), a.named_parameters(), b.named_parameters()):
            if r.grad is None:
                assert p.grad is None, k
                continue
            e_hip, e_cpu = robust_rel(p.grad.cpu(), r.grad), robust_rel(q.grad, r.grad)
            worst.append((e_hip / max(e_cpu, 2.5e-5), k, e_hip, e_cpu))
            # One flipped kink moves one channel by ~0.5% and everything upstream of it by up to a few 1e-3 (measured:
            # a single flip in BatchNorm 2, channel 32, gives L2 5e-3 / median 2e-3 upstream while all tensors
            # downstream of it agree to 3e-5..2e-4).  Which pre-activation flips depends on rounding details of either
            # implementation, so this full-size check only bounds the damage; the strict gradient checks are the
            # kink-free kernel tests (tests/test_gpu_kernels.py, 1e-4..2e-4) and the tiny-width reference fixtures
            # (5e-4).  A wrong formula or tile path shows up here as an O(1) error.
            assert e_hip < max(1e-2, 4 * e_cpu), (k, e_hip, e_cpu)
            assert rel_l2(p.grad.cpu(), r.grad) < 1e-1, k
    # size-independent properties at full size
    assert float(vid.abs().max()) <= 1.0
    seed_all(8)
    with torch.no_grad():
        again, _ = gen.sample_videos(B)
    # BN running stats do not enter train-mode outputs: same seed -> bit-identical frames (deterministic reductions)
    assert torch.equal(again, vid.detach())


def test_full_width_ucf_against_oracle():
    """BASELINE.json configs[3] shapes (UCF101: 3x64x64, ngf=ndf=64, ksize=4 video discriminator) at batch 4: frames,
    logits, loss against the fp32 oracle at 1e-4; gradients bounded as in the MNIST full-width test."""
    seed_all(17)
    gen, dv, di = G.build_ucf()
    o32 = M.build_ucf()
    for m, o in zip((gen, dv, di), o32):
        o.load_state_dict(m.state_dict())
    gen.cuda(); dv.cuda(); di.cuda()
    B = 4
    seed_all(18)
    vid, _ = gen.sample_videos(B)
    img, _ = gen.sample_images(B)
    pv, _ = dv(vid)
    pi, _ = di(img)
    loss = G.bce_with_logits_const(pv, 1.0) + G.bce_with_logits_const(pi, 1.0)
    loss.backward()
    og, ov, oi = o32
    seed_all(18)
    rvid, _ = og.sample_videos(B)
    rimg, _ = og.sample_images(B)
    rpv, _ = ov(rvid)
    rpi, _ = oi(rimg)
    bce = torch.nn.BCEWithLogitsLoss()
    rloss = bce(rpv, torch.ones_like(rpv)) + bce(rpi, torch.ones_like(rpi))
    rloss.backward()
    assert vid.shape == (B, 3, 16, 64, 64) and pv.shape == (B,) and pi.shape == (B, 4, 4)
    assert not vid.is_contiguous()          # a permuted view of the channels-last buffer, like the reference's view
    assert rel_err(vid.detach().cpu(), rvid.detach()) < TOL
    assert rel_err(img.detach().cpu(), rimg.detach()) < TOL
    assert rel_err(pv.detach().cpu(), rpv.detach()) < TOL and rel_err(pi.detach().cpu(), rpi.detach()) < TOL
    assert abs(float(loss.detach()) - float(rloss.detach())) / abs(float(rloss.detach())) < TOL
    for m, a in zip((gen, dv, di), o32):
        for (k, p), (_, q) in zip(m.named_parameters(), a.named_parameters()):
            if q.grad is None:
                assert p.grad is None, k
            else:
                assert robust_rel(p.grad.cpu(), q.grad) < 1e-2 and rel_l2(p.grad.cpu(), q.grad) < 1e-1, k


def test_unchanged_driver_surface_with_stock_adam_and_bce():
    """The reference driver's own import lines and loop body (mnist_moco_ode.py:5-6,86-89,113-163) with torch's
    stock Adam and BCEWithLogitsLoss on the drop-in classes, against the oracle driven the same way."""
    from models.mocogan import VideoDiscriminator, PatchImageDiscriminator
    from models.mocogan_ode import VideoGeneratorMNISTODE as VideoGeneratorMNIST
    import torch.nn as nn
    g = golden("train_mnist_tiny.npz")
    s = int(g["seed"])
    disVid, disImg = VideoDiscriminator(1, ksize=2, ndf=8), PatchImageDiscriminator(1, ndf=8)
    gen = VideoGeneratorMNIST(1, 50, 0, 16, 16, ngf=8)
    for m, p in ((gen, "gen"), (disVid, "vid"), (disImg, "img")):
        load_sd(m, g, f"w0/{p}")
    disVid.cuda(); disImg.cuda(); gen.cuda()
    mk = lambda m: torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-5)  # noqa: E731
    disVidOpt, disImgOpt, genOpt = mk(disVid), mk(disImg), mk(gen)
    loss = nn.BCEWithLogitsLoss()
    batch_size = 8
    for it in range(2):
        seed_all(s + 1 + it)
        for i in range(2):
            disImgOpt.zero_grad()
            real = _f32(g[f"real_img/{it}/{i}"]).cuda()
            pr, _ = disImg(real)
            with torch.no_grad():
                fake, _ = gen.sample_images(batch_size)
            pf, _ = disImg(fake)
            dis_img_loss = loss(pr, torch.ones_like(pr)) + loss(pf, torch.zeros_like(pf))
            dis_img_loss.backward()
            disImgOpt.step()
            disVidOpt.zero_grad()
            real = _f32(g[f"real_vid/{it}/{i}"]).cuda().transpose(1, 2)
            pr, _ = disVid(real)
            with torch.no_grad():
                fake, _ = gen.sample_videos(batch_size)
            pf, _ = disVid(fake)
            dis_vid_loss = loss(pr, torch.ones_like(pr)) + loss(pf, torch.zeros_like(pf))
            dis_vid_loss.backward()
            disVidOpt.step()
        genOpt.zero_grad()
        fakeVid, _ = gen.sample_videos(batch_size)
        fakeImg, _ = gen.sample_images(batch_size)
        pf_vid, _ = disVid(fakeVid)
        pf_img, _ = disImg(fakeImg)
        gen_loss = loss(pf_vid, torch.ones_like(pf_vid)) + loss(pf_img, torch.ones_like(pf_img))
        gen_loss.backward()
        genOpt.step()
        got = [dis_img_loss.item(), dis_vid_loss.item(), gen_loss.item()]
        assert np.allclose(got, g["losses"][it], rtol=2e-4, atol=0), (got, g["losses"][it])
    # the discriminators' .grad were filled by the generator step too, as in the reference (no freezing here)
    assert all(p.grad is not None for p in disVid.parameters())
    sd = gen.state_dict()
    assert float((sd["main.0.weight"].cpu() - torch.from_numpy(g["w1/gen/main.0.weight"])).abs().median()) < 2e-6


def test_reference_api_corner_cases():
    """video_len other than the constructor's (models/mocogan.py:271-272), linear=False (nn.Identity pre-net),
    eval-mode discriminator, and .squeeze() at batch 1 (models/mocogan.py:162 drops the batch dim too)."""
    seed_all(31)
    gen = G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8, linear=False)
    ref = M.Generator(1, 50, 0, 16, 16, ngf=8, mnist=True)
    ref.linear = torch.nn.Identity()
    ref.load_state_dict(gen.state_dict())
    assert not any(k.startswith("linear") for k in gen.state_dict())
    gen.cuda()
    seed_all(32)
    vid, _ = gen.sample_videos(3, video_len=8)
    seed_all(32)
    rvid, _ = ref.sample_videos(3, video_len=8)
    assert vid.shape == (3, 1, 8, 28, 28) and rel_err(vid.detach().cpu(), rvid.detach()) < TOL
    vid.sum().backward()
    rvid.sum().backward()
    assert robust_rel(gen.ode_fn.fn[0].weight.grad.cpu(), ref.ode_fn.fn[0].weight.grad) < 5e-3
    dis, rdis = G.VideoDiscriminator(1, ksize=2, ndf=8), M.VideoDisc(1, ksize=2, ndf=8)
    rdis.load_state_dict(dis.state_dict())
    dis.cuda()
    x = torch.rand(1, 1, 16, 28, 28, generator=torch.Generator().manual_seed(1))
    out, _ = dis(x.cuda())
    want, _ = rdis(x)
    assert out.shape == want.shape == (11, 2, 2) and rel_err(out.detach().cpu(), want.detach()) < TOL
    dis.eval(); rdis.eval()
    x4 = torch.rand(4, 1, 16, 28, 28, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        out, _ = dis(x4.cuda())
        want, _ = rdis(x4)
    assert rel_err(out.cpu(), want) < TOL


@pytest.mark.parametrize("B,rnn", [(1, False), (2, False), (5, False), (1, True), (5, True)])
def test_smallest_and_odd_batches_against_oracle(B, rnn):
    """Batch 1 (a single clip: the BatchNorm batches are its 16 frames / one image, the squeezed discriminator outputs lose
    the batch dimension, models/mocogan.py:162), 2 and an odd 5: sample_videos / sample_images frames at 1e-4 against the
    oracle on the same seeds, and one full training iteration (the joint generator pass falls back to the two calls where
    the rows do not split) with its three losses at 1e-4."""
    seed_all(61 + B)
    gen, dv, di = G.build_mnist(ngf=16, ndf=16)
    ogen, odv, odi = (M.build_mnist_odernn if rnn else M.build_mnist)(ngf=16, ndf=16)
    if rnn:      # (the ODE-RNN generator: adaptive solves with the error norm over the whole -- here tiny -- batch)
        gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=16)
    for m, o in zip((gen, dv, di), (ogen, odv, odi)):
        o.load_state_dict(m.state_dict())
    gen.cuda(); dv.cuda(); di.cuda()
    seed_all(70)
    vid, _ = gen.sample_videos(B)
    img, _ = gen.sample_images(B)
    seed_all(70)
    rvid, _ = ogen.sample_videos(B)
    rimg, _ = ogen.sample_images(B)
    assert vid.shape == rvid.shape == (B, 1, 16, 28, 28) and img.shape == rimg.shape == (B, 1, 28, 28)
    assert rel_err(vid.detach().cpu(), rvid.detach()) < TOL and rel_err(img.detach().cpu(), rimg.detach()) < TOL
    # (the two sampling calls above moved the BatchNorm running statistics of both generators alike)
    tr = G.GanTrainer(gen, dv, di)
    opts = M.make_optimizers(ogen, odv, odi)
    g = torch.Generator().manual_seed(6)
    imgs = [torch.rand(B, 1, 28, 28, generator=g) for _ in range(2)]
    vids = [torch.rand(B, 16, 1, 28, 28, generator=g) for _ in range(2)]
    seed_all(71)
    got = [float(v) for v in G.train_step(tr, [t.cuda() for t in imgs], [t.cuda() for t in vids])]
    seed_all(71)
    want = [float(v) for v in M.train_step(ogen, odv, odi, opts, imgs, vids)]
    assert np.allclose(got, want, rtol=1e-4, atol=0), (got, want)
    for (k, v), (_, w) in zip(gen.state_dict().items(), ogen.state_dict().items()):
        if "running_" in k:
            assert rel_err(v.cpu(), w) < 1e-4, k
        elif v.dtype == torch.int64:
            assert int(v) == int(w), k


def test_full_width_train_iterations_batch8_against_oracle():
    """BASELINE.json configs[0] (the reference's CPU-runnable case: batch 8, 16x1x28x28, ngf=ndf=64, rk4): two full
    training iterations (2 x [image-D, video-D] + G each) with FusedAdam / fused BCE against the oracle's
    train_step on stock torch.  Losses of every iteration at 1e-4 relative (iteration 1) / 1e-3 (iteration 2, which
    already depends on updated weights), BatchNorm running statistics at 5e-3, eval-mode samples at 3e-2."""
    seed_all(41)
    gen, dv, di = G.build_mnist()
    ogen, odv, odi = M.build_mnist()
    for m, o in zip((gen, dv, di), (ogen, odv, odi)):
        o.load_state_dict(m.state_dict())
    gen.cuda(); dv.cuda(); di.cuda()
    tr = G.GanTrainer(gen, dv, di)
    opts = M.make_optimizers(ogen, odv, odi)
    B = 8
    g = torch.Generator().manual_seed(5)
    for it in range(2):
        imgs = [torch.rand(B, 1, 28, 28, generator=g) for _ in range(2)]
        vids = [torch.rand(B, 16, 1, 28, 28, generator=g) for _ in range(2)]
        seed_all(100 + it)
        got = [float(v) for v in G.train_step(tr, [t.cuda() for t in imgs], [t.cuda() for t in vids])]
        seed_all(100 + it)
        want = [float(v) for v in M.train_step(ogen, odv, odi, opts, imgs, vids)]
        assert np.allclose(got, want, rtol=1e-4 if it == 0 else 1e-3, atol=0), (it, got, want)
    for m, o in zip((gen, dv, di), (ogen, odv, odi)):
        for (k, v), (_, w) in zip(m.state_dict().items(), o.state_dict().items()):
            if "running_" in k:
                # after the first Adam step weights whose gradient is within rounding of zero have moved by +-lr in
                # either direction (Adam normalises the step), so the second iteration's batch statistics differ
                # at the 1e-3 level between ANY two fp32 implementations
                assert rel_err(v.cpu(), w) < 5e-3, k
            elif v.dtype == torch.int64:
                assert int(v) == int(w), k
    gen.eval(); ogen.eval()
    seed_all(7)
    with torch.no_grad():
        ev, _ = gen.sample_videos(4)
    seed_all(7)
    with torch.no_grad():
        rev, _ = ogen.sample_videos(4)
    # two Adam steps in, a handful of +-lr weight flips (see above) are visible at the 1e-2 level in eval-mode frames
    assert rel_err(ev.cpu(), rev) < 3e-2


def test_checkpoint_interchange_with_stock_torch(tmp_path):
    """the checkpoint format of mnist_moco_ode.py:175-190 ({'epoch', 'model_state_dict': [...],
    'optimizer_state_dict': [...]}, torch.save) written from the HIP path loads into stock torch modules + torch.optim.Adam
    (and back), and training continues identically from it."""
    g = golden("train_mnist_tiny.npz")
    s = int(g["seed"])
    gen, dv, di = G.build_mnist(ngf=8, ndf=8)
    for m, p in ((gen, "gen"), (dv, "vid"), (di, "img")):
        load_sd(m, g, f"w0/{p}")
        m.cuda()
    tr = G.GanTrainer(gen, dv, di)
    imgs = [_f32(g[f"real_img/0/{i}"]) for i in range(2)]
    vids = [_f32(g[f"real_vid/0/{i}"]) for i in range(2)]
    seed_all(s + 1)
    G.train_step(tr, [t.cuda() for t in imgs], [t.cuda() for t in vids])
    path = tmp_path / "state_normal0.ckpt"
    torch.save({"epoch": 0,
                "model_state_dict": [gen.state_dict(), dv.state_dict(), di.state_dict()],
                "optimizer_state_dict": [tr.gen_opt.state_dict(), tr.vid_opt.state_dict(), tr.img_opt.state_dict()]}, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    ogen, odv, odi = M.build_mnist(ngf=8, ndf=8)
    opts = M.make_optimizers(ogen, odv, odi)
    for m, sd in zip((ogen, odv, odi), ck["model_state_dict"]):
        m.load_state_dict(sd)                                   # strict: identical keys
    for o, sd in zip(opts, ck["optimizer_state_dict"]):
        o.load_state_dict(sd)
    assert int(opts[0].state[ogen.main[0].weight]["step"]) == 1 and ogen.recurrent.weight_ih not in opts[0].state
    # continue one iteration on both sides from the checkpoint
    imgs1 = [_f32(g[f"real_img/1/{i}"]) for i in range(2)]
    vids1 = [_f32(g[f"real_vid/1/{i}"]) for i in range(2)]
    seed_all(s + 2)
    got = [float(v) for v in G.train_step(tr, [t.cuda() for t in imgs1], [t.cuda() for t in vids1])]
    seed_all(s + 2)
    want = [float(v) for v in M.train_step(ogen, odv, odi, opts, imgs1, vids1)]
    assert np.allclose(got, want, rtol=2e-4, atol=0), (got, want)
    assert np.allclose(got, g["losses"][1], rtol=2e-4, atol=0)
    # and the other direction: a stock-torch optimiser state loads into FusedAdam
    tr.gen_opt.load_state_dict(opts[0].state_dict())
    assert int(tr.gen_opt.state[gen.main[0].weight]["step"]) == 2


def test_data_parallel_step_path_on_one_rank_nccl():
    """The multi-GPU optimiser path (flat bucket -> RCCL all-reduce -> Adam reading the reduced bucket with 1/world
    folded in) exercised on a single-rank NCCL group with the world size forced to 2: with one rank the all-reduce is
    the identity, so every gradient is halved -- compared against a plain step fed the same halved gradients."""
    import os
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        g = golden("train_mnist_tiny.npz")
        s = int(g["seed"])
        nets = []
        for _ in range(2):
            gen, dv, di = G.build_mnist(ngf=8, ndf=8)
            for m, p in ((gen, "gen"), (dv, "vid"), (di, "img")):
                load_sd(m, g, f"w0/{p}")
                m.cuda()
            nets.append((gen, dv, di))
        tr_dp = G.GanTrainer(*nets[0], pair_d_passes=False)   # (the reference sequence below is the two-pass form)
        tr_dp.world = 2                                   # pretend there is a second rank
        tr_ref = G.GanTrainer(*nets[1])
        real_img = _f32(g["real_img/0/0"]).cuda()
        seed_all(s + 1)
        l_dp = tr_dp.d_image_step(real_img)
        # reference: same forward/backward, then Adam on gradients scaled by 1/2
        seed_all(s + 1)
        tr_ref.img_opt.zero_grad()
        pr, _ = tr_ref.dis_img(real_img)
        with torch.no_grad():
            fake, _ = tr_ref.gen.sample_images(real_img.shape[0])
        pf, _ = tr_ref.dis_img(fake)
        loss = G.bce_with_logits_const(pr, 1.0) + G.bce_with_logits_const(pf, 0.0)
        loss.backward()
        tr_ref.img_opt.step(gscale=0.5)
        assert abs(float(l_dp) - float(loss.detach())) < 1e-6
        for (k, a), (_, b) in zip(nets[0][2].state_dict().items(), nets[1][2].state_dict().items()):
            assert torch.equal(a, b), k
    finally:
        if created:
            dist.destroy_process_group()


def test_direct_gradient_arena_is_bitwise_the_autograd_accumulation():
    """GanTrainer(direct_grads=True): the backward kernels store / add parameter gradients straight into a per-network
    arena (p.grad = view) instead of returning them to autograd, which would add the two passes of a step with one
    elementwise kernel per tensor.  Same kernels, same operands, one fp32 add either way -> identical bits."""
    g = golden("train_mnist_tiny.npz")
    s = int(g["seed"])
    runs = []
    for direct in (True, False):
        gen, dv, di = G.build_mnist(ngf=8, ndf=8)
        for m, p in ((gen, "gen"), (dv, "vid"), (di, "img")):
            load_sd(m, g, f"w0/{p}")
            m.cuda()
        tr = G.GanTrainer(gen, dv, di, direct_grads=direct)
        assert bool(tr.arenas) == direct
        losses = []
        for it in range(2):
            imgs = [_f32(g[f"real_img/{it}/{i}"]).cuda() for i in range(2)]
            vids = [_f32(g[f"real_vid/{it}/{i}"]).cuda() for i in range(2)]
            seed_all(s + 1 + it)
            losses.append([float(v) for v in tr.step(imgs, vids)])
        if direct:      # gradients live in the arena; the unused GRU cell never received one
            assert gen.main[0].weight.grad.data_ptr() == tr.arenas[id(gen)].views[gen.main[0].weight].data_ptr()
            assert gen.recurrent.weight_ih.grad is None
        runs.append((losses, [p.detach().clone() for m in (gen, dv, di) for p in m.parameters()],
                     [p.grad.clone() for m in (gen,) for p in m.parameters() if p.grad is not None]))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)
    for a, b in zip(runs[0][2], runs[1][2]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("step_size", [0.05, 0.04])
def test_solver_grid_option_step_size_against_oracle(step_size):
    """`gen.ode_step_size = h` = torchdiffeq's options={'step_size': h}: rk4 on the solver's own grid (h = 0.05: the
    "RK4 20 steps" BASELINE.json words for config 1), outputs interpolated linearly, the adjoint solved per output
    interval on the reversed-span grid.  The reference never passes this option and torchdiffeq is not installed
    here, so this case is PARITY UNPINNED: it is checked against the oracle's restatement of the published
    FixedGridODESolver (oracle/ode_ref.py:fixed_grid_solve), motion latent first (tight), then frames and gradients."""
    g = golden("gen_mnist_tiny.npz")
    s = int(g["seed"])
    gen = G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8)
    load_sd(gen, g, "w")
    gen.cuda()
    gen.ode_step_size = step_size
    ogen = M.Generator(1, 50, 0, 16, 16, ngf=8, mnist=True)
    ogen.load_state_dict({k: v.detach().cpu().clone() for k, v in gen.state_dict().items()})
    ogen.ode_options = {"step_size": step_size}
    seed_all(s + 1)
    vid, _ = gen.sample_videos(4)
    seed_all(s + 2)
    img, _ = gen.sample_images(4)
    seed_all(s + 1)
    ovid, _ = ogen.sample_videos(4)
    seed_all(s + 2)
    oimg, _ = ogen.sample_images(4)
    assert rel_err(vid.detach().cpu(), ovid.detach()) < TOL
    assert rel_err(img.detach().cpu(), oimg.detach()) < TOL
    # and it is a different solve from the default grid (15 steps on the output times)
    assert rel_err(vid.detach().cpu(), g["videos"]) > 1e-6
    wv, wi = _f32(g["wv"]), _f32(g["wi"])
    ((vid * wv.cuda()).sum() + (img * wi.cuda()).sum()).backward()
    ((ovid * wv).sum() + (oimg * wi).sum()).backward()
    ref = dict(ogen.named_parameters())
    for k, p in gen.named_parameters():
        if ref[k].grad is None:
            assert p.grad is None, k
            continue
        assert robust_rel(p.grad.cpu(), ref[k].grad) < 5e-3, (k, robust_rel(p.grad.cpu(), ref[k].grad))
    for k in ("ode_fn.fn.0.weight", "ode_fn.fn.2.weight", "linear.0.weight", "linear.2.bias"):
        assert rel_err(dict(gen.named_parameters())[k].grad.cpu(), ref[k].grad) < 2e-3, k


@pytest.mark.parametrize("tag,mnist", [("ucf_tiny", False), ("mnist_tiny", True)])
def test_dopri5_method_against_oracle(tag, mnist):
    """`gen.ode_method = "dopri5"` (BASELINE configs[3] words the UCF run "dopri5 adaptive"; the reference code itself
    passes method='rk4'): torchdiffeq's adaptive solver over the 16 output times with dense-output interpolation, on
    the device in one launch.  PARITY UNPINNED (torchdiffeq is not installed; the reference holds no fixture):
    checked against the oracle's restatement (oracle/ode_ref.py:dopri5_solve + odeint_adjoint, which integrates the
    adjoint adaptively, as the device does: csrc/odernn_valu.hip, mixed norm over the whole batch)."""
    g = golden(f"gen_{tag}.npz")
    s = int(g["seed"])
    if mnist:
        gen = G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8)
        ogen = M.Generator(1, 50, 0, 16, 16, ngf=8, mnist=True, ode_method="dopri5")
    else:
        gen = G.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8)
        ogen = M.Generator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8, mnist=False, ode_method="dopri5")
    load_sd(gen, g, "w")
    gen.cuda()
    gen.ode_method = "dopri5"
    ogen.load_state_dict({k: v.detach().cpu().clone() for k, v in gen.state_dict().items()})
    seed_all(s + 1)
    vid, _ = gen.sample_videos(2)
    seed_all(s + 2)
    img, _ = gen.sample_images(3)
    seed_all(s + 1)
    ovid, _ = ogen.sample_videos(2)
    seed_all(s + 2)
    oimg, _ = ogen.sample_images(3)
    assert rel_err(vid.detach().cpu(), ovid.detach()) < TOL
    # sample_images: the reference integrates all B*T*2 trajectories under ONE step-size controller (error norm over
    # the whole batch); the build integrates only the B selected ones, so the accepted steps differ -- both solutions
    # are within the solver tolerance (1e-7) of the exact flow, hence the same bound
    assert rel_err(img.detach().cpu(), oimg.detach()) < TOL
    plan = gen._pool.plans[(2, 16, False)][0]
    assert 3 <= int(plan._nsteps[0]) < 200          # adaptive: a handful of accepted + rejected trial steps
    rng = torch.Generator().manual_seed(5)
    wv, wi = torch.randn(vid.shape, generator=rng), torch.randn(img.shape, generator=rng)
    ((vid * wv.cuda()).sum() + (img * wi.cuda()).sum()).backward()
    ((ovid * wv).sum() + (oimg * wi).sum()).backward()
    ref = dict(ogen.named_parameters())
    for k, p in gen.named_parameters():
        if ref[k].grad is None:
            assert p.grad is None, k
            continue
        assert robust_rel(p.grad.cpu(), ref[k].grad) < 5e-3, (k, robust_rel(p.grad.cpu(), ref[k].grad))
    for k in ("ode_fn.fn.0.weight", "ode_fn.fn.2.weight", "linear.0.weight"):
        assert rel_err(dict(gen.named_parameters())[k].grad.cpu(), ref[k].grad) < 2e-3, k


def test_gradient_arena_without_prenet_block():
    """linear=False (nn.Identity pre-net): the adjoint kernel then writes only the ODEFunc block of its gradient
    vector, and the arena hands it a base pointer 2128 floats BEFORE that block's first tensor -- nothing in front of
    the block may be touched, and the result must equal the stock autograd path bit for bit."""
    runs = []
    for direct in (True, False):
        seed_all(77)
        gen = G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8, linear=False)
        dv, di = G.VideoDiscriminator(1, ksize=2, ndf=8), G.PatchImageDiscriminator(1, ndf=8)
        for m in (gen, dv, di):
            m.cuda()
        tr = G.GanTrainer(gen, dv, di, direct_grads=direct)
        rng = torch.Generator().manual_seed(5)
        imgs = [torch.rand(4, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
        vids = [torch.rand(4, 16, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
        seed_all(78)
        losses = [float(v) for v in tr.step(imgs, vids)]
        if direct:
            arena = tr.arenas[id(gen)]
            # the ODEFunc block is the arena's tail; everything before it belongs to decoder tensors that received
            # their own gradients -- compared below against the stock path
            assert arena.params[-4] is gen.ode_fn.fn[0].weight and arena.params[-1] is gen.ode_fn.fn[2].bias
        runs.append((losses, [p.detach().clone() for m in (gen, dv, di) for p in m.parameters()],
                     [None if p.grad is None else p.grad.clone() for p in gen.parameters()]))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)
    for a, b in zip(runs[0][2], runs[1][2]):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
