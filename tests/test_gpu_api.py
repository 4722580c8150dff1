"""GPU: the rest of the reference's public surface and the robustness rules of the drop-in modules --
sample_z_m / sample_z_video (models/mocogan.py:217-269, models/mocogan_ode.py:133-148), eval-mode backward,
plan leases (a late backward raises instead of reading recycled buffers), the device-resident Rot-MNIST feeder."""
import numpy as np
import pytest
import torch

from conftest import golden, load_sd, rel_err, seed_all

import gan_ode_amd as G
from oracle import mocogan_ref as M

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _pair(mnist=True, ngf=8):
    if mnist:
        gen = G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=ngf)
        ref = M.Generator(1, 50, 0, 16, 16, ngf=ngf, mnist=True)
    else:
        gen = G.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=ngf)
        ref = M.Generator(3, 50, 0, 16, 16, dim_hidden=16, ngf=ngf, mnist=False)
    ref.load_state_dict(gen.state_dict())
    return gen.cuda(), ref


@pytest.mark.parametrize("mnist", [True, False])
def test_sample_z_m_and_sample_z_video_against_oracle(mnist):
    seed_all(3)
    gen, ref = _pair(mnist)
    seed_all(4)
    zm = gen.sample_z_m(5)
    seed_all(4)
    rzm = ref.sample_z_m(5)
    assert zm.shape == (5 * 16, 16) and zm.is_cuda
    assert rel_err(zm.detach().cpu(), rzm.detach()) < TOL
    # row n*T + t: frame 0 of every video is the pre-net output itself (Appendix A of SURVEY.md)
    w = torch.randn(zm.shape, generator=torch.Generator().manual_seed(1))
    (zm * w.cuda()).sum().backward()
    (rzm * w).sum().backward()
    for k in ("ode_fn.fn.0.weight", "ode_fn.fn.0.bias", "ode_fn.fn.2.weight", "ode_fn.fn.2.bias",
              "linear.0.weight", "linear.0.bias", "linear.2.weight", "linear.2.bias"):
        a, b = dict(gen.named_parameters())[k].grad, dict(ref.named_parameters())[k].grad
        assert rel_err(a.cpu(), b) < 2e-4, k
    # video_len argument and the RNG order of sample_z_video (NumPy content first, then torch.randn)
    seed_all(6)
    z, labels = gen.sample_z_video(3, 8)
    seed_all(6)
    rz, rlabels = ref.sample_z_video(3, 8)
    assert z.shape == (24, 66) and isinstance(labels, np.ndarray) and np.array_equal(labels, rlabels)
    assert torch.equal(z[:, :50].cpu(), rz[:, :50])
    assert rel_err(z[:, 50:].detach().cpu(), rz[:, 50:].detach()) < TOL
    assert gen.sample_z_categ(4)[0] is None
    # decoding those rows with the reference's own call sequence gives sample_videos' frames
    seed_all(7)
    vid, _ = gen.sample_videos(3, 8)
    seed_all(7)
    rvid, _ = ref.sample_videos(3, 8)
    assert rel_err(vid.detach().cpu(), rvid.detach()) < TOL


def test_sample_videos_labels_are_float64_zeros_and_cached():
    seed_all(1)
    gen, _ = _pair()
    with torch.no_grad():
        _, l1 = gen.sample_videos(4)
        _, l2 = gen.sample_videos(4)
    assert l1.dtype == torch.float64 and l1.shape == (4,) and float(l1.abs().sum()) == 0.0
    assert l1.data_ptr() == l2.data_ptr()


def test_eval_mode_backward_uses_running_statistics():
    """Advisor finding r1: an eval-mode forward normalises with the running statistics, which are constants -- its
    backward must not subtract the batch-mean terms.  Generator and both discriminators against the oracle in eval
    mode (input gradients and parameter gradients)."""
    g = golden("train_mnist_tiny.npz")
    gen, dv, di = G.build_mnist(ngf=8, ndf=8)
    ogen, odv, odi = M.build_mnist(ngf=8, ndf=8)
    for m, o, p in ((gen, ogen, "gen"), (dv, odv, "vid"), (di, odi, "img")):
        load_sd(m, g, f"w1/{p}")               # trained one step: running statistics differ from (0, 1)
        o.load_state_dict({k: v.cpu().clone() for k, v in m.state_dict().items()})
        m.cuda().eval(); o.eval()
    seed_all(9)
    vid, _ = gen.sample_videos(3)
    pv, _ = dv(vid)
    x = torch.rand(3, 1, 28, 28, generator=torch.Generator().manual_seed(2))
    xg = x.cuda().requires_grad_(True)
    pi, _ = di(xg)
    (G.bce_with_logits_const(pv, 1.0) + G.bce_with_logits_const(pi, 0.0)).backward()
    seed_all(9)
    rvid, _ = ogen.sample_videos(3)
    rpv, _ = odv(rvid)
    xr = x.clone().requires_grad_(True)
    rpi, _ = odi(xr)
    bce = torch.nn.BCEWithLogitsLoss()
    (bce(rpv, torch.ones_like(rpv)) + bce(rpi, torch.zeros_like(rpi))).backward()
    assert rel_err(vid.detach().cpu(), rvid.detach()) < TOL and rel_err(pv.detach().cpu(), rpv.detach()) < TOL
    assert rel_err(xg.grad.cpu(), xr.grad) < 5e-4
    for m, o in ((gen, ogen), (dv, odv), (di, odi)):
        for (k, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
            if q.grad is None:
                assert p.grad is None, k
            else:
                assert rel_err(p.grad.cpu(), q.grad) < 2e-3, (k, rel_err(p.grad.cpu(), q.grad))
    # running statistics untouched by eval-mode passes
    for (k, v), (_, w) in zip(gen.state_dict().items(), ogen.state_dict().items()):
        if "running_" in k or "num_batches" in k:
            assert torch.equal(v.cpu(), w), k


def test_late_backward_on_a_recycled_plan_raises_and_dropped_losses_release_plans():
    seed_all(5)
    dis = G.PatchImageDiscriminator(1, ndf=8).cuda()
    x = torch.rand(2, 1, 28, 28).cuda()
    first, _ = dis(x)
    kept = [dis(x)[0] for _ in range(G.modules._Pool.MAX_PLANS)]     # all plans of this shape checked out, oldest reused
    with pytest.raises(RuntimeError, match="reused by a later forward"):
        first.sum().backward()
    kept[-1].sum().backward()                                          # the newest forwards are intact
    # a forward whose loss is dropped gives its plan back when the graph dies: no growth beyond one plan
    dis2 = G.PatchImageDiscriminator(1, ndf=8).cuda()
    for _ in range(10):
        out, _ = dis2(x)
        del out
    assert len(dis2._pool.plans[tuple(x.shape)]) == 1
    # second backward through a retained graph after the plan was reused
    out, _ = dis2(x)
    out.sum().backward(retain_graph=True)
    dis2(x)
    with pytest.raises(RuntimeError, match="reused by a later forward"):
        out.sum().backward()


def test_rejected_constructor_arguments():
    with pytest.raises(NotImplementedError, match="dim_hidden"):
        G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, dim_hidden=32)
    with pytest.raises(NotImplementedError, match="dim_hidden"):
        G.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=8)
    with pytest.raises(NotImplementedError, match="ode_fn"):
        G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ode_fn=lambda dim, dim_hidden: torch.nn.Linear(dim, dim))


def test_invalidate_packs_after_a_write_behind_autograds_back():
    seed_all(2)
    dis = G.PatchImageDiscriminator(1, ndf=8).cuda()
    x = torch.rand(2, 1, 28, 28).cuda()
    with torch.no_grad():
        a, _ = dis(x)
        # .data write: no version bump -> packed panels are stale.  (An additive change: the stack is invariant to a
        # positive rescaling of this layer -- LeakyReLU is homogeneous and the next layer is batch-normalised.)
        dis.main[1].weight.data.add_(0.05)
        dis.invalidate_packs()
        b, _ = dis(x)
        ref = M.PatchImageDisc(1, ndf=8)
        ref.load_state_dict({k: v.cpu() for k, v in dis.state_dict().items()})
        want, _ = ref(x.cpu())
    assert rel_err(b.cpu(), want) < TOL and rel_err(a.cpu(), want) > 1e-2


def test_device_resident_rot_mnist_feeds_the_trainer_bit_identically():
    """SURVEY 8(f) rank 3 on the device: RotMnistOnDevice(device='cuda') batches (gathered on the GPU, read in place
    through strides by the first Conv3d) drive GanTrainer.step exactly like the same batches uploaded from host
    tensors: identical losses and weights, bit for bit."""
    from gan_ode_amd.data import RotMnistOnDevice
    X = torch.rand(40, 16, 1, 28, 28, generator=torch.Generator().manual_seed(0))
    runs = []
    for device in ("cuda", "cpu"):
        seed_all(13)
        gen, dv, di = G.build_mnist(ngf=8, ndf=8)
        gen.cuda(); dv.cuda(); di.cuda()
        tr = G.GanTrainer(gen, dv, di)
        feed = RotMnistOnDevice(X, device=device, seed=7)
        vs, ims = feed.videos(8), feed.images(8)
        losses = []
        for it in range(2):
            imgs = [next(ims) for _ in range(2)]
            vids = [next(vs) for _ in range(2)]
            assert all(t.device.type == device for t in imgs + vids)
            seed_all(100 + it)
            losses.append([float(v) for v in tr.step([t.cuda() for t in imgs], [t.cuda() for t in vids])])
        runs.append((losses, [p.detach().clone() for m in (gen, dv, di) for p in m.parameters()]))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


def test_image_discriminator_side_stream_is_bitwise_the_serial_schedule():
    """GanTrainer(overlap_image_d=True) runs the image-discriminator step on a side stream next to the video step; both
    streams execute the same kernels in the same order, generator work (and with it the BatchNorm running statistics)
    stays on the caller's stream: losses, weights, running statistics and Adam state are bit-identical."""
    runs = []
    for overlap in (True, False):
        seed_all(23)
        gen, dv, di = G.build_mnist(ngf=16, ndf=16)
        gen.cuda(); dv.cuda(); di.cuda()
        tr = G.GanTrainer(gen, dv, di, overlap_image_d=overlap)
        assert (tr._side is not None) == overlap
        rng = torch.Generator().manual_seed(9)
        losses = []
        for it in range(3):
            imgs = [torch.rand(8, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            vids = [torch.rand(8, 16, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            seed_all(200 + it)
            losses.append([float(v) for v in tr.step(imgs, vids)])
        torch.cuda.synchronize()
        state = [v.detach().clone() for m in (gen, dv, di) for v in m.state_dict().values()]
        state += [tr.img_opt.state[p]["exp_avg_sq"].clone() for p in di.parameters()]
        runs.append((losses, state))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


def test_graph_replayed_iterations_are_bitwise_the_eager_ones():
    """GanTrainer(graph=True): two eager iterations, then the whole iteration (both streams, the backward passes, five
    Adam launches) is captured in a HIP graph and replayed; the host still draws the noise from the NumPy / torch CPU
    generators in the reference's order (HostFeed) and feeds Adam's step-dependent coefficients.  Losses, weights,
    BatchNorm buffers and optimiser state are bit-identical to the eager schedule over six iterations, eager sampling
    between replays sees the updated weights, and the checkpointed step counts are right."""
    runs = []
    for graph in (True, False):
        seed_all(31)
        gen, dv, di = G.build_mnist(ngf=16, ndf=16)
        gen.cuda(); dv.cuda(); di.cuda()
        tr = G.GanTrainer(gen, dv, di, graph=graph)
        rng = torch.Generator().manual_seed(10)
        losses, samples = [], []
        for it in range(6):
            imgs = [torch.rand(8, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            vids = [torch.rand(8, 16, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            seed_all(300 + it)
            losses.append([float(v) for v in tr.step(imgs, vids)])
            if it == 3:                      # eager use of the same modules between two replays
                seed_all(77)
                with torch.no_grad():
                    samples.append(gen.sample_videos(3)[0].clone())
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        state = [v.detach().clone() for m in (gen, dv, di) for v in m.state_dict().values()]
        state += [tr.gen_opt.state[p]["exp_avg"].clone() for p in gen.parameters() if p in tr.gen_opt.state and tr.gen_opt.state[p]]
        steps = [int(tr.img_opt.state[next(di.parameters())]["step"]), int(tr.gen_opt.state[gen.main[0].weight]["step"])]
        runs.append((losses, state, samples, steps))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(runs[0][2][0], runs[1][2][0])
    assert runs[0][3] == runs[1][3] == [12, 6]


@pytest.mark.parametrize("odernn", [False, True])
def test_graph_capture_is_self_contained_with_respect_to_weight_packs(odernn):
    """Advisor r2: an eager sample_videos() between the last warm-up iteration and the capture iteration leaves the
    generator's packed panels marked fresh; without the invalidation in _capture() the graph would hold no pack launch
    for them and every replay would decode with the weights of iteration 2.  Eager passes between iteration index 1 and
    2 (generator and discriminator), and a load_state_dict of perturbed weights between two replays: still bit-identical
    to the eager schedule.  odernn=True: the same with the ODE-RNN generator, which now has a gradient arena and may be
    captured."""
    runs = []
    for graph in (True, False):
        seed_all(35)
        gen, dv, di = G.build_mnist(ngf=16, ndf=16)
        if odernn:
            gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=16)
        gen.cuda(); dv.cuda(); di.cuda()
        tr = G.GanTrainer(gen, dv, di, graph=graph)
        rng = torch.Generator().manual_seed(12)
        losses = []
        for it in range(6):
            imgs = [torch.rand(8, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            vids = [torch.rand(8, 16, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            if it == 2:                      # eager use right before the capture iteration
                seed_all(78)
                with torch.no_grad():
                    v, _ = gen.sample_videos(3)
                    dv(v); di(gen.sample_images(3)[0])
            if it == 4:                      # weights replaced between two replays
                for m in (gen, dv, di):
                    sd = {k: (t + 0.01 if t.dtype == torch.float32 and "running" not in k else t) for k, t in m.state_dict().items()}
                    m.load_state_dict(sd)
            seed_all(500 + it)
            losses.append([float(x) for x in tr.step(imgs, vids)])
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        runs.append((losses, [t.detach().clone() for m in (gen, dv, di) for t in m.state_dict().values()]))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("odernn", [False, True])
def test_prefetched_latents_are_bitwise_the_call_by_call_ones(odernn):
    """GanTrainer(prefetch_latents=True) (default): every sample_images / sample_videos call of an iteration is announced
    up front, the host draws are made in the same order and all latent solves go out at once on a side stream (ODE-RNN: one
    gode_odernn_fwd_multi launch, and the G step's two adjoints one gode_odernn_bwd_multi launch).  Same kernels on the
    same inputs: losses, weights, BatchNorm buffers and Adam state are bit-identical to the call-by-call schedule."""
    runs = []
    for prefetch in (True, False):
        seed_all(53)
        gen, dv, di = G.build_mnist(ngf=16, ndf=16)
        if odernn:
            gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=16)
        gen.cuda(); dv.cuda(); di.cuda()
        tr = G.GanTrainer(gen, dv, di, prefetch_latents=prefetch)
        rng = torch.Generator().manual_seed(14)
        losses = []
        for it in range(3):
            imgs = [torch.rand(8, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            vids = [torch.rand(8, 16, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            seed_all(600 + it)
            losses.append([float(x) for x in tr.step(imgs, vids)])
        torch.cuda.synchronize()
        assert not gen.__dict__.get("_prefetched")
        state = [t.detach().clone() for m in (gen, dv, di) for t in m.state_dict().values()]
        state += [tr.gen_opt.state[p]["exp_avg_sq"].clone() for p in gen.parameters() if tr.gen_opt.state.get(p)]
        runs.append((losses, state))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)
    # a call that does not match the announced order is refused, and the queue can be dropped
    gen.prefetch_latents([("videos", 4), ("images", 4)])
    with pytest.raises(RuntimeError, match="promised"):
        gen.sample_images(4)
    gen.discard_prefetched()
    with torch.no_grad():
        assert gen.sample_images(4)[0].shape == (4, 1, 28, 28)


def test_second_trainer_on_the_same_networks_does_not_detach_the_first_ones_arenas():
    """Advisor r2: GanTrainer.__init__ binds module._gode_arena; a second trainer on the same networks used to leave the
    first one reducing an arena nobody writes.  Each trainer now re-binds its own arenas at the start of every optimiser
    step: interleaved steps of two trainers fill the stepping trainer's arena (p.grad is a view of it)."""
    seed_all(36)
    gen, dv, di = G.build_mnist(ngf=8, ndf=8)
    gen.cuda(); dv.cuda(); di.cuda()
    tr1 = G.GanTrainer(gen, dv, di)
    tr2 = G.GanTrainer(gen, dv, di)
    x = torch.rand(4, 1, 28, 28).cuda()
    v = torch.rand(4, 16, 1, 28, 28).cuda()
    for tr in (tr1, tr2, tr1):
        seed_all(37)
        tr.d_image_step(x); tr.d_video_step(v); tr.g_step(4)
        for m in (gen, dv, di):
            a = tr.arenas[id(m)]
            assert m._gode_arena is a
            live = [p for p in m.parameters() if p.grad is not None]
            assert live and all(p.grad.data_ptr() == a.views[p].data_ptr() for p in live)
            assert float(a.flat.abs().sum()) > 0


@pytest.mark.parametrize("which", ["video", "image"])
def test_paired_discriminator_pass_equals_two_passes(which):
    """forward_pair(real, fake): ONE pass over [real; fake] with per-group BatchNorm statistics against the two separate
    calls of the reference loop (mnist_moco_ode.py:119-124,137-143) on the same module: logits and running statistics to
    fp32 rounding (1e-5), weight gradients to fp32 summation order (one reduction over both groups instead of store +
    add), running statistics updated real-then-fake, num_batches_tracked += 2;
    against the oracle at the usual tolerances; eval mode too."""
    seed_all(41)
    if which == "video":
        dis, ref = G.VideoDiscriminator(1, ksize=2, ndf=16), M.VideoDisc(1, ksize=2, ndf=16)
        real = torch.rand(6, 16, 1, 28, 28, generator=torch.Generator().manual_seed(1)).transpose(1, 2)   # strided view, as the loop feeds it
        fake = torch.rand(6, 1, 16, 28, 28, generator=torch.Generator().manual_seed(2)) * 2 - 1
    else:
        dis, ref = G.PatchImageDiscriminator(1, ndf=16), M.PatchImageDisc(1, ndf=16)
        real = torch.rand(6, 1, 28, 28, generator=torch.Generator().manual_seed(1))
        fake = torch.rand(6, 1, 28, 28, generator=torch.Generator().manual_seed(2)) * 2 - 1
    ref.load_state_dict(dis.state_dict())
    import copy
    dis2 = copy.deepcopy(dis)
    dis.cuda(); dis2.cuda()
    rc, fc = real.cuda(), fake.cuda()
    (pr, _), (pf, _) = dis.forward_pair(rc, fc)
    loss = G.bce_with_logits_pair(pr, 1.0, pf, 0.0)
    loss.backward()
    qr, _ = dis2(rc)
    qf, _ = dis2(fc)
    loss2 = G.bce_with_logits_pair(qr, 1.0, qf, 0.0)
    loss2.backward()
    assert pr.shape == qr.shape and pf.shape == qf.shape
    # (not bitwise: the paired stack reads materialised activations where the single pass fuses BatchNorm into the load,
    # and tile / split-K choices depend on the row count)
    assert rel_err(pr.detach().cpu(), qr.detach().cpu()) < 1e-5 and rel_err(pf.detach().cpu(), qf.detach().cpu()) < 1e-5
    assert abs(float(loss) - float(loss2)) < 1e-6
    for (k, p), (_, q) in zip(dis.named_parameters(), dis2.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad.cpu()) < 2e-5, k
    for (k, v), (_, w) in zip(dis.state_dict().items(), dis2.state_dict().items()):
        assert rel_err(v.cpu().double(), w.cpu().double()) < 1e-5, k    # weights untouched, running statistics + counters agree
    # the oracle's two calls
    bce = torch.nn.BCEWithLogitsLoss()
    orr, _ = ref(real)
    orf, _ = ref(fake)
    ol = bce(orr, torch.ones_like(orr)) + bce(orf, torch.zeros_like(orf))
    ol.backward()
    assert rel_err(pr.detach().cpu(), orr.detach()) < TOL and rel_err(pf.detach().cpu(), orf.detach()) < TOL
    assert abs(float(loss) - float(ol)) / abs(float(ol)) < TOL
    for (k, p), (_, q) in zip(dis.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < 5e-4, k
    for (k, v), (_, w) in zip(dis.state_dict().items(), ref.state_dict().items()):
        if "running_" in k:
            assert rel_err(v.cpu(), w) < TOL, k
        if "num_batches" in k:
            assert int(v) == int(w) == 2, k
    # the joint form the trainer uses, and eval mode (running statistics for both groups)
    joint = dis.forward_pair_joint(rc, fc)
    lj = G.bce_with_logits_halves(joint, 1.0, 0.0)
    (pr2, _), (pf2, _) = dis2.forward_pair(rc, fc)
    assert abs(float(lj) - float(G.bce_with_logits_pair(pr2, 1.0, pf2, 0.0))) < 1e-6
    ref.load_state_dict({k: v.cpu() for k, v in dis.state_dict().items()})     # (dis has seen two more training passes)
    dis.eval(); ref.eval()
    with torch.no_grad():
        (er, _), (ef, _) = dis.forward_pair(rc, fc)
        wr, _ = ref(real)
        wf, _ = ref(fake)
    assert rel_err(er.cpu(), wr) < TOL and rel_err(ef.cpu(), wf) < TOL


def test_trainer_with_paired_passes_matches_the_two_pass_schedule():
    """GanTrainer(pair_d_passes=True) (default) against pair_d_passes=False over three iterations: the same losses to
    fp32 rounding of the weight-gradient sums (and the same statistics), on the full-width MNIST networks at batch 8."""
    runs = []
    for pair in (True, False):
        seed_all(45)
        gen, dv, di = G.build_mnist()
        gen.cuda(); dv.cuda(); di.cuda()
        tr = G.GanTrainer(gen, dv, di, pair_d_passes=pair)
        rng = torch.Generator().manual_seed(11)
        losses = []
        for it in range(2):
            imgs = [torch.rand(8, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            vids = [torch.rand(8, 16, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
            seed_all(400 + it)
            losses.append([float(v) for v in tr.step(imgs, vids)])
        runs.append((losses, {k: v.detach().clone() for m in (dv, di) for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}))
    assert np.allclose(runs[0][0][0], runs[1][0][0], rtol=1e-5, atol=0), runs          # first iteration: same weights
    assert np.allclose(runs[0][0][1], runs[1][0][1], rtol=1e-3, atol=0), runs          # after Adam steps: sign-like updates
    for k in runs[0][1]:
        assert rel_err(runs[0][1][k].cpu().double(), runs[1][1][k].cpu().double()) < 5e-3, k


@pytest.mark.parametrize("kind", ["mnist", "odernn", "ucf"])
@pytest.mark.parametrize("images_first", [False, True])
def test_joint_generator_pass_equals_the_two_calls(kind, images_first):
    """VideoGenerator.sample_pair: sample_videos(B) and sample_images(B) decoded in ONE pass over [B*T video rows | B image
    rows] with two BatchNorm batches of unequal size.  Same draws, same per-call batch statistics, same running-stat
    update order as the two calls of the reference (models/mocogan.py:271-295); only the summation order of the weight
    gradients (one pass over 17 row blocks instead of 16 + 1) differs."""
    def make():
        seed_all(71)
        if kind == "ucf":
            gen = G.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8)
        elif kind == "odernn":
            gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=16)
        else:
            gen = G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=16)
        return gen.cuda()
    B = 4
    res = []
    for joint in (False, True):
        gen = make()
        seed_all(72)
        if joint:
            (vid, lab), (img, none) = gen.sample_pair(B, B, images_first=images_first)
            assert none is None and lab.dtype == torch.float64 and lab.shape == (B,)
        elif images_first:
            img, _ = gen.sample_images(B)
            vid, _ = gen.sample_videos(B)
        else:
            vid, _ = gen.sample_videos(B)
            img, _ = gen.sample_images(B)
        wv = torch.linspace(-1, 1, vid.numel(), device="cuda").view(vid.shape)
        wi = torch.linspace(1, -1, img.numel(), device="cuda").view(img.shape)
        loss = (vid * wv).sum() + 3.0 * (img * wi).sum()
        loss.backward()
        torch.cuda.synchronize()
        res.append((vid.detach().clone(), img.detach().clone(), [p.grad.clone() for p in gen.parameters() if p.grad is not None],
                    [b.clone() for b in gen.buffers()], [n for n, p in gen.named_parameters() if p.grad is not None]))
    (v0, i0, g0, b0, names), (v1, i1, g1, b1, _) = res
    assert v0.shape == v1.shape and i0.shape == i1.shape and len(g0) == len(g1) > 8
    assert float((v0 - v1).abs().max()) <= 1e-5 and float((i0 - i1).abs().max()) <= 1e-5    # (other tile shapes: fp32 rounding)
    for a, b in zip(b0, b1):            # running mean / var: both updates, in the reference's call order
        assert float((a.double() - b.double()).abs().max()) <= 1e-6 * (1 + float(a.abs().max()))
    for n, a, b in zip(names, g0, g1):
        scale = float(a.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 1e-3 * scale, (n, float((a - b).abs().max()), scale)   # (4-sample BN batches)


def test_sample_pair_falls_back_to_the_two_calls_when_the_rows_do_not_split():
    """A batch whose statistics rows do not split at the batch boundary (here 3 videos = 48 rows + 3 image rows: no tile
    boundary at row 48 in the first layers) cannot be decoded jointly: can_pair says so, sample_pair runs the two calls --
    bit-identical to calling them -- and the trainer's step() keeps working on such a batch size."""
    def make():
        seed_all(81)
        return G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=16).cuda()
    gen = make()
    if gen.can_pair(3, 3):
        pytest.skip("this build decodes 48 + 3 rows jointly")
    outs = []
    for joint in (True, False):
        gen = make()
        seed_all(82)
        with torch.no_grad():
            if joint:
                (v, _), (i, _) = gen.sample_pair(3, 3, images_first=True)
            else:
                i, _ = gen.sample_images(3)
                v, _ = gen.sample_videos(3)
        outs.append((v.clone(), i.clone(), [b.clone() for b in gen.buffers()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)
    seed_all(83)
    gen, dv, di = G.build_mnist(ngf=8, ndf=8)
    gen.cuda(); dv.cuda(); di.cuda()
    tr = G.GanTrainer(gen, dv, di)
    imgs = [torch.rand(3, 1, 28, 28).cuda() for _ in range(2)]
    vids = [torch.rand(3, 16, 1, 28, 28).cuda() for _ in range(2)]
    for _ in range(2):
        li, lv, lg = tr.step(imgs, vids)
    assert all(torch.isfinite(x) for x in (li, lv, lg))
