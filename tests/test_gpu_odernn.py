"""GPU: the ODE-RNN motion latent (BASELINE.json configs[4]; models/mocogan_ode_rnn.py) -- kernels through the C ABI
and the drop-in generator class against the CPU oracle (oracle.mocogan_ref.GeneratorOdeRnn on oracle.ode_ref's
dopri5).  Parity is unpinned for this row (reference file un-importable, torchdiffeq absent, no fixture).
Adaptive step sequences differ between two fp32 implementations, so states agree to the solver tolerance (~1e-6),
asserted at 2e-5; gradients: the device's ADAPTIVE adjoint (dopri5 on (y, a, g_theta) with torchdiffeq's mixed norm,
substeps=0, the default) against the oracle's adaptive adjoint at 1e-4 (measured ~1e-6), and the fixed 32-substep
discretisation kept as an option, also at 1e-4 (measured ~4e-6)."""
import numpy as np
import pytest
import torch

from conftest import rel_err, seed_all

import gan_ode_amd as G
import gan_ode_amd._lib as L
from oracle import mocogan_ref as M
from oracle import ode_ref

pytestmark = pytest.mark.gpu


def stream():
    return torch.cuda.current_stream().cuda_stream


# (33, 2): a second workgroup holding ONE trajectory; (4096, 2): the co-residency cap of the whole-batch norm exchange (128
# workgroups waiting for each other); (4128, 2): one workgroup more -- the MFMA fallback with one norm per 64 trajectories
@pytest.mark.parametrize("N,T", [(20, 5), (32, 16), (300, 3), (33, 2), (4096, 2), (4128, 2)])
def test_odernn_kernels_against_oracle(N, T):
    torch.manual_seed(N)
    f = M.OdeRhs(16, 16)
    gru = torch.nn.GRUCell(16, 16)
    noise = torch.randn(T + 1, N, 16)
    # oracle; the right-hand side counts its evaluations: a dopri5 call makes 2 (initial step selection) + 6 per trial step
    calls = [0]
    f.register_forward_pre_hook(lambda *_: calls.__setitem__(0, calls[0] + 1))
    h = [noise[0]]
    hps = []
    ref_trials = []
    t01 = torch.tensor([0.0, 1.0])
    for t in range(T):
        c0 = calls[0]
        hp = ode_ref.odeint_adjoint(f, h[-1], t01)[-1]
        ref_trials.append((calls[0] - c0 - 2) // 6)
        hps.append(hp)
        h.append(gru(noise[t + 1], hp))
    zref = torch.stack(h[1:], dim=1)                       # [N, T, 16]
    gup = torch.randn(N, T, 16, generator=torch.Generator().manual_seed(7))
    (zref * gup).sum().backward()
    ref_grads = [p.grad for p in list(f.parameters()) + [gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh]]
    # device
    P = [p.detach().cuda() for p in list(f.parameters()) + [gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh]]
    op = L.OdeRnnParams(*[p.data_ptr() for p in P])
    nz, content = noise.cuda(), torch.randn(N, 50).cuda()
    z = torch.full((N * T, 72), float("nan"), device="cuda")
    hp_d = torch.empty(N, T, 16, device="cuda")
    nst = torch.full((T,), -7, dtype=torch.int32, device="cuda")
    nsync = L.lib().gode_odernn_sync_size(N)
    sync = torch.zeros(max(nsync, 1), dtype=torch.int32, device="cuda")
    fop = L.OdeRnnFwdOp(p=op, noise=nz.data_ptr(), content=content.data_ptr(), sel_t=None, z=z.data_ptr(), hs=None,
                        hp=hp_d.data_ptr(), nsteps=nst.data_ptr(), N=N, T=T, rtol=1e-7, atol=1e-9, zcols=72,
                        sync=sync.data_ptr() if nsync else None)
    L.run_one(fop, stream())
    zz = z.cpu().view(N, T, 72)
    assert rel_err(hp_d.cpu(), torch.stack(hps, dim=1).detach()) < 2e-5
    assert rel_err(zz[:, :, :16], zref.detach()) < 2e-5
    assert torch.equal(zz[:, :, 16:66], content.cpu()[:, None, :].expand(N, T, 50)) and float(zz[:, :, 66:].abs().max()) == 0
    steps = nst.cpu()
    assert int(steps.min()) >= 2 and int(steps.max()) < 200, steps      # the controller converges in a handful of steps
    # whole-batch error norm (also across the 10 workgroups of N = 300): the device takes the oracle's step sequence
    # (an accept / reject decision within rounding of ratio = 1 may fall either way in two fp32 evaluations -- at rtol 1e-7 the
    # error estimate is a few ulps of the state: measured, one frame of the N = 20 case takes 6 trial steps where the oracle takes 7)
    diff = [abs(a - b) for a, b in zip(steps.tolist(), ref_trials)]
    if N <= 4096:        # (above: the fallback's norm is per 64-trajectory workgroup, its step sequence its own)
        assert max(diff) <= 1 and sum(1 for d in diff if d) <= max(1, T // 8), (steps.tolist(), ref_trials)
    gz = torch.zeros(N * T, 72, device="cuda")
    gz.view(N, T, 72)[:, :, :16] = gup.cuda()
    grads = torch.full((L.ODERNN_NPARAM,), float("nan"), device="cuda")
    work = torch.empty(L.lib().gode_odernn_bwd_work_size(N), device="cuda")
    nstb = torch.full((T,), -7, dtype=torch.int32, device="cuda")
    syncb = torch.zeros(max(nsync, 1), dtype=torch.int32, device="cuda")
    for substeps, tol in ((0, 1e-4), (32, 1e-4)):       # 0: adaptive adjoint (what torchdiffeq does); 32: fixed Kutta-3/8
        grads.fill_(float("nan"))
        bop = L.OdeRnnBwdOp(p=op, noise=nz.data_ptr(), hp=hp_d.data_ptr(), sel_t=None, gz=gz.data_ptr(), work=work.data_ptr(),
                            grads=grads.data_ptr(), N=N, T=T, substeps=substeps, accumulate=0, zcols=72, rtol=1e-7, atol=1e-9,
                            sync=syncb.data_ptr() if nsync else None, nsteps=nstb.data_ptr())
        L.run_one(bop, stream())
        if substeps == 0:
            sb = nstb.cpu()
            assert int(sb.min()) >= 2 and int(sb.max()) < 400, sb        # no stalled call (those report negative counts)
        g = grads.cpu()
        off = 0
        for name, r in zip(("W1", "b1", "W2", "b2", "Wih", "Whh", "bih", "bhh"), ref_grads):
            n = r.numel()
            assert rel_err(g[off:off + n].view_as(r), r) < tol, (name, substeps, rel_err(g[off:off + n].view_as(r), r))
            off += n


def test_odernn_generator_against_oracle():
    seed_all(5)
    gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=8)
    ref = M.GeneratorOdeRnn(1, 50, 0, 16, 16, ngf=8, mnist=True)
    ref.load_state_dict(gen.state_dict())
    gen.cuda()
    seed_all(6)
    vid, labels = gen.sample_videos(4)
    img, _ = gen.sample_images(4)
    seed_all(6)
    rvid, _ = ref.sample_videos(4)
    rimg, _ = ref.sample_images(4)
    assert vid.shape == (4, 1, 16, 28, 28) and img.shape == (4, 1, 28, 28)
    assert rel_err(vid.detach().cpu(), rvid.detach()) < 1e-4
    assert rel_err(img.detach().cpu(), rimg.detach()) < 1e-4
    wv = torch.randn(vid.shape, generator=torch.Generator().manual_seed(1))
    wi = torch.randn(img.shape, generator=torch.Generator().manual_seed(2))
    ((vid * wv.cuda()).sum() + (img * wi.cuda()).sum()).backward()
    ((rvid * wv).sum() + (rimg * wi).sum()).backward()
    for (k, p), (_, q) in zip(gen.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, k          # the pre-net `linear` is unused by the ODE-RNN variant
        else:
            d = (p.grad.cpu() - q.grad).abs()
            assert float(d.median() / (q.grad.norm() / q.grad.numel() ** 0.5)) < 5e-3, k
            assert rel_err(p.grad.cpu(), q.grad) < 5e-2, k
    assert gen.recurrent.weight_hh.grad is not None       # the GRU cell is live in this variant


@pytest.mark.parametrize("N,T,prenet", [(16, 16, True), (32, 16, False), (100, 6, True)])
def test_dopri5_latent_kernels_against_oracle(N, T, prenet):
    """gode_ode_fwd / gode_ode_bwd with method 1 (ode_method="dopri5" on the plain generators; BASELINE configs[3] wording)
    through the C ABI: pre-net + ONE adaptive solve over the T output times with dense output, and ONE adjoint call over the
    T - 1 intervals (theta state carried, solver restarted per interval), against oracle.ode_ref (parity unpinned).  N = 100:
    four workgroups exchanging the whole-batch norm.  States 2e-5, all eight gradient tensors 1e-4, trial-step counts of the
    forward call equal to the oracle's up to one borderline decision."""
    torch.manual_seed(N + T)
    f = M.OdeRhs(16, 16)
    pre = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.LeakyReLU(0.2), torch.nn.Linear(64, 16), torch.nn.LeakyReLU(0.2))
    x = torch.randn(N, 16)
    tt = torch.linspace(0, 1, T).float()
    calls = [0]
    f.register_forward_pre_hook(lambda *_: calls.__setitem__(0, calls[0] + 1))
    y0 = pre(x) if prenet else x.clone().requires_grad_(True)
    sol = ode_ref.odeint_adjoint(f, y0, tt, method="dopri5")              # [T, N, 16]
    ref_trials = (calls[0] - 2) // 6
    zref = sol.transpose(0, 1)                                            # [N, T, 16]
    gup = torch.randn(N, T, 16, generator=torch.Generator().manual_seed(3))
    (zref * gup).sum().backward()
    params = (list(pre.parameters()) if prenet else []) + list(f.parameters())
    ref_grads = [p.grad for p in params]
    # device
    dev = [p.detach().cuda() for p in params]
    ptrs = [p.data_ptr() for p in dev]
    op = L.OdeParams(*(ptrs if prenet else [None] * 4 + ptrs))
    xd, content, tout = x.cuda(), torch.randn(N, 50).cuda(), tt.cuda()
    z = torch.full((N * T, 72), float("nan"), device="cuda")
    traj = torch.empty(N, T, 16, device="cuda")
    nst = torch.full(((N + 63) // 64,), -7, dtype=torch.int32, device="cuda")
    nsync = L.lib().gode_odernn_sync_size(N)
    sync = torch.zeros(max(nsync, 1), dtype=torch.int32, device="cuda")
    fop = L.OdeFwdOp(p=op, x=xd.data_ptr(), content=content.data_ptr(), dt=None, sel_t=None, z=z.data_ptr(), traj=traj.data_ptr(),
                     N=N, T=T, substeps=1, prenet=1 if prenet else 0, zcols=72, method=1, rtol=1e-7, atol=1e-9, tout=tout.data_ptr(),
                     nsteps=nst.data_ptr(), sync=sync.data_ptr() if nsync else None)
    L.run_one(fop, stream())
    zz = z.cpu().view(N, T, 72)
    assert rel_err(traj.cpu(), zref.detach()) < 2e-5 and rel_err(zz[:, :, :16], zref.detach()) < 2e-5
    assert torch.equal(zz[:, :, 16:66], content.cpu()[:, None, :].expand(N, T, 50))
    assert abs(int(nst[0]) - ref_trials) <= 1, (int(nst[0]), ref_trials)
    gz = torch.zeros(N * T, 72, device="cuda")
    gz.view(N, T, 72)[:, :, :16] = gup.cuda()
    grads = torch.full((L.ODE_NPARAM,), float("nan"), device="cuda")
    work = torch.empty(L.lib().gode_ode_bwd_work_size(N), device="cuda")
    nstb = torch.full(((N + 63) // 64,), -7, dtype=torch.int32, device="cuda")
    syncb = torch.zeros(max(nsync, 1), dtype=torch.int32, device="cuda")
    bop = L.OdeBwdOp(p=op, x=xd.data_ptr(), traj=traj.data_ptr(), dt=None, sel_t=None, gz=gz.data_ptr(), work=work.data_ptr(),
                     grads=grads.data_ptr(), N=N, T=T, substeps=0, prenet=1 if prenet else 0, accumulate=0, zcols=72, method=1,
                     rtol=1e-7, atol=1e-9, tout=tout.data_ptr(), nsteps=nstb.data_ptr(), sync=syncb.data_ptr() if nsync else None)
    L.run_one(bop, stream())
    assert int(nstb[0]) >= T - 1, nstb
    g = grads.cpu()
    names = ("Wa", "ba", "Wb", "bb", "W1", "b1", "W2", "b2")
    offs = (0, 1024, 1088, 2112, 2128, 2384, 2400, 2656)
    lens = (1024, 64, 1024, 16, 256, 16, 256, 16)
    k0 = 0 if prenet else 4
    if not prenet:
        assert float(g[:2128].abs().max()) == 0.0
    for name, o, n_, r in zip(names[k0:], offs[k0:], lens[k0:], ref_grads):
        assert rel_err(g[o:o + n_].view_as(r), r) < 1e-4, (name, rel_err(g[o:o + n_].view_as(r), r))
    # accumulate flag
    base = grads.clone()
    bop.accumulate = 1
    L.run_one(bop, stream())
    assert rel_err(grads.cpu(), 2 * base.cpu()) < 1e-6


def test_odernn_generator_surface_video_len_eval_mode_and_more_than_one_workgroup():
    """The rest of the drop-in class's surface against the oracle: the video_len argument, 40 videos / 40 images (two
    workgroups exchanging the whole-batch error norm through the plan's sync words, the image path with row selection),
    eval mode (running statistics), state_dict round trip into the oracle class."""
    seed_all(15)
    gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=8)
    ref = M.GeneratorOdeRnn(1, 50, 0, 16, 16, ngf=8, mnist=True)
    ref.load_state_dict(gen.state_dict())
    gen.cuda()
    with torch.no_grad():
        seed_all(16)
        v8, _ = gen.sample_videos(3, 8)
        v40, lab = gen.sample_videos(40)
        i40, _ = gen.sample_images(40)
        seed_all(16)
        r8, _ = ref.sample_videos(3, 8)
        r40, _ = ref.sample_videos(40)
        ri40, _ = ref.sample_images(40)
    assert v8.shape == (3, 1, 8, 28, 28) and v40.shape == (40, 1, 16, 28, 28) and i40.shape == (40, 1, 28, 28)
    assert lab.dtype == torch.float64 and lab.shape == (40,)
    assert rel_err(v8.cpu(), r8) < 1e-4 and rel_err(v40.cpu(), r40) < 1e-4 and rel_err(i40.cpu(), ri40) < 1e-4
    plan = gen._pool.plans[(40, 16, False)][0]
    assert plan.sync_f is not None and int(plan.nsteps.min()) >= 2          # two workgroups, no stalled frame
    # running statistics moved identically; eval mode uses them
    for (k, a), (_, b) in zip(gen.state_dict().items(), ref.state_dict().items()):
        if "running_" in k:
            assert rel_err(a.cpu(), b) < 1e-4, k
        if "num_batches" in k:
            assert int(a) == int(b), k
    gen.eval(); ref.eval()
    with torch.no_grad():
        seed_all(17)
        ve, _ = gen.sample_videos(5)
        seed_all(17)
        re_, _ = ref.sample_videos(5)
    assert rel_err(ve.cpu(), re_) < 1e-4
    # gradients through two workgroups (adjoint exchange incl. the parameter-tensor norms)
    gen.train(); ref.train()
    seed_all(18)
    vg, _ = gen.sample_videos(40)
    seed_all(18)
    rg, _ = ref.sample_videos(40)
    w = torch.randn(vg.shape, generator=torch.Generator().manual_seed(2))
    (vg * w.cuda()).sum().backward()
    (rg * w).sum().backward()
    assert int(plan.nsteps_bwd.min()) >= 2
    refp = dict(ref.named_parameters())
    for k in ("ode_fn.fn.0.weight", "ode_fn.fn.0.bias", "ode_fn.fn.2.weight", "ode_fn.fn.2.bias", "recurrent.weight_ih",
              "recurrent.weight_hh", "recurrent.bias_ih", "recurrent.bias_hh"):
        assert rel_err(dict(gen.named_parameters())[k].grad.cpu(), refp[k].grad) < 5e-3, k
