"""CPU: host-side logic of the product -- C-ABI loading/symbols, drop-in construction (same seed -> the reference's
initial weights, same state_dict keys), refusal of CPU tensors (no fallback path), the host RNG / row-selection
contract of sample_images, and the data-parallel gradient bucket over gloo with two ranks."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, golden, seed_all

import gan_ode_amd as G
from gan_ode_amd import _lib as L
from oracle import mocogan_ref as M


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.lib()                      # also checks every op struct size against the compiled ABI
    assert lib.gode_version() == 105
    hdr = open(os.path.join(REPO, "include", "gode.h")).read()
    declared = set(re.findall(r"\b(gode_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(L.EXPORTS)


def test_argument_errors_are_reported_not_crashed():
    lib = L.lib()
    bad = L.IgemmOp()                  # null pointers
    assert lib.gode_igemm(C.byref(bad), None) < 0
    g = L.ConvGeom(1, 4, 4, 1, 8, 8, 1, 5, 5, 1, 4, 4, 1, 2, 2, 0, 1, 1)   # Ho should be 4: conv relation violated
    assert lib.gode_pack_size(C.byref(g), L.FPROP) < 0


@pytest.mark.parametrize("tag,ctor", [
    ("mnist_tiny", lambda: G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8)),
    ("ucf_tiny", lambda: G.VideoGenerator(3, 50, 0, 16, 16, dim_hidden=16, ngf=8)),
])
def test_same_seed_gives_the_reference_initial_weights(tag, ctor):
    g = golden(f"gen_{tag}.npz")
    seed_all(int(g["seed"]))
    gen = ctor()
    sd = gen.state_dict()
    want = {k[2:] for k in g.files if k.startswith("w/")}
    assert set(sd.keys()) == want
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g[f"w/{k}"]), k
    assert [k for k, _ in gen.named_parameters()][:4] == ["recurrent.weight_ih", "recurrent.weight_hh",
                                                          "recurrent.bias_ih", "recurrent.bias_hh"]


def test_discriminator_state_dict_keys_match_reference():
    for tag, ctor in (("vid_mnist_tiny", lambda: G.VideoDiscriminator(1, ksize=2, ndf=8)),
                      ("img_ucf_tiny", lambda: G.PatchImageDiscriminator(3, ndf=8))):
        g = golden(f"disc_{tag}.npz")
        assert set(ctor().state_dict().keys()) == {k[2:] for k in g.files if k.startswith("w/")}


def test_ucf_constructor_fails_like_the_reference_without_dim_hidden():
    with pytest.raises(TypeError):
        G.VideoGenerator(3, 50, 0, 16, 16)          # ucf_moco_ode.py:80 as shipped


def test_no_cpu_fallback():
    gen, dv, di = G.build_mnist(ngf=8, ndf=8)
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        gen.sample_videos(2)
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        dv(torch.zeros(2, 1, 16, 28, 28))
    with pytest.raises(RuntimeError):
        G.bce_with_logits_const(torch.zeros(3), 1.0)
    with pytest.raises(RuntimeError):
        gen.ode_fn(None, torch.zeros(1, 16))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        gen.prefetch_latents([("videos", 2)])
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        gen.sample_z_m(2)


def test_sample_images_host_draws_match_oracle_rng_order(monkeypatch):
    """The pruned sample_images must consume the host RNGs exactly like the reference (NumPy normal -> torch randn ->
    NumPy choice) and pick the same latent rows."""
    gen = G.VideoGeneratorMNISTODE(1, 50, 0, 16, 16, ngf=8)
    captured = {}

    def fake_run(n_traj, T, select, x, content, sel):
        captured.update(n=n_traj, T=T, select=select, x=x.clone(), content=content.clone(), sel=sel.clone())
        return torch.zeros(n_traj, 1, 28, 28, 1)

    monkeypatch.setattr(gen, "_run", fake_run)
    B, T = 5, 16
    seed_all(123)
    gen.sample_images(B)
    after_np, after_t = np.random.rand(), torch.rand(1)
    # replay of the reference's draw sequence (models/mocogan.py:288-290 via :252 and mocogan_ode.py:136)
    seed_all(123)
    content = np.random.normal(0, 1, (B * T * 2, 50)).astype(np.float32)
    x = torch.randn(B * T * 2, 16)
    j = np.sort(np.random.choice(B * T * 2 * T, B, replace=False)).astype(np.int64)
    assert np.random.rand() == after_np and torch.equal(torch.rand(1), after_t)   # generators left in the same state
    assert captured["select"] and captured["n"] == B
    assert torch.equal(captured["x"], x[j // T])
    assert np.array_equal(captured["content"].numpy(), content[j // T])
    assert np.array_equal(captured["sel"].numpy(), (j % T).astype(np.int32))
    # and the oracle's z rows agree with that reading: row j of z is (content[j // T], motion(traj j // T, time j % T))
    seed_all(123)
    ogen = M.Generator(1, 50, 0, 16, 16, ngf=8)
    seed_all(123)
    z, _ = ogen.sample_z_video(B * T * 2)
    assert np.array_equal(z[j, :50].detach().numpy(), content[j // T])


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2), torch.nn.GRUCell(2, 2))
    params = list(net.parameters())
    for i, p in enumerate(params[:4]):                       # the GRU cell (last 4) never gets a gradient
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    b = G.train.GradBucket(params)
    flat = b.gather()
    assert flat.numel() == sum(p.numel() for p in params[:4])
    b.all_reduce()
    views = b.views()
    ok = len(views) == 4
    for i, p in enumerate(params[:4]):
        ok = ok and torch.equal(views[p], torch.full_like(p, 3.0 * (i + 1)))     # (1 + 2) * (i + 1)
        ok = ok and views[p].data_ptr() >= flat.data_ptr()                        # views alias the bucket (no copy back)
    # the GradArena form of the same step (what the GPU trainer uses): writers store / add into the arena, p.grad is a
    # view of it, ONE all-reduce over the whole flat buffer, parameters nobody wrote keep grad None
    arena = G.train.GradArena(params)
    arena.begin()
    ok = ok and all(p.grad is None for p in params)
    for i, p in enumerate(params[:4]):
        v, acc = arena.target(p)
        ok = ok and not acc
        v.fill_(float(rank + 1) * (i + 1))                   # first pass of the step stores ...
        v2, acc2 = arena.target(p)
        ok = ok and acc2 and v2.data_ptr() == v.data_ptr()
        v2.add_(1.0)                                         # ... the second pass adds in place
    arena.end()
    dist.all_reduce(arena.flat)
    for i, p in enumerate(params[:4]):
        ok = ok and torch.equal(p.grad, torch.full_like(p, 3.0 * (i + 1) + 2.0)) and p.grad.data_ptr() == arena.views[p].data_ptr()
    ok = ok and all(p.grad is None for p in params[4:])
    q.put((rank, ok))
    dist.destroy_process_group()


def test_grad_bucket_all_reduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 500
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_rot_mnist_reader_and_device_streams(tmp_path):
    """Input side: the .mat contract of dataset/mnist_rotation.py and the shuffled drop_last
    streams, on a synthetic file of the right shape."""
    from scipy.io import savemat
    from gan_ode_amd.data import MNISTRotationImage, MNISTRotationVideo, RotMnistOnDevice
    rng = np.random.RandomState(0)
    X = rng.rand(40, 16, 784).astype(np.float32)
    savemat(tmp_path / "rot.mat", {"X": X, "Y": np.arange(40)[:, None]})
    vid = MNISTRotationVideo(str(tmp_path / "rot.mat"), N=30)
    img = MNISTRotationImage(str(tmp_path / "rot.mat"), N=30)
    assert len(vid) == 30 and vid[3][0].shape == (16, 1, 28, 28) and int(vid[3][1]) == 3
    assert np.array_equal(vid[3][0].numpy().reshape(16, 784), X[3])
    np.random.seed(5)
    f = np.random.randint(0, 16)
    np.random.seed(5)
    assert np.array_equal(img[7][0].numpy().reshape(784), X[7, f])
    with pytest.raises(FileExistsError):
        MNISTRotationVideo(str(tmp_path / "missing.mat"))
    dev = RotMnistOnDevice(vid.X, device="cpu", seed=1)
    vs, ims = dev.videos(8), dev.images(8)
    seen = []
    for _ in range(3):                       # 30 clips, batch 8, drop_last -> 3 batches per epoch
        v = next(vs)
        assert v.shape == (8, 16, 1, 28, 28)
        seen += [int(np.where((X[:30] == v[i].numpy().reshape(16, 784)).all(axis=(1, 2)))[0][0]) for i in range(8)]
    assert len(set(seen)) == 24              # no clip repeats inside an epoch
    assert next(ims).shape == (8, 1, 28, 28)


def test_odernn_dropin_constructs_and_draws_like_the_oracle():
    """models/mocogan_ode_rnn.py drop-in: same state_dict keys as the restated reference class, host noise drawn in
    the reference order (NumPy content, then h_0 and one e_t per frame from FloatTensor.normal_), no CPU path."""
    from models.mocogan_ode_rnn import VideoGeneratorMNISTODERNN
    seed_all(3)
    gen = VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=8)
    ref = M.GeneratorOdeRnn(1, 50, 0, 16, 16, ngf=8, mnist=True)
    assert list(gen.state_dict().keys()) == list(ref.state_dict().keys())
    seed_all(5)
    content, noise = gen._draw(4, 16)
    seed_all(5)
    oc = ref.sample_z_content(4)
    h0, e0 = ref._normal(4), ref._normal(4)
    assert noise.shape == (17, 4, 16) and torch.equal(noise[0], h0) and torch.equal(noise[1], e0)
    assert torch.equal(oc[::16], content)
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        gen.sample_videos(2)


def test_solver_grid_arrays_reproduce_the_fixed_grid_solver():
    """Host logic of the `ode_step_size` option (= torchdiffeq options={'step_size': h}): the step sizes, output
    slots and interpolation weights handed to the device kernel, replayed on the CPU with the oracle's Kutta-3/8
    increment, must give exactly what oracle/ode_ref.py:fixed_grid_solve (the restated FixedGridODESolver) returns --
    forward on the solver grid and per-interval reversed grids for the adjoint pass."""
    from gan_ode_amd.modules import solver_grid_arrays
    from oracle import ode_ref
    torch.manual_seed(3)
    W1, W2 = torch.randn(16, 16) * 0.4, torch.randn(16, 16) * 0.4
    f = lambda t, y: torch.tanh(y @ W1.T) @ W2.T          # noqa: E731  (autonomous, like ODEFunc)
    y0 = torch.randn(5, 16)
    T = 16
    t = torch.linspace(0, 1, T).float()
    for h in (0.05, 0.04, 1.0 / 15.0, 0.2):
        gd = solver_grid_arrays(T, h)
        want = ode_ref.fixed_grid_solve(f, y0, t, "rk4", step_size=h)
        got = [y0]
        y, j = y0, 1
        for i in range(gd["G"]):
            dt = gd["grid_dt"][i]
            y1 = y + ode_ref.kutta38_increment(f, torch.zeros(()), dt, dt, y)
            while j < T and int(gd["emit_at"][j]) == i:
                w = gd["emit_w"][j]
                got.append(y1 if float(w) == 1.0 else (y if float(w) == 0.0 else y + w * (y1 - y)))
                j += 1
            y = y1
        assert j == T and torch.equal(torch.stack(got), want), h
        # reversed-span grids of the adjoint pass: same constructor on (-t_i, -t_{i-1})
        for i in range(1, T):
            seg = -t[i - 1:i + 1].flip(0)
            rg = ode_ref._grid_from_step_size(seg, h)
            d = rg[1:] - rg[:-1]
            lo, hi = int(gd["bstep_off"][i - 1]), int(gd["bstep_off"][i])
            assert torch.equal(gd["bstep_dt"][lo:hi], d), (h, i)


def test_cost_model_and_statistics_layout_queries_are_host_side():
    """gode_igemm_model_cycles / gode_igemm_stats_segments (what ConvStack(split_images=...) plans with) do no GPU work:
    the joint 512 + 32-row decoder batch of configs[1] prices above its two parts on the layers whose 512 rows fill whole
    rounds of 256 workgroups (so the engine launches those GEMMs per part), its statistics rows split at the batch boundary,
    and an op off the modelled path answers -1."""
    import ctypes as C
    from gan_ode_amd.engine import make_geom
    lib = L.lib()

    def cyc(n, ci, co, hw):
        g = make_geom(n, ci, co, (1, 2 * hw, 2 * hw), (1, hw, hw), (1, 4, 4), (1, 2, 2), (0, 1, 1))
        return lib.gode_igemm_model_cycles(C.byref(L.IgemmOp(g=g, dir=L.DGRAD, tile=0))), g
    for ci, co, hw in ((256, 512, 4), (128, 256, 8)):
        joint, g = cyc(544, ci, co, hw)
        a, _ = cyc(512, ci, co, hw)
        b, _ = cyc(32, ci, co, hw)
        assert joint > 0 and a > 0 and b > 0
        assert a + b + 12000.0 < joint, (ci, co, joint, a, b)
    joint, g = cyc(544, 64, 128, 16)              # 4352 tiles of 128x64: the joint launch is the cheaper one
    a, _ = cyc(512, 64, 128, 16)
    b, _ = cyc(32, 64, 128, 16)
    assert joint < a + b + 12000.0
    seg = (C.c_int32 * 24)()
    nseg = lib.gode_igemm_stats_segments(C.byref(L.IgemmOp(g=g, dir=L.DGRAD, tile=0)), 512, seg)
    assert nseg == 4                              # one {begin, split, end} per stride phase
    for k in range(nseg):
        b0, sp, e0 = seg[3 * k], seg[3 * k + 1], seg[3 * k + 2]
        assert b0 < sp < e0 and (sp - b0) * 32 == (e0 - b0) * 512 * 32 // 544
    assert lib.gode_igemm_stats_segments(C.byref(L.IgemmOp(g=g, dir=L.DGRAD, tile=0)), 0, seg) < 0
    thin = make_geom(32, 1, 64, (1, 28, 28), (1, 32, 32), (1, 1, 1), (1, 1, 1), (0, 2, 2))
    assert lib.gode_igemm_model_cycles(C.byref(L.IgemmOp(g=thin, dir=L.DGRAD, tile=0))) == -1.0
