"""GPU: data-parallel semantics -- per-replica BatchNorm statistics, gradients averaged over the replicas
by ONE reduction per optimiser step, 1/world folded into Adam -- against the oracle's golden "S independent shard
passes with averaged gradients" (oracle.mocogan_ref.train_step_dp).

On one GPU the replicas are *virtual*: GanTrainer takes lists of shard tensors and accumulates the shards' gradients
in its gradient arena (the buffer RCCL reduces in a real multi-GPU run), which exercises exactly the arithmetic a
world of S ranks performs (sum of S shard gradients, scaled by 1/S inside the Adam kernel).  The real multi-process
path is rehearsed with two rank processes sharing the one GPU over gloo (RCCL refuses two ranks on one device);
hardware scaling numbers come from the driver's 8-GPU run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, assert_weights_after_adam, seed_all

import gan_ode_amd as G
from oracle import mocogan_ref as M

pytestmark = pytest.mark.gpu


def _f32(a):
    return torch.from_numpy(np.asarray(a).astype(np.float32))


def _check_weights(models, oracles, frac_tol=2e-3):
    for m, o in zip(models, oracles):
        for (k, v), (_, w) in zip(m.state_dict().items(), o.state_dict().items()):
            if v.dtype == torch.int64 or "running_" in k:
                continue
            assert_weights_after_adam(v, w, k, frac=frac_tol)


@pytest.mark.parametrize("width,B,S", [(8, 4, 2), (64, 16, 2)])
def test_virtual_replicas_match_oracle_averaged_shard_gradients(width, B, S):
    """Two shards of B through the arena (second shard accumulating, 1/2 folded into Adam) against two independent
    oracle passes with averaged gradients, tiny width and BASELINE width (2 x 16 = one batch of 32 split over two
    replicas): losses 1e-4 relative, updated weights as in the train-step fixtures."""
    seed_all(11)
    gen, dv, di = G.build_mnist(ngf=width, ndf=width)
    ogen, odv, odi = M.build_mnist(ngf=width, ndf=width)
    for m, o in zip((gen, dv, di), (ogen, odv, odi)):
        o.load_state_dict(m.state_dict())
    gen.cuda(); dv.cuda(); di.cuda()
    tr = G.GanTrainer(gen, dv, di)
    opts = M.make_optimizers(ogen, odv, odi)
    rng = torch.Generator().manual_seed(3)
    imgs = [[torch.rand(B, 1, 28, 28, generator=rng) for _ in range(S)] for _ in range(2)]
    vids = [[torch.rand(B, 16, 1, 28, 28, generator=rng) for _ in range(S)] for _ in range(2)]
    seed_all(12)
    got = [float(v) for v in tr.step([[x.cuda() for x in sh] for sh in imgs], [[x.cuda() for x in sh] for sh in vids])]
    seed_all(12)
    want = [float(v) for v in M.train_step_dp(ogen, odv, odi, opts, imgs, vids)]
    assert np.allclose(got, want, rtol=1e-4, atol=0), (got, want)
    _check_weights((gen, dv, di), (ogen, odv, odi))
    # and it is NOT the single-replica update on the concatenated batch (per-replica BatchNorm statistics differ)
    seed_all(11)
    ogen2, odv2, odi2 = M.build_mnist(ngf=width, ndf=width)
    opts2 = M.make_optimizers(ogen2, odv2, odi2)
    seed_all(12)
    one = [float(v) for v in M.train_step(ogen2, odv2, odi2, opts2, [torch.cat(sh) for sh in imgs],
                                          [torch.cat(sh) for sh in vids])]
    assert not np.allclose(one, want, rtol=1e-4, atol=0)


def test_config2_batch256_as_8_virtual_replicas_of_32():
    """BASELINE.json configs[2] -- global batch 256 = 8 replicas x 32 clips, ngf=ndf=64 -- one full training iteration
    with the 8 replicas run back to back on the one GPU, against the oracle's 8 shard passes with averaged
    gradients.  Losses 1e-4; weights as above."""
    S, B = 8, 32
    seed_all(21)
    gen, dv, di = G.build_mnist()
    ogen, odv, odi = M.build_mnist()
    for m, o in zip((gen, dv, di), (ogen, odv, odi)):
        o.load_state_dict(m.state_dict())
    gen.cuda(); dv.cuda(); di.cuda()
    tr = G.GanTrainer(gen, dv, di)
    opts = M.make_optimizers(ogen, odv, odi)
    rng = torch.Generator().manual_seed(4)
    imgs = [[torch.rand(B, 1, 28, 28, generator=rng) for _ in range(S)] for _ in range(2)]
    vids = [[torch.rand(B, 16, 1, 28, 28, generator=rng) for _ in range(S)] for _ in range(2)]
    seed_all(22)
    got = [float(v) for v in tr.step([[x.cuda() for x in sh] for sh in imgs], [[x.cuda() for x in sh] for sh in vids])]
    seed_all(22)
    want = [float(v) for v in M.train_step_dp(ogen, odv, odi, opts, imgs, vids)]
    assert np.allclose(got, want, rtol=1e-4, atol=0), (got, want)
    # the generator's first layer sits behind 9 BatchNorm/(Leaky)ReLU kinks of both networks; averaged over 8 shards
    # a slightly larger share of its 540k entries has a gradient within fp32 noise of zero (measured 2.1e-3 flipped)
    _check_weights((gen, dv, di), (ogen, odv, odi), frac_tol=5e-3)


def test_two_rank_processes_on_one_gpu_through_the_bench_launcher():
    """`python bench.py --gpus 2` as the driver calls it, self-launching (parent spawns the ranks before any GPU call),
    here with --backend gloo because both ranks share the box's single GPU.  Checks the contract fields and that the
    replicas stayed in step (the trainer broadcasts rank 0's state; identical all-reduced gradients keep them equal)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5",
                        "--warmup", "2", "--train-steps", "2"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["global_batch"] == 64 and rec["config"]["parallelism"] == "dp2"
    assert rec["scaling"] == "weak" and rec["value"] > 0 and rec["iteration_ms"] > 0
    ar = rec["allreduce"]
    assert ar["collectives_per_iteration"] == 5
    assert ar["bytes_per_iteration_per_rank"] == 2 * (ar["bucket_bytes"]["dis_img"] + ar["bucket_bytes"]["dis_vid"]) + ar["bucket_bytes"]["gen"]
    # the diagnosis fields of the first real multi-GPU run: every collective timed on its own, their sum per iteration, and
    # the iteration with / without them
    assert set(ar["ms_per_collective"]) == {"dis_img", "dis_vid", "gen"} and all(v > 0 for v in ar["ms_per_collective"].values())
    assert ar["ms_per_iteration"] > 0 and ar["iteration_ms_without_allreduce"] > 0 and ar["overlapped"] is True
    assert isinstance(ar["exposed_ms_per_iteration"], float)
    assert rec["cpu_baseline"] is None          # rank-0-at-N=1 only


@pytest.mark.parametrize("which", ["ode", "odernn"])
def test_overlapped_allreduce_schedule_is_bitwise_the_serial_one(which):
    """GanTrainer(overlap_allreduce=True) (default): asynchronous all-reduces, the optimiser step deferred to where the
    network is next needed, the generator's arena in two parts (decoder block under the latent adjoint, ODE tail after
    it) -- against the serial form (collective, then Adam) on two rank processes sharing the GPU over gloo: same losses,
    same weights and Adam moments, bit for bit; replicas equal among themselves."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_gode_launch", os.path.join(REPO, "gan-ode_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    recs = []
    for mode in ("overlap", "serial"):
        code, out = launch.spawn_ranks(2, [sys.executable, os.path.join(REPO, "tests", "_dp_rank.py"), mode, which], timeout=600)
        assert code == 0, out[-3000:]
        recs.append(json.loads([ln for ln in out.splitlines() if ln.startswith('{"mode"')][0]))
    assert recs[0]["losses"] == recs[1]["losses"], recs
    assert recs[0]["digest"] == recs[1]["digest"]


def test_single_gpu_bench_line_through_the_spawn_path():
    """--gpus 1 --spawn: the same launcher with one rank prints the same kind of line as the in-process run."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--spawn", "--steps", "5", "--warmup", "2",
                        "--train-steps", "2", "--no-cpu-baseline", "--no-live-traffic"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert rec["n_gpus"] == 1 and rec["roofline"]["bound"] == "mfma" and rec["iteration"]["algorithmic_gflop"] > 100
    assert rec["allreduce"]["bytes_per_iteration_per_rank"] == 0


