"""bench.py -- BASELINE.json's metric on MI355X: generated videos/s (16-frame clips) of gen.sample_videos(32) on
Rotated-MNIST-shaped synthetic data (batch 32, 16x1x28x28, ngf=ndf=64; configs[1]), plus G-step / D-step / full
iteration milliseconds, at N = 1/2/4/8 GPUs (one process per GPU; weak scaling, 32 clips per GPU; gradients are
all-reduced over RCCL once per optimiser step in the train-step timings).

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under a launcher that already set RANK/WORLD_SIZE (python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...) this process IS a rank; started
plainly, it becomes a parent that spawns N fresh rank processes (gan-ode_amd/launch.py) before anything touches the
GPU, relays rank 0's JSON line and exits non-zero if any rank did.

A "step" of the headline metric is one sample_videos(32) call (host noise draw + H2D of 8.4 KB + pre-net + RK4 +
decoder, train-mode BatchNorm, no_grad) -- exactly what the reference's "generated videos" are.  rank 0 prints ONE
JSON line.  `roofline` is measured live with HIP events around repeated launches of the dominant kernel (the
stride-2 ConvTranspose implicit GEMM) on the stream it runs on; `iteration` is the same figure for one whole
training iteration (algorithmic FLOPs of the convolution GEMMs it executes / its wall time) with a per-class time
split; `cpu_baseline` times the CPU oracle (a port of the reference path on stock torch CPU kernels) on this box's
host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# torch / numpy are imported by the RANK processes only (_rank_main): the launching parent of a multi-GPU run must
# not touch the GPU, and does not even import torch (gan-ode_amd/launch.py is standard library only).
np = torch = dist = None

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0           # same guide, "HBM3E peak BW" (spec)
B, T = 32, 16


def _sync_all(distributed):
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()


def _timed(fn, steps, warmup, distributed, after_warmup=None):
    for _ in range(warmup):
        fn()
    if after_warmup is not None:
        after_warmup()
    _sync_all(distributed)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _sync_all(distributed)
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def _pmc_traffic(layer):
    """HBM-side bytes per launch of that layer's kernel from the newest committed PMC passes (separate --pmc
    FETCH_SIZE / WRITE_SIZE runs of `bench.py --roofline-only`; not collectable from inside this process).  Returns
    (bytes or None, name of the file they come from)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic*.json")),
                   key=lambda f: (os.path.basename(f).split("_")[0], os.path.getmtime(f)))
    for f in reversed(files):
        try:
            pm = json.load(open(f))["per_launch"]
            return pm[layer]["traffic_bytes"], os.path.basename(f)
        except Exception:
            continue
    return None, None


def _live_pmc_traffic(layer_index, timeout_s=150):
    """HBM-side (L2-miss / fabric) bytes per launch of decoder layer `layer_index`, measured NOW: two child runs of
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --roofline-only` (separate passes, as
    MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled for gfx950's 64-B tally of 128-B requests; counter unit KB), last of
    the 30 timed launches of that layer.  Children, not this process: counters cannot be collected from inside.  Returns
    (bytes, description) or (None, reason)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env["TMPDIR"] = "/tmp"
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gode_pmc_", dir="/tmp")
        try:
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--",
                                sys.executable, os.path.abspath(__file__), "--roofline-only"], cwd="/tmp", env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}"
            path = None
            for root, _, files in os.walk(d):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            if path is None:
                return None, "no counter_collection.csv"
            rows = [q for q in csv.DictReader(open(path)) if q["Counter_Name"] == counter and "igemm_fast" in q["Kernel_Name"]]
            rows.sort(key=lambda q: int(q["Dispatch_Id"]))
            # --roofline-only: one sample_videos call (layers 0..3 once each), then 3 warm-up + 30 timed launches per layer
            if len(rows) != 4 + 4 * 33:
                return None, f"unexpected dispatch count {len(rows)}"
            vals[counter] = float(rows[4 + layer_index * 33 + 32]["Counter_Value"])
        except Exception as e:      # noqa: BLE001 -- a measurement aid: any failure falls back to the committed file
            return None, f"{type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024), \
        "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of `bench.py --roofline-only` in this run"


def _kernel_roofline(gen, reps=30, live_traffic=False):
    """Times each decoder GEMM launch on its own with HIP events on the launch stream and returns the roofline
    object of the launch class that dominates the forward pass."""
    import gan_ode_amd._lib as L
    from gan_ode_amd.engine import stream_ptr
    with torch.no_grad():
        gen.sample_videos(B)                       # builds the plan and fills every buffer
    plan = gen._pool.plans[(B, T, False)][0]
    prog, _ = plan.stack._fwd[True]
    rows = B * T
    per_layer = []
    igemms = [op for op in prog.ops if isinstance(op, L.IgemmOp)]
    # MACs per latent row (= per frame): Cin*Cout*taps*input positions (0.54M, 3 x 33.55M, 0.05M)
    macs_per_row = [66 * 512 * 16, 512 * 256 * 16 * 16, 256 * 128 * 16 * 64, 128 * 64 * 16 * 256, 64 * 28 * 28]
    names = ["convT0 66->512 (GEMM)", "convT1 512->256 k4s2", "convT2 256->128 k4s2", "convT3 128->64 k4s2",
             "convT4 64->1 k1 + tanh"]
    for op, macs, name in zip(igemms, macs_per_row, names):
        st = stream_ptr()
        for _ in range(3):
            L.run_one(op, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.run_one(op, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flop = 2.0 * macs * rows
        per_layer.append(dict(layer=name, ms=ms, gflop=flop / 1e9, tflops=flop / ms / 1e9))
    # the dominant kernel: the longest launch of the three 34-GFLOP layers.  Since round 3 they are within ~2 % of each other
    # (263-270 us); among those within 2 % of the longest the LAST layer is reported (the one with the most HBM traffic, and the
    # one every earlier round reported), so that the line does not flip between kernels from run to run
    longest = max(d["ms"] for d in per_layer[1:4])
    dom = [d for d in per_layer[1:4] if d["ms"] >= 0.98 * longest][-1]
    traffic, traffic_src = (None, None)
    if live_traffic:
        traffic, traffic_src = _live_pmc_traffic(per_layer.index(dom))
        if traffic is None:
            live_fail = traffic_src
            traffic, traffic_src = _pmc_traffic(dom["layer"])
            traffic_src = f"{traffic_src} (live collection failed: {live_fail})"
    else:
        traffic, traffic_src = _pmc_traffic(dom["layer"])
    roof = {"bound": "mfma", "kernel": "igemm_fast_kernel (fp32 MFMA 32x32x2), " + dom["layer"],
            "achieved": round(dom["tflops"], 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(dom["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launch_ms": round(dom["ms"], 4), "algorithmic_gflop_per_launch": round(dom["gflop"], 2),
            "per_layer": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()} for d in per_layer]}
    return roof


def _op_class_and_flop(op):
    """(class name, algorithmic FLOPs) of one traced libgode op.  Convolution GEMMs: 2 * output positions * Cout *
    Cin * taps in conv orientation (the stride-phase decomposition spends no MFMA on structural zeros, so this is
    also what a ConvTranspose / input-gradient launch executes); the generator's first layer is counted with its 66
    real latent columns, not the 96 it is padded to.  ODE: 61.4 kFLOP per trajectory forward, the
    adjoint recomputes the stages and adds the two vector-Jacobian products (3x)."""
    import gan_ode_amd._lib as L
    if isinstance(op, str):
        return ("ode" if op.startswith("ode") else op), 0.0
    k = op.KIND
    if k in (L.OP_IGEMM, L.OP_WGRAD):
        g = op.g
        co = 66 if (g.Co == 96 and g.Ho == 1 and g.Wo == 1) else g.Co
        fl = 2.0 * g.N * g.Do * g.Ho * g.Wo * co * g.Ci * g.kd * g.kh * g.kw
        if k == L.OP_WGRAD:
            return "wgrad", fl
        return ("conv_fwd/convT_dgrad" if op.dir == L.FPROP else "convT_fwd/conv_dgrad"), fl
    if k == L.OP_ODE_FWD:
        return "ode", op.N * (61440.0 * max(1, op.substeps) + 4096.0)
    if k == L.OP_ODE_BWD:
        return "ode", op.N * (3 * 61440.0 * max(1, op.substeps) + 8192.0)
    return {L.OP_BN_FINALIZE: "batchnorm", L.OP_BN_BWD: "batchnorm", L.OP_BN_APPLY: "batchnorm", L.OP_BCE: "loss",
            L.OP_ADAM: "adam", L.OP_PACK: "weight_pack", L.OP_ODERNN_FWD: "ode", L.OP_ODERNN_BWD: "ode"}.get(k, "other"), 0.0


def _iteration_roofline(tr, imgs, vids, it_ms):
    """One training iteration re-run with every libgode op launched between two stream events (L.TRACE): sums the
    algorithmic FLOPs of what was executed and the GPU time per op class.  `achieved` uses the un-traced wall time
    it_ms (the traced run serialises host and GPU and is only used for the split)."""
    import gan_ode_amd._lib as L
    L.TRACE = []
    try:
        tr.step(imgs, vids)
        torch.cuda.synchronize()
        trace = L.TRACE
    finally:
        L.TRACE = None
    flop, ms, n = {}, {}, {}
    for op, e0, e1 in trace:
        cls, fl = _op_class_and_flop(op)
        flop[cls] = flop.get(cls, 0.0) + fl
        ms[cls] = ms.get(cls, 0.0) + e0.elapsed_time(e1)
        n[cls] = n.get(cls, 0) + 1
    total = sum(flop.values())
    split = {c: {"launches": n[c], "ms": round(ms[c], 3), "gflop": round(flop[c] / 1e9, 2),
                 "tflops": round(flop[c] / ms[c] / 1e9, 1) if flop[c] else None} for c in sorted(ms, key=ms.get, reverse=True)}
    return {"bound": "mfma", "what": "one training iteration (2 x [image-D, video-D] + G), convolution GEMM FLOPs as executed "
            "(discriminator weight gradients skipped in the G step, sample_images pruned to the selected trajectories)",
            "achieved": round(total / it_ms / 1e9, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(total / it_ms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4), "algorithmic_gflop": round(total / 1e9, 2),
            "iteration_ms": round(it_ms, 3), "libgode_launches": len(trace),
            "traced_ms_by_class (each op between two events; includes ~2-3 us launch gap per op)": split}


def _cpu_baseline(budget_s=15.0, threads=None):
    """The oracle (CPU port of the reference path on stock torch kernels) timed on the host cores: same workload,
    bounded sample."""
    from oracle import mocogan_ref as M
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0); np.random.seed(0)
    gen, _, _ = M.build_mnist()
    with torch.no_grad():
        gen.sample_videos(B)
        n, t0 = 0, time.perf_counter()
        while True:
            gen.sample_videos(B)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 40:
                break
    return {"value": round(n * B / el, 2), "unit": "videos/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} calls of oracle sample_videos({B}) (ngf=64, T=16, train-mode BN, no_grad) in {el:.1f}s, "
                      f"os.cpu_count()={os.cpu_count()}, cgroup CPU quota={threads}"}


def _parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--train-steps", type=int, default=30, help="iterations for the G/D step / iteration timings (10 until round 3: too few for a figure steady to 1 %)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the HIP-graph replay timing of the training iteration")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="take roofline.traffic from the newest committed PMC file instead of measuring it in two rocprofv3 "
                         "child passes (~20 s each)")
    ap.add_argument("--spawn", action="store_true",
                    help="go through the rank launcher even for --gpus 1 (N > 1 always does unless a launcher already "
                         "set WORLD_SIZE)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the rank processes (nccl = RCCL; gloo only to rehearse several "
                         "ranks on ONE GPU, where RCCL refuses duplicate devices)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only build the plan and run the per-layer roofline launches (for `rocprofv3 --kernel-trace "
                         "--stats -- python bench.py --roofline-only`, whose per-kernel averages must agree with the "
                         "HIP-event timings printed here)")
    ap.add_argument("--iteration-only", type=int, default=0, metavar="K",
                    help="only run K training iterations after 3 warm-ups (for the rocprofv3 per-kernel table of one "
                         "iteration under profiles/)")
    ap.add_argument("--config", default="mnist", choices=["mnist", "ucf", "odernn"],
                    help="mnist = BASELINE configs[1] (default, the headline); ucf = configs[3] shapes (batch 16, "
                         "3x64x64, rk4 as the code does); odernn = configs[4] (ODE-RNN latent, batch 32)")
    ap.add_argument("--ode-step-size", type=float, default=None,
                    help="torchdiffeq options={'step_size': h} for the rk4 solve (0.05 = the \"RK4 20 steps\" of "
                         "BASELINE configs[1]); default: the reference's own call, 15 steps on the output times")
    ap.add_argument("--ode-method", default="rk4", choices=["rk4", "dopri5"],
                    help="rk4 = the reference's call; dopri5 = torchdiffeq's adaptive default (BASELINE configs[3] wording)")
    return ap.parse_args()


def main():
    a = _parse()
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not under_launcher and (a.gpus > 1 or a.spawn):
        # parent: no torch import, no GPU call -- start one fresh process per GPU and relay rank 0's line
        import importlib.util
        spec = importlib.util.spec_from_file_location("_gode_launch", os.path.join(REPO, "gan-ode_amd", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        argv = [sys.executable, os.path.abspath(__file__)] + [x for x in sys.argv[1:] if x != "--spawn"]
        code, out = launch.spawn_ranks(a.gpus, argv)
        sys.stdout.write(out)
        sys.stdout.flush()
        raise SystemExit(code)
    _rank_main(a)


def _rank_main(a):
    global np, torch, dist, B
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    distributed = world > 1 or os.environ.get("GODE_FORCE_DIST") == "1"   # (the latter: rehearse the RCCL path on 1 GPU)
    out = sys.stdout
    if distributed:
        # RCCL prints a version banner on the C-level stdout at initialisation; the contract is ONE JSON line on stdout.
        # Keep a private handle on the real stdout for that line and route everything else written to fd 1 to stderr.
        sys.stdout.flush()
        out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs visible (RCCL needs one device per rank)")
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    if distributed:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    import gan_ode_amd as G
    G.limit_host_threads()     # size torch's CPU pool to this rank's share of the cgroup quota (see its docstring)
    torch.manual_seed(1234 + rank); np.random.seed(1234 + rank)
    C_, HW = 1, 28
    if a.config == "ucf":
        gen, dv, di = G.build_ucf()
        B, C_, HW = 16, 3, 64
    elif a.config == "odernn":
        gen, dv, di = G.build_mnist()
        gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16)
    else:
        gen, dv, di = G.build_mnist()
    gen.cuda(); dv.cuda(); di.cuda()
    if a.ode_step_size is not None:
        gen.ode_step_size = a.ode_step_size
    if a.ode_method != "rk4" and a.config != "odernn":
        gen.ode_method = a.ode_method

    if a.roofline_only:
        out.write(json.dumps({"roofline": _kernel_roofline(gen)}) + "\n")
        out.flush()
        return

    # the trainer broadcasts rank 0's parameters and buffers to every replica when world > 1
    tr = G.GanTrainer(gen, dv, di)
    g = torch.Generator().manual_seed(99 + rank)
    imgs = [torch.rand(B, C_, HW, HW, generator=g).cuda() for _ in range(2)]
    vids = [torch.rand(B, T, C_, HW, HW, generator=g).cuda() for _ in range(2)]

    if a.iteration_only:
        for _ in range(3):
            tr.step(imgs, vids)
        it_ms = _timed(lambda: tr.step(imgs, vids), a.iteration_only, 0, distributed) / a.iteration_only * 1e3
        if rank == 0:
            out.write(json.dumps({"iteration_ms": round(it_ms, 3), "iterations": a.iteration_only}) + "\n")
            out.flush()
        if distributed:
            dist.destroy_process_group()
        return

    # G / D step and whole-iteration milliseconds (synthetic real data resident on the GPU).  These run BEFORE the
    # headline loop: they build every plan and bring the clocks up, so that the W warm-up steps of the headline are
    # spent on the headline's own steady state (round 1: the first ~25 ms of GEMM load in a process ran 8 % slow).
    k = max(1, a.train_steps)
    wtr = max(1, min(3, k))
    d_ms = _timed(lambda: (tr.d_image_step(imgs[0]), tr.d_video_step(vids[0])), k, wtr, distributed) / k * 1e3
    g_ms = _timed(lambda: tr.g_step(B), k, wtr, distributed) / k * 1e3
    it_ms = _timed(lambda: tr.step(imgs, vids), k, wtr, distributed, after_warmup=G.freeze_host_gc) / k * 1e3
    # multi-GPU diagnosis: what the five all-reduces of an iteration cost on their own, and what the iteration costs without
    # them (the replicas' weights drift apart in that leg -- it runs last among the training timings and only its clock is used)
    coll_ms, it_noar_ms = {}, None
    if distributed:
        tr.flush_all()
        coll_ms = tr.time_collectives()
        tr.skip_allreduce = True
        it_noar_ms = _timed(lambda: tr.step(imgs, vids), k, 1, distributed) / k * 1e3
        tr.skip_allreduce = False
    # the same iteration replayed from a HIP graph (single process only): the host then only draws the noise, uploads
    # it and launches the graph (GanTrainer(graph=True)); bit-identical results (tests/test_gpu_api.py)
    it_graph_ms = None
    if not distributed and not a.no_graph:
        trg = G.GanTrainer(gen, dv, di, graph=True)
        trg.gen_opt, trg.vid_opt, trg.img_opt = tr.gen_opt, tr.vid_opt, tr.img_opt     # continue the same optimiser state
        it_graph_ms = _timed(lambda: trg.step(imgs, vids), k, max(3, wtr), distributed) / k * 1e3

    def sample():
        with torch.no_grad():
            gen.sample_videos(B)

    # the headline: W untimed warm-up steps, then exactly K timed steps between barrier + synchronize, max over ranks.
    # freeze_host_gc keeps Python's full-heap garbage-collection pass (75-100 ms) out of the loop (see its docstring)
    dt = _timed(sample, a.steps, a.warmup, distributed, after_warmup=G.freeze_host_gc)
    vps = world * B * a.steps / dt

    # per-step spread (rank 0): GPU time between stream markers recorded after every step of a second, untimed-by-the-
    # contract pass of min(steps, 200) steps -- reported beside the headline, never used for `value`
    spread = None
    if rank == 0:
        k2 = min(a.steps, 200)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(k2 + 1)]
        marks[0].record()
        for i in range(k2):
            sample()
            marks[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(k2))
        spread = {"p10": round(per[int(0.1 * (k2 - 1))], 4), "p50": round(per[(k2 - 1) // 2], 4),
                  "p90": round(per[int(0.9 * (k2 - 1))], 4), "steps": k2}

    if distributed:
        dist.barrier()       # every rank leaves the collective phase before rank 0 goes off to its single-rank legs
    if rank == 0:
        # (live PMC passes only in single-GPU runs: in a distributed run the other ranks would sit in the final barrier)
        roof = _kernel_roofline(gen, live_traffic=not a.no_live_traffic and not distributed) if a.config == "mnist" else None
        it_roof = _iteration_roofline(tr, imgs, vids, min(it_ms, it_graph_ms) if it_graph_ms else it_ms) if not distributed else None
        cpu = None if a.no_cpu_baseline or distributed or a.config != "mnist" else _cpu_baseline(threads=G.host_cpu_quota())
        workload = {"mnist": "Rotated-MNIST MoCoGAN+ODE, gen.sample_videos(32): batch 32/GPU, 16x1x28x28, ngf=ndf=64, "
                             "rk4 (Kutta 3/8) on linspace(0,1,16) = 15 steps as the reference code does, train-mode BN, "
                             "random-init weights",
                    "ucf": "UCF101-shaped MoCoGAN+ODE, gen.sample_videos(16): batch 16/GPU, 16x3x64x64, ngf=ndf=64, "
                           "rk4 as the reference code does, dim_hidden=16",
                    "odernn": "Rotated-MNIST MoCoGAN+ODE-RNN, gen.sample_videos(32): batch 32/GPU, dopri5 (1e-7/1e-9) "
                              "+ GRUCell per frame"}[a.config]
        if a.ode_method != "rk4" and a.config != "odernn":
            workload += f"; ODE solved with {a.ode_method} (rtol 1e-7, atol 1e-9) instead of the reference's rk4 call"
        if a.ode_step_size is not None:
            workload += (f"; rk4 options step_size={a.ode_step_size} (solver grid of "
                         f"{int(np.ceil(1 / a.ode_step_size))} steps, outputs interpolated as torchdiffeq does)")
        # one flat fp32 bucket per network = its GradArena (the generator's includes the dead GRU cell's slots, which
        # stay zero: the arena is reduced whole, without packing copies)
        nets = (("gen", gen), ("dis_vid", dv), ("dis_img", di))
        arena_bytes = {n_: 4 * (tr.arenas[id(m)].flat.numel() if id(m) in tr.arenas
                                else sum(p.numel() for p in m.parameters())) for n_, m in nets}
        allreduce = {"collectives_per_iteration": 2 * tr.d_iters + 1 if world > 1 else 0,
                     "bytes_per_iteration_per_rank": (tr.d_iters * (arena_bytes["dis_img"] + arena_bytes["dis_vid"])
                                                      + arena_bytes["gen"]) if world > 1 else 0,
                     "bucket_bytes": arena_bytes, "backend": (a.backend if distributed else None),
                     # each collective timed on its own (same buffers, idle GPU), their sum per iteration, and what the
                     # iteration pays for them in place (iteration_ms minus the same schedule without collectives): with
                     # overlap_allreduce the second should stay well below the first
                     "ms_per_collective": {k_: round(v, 4) for k_, v in coll_ms.items()} if coll_ms else None,
                     "ms_per_iteration": round(tr.d_iters * (coll_ms.get("dis_img", 0.0) + coll_ms.get("dis_vid", 0.0))
                                               + coll_ms.get("gen", 0.0), 4) if coll_ms else None,
                     "iteration_ms_without_allreduce": None if it_noar_ms is None else round(it_noar_ms, 3),
                     "exposed_ms_per_iteration": None if it_noar_ms is None else round(it_ms - it_noar_ms, 3),
                     "overlapped": bool(tr.overlap_ar) if world > 1 else None}
        line = {
            "metric": "generated videos/sec (16-frame clips)", "value": round(vps, 2), "unit": "videos/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * B, "parallelism": f"dp{world}"},
            "global_batch": world * B,
            "step_ms_spread": spread,
            "d_step_ms": round(d_ms, 3), "g_step_ms": round(g_ms, 3),
            "iteration_ms": round(min(it_ms, it_graph_ms) if it_graph_ms else it_ms, 3),
            "iteration_ms_eager": round(it_ms, 3),
            "iteration_ms_graph": None if it_graph_ms is None else round(it_graph_ms, 3),
            "train_videos_per_s": round(world * B / ((min(it_ms, it_graph_ms) if it_graph_ms else it_ms) / 1e3), 2),
            "allreduce": allreduce,
            "roofline": roof, "iteration": it_roof, "cpu_baseline": cpu,
        }
        out.write(json.dumps(line) + "\n")
        out.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
