"""Rank program for tests/test_gpu_dataparallel.py: two replicas of a small GanTrainer sharing the box's one GPU over gloo.
argv[1] = "overlap" | "serial" (GanTrainer(overlap_allreduce=...)), argv[2] = generator ("ode" | "odernn").
Runs three iterations on rank-specific data and prints, from rank 0, one JSON line with the losses and a digest of every
weight, BatchNorm buffer and Adam moment; every rank asserts that it ended with rank 0's weights."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import gan_ode_amd as G

mode, which = sys.argv[1], sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
G.limit_host_threads()
torch.manual_seed(7 + rank); np.random.seed(7 + rank)        # different initial weights per rank: the trainer broadcasts rank 0's
gen, dv, di = G.build_mnist(ngf=16, ndf=16)
if which == "odernn":
    gen = G.VideoGeneratorMNISTODERNN(1, 50, 0, 16, 16, ngf=16)
gen.cuda(); dv.cuda(); di.cuda()
tr = G.GanTrainer(gen, dv, di, overlap_allreduce=(mode == "overlap"))
rng = torch.Generator().manual_seed(100 + rank)
losses = []
for it in range(3):
    imgs = [torch.rand(8, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
    vids = [torch.rand(8, 16, 1, 28, 28, generator=rng).cuda() for _ in range(2)]
    torch.manual_seed(1000 * rank + it); np.random.seed(1000 * rank + it)
    losses.append([float(v) for v in tr.step(imgs, vids)])
torch.cuda.synchronize()
assert not tr._pending
h = hashlib.sha256()
for m in (gen, dv, di):
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            continue                                         # per-replica BatchNorm statistics differ by design
        h.update(v.detach().cpu().numpy().tobytes())
for opt, m in ((tr.gen_opt, gen), (tr.vid_opt, dv), (tr.img_opt, di)):
    for p in m.parameters():
        st = opt.state.get(p)
        if st:
            h.update(st["exp_avg"].cpu().numpy().tobytes())
digest = h.hexdigest()
all_d = [None] * world
dist.all_gather_object(all_d, digest)
assert all(d == all_d[0] for d in all_d), "replicas diverged"
if rank == 0:
    print(json.dumps({"mode": mode, "digest": digest, "losses": losses}))
dist.barrier()
dist.destroy_process_group()
