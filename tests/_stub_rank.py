"""Stub rank program for tests/test_launcher.py: what bench.py's rank processes do, minus the GPU -- join the group
the launcher's environment describes (gloo), all-reduce, and let rank 0 print ONE JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "fail" and rank == 1:
    sys.stderr.write("rank 1 giving up on purpose\n")
    sys.exit(3)
dist.init_process_group("gloo")                 # MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE from the launcher
t = torch.tensor([float(rank + 1)])
if mode == "fail":
    # rank 0 would wait here for the dead peer; the launcher must stop it instead of hanging
    dist.all_reduce(t)
    sys.exit(0)
dist.all_reduce(t)
dist.barrier()
print(f"rank {rank} says hello on stdout")       # only rank 0's stdout is relayed to the parent's stdout
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": float(t), "local_world": os.environ["LOCAL_WORLD_SIZE"],
                      "spawned": os.environ.get("GODE_SPAWNED")}))
dist.destroy_process_group()
