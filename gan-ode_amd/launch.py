"""Single-node rank launcher: one fresh process per GPU, started BEFORE anything in the parent touches the GPU.

`python bench.py --gpus N` (N > 1, not already under torch.distributed.run) goes through here: the parent only
parses arguments, spawns N children with RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set (the torchrun environment contract), relays every child's stderr, prints what rank 0 wrote to
stdout and exits non-zero if any child did.  It never replaces its own process image and never initialises HIP
(a process that has initialised the GPU must not exec another program on this pool), which is why this file is
standard library only: bench.py loads it by path, without importing torch.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GODE_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this host driver
    return env


def _pump(stream, sink, prefix):
    for line in iter(stream.readline, b""):
        sink.write(prefix + line.decode(errors="replace"))
        sink.flush()
    stream.close()


def spawn_ranks(world, argv, timeout=None, env=None, port=None):
    """Starts `world` copies of argv (a full command line, e.g. [sys.executable, "bench.py", ...]), one per rank.
    Returns (exit_code, rank0_stdout): exit_code is 0 only if every rank exited 0; when one rank fails the others are
    terminated (a dead peer would otherwise leave them blocked in a collective until the store times out)."""
    port = port or free_port()
    procs, pumps = [], []
    out0 = []
    for r in range(world):
        p = subprocess.Popen(argv, env=rank_env(r, world, port, env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             start_new_session=True)
        procs.append(p)
        t = threading.Thread(target=_pump, args=(p.stderr, sys.stderr, f"[rank {r}] "), daemon=True)
        t.start()
        pumps.append(t)
        if r == 0:
            def _collect(stream=p.stdout):
                for line in iter(stream.readline, b""):
                    out0.append(line.decode(errors="replace"))
                stream.close()
            t0 = threading.Thread(target=_collect, daemon=True)
        else:
            t0 = threading.Thread(target=_pump, args=(p.stdout, sys.stderr, f"[rank {r} stdout] "), daemon=True)
        t0.start()
        pumps.append(t0)
    deadline = None if timeout is None else time.monotonic() + timeout
    code = 0
    live = set(range(world))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    sys.stderr.write(f"[launch] rank {r} exited with {rc}; stopping the other ranks\n")
                    for q in live:
                        _stop(procs[q])
        if deadline is not None and time.monotonic() > deadline and live:
            sys.stderr.write(f"[launch] timeout after {timeout}s; stopping ranks {sorted(live)}\n")
            code = code or 124
            for q in live:
                _stop(procs[q])
            deadline = None
        time.sleep(0.05)
    for t in pumps:
        t.join(5)
    return code, "".join(out0)


def _stop(p):
    """Ends exactly the process group this launcher started for that rank (never a pattern match)."""
    if p.poll() is not None:
        return
    try:
        os.killpg(p.pid, 15)
    except ProcessLookupError:
        return
    try:
        p.wait(10)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, 9)
        except ProcessLookupError:
            pass
