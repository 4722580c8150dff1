"""CPU: the single-node rank launcher behind `python bench.py --gpus N` (gan-ode_amd/launch.py) driven with a stub rank
program over gloo, and bench.py's parent path (no torch import, no GPU call, children's failure propagated)."""
import importlib.util
import json
import os
import subprocess
import sys

from conftest import REPO


def _launch():
    spec = importlib.util.spec_from_file_location("_gode_launch", os.path.join(REPO, "gan-ode_amd", "launch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


STUB = os.path.join(REPO, "tests", "_stub_rank.py")


def test_launcher_is_standard_library_only():
    src = open(os.path.join(REPO, "gan-ode_amd", "launch.py")).read()
    assert "import torch" not in src and "import numpy" not in src


def test_spawn_two_ranks_gloo_relays_rank0_line():
    code, out = _launch().spawn_ranks(2, [sys.executable, STUB], timeout=180)
    assert code == 0
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    rec = json.loads(lines[0])
    assert rec == {"n_gpus": 2, "sum": 3.0, "local_world": "2", "spawned": "1"}
    assert "rank 1 says hello" not in out and "rank 0 says hello" in out


def test_failed_rank_fails_the_launch_and_stops_its_peers():
    code, out = _launch().spawn_ranks(2, [sys.executable, STUB, "fail"], timeout=120)
    assert code == 3                       # the failing rank's exit code, not a hang until the store times out
    assert "{" not in out


def test_bench_parent_does_not_import_torch_and_propagates_failure():
    """Without a GPU the rank processes cannot start their work; the parent must come back non-zero, having neither
    imported torch nor printed a JSON line.  (On the GPU box the same path prints rank 0's line: -m gpu test.)"""
    probe = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0'];\n"
             "try:\n    runpy.run_path(r'%s', run_name='__main__')\nexcept SystemExit as e:\n"
             "    print('EXIT', e.code, 'torch' in sys.modules)\n" % os.path.join(REPO, "bench.py"))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300, env=env)
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("EXIT")]
    assert last, (r.stdout, r.stderr)
    _, code, torch_loaded = last[-1].split()
    assert code != "0" and torch_loaded == "False"
    assert not any(ln.startswith('{"metric"') for ln in r.stdout.splitlines())
