"""`python bench.py --gpus N` started WITHOUT a launcher must produce N ranks by itself (the driver starts the N = 1 case plainly;
a plain start with N > 1 used to render the whole frame on one GPU and report n_gpus 1).  Rehearsed here without a GPU: --rehearse
runs everything of the multi-rank path except the render (self-launch -> torch.distributed.run -> process group -> band plan ->
parallel.gather_film_rows in its default mode -> rank 0's JSON line) over gloo."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(n, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PTRS_DIST_BACKEND"] = "gloo"
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rehearse", "--steps", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


@pytest.mark.parametrize("n", [2, 3])
def test_plain_start_with_gpus_n_launches_n_ranks(n):
    p, lines = _run(n)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "exactly one JSON line (rank 0's): %r" % (p.stdout,)
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["rehearsal"] is True and j["gathered_film_ok"] is True
    d = j["config"]["dist"]
    assert d["world_size"] == n and d["backend"] == "gloo" and "self_launch" in d["launched_by"]
    b = j["config"]["band_plan"]
    assert len(b) == n + 1 and b[0] == 0 and b[-1] == 96 and len({b[i + 1] - b[i] for i in range(n)}) > 1, "cost-planned bands of unequal height went through the gather"


def test_single_gpu_start_launches_nothing():
    p, lines = _run(1)
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["dist"]["launched_by"] == "single process" and "launching" not in p.stderr


def test_child_failure_is_the_parents_exit_code():
    """a rank that dies takes the launch down and the parent reports it (no JSON line, non-zero exit)"""
    p, lines = _run(2, {"PTRS_DIST_BACKEND": "no-such-backend"})
    assert p.returncode != 0 and not lines
