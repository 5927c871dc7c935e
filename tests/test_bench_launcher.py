"""`python bench.py --gpus N` must start its own ranks.  Rehearsed on CPU: --dry-run runs the
rank protocol (rendezvous, sharding, barriers, MAX over ranks, final gather of per-stream tracks)
over gloo without launching a kernel."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--dry-run", *extra],
                          env=env, capture_output=True, text=True, timeout=240)


def test_self_launch_two_ranks():
    p = _run("--gpus", "2", "--streams-per-gpu", "3")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] and rec["gather_ok"]
    assert rec["gathered_shape"][0] == 6        # every stream of both ranks, in global order


def test_self_launch_strong_scaling_split():
    p = _run("--gpus", "2", "--total-streams", "8")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["streams"] == 8


def test_failed_rank_fails_the_launch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "no-such-backend", "--dry-run",
                        "--gpus", "2"], capture_output=True, text=True, timeout=240,
                       env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert p.returncode != 0


def test_single_rank_dry_run_needs_no_launcher():
    p = _run("--gpus", "1", "--streams-per-gpu", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["gather_ok"]
