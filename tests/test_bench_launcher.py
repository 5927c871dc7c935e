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


def test_self_launch_eight_ranks():
    """The scaling run's own shape, `--gpus 8`: eight self-launched ranks meet, shard 16 streams (stream s on rank
    s mod 8), gather them in global order, and rank 0 reports every rank's elapsed time."""
    p = _run("--gpus", "8", "--streams-per-gpu", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["gather_ok"] and rec["gathered_shape"][0] == 16
    assert len(rec["per_rank_elapsed_s"]) == 8 and max(rec["per_rank_elapsed_s"]) == rec["max_elapsed_s"]


def test_stream_cache_round_trip(tmp_path):
    """A stream synthesised once is loaded from the cache directory afterwards, bit for bit (ranks of later runs on
    one machine skip the seconds of scipy before the first barrier); a damaged file is synthesised again."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    old = (bench.H, bench.W, bench.N_KP)
    bench.H, bench.W, bench.N_KP = 48, 64, 20          # a small frame: the protocol, not the size
    try:
        a = bench.make_streams([3, 11], 2, cache_dir=str(tmp_path))
        files = sorted(os.listdir(tmp_path))
        assert len(files) == 2 and all(f.endswith(".npz") for f in files)
        b = bench.make_streams([3, 11], 2, cache_dir=str(tmp_path))
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        with open(os.path.join(tmp_path, files[0]), "wb") as f:
            f.write(b"not an archive")
        c = bench.make_streams([3, 11], 2, cache_dir=str(tmp_path))
        for x, y in zip(a, c):
            assert np.array_equal(x, y)
        # the true flow: a keypoint of frame 0 lies at moved[1] in frame 1 (ring-wise), within the synthetic motion's range
        frames, pts, moved = a
        assert moved.shape == (2, 2, 20, 2) and np.abs(moved[1] - pts).max() < 8.0
        X, poses = bench.make_landmarks([3, 11], moved, 0.0)
        K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]])
        proj = (K @ (poses[1, 0, :, :3] @ X[1, 0].T.astype(np.float64) + poses[1, 0, :, 3:])).T
        assert np.abs(proj[:, :2] / proj[:, 2:] - moved[1, 0]).max() < 1e-2       # landmarks project onto the flow (float32 storage)
    finally:
        bench.H, bench.W, bench.N_KP = old


def test_frame_mode_figure_never_breaks_the_headline(monkeypatch):
    """The default line's `frame_mode` field comes from a child process; a child that fails or prints nothing usable
    yields an `error` entry, a good one the figure with its command."""
    import json as _json
    import subprocess as _sp
    import bench

    class _Done:
        def __init__(self, out):
            self.stdout, self.stderr, self.returncode = out, "", 0

    monkeypatch.setattr(_sp, "run", lambda *a, **k: _Done("warming up\n"))
    assert "error" in bench.frame_mode_figure()

    def boom(*a, **k):
        raise _sp.TimeoutExpired(cmd="bench", timeout=1)
    monkeypatch.setattr(_sp, "run", boom)
    assert "TimeoutExpired" in bench.frame_mode_figure()["error"]
    line = {"metric": "m", "value": 5.0, "unit": "frames/s", "steps": 300, "ms_per_step": 0.5,
            "config": {"streams_per_gpu": 64, "pose_check": {"streams_with_pose": 64}}}
    monkeypatch.setattr(_sp, "run", lambda *a, **k: _Done("noise\n" + _json.dumps(line) + "\n"))
    got = bench.frame_mode_figure(steps=300)
    assert got["value"] == 5.0 and got["pose_check"]["streams_with_pose"] == 64 and "--mode frame" in got["command"]
