#!/usr/bin/env python3
"""Headline benchmark: tracked frames/s of the pyramid + pyramidal-LK hot path on
synthetic KITTI-shaped frames (1241x376, 2000 keypoints, 4-level pyramid).

A step = every local stream advances one frame: the new frame (already resident
in HBM) gets its pyramid built and the stream's 2000 keypoints are tracked from
the previous frame's pyramid (kept from the step before) into it.  Streams are
independent, so N GPUs run N x streams-per-GPU streams with no per-frame
communication (weak scaling); ranks only meet at the barriers around the timed
region and in a final gather of every stream's tracked positions.

Launching: `python bench.py --gpus N` starts N rank processes by itself (the parent
never touches a GPU); under `python -m torch.distributed.run` (WORLD_SIZE set) the
process is one of the ranks.  Prints ONE JSON line on rank 0.

`--mode frame` (the default `--mode klt` is the headline above) times the whole tracked frame of every stream --
pyramid, tracker, status filter, P3P RANSAC (the reference's KITTI parameters: 1 px, confidence 0.999, 100
iterations) and the refit -- through vo_track_pnp_batch: frames/s with a pose per frame.  Each stream's
landmarks project onto the true positions of its keypoints in the next frame under a seeded pose, 10 % of them
displaced (outliers).  A group begins its step's chain, and waits for the chain of its previous step only when it
comes round again: one synchronisation per step and group, the other groups' chains running meanwhile.
The default single-GPU line carries that figure as `frame_mode`: a child process runs `--mode frame` for 300 steps
before this process touches the GPU (`--no-frame-line`, or `--no-cpu-baseline` as profiling runs pass it, leave it out).

A GPU's streams are dealt to `--groups` launch groups (default 2).  Each group owns a
tracker HIP stream and a pyramid HIP stream and issues one tracker launch per step, so
the launches of different groups overlap: one group's launch fills the drain of the
other's.  roofline: `launch_ms` is the mean duration of one tracker launch, taken with
HIP events on the launch's own stream around every fourth launch of every group in the
timed region (an event pair costs a few microseconds of idle queue, so not every step
carries one); `launches_in_flight` = (sum of all launch durations) / (timed wall time) is
the measured overlap; `achieved` = tracker algorithmic bytes of one launch x
launches_in_flight / launch_ms, i.e. the bytes per second the kernel's launches process
while it runs on the device (`achieved_per_launch` leaves the overlap out).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 376, 1241
N_KP = 2000
MAX_LEVEL = 3            # "4-level pyramid": levels 0..3
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak


def level_pixels(max_level):
    w, h, px = W, H, []
    for _ in range(max_level + 1):
        px.append(w * h)
        w, h = (w + 1) // 2, (h + 1) // 2
    return px


def algorithmic_bytes(max_level, n_kp):
    """Per tracked frame: read the new frame, write its upper pyramid levels, the
    tracker reads the previous and the next pyramid once, plus 21 B per keypoint
    (8 in, 8 out, 1 status, 4 min-eigenvalue)."""
    px = level_pixels(max_level)
    pyramid = px[0] + sum(px[1:])
    tracker = 2 * sum(px) + n_kp * 21
    return pyramid + tracker, tracker


def _synth_stream(seed, ring):
    """One stream: `ring` frames of a texture seen through successive small warps, its keypoints, and where those
    keypoints -- placed in frame r-1 -- lie in frame r (`moved[r]`, the true flow of the pair r-1 -> r, ring-wise)."""
    from scipy import ndimage
    from visual_odometry_amd import synth
    frames = np.empty((ring, H, W), np.uint8)
    margin = 40
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    rng = np.random.default_rng(seed)
    tex = synth.texture((H, W), rng, margin)
    off = np.zeros(2)
    offs, affs = [], []
    c = np.array([(W - 1) / 2, (H - 1) / 2])
    for r in range(ring):
        if r:
            off = off + rng.uniform(-6, 6, 2)
            off = np.clip(off, -30, 30)
        aff = np.eye(2) + (rng.uniform(-0.005, 0.005, (2, 2)) if r else 0)
        sx = aff[0, 0] * (xs - c[0]) + aff[0, 1] * (ys - c[1]) + c[0] - off[0] + margin
        sy = aff[1, 0] * (xs - c[0]) + aff[1, 1] * (ys - c[1]) + c[1] - off[1] + margin
        img = ndimage.map_coordinates(tex, [sy, sx], order=1, mode="nearest")
        img = img + rng.normal(0, 1.0, img.shape)
        frames[r] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        offs.append(off.copy())
        affs.append(aff.copy())
    pts = synth.uniform_keypoints((H, W), N_KP, seed=seed, margin=11)
    moved = np.empty((ring, N_KP, 2), np.float64)
    for r in range(ring):
        q = (r - 1) % ring          # texture coordinate of x in frame q: A_q (x - c) + c - off_q; the same in frame r
        u = (pts.astype(np.float64) - c) @ affs[q].T - offs[q] + offs[r]
        moved[r] = u @ np.linalg.inv(affs[r]).T + c
    return frames, pts, moved


def make_streams(stream_ids, ring, base_seed=2021, cache_dir=None):
    """Per stream a ring of `ring` frames, a seeded keypoint set and the keypoints' true flow; a stream's data depends
    on its global id only.  cache_dir: a stream synthesised once is kept there (keyed by seed and ring), so ranks of
    later runs on the same machine load it instead of spending seconds in scipy before the first barrier."""
    from visual_odometry_amd import streams
    n_streams = len(stream_ids)
    frames = np.empty((ring, n_streams, H, W), np.uint8)
    pts = np.empty((n_streams, N_KP, 2), np.float32)
    moved = np.empty((ring, n_streams, N_KP, 2), np.float64)
    for s, sid in enumerate(stream_ids):
        seed = streams.stream_seed(base_seed, sid)
        path = os.path.join(cache_dir, f"stream_{seed}_{ring}_{H}x{W}_{N_KP}.npz") if cache_dir else None
        got = None
        if path and os.path.exists(path):
            try:
                with np.load(path) as z:
                    got = (z["frames"], z["pts"], z["moved"])
                if got[0].shape != (ring, H, W) or got[1].shape != (N_KP, 2) or got[2].shape != (ring, N_KP, 2):
                    got = None
            except Exception:
                got = None
        if got is None:
            got = _synth_stream(seed, ring)
            if path:
                try:
                    os.makedirs(cache_dir, exist_ok=True)
                    tmp = f"{path}.{os.getpid()}.tmp.npz"
                    np.savez(tmp, frames=got[0], pts=got[1], moved=got[2])
                    os.replace(tmp, path)          # atomic: a concurrent rank sees the whole file or none
                except Exception:
                    pass
        frames[:, s], pts[s], moved[:, s] = got
    return frames, pts, moved


def make_landmarks(stream_ids, moved, outliers, base_seed=2021):
    """Per ring slot r and stream: landmarks (N_KP,3) float32 that project onto moved[r] under a seeded pose of the
    KITTI camera, a fraction `outliers` of them displaced by ~3 units -> (X (ring,S,N_KP,3), poses (ring,S,3,4))."""
    from visual_odometry_amd import streams, synth
    Kd = synth.K_KITTI.astype(np.float64)
    ring, S = moved.shape[:2]
    X = np.empty((ring, S, N_KP, 3), np.float32)
    poses = np.empty((ring, S, 3, 4))
    for s, sid in enumerate(stream_ids):
        rng = np.random.default_rng(streams.stream_seed(base_seed, sid) + 7919)
        for r in range(ring):
            R = synth.rodrigues(rng.normal(0, 0.02, 3))
            t = rng.normal(0, 0.3, 3)
            z = rng.uniform(4, 40, N_KP)
            rays = np.linalg.solve(Kd, np.hstack([moved[r, s], np.ones((N_KP, 1))]).T).T
            Xw = (R.T @ (rays * z[:, None] - t).T).T
            bad = rng.uniform(size=N_KP) < outliers
            Xw[bad] += rng.normal(0, 3.0, (int(bad.sum()), 3))
            X[r, s] = Xw.astype(np.float32)
            poses[r, s] = np.hstack([R, t[:, None]])
    return X, poses


def host_cores():
    """Worker threads for the all-cores CPU column: the process's CPU share, at most 16
    (a one-GPU slice of a shared box shows every CPU of the machine in os.cpu_count())."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            n = min(n, int(v))
    return max(1, min(n, 16))


def cpu_baseline(frames, pts, max_level=MAX_LEVEL, budget_s=10.0, cores=0, frame_mode=None):
    """CPU oracle on the same workload (bounded sample): one thread, then `cores` threads
    (the C oracle is called through ctypes, which releases the interpreter lock).  frame_mode = (landmarks
    (ring,S,N,3), K, thr, conf, max_iters): each pair also runs the status filter and the CPU restatement of
    cv.solvePnPRansac(P3P) with its refit, as a tracked frame of the reference does."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    ring, n_streams = frames.shape[:2]

    def pair(i):
        s, r = i % n_streams, (i // n_streams) % ring
        out, st, _ = oracle.klt(frames[r, s], frames[(r + 1) % ring, s], pts[s], max_level=max_level)
        if frame_mode is not None:
            from oracle.pnp_ransac import cpu_pnp_ransac
            X, Kd, thr, conf, iters = frame_mode
            keep = st.ravel() == 1
            cpu_pnp_ransac(X[(r + 1) % ring, s][keep], out[keep], Kd, thr, conf, iters)

    done, t0 = 0, time.perf_counter()
    while True:
        pair(done)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s and done >= 3:
            break
    what = "tracked + PnP-RANSAC'd (C oracle LK, numpy P3P, C scoring, least-squares refit)" if frame_mode is not None else "tracked"
    res = {"value": done / el, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} frame pairs of the same streams {what} ({N_KP} keypoints each, pyramids of both frames "
                     f"rebuilt per pair as the reference's cv2 call does), {el:.1f} s"}
    cores = cores or host_cores()
    if cores > 1:
        per = max(1, int(round(done / el * budget_s)))    # pairs one thread finishes in the budget
        total = per * cores
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            list(pool.map(pair, range(total)))
        el = time.perf_counter() - t0
        res["all_cores"] = {"value": total / el, "unit": "frames/s", "cores": cores, "kind": "port",
                            "sample": f"{total} frame pairs over {cores} threads, {el:.1f} s"}
    return res


def frame_mode_figure(steps=300):
    """The tracked frame with a pose (`--mode frame`) measured by a child process of this script, for the default
    line's `frame_mode` field.  Called before this process touches torch or the GPU: the child has the device to
    itself and has exited when the headline measurement starts."""
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "frame", "--steps", str(steps), "--warmup", "20", "--no-cpu-baseline"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        rec = json.loads(line)
        return {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "steps": rec["steps"], "ms_per_step": rec["ms_per_step"],
                "streams_per_gpu": rec["config"]["streams_per_gpu"], "pose_check": rec["config"].get("pose_check"),
                "command": "python bench.py --mode frame --steps %d --warmup 20 --no-cpu-baseline" % steps}
    except Exception as e:      # the headline line does not depend on it
        return {"error": "%s: %s" % (type(e).__name__, e)}


def launch_ranks(n, argv):
    """Parent of a self-launched multi-GPU run: start one fresh process per rank, wait, and pass
    on the first failure.  This process never initialises a GPU and replaces no running program."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:           # a rank failed: the others would wait at a barrier for ever
                    q.terminate()
    return rc


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", choices=("klt", "frame"), default="klt",
                    help="klt (headline): pyramid + tracker per frame; frame: the whole tracked frame with a pose "
                         "(pyramid, tracker, status filter, P3P RANSAC, refit) through vo_track_pnp_batch")
    ap.add_argument("--priorities", default="",
                    help="HIP stream priorities 'tracker,pyramid' (lower = more urgent; default -1,0)")
    ap.add_argument("--depth", type=int, default=1,
                    help="frame mode: batches a group keeps in flight (2: a step's chain is begun before the host waits for "
                         "the step before, so the group's HIP stream never idles through the host's turnaround; measured "
                         "equal to 1 on an idle host: the other group's chain already covers the turnaround)")
    ap.add_argument("--refine-iters", type=int, default=10, help="frame mode: Gauss-Newton passes of the refit (0: none; development aid)")
    ap.add_argument("--outliers", type=float, default=0.1, help="frame mode: fraction of displaced landmarks per stream")
    ap.add_argument("--stream-cache", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "vo_bench_streams"),
                    help="directory that keeps synthesised streams from run to run ('' = none)")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed steps are run for this long before the warmup steps (clock ramp)")
    ap.add_argument("--streams-per-gpu", type=int, default=64)
    ap.add_argument("--total-streams", type=int, default=0,
                    help="strong scaling: this many streams in all, divided over the GPUs (must divide evenly); "
                         "default 0 = weak scaling with --streams-per-gpu streams on every GPU")
    ap.add_argument("--groups", type=int, default=2,
                    help="launch groups per GPU: the GPU's streams are dealt to this many groups, each with its own "
                         "tracker/pyramid HIP streams and one tracker launch per step (1 = one launch per step)")
    ap.add_argument("--ring", type=int, default=3, help="distinct frames per stream, cycled through")
    ap.add_argument("--max-level", type=int, default=MAX_LEVEL,
                    help="requested last pyramid level (default 3 = the headline 4-level configuration; the reference "
                         "asks for 8, which the level-size rule clamps to 4 on 1241x376 frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-line", action="store_true",
                    help="leave out the `frame_mode` field of the default line (a `--mode frame` run in a child process first)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="threads of the all-cores CPU column (0 = the CPU share)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="build each frame's pyramid on the tracker's stream, right before it is tracked")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearsal of the multi-rank protocol without a GPU: rendezvous, sharding, barriers, MAX over "
                         "ranks and the final gather run on CPU tensors (use --backend gloo); no kernel is launched")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal aid: every rank uses GPU 0 (one-GPU box); never used for reported numbers")
    ap.add_argument("--distinct", type=int, default=0,
                    help="development aid: synthesise only this many distinct streams and repeat them (0 = all distinct)")
    return ap.parse_args(argv)


def dry_run(args, rank, world):
    """The rank protocol of the benchmark on CPU tensors: what a stream 'tracked' is a function of its
    global id, so rank 0 can check that the gather returned every stream's rows in global order."""
    import torch
    import torch.distributed as dist
    from visual_odometry_amd import streams
    dev = torch.device("cpu")
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    S = args.total_streams // world if args.total_streams else args.streams_per_gpu
    owned = streams.shard_streams(S * world, rank, world)
    n_kp = 8
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    local = torch.stack([torch.full((n_kp, 2), float(streams.stream_seed(2021, s))) for s in owned])
    if world > 1:
        dist.barrier()
    elapsed_local = time.perf_counter() - t0 + 1e-3 * (rank + 1)
    elapsed = streams.max_over_ranks(elapsed_local, dev, world)
    per_rank = streams.gather_scalars(elapsed_local, dev, world)
    tracks = streams.gather_per_stream(local, owned, S * world, world)
    ok = bool((tracks[:, 0, 0] == torch.tensor([float(streams.stream_seed(2021, s)) for s in range(S * world)])).all())
    if rank == 0:
        print(json.dumps({"metric": "tracked frames/s (KITTI 1241x376, 2k kp)", "value": None, "unit": "frames/s",
                          "dry_run": True, "n_gpus": world, "streams": S * world, "gather_ok": ok,
                          "gathered_shape": list(tracks.shape), "max_elapsed_s": elapsed,
                          "per_rank_elapsed_s": per_rank}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))     # before torch or HIP is touched in this process

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        sys.exit(dry_run(args, rank, world))
    frame_extra = None
    if args.mode == "klt" and world == 1 and not args.no_cpu_baseline and not args.no_frame_line:
        frame_extra = frame_mode_figure()

    import torch
    import torch.distributed as dist
    from visual_odometry_amd import hip, streams

    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    if args.total_streams:
        if args.total_streams % world:
            raise SystemExit(f"--total-streams {args.total_streams} does not divide over {world} GPUs")
        args.streams_per_gpu = args.total_streams // world
    S, R = args.streams_per_gpu, args.ring
    G = max(1, min(args.groups, S))
    frame_mode = args.mode == "frame"
    owned = streams.shard_streams(S * world, rank, world)      # weak scaling: S streams per GPU
    cache = args.stream_cache or None
    if args.distinct and args.distinct < S:
        f0, p0, m0 = make_streams(owned[:args.distinct], R, cache_dir=cache)
        rep = (S + args.distinct - 1) // args.distinct
        frames_h = np.tile(f0, (1, rep, 1, 1))[:, :S].copy()
        pts_h = np.tile(p0, (rep, 1, 1))[:S].copy()
        moved_h = np.tile(m0, (1, rep, 1, 1))[:, :S].copy()
        owned_data = (owned[:args.distinct] * rep)[:S]
    else:
        frames_h, pts_h, moved_h = make_streams(owned, R, cache_dir=cache)
        owned_data = owned
    frames = torch.from_numpy(frames_h).to(dev)          # (R, S, H, W) resident in HBM
    pts = torch.from_numpy(pts_h).to(dev)
    out_pts = torch.empty_like(pts)
    out_status = torch.empty((S, N_KP), dtype=torch.uint8, device=dev)
    out_err = torch.empty((S, N_KP), dtype=torch.float32, device=dev)
    if frame_mode:
        out_pts = [out_pts] + [torch.empty_like(pts) for _ in range(max(1, args.depth) - 1)]
        out_status = [out_status] + [torch.empty_like(out_status) for _ in range(max(1, args.depth) - 1)]
        out_mask = [torch.empty((S, N_KP), dtype=torch.uint8, device=dev) for _ in range(max(1, args.depth))]
        from visual_odometry_amd import params as vo_params, synth
        X_h, poses_true = make_landmarks(owned_data, moved_h, args.outliers)
        X = torch.from_numpy(X_h).to(dev)                # (R, S, N_KP, 3) resident in HBM
        K_cam = synth.K_KITTI.astype(np.float64)
        pnp = (float(vo_params.PNP_RANSAC_REPROJ_THRESHOLD), float(vo_params.PNP_RANSAC_CONFIDENCE), int(vo_params.PNP_RANSAC_MAX_ITERS))
        last = {"Rt": np.zeros((S, 3, 4)), "good": np.zeros(S, np.int32), "tracked": np.zeros(S, np.int32),
                "path": np.zeros(S, np.int32), "slot": 0, "d": 0, "slow": 0, "frames": 0}

    # Launch groups.  A group owns local streams lo..hi, two HIP streams and three pyramid slots.  Its
    # tracker runs on a high-priority HIP stream; the pyramid of the frame after next is built on a
    # normal-priority one, so its workgroups are dispatched where the trackers leave room.  Step k of a
    # group tracks slot k%3 -> (k+1)%3 while frame k+2 is built into slot (k+2)%3, which the tracker of
    # step k-1 was the last to read.  Every step still builds one pyramid and tracks one frame per stream.
    bounds = [S * g // G for g in range(G + 1)]
    prio = [int(v) for v in args.priorities.split(",")] if args.priorities else [-1, 0]
    groups = []
    for g in range(G):
        lo, hi = bounds[g], bounds[g + 1]
        if args.no_pipeline:
            s_klt = s_pyr = torch.cuda.Stream(device=dev, priority=prio[0]) if G > 1 else torch.cuda.current_stream(dev)
        else:
            s_klt = torch.cuda.Stream(device=dev, priority=prio[0])
            s_pyr = torch.cuda.Stream(device=dev, priority=prio[1])
        with torch.cuda.stream(s_klt):
            ctx = hip.Context.from_torch(local_rank)
        with torch.cuda.stream(s_pyr):
            ctx_pyr = hip.Context.from_torch(local_rank)
        q = {"n": hi - lo, "lo": lo, "hi": hi, "s_klt": s_klt, "s_pyr": s_pyr, "ctx": ctx,
             "frames": [frames[r, lo:hi] for r in range(R)], "pts": pts[lo:hi],
             "out": None if frame_mode else (out_pts[lo:hi], out_status[lo:hi], out_err[lo:hi]),
             "pyrs": [hip.Pyramid(ctx_pyr, hi - lo, H, W, args.max_level) for _ in range(3)],
             "pyr_done": [torch.cuda.Event() for _ in range(3)], "klt_done": [torch.cuda.Event() for _ in range(3)],
             "evs": [], "ms": []}
        if frame_mode:
            # a context holds one batch in flight: `depth` contexts on the group's tracker stream, used in turn, each
            # with its own output arrays
            D = max(1, args.depth)
            q["X"] = [X[r, lo:hi] for r in range(R)]
            q["ctxs"] = [ctx]
            with torch.cuda.stream(s_klt):
                q["ctxs"] += [hip.Context.from_torch(local_rank) for _ in range(D - 1)]
            q["fout"] = [(out_pts[d][lo:hi], out_status[d][lo:hi], out_mask[d][lo:hi]) for d in range(D)]
            q["jobs"] = [hip.TrackPnpBatch(c) for c in q["ctxs"]]
            q["pending"] = []
        groups.append(q)
    eff_level = groups[0]["pyrs"][0].max_level
    torch.cuda.synchronize()

    for q in groups:
        for f in (0, 1):
            q["pyrs"][f].build(q["frames"][f % R])
            q["pyr_done"][f].record(q["s_pyr"])
        q["klt_done"][2].record(q["s_klt"])      # slot 2 is free from the start
    state = {"t": 0}

    host = {"finish": 0.0, "begin": 0.0, "pyr": 0.0} if os.environ.get("VO_BENCH_HOST_PARTS") else None

    def finish(q):
        """Frame mode: wait for the group's chain in flight and take its poses (the one synchronisation of its step)."""
        if not q["pending"]:
            return
        d, slot, timed = q["pending"].pop(0)
        if host is not None:
            t_h = time.perf_counter()
        _, _, Rt, _, tracked, good, _, path = q["jobs"][d].finish()
        if host is not None:
            host["finish"] += time.perf_counter() - t_h
        if timed:
            q["ms"].append(q["ctxs"][d].profile_tracker_ms())
        lo, hi = q["lo"], q["hi"]
        last["Rt"][lo:hi], last["good"][lo:hi], last["tracked"][lo:hi], last["path"][lo:hi] = Rt, good, tracked, path
        last["slot"], last["d"] = slot, d
        last["slow"] += int((path >= 2).sum())
        last["frames"] += hi - lo

    def step(timed=False):
        k = state["t"]
        for q in groups:
            if frame_mode and len(q["pending"]) >= len(q["jobs"]):
                finish(q)                     # the oldest batch of this group; its context is needed again
            if host is not None:
                t_h = time.perf_counter()
            q["s_pyr"].wait_event(q["klt_done"][(k + 2) % 3])          # = step k-1, the last reader of the slot
            q["pyrs"][(k + 2) % 3].build(q["frames"][(k + 2) % R])
            q["pyr_done"][(k + 2) % 3].record(q["s_pyr"])
            q["s_klt"].wait_event(q["pyr_done"][(k + 1) % 3])
            if host is not None:
                host["pyr"] += time.perf_counter() - t_h
                t_h = time.perf_counter()
            if frame_mode:
                d = k % len(q["jobs"])
                q["ctxs"][d].profile_tracker(timed)
                q["jobs"][d].begin(q["pyrs"][k % 3], q["pyrs"][(k + 1) % 3], q["pts"], q["X"][(k + 1) % R], K_cam, pnp[0], pnp[1], pnp[2],
                                   refine_iters=args.refine_iters, out=q["fout"][d])
                q["pending"].append((d, (k + 1) % R, timed))
                if host is not None:
                    host["begin"] += time.perf_counter() - t_h
                while len(q["pending"]) > len(q["jobs"]) - 1 and len(q["jobs"]) > 1:
                    finish(q)                 # one synchronisation per step and group: the step before, with this step's
                                              # chain already queued behind it
            else:
                if timed:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    q["evs"].append(ev)
                    ev[0].record(q["s_klt"])
                hip.klt_track(q["ctx"], q["pyrs"][k % 3], q["pyrs"][(k + 1) % 3], q["pts"], out=q["out"])
                if timed:
                    ev[1].record(q["s_klt"])
            q["klt_done"][k % 3].record(q["s_klt"])
        state["t"] = k + 1

    def drain():
        if frame_mode:
            for q in groups:
                while q["pending"]:
                    finish(q)
        torch.cuda.synchronize()

    def fence():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the GPU's clocks take ~50 ms of load to settle: run untimed steps for a fixed wall time first,
    # so that short timed regions measure the same sustained state as long ones
    t_settle = time.perf_counter() + args.settle_ms * 1e-3
    while time.perf_counter() < t_settle:
        for _ in range(16):
            step()
        drain()
    for _ in range(args.warmup):
        step()
    EV_EVERY = 4
    fence()
    if frame_mode:
        last["slow"] = last["frames"] = 0
    if host is not None:
        host.update(finish=0.0, begin=0.0, pyr=0.0)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(timed=(k % EV_EVERY == 0))
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if host is not None:       # where the host thread spent the timed region (development aid, stderr)
        print({k: round(v / args.steps * 1e3, 4) for k, v in host.items()}, "ms per step; step", round(elapsed / args.steps * 1e3, 4),
              file=sys.stderr, flush=True)
    elapsed = streams.max_over_ranks(elapsed, dev, world)
    per_rank_ms = [e / args.steps * 1e3 for e in streams.gather_scalars(elapsed_local, dev, world)]

    # per-launch durations, per group size (groups may differ by one stream)
    if frame_mode:
        per_group = [float(np.mean(q["ms"])) for q in groups]
        n_timed = sum(len(q["ms"]) for q in groups)
    else:
        per_group = [float(np.mean([a.elapsed_time(b) for a, b in q["evs"]])) for q in groups]
        n_timed = sum(len(q["evs"]) for q in groups)
    launch_ms = float(np.mean(per_group))
    busy_ms = sum(m * args.steps for m in per_group)
    in_flight = busy_ms / (elapsed_local * 1e3)
    # The only data exchange: every stream's tracked positions and flags of the last step, gathered to
    # global stream order (S*world x 2000 x 2 float32 -- 16 KB per stream), outside the timed region;
    # in frame mode every stream's last pose as well.
    if frame_mode:
        out_pts, out_status = out_pts[last["d"]], out_status[last["d"]]
    tracks = streams.gather_per_stream(out_pts, owned, S * world, world)
    status = streams.gather_per_stream(out_status, owned, S * world, world)
    tracked_frac = float(status.float().mean().item())
    gather_bytes = tracks.numel() * 4 + status.numel()
    if frame_mode:
        poses = streams.gather_per_stream(torch.from_numpy(last["Rt"]).to(dev), owned, S * world, world)
        gather_bytes += poses.numel() * 8
        # the poses of the last step against the poses the landmarks were made with
        truth = poses_true[last["slot"]]
        found = last["good"] > 0
        rot_err = [float(np.degrees(np.arccos(np.clip((np.trace(last["Rt"][s, :, :3] @ truth[s, :, :3].T) - 1) / 2, -1, 1))))
                   for s in range(S) if found[s]]
        pose_check = {"streams_with_pose": int(found.sum()), "max_rotation_error_deg": max(rot_err) if rot_err else None,
                      "max_translation_error": float(max(np.linalg.norm(last["Rt"][s, :, 3] - truth[s, :, 3]) for s in range(S) if found[s]))
                      if found.any() else None,
                      "mean_inlier_fraction": float((last["good"][found] / np.maximum(last["tracked"][found], 1)).mean()) if found.any() else None,
                      "frames_finished_beyond_first_chunk": last["slow"], "frames_timed_local": last["frames"]}

    if rank == 0:
        frames_total = S * world * args.steps
        fps = frames_total / elapsed
        path_bytes, klt_bytes = algorithmic_bytes(eff_level, N_KP)
        if frame_mode:
            # + landmarks read (12 B), inlier mask written (1 B) per keypoint, one pose (96 B) per frame
            path_bytes += N_KP * 13 + 96
        per_launch = klt_bytes * (S / G) / (launch_ms * 1e-3) / 1e9
        achieved = per_launch * in_flight
        roof = {"bound": "hbm", "kernel": "klt_track3_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "launch_ms": launch_ms, "frames_per_launch": S / G, "launches_in_flight": in_flight,
                "achieved_per_launch": per_launch,
                "bytes_per_frame_kernel": klt_bytes, "bytes_per_frame_path": path_bytes,
                "timed_launches": n_timed, "path_achieved": path_bytes * fps / world / 1e9}
        # Counter-derived figures of the same build and workload (rocprofv3 --pmc passes, committed):
        # HBM bytes per launch, VALU issue share, LDS bank-conflict share.
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if rec.get("kernel") == roof["kernel"] and rec.get("frames_per_launch") == S / G:
                    roof["traffic"] = rec.get("hbm_bytes_per_launch")
                    for key in ("valu_issue", "lds_conflict_frac", "lds_busy_frac", "valu_insts_per_keypoint", "counters_from"):
                        if key in rec:
                            roof[key] = rec[key]
            except Exception:
                pass
        if frame_mode:
            # what the rest of the chain costs, from the committed rocprofv3 kernel statistics of this command
            ff = os.path.join(ROOT, "profiles", "frame_chain.json")
            if os.path.exists(ff):
                try:
                    roof["chain"] = json.load(open(ff))
                except Exception:
                    pass
        workload = (f"KITTI-shaped 1241x376 u8 frames, 2000 keypoints/frame, {eff_level + 1}-level pyramid "
                    f"(maxLevel={eff_level}), win 21, 10 iters, eps 1e-3, minEig 1e-2")
        res = {
            "metric": "tracked frames/s (KITTI 1241x376, 2k kp)", "value": fps, "unit": "frames/s",
            "keypoints_per_s": fps * N_KP, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "per_rank_ms_per_step": per_rank_ms, "higher_is_better": True,
            "scaling": "strong" if args.total_streams else "weak",
            "vs_baseline": None, "dtype": "u8 images, int32 fixed-point bilinear, f32 normal equations",
            "data": "synthetic",
            "config": {"workload": workload,
                       "streams_per_gpu": S, "launch_groups": G, "frames_per_launch": S / G, "ring_frames": R,
                       "settle_ms": args.settle_ms, "stream_priorities": {"tracker": prio[0], "pyramid": prio[1]},
                       "schedule": "pyramid right before its tracker, one HIP stream per group" if args.no_pipeline
                                   else "per group: pyramid of frame k+2 on a second HIP stream during tracker k",
                       "distinct_streams": (args.distinct if args.distinct and args.distinct < S else S) * world,
                       "tracked_fraction": tracked_frac,
                       "final_gather": {"tracks_shape": list(tracks.shape), "bytes": gather_bytes},
                       "parallelism": f"stream-parallel x{world}"},
            "roofline": roof,
        }
        if frame_mode:
            res["metric"] = "tracked frames/s with a pose per frame (KITTI 1241x376, 2k kp; tracker + status filter + P3P RANSAC + refit)"
            res["dtype"] = "u8 images, int32 fixed-point bilinear, f32 normal equations; f64 P3P / reprojection / refit"
            res["config"]["workload"] = (workload + f"; then per frame the status filter, cv.solvePnPRansac(P3P) restated (reprojection "
                                         f"{pnp[0]} px, confidence {pnp[1]}, {pnp[2]} iterations, adaptive stop) and the refit on the inliers; "
                                         f"{args.outliers:.0%} of the landmarks displaced")
            res["config"]["synchronisations_per_step"] = G
            res["config"]["batches_in_flight_per_group"] = max(1, args.depth)
            res["config"]["pose_check"] = pose_check
        if frame_extra is not None:
            res["frame_mode"] = frame_extra        # the same streams with a pose per frame, measured by `--mode frame`
        if elapsed * 1e3 < 50.0:
            res["short_timed_region"] = True       # steps x ms_per_step < 50 ms: one scheduler hiccup away from noise
        if world == 1 and not args.no_cpu_baseline:
            fm = (X_h, K_cam, pnp[0], pnp[1], pnp[2]) if frame_mode else None
            res["cpu_baseline"] = cpu_baseline(frames_h, pts_h, max_level=eff_level, cores=args.cpu_cores, frame_mode=fm)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
