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


def make_streams(stream_ids, ring, base_seed=2021):
    """Per stream a ring of `ring` frames (texture seen through successive small
    warps) and a seeded keypoint set; a stream's data depends on its global id only."""
    from scipy import ndimage
    from visual_odometry_amd import synth
    from visual_odometry_amd import streams
    n_streams = len(stream_ids)
    frames = np.empty((ring, n_streams, H, W), np.uint8)
    pts = np.empty((n_streams, N_KP, 2), np.float32)
    margin = 40
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    for s, sid in enumerate(stream_ids):
        seed = streams.stream_seed(base_seed, sid)
        rng = np.random.default_rng(seed)
        tex = synth.texture((H, W), rng, margin)
        off = np.zeros(2)
        for r in range(ring):
            if r:
                off = off + rng.uniform(-6, 6, 2)
                off = np.clip(off, -30, 30)
            aff = np.eye(2) + (rng.uniform(-0.005, 0.005, (2, 2)) if r else 0)
            cx, cy = (W - 1) / 2, (H - 1) / 2
            sx = aff[0, 0] * (xs - cx) + aff[0, 1] * (ys - cy) + cx - off[0] + margin
            sy = aff[1, 0] * (xs - cx) + aff[1, 1] * (ys - cy) + cy - off[1] + margin
            img = ndimage.map_coordinates(tex, [sy, sx], order=1, mode="nearest")
            img = img + rng.normal(0, 1.0, img.shape)
            frames[r, s] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        pts[s] = synth.uniform_keypoints((H, W), N_KP, seed=seed, margin=11)
    return frames, pts


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


def cpu_baseline(frames, pts, max_level=MAX_LEVEL, budget_s=10.0, cores=0):
    """CPU oracle on the same workload (bounded sample): one thread, then `cores` threads
    (the C oracle is called through ctypes, which releases the interpreter lock)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    ring, n_streams = frames.shape[:2]

    def pair(i):
        s, r = i % n_streams, (i // n_streams) % ring
        oracle.klt(frames[r, s], frames[(r + 1) % ring, s], pts[s], max_level=max_level)

    done, t0 = 0, time.perf_counter()
    while True:
        pair(done)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s and done >= 3:
            break
    res = {"value": done / el, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} frame pairs of the same streams ({N_KP} keypoints each, pyramids of both frames "
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
    elapsed = streams.max_over_ranks(time.perf_counter() - t0 + 1e-3 * (rank + 1), dev, world)
    tracks = streams.gather_per_stream(local, owned, S * world, world)
    ok = bool((tracks[:, 0, 0] == torch.tensor([float(streams.stream_seed(2021, s)) for s in range(S * world)])).all())
    if rank == 0:
        print(json.dumps({"metric": "tracked frames/s (KITTI 1241x376, 2k kp)", "value": None, "unit": "frames/s",
                          "dry_run": True, "n_gpus": world, "streams": S * world, "gather_ok": ok,
                          "gathered_shape": list(tracks.shape), "max_elapsed_s": elapsed}), flush=True)
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
    owned = streams.shard_streams(S * world, rank, world)      # weak scaling: S streams per GPU
    if args.distinct and args.distinct < S:
        f0, p0 = make_streams(owned[:args.distinct], R)
        rep = (S + args.distinct - 1) // args.distinct
        frames_h = np.tile(f0, (1, rep, 1, 1))[:, :S].copy()
        pts_h = np.tile(p0, (rep, 1, 1))[:S].copy()
    else:
        frames_h, pts_h = make_streams(owned, R)
    frames = torch.from_numpy(frames_h).to(dev)          # (R, S, H, W) resident in HBM
    pts = torch.from_numpy(pts_h).to(dev)
    out_pts = torch.empty_like(pts)
    out_status = torch.empty((S, N_KP), dtype=torch.uint8, device=dev)
    out_err = torch.empty((S, N_KP), dtype=torch.float32, device=dev)

    # Launch groups.  A group owns local streams lo..hi, two HIP streams and three pyramid slots.  Its
    # tracker runs on a high-priority HIP stream; the pyramid of the frame after next is built on a
    # normal-priority one, so its workgroups are dispatched where the trackers leave room.  Step k of a
    # group tracks slot k%3 -> (k+1)%3 while frame k+2 is built into slot (k+2)%3, which the tracker of
    # step k-1 was the last to read.  Every step still builds one pyramid and tracks one frame per stream.
    bounds = [S * g // G for g in range(G + 1)]
    groups = []
    for g in range(G):
        lo, hi = bounds[g], bounds[g + 1]
        if args.no_pipeline:
            s_klt = s_pyr = torch.cuda.Stream(device=dev, priority=-1) if G > 1 else torch.cuda.current_stream(dev)
        else:
            s_klt = torch.cuda.Stream(device=dev, priority=-1)
            s_pyr = torch.cuda.Stream(device=dev, priority=0)
        with torch.cuda.stream(s_klt):
            ctx = hip.Context.from_torch(local_rank)
        with torch.cuda.stream(s_pyr):
            ctx_pyr = hip.Context.from_torch(local_rank)
        groups.append({
            "n": hi - lo, "s_klt": s_klt, "s_pyr": s_pyr, "ctx": ctx,
            "frames": [frames[r, lo:hi] for r in range(R)], "pts": pts[lo:hi],
            "out": (out_pts[lo:hi], out_status[lo:hi], out_err[lo:hi]),
            "pyrs": [hip.Pyramid(ctx_pyr, hi - lo, H, W, args.max_level) for _ in range(3)],
            "pyr_done": [torch.cuda.Event() for _ in range(3)], "klt_done": [torch.cuda.Event() for _ in range(3)],
            "evs": []})
    eff_level = groups[0]["pyrs"][0].max_level
    torch.cuda.synchronize()

    for q in groups:
        for f in (0, 1):
            q["pyrs"][f].build(q["frames"][f % R])
            q["pyr_done"][f].record(q["s_pyr"])
        q["klt_done"][2].record(q["s_klt"])      # slot 2 is free from the start
    state = {"t": 0}

    def step(timed=False):
        k = state["t"]
        for q in groups:
            q["s_pyr"].wait_event(q["klt_done"][(k + 2) % 3])          # = step k-1, the last reader of the slot
            q["pyrs"][(k + 2) % 3].build(q["frames"][(k + 2) % R])
            q["pyr_done"][(k + 2) % 3].record(q["s_pyr"])
            q["s_klt"].wait_event(q["pyr_done"][(k + 1) % 3])
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                q["evs"].append(ev)
                ev[0].record(q["s_klt"])
            hip.klt_track(q["ctx"], q["pyrs"][k % 3], q["pyrs"][(k + 1) % 3], q["pts"], out=q["out"])
            if timed:
                ev[1].record(q["s_klt"])
            q["klt_done"][k % 3].record(q["s_klt"])
        state["t"] = k + 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the GPU's clocks take ~50 ms of load to settle: run untimed steps for a fixed wall time first,
    # so that short timed regions measure the same sustained state as long ones
    t_settle = time.perf_counter() + args.settle_ms * 1e-3
    while time.perf_counter() < t_settle:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    EV_EVERY = 4
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(timed=(k % EV_EVERY == 0))
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    elapsed = streams.max_over_ranks(elapsed, dev, world)

    # per-launch durations, per group size (groups may differ by one stream)
    launch_ms = float(np.mean([a.elapsed_time(b) for q in groups for a, b in q["evs"]]))
    busy_ms = sum(float(np.mean([a.elapsed_time(b) for a, b in q["evs"]])) * args.steps for q in groups)
    in_flight = busy_ms / (elapsed_local * 1e3)
    # The only data exchange: every stream's tracked positions and flags of the last step, gathered to
    # global stream order (S*world x 2000 x 2 float32 -- 16 KB per stream), outside the timed region.
    tracks = streams.gather_per_stream(out_pts, owned, S * world, world)
    status = streams.gather_per_stream(out_status, owned, S * world, world)
    tracked_frac = float(status.float().mean().item())

    if rank == 0:
        frames_total = S * world * args.steps
        fps = frames_total / elapsed
        path_bytes, klt_bytes = algorithmic_bytes(eff_level, N_KP)
        per_launch = klt_bytes * (S / G) / (launch_ms * 1e-3) / 1e9
        achieved = per_launch * in_flight
        roof = {"bound": "hbm", "kernel": "klt_track3_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "launch_ms": launch_ms, "frames_per_launch": S / G, "launches_in_flight": in_flight,
                "achieved_per_launch": per_launch,
                "bytes_per_frame_kernel": klt_bytes, "bytes_per_frame_path": path_bytes,
                "timed_launches": sum(len(q["evs"]) for q in groups), "path_achieved": path_bytes * fps / world / 1e9}
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
        res = {
            "metric": "tracked frames/s (KITTI 1241x376, 2k kp)", "value": fps, "unit": "frames/s",
            "keypoints_per_s": fps * N_KP, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.total_streams else "weak",
            "vs_baseline": None, "dtype": "u8 images, int32 fixed-point bilinear, f32 normal equations",
            "data": "synthetic",
            "config": {"workload": f"KITTI-shaped 1241x376 u8 frames, 2000 keypoints/frame, {eff_level + 1}-level pyramid "
                                   f"(maxLevel={eff_level}), win 21, 10 iters, eps 1e-3, minEig 1e-2",
                       "streams_per_gpu": S, "launch_groups": G, "frames_per_launch": S / G, "ring_frames": R,
                       "settle_ms": args.settle_ms,
                       "schedule": "pyramid right before its tracker, one HIP stream per group" if args.no_pipeline
                                   else "per group: pyramid of frame k+2 on a second HIP stream during tracker k",
                       "distinct_streams": (args.distinct if args.distinct and args.distinct < S else S) * world,
                       "tracked_fraction": tracked_frac,
                       "final_gather": {"tracks_shape": list(tracks.shape), "bytes": tracks.numel() * 4 + status.numel()},
                       "parallelism": f"stream-parallel x{world}"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(frames_h, pts_h, max_level=eff_level, cores=args.cpu_cores)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
