#!/usr/bin/env python3
"""Headline benchmark: tracked frames/s of the pyramid + pyramidal-LK hot path on
synthetic KITTI-shaped frames (1241x376, 2000 keypoints, 4-level pyramid).

A step = every local stream advances one frame: the new frame (already resident
in HBM) gets its pyramid built and the stream's 2000 keypoints are tracked from
the previous frame's pyramid (kept from the step before) into it.  Streams are
independent, so N GPUs run N x streams-per-GPU streams with no per-frame
communication (weak scaling); ranks only meet at the barriers around the timed
region and in a final gather of per-stream tracked counts.

Prints ONE JSON line on rank 0.  roofline.achieved prices the tracker kernel's own
algorithmic bytes against its launch duration, taken with HIP events on the launch
stream around every fourth tracker launch of the timed region (an event pair costs a
few microseconds of idle queue, so not every step carries one).
"""
import argparse
import json
import os
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


def cpu_baseline(frames, pts, max_level=MAX_LEVEL, budget_s=12.0):
    """Single-thread CPU oracle on the same workload (bounded sample)."""
    from oracle import oracle
    ring, n_streams = frames.shape[:2]
    done, t0 = 0, time.perf_counter()
    while True:
        s, r = done % n_streams, (done // n_streams) % ring
        oracle.klt(frames[r, s], frames[(r + 1) % ring, s], pts[s], max_level=max_level)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s and done >= 3:
            break
    return {"value": done / el, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} frame pairs of the same streams ({N_KP} keypoints each, pyramids of both frames "
                      f"rebuilt per pair as the reference's cv2 call does), {el:.1f} s"}


def main():
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
    ap.add_argument("--ring", type=int, default=3)
    ap.add_argument("--max-level", type=int, default=MAX_LEVEL,
                    help="requested last pyramid level (default 3 = the headline 4-level configuration; the reference "
                         "asks for 8, which the level-size rule clamps to 4 on 1241x376 frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="build each frame's pyramid on the tracker's stream, right before it is tracked")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal aid: every rank uses GPU 0 (one-GPU box); never used for reported numbers")
    ap.add_argument("--distinct", type=int, default=0,
                    help="development aid: synthesise only this many distinct streams and repeat them (0 = all distinct)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from visual_odometry_amd import hip, streams

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
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
    # Two HIP streams per GPU.  The tracker runs on a high-priority stream; the pyramid of the frame
    # after next is built on a normal-priority one, so its workgroups are dispatched where the
    # tracker leaves room -- in practice while the tracker's last workgroups drain.  Three pyramid
    # slots: step k tracks slot k%3 -> (k+1)%3 while frame k+2 is built into slot (k+2)%3, which
    # the tracker of step k-1 was the last to read.  Every step still builds one pyramid and
    # tracks one frame per stream.
    if args.no_pipeline:
        s_klt = s_pyr = torch.cuda.current_stream(dev)
    else:
        s_klt = torch.cuda.Stream(device=dev, priority=-1)
        s_pyr = torch.cuda.Stream(device=dev, priority=0)
    with torch.cuda.stream(s_klt):
        ctx = hip.Context.from_torch(local_rank)
    with torch.cuda.stream(s_pyr):
        ctx_pyr = hip.Context.from_torch(local_rank)
    pyrs = [hip.Pyramid(ctx_pyr, S, H, W, args.max_level) for _ in range(3)]
    eff_level = pyrs[0].max_level
    out = (torch.empty_like(pts), torch.empty((S, N_KP), dtype=torch.uint8, device=dev),
           torch.empty((S, N_KP), dtype=torch.float32, device=dev))
    pyr_done = [torch.cuda.Event() for _ in range(3)]
    klt_done = [torch.cuda.Event() for _ in range(3)]
    torch.cuda.synchronize()

    state = {"t": 0}
    for f in (0, 1):
        pyrs[f].build(frames[f % R])
        pyr_done[f].record(s_pyr)
    klt_done[2].record(s_klt)      # slot 2 is free from the start

    def step(ev=None):
        k = state["t"]
        s_pyr.wait_event(klt_done[(k + 2) % 3])          # = step k-1, the last reader of the slot
        pyrs[(k + 2) % 3].build(frames[(k + 2) % R])
        pyr_done[(k + 2) % 3].record(s_pyr)
        s_klt.wait_event(pyr_done[(k + 1) % 3])
        if ev:
            ev[0].record(s_klt)
        hip.klt_track(ctx, pyrs[k % 3], pyrs[(k + 1) % 3], pts, out=out)
        if ev:
            ev[1].record(s_klt)
        klt_done[k % 3].record(s_klt)
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
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range((args.steps + EV_EVERY - 1) // EV_EVERY)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k // EV_EVERY] if k % EV_EVERY == 0 else None)
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = streams.max_over_ranks(elapsed, dev, world)

    klt_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    tracked = out[1].sum(dim=1).to(torch.int32)           # per-stream tracked keypoints of the last step
    tracked = streams.gather_per_stream(tracked, owned, S * world, world)   # the only data exchange: a final gather
    tracked_frac = float(tracked.float().mean().item()) / N_KP

    if rank == 0:
        frames_total = S * world * args.steps
        fps = frames_total / elapsed
        path_bytes, klt_bytes = algorithmic_bytes(eff_level, N_KP)
        achieved = klt_bytes * S / (klt_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if rec.get("streams_per_gpu") == S and rec.get("kernel") == "klt_track3_kernel":
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "tracked frames/s (KITTI 1241x376, 2k kp)", "value": fps, "unit": "frames/s",
            "keypoints_per_s": fps * N_KP, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.total_streams else "weak",
            "vs_baseline": None, "dtype": "u8 images, int32 fixed-point bilinear, f32 normal equations",
            "data": "synthetic",
            "config": {"workload": f"KITTI-shaped 1241x376 u8 frames, 2000 keypoints/frame, {eff_level + 1}-level pyramid "
                                   f"(maxLevel={eff_level}), win 21, 10 iters, eps 1e-3, minEig 1e-2",
                       "streams_per_gpu": S, "frames_per_launch": S, "ring_frames": R, "settle_ms": args.settle_ms,
                       "schedule": "one stream" if args.no_pipeline else "pyramid of frame k+2 on a second HIP stream during tracker k",
                       "distinct_streams": (args.distinct if args.distinct and args.distinct < S else S) * world,
                       "tracked_fraction": tracked_frac, "parallelism": f"stream-parallel x{world}"},
            "roofline": {"bound": "hbm", "kernel": "klt_track3_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": klt_ms, "bytes_per_frame_kernel": klt_bytes, "bytes_per_frame_path": path_bytes,
                         "timed_launches": len(evs), "path_achieved": path_bytes * fps / world / 1e9},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(frames_h, pts_h, max_level=eff_level)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
