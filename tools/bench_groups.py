#!/usr/bin/env python3
"""Experiment beside bench.py: the 64 streams of one GPU split into G groups, each with its own pair
of HIP streams (tracker / pyramid), so that one group's tracker launch fills the drain of another's.
Reports frames/s only: launches of different groups overlap, so a per-launch duration no longer
describes one kernel (which is why bench.py keeps one group)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--distinct", type=int, default=16)
    a = ap.parse_args()
    import torch
    from visual_odometry_amd import hip
    dev = torch.device("cuda", 0)
    S, G, R = a.streams, a.groups, 3
    assert S % G == 0
    Sg = S // G
    f0, p0, _ = bench.make_streams(list(range(a.distinct)), R)
    rep = (S + a.distinct - 1) // a.distinct
    frames = torch.from_numpy(np.tile(f0, (1, rep, 1, 1))[:, :S].copy()).to(dev)
    pts = torch.from_numpy(np.tile(p0, (rep, 1, 1))[:S].copy()).to(dev)
    groups = []
    for g in range(G):
        s_klt, s_pyr = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=0)
        with torch.cuda.stream(s_klt):
            ctx = hip.Context.from_torch(0)
        with torch.cuda.stream(s_pyr):
            ctx_pyr = hip.Context.from_torch(0)
        sl = slice(g * Sg, (g + 1) * Sg)
        gp = pts[sl].contiguous()
        groups.append(dict(s_klt=s_klt, s_pyr=s_pyr, ctx=ctx, fr=[frames[r, sl] for r in range(R)], pts=gp,
                           pyrs=[hip.Pyramid(ctx_pyr, Sg, bench.H, bench.W, bench.MAX_LEVEL) for _ in range(3)],
                           out=(torch.empty_like(gp), torch.empty((Sg, bench.N_KP), dtype=torch.uint8, device=dev),
                                torch.empty((Sg, bench.N_KP), dtype=torch.float32, device=dev)),
                           pyr_done=[torch.cuda.Event() for _ in range(3)], klt_done=[torch.cuda.Event() for _ in range(3)]))
    torch.cuda.synchronize()
    for q in groups:
        for f in (0, 1):
            q["pyrs"][f].build(q["fr"][f % R])
            q["pyr_done"][f].record(q["s_pyr"])
        q["klt_done"][2].record(q["s_klt"])
    state = {"t": 0}

    def step():
        k = state["t"]
        for q in groups:
            q["s_pyr"].wait_event(q["klt_done"][(k + 2) % 3])
            q["pyrs"][(k + 2) % 3].build(q["fr"][(k + 2) % R])
            q["pyr_done"][(k + 2) % 3].record(q["s_pyr"])
            q["s_klt"].wait_event(q["pyr_done"][(k + 1) % 3])
            hip.klt_track(q["ctx"], q["pyrs"][k % 3], q["pyrs"][(k + 1) % 3], q["pts"], out=q["out"])
            q["klt_done"][k % 3].record(q["s_klt"])
        state["t"] = k + 1

    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        for _ in range(16):
            step()
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tracked = float(sum(q["out"][1].float().mean().item() for q in groups) / G)
    print(json.dumps({"groups": G, "streams": S, "frames_per_s": S * a.steps / el, "ms_per_step": el / a.steps * 1e3,
                      "tracked_fraction": tracked}))


if __name__ == "__main__":
    main()
