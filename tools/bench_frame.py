#!/usr/bin/env python3
"""Timing of the single-frame calls of a tracked frame on one scene: vo_track_pnp (one call) against
vo_klt_track + filter + vo_solve_pnp_ransac, host arrays in and out, and the refit alone."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from visual_odometry_amd import hip, synth
    from tests.test_frame_gpu import _scene, K
    ctx = hip.Context(0)
    img0, img1, pts, X, R, t = _scene(shape=synth.PARKING_SHAPE, n=1500, seed=3)
    h, w = img0.shape
    p0 = hip.Pyramid(ctx, 1, h, w, 3).build(img0)
    p1 = hip.Pyramid(ctx, 1, h, w, 3)

    def fused():
        return hip.track_pnp(ctx, p0, p1, pts, X, K, 1.0, 0.999, 10000, next_image=img1)

    def separate():
        p1.build(img1)
        out, st, _ = hip.klt_track(ctx, p0, p1, pts)
        keep = st[0] == 1
        return hip.solve_pnp_ransac(ctx, X[keep], out[0][keep], K, 1.0, 0.999, 10000)

    out, st, _ = hip.klt_track(ctx, p0, p1.build(img1), pts)
    keep = st[0] == 1
    Xk, xk = X[keep], out[0][keep]
    Rt, mask, good, _ = hip.solve_pnp_ransac(ctx, Xk, xk, K, 1.0, 0.999, 10000)
    from scipy.spatial.transform import Rotation
    rv0 = Rotation.from_matrix(Rt[:, :3]).as_rotvec() + 1e-3

    def refit():
        return hip.refine_pose(ctx, Xk, xk, K, rv0, Rt[:, 3], mask=mask, max_iters=10)

    res = {"lib": os.environ.get("VO_HIP_LIB", "default"), "points": int(len(pts)), "tracked": int(keep.sum()), "inliers": int(good)}
    for name, fn in (("track_pnp_ms", fused), ("klt_then_pnp_ms", separate), ("refine_pose_ms", refit)):
        for _ in range(20):
            fn()
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        res[name] = float(np.median(ts) * 1e3)
    res["path"] = hip.track_pnp_path(ctx) if hasattr(hip, "track_pnp_path") else None
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
