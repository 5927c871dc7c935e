#!/usr/bin/env python3
"""Seed/shape sweep of every GPU operator against the CPU oracle, beyond the fixed cases of tests/.
Prints one JSON line per operator with the worst deviation seen; exits non-zero at the first breach
of the parity bars (bit-exact integer/byte/index work and float32 maps; LK positions 0.05 px with
>= 99.5 % status agreement)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    a = ap.parse_args()
    from oracle import oracle
    from visual_odometry_amd import hip, synth
    ctx = hip.Context(0)
    rng = np.random.default_rng(1234)
    worst = {"klt_px": 0.0, "klt_status": 1.0, "klt_points": 0}
    for seed in range(a.seeds):
        h, w = int(rng.integers(60, 400)), int(rng.integers(80, 700))
        img0, img1, _ = synth.frame_pair((h, w), seed=100 + seed, max_shift=float(rng.uniform(1, 8)))
        # pyramid
        pyr = hip.Pyramid(ctx, 1, h, w, 8).build(img0)
        for lvl in range(pyr.max_level + 1):
            assert (pyr.download(0, lvl) == oracle.pyr_level(img0, lvl)).all(), ("pyramid", seed, lvl)
        # LK, both kernels
        n = int(rng.integers(50, 1500))
        pts = synth.uniform_keypoints((h, w), n, seed=seed, margin=int(rng.integers(0, 12)))
        ro, rs, _ = oracle.klt(img0, img1, pts, max_level=8)
        for variant in ("1", "3"):
            os.environ["VO_KLT_VARIANT"] = variant
            p1 = hip.Pyramid(ctx, 1, h, w, 8).build(img1)
            out, st, _ = hip.klt_track(ctx, pyr, p1, pts)
            agree = float((st[0] == rs.ravel()).mean())
            both = (st[0] == 1) & (rs.ravel() == 1)
            d = float(np.abs(out[0][both] - ro[both]).max()) if both.any() else 0.0
            worst["klt_px"] = max(worst["klt_px"], d)
            worst["klt_status"] = min(worst["klt_status"], agree)
            worst["klt_points"] += n
            assert d <= 0.05 and agree >= 0.995, ("klt", seed, variant, d, agree)
        os.environ.pop("VO_KLT_VARIANT", None)
        # corner response, NMS/top-K, mask, sub-pixel refinement
        for mode in (hip.HARRIS, hip.MIN_EIGEN):
            k = float(rng.choice([0.04, 0.1]))
            R = hip.harris_response(ctx, img0[None], k, mode)[0]
            Ro = oracle.harris_response(img0, k, mode)
            assert (R.view(np.uint32) == Ro.view(np.uint32)).all(), ("response", seed, mode)
            radius, K = int(rng.integers(1, 12)), int(rng.integers(10, 3000))
            xy, sc, cnt = hip.detect_corners(ctx, img0[None], K, 0.01, radius, k, mode)
            oxy, osc, om = oracle.nms_topk(Ro, K, 0.01, radius)
            assert int(cnt[0]) == om and (xy[0, :om] == oxy[:om]).all() and (sc[0, :om] == osc[:om]).all(), ("corners", seed, mode, radius, K)
            assert (hip.harris_mask(ctx, img0, k, 0.01, mode) == oracle.harris_mask(img0, k, 0.01, mode)).all(), ("mask", seed)
    print(json.dumps({"op": "pyramid/klt/corners/mask", "seeds": a.seeds, **worst}), flush=True)
    # sub-pixel refinement, own loop so that both sides get identical arguments
    for seed in range(a.seeds):
        h, w = int(rng.integers(40, 300)), int(rng.integers(40, 300))
        img = synth.frame_pair((h, w), seed=300 + seed)[0]
        pts = synth.uniform_keypoints((h, w), int(rng.integers(1, 800)), seed=seed, margin=0)
        win, iters = (int(rng.integers(1, 7)), int(rng.integers(1, 7))), int(rng.integers(1, 60))
        got = hip.corner_subpix(ctx, img, pts, win, iters, 0.001)
        ref = oracle.corner_subpix(img, pts, win, iters, 0.001)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all(), ("subpix", seed, win, iters)
    print(json.dumps({"op": "corner_subpix", "seeds": a.seeds, "bit_exact": True}), flush=True)
    # scoring and minimal-solver pipelines
    for seed in range(a.seeds):
        n, H = int(rng.integers(20, 4000)), int(rng.integers(1, 600))
        sc = synth.two_view_scene(n=n, seed=seed, inlier_frac=float(rng.uniform(0.3, 1.0)), noise=float(rng.uniform(0, 1)))
        Fh, Rt = synth.fundamental_hypotheses(sc, H, seed=seed), synth.pose_hypotheses(sc, H, seed=seed)
        thr = float(rng.uniform(0.05, 2.0))
        c, m = hip.score_fundamental(ctx, sc["m1"], sc["m2"], Fh, thr)
        oc, om = oracle.score_fundamental(sc["m1"], sc["m2"], Fh, thr)
        assert (c == oc).all() and (m == om).all(), ("score_f", seed)
        c, m = hip.score_pnp(ctx, sc["X"], sc["m2"], sc["K"], Rt, thr)
        oc, om = oracle.score_pnp(sc["X"], sc["m2"], sc["K"], Rt, thr)
        assert (c == oc).all() and (m == om).all(), ("score_pnp", seed)
    print(json.dumps({"op": "score_fundamental/score_pnp", "seeds": a.seeds, "bit_exact": True}), flush=True)
    # BA residual / normal equations
    worst_ba = 0.0
    for seed in range(a.seeds):
        F, P = int(rng.integers(2, 24)), int(rng.integers(5, 3000))
        pb = synth.ba_problem(F=F, P=P, visibility=float(rng.uniform(0.3, 1.0)), seed=seed)
        start = int(np.count_nonzero(pb["ci"] < int(F / 4) + 1))
        got = hip.ba_normal_eq(ctx, pb["x"], F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"], start)
        ref = oracle.ba_normal_eq(pb["x"], F, P, pb["ci"], pb["pi"], pb["obs"], pb["K"], start)
        for key in ("r", "cost", "Hcc", "Hpp", "Hcp", "g"):
            g, r = np.asarray(got[key], np.float64), np.asarray(ref[key], np.float64)
            scale = max(np.abs(r).max(), 1e-300)
            worst_ba = max(worst_ba, float(np.abs(g - r).max() / scale))
        assert worst_ba < 1e-9, ("ba", seed, worst_ba)
    print(json.dumps({"op": "ba_normal_eq", "seeds": a.seeds, "max_rel_dev": worst_ba}), flush=True)
    # descriptor matching, both kernels
    for seed in range(a.seeds):
        n1, n2 = int(rng.integers(1, 2500)), int(rng.integers(2, 2500))
        d1, d2, _ = synth.sift_like_descriptors(n1, n2, shared=float(rng.uniform(0, 1)), seed=seed)
        for scale in (np.float32(1), np.float32(rng.uniform(0.001, 3.0))):
            x, y = d1 * scale, d2 * scale
            ratio = float(rng.uniform(0.5, 1.0))
            gi, gd, gk = hip.knn_match2(ctx, x, y, ratio)
            ri, rd, rk = oracle.knn_match2(x, y, ratio)
            assert (gi == ri).all() and (gd.view(np.uint32) == rd.view(np.uint32)).all() and (gk == rk).all(), ("match", seed, n1, n2)
    print(json.dumps({"op": "knn_match2", "seeds": a.seeds, "bit_exact": True}), flush=True)
    # connected components in both numberings, and the fused FeatureExtractor.harris call
    labels_seen = 0
    for seed in range(a.seeds):
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 500))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            mask = (rng.random((h, w)) < rng.uniform(0.02, 0.9)).astype(np.uint8) * int(rng.integers(1, 256))
        elif kind == 1:
            mask = np.zeros((h, w), np.uint8); mask[::int(rng.integers(1, 4)), ::int(rng.integers(1, 4))] = 255
        elif kind == 2 and h >= 3 and w >= 3:
            mask = oracle.harris_mask(synth.frame_pair((max(h, 24), max(w, 24)), seed=500 + seed)[0], 0.1, float(rng.uniform(0.001, 0.2)))
        else:
            mask = (np.add.outer(np.arange(h), np.arange(w)) % int(rng.integers(2, 9)) == 0).astype(np.uint8)      # diagonals
        for order in (0, 1):
            n, lab, cen, ar = hip.connected_components(ctx, mask, order)
            rn, rl, rc, ra = oracle.connected_components(mask, order)
            assert n == rn and (lab == rl).all() and (ar == ra).all() and (cen.view(np.uint64) == rc.view(np.uint64)).all(), ("cc", seed, order)
            labels_seen += n
        hh, ww = int(rng.integers(8, 260)), int(rng.integers(8, 400))
        img = synth.frame_pair((max(hh, 24), max(ww, 24)), seed=600 + seed)[0][:hh, :ww].copy()
        order = int(rng.integers(0, 2))
        got, ref = hip.harris_corners(ctx, img, order=order), oracle.harris_corners(img, order=order)
        assert got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all(), ("harris_corners", seed)
    print(json.dumps({"op": "connected_components/harris_corners", "seeds": a.seeds, "labels": labels_seen, "bit_exact": True}), flush=True)
    # SIFT: keypoints (six fields) and descriptors
    kp_seen = 0
    for seed in range(a.seeds):
        h, w = int(rng.integers(24, 260)), int(rng.integers(24, 400))
        img = synth.frame_pair((h, w), seed=700 + seed, noise=float(rng.uniform(0.2, 4.0)))[0]
        args = dict(n_octave_layers=int(rng.integers(2, 6)), contrast_threshold=float(rng.uniform(0.01, 0.08)),
                    edge_threshold=float(rng.uniform(4, 20)), sigma=float(rng.uniform(1.0, 2.2)))
        rk, rd = oracle.sift(img, **args)
        kp, des = hip.sift_detect_compute(ctx, img, **args)
        assert len(kp) == len(rk), ("sift count", seed, len(kp), len(rk))
        for f in ("x", "y", "size", "angle", "response"):
            assert (kp[f].view(np.uint32) == rk[f].view(np.uint32)).all(), ("sift", seed, f)
        assert (kp["octave"] == rk["octave"]).all() and (des == rd).all(), ("sift descriptors", seed)
        kp_seen += len(kp)
    print(json.dumps({"op": "sift_detect_compute", "seeds": a.seeds, "keypoints": kp_seen, "bit_exact": True}), flush=True)
    # two-view cheirality vote
    from visual_odometry_amd.vo_pipeline.bootstrap import BootstrapInitializer
    from oracle import solvers_np
    worst_cloud = 0.0
    for seed in range(min(a.seeds, 12)):
        n = int(rng.integers(1, 400))
        sc = synth.two_view_scene(n=n, seed=seed, inlier_frac=1.0, noise=float(rng.uniform(0, 0.3)))
        Kf = sc["K"].astype(np.float32)
        motions = BootstrapInitializer.motion_hypotheses(Kf.T @ sc["F"] @ Kf)
        M1, M2 = Kf @ np.eye(3, 4, dtype=np.float32), np.stack([Kf @ m for m in motions])
        p1 = np.hstack([sc["m1"], np.ones((n, 1), np.float32)]).astype(np.float32)
        p2 = np.hstack([sc["m2"], np.ones((n, 1), np.float32)]).astype(np.float32)
        clouds, votes = hip.cheirality_vote(ctx, p1, p2, M1, M2, motions)
        rc, rv = solvers_np.cheirality_votes(p1, p2, M1, M2, motions)
        assert (votes == rv).all(), ("cheirality votes", seed)
        rel = np.abs(clouds - rc)[..., :3] / np.maximum(np.abs(rc[..., :3]), 1.0)
        worst_cloud = max(worst_cloud, float(np.median(rel)))
        assert np.median(rel) < 1e-5, ("cheirality clouds", seed)
    print(json.dumps({"op": "cheirality_vote", "seeds": min(a.seeds, 12), "votes_exact": True, "worst_median_rel_cloud_dev": worst_cloud}), flush=True)


if __name__ == "__main__":
    main()
