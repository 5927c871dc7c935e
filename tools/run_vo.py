#!/usr/bin/env python3
"""End-to-end run of the ContinuousVO loop on the synthetic 640x480 parking-shaped sequence:
frames/s of the whole loop (host numpy in and out of every operator, as the reference calls
them) and the error of the recovered trajectory against the scene's ground truth."""
import argparse
import gc
import json
import logging
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from visual_odometry_amd import params                                      # noqa: E402
from visual_odometry_amd.utils import loadData                              # noqa: E402
from visual_odometry_amd.utils.matrix import hom_inv                        # noqa: E402
from visual_odometry_amd.vo_pipeline.continuousVO import ContinuousVO      # noqa: E402


def trajectory_error(vo, gt_poses, first):
    """Camera centres of frames >= first in the frame-0 camera system, estimated vs ground truth,
    after fitting the one free scale by least squares over the path.  -> (rms position error / path length,
    max rotation error in degrees)."""
    est = {fs.idx: np.asarray(fs.pose, np.float64) for fs in vo.frame_queue.queue}
    idx = sorted(i for i in est if i >= first)
    c_est = np.array([hom_inv(est[i])[:3, 3] for i in idx])
    c_gt = np.array([gt_poses[i][:3, 3] for i in idx])
    scale = float((c_gt * c_est).sum() / (c_est * c_est).sum())
    err = np.linalg.norm(c_est * scale - c_gt, axis=1)
    rot = []
    for i in idx:
        dR = est[i][:3, :3] @ gt_poses[i][:3, :3]            # world->camera times camera->world
        rot.append(np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))))
    return float(np.sqrt(np.mean(err ** 2)) / np.linalg.norm(c_gt[-1])), float(max(rot)), scale


from oracle.pnp_ransac import cpu_pnp_ransac                               # noqa: E402


def cpu_frame_baseline(calls, K):
    """The recorded KLT and PnP calls of tracked frames replayed on one CPU core through the oracle
    operators; also how far the CPU pose lands from the pose the GPU call returned."""
    from oracle import oracle
    t_klt = t_pnp = 0.0
    rot_diff, trans_diff, iters = [], [], []
    for c in calls:
        t0 = time.perf_counter()
        oracle.klt(c["img0"], c["img1"], c["pts"], win=params.KLT_RADIUS, max_level=params.KLT_NUM_PYRAMIDS,
                   max_iters=params.KLT_N_ITERS, eps=0.001, min_eig=params.KLT_MIN_EIGEN_THRESHOLD)
        t1 = time.perf_counter()
        R, t, inl, it = cpu_pnp_ransac(c["X"], c["x"], K, params.PNP_RANSAC_REPROJ_THRESHOLD, params.PNP_RANSAC_CONFIDENCE,
                                       params.PNP_RANSAC_MAX_ITERS)
        t2 = time.perf_counter()
        t_klt += t1 - t0
        t_pnp += t2 - t1
        iters.append(it)
        if R is not None:
            dR = R @ c["M"][:3, :3].T
            rot_diff.append(np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))))
            trans_diff.append(np.linalg.norm(t - c["M"][:3, 3]) / max(np.linalg.norm(c["M"][:3, 3]), 1e-9))
    m = len(calls)
    return {"sample": f"{m} tracked frames of this run (their recorded KLT and PnP inputs), 1 core: C oracle LK with the "
                      "reference's arguments + OpenCV-style sequential P3P RANSAC (numpy solver, C scoring, adaptive "
                      "iteration count) + least-squares refit", "cores": 1, "kind": "port",
            "ms_per_tracked_frame": (t_klt + t_pnp) / m * 1e3, "klt_ms": t_klt / m * 1e3, "pnp_ms": t_pnp / m * 1e3,
            "frames_per_s": m / (t_klt + t_pnp), "ransac_iterations_mean": float(np.mean(iters)),
            "max_rotation_diff_deg_vs_gpu_pose": float(max(rot_diff)) if rot_diff else None,
            "max_rel_translation_diff_vs_gpu_pose": float(max(trans_diff)) if trans_diff else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-baseline", type=int, default=0,
                    help="replay the KLT and PnP calls of this many tracked frames on the CPU oracle after the run")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--dataset-params", default="PARKING")
    ap.add_argument("--bootstrap", default="sift", choices=["sift", "klt"],
                    help="correspondences of the two-view bootstraps: SIFT + matching (the reference's route) or corners + KLT")
    ap.add_argument("--separate-calls", action="store_true", help="KLT and PnP as two calls per tracked frame, as the reference makes them")
    ap.add_argument("--no-overlap", action="store_true", help="at a keyframe run the BA window and the re-bootstrap one after the other, as the reference does")
    ap.add_argument("--profile", action="store_true", help="median time of each part of a tracked frame's step")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    if a.v:
        logging.basicConfig(level=logging.INFO, format="%(message)s")
    params.use_dataset(a.dataset_params)
    data = loadData.synthetic_parking(a.frames, a.seed)
    frames = list(data.frames)                   # rendering is not part of the timed loop
    data.frames = iter(frames)
    vo = ContinuousVO(data, rescale_bootstrap=True, bootstrap_correspondences=a.bootstrap,
                      fused_step=not a.separate_calls, overlap_keyframe=not a.no_overlap)   # one world scale across keyframes, so the path compares with ground truth
    calls = []
    if a.cpu_baseline:
        from visual_odometry_amd.vo_pipeline.poseEstimation import PoseEstimation
        klt0, pnp0 = PoseEstimation.KLT, PoseEstimation.PnP
        last = {}

        def klt_rec(img0, img1, pts, **kw):
            out = klt0(img0, img1, pts, **kw)
            last.update(img0=np.array(img0), img1=np.array(img1), pts=np.array(pts, np.float32))
            return out

        def pnp_rec(self, X, x):
            M, inl = pnp0(self, X, x)
            if last and len(calls) < 4 * a.cpu_baseline:
                calls.append(dict(last, X=np.array(X), x=np.array(x), M=np.array(M)))
            return M, inl

        PoseEstimation.KLT = staticmethod(klt_rec)
        PoseEstimation.PnP = pnp_rec
    parts, windows = {}, []       # windows: (cameras, landmarks, observations, trial steps, accepted steps) per BA call
    if a.profile:
        from visual_odometry_amd.vo_pipeline import continuousVO as cvo, keypointTrajectory as kt
        from visual_odometry_amd.vo_pipeline.poseEstimation import PoseEstimation as PE

        def timed(owner, name, label, static=False):
            fn = getattr(owner, name)

            def wrapper(*args, **kw):
                t0 = time.perf_counter()
                out = fn(*args, **kw)
                parts.setdefault(label, []).append(time.perf_counter() - t0)
                return out
            setattr(owner, name, staticmethod(wrapper) if static else wrapper)
        timed(PE, "KLT", "KLT", static=True)
        timed(PE, "PnP", "PnP")
        timed(cvo, "filter_tracked", "filter_tracked")
        timed(cvo, "baseline_uncertainty", "baseline_uncertainty")
        timed(kt.KeypointTrajectories, "at_frame", "at_frame")
        timed(kt.KeypointTrajectories, "tracked_to_many", "tracked_to_many")
        for name in ("track_and_pose",):
            if hasattr(PE, name):
                timed(PE, name, name)
        from visual_odometry_amd.vo_pipeline import bootstrap as bs, ransac as rs, featureExtraction as fe, featureMatching as fm
        timed(cvo.ContinuousVO, "_bootstrap", "_bootstrap (whole)")
        timed(cvo.ContinuousVO, "_bundle_adjustment", "_bundle_adjustment (whole)")
        timed(cvo.ContinuousVO, "_bundle_adjustment_begin", "_bundle_adjustment_begin")
        timed(cvo.ContinuousVO, "_bundle_adjustment_finish", "_bundle_adjustment_finish")
        timed(cvo.ContinuousVO, "_bootstrap_views", "_bootstrap_views (two-view reconstruction)")
        timed(fe.FeatureExtractor, "sift_arrays", "sift_arrays")
        timed(fm.FeatureMatcher, "match_arrays", "match_arrays")
        timed(bs.BootstrapInitializer, "_estimate_fundamental", "_estimate_fundamental (sift x2 + match + RANSAC)")
        timed(bs.BootstrapInitializer, "_transform_matrix", "_transform_matrix")
        timed(fe.FeatureExtractor, "corners", "corners")
        from visual_odometry_amd import hip as _hip
        from visual_odometry_amd.vo_pipeline import bundleAdjustment as bam
        timed(cvo.ContinuousVO, "_ba_window", "_ba_window")
        timed(bam.BundleAdjustment, "bundle_adjustment", "bundle_adjustment")
        timed(_hip.BaSolver, "__init__", "BaSolver.__init__")
        timed(_hip.BaSolver, "filtered", "BaSolver.filtered", static=True)      # a bound classmethod: no self to pass on
        ba_run = bam.BundleAdjustment.bundle_adjustment

        def ba_recorded(self, M, landmarks, ci, *args, **kw):
            out = ba_run(self, M, landmarks, ci, *args, **kw)
            st = getattr(self, "last_stats", None) or {}
            windows.append([int(M.shape[0]), int(landmarks.shape[0]), int(np.asarray(ci).size), int(st.get("trial_steps", 0)),
                            int(st.get("accepted", 0))])
            return out
        bam.BundleAdjustment.bundle_adjustment = ba_recorded
        timed(_hip.BaSolver, "try_step", "BaSolver.try_step")
        timed(_hip.BaSolver, "solve", "BaSolver.solve")
        timed(_hip.BaSolver, "solve_begin", "BaSolver.solve_begin")
        timed(_hip.BaSolver, "solve_finish", "BaSolver.solve_finish")
        timed(_hip.BaSolver, "linearize", "BaSolver.linearize")
        timed(_hip.BaSolver, "params", "BaSolver.params")
        timed(_hip, "ba_residual", "hip.ba_residual")
        timed(_hip, "_ba_prepare", "hip._ba_prepare")
        timed(_hip.BaSolver, "accept", "BaSolver.accept")
        timed(_hip.BaSolver, "close", "BaSolver.close")
        timed(bam.BundleAdjustment, "_preprocess_data", "BA._preprocess_data")
        timed(bam.BundleAdjustment, "_camera_params", "BA._camera_params", static=True)
    # everything imported and built so far (torch, scipy, the synthetic sequence) into the collector's permanent
    # generation: a full collection inside the loop then walks the loop's own objects, not a million module-level
    # ones (such a pause is 5-10 ms, a tracked frame 0.3 ms)
    gc.collect()
    gc.freeze()
    per_frame = []
    while True:
        t0 = time.perf_counter()
        if vo.step() is None:
            break
        per_frame.append(time.perf_counter() - t0)
    per_frame = np.array(per_frame)
    rel, rot, scale = trajectory_error(vo, np.asarray(data.T, np.float64), vo.frames_to_skip)
    tracked = per_frame[vo.frames_to_skip + 1:]
    out_cpu = {}
    if a.cpu_baseline and calls:
        step = max(1, len(calls) // a.cpu_baseline)
        out_cpu = {"cpu_baseline": cpu_frame_baseline(calls[::step][:a.cpu_baseline], np.asarray(data.K, np.float64))}
    print(json.dumps({"frames": len(per_frame), "bootstrap": a.bootstrap, "fused_step": not a.separate_calls, "overlap_keyframe": not a.no_overlap, "keyframes": [k.idx for k in vo.keyframes],
                      "loop_frames_per_s": float(len(tracked) / tracked.sum()),
                      "median_ms_per_tracked_frame": float(np.median(tracked) * 1e3),
                      "bootstrap_ms": float(per_frame[vo.frames_to_skip] * 1e3),
                      "ms_of_keyframe_steps": [round(float(per_frame[k.idx]) * 1e3, 3) for k in vo.keyframes[1:]],
                      "slowest_other_steps": [(int(i), round(float(per_frame[i]) * 1e3, 3))
                                              for i in sorted((i for i in range(vo.frames_to_skip + 1, len(per_frame))
                                                               if i not in {k.idx for k in vo.keyframes}),
                                                              key=lambda i: -per_frame[i])[:5]],
                      "rel_position_rms": rel, "max_rotation_deg": rot, "scale": scale,
                      "landmarks": len(vo.keypoint_trajectories.landmarks), **out_cpu,
                      **({"median_ms_of_parts": {k: float(np.median(v) * 1e3) for k, v in parts.items()},
                          "total_ms_of_parts": {k: float(np.sum(v) * 1e3) for k, v in parts.items()},
                          "calls_of_parts": {k: len(v) for k, v in parts.items()}, "ba_windows": windows,
                          "ba_ms_per_window": {k: [round(t * 1e3, 3) for t in parts.get(k, [])]
                                               for k in ("BaSolver.filtered", "BaSolver.solve", "BaSolver.solve_begin", "BaSolver.solve_finish", "_ba_window")}} if parts else {})}))


if __name__ == "__main__":
    main()
