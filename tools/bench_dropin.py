#!/usr/bin/env python3
"""Latency of the drop-in (host-memory, PCIe-inclusive, synchronous) calls -- the way the
reference's single-stream loop would use them.  Not the headline metric."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


SPIKES = {}


def t(fn, reps=40, warm=3):
    """Median latency of fn in ms (a call in a few hundred takes milliseconds -- allocator or scheduler --, which a mean
    over tens of calls would smear over all of them); the largest time seen is kept in SPIKES."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    SPIKES[len(SPIKES)] = max(ts) * 1e3
    return float(np.median(ts)) * 1e3


def _tm(p1, p2, sc, synth):
    from visual_odometry_amd.vo_pipeline.bootstrap import BootstrapInitializer
    b = object.__new__(BootstrapInitializer)
    b.K, b.max_point_dist = synth.K_KITTI, 100
    return b._transform_matrix(sc["F"], p1, p2)


def main():
    from visual_odometry_amd import synth
    from visual_odometry_amd.vo_pipeline.bootstrap import BootstrapInitializer
    from visual_odometry_amd.vo_pipeline.featureExtraction import FeatureExtractor
    from visual_odometry_amd.vo_pipeline.poseEstimation import AlgoMethod, PoseEstimation
    from visual_odometry_amd.vo_pipeline.trackPoints import TrackPoints
    img0, img1, _ = synth.frame_pair(synth.KITTI_SHAPE, seed=2021)
    pts = synth.uniform_keypoints(synth.KITTI_SHAPE, 2000, seed=0)
    sc = synth.two_view_scene(n=2000, K=synth.K_KITTI, shape=synth.KITTI_SHAPE, seed=3)
    pe = PoseEstimation(synth.K_KITTI, algo_method_type=AlgoMethod.P3P)
    tp = TrackPoints(21, 10, 0.1)
    from scipy.spatial.transform import Rotation
    from visual_odometry_amd.vo_pipeline.bundleAdjustment import BundleAdjustment
    pb = synth.ba_problem(F=10, P=2000, visibility=0.8, seed=1, obs_noise=0.3)
    rng = np.random.default_rng(0)
    cams = pb["x"][:60].reshape(10, 6).copy()
    cams[3:, 3:] += rng.normal(0, 0.02, (7, 3))
    M = np.zeros((10, 3, 4), np.float32)
    for i in range(10):
        M[i, :, :3] = Rotation.from_rotvec(cams[i, :3]).as_matrix()
        M[i, :, 3] = cams[i, 3:]
    landmarks = (pb["x"][60:].reshape(-1, 3) + rng.normal(0, 0.1, (2000, 3))).astype(np.float32)
    ba = BundleAdjustment(synth.K_KITTI)
    p1 = np.hstack([sc["m1"], np.ones((2000, 1), np.float32)]).astype(np.float32)
    p2 = np.hstack([sc["m2"], np.ones((2000, 1), np.float32)]).astype(np.float32)
    M1 = (synth.K_KITTI @ np.eye(3, 4)).astype(np.float32)
    M2 = synth.K_KITTI.astype(np.float64) @ np.hstack([sc["R"], sc["t"].reshape(3, 1)])
    from visual_odometry_amd.vo_pipeline.featureMatching import FeatureMatcher, MatcherType
    des0, des1, _ = synth.sift_like_descriptors(3000, 3000, seed=5)
    fm = FeatureMatcher(MatcherType.FLANN, k=2, matching_threshold=0.8)
    res = {
        "PoseEstimation.KLT ms (1241x376, 2000 kp, host arrays, both pyramids rebuilt)": t(lambda: PoseEstimation.KLT(img0, img1, pts)),
        "TrackPoints.trackKLT ms (forward+backward)": t(lambda: tp.trackKLT(img0, img1, pts)),
        "PoseEstimation.PnP ms (2000 points, 100 RANSAC iterations + refit)": t(lambda: pe.PnP(sc["X"], sc["m2"]), reps=10),
        "findFundamentalMat ms (2000 points, 100 iterations)": t(lambda: BootstrapInitializer.findFundamentalMat(sc["m1"], sc["m2"], 0.1, 0.9999, 100), reps=10),
        "FeatureExtractor.harris ms (1241x376: mask, connected components, centroids, cornerSubPix in one GPU call)": t(lambda: FeatureExtractor.harris(img0), reps=10),
        "FeatureExtractor.corners ms (top-2000)": t(lambda: FeatureExtractor.corners(img0, 2000)),
        "BootstrapInitializer._transform_matrix ms (2000 pairs: E decomposition on the host, four triangulations + vote in one launch)":
            t(lambda: _tm(p1, p2, sc, synth)),
        "BootstrapInitializer._linear_triangulation ms (2000 pairs)": t(lambda: BootstrapInitializer._linear_triangulation(p1, p2, M1, M2)),
        "FeatureMatcher.match_descriptors ms (3000 x 3000 descriptors, list of DMatch objects)": t(lambda: fm.match_descriptors(des0, des1)),
        "FeatureMatcher.match_arrays ms (3000 x 3000 descriptors, index/distance/mask arrays)": t(lambda: fm.match_arrays(des0, des1)),
        f"BundleAdjustment.bundle_adjustment ms (10 keyframes x 2000 landmarks, {pb['ci'].size} observations, to ftol 1e-4)":
            t(lambda: ba.bundle_adjustment(M, landmarks, pb["ci"], pb["pi"], pb["obs"]), reps=5, warm=1),
    }
    res["largest single call per entry, ms"] = [round(v, 3) for v in SPIKES.values()]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
