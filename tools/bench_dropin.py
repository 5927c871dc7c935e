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


def t(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


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
    res = {
        "PoseEstimation.KLT ms (1241x376, 2000 kp, host arrays, both pyramids rebuilt)": t(lambda: PoseEstimation.KLT(img0, img1, pts)),
        "TrackPoints.trackKLT ms (forward+backward)": t(lambda: tp.trackKLT(img0, img1, pts)),
        "PoseEstimation.PnP ms (2000 points, 100 RANSAC iterations + refit)": t(lambda: pe.PnP(sc["X"], sc["m2"]), reps=10),
        "findFundamentalMat ms (2000 points, 100 iterations)": t(lambda: BootstrapInitializer.findFundamentalMat(sc["m1"], sc["m2"], 0.1, 0.9999, 100), reps=10),
        "FeatureExtractor.corners ms (top-2000)": t(lambda: FeatureExtractor.corners(img0, 2000)),
    }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
