#!/usr/bin/env python3
"""End-to-end run of the ContinuousVO loop on the synthetic 640x480 parking-shaped sequence:
frames/s of the whole loop (host numpy in and out of every operator, as the reference calls
them) and the error of the recovered trajectory against the scene's ground truth."""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--dataset-params", default="PARKING")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    if a.v:
        logging.basicConfig(level=logging.INFO, format="%(message)s")
    params.use_dataset(a.dataset_params)
    data = loadData.synthetic_parking(a.frames, a.seed)
    frames = list(data.frames)                   # rendering is not part of the timed loop
    data.frames = iter(frames)
    vo = ContinuousVO(data)
    per_frame = []
    while True:
        t0 = time.perf_counter()
        if vo.step() is None:
            break
        per_frame.append(time.perf_counter() - t0)
    per_frame = np.array(per_frame)
    rel, rot, scale = trajectory_error(vo, np.asarray(data.T, np.float64), vo.frames_to_skip)
    tracked = per_frame[vo.frames_to_skip + 1:]
    print(json.dumps({"frames": len(per_frame), "keyframes": [k.idx for k in vo.keyframes],
                      "loop_frames_per_s": float(len(tracked) / tracked.sum()),
                      "median_ms_per_tracked_frame": float(np.median(tracked) * 1e3),
                      "bootstrap_ms": float(per_frame[vo.frames_to_skip] * 1e3),
                      "rel_position_rms": rel, "max_rotation_deg": rot, "scale": scale,
                      "landmarks": len(vo.keypoint_trajectories.landmarks)}))


if __name__ == "__main__":
    main()
