#!/usr/bin/env python3
"""SIFT on one KITTI-shaped frame, repeated (run it under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from visual_odometry_amd import hip, synth
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = hip.Context.from_torch(0)
    img = torch.from_numpy(synth.frame_pair(synth.KITTI_SHAPE, seed=2021)[0]).cuda()
    kp, des = hip.sift_detect_compute(ctx, img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.sift_detect_compute(ctx, img)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(json.dumps({"op": "sift_detect_compute", "shape": list(synth.KITTI_SHAPE), "keypoints": int(len(kp)), "ms": ms}))


if __name__ == "__main__":
    main()
