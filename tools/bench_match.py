#!/usr/bin/env python3
"""Timing of vo_knn_match2 alone (device-resident descriptors, HIP events on the context's stream)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_ops import timed  # noqa: E402


def main():
    import torch
    from visual_odometry_amd import hip, synth
    ctx = hip.Context.from_torch(0)
    dev = torch.device("cuda", 0)
    for n1, n2, kind in [(3000, 3000, "bytes"), (3000, 3000, "float"), (16384, 16384, "bytes"), (16384, 16384, "float"),
                         (65536, 65536, "bytes")]:
        d1, d2, _ = synth.sift_like_descriptors(n1, n2, seed=5)
        if kind == "float":
            d1, d2 = d1 * np.float32(1.0 / 512), d2 * np.float32(1.0 / 512)
        a, b = torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev)
        ms = timed(ctx, lambda: hip.knn_match2(ctx, a, b, 0.8))
        print(json.dumps({"n1": n1, "n2": n2, "kind": kind, "ms": ms, "tflops": 2.0 * n1 * n2 * 128 / (ms * 1e-3) / 1e12}), flush=True)


if __name__ == "__main__":
    main()
