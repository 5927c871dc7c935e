#!/usr/bin/env python3
"""Seed sweep of the corner operators on small, odd and padded inputs: response maps of images from 3x3 up,
device images with padded rows, NMS/top-K of arbitrary float maps (ties, constant maps, negative values) with
radii up to 32 and K up to 4096, masks, all bit-exact against the CPU oracle."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=60)
    a = ap.parse_args()
    import torch
    from oracle import oracle
    from visual_odometry_amd import hip
    ctx = hip.Context(0)
    tctx = hip.Context.from_torch(0)
    rng = np.random.default_rng(2468)
    n_resp = n_nms = 0
    for seed in range(a.seeds):
        h, w = int(rng.integers(3, 150)), int(rng.integers(3, 200))
        kind = seed % 4
        if kind == 0:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:
            img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)          # flat: response zero everywhere
        elif kind == 2:
            img = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)           # binary: many equal responses
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            img = ((yy // 7 + xx // 5) % 2 * 180 + 30).astype(np.uint8)
        k = float(rng.choice([0.04, 0.1, 0.2]))
        for mode in (hip.HARRIS, hip.MIN_EIGEN):
            Ro = oracle.harris_response(img, k, mode)
            R = hip.harris_response(ctx, img, k, mode)[0]
            assert (R.view(np.uint32) == Ro.view(np.uint32)).all(), ("response", seed, mode, h, w)
            # device image with padded rows
            big = torch.from_numpy(rng.integers(0, 256, (h + 2, w + 7), dtype=np.uint8)).cuda()
            big[1:1 + h, 3:3 + w] = torch.from_numpy(img).cuda()
            Rd = hip.harris_response(tctx, big[1:1 + h, 3:3 + w], k, mode)
            tctx.sync()
            assert (Rd[0].cpu().numpy().view(np.uint32) == Ro.view(np.uint32)).all(), ("response strided", seed, mode)
            rel = float(rng.choice([0.0, 0.01, 0.3, 0.999]))
            assert (hip.harris_mask(ctx, img, k, rel, mode) == oracle.harris_mask(img, k, rel, mode)).all(), ("mask", seed, mode, rel)
            n_resp += 1
            radius, K = int(rng.integers(1, 33)), int(rng.choice([1, 2, 17, 500, 4096]))
            xy, sc, cnt = hip.detect_corners(ctx, img[None], K, rel, radius, k, mode)
            oxy, osc, om = oracle.nms_topk(Ro, K, rel, radius)
            assert int(cnt[0]) == om and (xy[0, :om] == oxy[:om]).all() and (sc[0, :om] == osc[:om]).all(), ("detect", seed, mode, radius, K, rel)
        # NMS / top-K of arbitrary maps
        hh, ww = int(rng.integers(1, 120)), int(rng.integers(1, 160))
        choice = seed % 3
        if choice == 0:
            Rm = rng.normal(0, 1, (hh, ww)).astype(np.float32)
        elif choice == 1:
            Rm = rng.integers(-2, 3, (hh, ww)).astype(np.float32)                # plateaus and ties
        else:
            Rm = np.full((hh, ww), float(rng.choice([-1.0, 0.0, 2.5])), np.float32)
        for radius in (1, int(rng.integers(2, 33))):
            K = int(rng.choice([1, 3, 100, 4096]))
            rel = float(rng.choice([0.0, 0.01, 0.5]))
            xy, sc, cnt = hip.nms_topk(ctx, Rm[None], K, rel, radius)
            oxy, osc, om = oracle.nms_topk(Rm, K, rel, radius)
            assert int(cnt[0]) == om and (xy[0, :om] == oxy[:om]).all() and (sc[0, :om] == osc[:om]).all(), ("nms", seed, hh, ww, radius, K, rel, choice)
            n_nms += 1
    print(json.dumps({"op": "corners", "seeds": a.seeds, "response_cases": n_resp, "nms_cases": n_nms, "bit_exact": True}), flush=True)


if __name__ == "__main__":
    main()
