#!/usr/bin/env python3
"""Seed sweep of the pyramid builder: random sizes from 22 pixels up, batches of 1..17 images (multiples of 8
take the image-per-XCD workgroup order), host images and device images with padded rows and image strides,
random requested depth; every level of every image bit-exact against the CPU oracle."""
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
    rng = np.random.default_rng(1357)
    levels = 0
    for seed in range(a.seeds):
        h, w = int(rng.integers(22, 500)), int(rng.integers(22, 1300))
        if seed % 5 == 0:
            h, w = int(rng.integers(22, 60)), int(rng.integers(22, 60))
        B = int(rng.choice([1, 2, 3, 8, 16, 17]))
        ml = int(rng.integers(0, 9))
        imgs = rng.integers(0, 256, (B, h, w), dtype=np.uint8)
        want_ml = oracle.pyr_max_level(w, h, ml)
        for device in (False, True):
            if device:
                pr, pc, pi = int(rng.integers(0, 9)), int(rng.integers(0, 70)), int(rng.integers(0, 4))
                big = torch.from_numpy(rng.integers(0, 256, (B, h + pr + pi, w + pc), dtype=np.uint8)).cuda()
                x0 = int(rng.integers(0, pc + 1))
                big[:, pr:pr + h, x0:x0 + w] = torch.from_numpy(imgs).cuda()
                pyr = hip.Pyramid(tctx, B, h, w, ml).build(big[:, pr:pr + h, x0:x0 + w])
                tctx.sync()
            else:
                pyr = hip.Pyramid(ctx, B, h, w, ml).build(imgs)
            assert pyr.max_level == want_ml, ("levels", seed, h, w, ml, pyr.max_level, want_ml)
            for b in sorted(set([0, B - 1, int(rng.integers(0, B))])):
                for lvl in range(pyr.max_level + 1):
                    assert (pyr.download(b, lvl) == oracle.pyr_level(imgs[b], lvl)).all(), ("pyramid", seed, device, b, lvl, h, w, B)
                    levels += 1
            pyr.close()
    print(json.dumps({"op": "pyramid", "seeds": a.seeds, "levels_compared": levels, "bit_exact": True}), flush=True)


if __name__ == "__main__":
    main()
