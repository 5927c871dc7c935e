#!/usr/bin/env python3
"""How much does lock-step iteration of three keypoints per wavefront cost?  Runs the CPU oracle's LK on
frame pairs of the benchmark's streams with its per-keypoint, per-level iteration log, groups keypoints
in threes exactly as klt_track3_kernel does (consecutive indices) and compares the mean iteration count
of a keypoint with the mean of the maximum over its group -- the count the wavefront executes.  Also
reports how often a keypoint's tile offset leaves the staging position during a level (the reads whose
banks can collide).  CPU only."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--max-level", type=int, default=bench.MAX_LEVEL)
    a = ap.parse_args()
    frames, pts, _ = bench.make_streams(list(range(a.streams)), 3)
    L = oracle.lib()
    L.vo_or_klt_set_iter_log.argtypes = [ctypes.c_void_p]
    per_kp, per_wave, levels_run = [], [], 0
    for s in range(a.streams):
        for r in range(2):
            log = np.zeros((bench.N_KP, 16), np.int32)
            L.vo_or_klt_set_iter_log(log.ctypes.data)
            oracle.klt(frames[r, s], frames[r + 1, s], pts[s], max_level=a.max_level)
            L.vo_or_klt_set_iter_log(None)
            it = log[:, :a.max_level + 1].astype(np.float64)            # (keypoint, level)
            n3 = bench.N_KP // 3 * 3
            grp = it[:n3].reshape(-1, 3, it.shape[1])
            per_kp.append(it[:n3].mean(axis=0))
            per_wave.append(grp.max(axis=1).mean(axis=0))
    # Chained tracks: does the count a keypoint needed from frame t-1 to t predict the count from t to t+1?  If it did,
    # ordering a stream's keypoints by their last count (free when they are compacted) would put keypoints of similar
    # length into one wavefront.  Per stream: track 0 -> 1, then the tracked positions 1 -> 2; correlate the per-keypoint
    # totals over the levels, and regroup the second pair's keypoints in threes after sorting by the first pair's total.
    corr, over_plain, over_sorted, over_ideal = [], [], [], []
    for s in range(a.streams):
        logs, p = [], pts[s]
        ok = np.ones(bench.N_KP, bool)
        for r in range(2):
            log = np.zeros((bench.N_KP, 16), np.int32)
            L.vo_or_klt_set_iter_log(log.ctypes.data)
            out, st, _ = oracle.klt(frames[r, s], frames[r + 1, s], p, max_level=a.max_level)
            L.vo_or_klt_set_iter_log(None)
            logs.append(log[:, :a.max_level + 1].astype(np.float64))
            ok &= st.ravel() == 1
            p = out
        a0, a1 = logs[0][ok], logs[1][ok]
        corr.append(float(np.corrcoef(a0.sum(axis=1), a1.sum(axis=1))[0, 1]))
        n3 = a1.shape[0] // 3 * 3

        def overhead(order):
            g = a1[order[:n3]].reshape(-1, 3, a1.shape[1])
            return float(3.0 * g.max(axis=1).sum() / a1[order[:n3]].sum() - 1.0)     # a wavefront runs its longest keypoint's count for all three
        over_plain.append(overhead(np.arange(a1.shape[0])))
        over_sorted.append(overhead(np.argsort(a0.sum(axis=1), kind="stable")))
        over_ideal.append(overhead(np.argsort(a1.sum(axis=1), kind="stable")))       # an oracle that knew this frame's counts
    per_kp, per_wave = np.mean(per_kp, axis=0), np.mean(per_wave, axis=0)
    res = {"streams": a.streams, "frame_pairs": 2 * a.streams, "levels": a.max_level + 1,
           "iterations_per_keypoint_by_level": [round(float(v), 3) for v in per_kp],
           "iterations_per_wavefront_by_level": [round(float(v), 3) for v in per_wave],
           "mean_keypoint": round(float(per_kp.mean()), 3), "mean_wavefront": round(float(per_wave.mean()), 3),
           "lockstep_overhead": round(float(per_wave.sum() / per_kp.sum() - 1.0), 4),
           "chained": {"correlation_of_consecutive_frames_counts": round(float(np.mean(corr)), 3),
                       "lockstep_overhead_in_index_order": round(float(np.mean(over_plain)), 4),
                       "lockstep_overhead_sorted_by_last_frames_count": round(float(np.mean(over_sorted)), 4),
                       "lockstep_overhead_sorted_by_this_frames_count": round(float(np.mean(over_ideal)), 4)}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
