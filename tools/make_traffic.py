#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py) of the tracker kernel: HBM bytes per launch,
VALU issue share, LDS conflict / busy shares, VALU instructions per keypoint.  bench.py puts these on its line.
usage: tools/make_traffic.py <pmc_summary.json> <frames_per_launch> <keypoints_per_frame> <command string> > profiles/traffic.json"""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench

summary, frames, n_kp, command = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
k = json.load(open(summary))["klt_track3_kernel"]
_, klt_bytes = bench.algorithmic_bytes(bench.MAX_LEVEL, n_kp)
hbm = k["FETCH_SIZE"] * 1024 * 2 + k["WRITE_SIZE"] * 1024
xcd_cycles = k["GRBM_GUI_ACTIVE"] / 8
out = {
    "kernel": "klt_track3_kernel", "frames_per_launch": frames, "round": 3, "command": command,
    "FETCH_SIZE_KB_per_launch": k["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": k["WRITE_SIZE"],
    "TCC_MISS_per_launch": k["TCC_MISS_sum"], "TCC_REQ_per_launch": k["TCC_REQ_sum"],
    "hbm_bytes_per_launch": int(round(hbm)), "algorithmic_bytes_per_launch": int(klt_bytes * frames),
    "traffic_over_algorithmic": round(hbm / (klt_bytes * frames), 3),
    "correction": "gfx950 tallies 128-B read requests at 64 B, so FETCH_SIZE is doubled; WRITE_SIZE taken as is. "
                  "TCC_MISS_sum x 128 B is given for comparison.",
    "TCC_MISS_x128_bytes": int(k["TCC_MISS_sum"] * 128),
    "valu_issue": round(k["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * xcd_cycles), 4),
    "valu_issue_definition": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
    "lds_conflict_frac": round(k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"], 4),
    "lds_conflict_definition": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
    "lds_busy_frac": round(k["SQ_LDS_IDX_ACTIVE"] / (256 * xcd_cycles), 4),
    "valu_insts_per_keypoint": round(k["SQ_INSTS_VALU"] / (frames * n_kp), 1),
    "salu_insts_per_keypoint": round(k["SQ_INSTS_SALU"] / (frames * n_kp), 1),
    "counters_from": summary,
}
print(json.dumps(out, indent=1))
