#!/usr/bin/env python3
"""profiles/frame_chain.json: what the launch chain behind the tracker costs in `bench.py --mode frame`, from the
rocprofv3 kernel statistics of that command and a PMC summary (tools/pmc_summary.py) of the same workload run on one
stream with one launch group.  bench.py puts the record on its frame-mode line as roofline.chain.
usage: tools/make_frame_chain.py <kernel_stats.csv> <pmc_summary.json> <streams per launch in the PMC run> <hypotheses scored per stream> <command string>"""
import csv
import json
import sys

stats, summary, streams, hyps, command = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
N_KP = 2000
rows = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(stats))}
klt = rows["klt_track3_kernel"]
steps = int(klt["Calls"])
chain = {}
for name in ("frame_prep_kernel", "p3p_kernel", "score_kernel<true>", "frame_bound_kernel", "frame_mask_kernel", "refine_pose_kernel",
             "copy_pinned_wide_kernel", "pyr_down_rows_kernel<4>", "pyr_down_rows_kernel<8>", "klt_track3_kernel"):
    if name in rows:
        r = rows[name]
        chain[name] = {"launches_per_tracker_launch": round(int(r["Calls"]) / steps, 2), "average_us": round(float(r["AverageNs"]) / 1e3, 2),
                       "us_per_tracker_launch": round(float(r["TotalDurationNs"]) / steps / 1e3, 2)}
behind = sum(v["us_per_tracker_launch"] for k, v in chain.items() if not k.startswith(("klt", "pyr")))
pmc = {k.replace("void ", ""): v for k, v in json.load(open(summary)).items()}
out = {"command": command, "in_the_benchmarks_schedule": {"note": "two launch groups: a kernel's duration includes waiting for compute units the other "
                                                                   "group's tracker holds", "kernels": chain,
                                          "chain_us_per_tracker_launch": round(behind, 1),
                                          "tracker_us_per_launch": chain["klt_track3_kernel"]["us_per_tracker_launch"]}}
# the same kernels alone on the device: the kernel trace of a counter pass (one launch group, launches serialised)
import glob
import os
alone = {}
for f in glob.glob(os.path.join(os.path.dirname(summary), "pmc", "pass1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        alone.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if alone:
    n_klt = len(alone.get("klt_track3_kernel", [1]))
    out["alone_on_the_device"] = {"streams_per_launch": streams,
                                  "kernels": {k: {"launches_per_tracker_launch": round(len(v) / n_klt, 2), "average_us": round(sum(v) / len(v), 2)}
                                              for k, v in alone.items() if k in chain},
                                  "chain_us_per_tracker_launch": round(sum(sum(v) for k, v in alone.items()
                                                                           if k in chain and not k.startswith(("klt", "pyr"))) / n_klt, 1)}
for name, key in (("score_kernel<true>", "score_kernel"), ("p3p_kernel", "p3p_kernel"), ("refine_pose_kernel", "refine_pose_kernel"),
                  ("frame_prep_kernel", "frame_prep_kernel"), ("frame_mask_kernel", "frame_mask_kernel")):
    k = pmc.get(name) or pmc.get(key)
    if not k:
        continue
    # SQ_ACTIVE_INST_VALU, SQ_WAIT_* and SQ_WAVE_CYCLES count in the same unit (four clock cycles): shares of the cycles
    # the kernel's wavefronts were resident
    rec = {"valu_share_of_wave_cycles": round(k["SQ_ACTIVE_INST_VALU"] / max(k["SQ_WAVE_CYCLES"], 1), 4),
           "waiting_for_an_instruction_share": round(k["SQ_WAIT_INST_ANY"] / max(k["SQ_WAVE_CYCLES"], 1), 4),
           "waiting_for_anything_share": round(k["SQ_WAIT_ANY"] / max(k["SQ_WAVE_CYCLES"], 1), 4),
           "valu_wave_instructions": round(k["SQ_INSTS_VALU"]), "wavefronts": round(k["SQ_WAVES"]), "dispatches": k["dispatches"]}
    if key == "score_kernel":
        # the two launches of a staged chunk: the mean dispatch scores half the stream's hypotheses
        # a staged chunk is two launches; those of the second beyond the loop's bound leave at once, so the counted
        # instructions are the first stage's 32 hypotheses per stream (clean streams end their loop inside it)
        evals = streams * 32 * N_KP
        rec["lane_instructions_per_evaluation"] = round(2 * k["SQ_INSTS_VALU"] * 64 / evals, 1)
        t = sorted(alone.get("score_kernel<true>", []))
        if t:
            first = sum(t[len(t) // 2:]) / len(t[len(t) // 2:])          # the longer half of the launches: the first stages
            rate = rec["lane_instructions_per_evaluation"] * evals / (first * 1e-6) / 1e12
            rec["first_stage_us_alone"] = round(first, 2)
            rec["float64_lane_instructions_per_s"] = round(rate * 1e12)
            rec["float64_issue_peak_lane_instructions_per_s"] = 256 * 4 * 16 * 2.4e9      # 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz
            rec["frac_of_float64_issue_peak"] = round(rate * 1e12 / (256 * 4 * 16 * 2.4e9), 3)
    out.setdefault("counters", {})[key] = rec
out["binder"] = ("latency: every kernel of the chain is a few wavefronts per compute unit for a few microseconds (float64 chains of "
                 "dependent instructions; VALU share of the resident wavefronts' cycles above), far from both the HBM and the float64 "
                 "vector roof; the chain costs what its launches and their dependent latencies cost")
print(json.dumps(out, indent=1))
