"""Kernel timeline of single-stream tracked frames from a rocprofv3 --kernel-trace csv of tools/run_vo.py: for every
klt_track launch, the chain up to the next result copy -- name, start offset, duration (us) -- medians over frames.
usage: frame_timeline.py TRACE_DIR"""
import csv
import glob
import statistics
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
chains = []
i = 0
while i < len(rows):
    if names[i].startswith("copy_pinned_wide_kernel") and i + 4 < len(rows) and names[i + 1].startswith("pyr_down"):
        j = i
        while j < len(rows) and not names[j].startswith("refine_pose_kernel") and j - i < 24:
            j += 1
        if j < len(rows) and j - i < 24:
            j = min(j + 1, len(rows) - 1)                   # the result copy behind the refit
            chains.append([(names[k], (starts[k] - starts[i]) / 1e3, (ends[k] - starts[k]) / 1e3) for k in range(i, j + 1)])
            i = j
    i += 1
lengths = [len(c) for c in chains]
mode = statistics.mode(lengths)
sel = [c for c in chains if len(c) == mode]
print("frames with the usual chain of", mode, "kernels:", len(sel), "of", len(chains))
for k in range(mode):
    print("%-34s start %7.1f us  duration %6.1f us" % (sel[0][k][0][:34], statistics.median(c[k][1] for c in sel),
                                                        statistics.median(c[k][2] for c in sel)))
print("chain end %.1f us after its first kernel's start" % statistics.median(c[-1][1] + c[-1][2] for c in sel))
