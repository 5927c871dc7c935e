#!/bin/bash
# Evidence for the bundle-adjustment trial step: tools/evidence_ba.sh OUTDIR
#   step times on loop-shaped windows (default kernels and the strip-wise Schur kernel), kernel durations from a
#   trace, counters of the step's kernels, and the allocation calls of a 120-frame loop.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
cd "$REPO"
{ echo "# tools/ab/ba_window_step.py: median wall time of vo_ba_try_step (ms), default kernels then VO_BA_SCHUR=strips";
  timeout -k 10 200 python3 tools/ab/ba_window_step.py;
  VO_BA_SCHUR=strips timeout -k 10 200 python3 tools/ab/ba_window_step.py; } > "$OUT/ba_step.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/ab/ba_window_step.py" 4 > /dev/null 2>&1
{ echo "# kernel durations (us) over windows of 47, 77, 107 cameras, rocprofv3 --kernel-trace"; python3 "$REPO/tools/ab/kernel_times.py" "$OUT/trace" ba_; } >> "$OUT/ba_step.txt"
bash "$REPO/tools/pmc_cmd.sh" "$OUT/pmc" ba_ "$REPO/tools/ab/ba_window_step.py" 4 > "$OUT/pmc_ba_step.txt"
python3 "$REPO/tools/pmc_summary.py" "$OUT/pmc" > "$OUT/pmc_ba_step.json"
cd /tmp
timeout -k 10 300 rocprofv3 --hip-trace --output-format csv -d "$OUT/hip" -- python3 "$REPO/tools/run_vo.py" --frames 120 --bootstrap klt > "$OUT/hip_run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/hip/**/*hip_api_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0, last = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
names = ("hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipMallocAsync", "hipFreeAsync", "hipExtMallocWithFlags")
with open(out + "/loop_hip_alloc.txt", "w") as o:
    o.write("# allocation calls of tools/run_vo.py --frames 120 --bootstrap klt (rocprofv3 --hip-trace): ms since the first HIP call, duration, ms before the last call\n")
    o.write("# %d HIP calls in all; the frame loop is the last ~100 ms before the frees of the teardown\n" % len(rows))
    for r in rows:
        if r["Function"] in names:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            o.write("%-16s %9.1f ms  %7.1f us  %8.1f ms before end\n" % (r["Function"], (s - t0) / 1e6, (e - s) / 1e3, (last - s) / 1e6))
PY
grep "^{" "$OUT/hip_run.log" | cut -c1-200
