#!/bin/bash
# Allocation calls inside a loop run, from a HIP API trace: tools/ab/loop_allocs.sh OUTDIR route frames
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-trace --output-format csv -d "$OUT/hip" -- python3 "$REPO/tools/run_vo.py" --frames ${3:-120} --bootstrap ${2:-sift} > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/hip/**/*hip_api_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0, last = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
names = ("hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipMallocAsync", "hipFreeAsync", "hipExtMallocWithFlags")
for r in rows:
    if r["Function"] in names:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-16s %9.1f ms  %8.1f us  %8.1f ms before end" % (r["Function"], (s - t0) / 1e6, (e - s) / 1e3, (last - s) / 1e6))
PY
grep "^{" "$OUT/run.log" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['loop_frames_per_s']), d['ms_of_keyframe_steps'])"
