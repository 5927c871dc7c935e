#!/bin/bash
# HBM read traffic and duration of sift_extrema_kernel per octave-0 launch (rocprofv3 --pmc FETCH_SIZE, kernel trace).
# usage: tools/ab/sift_extrema_traffic.sh <outdir>
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/${1:-gpurun_out/sift_extrema}; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/pmc -- python3 $R/tools/bench_sift.py 3 > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_sift.py 20 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(float); grid = {}
for r in csv.DictReader(open(f)):
    if "sift_extrema" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        per[r["Dispatch_Id"]] += float(r["Counter_Value"]); grid[r["Dispatch_Id"]] = int(r["Grid_Size"])
big = max(grid.values())
v = [per[d] for d in per if grid[d] == big]
print("octave-0 launches:", len(v), "FETCH_SIZE KB mean", sum(v) / len(v), "-> HBM read bytes (x2 on gfx950)", 2048 * sum(v) / len(v))
s = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(s)):
    if "sift_extrema" in r["Name"] or "sift_blur" in r["Name"]:
        print(r["Name"].split("(")[0], r["Calls"], "calls, total us per SIFT call:", float(r["TotalDurationNs"]) / 21 / 1e3)
PY
