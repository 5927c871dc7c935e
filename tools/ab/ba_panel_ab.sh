#!/bin/bash
# Step times and kernel durations of the bundle-adjustment trial step: tools/ab/ba_panel_ab.sh OUTDIR
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 200 python3 -u -m pytest tests/test_ba_solver_gpu.py tests/test_ba_reference_window_gpu.py -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_ba.log" 2>&1
tail -2 "$OUT/pytest_ba.log"
timeout -k 10 200 python3 tools/ab/ba_window_step.py > "$OUT/ba_step.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/ab/ba_window_step.py" 4 > /dev/null 2>&1
python3 "$REPO/tools/ab/kernel_times.py" "$OUT/trace" ba_ >> "$OUT/ba_step.txt"
cat "$OUT/ba_step.txt"
