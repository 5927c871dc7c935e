#!/bin/bash
# Interleaved repetitions of the loop with and without the helper threads: helper_thread_reps.sh OUTFILE [reps] [route] [frames]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m "$1"); : > "$OUT"
ROUTE=${3:-sift}; FRAMES=${4:-60}
cd "$REPO"
for rep in $(seq ${2:-5}); do
  for h in 1 0; do
    VO_HELPER_THREAD=$h timeout -k 10 200 python3 tools/run_vo.py --frames $FRAMES --bootstrap $ROUTE | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$ROUTE $FRAMES helper=$h', round(d['loop_frames_per_s'],1), round(d['median_ms_per_tracked_frame'],3), round(d['bootstrap_ms'],1))" >> "$OUT" || exit 1
  done
done
cat "$OUT"
