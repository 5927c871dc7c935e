#!/bin/bash
# The whole loop on both bootstrap routes, 60 and 120 frames, keyframe overlap on and off, then with per-part timers:
# tools/evidence_loop.sh OUTFILE
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m "$1"); mkdir -p "$(dirname "$OUT")"; : > "$OUT"
cd "$REPO"
for route in klt sift; do
  for frames in 60 120; do
    for flag in "" "--no-overlap"; do
      timeout -k 10 200 python3 tools/run_vo.py --frames $frames --bootstrap $route $flag | grep "^{" >> "$OUT" || exit 1
    done
  done
done
for route in klt sift; do
  VO_HELPER_THREAD=0 timeout -k 10 200 python3 tools/run_vo.py --frames 60 --bootstrap $route | grep "^{" | sed 's/^{/{"helper_thread": false, /' >> "$OUT" || exit 1
  timeout -k 10 200 python3 tools/run_vo.py --frames 120 --bootstrap $route --profile | grep "^{" >> "$OUT" || exit 1
done
cut -c1-260 "$OUT"
