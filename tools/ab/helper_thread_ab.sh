#!/bin/bash
# The SIFT route with and without the helper thread, and the new test: tools/ab/helper_thread_ab.sh OUTDIR
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 300 python3 -u -m pytest tests/test_pipeline_gpu.py -x -q --timeout 120 --timeout-method thread > "$OUT/pytest_pipeline.log" 2>&1 || { tail -30 "$OUT/pytest_pipeline.log"; exit 1; }
tail -2 "$OUT/pytest_pipeline.log"
for rep in 1 2; do
  for h in 1 0; do
    echo "# VO_HELPER_THREAD=$h" >> "$OUT/sift_route.txt"
    VO_HELPER_THREAD=$h timeout -k 10 200 python3 tools/run_vo.py --frames 60 --bootstrap sift | grep "^{" | cut -c1-330 >> "$OUT/sift_route.txt"
  done
done
VO_HELPER_THREAD=1 timeout -k 10 200 python3 tools/run_vo.py --frames 120 --bootstrap klt | grep "^{" | cut -c1-330 >> "$OUT/sift_route.txt"
VO_HELPER_THREAD=1 timeout -k 10 200 python3 tools/run_vo.py --frames 60 --bootstrap sift --profile | grep "^{" > "$OUT/sift_route_profile.txt"
cat "$OUT/sift_route.txt"
