#!/bin/bash
# Full evidence set for one build: GPU test suite, default bench line, kernel-trace
# stats of the same bench command, PMC passes.  usage: tools/evidence.sh <tag>
TAG=${1:-ev}
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=$R/gpurun_out/$TAG
rm -rf $O
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { echo tests failed; tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python -u bench.py > $O/bench.log 2>&1 || { echo bench failed; tail -20 $O/bench.log; exit 1; }
tail -1 $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1 || { echo stats failed; tail -20 $O/stats.log; exit 1; }
cd $R
# counters are device-wide: one stream, so that no other kernel runs beside the one being counted
bash tools/pmc_passes.sh $O/pmc --steps 3 --warmup 1 --settle-ms 0 --no-pipeline || exit 1
echo evidence done
