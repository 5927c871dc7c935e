#!/bin/bash
# Evidence of the batched tracked frame (bench.py --mode frame): bench line, kernel statistics of the same command, PMC
# passes on one stream / one launch group, profiles/frame_chain.json.  usage: tools/evidence_frame.sh <tag>
TAG=${1:-frame}
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 300 python -u bench.py --mode frame > $O/bench_frame.log 2>&1 || { echo bench failed; tail -20 $O/bench_frame.log; exit 1; }
tail -1 $O/bench_frame.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --mode frame --no-cpu-baseline > $O/stats.log 2>&1 || { echo stats failed; tail -20 $O/stats.log; exit 1; }
cd $R
bash tools/pmc_passes.sh $O/pmc --mode frame --steps 3 --warmup 1 --settle-ms 0 --no-pipeline --groups 1 || exit 1
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.json || exit 1
python tools/make_frame_chain.py $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/pmc_summary.json 64 100 \
  "rocprofv3 --kernel-trace --stats -- python3 bench.py --mode frame --no-cpu-baseline; counters: tools/pmc_passes.sh <dir> --mode frame --steps 3 --warmup 1 --settle-ms 0 --no-pipeline --groups 1" > $O/frame_chain.json || exit 1
echo frame evidence done
