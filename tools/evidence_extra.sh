#!/bin/bash
# Second evidence set for one build: the long seed sweeps, the per-operator bench, drop-in call latencies,
# the whole loop with both bootstrap routes, other points of the headline workload.  usage: tools/evidence_extra.sh <tag>
TAG=${1:-ev}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/$TAG
mkdir -p $O
run() { # name, seconds, command...
    local name=$1 secs=$2; shift 2
    timeout -k 10 $secs "$@" > $O/$name.log 2>&1 || { echo "$name failed"; tail -20 $O/$name.log; exit 1; }
    tail -1 $O/$name.log
}
run fuzz_parity 500 python -u tools/fuzz_parity.py --seeds 24
run ops 300 python -u tools/bench_ops.py
run sift 120 python -u tools/bench_sift.py
run dropin 200 python -u tools/bench_dropin.py
run vo_sift 200 python -u tools/run_vo.py --frames 60 --bootstrap sift --profile
run vo_klt 200 python -u tools/run_vo.py --frames 60 --bootstrap klt --profile
run vo_klt_separate 200 python -u tools/run_vo.py --frames 60 --bootstrap klt --separate-calls
run frame 200 python -u tools/bench_frame.py
run fuzz_geometry 300 python -u tools/fuzz_geometry.py --seeds 24
run fuzz_ba 300 python -u tools/fuzz_ba.py --seeds 20
run fuzz_klt 400 python -u tools/fuzz_klt.py --seeds 60
run bench_ring16 300 python -u bench.py --ring 16 --no-cpu-baseline
run bench_groups1 300 python -u bench.py --groups 1 --no-cpu-baseline
run bench_level8 300 python -u bench.py --max-level 8 --no-cpu-baseline
echo extra evidence done
