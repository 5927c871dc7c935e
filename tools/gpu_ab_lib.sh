#!/bin/bash
# A/B of two builds of the library on one box: the in-tree libvo_hip.so against tools/ab/libvo_hip_base.so
# (alternating runs, same process layout); extra arguments go to bench.py
cd ${GRAFT_REPO_ROOT:-/root/repo}
BASE=$PWD/tools/ab/libvo_hip_base.so
for v in new base new base; do
  if [ $v = base ]; then export VO_HIP_LIB=$BASE; else unset VO_HIP_LIB; fi
  timeout -k 10 300 python -u bench.py --no-cpu-baseline "$@" > gpurun_out/abl_$v.log 2>&1 || { echo bench failed; tail -5 gpurun_out/abl_$v.log; exit 1; }
  tail -1 gpurun_out/abl_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v fps', round(d['value']), 'launch_ms', round(d['roofline']['launch_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))"
done
