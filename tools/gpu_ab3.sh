#!/bin/bash
# A/B of the two tracker kernels on one box: LK tests with each, then short benches
cd ${GRAFT_REPO_ROOT:-/root/repo}
export VO_KLT_VARIANT=3
timeout -k 10 300 python -u -m pytest tests/test_klt_gpu.py tests/test_dropin_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab3_test.log 2>&1; echo test3_exit=$?; tail -15 gpurun_out/ab3_test.log
for v in 3 1 3 1; do
  export VO_KLT_VARIANT=$v
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --distinct 8 "$@" > gpurun_out/ab3_bench_$v.log 2>&1 || { echo bench failed; tail -5 gpurun_out/ab3_bench_$v.log; exit 1; }
  tail -1 gpurun_out/ab3_bench_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v fps', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'tracked', d['config']['tracked_fraction'])"
done
