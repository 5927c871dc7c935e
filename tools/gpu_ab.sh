#!/bin/bash
# quick A/B: LK tests + short bench (8 distinct streams repeated)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 python -u -m pytest tests/test_klt_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_test.log 2>&1; echo test_exit=$?; tail -2 gpurun_out/ab_test.log
timeout -k 10 300 python -u bench.py --no-cpu-baseline --distinct 8 "$@" > gpurun_out/ab_bench.log 2>&1; echo bench_exit=$?
tail -1 gpurun_out/ab_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'tracked', d['config']['tracked_fraction'])"
