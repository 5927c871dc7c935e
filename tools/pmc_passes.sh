#!/bin/bash
# PMC passes over bench.py (one counter set per run; kernel-trace only).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for CTRS in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
  echo "pass $i done"
done
