#!/bin/bash
# two SQ counter passes over a short bench run
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for CTRS in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$REPO/bench.py" --no-cpu-baseline --distinct 8 --steps 3 --warmup 1 "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
done
python3 $REPO/tools/pmc_summary.py $OUT | python3 -c "
import sys,json
d=json.load(sys.stdin)
k=d.get('klt_track3_kernel') or d['klt_track_kernel']; w=k['SQ_WAVES']
for c in sorted(k): print(c, round(k[c]/w,1) if c!='dispatches' else k[c])
"
