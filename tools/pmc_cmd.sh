#!/bin/bash
# Two SQ counter passes and a kernel trace over a python script of this repository:
#   tools/pmc_cmd.sh OUTDIR KERNEL_SUBSTRING script.py [args...]
# prints per-wave means of the counters for the kernels whose name contains KERNEL_SUBSTRING.
OUT=$(realpath -m "$1"); KER="$2"; shift 2
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SCRIPT=$(realpath "$1"); shift
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 "$SCRIPT" "$@" > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
done
python3 $REPO/tools/pmc_summary.py $OUT | KER="$KER" python3 -c "
import sys,json,os
d=json.load(sys.stdin)
for name,k in d.items():
    if os.environ['KER'] not in name: continue
    w=k['SQ_WAVES']
    print(name, 'dispatches', k['dispatches'], 'waves', w)
    for c in sorted(k):
        if c not in ('dispatches','SQ_WAVES'): print('  ', c, round(k[c]/w,1))
"
