#!/bin/bash
# Frame-mode benchmark under different schedules: stream priorities, staged first chunk or not,
# launch groups.  usage: tools/ab/frame_sched.sh <out.jsonl>
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/frame_sched.jsonl}
: > $OUT
run() {
    echo "# ${ENVV[*]} $*" >> $OUT
    env "${ENVV[@]}" timeout -k 10 200 python bench.py --mode frame --steps 400 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> $OUT || exit 1
}
ENVV=(X=1); run
ENVV=(X=1); run --priorities=0,0
ENVV=(X=1); run --priorities=-1,-1
ENVV=(VO_FRAME_STAGE1=0); run
ENVV=(X=1); run --groups 3
ENVV=(X=1); run --groups 4
ENVV=(X=1); run --groups 1
python - <<PY
import json
for l in open("$OUT"):
    if l.startswith('#'): print(l.strip(), end=' -> ')
    else:
        d=json.loads(l); print(round(d['value']), 'frames/s', round(d['ms_per_step'],4), 'ms/step; klt launch', round(d['roofline']['launch_ms'],4), 'in flight', round(d['roofline']['launches_in_flight'],2), d['config']['pose_check']['max_rotation_error_deg'])
PY
