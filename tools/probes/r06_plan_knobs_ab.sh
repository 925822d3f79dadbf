#!/bin/bash
# same-box A/B of the launch-plan knobs on the bench workload (K_e = 1024): does any forced schedule beat the planners' choices in the region steps?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06
for knob in "" "attn_streamk=0" "attn_split=0" "gemm_quarter=0" "gemm_pieces=1" "gemm_quarter=1"; do
  RGN_PLAN_OVERRIDE="$knob" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla --no-5pct 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d['step_ms_by_kind']
print('%-18s steps/s %.2f  F %.2f ms  R %.2f ms (min %.2f)' % ('$knob' or 'planner (shipped)', d['value'], s['F']['avg_ms'], s['R']['avg_ms'], s['R']['min_ms']))"
done
