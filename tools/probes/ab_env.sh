#!/bin/bash
# same-box A/B of an environment switch: AB_VAR=NAME AB_VALUES="0 1" [AB_TESTS=expr] bash tools/probes/ab_env.sh
cd $GRAFT_REPO_ROOT
for v in $AB_VALUES $AB_VALUES; do
  export $AB_VAR=$v; echo "$AB_VAR=$v"
  [ -n "$AB_TESTS" ] && python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "$AB_TESTS" 2>&1 | tail -1
  python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), round(d['roofline']['frac'],4), {k:round(v['avg_ms'],2) for k,v in d['step_ms_by_kind'].items()}, 'full-token', round(d['full_token']['edit_wall_clock_s'],4), 'psnr', d.get('parity_smoke_db', d.get('latent_psnr_vs_full_token_random_weights_db')))"
done
