#!/bin/bash
# round-5 measurement pass on the final kernel sources: suite (+ durations), the gpu_long parity cases, counters / clocks / bench lines,
# every BASELINE configuration at full size
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu_final.txt 2>&1; tail -22 $O/pytest_gpu_final.txt
timeout 900 python -m pytest tests -m gpu_long -q --durations=8 > $O/pytest_gpu_long.txt 2>&1; tail -12 $O/pytest_gpu_long.txt
bash tools/probes/measure_counters.sh > $O/measure_counters.log 2>&1; tail -25 $O/measure_counters.log | cut -c1-300
timeout 3000 python tools/run_configs.py flux_sweep flux_cfg step1x_512 step1x_1024 qwen_1024 qwen_sweep step1x_v1p2_2048 step1x_v1p2_2048_50_fp8 --out $O/configs_full_size.json > $O/configs_full_size.log 2>&1
python - <<PY
import json
for r in json.load(open("$O/configs_full_size.json")):
    print(r.get("case","?")[:70], "K_e", r.get("K_e"), "regione", round(r.get("regione_edit_s",0),3), "full", round(r.get("full_token_edit_s",0),3), "x", round(r.get("speedup",0),3), r.get("step_ms_by_kind"), r.get("error"))
PY
