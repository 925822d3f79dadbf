#!/bin/bash
# round-5 measurement pass on the final kernel sources: counters / clocks / bench lines, the gpu_long parity cases, every BASELINE
# configuration at full size, the -m gpu suite with durations
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
bash tools/probes/measure_counters.sh > $O/measure_counters.log 2>&1; tail -3 $O/measure_counters.log | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu_final.txt 2>&1; tail -20 $O/pytest_gpu_final.txt
if [ "$1" != "quick" ]; then
timeout 900 python -m pytest tests -m gpu_long -q --durations=8 > $O/pytest_gpu_long.txt 2>&1; tail -10 $O/pytest_gpu_long.txt
timeout 3000 python tools/run_configs.py flux_sweep flux_cfg step1x_512 step1x_1024 qwen_1024 qwen_sweep step1x_v1p2_2048 step1x_v1p2_2048_50_fp8 --out $O/configs_full_size.json > $O/configs_full_size.log 2>&1
python - <<PY
import json
for r in json.load(open("$O/configs_full_size.json")):
    print(r.get("case","?")[:70], "K_e", r.get("K_e"), "regione", round(r.get("regione_edit_s",0),3), "full", round(r.get("full_token_edit_s",0),3), "x", round(r.get("speedup",0),3), {k: v.get("avg_ms") for k, v in (r.get("step_ms_by_kind") or {}).items()}, r.get("error"))
PY
fi
