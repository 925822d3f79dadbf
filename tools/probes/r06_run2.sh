set -x
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_full_depth.py tests/test_hosted_pipelines.py -m gpu -q -x -s -k "28_steps or hosted or stock" > gpurun_out/r6c/pytest_depth_hosted.txt 2>&1
tail -25 gpurun_out/r6c/pytest_depth_hosted.txt
python bench.py > gpurun_out/r6c/bench_default.json 2> gpurun_out/r6c/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6c/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','step_ms_by_kind','end_to_end','speedup_vs_full_token')})
print(d['roofline'].get('frac'), d.get('roofline_attention',{}).get('frac'))
PY
