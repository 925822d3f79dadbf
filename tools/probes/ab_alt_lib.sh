#!/bin/bash
# same-box A/B of two builds of the kernel library: regione_amd/lib/libregione_hip_alt.so (RGN_LIB) against the in-tree build
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "${AB_TESTS:-ln_modulate}" 2>&1 | tail -2
for lib in alt main alt main; do
  if [ $lib = alt ]; then export RGN_LIB=$GRAFT_REPO_ROOT/regione_amd/lib/libregione_hip_alt.so; else unset RGN_LIB; fi
  echo $lib
  [ -n "$AB_PROBE" ] && python $AB_PROBE 2>&1 | grep -v amdgpu
  python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), round(d['roofline']['frac'],4), {k:round(v['avg_ms'],2) for k,v in d['step_ms_by_kind'].items()}, 'full-token', round(d['full_token']['edit_wall_clock_s'],4))"
done
