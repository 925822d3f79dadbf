#!/bin/bash
# round-5 first GPU pass: counter inventory (is there a MALL / HBM-side counter?), vendor attention stick, GPU-suite durations
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -i -E "mall|hbm|dram|EA0?_|umc|hit|miss" $O/counters_list.txt | head -150 > $O/counters_memory_side.txt
cd $R
python tools/probes/vendor_attn.py > $O/vendor_attn.json 2> $O/vendor_attn.err
for k in vendor_attn vendor_attn_region attn attn_region; do python tools/clock_probe.py $k --json 2>/dev/null | tail -1 > $O/clock_$k.json; done
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > $O/pytest_gpu_durations.txt 2>&1
tail -5 $O/pytest_gpu_durations.txt
cat $O/vendor_attn.json | head -80
