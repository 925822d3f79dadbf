#!/bin/bash
# round-5 second GPU pass: the refactored suite, the f4 host-side timing, what the SMI tools expose about HBM activity, the parity one-offs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu_pass2.txt 2>&1
tail -30 $O/pytest_gpu_pass2.txt
python tools/f4_host_side.py > $O/f4_host_side.json 2> $O/f4_host_side.err; tail -5 $O/f4_host_side.err; cat $O/f4_host_side.json
{ rocm-smi --showmemuse --showuse; echo ----; amd-smi metric --usage 2>&1 | head -60; echo ----; amd-smi metric --help 2>&1 | head -60; } > $O/smi_mem_activity.txt 2>&1
timeout 2400 python tools/parity_full_depth.py --cases flux_headline --out $O/parity_headline.json > $O/parity_headline.log 2>&1; tail -15 $O/parity_headline.log
timeout 2400 python tools/parity_full_depth.py --cases flux_fullloop,qwen_fullloop --out $O/parity_fullloops.json > $O/parity_fullloops.log 2>&1; tail -12 $O/parity_fullloops.log
