# round-3 artifacts, part 2: kernel trace of 5 edits (no K_e = 5 % leg), full-size runs of the other BASELINE configurations,
# full-dimension parity log
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
cd /tmp
RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla > $O/bench_rocprof_run.json 2> $O/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) "RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla: 5 RegionE edits (1 warm-up + 3 timed + 1 characterising)" > $O/kernel_stats.txt
rm -rf $O/kt
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_full_dims.py tests/test_adapters.py -m gpu -q -s 2>&1 | grep -E "parity\]|x-check\]|passed|failed" > $O/full_dims_parity.txt
python tools/run_configs.py flux_sweep flux_cfg step1x_512 qwen_1024 qwen_sweep step1x_v1p2_2048 --out $O/configs_full_size.json > $O/configs.log 2>&1
tail -3 $O/configs.log | cut -c1-300
