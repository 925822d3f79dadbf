#!/bin/bash
# re-run of the two traffic counter passes of measure_counters.sh (bench.py's characterising leg is TWO instrumented edits since round 6:
# 3 edits per process, not 2) + the default bench lines on the final profiles/ files
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prd /tmp/pwr /tmp/plat
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla --no-5pct"
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d /tmp/prd -o rd -- $BENCH > $O/pmc/rd_bench.json 2> $O/pmc/rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pwr -o wr -- $BENCH > $O/pmc/wr_bench.json 2> $O/pmc/wr.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d /tmp/plat -o lat -- $BENCH > $O/pmc/lat_bench.json 2> $O/pmc/lat.err
cd $R
LAT=$(find /tmp/plat -name "lat_results.db" | head -1)
python tools/pmc_traffic.py $(find /tmp/prd -name "rd_results.db" | head -1) $(find /tmp/pwr -name "wr_results.db" | head -1) --edits 3 ${LAT:+--lat-db $LAT} > $O/pmc_traffic.json
cp $O/pmc_traffic.json $R/profiles/r06_pmc_traffic.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2>/dev/null
tail -c 400 $O/bench_default.json
