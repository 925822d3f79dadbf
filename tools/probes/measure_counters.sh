#!/bin/bash
# Measurement pass (round = $RND, default r06) on one MI355X box (everything the bench line's roofline / mfma_busy / traffic fields and DESIGN section 5
# quote): sustained clock / power probes, rocprofv3 --pmc passes (own runs, --kernel-trace only), a kernel-trace summary and the default
# bench line, all from the SAME kernel sources (csrc_sha16 stamp).  Writes gpurun_out/r05/; the summaries are copied to profiles/.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=${RND:-r06}
O=$R/gpurun_out/$RND
mkdir -p $O/pmc
cd $R
for k in gemm attn vendor vendor_attn gemv edit; do python tools/clock_probe.py $k --json 2>/dev/null | tail -1 > $O/clock_$k.json; done
python -c "
import json
print(json.dumps({k: json.load(open('$O/clock_%s.json' % k)) for k in ('gemm', 'attn', 'vendor', 'vendor_attn', 'gemv', 'edit')}, indent=1))" > $O/clock_probe.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm /tmp/prd /tmp/pwr /tmp/kt /tmp/plat
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla --no-5pct"          # 2 edits: 1 timed + 1 characterising
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pm -o m -- $BENCH > $O/pmc/mfma_bench.json 2> $O/pmc/mfma.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d /tmp/prd -o rd -- $BENCH > $O/pmc/rd_bench.json 2> $O/pmc/rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pwr -o wr -- $BENCH > $O/pmc/wr_bench.json 2> $O/pmc/wr.err
# the closest thing to an HBM-side view this image offers (no MALL / UMC counter): average L2->fabric read latency per kernel family
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d /tmp/plat -o lat -- $BENCH > $O/pmc/lat_bench.json 2> $O/pmc/lat.err
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla --no-5pct > $O/bench_rocprof_run.json 2> $O/pmc/kt.err
cd $R
python tools/pmc_summary.py $(find /tmp/pm -name "m_results.db" | head -1) > $O/pmc_mfma.json
LAT=$(find /tmp/plat -name "lat_results.db" | head -1)
python tools/pmc_traffic.py $(find /tmp/prd -name "rd_results.db" | head -1) $(find /tmp/pwr -name "wr_results.db" | head -1) --edits 3 ${LAT:+--lat-db $LAT} > $O/pmc_traffic.json
{ echo "# rocprofv3 --kernel-trace of: python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla --no-5pct (5 RegionE edits: 1 warm-up, 1 characterising, 3 timed); bench line of the same process: bench_rocprof_run.json"; python tools/probes/kernel_avg.py $(find /tmp/kt -name "kt_results.db" | head -1); } > $O/kernel_stats.txt
# the bench line quotes traffic / mfma_busy only from profiles/ files of the SAME kernel sources: put this pass's summaries there first
cp $O/pmc_mfma.json $R/profiles/${RND}_pmc_mfma.json; cp $O/pmc_traffic.json $R/profiles/${RND}_pmc_traffic.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2>/dev/null
tail -c 600 $O/bench_default.json; head -14 $O/kernel_stats.txt | cut -c1-160
