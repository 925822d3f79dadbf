set -x
mkdir -p gpurun_out/r04/pmc
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
for k in gemm attn vendor edit; do python tools/clock_probe.py $k --json 2>/dev/null | tail -1 > $O/clock_$k.json; done
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pm -o m -- $BENCH > $O/pmc/mfma_bench.json 2> $O/pmc/mfma.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d /tmp/prd -o rd -- $BENCH > $O/pmc/rd_bench.json 2> $O/pmc/rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pwr -o wr -- $BENCH > $O/pmc/wr_bench.json 2> $O/pmc/wr.err
cd $R
python tools/pmc_summary.py $(find /tmp/pm -name "m_results.db" | head -1) > $O/pmc_mfma.json
python tools/pmc_traffic.py $(find /tmp/prd -name "rd_results.db" | head -1) $(find /tmp/pwr -name "wr_results.db" | head -1) > $O/pmc_traffic.json
head -c 600 $O/pmc_traffic.json; python -c "
import json; d=json.load(open('$O/pmc_mfma.json')); print({k:v.get('mfma_util') for k,v in d.items() if isinstance(v,dict)})"
cat $O/clock_*.json
