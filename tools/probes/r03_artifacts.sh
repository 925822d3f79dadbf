set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla > $O/bench_rocprof_run.json 2> $O/kt.err
ls -R $O/kt | head -20
DB=$(find $O/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla" > $O/kernel_stats.txt
head -12 $O/kernel_stats.txt
RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rd -o rd -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla > $O/pmc_rd.json 2> $O/rd.err
RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla > $O/pmc_wr.json 2> $O/wr.err
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $(find $O/rd -name "*.db" | head -1) $(find $O/wr -name "*.db" | head -1) > $O/pmc_traffic.json
cat $O/pmc_traffic.json | head -30
# the databases are large: keep the summaries only
rm -rf $O/kt $O/rd $O/wr
