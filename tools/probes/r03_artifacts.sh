# round-3 artifacts: the default bench line, a kernel trace of 5 edits, two PMC passes -> gpurun_out/r03/ (copied to profiles/r03_*)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
cd /tmp
RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rd -o rd -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla > $O/pmc_rd.json 2> $O/rd.err
RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla > $O/pmc_wr.json 2> $O/wr.err
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $(find $O/rd -name "*.db" | head -1) $(find $O/wr -name "*.db" | head -1) > $O/pmc_traffic.json
cp $O/pmc_traffic.json $GRAFT_REPO_ROOT/profiles/r03_pmc_traffic.json      # bench.py quotes it when the csrc stamp matches
rm -rf $O/rd $O/wr
RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla > $O/bench_rocprof_run.json 2> $O/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) "RGN_BENCH_NO_5PCT=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla: 5 RegionE edits (1 warm-up + 3 timed + 1 characterising)" > $O/kernel_stats.txt
rm -rf $O/kt
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err
head -c 600 $O/bench_default.json
