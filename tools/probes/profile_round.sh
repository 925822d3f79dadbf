# Round profile: kernel trace + the two PMC passes + summaries (GPU box).  Usage: bash tools/probes/profile_round.sh r02
set -x
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla > $O/bench_rocprof_run.json 2> $O/kt.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rd -o rd -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla > /dev/null 2> $O/rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o wr -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vanilla > /dev/null 2> $O/wr.err
cd $R
KT=$(find $O/kt -name "*results.db" | head -1); RD=$(find $O/rd -name "*results.db" | head -1); WR=$(find $O/wr -name "*results.db" | head -1)
python tools/rocpd_summary.py $KT "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla: 5 RegionE edits (1 characterising + 1 warm-up + 3 timed)" > $O/kernel_stats.txt
python tools/pmc_traffic.py $RD $WR > $O/pmc_traffic.json
find $O -name "*stats*.csv" | head; 
rm -rf $O/rd $O/wr; find $O/kt -name "*.db" -size +40M -delete
head -30 $O/kernel_stats.txt; cat $O/pmc_traffic.json | head -20; tail -3 $O/kt.err
