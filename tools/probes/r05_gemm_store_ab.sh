#!/bin/bash
# HISTORICAL (ran at commit a185b89; output = profiles/r05_gemm_store_loop_ab.txt): the `gemm_persistent` knob and the old-store-loop
# library build it compares no longer exist (the persistent loop measured neutral and was removed in b06e038).  Kept as the record of
# how the A/B was made; to re-run, check out a185b89.
# round-5 same-box A/B of the GEMM epilogue's store loop and the persistent tile loop:
#   old   = the round-4 store loop (join-point `s_waitcnt vmcnt(0)` in front of every C store), one workgroup per tile
#   fast  = load-free store loop instance, one workgroup per tile          (RGN_PLAN_OVERRIDE=gemm_persistent=0)
#   pers  = load-free store loop + persistent tile loop (the shipped default)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
OLD=$R/tools/probes/bin/libregione_hip_oldstore.so
run() { # label, env...
  local label=$1; shift
  env "$@" GEMM_COLD=1 python tools/bench_kernels.py gemm 2>/dev/null | sed "s/^/[$label] /"
}
{
for rep in 1 2; do
  run old RGN_LIB=$OLD RGN_PLAN_OVERRIDE=gemm_persistent=0
  run fast RGN_PLAN_OVERRIDE=gemm_persistent=0
  run pers RGN_PLAN_OVERRIDE=
done
} > $O/gemm_store_ab_kernels.txt 2>&1
grep -E "kvq\+mlp|proj_out|ff1|img out" $O/gemm_store_ab_kernels.txt | sort | head -60
# epilogue variants (bias / GELU / fused Q/K/V) isolated
for v in old fast pers; do
  case $v in old) E="RGN_LIB=$OLD RGN_PLAN_OVERRIDE=gemm_persistent=0";; fast) E="RGN_PLAN_OVERRIDE=gemm_persistent=0";; pers) E="RGN_PLAN_OVERRIDE=";; esac
  env $E python tools/bench_epilogues.py 2>/dev/null | sed "s/^/[$v] /"
done > $O/gemm_store_ab_epilogues.txt 2>&1
cat $O/gemm_store_ab_epilogues.txt | head -40
# the whole edit, two runs each, interleaved
for rep in 1 2; do
  for v in old fast pers; do
    case $v in old) E="RGN_LIB=$OLD RGN_PLAN_OVERRIDE=gemm_persistent=0";; fast) E="RGN_PLAN_OVERRIDE=gemm_persistent=0";; pers) E="RGN_PLAN_OVERRIDE=";; esac
    env $E python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-5pct --no-vanilla 2>/dev/null > $O/gemm_store_ab_bench_${v}_$rep.json
    python - <<PY
import json
d=json.load(open("$O/gemm_store_ab_bench_${v}_$rep.json"))
print("$v $rep", round(d["value"],3), "steps/s  F", round(d["step_ms_by_kind"]["F"]["avg_ms"],2), "R", round(d["step_ms_by_kind"]["R"]["avg_ms"],2), "gemm frac", round(d["roofline"]["frac"],4), "attn frac", round(d["roofline_attention"]["frac"],4))
PY
  done
done | tee $O/gemm_store_ab_bench.txt
