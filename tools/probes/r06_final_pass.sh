#!/bin/bash
# round-6 measurement pass on the final kernel sources: counters / clocks / bench lines, parity reports, every BASELINE configuration at full
# size, region-step traces, the VAE kernels, the -m gpu suite with durations.  Everything lands in gpurun_out/r06/ (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
export RND=r06
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
STAGE=${1:-all}
if [ "$STAGE" = "all" ] || [ "$STAGE" = "counters" ]; then
bash tools/probes/measure_counters.sh > $O/measure_counters.log 2>&1; tail -3 $O/measure_counters.log | cut -c1-300
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "tests" ]; then
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.txt 2>&1; tail -32 $O/pytest_gpu.txt
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "parity" ]; then
python tools/parity_full_depth.py --cases flux_headline --out $O/parity_headline.json > $O/parity_headline.log 2>&1; grep "full-depth parity" $O/parity_headline.log | tail -8
python tools/parity_full_depth.py --cases flux_loop,qwen_loop,step1x_v1p2_loop,flux_width,qwen_width,qwen_fullloop --out $O/parity_full_depth.json > $O/parity_full_depth.log 2>&1; grep "full-depth parity" $O/parity_full_depth.log | tail -40
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "configs" ]; then
timeout 3000 python tools/run_configs.py flux_sweep flux_cfg step1x_512 step1x_1024 qwen_1024 qwen_sweep step1x_v1p2_2048 step1x_v1p2_2048_50_fp8 --out $O/configs_full_size.json > $O/configs_full_size.log 2>&1
python - <<PY
import json
for r in json.load(open("$O/configs_full_size.json")):
    print(r.get("case","?")[:70], "K_e", r.get("K_e"), "regione", round(r.get("regione_edit_s",0),3), "full", round(r.get("full_token_edit_s",0),3), "x", round(r.get("speedup",0),3), {k: v.get("avg_ms") for k, v in (r.get("step_ms_by_kind") or {}).items()}, r.get("error"))
PY
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "traces" ]; then
python tools/f4_host_side.py > $O/f4_host_side.json 2> $O/f4_host_side.err; tail -12 $O/f4_host_side.json
python tools/vae_decode_bench.py --reps 10 > $O/vae_decode_bench.json 2>/dev/null; cat $O/vae_decode_bench.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/vp /tmp/rt
rocprofv3 --kernel-trace -d /tmp/vp -o vae -- python $R/tools/vae_decode_bench.py --reps 5 --no-eager > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace of tools/vae_decode_bench.py --reps 5 --no-eager (8 decodes of a 128 x 128 x 16 latent -> 1024 x 1024)"; python $R/tools/probes/kernel_avg.py $(find /tmp/vp -name "vae_results.db" | head -1); } > $O/vae_kernel_stats.txt
for fam in flux qwen; do
rm -rf /tmp/rt_$fam
rocprofv3 --kernel-trace -d /tmp/rt_$fam -o rt -- python $R/tools/probes/region_step_trace.py run $fam 6 > $O/region_step_trace_$fam.log 2>&1
python $R/tools/probes/region_step_trace.py report $(find /tmp/rt_$fam -name "rt_results.db" | head -1) > $O/region_step_trace_$fam.txt 2>&1
done
cd $R
head -5 $O/region_step_trace_flux.txt | cut -c1-200
fi
