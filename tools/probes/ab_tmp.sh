for q in 1 2 0; do
  echo "== RGN_GEMM_QUARTER=$q (gate epilogue, cold W)"; RGN_GEMM_QUARTER=$q GEMM_COLD=1 GEMM_EPI=gate timeout 300 python tools/bench_kernels.py small 2>&1 | grep -E "R out|R qkv|R ff1|R ff2|R kvq|R proj" 
done
echo "== variants for R out"; GEMM_ONLY="R out" GEMM_VARIANTS=auto,1,2,3 GEMM_COLD=1 GEMM_EPI=gate timeout 300 python tools/bench_kernels.py small 2>&1 | grep "R out"
