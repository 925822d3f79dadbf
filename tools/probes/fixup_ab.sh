cd $GRAFT_REPO_ROOT
export GEMM_COLD=1 GEMM_EPI=gate
for ns in 0 5 6 7 8; do
  echo "== NSPLIT=$ns V1"
  env RGN_GEMM_NSPLIT=$ns python tools/bench_kernels.py small 2>&1 | grep -E "gemm|pair"
done
echo "== NSPLIT=0 V0"; RGN_GEMM_ASM_SPLIT=0 python tools/bench_kernels.py small 2>&1 | grep -E "gemm|pair"
