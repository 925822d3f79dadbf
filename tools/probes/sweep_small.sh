cd $GRAFT_REPO_ROOT
for ns in 0 1 2 3 4 5 6 8; do echo "== NSPLIT $ns (variant 3)"; RGN_GEMM_NSPLIT=$ns GEMM_VARIANTS=3 python tools/bench_kernels.py small 2>&1 | grep -v amdgpu.ids; done
echo "== auto"; python tools/bench_kernels.py small 2>&1 | grep -v amdgpu.ids
echo "== auto variant 1"; GEMM_VARIANTS=1 python tools/bench_kernels.py small 2>&1 | grep -v amdgpu.ids
echo "== gated auto"; GEMM_EPI=gate python tools/bench_kernels.py small 2>&1 | grep -v amdgpu.ids
ATTN_MORE=1 python tools/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids
