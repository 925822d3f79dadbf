set -x
mkdir -p gpurun_out/r6a
python tools/parity_full_depth.py --cases flux_headline --out gpurun_out/r6a/parity_headline.json --save-fixture gpurun_out/r6a/headline_flux.npz > gpurun_out/r6a/parity_headline.log 2>&1
cp gpurun_out/r6a/headline_flux.npz tests/golden/headline_flux.npz
python -m pytest tests -m gpu -q -x --durations=40 > gpurun_out/r6a/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r6a/pytest_gpu.txt
PROFILE_FIRST_R=1 python tools/host_overhead.py 0.25 > gpurun_out/r6a/host_overhead.txt 2>&1
tail -40 gpurun_out/r6a/host_overhead.txt
