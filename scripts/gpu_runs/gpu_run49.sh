mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vqvae_gpu.py tests/test_extract_gpu.py -q --tb=short -x -p no:cacheprovider > gpurun_out/tests49.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests49.log | tail -2; grep -E "^E  |Error" gpurun_out/tests49.log | cut -c1-300 | head -20
timeout 300 python scripts/bench_kernels.py vqvae 2>&1 | grep -E "VQ-VAE"
