mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vqvae_gpu.py tests/test_extract_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests14.log 2>&1; echo "tests exit $?"
timeout 300 python scripts/bench_kernels.py res > gpurun_out/bench_res_mfma.log 2>&1
LLARK_RESBLOCK_VALU=1 timeout 300 python scripts/bench_kernels.py res > gpurun_out/bench_res_valu.log 2>&1
grep -E "passed|failed" gpurun_out/tests14.log | tail -2; grep -E "^E  " gpurun_out/tests14.log | cut -c1-300 | head -6; echo MFMA; grep resblock gpurun_out/bench_res_mfma.log; echo VALU; grep resblock gpurun_out/bench_res_valu.log
