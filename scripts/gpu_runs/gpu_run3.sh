mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.txt
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_prior_gpu.py -m gpu -q --tb=short -rA -p no:cacheprovider > gpurun_out/tests3.log 2>&1; echo "tests exit $?" >> gpurun_out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 900 python scripts/bench_gemm.py > gpurun_out/bench_gemm.log 2>&1; echo "bench_gemm exit $?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt; grep -E "passed|failed" gpurun_out/tests3.log | tail -3; grep -v "^{" gpurun_out/bench_gemm.log | tail -60
