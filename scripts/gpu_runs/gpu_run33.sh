mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py -q --tb=short -x -p no:cacheprovider -k skinny > gpurun_out/tests33.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests33.log | tail -2; grep -E "^E  " gpurun_out/tests33.log | cut -c1-300 | head -20
for kw in 4 8 16 0; do echo "LLARK_SKINNY_KW=$kw"; LLARK_SKINNY_KW=$kw LLARK_DECODE_GRAPH=0 timeout 600 python scripts/bench_kernels.py decode 2>&1 | grep "eager"; done
