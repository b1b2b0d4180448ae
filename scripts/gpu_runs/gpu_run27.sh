mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/bench_gemm.py 12,20 2>&1 | grep -v "^{" | tee gpurun_out/bench_gemm6.log
