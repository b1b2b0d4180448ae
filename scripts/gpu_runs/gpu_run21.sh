mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/bench_gemm.py 12,11,100 2>&1 | grep -v "^{" | tee gpurun_out/bench_gemm5.log
