export TMPDIR=/tmp
timeout 300 python scripts/bench_gemm.py 11,12,100,101 2>&1 | grep -E "^bf16"
