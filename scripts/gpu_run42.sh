export TMPDIR=/tmp
timeout 300 python scripts/bench_gemm.py 1,11,12,100 2>&1 | grep -E "^split"
