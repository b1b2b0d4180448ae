export TMPDIR=/tmp
timeout 300 python scripts/bench_gemm.py 12,13,20,21 2>&1 | grep -E "^split|^variant"
