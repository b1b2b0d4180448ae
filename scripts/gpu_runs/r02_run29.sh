R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_29
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_lo8_gpu.py tests/test_prior_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | sed 's/split f16 //' | cut -c1-120 | tee -a $O/overlap.log; done
