R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_32
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_prior_gpu.py tests/test_lo8_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 200 python scripts/bench_kernels.py attn 2>&1 | grep "^prior_attn" | tee $O/attn_new.log
