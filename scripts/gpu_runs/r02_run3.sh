# round 2, GPU run 3: lo8 split GEMM -- parity tests, A/B timing against the f16x2 kernels, tiny/full-width prior
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_3
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -s -p no:cacheprovider > $O/t_lo8.log 2>&1; echo "lo8 tests exit $?"; grep -E "lo8 gemm|passed|failed|Error|error|rel err|assert" $O/t_lo8.log | cut -c1-300 | tail -30
timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -p no:cacheprovider -k "gemm256 or persistent or tile_variants or layernorm" > $O/t_prior_subset.log 2>&1; echo "prior subset exit $?"; tail -3 $O/t_prior_subset.log
timeout 400 python scripts/bench_gemm256.py 30,40 > $O/bench_gemm_lo8.log 2>&1; echo "bench exit $?"; grep "split f16\|lo8" $O/bench_gemm_lo8.log
