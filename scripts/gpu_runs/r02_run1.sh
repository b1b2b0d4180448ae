# round 2, GPU run 1: the new 256x256 ring GEMM (parity + A/B timing), full-depth parity fixtures, e2e bench
O=gpurun_out/r02_1
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prior_gpu.py -x -q -k "gemm256" > $O/t_gemm256.log 2>&1; echo "gemm256 tests exit $?"; tail -5 $O/t_gemm256.log
timeout 400 python scripts/bench_gemm256.py 12,20,30 > $O/bench_gemm256.log 2>&1; echo "bench_gemm256 exit $?"; grep "split f16" $O/bench_gemm256.log
timeout 900 python -m pytest tests/test_fulldepth_gpu.py -x -q -s > $O/t_fulldepth.log 2>&1; echo "fulldepth exit $?"; grep "fulldepth\]\|passed\|failed\|Error" $O/t_fulldepth.log | tail -12
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_e2e.log 2>&1; echo "bench exit $?"; tail -1 $O/bench_e2e.log | cut -c1-1500
