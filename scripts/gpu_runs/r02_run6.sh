# round 2, GPU run 6: lo8 with alternating wave priority
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_6
mkdir -p $O
export TMPDIR=/tmp
cd $R
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof_prio.so timeout 300 python scripts/prof_lo8.py lo8 > $O/prof_lo8_altprio.txt 2>&1; echo "== altprio exit $?"; grep -v amdgpu.ids $O/prof_lo8_altprio.txt | cut -c1-330
timeout 400 python scripts/bench_gemm256.py 30,40 > $O/bench_gemm_lo8.log 2>&1; echo "bench exit $?"; grep "split f16\|lo8" $O/bench_gemm_lo8.log
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -p no:cacheprovider -k "gemm" > $O/t_lo8.log 2>&1; echo "lo8 tests exit $?"; tail -2 $O/t_lo8.log
