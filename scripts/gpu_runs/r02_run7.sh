# round 2, GPU run 7: staged-W8 lo8 kernel: parity, phase cycles, A/B timing
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_7
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -p no:cacheprovider > $O/t_lo8.log 2>&1; echo "lo8 tests exit $?"; grep -E "passed|failed|Error|error|assert" $O/t_lo8.log | cut -c1-300 | tail -8
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof.so timeout 300 python scripts/prof_lo8.py lo8s > $O/prof_lo8s.txt 2>&1; echo "== lo8s prof exit $?"; grep -v amdgpu.ids $O/prof_lo8s.txt | cut -c1-330
timeout 400 python scripts/bench_gemm256.py 30,40,41 > $O/bench_gemm_lo8.log 2>&1; echo "bench exit $?"; grep "split f16\|lo8" $O/bench_gemm_lo8.log
