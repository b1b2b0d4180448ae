R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_14
mkdir -p $O
export TMPDIR=/tmp
cd $R
LLARK_LO8_FORM=q timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -p no:cacheprovider > $O/t_lo8_q.log 2>&1; echo "lo8 tests (form q) exit $?"; grep -E "passed|failed|Error|error|assert" $O/t_lo8_q.log | cut -c1-300 | tail -6
LLARK_LO8_FORM=q LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof.so timeout 300 python scripts/prof_lo8.py lo8s > $O/prof_lo8q.txt 2>&1; echo "== lo8q prof exit $?"; grep -v amdgpu.ids $O/prof_lo8q.txt | cut -c1-330
timeout 400 python scripts/bench_gemm256.py 30,41 > $O/bench_gemm_s.log 2>&1; echo "bench s exit $?"; grep "v41" $O/bench_gemm_s.log
LLARK_LO8_FORM=q timeout 400 python scripts/bench_gemm256.py 41 > $O/bench_gemm_q.log 2>&1; echo "bench q exit $?"; grep "v41" $O/bench_gemm_q.log
