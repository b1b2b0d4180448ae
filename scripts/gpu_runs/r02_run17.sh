R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_17
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python scripts/debug_lo8_forms.py > $O/debug_n.txt 2>&1; echo "== debug (form n) exit $?"; grep -v amdgpu.ids $O/debug_n.txt | grep -v "hist\|tile (row\|outside" | cut -c1-250
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -p no:cacheprovider > $O/t_lo8_n.log 2>&1; echo "lo8 tests (form n) exit $?"; grep -E "passed|failed|Error|error|assert" $O/t_lo8_n.log | cut -c1-300 | tail -6
timeout 400 python scripts/bench_gemm256.py 30,41 > $O/bench_gemm_n.log 2>&1; echo "bench n exit $?"; grep "split f16" $O/bench_gemm_n.log
LLARK_LO8_FORM=s timeout 400 python scripts/bench_gemm256.py 41 > $O/bench_gemm_s.log 2>&1; echo "bench s exit $?"; grep "split f16" $O/bench_gemm_s.log
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof.so timeout 300 python scripts/prof_lo8.py lo8s > $O/prof_lo8n.txt 2>&1; echo "== lo8n prof exit $?"; grep -v amdgpu.ids $O/prof_lo8n.txt | cut -c1-330
