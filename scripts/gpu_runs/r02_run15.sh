R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_15
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python scripts/debug_lo8_forms.py > $O/debug_s.txt 2>&1; echo "== debug exit $?"; grep -v amdgpu.ids $O/debug_s.txt | grep -v "hist\|tile (row\|outside" | cut -c1-250
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -p no:cacheprovider > $O/t_lo8_s.log 2>&1; echo "lo8 tests exit $?"; grep -E "passed|failed|Error|error|assert" $O/t_lo8_s.log | cut -c1-300 | tail -6
timeout 400 python scripts/bench_gemm256.py 30,41 > $O/bench_gemm_s.log 2>&1; echo "bench s exit $?"; grep "split f16" $O/bench_gemm_s.log
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof.so timeout 300 python scripts/prof_lo8.py lo8s > $O/prof_lo8s.txt 2>&1; echo "== lo8s prof exit $?"; grep -v amdgpu.ids $O/prof_lo8s.txt | cut -c1-330
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > $O/tests_full.log 2>&1; echo "full suite exit $?"; grep -E "fulldepth\]|passed|failed|^E  |^FAILED" $O/tests_full.log | cut -c1-300 | tail -24
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_e2e_lo8.log 2>&1; echo "e2e lo8 exit $?"; tail -1 $O/bench_e2e_lo8.log | cut -c1-700
