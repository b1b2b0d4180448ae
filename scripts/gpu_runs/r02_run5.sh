# round 2, GPU run 5: phase-cycle breakdown: lo8 with / without the static priority of the younger half, and the f16x2 kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_5
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in prio noprio; do
  LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof_$v.so timeout 300 python scripts/prof_lo8.py lo8 > $O/prof_lo8_$v.txt 2>&1; echo "== $v exit $?"; grep -v amdgpu.ids $O/prof_lo8_$v.txt | cut -c1-330
done
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof_prio.so timeout 300 python scripts/prof_lo8.py f16x2 > $O/prof_f16x2.txt 2>&1; echo "== f16x2 exit $?"; grep -v amdgpu.ids $O/prof_f16x2.txt | cut -c1-330
timeout 400 python scripts/bench_gemm256.py 30,40 > $O/bench_gemm_lo8.log 2>&1; echo "bench exit $?"; grep "split f16\|lo8" $O/bench_gemm_lo8.log
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -s -p no:cacheprovider -k "prior" > $O/t_lo8.log 2>&1; echo "lo8 tests exit $?"; grep -E "passed|failed|Error|error|rel err|assert" $O/t_lo8.log | cut -c1-300 | tail -12
