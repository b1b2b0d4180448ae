# round 2, GPU run 4: per-phase cycle breakdown of the lo8 GEMM (profiling build), lo8 prior tests
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_4
mkdir -p $O
export TMPDIR=/tmp
cd $R
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_lo8prof.so timeout 300 python scripts/prof_lo8.py > $O/prof_lo8.txt 2>&1; echo "prof exit $?"; cat $O/prof_lo8.txt | cut -c1-400
timeout 900 python -m pytest tests/test_lo8_gpu.py -x -q -s -p no:cacheprovider -k "prior or layernorm or qgelu or residual" > $O/t_lo8.log 2>&1; echo "lo8 tests exit $?"; grep -E "passed|failed|Error|error|rel err|assert" $O/t_lo8.log | cut -c1-300 | tail -12
