R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_30
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in head new head new; do
  if [ $v = new ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_headlo8n.so; fi
  echo "== $v"; timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | sed 's/split f16 //' | cut -c1-120 | tee -a $O/ab_$v.log
done
unset LLARK_HIP_LIB
timeout 900 python bench.py --no-cpu-baseline > $O/bench_e2e.log 2>&1; echo "e2e exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}\|"frac": [0-9.]*' $O/bench_e2e.log | tr '\n' ' ')"
LLARK_HIP_LIB=$R/llark_amd/libllark_hip_headlo8n.so timeout 900 python bench.py --no-cpu-baseline > $O/bench_e2e_head.log 2>&1; echo "e2e head exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_e2e_head.log | tr '\n' ' ')"
