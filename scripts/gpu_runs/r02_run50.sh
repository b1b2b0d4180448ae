R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_50
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in noskew skew noskew skew; do
  if [ $v = skew ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_$v.so; fi
  timeout 600 python bench.py --no-cpu-baseline > $O/bench_e2e_$v.log 2>&1; echo "e2e $v exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_e2e_$v.log | head -3 | tr '\n' ' ')" | tee -a $O/summary.log
done
