R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_46
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in base skew skew_NOWAIT skew_NODUP base skew_NOWAIT skew_NODUP; do
  if [ $v = base ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_$v.so; fi
  echo "== $v" | tee -a $O/skew.log; timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | sed 's/split f16 //' | cut -c1-80 | tee -a $O/skew.log
done
