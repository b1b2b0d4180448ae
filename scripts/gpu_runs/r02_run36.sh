R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_36
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in base fakefrag base fakefrag; do
  if [ $v = base ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_$v.so; fi
  echo "== $v" | tee -a $O/decode_ab.log; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "^decode" | tee -a $O/decode_ab.log
done
