R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_26
mkdir -p $O
export TMPDIR=/tmp
cd $R
for v in base nt d8 ntd8; do
  if [ $v = base ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$R/llark_amd/libllark_hip_sk_$v.so; fi
  echo "== $v"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "^decode"
done
unset LLARK_HIP_LIB
timeout 300 python bench.py --stages llama --no-cpu-baseline --steps 3 > $O/bench_llama_both.log 2>&1; echo "llama exit $?"; tail -1 $O/bench_llama_both.log | grep -o '"roofline_llm[a-z_0-9]*": {[^}]*}' | cut -c1-400
