R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_27
mkdir -p $O
export TMPDIR=/tmp
cd $R
for st in 0 200 400 700 1200 2000; do
  echo "== stagger $st (x10 ns per XCD)"
  LLARK_GEMM_STAGGER=$st timeout 300 python scripts/bench_gemm256.py 41 2>&1 | grep "^split" | sed 's/split f16 //' | cut -c1-120 | tee -a $O/stagger_$st.log
done
