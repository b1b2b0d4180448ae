R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_42
mkdir -p $O
export TMPDIR=/tmp
cd $R
for kw in 0 16 8 4; do
  echo "== LLARK_SKINNY_KW=$kw" | tee -a $O/decode_kw.log
  if [ $kw = 0 ]; then unset LLARK_SKINNY_KW; else export LLARK_SKINNY_KW=$kw; fi
  timeout 300 python scripts/bench_decode.py split 2>&1 | grep "^decode" | grep "B=1" | tee -a $O/decode_kw.log
done
