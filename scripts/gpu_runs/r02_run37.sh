R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_37
mkdir -p $O
export TMPDIR=/tmp
cd $R
echo "== cold weights, library rule" | tee -a $O/cold371.log
timeout 300 python scripts/bench_streamk.py 371 --cold 2>&1 | grep "^M=" | grep -v stream-k | tee -a $O/cold371.log
echo "== cold weights, LLARK_SK_KS=4" | tee -a $O/cold371.log
LLARK_SK_KS=4 timeout 300 python scripts/bench_streamk.py 371 --cold 2>&1 | grep "^M=" | grep "library" | tee -a $O/cold371.log
