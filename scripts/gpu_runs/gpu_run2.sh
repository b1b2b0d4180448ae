mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/summary.txt
for f in test_llama_gpu test_extract_gpu; do
  timeout 900 python -m pytest tests/$f.py -m gpu -q --tb=short -rA -p no:cacheprovider > gpurun_out/$f.log 2>&1; echo "$f exit $?" >> gpurun_out/summary.txt
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 1200 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_e2e.log 2>&1; echo "bench exit $?" >> gpurun_out/summary.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_e2e -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_e2e.log 2>&1; echo "rocprof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/summary.txt
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_e2e -name "*stats*" | head; find gpurun_out/prof_e2e -name "*kernel_trace*" -size +20M -delete
cat gpurun_out/summary.txt; tail -3 gpurun_out/bench_e2e.log
