mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for fr in 1 0; do
  LLARK_FRAG=$fr timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_frag$fr -o r -- python $R/bench.py --stages jukebox --depth 6 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_frag$fr.log 2>&1; echo "exit $?"
  python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/prof_frag$fr -name "*.db" | head -1) | grep -E "gemm" | cut -c1-200
  rm -rf $R/gpurun_out/prof_frag$fr
done
cd $R
timeout 900 python -m pytest tests/test_llama_gpu.py -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/tests25.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/tests25.log | tail -2; grep -E "^E  " gpurun_out/tests25.log | cut -c1-300 | head -20
