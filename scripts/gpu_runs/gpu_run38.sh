mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e2e4 -o r -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_e2e4.log 2>&1; echo "exit $?"
python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/prof_e2e4 -name "*.db" | head -1) > $R/gpurun_out/prof_e2e4_stats.txt
head -34 $R/gpurun_out/prof_e2e4_stats.txt | cut -c1-175
tail -c 600 $R/gpurun_out/prof_e2e4.log
rm -rf $R/gpurun_out/prof_e2e4
