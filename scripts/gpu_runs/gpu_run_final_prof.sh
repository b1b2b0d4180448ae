#!/bin/bash
# final-code kernel trace of the headline bench command (1 timed step)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_e2e5 -o r01 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_e2e5.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py gpurun_out/prof_e2e5/r01_results.db gpurun_out/prof_e2e5_stats.txt; rm -rf gpurun_out/prof_e2e5
head -12 gpurun_out/prof_e2e5_stats.txt | cut -c1-170; grep '^{' gpurun_out/prof_e2e5.log | cut -c1-400
