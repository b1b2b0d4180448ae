#!/bin/bash
# kernel trace of the generate stage (B=1, split) to see what the 3.5 ms decode step is made of
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/dec
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dec/prof -o gen -- python $R/bench.py --stages generate --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/dec/prof.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py gpurun_out/dec/prof/gen_results.db gpurun_out/dec/gen_b1_kernel_stats.txt; rm -rf gpurun_out/dec/prof
head -24 gpurun_out/dec/gen_b1_kernel_stats.txt | cut -c1-190
