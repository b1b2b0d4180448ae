#!/bin/bash
# Round 4: rocprofv3 --kernel-trace --stats of the e2e step on the tree as committed (RoPE in the q|k|v epilogue)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 280 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_e2e2 -o a -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-alt-precision > gpurun_out/r04/run_last4_prof.log 2>&1
  f=$(find gpurun_out/r04/prof_e2e2 -name "*.db" | head -1); python scripts/rocprof_summary.py $f gpurun_out/r04/e2e_kernel_stats_rope_fused.txt )
rm -rf gpurun_out/r04/prof_e2e2
tail -1 gpurun_out/r04/run_last4_prof.log | cut -c1-400; head -24 gpurun_out/r04/e2e_kernel_stats_rope_fused.txt | cut -c1-190
