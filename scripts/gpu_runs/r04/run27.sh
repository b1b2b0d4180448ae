#!/bin/bash
# round 4, run 27: rocprofv3 --kernel-trace --stats summary of the e2e step as committed (LayerNorm folded) -> profiles/r04_e2e_kernel_stats.txt
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_e2e -o a -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-alt-precision > gpurun_out/r04/run27_prof.log 2>&1
  f=$(find gpurun_out/r04/prof_e2e -name "*.db" | head -1); python scripts/rocprof_summary.py $f gpurun_out/r04/e2e_kernel_stats_lnfold.txt )
rm -rf gpurun_out/r04/prof_e2e
grep "^{" gpurun_out/r04/run27_prof.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_launch_ms', d['roofline']['avg_launch_ms'], 'launches', d['roofline']['launches'])"
head -24 gpurun_out/r04/e2e_kernel_stats_lnfold.txt | cut -c1-180
