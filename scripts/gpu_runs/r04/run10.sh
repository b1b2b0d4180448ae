#!/bin/bash
# round 4, run 10: validation -- the whole GPU suite, smoke(), the default bench line exactly as the driver runs it, and the rocprofv3
# --kernel-trace --stats summary of the e2e step
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r04/run10_suite.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r04/run10_smoke.txt
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04/run10_bench.txt 2>&1
( timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_e2e -o a -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-alt-precision > gpurun_out/r04/run10_prof.log 2>&1
  f=$(find gpurun_out/r04/prof_e2e -name "*.db" | head -1); python scripts/rocprof_summary.py $f gpurun_out/r04/e2e_kernel_stats.txt )
rm -rf gpurun_out/r04/prof_e2e
tail -3 gpurun_out/r04/run10_suite.txt; cat gpurun_out/r04/run10_smoke.txt; tail -4 gpurun_out/r04/run10_bench.txt | cut -c1-600; head -30 gpurun_out/r04/e2e_kernel_stats.txt
