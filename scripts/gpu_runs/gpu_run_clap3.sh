#!/bin/bash
# CLAP bench: fp32-class and bf16 modes at B=64, the configs[4] stage with the real encoder, kernel trace of the clap stage
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/clap
export TMPDIR=/tmp
cd $R
timeout 300 python bench.py --stages clap --steps 3 --warmup 1 > gpurun_out/clap/bench_clap_fp32.log 2>&1
timeout 300 python bench.py --stages clap --steps 3 --warmup 1 --llm-precision bf16 --no-cpu-baseline > gpurun_out/clap/bench_clap_bf16.log 2>&1
timeout 300 python bench.py --stages mpt --steps 3 --warmup 1 > gpurun_out/clap/bench_mpt_clap.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/clap/prof -o clap -- python $R/bench.py --stages clap --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/clap/prof.log 2>&1; echo "rocprof exit $?"
cd $R
python scripts/rocprof_summary.py gpurun_out/clap/prof/clap_results.db gpurun_out/clap/kernel_stats.txt; rm -rf gpurun_out/clap/prof
for f in gpurun_out/clap/bench_*.log; do echo "== $f"; grep -v amdgpu.ids $f | tail -3; done
head -30 gpurun_out/clap/kernel_stats.txt
