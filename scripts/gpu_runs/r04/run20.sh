#!/bin/bash
# round 4, run 20: per-kernel time of the folded-LayerNorm variants of gemm256x against the plain ones (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fold in 1 0; do
  rm -rf /tmp/prof$fold
  LLARK_PRIOR_LN_FOLD=$fold timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$fold -- python bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --stages jukebox > /tmp/prof$fold.log 2>&1
  f=$(find /tmp/prof$fold -name "*kernel_stats.csv" | head -1)
  head -14 "$f" | cut -c1-260 > gpurun_out/r04/run20_stats_fold$fold.csv
  cat gpurun_out/r04/run20_stats_fold$fold.csv
done
