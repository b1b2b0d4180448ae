#!/bin/bash
# round 4, run 24: consumer vectors + statistics loaded ahead of the stores (LDS-parked), after the producer got the same treatment in run 23
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_prior_gpu.py -q -m gpu -x -s -k "ln_folded or folded_layernorm or gemm256x or full_width" 2>&1 | tail -25 ) > gpurun_out/r04/run24_tests.txt
tail -12 gpurun_out/r04/run24_tests.txt
if grep -q failed gpurun_out/r04/run24_tests.txt; then exit 1; fi
for fold in 1 0 1 0; do
  LLARK_PRIOR_LN_FOLD=$fold timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r04/run24_bench_fold$fold.txt 2>&1
  python - $fold <<'PY'
import json, sys
for l in open(f"gpurun_out/r04/run24_bench_fold{sys.argv[1]}.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("fold", sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
done
for fold in 1 0; do
  rm -rf /tmp/prof$fold
  LLARK_PRIOR_LN_FOLD=$fold LLARK_PRIOR_PRECISION=f16x2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$fold -- python bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /tmp/prof$fold.log 2>&1
  f=$(find /tmp/prof$fold -name "*kernel_stats.csv" | head -1)
  head -12 "$f" | cut -c1-200 > gpurun_out/r04/run24_stats_fold$fold.csv
  cat gpurun_out/r04/run24_stats_fold$fold.csv
done
