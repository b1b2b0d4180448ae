#!/bin/bash
# round 4, run 19: LayerNorm folded into the gemm256x epilogues -- op-level + 3-layer parity tests, then the bench A/B (fold on / off)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_prior_gpu.py -q -m gpu -x -s -k "ln_folded or folded_layernorm" 2>&1 | tail -25 ) > gpurun_out/r04/run19_tests.txt
cat gpurun_out/r04/run19_tests.txt | tail -12
if grep -q failed gpurun_out/r04/run19_tests.txt; then exit 1; fi
for fold in 1 0 1 0; do
  LLARK_PRIOR_LN_FOLD=$fold timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r04/run19_bench_fold$fold.txt 2>&1
  python - $fold <<'PY'
import json, sys
for l in open(f"gpurun_out/r04/run19_bench_fold{sys.argv[1]}.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("fold", sys.argv[1], "value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "avg_us", d["roofline"].get("avg_launch_us"))
PY
done
