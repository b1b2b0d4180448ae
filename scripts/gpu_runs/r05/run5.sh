#!/bin/bash
# Round 5, call 5: folded LayerNorm with predicted row statistics (llark_gemm16_ln_p): role / adversarial-row tests, the 36-layer
# fixtures with the prediction on (default) and off (LLARK_PRIOR_LN_PRED=0), and the jukebox stage both ways.
mkdir -p gpurun_out/r05
out=gpurun_out/r05/run5.txt
: > $out
timeout 600 python -m pytest tests/test_prior_gpu.py -q -s -k "ln or layernorm or prior_full or prior_tiny" 2>&1 | grep -E "ln-pred|ln-fold|passed|failed|Error|error" | cut -c1-400 >> $out
for pred in 1 0; do
  echo "== LLARK_PRIOR_LN_PRED=$pred" >> $out
  LLARK_PRIOR_LN_PRED=$pred timeout 900 python -m pytest tests/test_fulldepth_gpu.py -q -s -k "jukebox and not lo8" 2>&1 | grep -E "fulldepth\]|wide\]|passed|failed|Error|assert" | cut -c1-500 >> $out
  LLARK_PRIOR_LN_PRED=$pred timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --no-extras --no-cpu-baseline --stages jukebox > /tmp/b.txt 2>&1
  python - <<'PY' >> $out
import json
for l in open("/tmp/b.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print("jukebox stage: value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", {k: v for k, v in d["kernel_ms"].items()})
PY
done
cat $out
