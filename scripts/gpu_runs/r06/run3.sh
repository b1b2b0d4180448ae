#!/bin/bash
# Round 6: (a) train stage A/B at the recipe shape, un-profiled; (b) activation-flow probe with the 1.5-pass (16-bit hi + MX-fp8 lo) rounding points.
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
{
echo "== train stage, recipe shape (8 x 2048 = micro-batch 2 x accumulation 4)"
for tw in 0 1; do
  echo "-- LLARK_TRAIN_TWINS=$tw LLARK_TRAIN_DW_FRAGW=$tw LLARK_TRAIN_ATTN_GLUE_FUSED=$tw"
  LLARK_TRAIN_TWINS=$tw LLARK_TRAIN_DW_FRAGW=$tw LLARK_TRAIN_ATTN_GLUE_FUSED=$tw timeout 900 python bench.py --stages train --batch 8 --micro-batch 2 --train-seq 2048 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'mfu', d.get('mfu') or d.get('roofline',{}).get('frac'), 'peak_hbm_gb', d.get('peak_hbm_gb'), d.get('last_micro_batch_ms'))
"
done
} 2>&1 | tee gpurun_out/r06/train_ab.txt
{
echo "== flow probe"
timeout 1200 python scripts/probes/llama_flow_error.py 2>&1 | tail -12
cp gpurun_out/llama_flow_error.json gpurun_out/r06/llama_flow_error.json
} 2>&1 | tee gpurun_out/r06/flow_probe.txt
