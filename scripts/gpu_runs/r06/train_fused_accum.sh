#!/bin/bash
# The recipe's accumulation (micro-batch 2 x 4) against the same 8 clips x 2048 tokens as one pass with per-group loss normalisation.
mkdir -p gpurun_out/r06
out=gpurun_out/r06/train_fused_accum.txt
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -k "fused_accumulation or twin_paths" 2>&1 | tail -3 > $out
for rep in 1 2; do
  echo "-- micro-batch 2 x accumulation 4 (rep $rep)" >> $out
  timeout 900 python bench.py --stages train --batch 8 --micro-batch 2 --train-seq 2048 --grad-comm bf16 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'mfu', d.get('mfu'), 'peak_hbm_gb', d.get('peak_hbm_gb'))" >> $out
  echo "-- one pass of 8 clips, 4 loss groups (rep $rep)" >> $out
  timeout 900 python bench.py --stages train --batch 8 --micro-batch 8 --loss-groups 4 --train-seq 2048 --grad-comm bf16 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'mfu', d.get('mfu'), 'peak_hbm_gb', d.get('peak_hbm_gb'))" >> $out
done
cat $out
