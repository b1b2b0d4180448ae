#!/bin/bash
# Training step: dX = dY . W on W as stored (llark_gemm16_t, default) against dX on a per-step W^T + fragment-major twin through the
# B-direct DMA loop (gemm_bda, plain bf16), LLARK_TRAIN_DX_DIRECT_USES = number of dX products per weight that stay on gemm_t.
mkdir -p gpurun_out/r05
{
for v in 1073741824 0 1; do
  echo "== LLARK_TRAIN_DX_DIRECT_USES=$v"
  LLARK_TRAIN_DX_DIRECT_USES=$v python bench.py --stages train --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'mfu', d.get('mfu') or d.get('roofline',{}).get('frac'), 'peak_hbm_gb', d.get('peak_hbm_gb'), d.get('last_micro_batch_ms'))
"
done
} 2>&1 | tee gpurun_out/r05/train_dx_ab.txt
