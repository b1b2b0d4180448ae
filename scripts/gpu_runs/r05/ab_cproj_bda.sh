#!/bin/bash
# In-step A/B of LLARK_PRIOR_CPROJ_BDA (the attention-output product as LayerNorm producer on gemm_bda's tiles vs the persistent tile):
# the Jukebox stage of bench.py twice each, interleaved, + the 36-layer parity fixture under the knob.
mkdir -p gpurun_out/r05
{
for rep in 1 2; do
  for v in 0 1; do
    echo "== LLARK_PRIOR_CPROJ_BDA=$v rep $rep"
    LLARK_PRIOR_CPROJ_BDA=$v python bench.py --stages jukebox --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_launch_ms', d['roofline'].get('avg_launch_ms'))
"
  done
done
echo "== full-depth parity under LLARK_PRIOR_CPROJ_BDA=1"
LLARK_PRIOR_CPROJ_BDA=1 python -m pytest tests/test_fulldepth_gpu.py -q -x -k "36_layers_batch8" 2>&1 | tail -8
} 2>&1 | tee gpurun_out/r05/cproj_bda_ab.txt
