#!/bin/bash
# Round 6: chained decode launches -- parity test, then the generate stage A/B (LLARK_DECODE_CHAIN=0/1), one box.
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_llama_gpu.py -q -x --tb=short -p no:cacheprovider -k "chained or decode" 2>&1 | tail -6
for rep in 1 2; do
for ch in 0 1; do
  echo "-- LLARK_DECODE_CHAIN=$ch rep $rep"
  LLARK_DECODE_CHAIN=$ch timeout 600 python bench.py --stages generate --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('ms_per_clip', d.get('ms_per_step') or d.get('ms_per_clip'), 'value', d['value'], 'decode', (d.get('roofline_decode') or {}).get('decode_ms_per_token'), (d.get('roofline_decode') or {}).get('frac'))
"
done
done
} 2>&1 | tee gpurun_out/r06/decode_chain_ab.txt
