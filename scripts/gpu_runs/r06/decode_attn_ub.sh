#!/bin/bash
# Round 6: decode attention with 8 loads in flight per lane (one round of memory latency per pass up to 512 keys) against 4 (two rounds at the
# 25 s prompt's 371 .. 435 keys): parity tests, then the generate stage with both libraries alternating on one box.
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_llama_gpu.py tests/test_mpt_gpu.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
for rep in 1 2; do
for lib in libllark_hip_ub4.so libllark_hip.so; do
  echo "-- $lib rep $rep"
  LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/$lib timeout 600 python bench.py --stages generate --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('ms_per_clip', d.get('ms_per_step') or d.get('ms_per_clip'), 'value', d['value'])
"
done
done
} 2>&1 | tee gpurun_out/r06/decode_attn_ub_ab.txt
