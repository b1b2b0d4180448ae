#!/bin/bash
mkdir -p gpurun_out/dec
for cfg in "0 48" "1 48" "1 24" "1 96"; do set -- $cfg
LLARK_DECODE_PREFETCH=$1 LLARK_DECODE_PREFETCH_MB=$2 timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_pf$1_$2.log 2>&1; echo "gen B=1 split prefetch=$1 mb=$2: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/gen_b1_pf$1_$2.log | tr '\n' ' ')"
done
