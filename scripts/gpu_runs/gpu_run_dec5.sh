#!/bin/bash
mkdir -p gpurun_out/dec
for nw in 4 16 4 16; do
LLARK_ATTN_DECODE_NW=$nw timeout 600 python bench.py --stages mpt --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/dec/mpt_nw$nw.log 2>&1; echo "mpt NW=$nw: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/mpt_nw$nw.log | tr '\n' ' ')"
done
LLARK_ATTN_DECODE_NW=4 timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_nw4.log 2>&1; echo "gen B=1 split NW=4: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/gen_b1_nw4.log | tr '\n' ' ')"
LLARK_ATTN_DECODE_NW=16 timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_nw16.log 2>&1; echo "gen B=1 split NW=16: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/gen_b1_nw16.log | tr '\n' ' ')"
