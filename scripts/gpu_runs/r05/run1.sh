#!/bin/bash
# Round 5, call 1: the Llama half at full depth in BOTH activation flows (tests/test_fulldepth_gpu.py) and the
# activation-flow probe (scripts/probes/llama_flow_error.py: oracle forward on the GPU with bf16 / fp16 rounding points).
mkdir -p gpurun_out/r05
{
  timeout 600 python -m pytest tests/test_fulldepth_gpu.py -q -s -k "llama" 2>&1 | grep -v "^$" | tail -30
  cp gpurun_out/llama_fulldepth_parity.json gpurun_out/r05/ 2>/dev/null
  timeout 600 python scripts/probes/llama_flow_error.py 2>&1 | tail -12
  cp gpurun_out/llama_flow_error.json gpurun_out/r05/ 2>/dev/null
} > gpurun_out/r05/run1.txt 2>&1
cat gpurun_out/r05/run1.txt
