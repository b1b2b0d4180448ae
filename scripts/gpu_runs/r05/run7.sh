#!/bin/bash
# Round 5, call 7: Llama-side GPU tests after the lazy fused-RoPE weights, the C-side whole-tile rule and output_hidden_states.
mkdir -p gpurun_out/r05
{
  timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_infer_driver.py tests/test_gemv_dma_gpu.py -q -x 2>&1 | tail -8
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
} > gpurun_out/r05/run7.txt 2>&1
cat gpurun_out/r05/run7.txt
