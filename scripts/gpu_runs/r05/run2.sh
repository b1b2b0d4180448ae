#!/bin/bash
# Round 5, call 2: bf16 flow of the Llama half at full depth (figures recorded before the assertion), and the per-tile time stamps of
# gemm256x (profiling build) at 256 / 128 / 64 / 32 resident workgroups.
mkdir -p gpurun_out/r05
{
  LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_prof.so timeout 600 python scripts/probes/gemm256x_tile_times.py 256 128 64 32 2>&1 | grep -v amdgpu.ids
  timeout 600 python -m pytest tests/test_fulldepth_gpu.py -q -s -k "llama and bf16" 2>&1 | grep -E "fulldepth\]|passed|failed|Error" | cut -c1-600
  cp gpurun_out/llama_fulldepth_parity.json gpurun_out/r05/llama_fulldepth_parity_bf16.json 2>/dev/null
} > gpurun_out/r05/run2.txt 2>&1
cat gpurun_out/r05/run2.txt
