#!/bin/bash
# Round 4, last GPU call (8 GPU-minutes left): the fused q|k|v + RoPE epilogue -- parity tests, then the same-engine A/B --
# and the file-level audio path with the restated resampler.  Everything under its own timeout.
mkdir -p gpurun_out/r04
{
  timeout 150 python -m pytest tests/test_llama_gpu.py -q -x -k "rope_qkv_epilogue or prefill_rope_fused" 2>&1 | tail -8
  timeout 150 python scripts/gpu_runs/r04/rope_fuse_ab.py 2>&1 | tail -6
  timeout 90 python -m pytest tests/test_extract_gpu.py -q -x 2>&1 | tail -4
} > gpurun_out/r04/run_last.txt 2>&1
cat gpurun_out/r04/run_last.txt
