#!/bin/bash
# Round 4, after the fused q|k|v + RoPE epilogue became the engine's default ("auto"): every Llama-side GPU test, the 32-layer 7B
# fixture, smoke, and the Llama stage of bench.py for the record.
mkdir -p gpurun_out/r04
{
  timeout 200 python -m pytest tests/test_llama_gpu.py tests/test_infer_driver.py -q -x 2>&1 | tail -5
  timeout 200 python -m pytest tests/test_fulldepth_gpu.py -q -x -k "llama" 2>&1 | tail -5
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  timeout 200 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r04/bench_llama_rope_fused.json
  cat gpurun_out/r04/bench_llama_rope_fused.json | cut -c1-1500
} > gpurun_out/r04/run_last2.txt 2>&1
cat gpurun_out/r04/run_last2.txt
