#!/bin/bash
# Round 4, last validation: the tests of every other user of HipLlamaEngine (training step, attention backward, transposed GEMMs, MPT)
# on the tree with the fused q|k|v + RoPE epilogue as the default, then one short end-to-end line for the record.
mkdir -p gpurun_out/r04
{
  timeout 240 python -m pytest tests/test_train_gpu.py tests/test_attn_bwd_gpu.py tests/test_gemm_tn_gpu.py tests/test_mpt_gpu.py tests/test_gemv_dma_gpu.py -q -x 2>&1 | tail -5
  timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-alt-precision 2>&1 | tail -1 > gpurun_out/r04/bench_e2e_rope_fused.json
  cut -c1-700 gpurun_out/r04/bench_e2e_rope_fused.json
} > gpurun_out/r04/run_last3.txt 2>&1
cat gpurun_out/r04/run_last3.txt
