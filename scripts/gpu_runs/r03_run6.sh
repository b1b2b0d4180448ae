#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 600 python scripts/debug/vq_fused_diff.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_vq_fused_diff_v2.txt; cat gpurun_out/r03_vq_fused_diff_v2.txt
( timeout 900 python -m pytest tests/test_vqvae_gpu.py -x -q -s 2>&1 | grep "vqvae fused\|passed\|failed\|Error" ) > gpurun_out/r03_vqvae_fused_tests.txt; cat gpurun_out/r03_vqvae_fused_tests.txt
( timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_mpt_gpu.py tests/test_clap_gpu.py -x -q 2>&1 | tail -3 ) > gpurun_out/r03_run6_llm_tests.txt; cat gpurun_out/r03_run6_llm_tests.txt
timeout 600 python bench.py --stages llama --llm-precision bf16 --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/r03_bench_llama_bf16_v1.json; cat gpurun_out/r03_bench_llama_bf16_v1.json | cut -c1-1500
