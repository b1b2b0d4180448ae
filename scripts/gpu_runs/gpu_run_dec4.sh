#!/bin/bash
# decode attention with 16 waves per (batch, head) for small grids: tests + generate B=1 / B=8 + mpt
mkdir -p gpurun_out/dec
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_infer_driver.py tests/test_mpt_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/dec/tests4.log 2>&1; echo "tests exit $?"; grep -v amdgpu.ids gpurun_out/dec/tests4.log | tail -5
timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_v2.log 2>&1; echo "gen B=1 split: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b1_v2.log | tr '\n' ' ')"
timeout 600 python bench.py --stages generate --llm-precision bf16 --no-cpu-baseline > gpurun_out/dec/gen_b1_bf16_v2.log 2>&1; echo "gen B=1 bf16: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b1_bf16_v2.log | tr '\n' ' ')"
timeout 600 python bench.py --stages generate --batch 8 --no-cpu-baseline > gpurun_out/dec/gen_b8_v2.log 2>&1; echo "gen B=8 split: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b8_v2.log | tr '\n' ' ')"
timeout 600 python bench.py --stages mpt --no-cpu-baseline > gpurun_out/dec/mpt_v2.log 2>&1; echo "mpt: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/mpt_v2.log | tr '\n' ' ')"
