#!/bin/bash
# fused-norm (norm-A) skinny GEMM after the load-scheduling fix: tests + generate B=1 / decode B=8 with and without
mkdir -p gpurun_out/dec
timeout 600 python -m pytest tests/test_llama_gpu.py tests/test_infer_driver.py -x -q -m gpu > gpurun_out/dec/tests3.log 2>&1; echo "tests exit $?"; grep -v amdgpu.ids gpurun_out/dec/tests3.log | tail -5
for na in 0 1; do
LLARK_DECODE_FUSE_NORM_A=$na timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_na$na.log 2>&1; echo "gen B=1 split norm_a=$na: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b1_na$na.log | tr '\n' ' ')"
LLARK_DECODE_FUSE_NORM_A=$na timeout 600 python bench.py --stages generate --llm-precision bf16 --no-cpu-baseline > gpurun_out/dec/gen_b1_bf16_na$na.log 2>&1; echo "gen B=1 bf16 norm_a=$na: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b1_bf16_na$na.log | tr '\n' ' ')"
LLARK_DECODE_FUSE_NORM_A=$na timeout 600 python bench.py --stages generate --batch 8 --no-cpu-baseline > gpurun_out/dec/gen_b8_na$na.log 2>&1; echo "gen B=8 split norm_a=$na: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b8_na$na.log | tr '\n' ' ')"
done
