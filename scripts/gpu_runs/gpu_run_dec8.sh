#!/bin/bash
# fused RoPE + cache append + decode attention: parity tests in both modes, then the decode-heavy benches
mkdir -p gpurun_out/dec
timeout 400 python -m pytest tests/test_llama_gpu.py tests/test_mpt_gpu.py tests/test_infer_driver.py -q -m gpu > gpurun_out/dec/tests8_fused.log 2>&1; echo "fused tests exit $?"; grep -v amdgpu.ids gpurun_out/dec/tests8_fused.log | tail -4
LLARK_DECODE_FUSE_ROPE=0 timeout 400 python -m pytest tests/test_llama_gpu.py tests/test_mpt_gpu.py tests/test_infer_driver.py -q -m gpu > gpurun_out/dec/tests8_unfused.log 2>&1; echo "unfused tests exit $?"; grep -v amdgpu.ids gpurun_out/dec/tests8_unfused.log | tail -2
timeout 300 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_v4.log 2>&1; echo "gen B=1 split fused: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/gen_b1_v4.log | tr '\n' ' ')"
LLARK_DECODE_FUSE_ROPE=0 timeout 300 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_v4_unfused.log 2>&1; echo "gen B=1 split unfused: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/gen_b1_v4_unfused.log | tr '\n' ' ')"
timeout 300 python bench.py --stages mpt --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/dec/mpt_v4.log 2>&1; echo "mpt fused: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/mpt_v4.log | tr '\n' ' ')"
