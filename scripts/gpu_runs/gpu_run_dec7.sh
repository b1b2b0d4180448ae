#!/bin/bash
# norm kernels with weight loads hoisted above the reductions: parity tests + decode-heavy benches
mkdir -p gpurun_out/dec
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_mpt_gpu.py tests/test_clap_gpu.py -x -q -m gpu > gpurun_out/dec/tests7.log 2>&1; echo "tests exit $?"; grep -v amdgpu.ids gpurun_out/dec/tests7.log | tail -4
timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_v3.log 2>&1; echo "gen B=1 split: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/gen_b1_v3.log | tr '\n' ' ')"
timeout 600 python bench.py --stages mpt --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/dec/mpt_v3.log 2>&1; echo "mpt: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/dec/mpt_v3.log | tr '\n' ' ')"
