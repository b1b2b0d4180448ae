#!/bin/bash
# CLAP: fp32-MFMA window attention + pick_variant change; tests (clap + everything that uses non-split GEMMs), then bench
mkdir -p gpurun_out/clap
timeout 900 python -m pytest tests/test_clap_gpu.py tests/test_mpt_gpu.py tests/test_prior_gpu.py tests/test_llama_gpu.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/clap/tests7.log 2>&1; echo "tests exit $?"
grep -v amdgpu.ids gpurun_out/clap/tests7.log | tail -15
timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/clap/bench_clap_fp32_v4.log 2>&1
LLARK_CLAP_ATTN_VALU=1 timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/clap/bench_clap_fp32_v4_valu.log 2>&1
timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --llm-precision bf16 --no-cpu-baseline > gpurun_out/clap/bench_clap_bf16_v4.log 2>&1
for f in gpurun_out/clap/bench_*_v4*.log; do echo "== $f"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' $f | tr '\n' ' '; echo; done
