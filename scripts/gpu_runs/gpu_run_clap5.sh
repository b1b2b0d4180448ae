#!/bin/bash
# CLAP: one-launch K-concatenated linears + fused GELU epilogue; tests then bench
mkdir -p gpurun_out/clap
timeout 600 python -m pytest tests/test_clap_gpu.py tests/test_mpt_gpu.py tests/test_prior_gpu.py -x -q -m gpu > gpurun_out/clap/tests5.log 2>&1; echo "tests exit $?"
grep -v amdgpu.ids gpurun_out/clap/tests5.log | tail -15
timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/clap/bench_clap_fp32_v3.log 2>&1
LLARK_CLAP_KCAT=0 timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/clap/bench_clap_fp32_v3_2launch.log 2>&1
timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --llm-precision bf16 --no-cpu-baseline > gpurun_out/clap/bench_clap_bf16_v3.log 2>&1
for f in gpurun_out/clap/bench_*_v3*.log; do echo "== $f"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' $f | tr '\n' ' '; echo; done
