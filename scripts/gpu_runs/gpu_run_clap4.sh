#!/bin/bash
# CLAP: launch-list replay + workspace arena; tests then bench (fp32 / bf16) and the configs[4] stage
mkdir -p gpurun_out/clap
timeout 600 python -m pytest tests/test_clap_gpu.py -x -q -m gpu > gpurun_out/clap/tests4.log 2>&1; echo "tests exit $?"
grep -v amdgpu.ids gpurun_out/clap/tests4.log | tail -15
timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/clap/bench_clap_fp32_v2.log 2>&1
timeout 300 python bench.py --stages clap --steps 5 --warmup 2 --llm-precision bf16 --no-cpu-baseline > gpurun_out/clap/bench_clap_bf16_v2.log 2>&1
timeout 300 python bench.py --stages mpt --steps 3 --warmup 2 > gpurun_out/clap/bench_mpt_clap_v2.log 2>&1
for f in gpurun_out/clap/bench_*_v2.log; do echo "== $f"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' $f | tr '\n' ' '; echo; done
