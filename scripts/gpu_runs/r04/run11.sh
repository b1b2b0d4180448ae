#!/bin/bash
# round 4, run 11: split B-direct K-step with all hi products before all lo products (GEMM_BD_SPLIT_ORDER = 1) against pairs back to back
# (libllark_hip_s0.so): bit-identity tests, then the Llama stage in both libraries, alternating
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_llama_gpu.py -x -q -k "fragment or llama or engine or decode or split" 2>&1 | tail -4 ) > gpurun_out/r04/run11_tests.txt
rm -f gpurun_out/r04/llama_bd_split_order_ab.txt
for rep in 1 2; do
  for lib in s0 new; do
    if [ $lib = s0 ]; then export LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_s0.so; else unset LLARK_HIP_LIB; fi
    timeout 600 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib rep $rep', 'ms/step', d['ms_per_step'], 'split gemm frac', d['roofline_llm']['frac'], 'avg', d['roofline_llm']['avg_launch_ms'], '| bf16 fwd ms', d['roofline_llm_bf16']['llama_ms_per_step'], 'gemm frac', d['roofline_llm_bf16']['frac'])
" >> gpurun_out/r04/llama_bd_split_order_ab.txt
  done
done
tail -3 gpurun_out/r04/run11_tests.txt; cat gpurun_out/r04/llama_bd_split_order_ab.txt
