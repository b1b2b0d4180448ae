#!/bin/bash
# round 4, run 5: B-direct kernels with the MFMA-first K-step (bd_kstep, GEMM_BD_SCHED = 1) against round 3's order (libllark_hip_s0.so):
# bit-identity / parity tests, then the Llama stage in both libraries, alternating (same box)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_llama_gpu.py tests/test_fulldepth_gpu.py -x -q -k "not jukebox" 2>&1 | tail -6 ) > gpurun_out/r04/run5_tests.txt
for rep in 1 2; do
  for lib in s0 new; do
    if [ $lib = s0 ]; then export LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_s0.so; else unset LLARK_HIP_LIB; fi
    timeout 600 python bench.py --stages llama --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib rep $rep', 'ms/step', d['ms_per_step'], 'split gemm frac', d['roofline_llm']['frac'], 'avg', d['roofline_llm']['avg_launch_ms'], '| bf16 fwd ms', d['roofline_llm_bf16']['llama_ms_per_step'], 'gemm frac', d['roofline_llm_bf16']['frac'])
" >> gpurun_out/r04/llama_bd_ab.txt
  done
done
tail -4 gpurun_out/r04/run5_tests.txt; cat gpurun_out/r04/llama_bd_ab.txt
