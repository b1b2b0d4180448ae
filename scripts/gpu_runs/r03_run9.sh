#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_prior_gpu.py tests/test_lo8_gpu.py tests/test_fulldepth_gpu.py -x -q -s 2>&1 | grep "fulldepth\|passed\|failed\|Error" | cut -c1-400 ) > gpurun_out/r03_run9_tests.txt; cat gpurun_out/r03_run9_tests.txt
LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_lo8prof.so timeout 300 python scripts/prof_lo8.py f16x2n 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_phase_cycles_v2.txt; cat gpurun_out/r03_phase_cycles_v2.txt | cut -c1-300
timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/r03_bench_e2e_v3.json; cut -c1-400 gpurun_out/r03_bench_e2e_v3.json
