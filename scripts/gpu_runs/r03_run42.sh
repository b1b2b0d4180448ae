#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gemv_dma_gpu.py tests/test_llama_gpu.py tests/test_fulldepth_gpu.py tests/test_mpt_gpu.py tests/test_infer_driver.py -q 2>&1 | tail -8 ) > gpurun_out/r03_run42_tests.txt; cat gpurun_out/r03_run42_tests.txt
echo "== decode"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1" | tee gpurun_out/r03_decode_gemv_dma.txt
timeout 300 python scripts/bench_decode.py bf16 2>&1 | grep "decode" | grep "B=1" | tee -a gpurun_out/r03_decode_gemv_dma.txt
timeout 900 python bench.py --stages generate --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_generate_v2.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_generate_v2.json'));print('generate:',d['ms_per_step'],d['value'])"
