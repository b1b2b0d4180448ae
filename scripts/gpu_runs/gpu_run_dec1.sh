#!/bin/bash
# decode step: eager vs host launch-list replay vs hipGraph; generate stage with replay
mkdir -p gpurun_out/dec
timeout 300 python -m pytest tests/test_llama_gpu.py -x -q -m gpu -k "decode_graph" > gpurun_out/dec/tests.log 2>&1; echo "tests exit $?"; grep -v amdgpu.ids gpurun_out/dec/tests.log | tail -5
timeout 600 python scripts/bench_kernels.py decode > gpurun_out/dec/decode.txt 2>&1; grep "decode step" gpurun_out/dec/decode.txt
LLARK_DECODE_REPLAY=1 timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_replay.log 2>&1; echo "gen replay: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b1_replay.log | tr '\n' ' ')"
timeout 600 python bench.py --stages generate --no-cpu-baseline > gpurun_out/dec/gen_b1_eager.log 2>&1; echo "gen eager: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"decode_ms_per_token": [0-9.]*' gpurun_out/dec/gen_b1_eager.log | tr '\n' ' ')"
