#!/bin/bash
mkdir -p gpurun_out/dec
timeout 200 python -m pytest tests/test_fuzz_gpu.py tests/test_clap_gpu.py -q -m gpu -k "attention or config5 or attn" > gpurun_out/dec/tests_last.log 2>&1; echo "tests exit $?"; grep -v amdgpu.ids gpurun_out/dec/tests_last.log | tail -3
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
