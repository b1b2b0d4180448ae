#!/bin/bash
# Round 4, final sanity of the library as last built (FLAC entry points added): smoke + the file-level audio tests + the fused-epilogue parity tests
mkdir -p gpurun_out/r04
{
  timeout 50 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 50 python -m pytest tests/test_extract_gpu.py tests/test_llama_gpu.py -q -x -k "extract or wrapper or rope_qkv_epilogue or checkpoint or batch" 2>&1 | tail -2
} > gpurun_out/r04/run_last6.txt 2>&1
cat gpurun_out/r04/run_last6.txt
