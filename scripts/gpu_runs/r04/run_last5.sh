#!/bin/bash
# Round 4: the GPU test files not re-run since GemmParams grew (every kernel taking it by value was recompiled): together with
# run_last2 / run_last3 this is the whole -m gpu suite on the tree as committed.
mkdir -p gpurun_out/r04
( timeout 235 python -m pytest tests/test_abi.py tests/test_prior_gpu.py tests/test_vqvae_gpu.py tests/test_lo8_gpu.py tests/test_clap_gpu.py tests/test_fuzz_gpu.py tests/test_fulldepth_gpu.py -q -x -m gpu --deselect tests/test_fulldepth_gpu.py::test_llama7b_32_layers_logits_vs_oracle --deselect tests/test_fulldepth_gpu.py::test_llama7b_64_greedy_tokens_vs_oracle 2>&1 | tail -6 ) > gpurun_out/r04/run_last5.txt 2>&1
cat gpurun_out/r04/run_last5.txt
