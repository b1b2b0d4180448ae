#!/bin/bash
# round 4, run 2: new tests (trainer bookkeeping, decode auto fusion, VQ certificate with the final thresholds), the default bench line with
# its new legs (wall time of the whole default invocation), bf16 16x16x32 probe
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_vqvae_gpu.py tests/test_train_gpu.py tests/test_llama_gpu.py tests/test_gemv_dma_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r04/run2_tests.txt
( time timeout 900 python bench.py --steps 5 --warmup 2 ) > gpurun_out/r04/run2_bench.txt 2>&1
( timeout 120 scripts/probes/mx_probe rates 2>&1 | tail -8 ) > gpurun_out/r04/mx_probe_rates2.txt
tail -5 gpurun_out/r04/run2_tests.txt; tail -8 gpurun_out/r04/run2_bench.txt | cut -c1-3000; tail -8 gpurun_out/r04/mx_probe_rates2.txt
