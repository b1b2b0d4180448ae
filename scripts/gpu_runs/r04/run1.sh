#!/bin/bash
# round 4, run 1: near-tie certificate (tests + statistics + timing), wide full-depth fixture, MFMA shape probe
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_vqvae_gpu.py -x -q -s 2>&1 | tail -40 ) > gpurun_out/r04/run1_vq_tests.txt
( timeout 600 python scripts/gpu_runs/r04/vq_near_tie_stats.py 2>&1 | tail -80 ) > gpurun_out/r04/vq_near_tie_stats.txt
( timeout 900 python -m pytest tests/test_fulldepth_gpu.py -x -q -s -k "jukebox" 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r04/run1_fulldepth.txt
( timeout 120 scripts/probes/mx_probe rates 2>&1 | tail -30 ) > gpurun_out/r04/mx_probe_rates.txt
tail -5 gpurun_out/r04/run1_vq_tests.txt; tail -30 gpurun_out/r04/vq_near_tie_stats.txt; tail -12 gpurun_out/r04/run1_fulldepth.txt; tail -8 gpurun_out/r04/mx_probe_rates.txt
