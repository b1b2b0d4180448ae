#!/bin/bash
# round 4, run 14: gemm256x A/B knobs (G256X_R0 = 2, G256X_PRIO = 1, both) as separate libraries, alternating with the default, two rounds
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04/gemm256x_knobs.txt
for rep in 1 2; do
  for lib in default r2 prio both; do
    if [ $lib = default ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_$lib.so; fi
    echo "== $lib rep $rep" >> gpurun_out/r04/gemm256x_knobs.txt
    timeout 300 python scripts/bench_gemm256.py 32 2>/dev/null | grep "median" | awk '{print $3, $4, $5, $6, $9, $10, $11}' >> gpurun_out/r04/gemm256x_knobs.txt
  done
done
cat gpurun_out/r04/gemm256x_knobs.txt
