#!/bin/bash
# round 4, run 17: gemm256x epilogue stores with the write-through (sc1) / nt policy through buffer descriptors, against plain stores
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04/gemm256x_store_policy.txt
( timeout 300 env LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_sc1.so python -m pytest tests/test_prior_gpu.py -x -q -k gemm256x 2>&1 | tail -1 ) >> gpurun_out/r04/gemm256x_store_policy.txt
for rep in 1 2; do
  for lib in default sc1 nt; do
    if [ $lib = default ]; then unset LLARK_HIP_LIB; else export LLARK_HIP_LIB=$GRAFT_REPO_ROOT/llark_amd/libllark_hip_$lib.so; fi
    echo "== $lib rep $rep" >> gpurun_out/r04/gemm256x_store_policy.txt
    timeout 300 python scripts/bench_gemm256.py 32 2>/dev/null | grep "median" | awk '{print $3, $4, $5, $6, $9, $10, $11}' >> gpurun_out/r04/gemm256x_store_policy.txt
  done
done
cat gpurun_out/r04/gemm256x_store_policy.txt
