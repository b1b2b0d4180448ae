mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== full"; timeout 300 python scripts/bench_gemm.py 12 2>&1 | grep -E "fc_qgelu|proj_resid |gate_up|Error|error" | cut -c1-300
for ab in 4 5; do
  echo "== ablate $ab (4 DMA only, 5 DMA only all-L2-hit)"; LLARK_SKIP_CHECK=1 LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_ab$ab.so timeout 300 python scripts/bench_gemm.py 12 2>&1 | grep -E "fc_qgelu|proj_resid |gate_up|Error|error" | cut -c1-300
done
