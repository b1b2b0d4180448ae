mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== full"; timeout 300 python scripts/bench_gemm.py 2 2>&1 | grep -E "fc_qgelu|proj_resid |gate_up"
for ab in 1 2 3; do
  echo "== ablate $ab (1 no DMA, 2 no LDS reads, 3 no MFMA)"; LLARK_SKIP_CHECK=1 LLARK_HIP_LIB=$PWD/llark_amd/libllark_hip_ab$ab.so timeout 300 python scripts/bench_gemm.py 2 2>&1 | grep -E "fc_qgelu|proj_resid |gate_up"
done
